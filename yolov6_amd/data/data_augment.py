"""letterbox on the device (csrc/preproc.hip): the reference's pre-processing step between a decoded frame and the network.

Reference: yolov6/data/data_augment.py:29-58 (`letterbox`: cv2.resize INTER_LINEAR + cv2.copyMakeBorder) and
yolov6/core/inferer.py:162-172 (`Inferer.process_image`: letterbox, HWC -> CHW, BGR -> RGB, uint8 -> fp16 / 255).

`letterbox` keeps the reference's signature and return value `(image, ratio, (left, top))` for a CUDA uint8 HWC tensor;
`process_image` returns the uint8 RGB planes the HIP model's first conv reads directly (`/ 255` is folded into its load, so
`model(process_image(frame, ...)[None])` equals the reference's `model(process_image(...)[0][None])` without the fp16 image
ever existing in HBM) - or, with `as_uint8=False`, the reference's fp16 / fp32 tensor.  The augmentation functions of the
reference's module (mosaic, mixup, HSV, random_affine: CPU data-loader code) are not part of the hot path; under the
overlay (`install_as_yolov6`) they keep coming from the reference checkout.
"""
import ctypes as C

import numpy as np
import torch

from .. import _lib


def _geometry(shape, new_shape, auto, scaleup, stride):
    """data_augment.py:31-56."""
    if isinstance(new_shape, int):
        new_shape = (new_shape, new_shape)
    elif isinstance(new_shape, list) and len(new_shape) == 1:
        new_shape = (new_shape[0], new_shape[0])
    r = min(new_shape[0] / shape[0], new_shape[1] / shape[1])
    if not scaleup:  # only scale down, do not scale up (for better val mAP)
        r = min(r, 1.0)
    new_unpad = int(round(shape[1] * r)), int(round(shape[0] * r))
    dw, dh = new_shape[1] - new_unpad[0], new_shape[0] - new_unpad[1]  # wh padding
    if auto:  # minimum rectangle
        dw, dh = np.mod(dw, stride), np.mod(dh, stride)
    dw /= 2
    dh /= 2
    top, bottom = int(round(dh - 0.1)), int(round(dh + 0.1))
    left, right = int(round(dw - 0.1)), int(round(dw + 0.1))
    return new_unpad, r, top, bottom, left, right


def _run(im, new_unpad, top, bottom, left, right, color, planar, reverse):
    if not (isinstance(im, torch.Tensor) and im.is_cuda and im.dtype == torch.uint8 and im.dim() == 3 and im.shape[2] == 3):
        raise RuntimeError("yolov6_amd: letterbox takes a CUDA uint8 HWC image with 3 channels (there is no CPU path; numpy "
                           "images belong to the reference's own data pipeline)")
    im = im.contiguous()
    H, W = int(im.shape[0]), int(im.shape[1])
    oh, ow = new_unpad[1] + top + bottom, new_unpad[0] + left + right
    out = torch.empty((3, oh, ow) if planar else (oh, ow, 3), dtype=torch.uint8, device=im.device)
    d = _lib.LetterboxDesc()
    d.src, d.H, d.W = C.c_void_p(im.data_ptr()), H, W
    d.dst, d.out_h, d.out_w = C.c_void_p(out.data_ptr()), oh, ow
    d.new_h, d.new_w, d.top, d.left = new_unpad[1], new_unpad[0], top, left
    d.planar, d.reverse_channels = int(planar), int(reverse)
    for c in range(3):
        d.pad[c] = int(color[c])
    with torch.cuda.device(im.device):
        _lib.check(_lib.load().y6_letterbox(C.byref(d), _lib.current_stream_ptr()), "letterbox")
    return out


def letterbox(im, new_shape=(640, 640), color=(114, 114, 114), auto=True, scaleup=True, stride=32):
    '''Resize and pad image while meeting stride-multiple constraints.  im: CUDA uint8 HWC tensor.'''
    shape = tuple(int(v) for v in im.shape[:2])
    new_unpad, r, top, bottom, left, right = _geometry(shape, new_shape, auto, scaleup, stride)
    return _run(im, new_unpad, top, bottom, left, right, color, planar=False, reverse=False), r, (left, top)


def process_image(img_src, img_size, stride, half=True, as_uint8=True):
    '''Inferer.process_image on the device.  -> (image [3, H, W], img_src).  as_uint8: RGB uint8 planes for the HIP model's
    uint8 image conv (default); else the reference's fp16 (half) / fp32 tensor in 0..1.'''
    shape = tuple(int(v) for v in img_src.shape[:2])
    new_unpad, r, top, bottom, left, right = _geometry(shape, img_size, True, True, stride)
    planes = _run(img_src, new_unpad, top, bottom, left, right, (114, 114, 114), planar=True, reverse=True)
    if as_uint8:
        return planes, img_src
    image = planes.half() if half else planes.float()
    image /= 255
    return image, img_src
