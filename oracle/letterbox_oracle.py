"""ORACLE (test infrastructure, CPU, numpy) - never imported by the product path (yolov6_amd/).

Restates `letterbox()` of the reference, yolov6/data/data_augment.py:29-58, and the layout half of
`Inferer.process_image`, yolov6/core/inferer.py:162-172.

PARITY UNPINNED.  letterbox() is cv2.resize(INTER_LINEAR) + cv2.copyMakeBorder; opencv-python (requirements.txt:7,
`opencv-python>=4.1.2`, un-pinned) is neither vendored by the reference nor installed in this environment, and the reference
holds no vectors for it.  `resize_linear_u8` restates the published algorithm of opencv's imgproc/src/resize.cpp for 8-bit
images (resizeGeneric_ / HResizeLinear / VResizeLinear with INTER_RESIZE_COEF_BITS = 11, and the `iscale == 2` shortcut to the
INTER_AREA fast path); the host logic around it (scale ratio, padding split, `auto` stride modulus) is the reference's own
code restated line by line and needs no cv2.
"""
import numpy as np


def _cv_round_short(v):
    """saturate_cast<short>(float): cvRound (round half to even), saturated."""
    return np.clip(np.rint(v), -32768, 32767).astype(np.int32)


def resize_linear_u8(im, new_w, new_h):
    """cv2.resize(im, (new_w, new_h), interpolation=cv2.INTER_LINEAR) for a uint8 HWC image (resize.cpp)."""
    H, W = im.shape[:2]
    if (new_w, new_h) == (W, H):
        return im.copy()
    inv_x, inv_y = new_w / W, new_h / H            # double(dsize) / ssize
    scale_x, scale_y = 1.0 / inv_x, 1.0 / inv_y
    if 2 * new_w == W and 2 * new_h == H:          # is_area_fast, iscale_x == iscale_y == 2: INTER_AREA's fast path
        s = im.astype(np.int32)
        return ((s[0::2, 0::2] + s[0::2, 1::2] + s[1::2, 0::2] + s[1::2, 1::2] + 2) >> 2).astype(np.uint8)
    dx = np.arange(new_w, dtype=np.float64)
    fx = ((dx + 0.5) * scale_x - 0.5).astype(np.float32)
    sx = np.floor(fx).astype(np.int32)
    fx = fx - sx.astype(np.float32)
    lo, hi = sx < 0, sx >= W - 1
    fx = np.where(lo | hi, np.float32(0), fx).astype(np.float32)
    sx = np.where(lo, 0, np.where(hi, W - 1, sx))
    a0, a1 = _cv_round_short((np.float32(1) - fx) * np.float32(2048)), _cv_round_short(fx * np.float32(2048))
    dy = np.arange(new_h, dtype=np.float64)
    fy = ((dy + 0.5) * scale_y - 0.5).astype(np.float32)
    sy = np.floor(fy).astype(np.int32)
    fy = (fy - sy.astype(np.float32)).astype(np.float32)
    b0, b1 = _cv_round_short((np.float32(1) - fy) * np.float32(2048)), _cv_round_short(fy * np.float32(2048))
    y0, y1 = np.clip(sy, 0, H - 1), np.clip(sy + 1, 0, H - 1)
    x1 = np.minimum(sx + 1, W - 1)
    s = im.astype(np.int32)
    # horizontal pass (int32, scale 2048): every source row that is needed
    hrow = s[:, sx] * a0[None, :, None] + s[:, x1] * a1[None, :, None]            # [H, new_w, C]
    r0, r1 = hrow[y0], hrow[y1]                                                    # [new_h, new_w, C]
    out = (((b0[:, None, None] * (r0 >> 4)) >> 16) + ((b1[:, None, None] * (r1 >> 4)) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)


def letterbox_geometry(shape, new_shape=(640, 640), auto=True, scaleup=True, stride=32):
    """data_augment.py:31-56 without the pixels: -> (new_unpad (w, h), r, top, bottom, left, right)."""
    if isinstance(new_shape, int):
        new_shape = (new_shape, new_shape)
    elif isinstance(new_shape, list) and len(new_shape) == 1:
        new_shape = (new_shape[0], new_shape[0])
    r = min(new_shape[0] / shape[0], new_shape[1] / shape[1])
    if not scaleup:
        r = min(r, 1.0)
    new_unpad = int(round(shape[1] * r)), int(round(shape[0] * r))
    dw, dh = new_shape[1] - new_unpad[0], new_shape[0] - new_unpad[1]
    if auto:
        dw, dh = np.mod(dw, stride), np.mod(dh, stride)
    dw /= 2
    dh /= 2
    top, bottom = int(round(dh - 0.1)), int(round(dh + 0.1))
    left, right = int(round(dw - 0.1)), int(round(dw + 0.1))
    return new_unpad, r, top, bottom, left, right


def letterbox(im, new_shape=(640, 640), color=(114, 114, 114), auto=True, scaleup=True, stride=32):
    """-> (padded uint8 HWC image, r, (left, top))."""
    shape = im.shape[:2]
    new_unpad, r, top, bottom, left, right = letterbox_geometry(shape, new_shape, auto, scaleup, stride)
    if shape[::-1] != new_unpad:
        im = resize_linear_u8(im, new_unpad[0], new_unpad[1])
    out = np.empty((im.shape[0] + top + bottom, im.shape[1] + left + right, 3), np.uint8)
    out[...] = np.asarray(color, np.uint8)
    out[top:top + im.shape[0], left:left + im.shape[1]] = im
    return out, r, (left, top)


def process_image(img_src, img_size, stride, half=True):
    """Inferer.process_image (inferer.py:162-172): letterbox, HWC -> CHW, BGR -> RGB, /255 (fp16 when `half`)."""
    import torch
    image = letterbox(img_src, img_size, stride=stride)[0]
    image = image.transpose((2, 0, 1))[::-1]
    image = torch.from_numpy(np.ascontiguousarray(image))
    image = image.half() if half else image.float()
    image /= 255
    return image
