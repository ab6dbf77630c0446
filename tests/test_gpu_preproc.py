"""GPU: letterbox on the device (csrc/preproc.hip, yolov6_amd/data/data_augment.py) against the CPU oracle's restatement of
cv2.resize(INTER_LINEAR) + copyMakeBorder (oracle/letterbox_oracle.py - PARITY UNPINNED: opencv is not installed here) -
bit for bit (integer arithmetic) - and the whole chain frame -> uint8 RGB planes -> HIP model against the reference-shaped
fp16 `/ 255` input."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("shape", [(480, 640), (1080, 1920), (375, 500), (640, 640), (1280, 960), (333, 1000), (97, 61), (1280, 1280)])
@pytest.mark.parametrize("new_shape,auto,scaleup", [((640, 640), True, True), (640, False, True), ((1280, 1280), True, False), ([416], True, True)])
def test_letterbox_equals_oracle(shape, new_shape, auto, scaleup):
    from oracle import letterbox_oracle as LO
    from yolov6_amd.data.data_augment import letterbox
    rng = np.random.default_rng(11)
    im = rng.integers(0, 256, shape + (3,), dtype=np.uint8)
    exp, r_exp, off_exp = LO.letterbox(im, new_shape, (114, 114, 114), auto, scaleup, 32)
    got, r, off = letterbox(torch.from_numpy(im).to(DEV), new_shape, (114, 114, 114), auto, scaleup, 32)
    torch.cuda.synchronize()
    assert r == r_exp and tuple(off) == tuple(off_exp)
    assert tuple(got.shape) == exp.shape and got.dtype == torch.uint8
    assert np.array_equal(got.cpu().numpy(), exp), f"{int((got.cpu().numpy() != exp).sum())} bytes differ"


def test_letterbox_colour_and_smooth_image():
    """A smooth image (where bilinear weights matter in every pixel), a non-grey border colour per channel."""
    from oracle import letterbox_oracle as LO
    from yolov6_amd.data.data_augment import letterbox
    yy, xx = np.mgrid[0:300, 0:523]
    im = np.stack([(xx * 255 // 522), (yy * 255 // 299), ((xx + yy) % 256)], -1).astype(np.uint8)
    exp, _, _ = LO.letterbox(im, (640, 640), (10, 200, 77), True, True, 32)
    got, _, _ = letterbox(torch.from_numpy(im).to(DEV), (640, 640), (10, 200, 77), True, True, 32)
    assert np.array_equal(got.cpu().numpy(), exp)


def test_process_image_planes_feed_the_uint8_image_conv():
    """Inferer.process_image on the device: uint8 RGB planes == the oracle's (letterbox, HWC -> CHW, BGR -> RGB) bytes, the fp16
    form == the reference-shaped `image.half() / 255`, and the HIP model gives the same detections from either (to the fp32 summation order of two separately tuned plans)."""
    from oracle import letterbox_oracle as LO, synth
    from yolov6_amd.configs import tiny_config
    from yolov6_amd.data.data_augment import process_image
    from yolov6_amd.models.yolo import build_model
    rng = np.random.default_rng(12)
    frame = rng.integers(0, 256, (270, 480, 3), dtype=np.uint8)
    ref = LO.process_image(frame, 256, 32, half=True)                       # [3, H, W] fp16 in 0..1
    planes, src = process_image(torch.from_numpy(frame).to(DEV), 256, 32)
    assert planes.dtype == torch.uint8 and tuple(planes.shape) == tuple(ref.shape)
    assert torch.equal((planes.cpu().half() / 255), ref)
    half, _ = process_image(torch.from_numpy(frame).to(DEV), 256, 32, half=True, as_uint8=False)
    assert half.dtype == torch.float16 and torch.equal(half.cpu(), ref)
    cfg = tiny_config(width=0.25, depth=0.33)
    model = build_model(cfg, 80, "cpu").eval()
    model.load_state_dict(synth.synth_state_dict(model.state_dict(), seed=0))
    model = model.to(DEV).half()
    # (shape-derived kernel tables: two plans that are each autotuned on their own may pick different conv variants - different fp32
    # summation orders, up to 3e-3 apart on this model, which made this test fail one run in seven in round 5; with the same
    # kernels the two input forms differ by nothing but the stem's load path)
    det_u8 = model.compile(planes[None].contiguous(), autotune=False).run().clone()
    det_f16 = model.compile(half[None].contiguous(), autotune=False).run().clone()
    torch.cuda.synchronize()
    err = ((det_u8 - det_f16).abs() / det_f16.abs().clamp(min=1.0)).max()
    assert float(err) < 2e-3, float(err)
