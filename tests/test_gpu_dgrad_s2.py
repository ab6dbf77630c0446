"""GPU: csrc/dgrad_s2.hip - the data gradient of a 3x3 stride-2 conv (+ the 1x1 stride-2 conv of the same input) from the compact
output gradients, against torch-CPU fp32 `conv_transpose2d` on the same fp16-rounded operands (= autograd's input gradient of
F.conv2d(stride=2), reference yolov6/layers/common.py:250-255 under core/engine.py:173).

Tolerance: fp16 operands, fp32 accumulation over up to 10 * M products, one fp16 rounding of the result (two when accumulating):
3e-3 of the tensor's max - the bar of the stride-1 data-gradient convs in tests/train_replay.py."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

from yolov6_amd import _lib
from yolov6_amd.engine import TRef, _null_tensor

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _nhwc(t_nchw, cstride=None, coff=0):
    B, Cn, H, W = t_nchw.shape
    cs = cstride or Cn
    buf = torch.zeros((B, H, W, cs), dtype=torch.float16, device=DEV)
    buf[..., coff:coff + Cn] = t_nchw.permute(0, 2, 3, 1).contiguous().half().to(DEV)
    return TRef(buf, B, H, W, Cn, cs, coff)


def _back(ref):
    return ref.to_nhwc_tensor().float().cpu().permute(0, 3, 1, 2).contiguous()


def _pack_dgrad(w):
    """The data-gradient image of an OIHW fp32 weight (y6_pack_job kind 1), as the training plan's per-step packing makes it."""
    lib = _lib.load()
    Cout, Cin, K, _ = w.shape
    wd = w.float().contiguous().to(DEV)
    n = int(lib.y6_pack_job_elems(1, Cout, Cin, K))
    dst = torch.empty(n, dtype=torch.float16, device=DEV)
    jobs = (_lib.PackJob * 1)()
    jobs[0].src, jobs[0].dst, jobs[0].kind, jobs[0].Cout, jobs[0].Cin, jobs[0].K, jobs[0].first = wd.data_ptr(), dst.data_ptr(), 1, Cout, Cin, K, 0
    tab = torch.frombuffer(bytearray(bytes(jobs)), dtype=torch.uint8).to(DEV)
    d = _lib.PackBatchDesc()
    d.jobs, d.njobs, d.total = tab.data_ptr(), 1, n
    _lib.check(lib.y6_pack_weights_batched(C.byref(d), _lib.current_stream_ptr()), "pack_weights_batched")
    torch.cuda.synchronize()
    return dst, wd


CASES = [
    # B, M (conv couts), N (conv cins), Ho, Wo, 1x1 branch, accumulate
    (2, 64, 32, 16, 16, True, False),       # the 32 -> 64 block: one cout fragment, two pixel fragments per wave
    (2, 64, 32, 13, 19, True, True),        # ragged tiles, accumulate
    (2, 128, 64, 20, 20, True, False),      # two cout fragments per block
    (1, 128, 64, 9, 40, False, True),       # no 1x1 branch (ConvBNReLU k3 s2), wide tile
    (2, 256, 128, 10, 10, True, True),      # two cout blocks per tile
    (1, 512, 256, 5, 7, True, False),       # sixteen chunks, four cout blocks
    (1, 96, 96, 6, 6, False, False),        # 96 channels: one fragment per block, three blocks
]


@pytest.mark.parametrize("B,M,N,Ho,Wo,has1,acc", CASES)
def test_dgrad_s2_matches_conv_transpose(B, M, N, Ho, Wo, has1, acc):
    lib = _lib.load()
    g = torch.Generator().manual_seed(B * 1000 + M + N + Ho * 7 + Wo)
    dy3 = torch.randn((B, M, Ho, Wo), generator=g).half().float()
    dy1 = torch.randn((B, M, Ho, Wo), generator=g).half().float()
    w3 = (torch.randn((M, N, 3, 3), generator=g) * (1.0 / (9 * M) ** 0.5))
    w1 = (torch.randn((M, N, 1, 1), generator=g) * (1.0 / M ** 0.5))
    old = torch.randn((B, N, 2 * Ho, 2 * Wo), generator=g).half().float()
    ref = F.conv_transpose2d(dy3, w3.half().float(), None, stride=2, padding=1, output_padding=1)
    if has1:
        ref = ref + F.conv_transpose2d(dy1, w1.half().float(), None, stride=2, padding=0, output_padding=1)
    assert ref.shape == old.shape
    if acc:
        ref = ref.half().float() + old
    # views into wider buffers: the kernel must respect channel strides / offsets of all three tensors
    t3 = _nhwc(dy3, cstride=M + 16, coff=8)
    t1 = _nhwc(dy1)
    tx = _nhwc(old, cstride=N + 32, coff=16)
    p3, keep3 = _pack_dgrad(w3)
    p1, keep1 = _pack_dgrad(w1)
    d = _lib.DgradS2Desc()
    d.dy3 = t3.ct()
    d.dy1 = t1.ct() if has1 else _null_tensor()
    d.dx = tx.ct()
    d.w3_packed = p3.data_ptr()
    d.w1_packed = p1.data_ptr() if has1 else None
    d.accumulate = int(acc)
    assert lib.y6_dgrad_s2_supported(C.byref(d)) == 1
    _lib.check(lib.y6_dgrad_s2(C.byref(d), _lib.current_stream_ptr()), "dgrad_s2")
    torch.cuda.synchronize()
    got = _back(tx)
    err = float((got - ref).abs().max()) / float(ref.abs().max())
    assert err < 3e-3, err
    # nothing outside the view was touched
    assert float(tx.buf[..., :16].abs().max()) == 0.0 and float(tx.buf[..., 16 + N:].abs().max()) == 0.0


def test_dgrad_s2_refuses_what_it_cannot_do():
    lib = _lib.load()
    dy = _nhwc(torch.zeros(1, 48, 4, 4))
    dx = _nhwc(torch.zeros(1, 32, 8, 8))
    w = torch.zeros(1 << 16, dtype=torch.float16, device=DEV)
    d = _lib.DgradS2Desc()
    d.dy3, d.dx, d.w3_packed = dy.ct(), dx.ct(), w.data_ptr()
    d.dy1 = _null_tensor()
    assert lib.y6_dgrad_s2_supported(C.byref(d)) == 0            # 48 channels
    with pytest.raises(RuntimeError):
        _lib.check(lib.y6_dgrad_s2(C.byref(d), _lib.current_stream_ptr()), "dgrad_s2")
    dy = _nhwc(torch.zeros(1, 64, 4, 4))
    dx = _nhwc(torch.zeros(1, 32, 8, 9))
    d.dy3, d.dx = dy.ct(), dx.ct()
    assert lib.y6_dgrad_s2_supported(C.byref(d)) == 0            # dx is not [2*Ho, 2*Wo]


@pytest.mark.parametrize("M,N,Ho", [(64, 32, 160), (128, 64, 80), (256, 128, 40), (512, 256, 20)])
def test_dgrad_s2_is_the_adjoint_of_the_forward_convs_at_the_training_batch(M, N, Ho):
    """BASELINE configs[2] sizes (YOLOv6-S b64, the four stride-2 RepVGG blocks of the backbone), where an element-wise CPU reference
    of dx would take minutes: the size-independent property instead.  dx = conv3^T(dy3) + conv1^T(dy1) is the adjoint of the block's
    two convs, so for any x:  <dx, x> = <dy3, conv3x3_s2(x)> + <dy1, conv1x1_s2(x)>  (fp64 inner products; the right side from
    torch-CPU fp32 convs of the same fp16-rounded operands).  dx is stored in fp16: independent roundings of 26-105 M elements
    leave the inner product well inside 1e-3 of its scale."""
    lib = _lib.load()
    B = 64
    g = torch.Generator().manual_seed(M + Ho)
    dy3 = torch.randn((B, M, Ho, Ho), generator=g).half()
    dy1 = torch.randn((B, M, Ho, Ho), generator=g).half()
    x = torch.randn((B, N, 2 * Ho, 2 * Ho), generator=g).half()
    w3 = (torch.randn((M, N, 3, 3), generator=g) * (1.0 / (9 * M) ** 0.5)).half().float()
    w1 = (torch.randn((M, N, 1, 1), generator=g) * (1.0 / M ** 0.5)).half().float()
    with torch.no_grad():
        y3 = F.conv2d(x.float(), w3, None, stride=2, padding=1)
        y1 = F.conv2d(x.float(), w1, None, stride=2, padding=0)
    rhs = float((dy3.double() * y3.double()).sum() + (dy1.double() * y1.double()).sum())
    scale = float((dy3.double().pow(2).sum() * y3.double().pow(2).sum()).sqrt() + (dy1.double().pow(2).sum() * y1.double().pow(2).sum()).sqrt())
    del y3, y1
    t3 = TRef(dy3.permute(0, 2, 3, 1).contiguous().to(DEV), B, Ho, Ho, M, M, 0)
    t1 = TRef(dy1.permute(0, 2, 3, 1).contiguous().to(DEV), B, Ho, Ho, M, M, 0)
    dxb = torch.empty((B, 2 * Ho, 2 * Ho, N), dtype=torch.float16, device=DEV)
    tx = TRef(dxb, B, 2 * Ho, 2 * Ho, N, N, 0)
    p3, k3 = _pack_dgrad(w3)
    p1, k1 = _pack_dgrad(w1)
    d = _lib.DgradS2Desc()
    d.dy3, d.dy1, d.dx = t3.ct(), t1.ct(), tx.ct()
    d.w3_packed, d.w1_packed, d.accumulate = p3.data_ptr(), p1.data_ptr(), 0
    _lib.check(lib.y6_dgrad_s2(C.byref(d), _lib.current_stream_ptr()), "dgrad_s2")
    torch.cuda.synchronize()
    lhs = float((dxb.double() * x.permute(0, 2, 3, 1).contiguous().to(DEV).double()).sum())
    assert abs(lhs - rhs) < 1e-3 * scale, (lhs, rhs, scale)
    # and the accumulating form adds exactly one more copy (a second launch over what the first left: dx + dx, rounded once more)
    d.accumulate = 1
    first = dxb.clone()
    _lib.check(lib.y6_dgrad_s2(C.byref(d), _lib.current_stream_ptr()), "dgrad_s2")
    torch.cuda.synchronize()
    assert torch.equal(dxb, (first.float() + first.float()).half())
