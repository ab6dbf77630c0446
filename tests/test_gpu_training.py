"""GPU: the training step (BASELINE configs[2]) - kernels one by one against torch-CPU fp32 autograd of the same op, then the
whole training graph (train-form forward with batch statistics, backward, parameter gradients, running statistics) against
oracle/model_oracle.py::TrainOracle, which is pinned to the unmodified reference's train-mode goldens
(tests/golden/train_*.npz, incl. reference parameter gradients), and the loss gradient against
oracle/loss_grad_oracle.py, pinned to the reference's autograd (tests/golden/lossgrad_*.npz).

Tolerances: weight-gradient / BatchNorm / pooling kernels see exact fp16 inputs and accumulate in fp32 / double ->
1e-3 of the tensor's max.  Whole-model gradients pass ~40 fp16-stored layers forward and backward: stated per tensor as
max|g - g_ref| / max|g_ref| against the fp32 oracle, bound measured on MI355X + 20 % (see the test)."""
import ctypes as C
import json
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from yolov6_amd import _lib
from yolov6_amd.engine import TRef

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _stream():
    return _lib.current_stream_ptr()


def _nhwc(t_nchw):
    """CPU NCHW fp32 -> GPU NHWC fp16 buffer + TRef."""
    B, Cn, H, W = t_nchw.shape
    buf = t_nchw.permute(0, 2, 3, 1).contiguous().half().to(DEV)
    return TRef(buf, B, H, W, Cn, Cn, 0)


def _back(ref):
    return ref.to_nhwc_tensor().float().cpu().permute(0, 3, 1, 2).contiguous()


def _rup(v, m):
    return (v + m - 1) // m * m


def _transpose(src: TRef, sy, sx, oy, ox, R, Q, nchw=None):
    lib = _lib.load()
    Cn, B = (nchw.shape[1], nchw.shape[0]) if nchw is not None else (src.C, src.B)
    dst = torch.empty(Cn * B * R * Q, dtype=torch.float16, device=DEV)
    d = _lib.WgradTDesc()
    if nchw is not None:
        Bn, Cc, Hn, Wn = nchw.shape
        d.src = _lib.Tensor(C.c_void_p(nchw.data_ptr()), Bn, Hn, Wn, Cc, Cc, 0)
        d.nchw, d.src_dtype = 1, (_lib.Y6_F16 if nchw.dtype == torch.float16 else _lib.Y6_F32)
    else:
        d.src = src.ct()
    d.sy, d.sx, d.oy, d.ox, d.R, d.Q = sy, sx, oy, ox, R, Q
    d.dst = dst.data_ptr()
    _lib.check(lib.y6_wgrad_transpose(C.byref(d), _stream()), "wgrad_transpose")
    return dst


def _wgrad(mode, a, planes, M, N, B, Q, rows, T):
    lib = _lib.load()
    out = torch.zeros(M * N * T, dtype=torch.float32, device=DEV)
    w = _lib.WgradDesc()
    w.mode = mode
    w.a = a.data_ptr()
    w.M, w.N, w.B, w.Q, w.rows, w.a_rows = M, N, B, Q, rows, rows
    w.a_channels, w.plane_channels = M, N
    for i, (t, prow, drow) in enumerate(planes):
        w.plane[i], w.plane_rows[i], w.drow[i] = t.data_ptr(), prow, drow
    w.out = out.data_ptr()
    w.sm, w.sn, w.st = N * T, T, 1
    ws = torch.empty(32 << 20, dtype=torch.uint8, device=DEV)
    w.workspace, w.workspace_bytes = ws.data_ptr(), ws.numel()
    _lib.check(lib.y6_wgrad(C.byref(w), _stream()), "wgrad")
    torch.cuda.synchronize()
    return out.cpu()


def test_transpose_sampling_exact():
    g = torch.Generator().manual_seed(0)
    x = torch.rand((2, 24, 7, 11), generator=g).half().float()
    xr = _nhwc(x)
    for (sy, sx, oy, ox, R, Q) in [(1, 1, -1, 0, 9, 16), (1, 1, 0, 0, 7, 16), (2, 2, 0, 1, 4, 16), (2, 2, -1, 0, 5, 16)]:
        t = _transpose(xr, sy, sx, oy, ox, R, Q).float().cpu().view(2, R, Q // 8, 24, 8).permute(3, 0, 1, 2, 4).reshape(24, 2, R, Q)
        ref = torch.zeros(24, 2, R, Q)
        for r in range(R):
            for q in range(Q):
                y, xx = r * sy + oy, q * sx + ox
                if 0 <= y < 7 and 0 <= xx < 11:
                    ref[:, :, r, q] = x[:, :, y, xx].t()
        assert torch.equal(t, ref), (sy, sx, oy, ox)
    img = torch.rand((2, 3, 8, 12), generator=g)
    t = _transpose(None, 2, 2, -1, 1, 5, 16, nchw=img.to(DEV)).float().cpu().view(2, 5, 2, 3, 8).permute(3, 0, 1, 2, 4).reshape(3, 2, 5, 16)
    ref = torch.zeros(3, 2, 5, 16)
    for r in range(5):
        for q in range(16):
            y, xx = 2 * r - 1, 2 * q + 1
            if 0 <= y < 8 and 0 <= xx < 12:
                ref[:, :, r, q] = img[:, :, y, xx].half().float().t()
    assert torch.equal(t, ref)


@pytest.mark.parametrize("cin,cout,B,H,W", [(16, 24, 3, 13, 17), (64, 64, 2, 20, 20), (40, 80, 2, 9, 33), (128, 32, 1, 40, 40)])
def test_wgrad_3x3_s1_and_1x1(cin, cout, B, H, W):
    g = torch.Generator().manual_seed(cin + cout)
    x = (torch.rand((B, cin, H, W), generator=g) - 0.5).half().float()
    dy = (torch.rand((B, cout, H, W), generator=g) - 0.5).half().float()
    Q = _rup(W, 16)
    a = _transpose(_nhwc(dy), 1, 1, 0, 0, H, Q)
    xr = _nhwc(x)
    p = _transpose(xr, 1, 1, -1, 0, H + 2, Q)
    got3 = _wgrad(_lib.WG_3X3S1, a, [(p, H + 2, 0), (p, H + 2, 1), (p, H + 2, 2)], cout, cin, B, Q, H, 9).view(cout, cin, 3, 3)
    ref3 = torch.nn.grad.conv2d_weight(x, (cout, cin, 3, 3), dy, stride=1, padding=1)
    assert float((got3 - ref3).abs().max()) <= 1e-3 * float(ref3.abs().max())
    # transpose-detecting: the centre tap differs from every other tap
    got1 = _wgrad(_lib.WG_1X1, a, [(p, H + 2, 1)], cout, cin, B, Q, H, 1).view(cout, cin, 1, 1)
    ref1 = torch.nn.grad.conv2d_weight(x, (cout, cin, 1, 1), dy, stride=1, padding=0)
    assert float((got1 - ref1).abs().max()) <= 1e-3 * float(ref1.abs().max())


def _wgrad_nhwc(ksize, dyv: TRef, xv: TRef, M, N):
    lib = _lib.load()
    T = ksize * ksize
    out = torch.zeros(M * N * T, dtype=torch.float32, device=DEV)
    w = _lib.WgradNhwcDesc()
    w.ksize, w.dy, w.x, w.M, w.N = ksize, dyv.ct(), xv.ct(), M, N
    w.out = out.data_ptr()
    w.sm, w.sn, w.st = N * T, T, 1
    ws = torch.empty(64 << 20, dtype=torch.uint8, device=DEV)
    w.workspace, w.workspace_bytes = ws.data_ptr(), ws.numel()
    assert lib.y6_wgrad_nhwc_supported(C.byref(w)) == 1
    _lib.check(lib.y6_wgrad_nhwc(C.byref(w), _stream()), "wgrad_nhwc")
    torch.cuda.synchronize()
    return out.cpu()


@pytest.mark.parametrize("kernel", ["flat", "ring"])
@pytest.mark.parametrize("cin,cout,B,H,W", [(16, 24, 3, 13, 17), (64, 64, 2, 20, 20), (40, 80, 2, 9, 33), (128, 32, 1, 40, 40),
                                            (64, 64, 3, 37, 160), (256, 128, 5, 20, 20), (32, 96, 2, 7, 80), (64, 68, 2, 11, 40),
                                            (128, 256, 16, 40, 40), (72, 136, 9, 20, 20), (64, 192, 3, 80, 80)])
def test_wgrad_nhwc_3x3_s1_and_1x1(cin, cout, B, H, W, kernel, monkeypatch):
    """The NHWC-fed weight gradients against torch's conv2d_weight - `flat`: the flat-index block-tiled kernel (csrc/wgrad_flat.hip:
    LDS-DMA chunks of the whole batch's padded pixel sequence + ds_read_b64_tr_b16; the 160-wide case does not fit its stages and
    takes the row ring), `ring`: the row-ring kernel (csrc/wgrad.hip, Y6_WGRAD_FLAT=0).  Ragged widths (pad cells must read as
    zeros), chunks and slices that cross row and image borders, channel counts that are not multiples of the 32-channel image /
    the block tile (two cout tiles, ragged last tile), and an 8-channel padded dy view (68 -> 72, zero pad channels) as the
    prediction convs' gradients are stored."""
    monkeypatch.setenv("Y6_WGRAD_FLAT", "1" if kernel == "flat" else "0")
    g = torch.Generator().manual_seed(cin + cout + W)
    x = (torch.rand((B, cin, H, W), generator=g) - 0.5).half().float()
    dy = (torch.rand((B, cout, H, W), generator=g) - 0.5).half().float()
    cpad = _rup(cout, 8)
    dyp = torch.zeros((B, cpad, H, W))
    dyp[:, :cout] = dy
    # both operands as channel slices of wider buffers (concat slots): cstride > C, coff > 0
    xbuf = torch.full((B, H, W, cin + 16), float("nan"), dtype=torch.float16, device=DEV)
    xbuf[..., 8:8 + cin] = x.permute(0, 2, 3, 1).to(DEV).half()
    xv = TRef(xbuf, B, H, W, cin, cin + 16, 8)
    dbuf = torch.full((B, H, W, cpad + 8), float("nan"), dtype=torch.float16, device=DEV)
    dbuf[..., :cpad] = dyp.permute(0, 2, 3, 1).to(DEV).half()
    dyv = TRef(dbuf, B, H, W, cpad, cpad + 8, 0)
    got3 = _wgrad_nhwc(3, dyv, xv, cout, cin).view(cout, cin, 3, 3)
    ref3 = torch.nn.grad.conv2d_weight(x, (cout, cin, 3, 3), dy, stride=1, padding=1)
    err3 = float((got3 - ref3).abs().max()) / float(ref3.abs().max())
    got1 = _wgrad_nhwc(1, dyv, xv, cout, cin).view(cout, cin, 1, 1)
    ref1 = torch.nn.grad.conv2d_weight(x, (cout, cin, 1, 1), dy, stride=1, padding=0)
    err1 = float((got1 - ref1).abs().max()) / float(ref1.abs().max())
    print(f"wgrad_nhwc {cin}->{cout} b{B} {H}x{W}: 3x3 {err3:.2e}  1x1 {err1:.2e}")
    if err3 > 1e-3:      # which taps are off (transpose / shift errors show as whole taps)
        per_tap = (got3 - ref3).abs().amax(dim=(0, 1)) / float(ref3.abs().max())
        print("per-tap error:", per_tap)
    assert err3 <= 1e-3 and err1 <= 1e-3
    # deterministic: slices are summed in a fixed order
    assert torch.equal(_wgrad_nhwc(3, dyv, xv, cout, cin), got3.reshape(-1))


@pytest.mark.parametrize("dilated", [False, True])
@pytest.mark.parametrize("cin,cout,B,H,W", [(16, 32, 2, 12, 20), (32, 64, 3, 16, 16), (64, 128, 4, 80, 80), (128, 256, 9, 40, 40),
                                            (72, 40, 5, 26, 44), (64, 64, 2, 160, 160), (64, 128, 2, 160, 160)])
def test_wgrad_nhwc_stride2(cin, cout, B, H, W, dilated):
    """Stride-2 weight gradients (3x3 pad 1 and 1x1) of the flat-index kernel (csrc/wgrad_flat.hip: four parity-plane images per
    stage) straight from NHWC, with dy compact [B, H/2, W/2, M] or zero-inserted at (2y, 2x) of an [B, H, W, M] buffer (how the
    training graph stores the gradient of a stride-2 conv for its data gradient), against torch's conv2d_weight."""
    g = torch.Generator().manual_seed(cin + cout + W)
    Ho, Wo = H // 2, W // 2
    x = (torch.rand((B, cin, H, W), generator=g) - 0.5).half().float()
    dy = (torch.rand((B, cout, Ho, Wo), generator=g) - 0.5).half().float()
    xbuf = torch.full((B, H, W, cin + 8), float("nan"), dtype=torch.float16, device=DEV)
    xbuf[..., 8:] = x.permute(0, 2, 3, 1).to(DEV).half()
    xv = TRef(xbuf, B, H, W, cin, cin + 8, 8)
    if dilated:
        dbuf = torch.zeros((B, H, W, cout), dtype=torch.float16, device=DEV)
        dbuf[:, ::2, ::2] = dy.permute(0, 2, 3, 1).to(DEV).half()
        dyv = TRef(dbuf, B, H, W, cout, cout, 0)
    else:
        dbuf = dy.permute(0, 2, 3, 1).contiguous().to(DEV).half()
        dyv = TRef(dbuf, B, Ho, Wo, cout, cout, 0)
    lib = _lib.load()
    for K in (3, 1):
        T = K * K
        out = torch.zeros(cout * cin * T, dtype=torch.float32, device=DEV)
        w = _lib.WgradNhwcDesc()
        w.ksize, w.dy, w.x, w.M, w.N, w.stride = K, dyv.ct(), xv.ct(), cout, cin, 2
        w.out = out.data_ptr()
        w.sm, w.sn, w.st = cin * T, T, 1
        ws = torch.empty(128 << 20, dtype=torch.uint8, device=DEV)
        w.workspace, w.workspace_bytes = ws.data_ptr(), ws.numel()
        if lib.y6_wgrad_nhwc_supported(C.byref(w)) == 0:
            # the four plane images of a wide output (80 columns and 128 couts' worth of dy images) do not fit the 80 KiB stage:
            # the engine keeps the plane-fed kernel for such a conv
            assert W >= 160 and K == 3
            continue
        assert lib.y6_wgrad_nhwc_supported(C.byref(w)) == 1
        _lib.check(lib.y6_wgrad_nhwc(C.byref(w), _stream()), "wgrad_nhwc")
        torch.cuda.synchronize()
        got = out.cpu().view(cout, cin, K, K)
        ref = torch.nn.grad.conv2d_weight(x, (cout, cin, K, K), dy, stride=2, padding=K // 2)
        err = float((got - ref).abs().max()) / float(ref.abs().max())
        print(f"wgrad_nhwc s2 {cin}->{cout} b{B} {H}x{W} k{K} dilated={dilated}: {err:.2e}")
        if err > 1e-3 and K == 3:
            print("per-tap error:", (got - ref).abs().amax(dim=(0, 1)) / float(ref.abs().max()))
        assert err <= 1e-3


@pytest.mark.parametrize("cin,cout,B,H,W", [(16, 32, 2, 12, 20), (32, 64, 3, 16, 16), (3, 16, 2, 24, 40)])
def test_wgrad_stride2(cin, cout, B, H, W):
    g = torch.Generator().manual_seed(7 + cin)
    x = (torch.rand((B, cin, H, W), generator=g) - 0.5).half().float()
    Ho, Wo = H // 2, W // 2
    dy = (torch.rand((B, cout, Ho, Wo), generator=g) - 0.5).half().float()
    Q = _rup(Wo, 16)
    # the backward sees dy zero-inserted at (2y, 2x): sample it back with stride 2
    dil = torch.zeros((B, cout, H, W))
    dil[:, :, ::2, ::2] = dy
    a = _transpose(_nhwc(dil), 2, 2, 0, 0, Ho, Q)
    nchw = x.to(DEV) if cin == 3 else None            # the stem reads the caller's NCHW image
    xr = None if cin == 3 else _nhwc(x)
    pe = [_transpose(xr, 2, 2, 0, cp, Ho, Q, nchw) for cp in (0, 1)]
    po = [_transpose(xr, 2, 2, -1, cp, Ho + 1, Q, nchw) for cp in (0, 1)]
    planes = []
    for ky in range(3):
        for cp in range(2):
            planes.append((pe[cp], Ho, 0) if ky == 1 else (po[cp], Ho + 1, 0 if ky == 0 else 1))
    got = _wgrad(_lib.WG_3X3S2, a, planes, cout, cin, B, Q, Ho, 9).view(cout, cin, 3, 3)
    ref = torch.nn.grad.conv2d_weight(x, (cout, cin, 3, 3), dy, stride=2, padding=1)
    assert float((got - ref).abs().max()) <= 1e-3 * float(ref.abs().max())
    got1 = _wgrad(_lib.WG_1X1, a, [(pe[0], Ho, 0)], cout, cin, B, Q, Ho, 1).view(cout, cin, 1, 1)
    ref1 = torch.nn.grad.conv2d_weight(x, (cout, cin, 1, 1), dy, stride=2, padding=0)
    assert float((got1 - ref1).abs().max()) <= 1e-3 * float(ref1.abs().max())


def test_wgrad_convt():
    g = torch.Generator().manual_seed(3)
    B, cin, cout, H, W = 2, 32, 64, 10, 14
    x = (torch.rand((B, cin, H, W), generator=g) - 0.5).half().float()
    dout = (torch.rand((B, cout, 2 * H, 2 * W), generator=g) - 0.5).half().float()
    Q = _rup(W, 16)
    a = _transpose(_nhwc(x), 1, 1, 0, 0, H, Q)
    dr = _nhwc(dout)
    planes = [(_transpose(dr, 2, 2, sub >> 1, sub & 1, H, Q), H, 0) for sub in range(4)]
    got = _wgrad(_lib.WG_CONVT, a, planes, cin, cout, B, Q, H, 4).view(cin, cout, 2, 2)
    w = torch.zeros((cin, cout, 2, 2), requires_grad=True)
    F.conv_transpose2d(x, w, stride=2).backward(dout)
    assert float((got - w.grad).abs().max()) <= 1e-3 * float(w.grad.abs().max())


@pytest.mark.parametrize("B,Cin,H,W,Cout,with_1x1", [(2, 3, 64, 64, 32, True), (1, 3, 96, 160, 16, True), (1, 3, 32, 704, 64, True),
                                                      (2, 3, 64, 96, 48, False), (1, 1, 64, 64, 32, True), (3, 3, 70, 74, 32, True)])
def test_wgrad_stem_one_pass_vs_autograd(B, Cin, H, W, Cout, with_1x1):
    """csrc/wgrad_stem.hip: the 3x3 s2 and 1x1 s2 weight gradients of the stem block (efficientrep.py:28-41, train form
    common.py:250-255) from the NCHW image and the compact NHWC gradients in one pass; accumulates; two runs agree bit for bit."""
    lib = _lib.load()
    g = torch.Generator().manual_seed(13)
    x = (torch.rand((B, Cin, H, W), generator=g) - 0.3).half()
    Ho, Wo = H // 2, W // 2
    dy3 = (torch.randn((B, Cout, Ho, Wo), generator=g) * 0.5).half().float()
    dy1 = (torch.randn((B, Cout, Ho, Wo), generator=g) * 0.5).half().float()
    w3 = torch.zeros((Cout, Cin, 3, 3), requires_grad=True)
    w1 = torch.zeros((Cout, Cin, 1, 1), requires_grad=True)
    F.conv2d(x.float(), w3, None, stride=2, padding=1).backward(dy3)
    F.conv2d(x.float(), w1, None, stride=2).backward(dy1)
    xd = x.to(DEV).contiguous()
    r3, r1 = _nhwc(dy3), _nhwc(dy1)
    ws = torch.empty(int(lib.y6_wgrad_stem_workspace_bytes(Cout)), dtype=torch.uint8, device=DEV)
    outs = []
    for _ in range(2):
        o3 = torch.full((Cout, Cin, 3, 3), 1.0, dtype=torch.float32, device=DEV)       # the op accumulates
        o1 = torch.full((Cout, Cin), -2.0, dtype=torch.float32, device=DEV)
        d = _lib.WgradStemDesc()
        d.x, d.in_dtype = xd.data_ptr(), _lib.Y6_F16
        d.B, d.Cin, d.H, d.W, d.Cout = B, Cin, H, W, Cout
        d.dy3 = r3.ct()
        d.out3 = o3.data_ptr()
        if with_1x1:
            d.dy1 = r1.ct()
            d.out1 = o1.data_ptr()
        d.workspace, d.workspace_bytes = ws.data_ptr(), ws.numel()
        if H % 2 or W % 2:
            assert lib.y6_wgrad_stem_supported(C.byref(d)) == 0
            return
        assert lib.y6_wgrad_stem_supported(C.byref(d)) == 1
        _lib.check(lib.y6_wgrad_stem(C.byref(d), _stream()), "wgrad_stem")
        torch.cuda.synchronize()
        outs.append((o3.cpu(), o1.cpu()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]), "two runs differ"
    g3 = outs[0][0] - 1.0
    assert float((g3 - w3.grad).abs().max()) <= 1e-3 * float(w3.grad.abs().max()) + 1e-4
    if with_1x1:
        g1 = outs[0][1] + 2.0
        assert float((g1 - w1.grad.view(Cout, Cin)).abs().max()) <= 1e-3 * float(w1.grad.abs().max()) + 1e-4
    else:
        assert torch.equal(outs[0][1], torch.full((Cout, Cin), -2.0))


def _bn_stats(ref: TRef, gamma, beta, rm, rv, eps=1e-3, mom=0.03):
    lib = _lib.load()
    Cn = ref.C
    outs = [torch.empty(Cn, dtype=torch.float32, device=DEV) for _ in range(4)]
    ws = torch.zeros(int(lib.y6_bn_stats_workspace_bytes(Cn)), dtype=torch.uint8, device=DEV)
    nb = torch.zeros((), dtype=torch.int64, device=DEV)
    d = _lib.BnTrainDesc()
    d.x = ref.ct()
    d.gamma, d.beta = gamma.data_ptr(), beta.data_ptr()
    d.running_mean, d.running_var, d.num_batches_tracked = rm.data_ptr(), rv.data_ptr(), nb.data_ptr()
    d.momentum, d.eps = mom, eps
    d.scale, d.shift, d.mean, d.invstd = (t.data_ptr() for t in outs)
    d.workspace, d.workspace_bytes = ws.data_ptr(), ws.numel()
    _lib.check(lib.y6_bn_train_stats(C.byref(d), _stream()), "bn_train_stats")
    return outs, nb


@pytest.mark.parametrize("shape,n", [((4, 64, 40, 40), 3), ((2, 128, 33, 20), 2), ((3, 24, 6, 10), 3), ((2, 256, 20, 20), 1)])
def test_bn_stats_multi_equals_the_single_launches_bit_for_bit(shape, n):
    """y6_bn_train_stats_multi (the statistics of a RepVGG block's branch tensors in one launch pair, common.py:250-255) leaves
    the bits of n calls of y6_bn_train_stats - scale / shift / mean / invstd, running statistics, num_batches_tracked - and both
    agree with torch's batch statistics."""
    lib = _lib.load()
    g = torch.Generator().manual_seed(5)
    B, Cn, H, W = shape
    ys = [(torch.randn((B, Cn, H, W), generator=g) * s + m).half().float() for s, m in ((1.0, 0.0), (0.5, 0.3), (2.0, -1.0))][:n]
    gam = [(torch.rand(Cn, generator=g) + 0.5).to(DEV) for _ in range(n)]
    bet = [(torch.randn(Cn, generator=g) * 0.1).to(DEV) for _ in range(n)]
    refs = [_nhwc(y) for y in ys]
    single = []
    for y, r, ga, be in zip(ys, refs, gam, bet):
        rm, rv = torch.zeros(Cn, device=DEV), torch.ones(Cn, device=DEV)
        outs, nb = _bn_stats(r, ga, be, rm, rv)
        single.append(outs + [rm, rv, nb])
    md = _lib.BnTrainMultiDesc()
    md.n = n
    keep, multi = [], []
    for t, (r, ga, be) in enumerate(zip(refs, gam, bet)):
        outs = [torch.empty(Cn, dtype=torch.float32, device=DEV) for _ in range(4)]
        ws = torch.zeros(int(lib.y6_bn_stats_workspace_bytes(Cn)), dtype=torch.uint8, device=DEV)
        rm, rv, nb = torch.zeros(Cn, device=DEV), torch.ones(Cn, device=DEV), torch.zeros((), dtype=torch.int64, device=DEV)
        d = md.d[t]
        d.x = r.ct()
        d.gamma, d.beta = ga.data_ptr(), be.data_ptr()
        d.running_mean, d.running_var, d.num_batches_tracked = rm.data_ptr(), rv.data_ptr(), nb.data_ptr()
        d.momentum, d.eps = 0.03, 1e-3
        d.scale, d.shift, d.mean, d.invstd = (o.data_ptr() for o in outs)
        d.workspace, d.workspace_bytes, d.workspace_clean = ws.data_ptr(), ws.numel(), 1
        keep.append(ws)
        multi.append(outs + [rm, rv, nb])
    _lib.check(lib.y6_bn_train_stats_multi(C.byref(md), _stream()), "bn_train_stats_multi")
    torch.cuda.synchronize()
    for t in range(n):
        for i, (a, b) in enumerate(zip(single[t], multi[t])):
            assert torch.equal(a, b), f"tensor {t}, output {i}: the shared launch changed bits"
        assert int(multi[t][6]) == 1
        mean = ys[t].double().mean((0, 2, 3))
        var = ys[t].double().var((0, 2, 3), unbiased=False)
        assert float((multi[t][2].cpu().double() - mean).abs().max()) <= 1e-5 * max(1.0, float(mean.abs().max()))
        assert float((multi[t][3].cpu().double() * torch.sqrt(var + 1e-3) - 1.0).abs().max()) <= 1e-5
    # entries that share a BatchNorm's outputs are refused, not raced
    if n >= 2:
        md.d[1].scale = md.d[0].scale
        rc = lib.y6_bn_train_stats_multi(C.byref(md), _stream())
        assert rc != 0


@pytest.mark.parametrize("act,with_res,dil", [("relu", False, 1), ("silu", False, 1), ("relu", True, 1), ("relu", False, 2), (None, False, 1)])
def test_bnact_forward_backward_vs_autograd(act, with_res, dil):
    """ReLU(bn(y3) + bn(y1) + bn_id(x)) [+ alpha*res] with batch statistics, and every gradient of it."""
    _run_bnact_case(act, with_res, dil, (3, 24, 6, 10))


@pytest.mark.parametrize("shape", [(4, 64, 40, 40), (2, 128, 33, 20)])
def test_bnact_sums_many_blocks_reproducible(shape):
    """The 8-channels-per-thread kernels at sizes where the per-channel sums span many blocks (block partials + ordered
    second-level sums, no atomics): every output equals autograd, and two runs agree bit for bit."""
    a = _run_bnact_case("relu", True, 1, shape)
    b = _run_bnact_case("relu", True, 1, shape)
    assert len(a) == len(b)
    for i, (x, y) in enumerate(zip(a, b)):
        assert torch.equal(x, y), f"output {i} differs between two runs"


def _run_bnact_case(act, with_res, dil, shape):
    lib = _lib.load()
    g = torch.Generator().manual_seed(11)
    B, Cn, H, W = shape
    ys = [(torch.randn((B, Cn, H, W), generator=g) * s).half().float() for s in (1.0, 0.5, 2.0)]
    res = torch.randn((B, Cn, H, W), generator=g).half().float()
    gam = [torch.rand(Cn, generator=g) + 0.5 for _ in range(3)]
    bet = [torch.randn(Cn, generator=g) * 0.1 for _ in range(3)]
    alpha = torch.tensor([0.7])
    dout = torch.randn((B, Cn, H, W), generator=g).half().float()
    # ---- torch reference (fp32 autograd)
    yv = [y.clone().requires_grad_(True) for y in ys]
    gv = [t.clone().requires_grad_(True) for t in gam]
    bv = [t.clone().requires_grad_(True) for t in bet]
    rv_ = res.clone().requires_grad_(True)
    av = alpha.clone().requires_grad_(True)
    rms = [torch.zeros(Cn) for _ in range(3)]
    rvs = [torch.ones(Cn) for _ in range(3)]
    z = sum(F.batch_norm(y, rm, rvv, w, b, True, 0.03, 1e-3) for y, rm, rvv, w, b in zip(yv, rms, rvs, gv, bv))
    o = {"relu": F.relu, "silu": F.silu, None: lambda t: t}[act](z)
    if with_res:
        o = o + av * rv_
    o.backward(dout)
    # ---- HIP
    refs = [_nhwc(y) for y in ys]
    dgam = [t.to(DEV) for t in gam]
    dbet = [t.to(DEV) for t in bet]
    stats, rmd, rvd = [], [], []
    for i in range(3):
        rm, rvv = torch.zeros(Cn, device=DEV), torch.ones(Cn, device=DEV)
        st, nb = _bn_stats(refs[i], dgam[i], dbet[i], rm, rvv)
        stats.append(st)
        rmd.append(rm)
        rvd.append(rvv)
    out = TRef(torch.empty((B, H, W, Cn), dtype=torch.float16, device=DEV), B, H, W, Cn, Cn, 0)
    rref = _nhwc(res)
    adev = alpha.to(DEV)
    d = _lib.BnActDesc()
    d.n = 3
    for i in range(3):
        d.x[i] = refs[i].ct()
        d.scale[i], d.shift[i] = stats[i][0].data_ptr(), stats[i][1].data_ptr()
    if with_res:
        d.res = rref.ct()
        d.res_alpha = adev.data_ptr()
    d.out = out.ct()
    d.act = _lib.ACT_BY_NAME[act]
    _lib.check(lib.y6_bnact_forward(C.byref(d), _stream()), "bnact_forward")
    torch.cuda.synchronize()
    assert float((_back(out) - o.detach()).abs().max()) < 4e-3 * max(1.0, float(o.abs().max()))     # fp16 output
    for i in range(3):
        assert torch.allclose(rmd[i].cpu(), rms[i], atol=1e-5) and torch.allclose(rvd[i].cpu(), rvs[i], atol=1e-5, rtol=1e-4)
    assert int(nb) == 1
    b = _lib.BnActBwdDesc()
    b.fwd = d
    dref = _nhwc(dout)
    b.dout = dref.ct()
    dxs = []
    dgd = [torch.zeros(Cn, device=DEV) for _ in range(3)]
    dbd = [torch.zeros(Cn, device=DEV) for _ in range(3)]
    for i in range(3):
        b.mean[i], b.invstd[i] = stats[i][2].data_ptr(), stats[i][3].data_ptr()
        b.gamma[i] = dgam[i].data_ptr()
        dil_i = dil if i == 0 else 1
        t = torch.zeros((B, H * dil_i, W * dil_i, Cn), dtype=torch.float16, device=DEV)
        if i == 2:
            t += 1.0                          # accumulate mode: the buffer already holds a gradient of ones
        dxs.append(TRef(t, B, H * dil_i, W * dil_i, Cn, Cn, 0))
        b.dx[i] = dxs[i].ct()
        b.dx_dil[i] = dil_i
        b.dx_acc[i] = 1 if i == 2 else 0
        b.dgamma[i], b.dbeta[i] = dgd[i].data_ptr(), dbd[i].data_ptr()
    dres = TRef(torch.zeros((B, H, W, Cn), dtype=torch.float16, device=DEV), B, H, W, Cn, Cn, 0)
    dal = torch.zeros(1, device=DEV)
    if with_res:
        b.dres = dres.ct()
        b.dalpha = dal.data_ptr()
    ws = torch.zeros(int(lib.y6_bnact_bwd_workspace_bytes(Cn)), dtype=torch.uint8, device=DEV)
    b.workspace, b.workspace_bytes = ws.data_ptr(), ws.numel()
    _lib.check(lib.y6_bnact_backward(C.byref(b), _stream()), "bnact_backward")
    torch.cuda.synchronize()
    for i in range(3):
        got = _back(dxs[i])
        if i == 0 and dil == 2:
            full = got
            got = full[:, :, ::2, ::2].clone()
            full[:, :, ::2, ::2] = 0
            assert float(full.abs().max()) == 0.0           # nothing but the (2y, 2x) positions is written
        if i == 2:
            got = got - 1.0
        ref = yv[i].grad
        assert float((got - ref).abs().max()) < 3e-3 * max(1.0, float(ref.abs().max())), f"dx[{i}]"
        assert float((dgd[i].cpu() - gv[i].grad).abs().max()) < 2e-3 * float(gv[i].grad.abs().max()), f"dgamma[{i}]"
        assert float((dbd[i].cpu() - bv[i].grad).abs().max()) < 2e-3 * float(bv[i].grad.abs().max()), f"dbeta[{i}]"
    if with_res:
        assert float((_back(dres) - rv_.grad).abs().max()) < 3e-3 * float(rv_.grad.abs().max())
        assert abs(float(dal) - float(av.grad)) < 2e-3 * abs(float(av.grad))
    outs = [t.cpu() for st in stats for t in st] + [t.cpu() for t in rmd + rvd + dgd + dbd] + [_back(t) for t in dxs] + [_back(out), dal.cpu()]
    return outs


def test_sppf_pool_backward_first_max_semantics():
    lib = _lib.load()
    g = torch.Generator().manual_seed(5)
    B, Cn, H, W = 2, 16, 20, 20
    # ReLU-like input: many exact ties (zeros) - the gradient must go to the FIRST maximum in window order
    x = F.relu(torch.randn((B, Cn, H, W), generator=g)).half().float().requires_grad_(True)
    y1 = F.max_pool2d(x, 5, 1, 2)
    y2 = F.max_pool2d(y1, 5, 1, 2)
    y3 = F.max_pool2d(y2, 5, 1, 2)
    d0, d1, d2, d3 = [torch.randn((B, Cn, H, W), generator=g).half().float() for _ in range(4)]
    ((x * d0).sum() + (y1 * d1).sum() + (y2 * d2).sum() + (y3 * d3).sum()).backward()
    ref = x.grad
    xr, y1r, y2r = _nhwc(x.detach()), _nhwc(y1.detach()), _nhwc(y2.detach())
    dx, g1, g2, g3 = _nhwc(d0), _nhwc(d1), _nhwc(d2), _nhwc(d3)
    d = _lib.SppfBwdDesc()
    d.x, d.y1, d.y2 = xr.ct(), y1r.ct(), y2r.ct()
    d.dy1, d.dy2, d.dy3, d.dx = g1.ct(), g2.ct(), g3.ct(), dx.ct()
    d.dx_acc = 1
    _lib.check(lib.y6_sppf_pool_backward(C.byref(d), _stream()), "sppf_pool_backward")
    torch.cuda.synchronize()
    assert float((_back(dx) - ref).abs().max()) < 2e-2      # sums of up to ~75 fp16 values, stored in fp16


def test_head_pack_and_unpack():
    lib = _lib.load()
    g = torch.Generator().manual_seed(2)
    B, nc, nreg = 2, 80, 68
    sizes = [(8, 8), (4, 4), (2, 2)]
    cls = [torch.randn((B, nc, h, w), generator=g).half().float() for h, w in sizes]
    reg = [torch.randn((B, nreg, h, w), generator=g).half().float() for h, w in sizes]
    A = sum(h * w for h, w in sizes)
    d = _lib.HeadPackDesc()
    d.n_levels = 3
    crefs, rrefs = [_nhwc(t) for t in cls], [_nhwc(t) for t in reg]
    for i in range(3):
        d.cls[i], d.reg[i] = crefs[i].ct(), rrefs[i].ct()
    scores = torch.empty((B, A, nc), device=DEV)
    distri = torch.empty((B, A, nreg), device=DEV)
    d.scores, d.distri, d.nc, d.nreg = scores.data_ptr(), distri.data_ptr(), nc, nreg
    _lib.check(lib.y6_head_pack(C.byref(d), _stream()), "head_pack")
    ref_s = torch.cat([torch.sigmoid(t).flatten(2).permute(0, 2, 1) for t in cls], 1)
    ref_d = torch.cat([t.flatten(2).permute(0, 2, 1) for t in reg], 1)
    assert torch.allclose(scores.cpu(), ref_s, atol=1e-6) and torch.equal(distri.cpu(), ref_d)
    ds, dd = torch.randn((B, A, nc), generator=g), torch.randn((B, A, nreg), generator=g)
    dsd, ddd = ds.to(DEV), dd.to(DEV)
    gcls = [TRef(torch.zeros((B, h, w, nc), dtype=torch.float16, device=DEV), B, h, w, nc, nc, 0) for h, w in sizes]
    greg = [TRef(torch.zeros((B, h, w, 72), dtype=torch.float16, device=DEV), B, h, w, nreg, 72, 0) for h, w in sizes]
    u = _lib.HeadPackDesc()
    u.n_levels = 3
    for i in range(3):
        u.cls[i], u.reg[i] = gcls[i].ct(), greg[i].ct()
    u.scores, u.dscores, u.ddistri, u.nc, u.nreg = scores.data_ptr(), dsd.data_ptr(), ddd.data_ptr(), nc, nreg
    _lib.check(lib.y6_head_unpack_backward(C.byref(u), _stream()), "head_unpack_backward")
    torch.cuda.synchronize()
    a0 = 0
    for i, (h, w) in enumerate(sizes):
        n = h * w
        p = ref_s[:, a0:a0 + n]
        want_c = (ds[:, a0:a0 + n] * p * (1 - p)).permute(0, 2, 1).reshape(B, nc, h, w)
        want_r = dd[:, a0:a0 + n].permute(0, 2, 1).reshape(B, nreg, h, w)
        assert float((_back(gcls[i]) - want_c).abs().max()) < 2e-3
        assert float((_back(greg[i]) - want_r).abs().max()) < 3e-3
        assert float(greg[i].buf[..., nreg:].abs().max()) == 0.0       # pad channels stay zero
        a0 += n


def test_sgd_step_matches_torch_and_skips_on_overflow():
    lib = _lib.load()
    g = torch.Generator().manual_seed(9)
    n = 10007
    p0, gr = torch.randn(n, generator=g), torch.randn(n, generator=g)
    ref = torch.nn.Parameter(p0.clone())
    opt = torch.optim.SGD([ref], lr=0.02, momentum=0.937, nesterov=True, weight_decay=5e-4)
    p, m = p0.clone().to(DEV), torch.zeros(n, device=DEV)
    scale = torch.tensor([1024.0], device=DEV)
    flag = torch.zeros(1, dtype=torch.int32, device=DEV)
    for step in range(3):
        ref.grad = gr.clone() * (step + 1)
        opt.step()
        gd = (gr * (step + 1) * 1024.0).to(DEV)
        _lib.check(lib.y6_grad_finite_check(gd.data_ptr(), n, flag.data_ptr(), _stream()), "finite")
        _lib.check(lib.y6_sgd_step(p.data_ptr(), gd.data_ptr(), m.data_ptr(), n, 0.02, 0.937, 5e-4, 1, int(step == 0), scale.data_ptr(),
                                   flag.data_ptr(), _stream()), "sgd")
    assert torch.allclose(p.cpu(), ref.detach(), atol=1e-5, rtol=1e-5)
    before = p.clone()
    gd[5] = float("inf")
    tracker = torch.zeros(1, dtype=torch.int32, device=DEV)
    _lib.check(lib.y6_grad_finite_check(gd.data_ptr(), n, flag.data_ptr(), _stream()), "finite")
    _lib.check(lib.y6_sgd_step(p.data_ptr(), gd.data_ptr(), m.data_ptr(), n, 0.02, 0.937, 5e-4, 1, 0, scale.data_ptr(), flag.data_ptr(),
                               _stream()), "sgd")
    _lib.check(lib.y6_scaler_update(scale.data_ptr(), flag.data_ptr(), tracker.data_ptr(), 2.0, 0.5, 2000, _stream()), "scaler")
    assert torch.equal(p, before) and float(scale) == 512.0 and int(flag) == 0


def _backward_at_a_scale_that_fits(model, scalar, S=256.0):
    """`(scalar * S).backward()` with S halved until no parameter gradient is non-finite - what solver.LossScaler does to a
    training run (an overflowing step is skipped, the scale halved).  Activation gradients are fp16: on these small random
    models the gradient arriving at the stem can exceed fp16's range at S = 256, and a NaN error compares False against any
    bound - which is how such parameters passed unexamined until round 4.  Returns the scale used."""
    while True:
        for p_ in model.parameters():
            p_.grad = None
        (scalar * S).backward(retain_graph=True)
        torch.cuda.synchronize()
        if all(bool(torch.isfinite(p_.grad).all()) for p_ in model.parameters() if p_.grad is not None) or S <= 4.0:
            return S
        S /= 2.0


LOSS_CASES = sorted(f[len("lossgrad_"):-len(".npz")] for f in os.listdir(GOLDEN) if f.startswith("lossgrad_"))


@pytest.mark.parametrize("case", LOSS_CASES)
def test_compute_loss_gradient_vs_reference_autograd(case):
    """ComputeLoss(...)[0].backward() on the HIP path == the gradients the unmodified reference back-propagates."""
    from yolov6_amd.models.losses.loss import ComputeLoss
    from yolov6_amd.utils import synth
    gl = np.load(os.path.join(GOLDEN, f"loss_{case}.npz"))
    gg = np.load(os.path.join(GOLDEN, f"lossgrad_{case}.npz"))
    m = json.loads(str(gl["meta"]))
    inp = synth.synth_loss_inputs(m["B"], m["feat_sizes"], m["strides"], m["C"], m["reg_max"], m["use_dfl"], seed=m["seed"])
    targets = inp["targets"] if case != "no_targets" else inp["targets"][:0]
    crit = ComputeLoss(fpn_strides=m["strides"], num_classes=m["C"], ori_img_size=inp["img"], warmup_epoch=4, use_dfl=m["use_dfl"],
                       reg_max=m["reg_max"], iou_type=m["iou_type"])
    feats = [torch.zeros(m["B"], 1, h, w, device=DEV) for h, w in m["feat_sizes"]]
    ps = inp["pred_scores"].to(DEV).requires_grad_(True)
    pd = inp["pred_distri"].to(DEV).requires_grad_(True)
    loss, items = crit((feats, ps, pd), targets.to(DEV), m["epoch"], 1, inp["img"], inp["img"])
    (loss * 8.0).backward()                       # the incoming gradient (a loss scale) is applied inside the kernels
    torch.cuda.synchronize()
    np.testing.assert_allclose(float(loss), float(gg["loss"]), rtol=1e-4)
    for got, name in ((ps.grad, "dscores"), (pd.grad, "ddistri")):
        ref = gg[name].astype(np.float64) * 8.0
        scale = max(float(np.abs(ref).max()), 1e-12)
        err = float(np.abs(got.cpu().numpy().astype(np.float64) - ref).max()) / scale
        assert err < 1e-4, f"{case}: {name} deviates by {err:.3e} of its max"


def _tiny_train_model(case):
    from tests.helpers import case_config, synth_sd_from_keys
    from yolov6_amd.models.yolo import build_model
    cfg, meta = case_config(case)
    model = build_model(cfg, meta["num_classes"], "cpu")
    sd = synth_sd_from_keys(meta["train"])
    model.load_state_dict(sd)
    return cfg, meta, sd, model


def _oracle_run(cfg, sd, nc, x, amp):
    """TrainOracle forward + backward of the goldens' scalar; returns (scalar, cls, reg, stems, {param: grad}, new_stats)."""
    from oracle.model_oracle import TrainOracle
    params = {k: v.clone().float().requires_grad_(True) for k, v in sd.items() if v.dtype.is_floating_point and "running_" not in k}
    orc = TrainOracle(cfg, sd, nc, amp_fp16=amp)
    orc.sd = {k: (params[k] if k in params else v.float()) for k, v in sd.items()}
    (xs, cls_o, reg_o), _ = orc.forward_train(x)
    scalar = (cls_o * cls_o).sum() + reg_o.square().mean()
    scalar.backward()
    grads = {k: p.grad.detach() for k, p in params.items() if p.grad is not None}
    return float(scalar), cls_o.detach(), reg_o.detach(), [t.detach() for t in xs], grads, orc.new_stats


def _rel_l2(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a - b).norm() / b.norm().clamp(min=1e-30))


@pytest.mark.parametrize("case,size,batch", [("tiny", 64, 2), ("tiny", 192, 4)])
def test_training_graph_forward_backward_vs_oracle(case, size, batch):
    """Whole model, train form: head outputs, BatchNorm running statistics and EVERY parameter gradient.

    Reference point: TrainOracle in fp32 (CPU autograd; at the golden size it IS the run that produced the reference's
    goldens: scalar, outputs, running statistics and three reference gradients are pinned by tests/test_oracle_cpu.py).
    fp16 activations put any AMP-style pipeline a few 1e-2 away from the fp32 graph on this random-weight network (batch
    statistics re-normalise every layer, so rounding noise is not damped); the bar is therefore stated against the noise
    floor measured with the same oracle run with fp16-rounded activations (`amp_fp16=True`, straight-through gradients):
        err(HIP, fp32 oracle) <= 2 x err(fp16-activation oracle, fp32 oracle) + 2e-3      (relative L2 per tensor)
    A wiring error (missing branch, wrong accumulation, flipped kernel) shows up as an O(0.3 ... 1) deviation."""
    from oracle import synth
    cfg, meta, sd, model = _tiny_train_model(case)
    nc = meta["num_classes"]
    x = synth.synth_images(batch, size, seed=21)
    xh = x.half().float()
    sc32, cls32, reg32, xs32, g32, stats32 = _oracle_run(cfg, sd, nc, xh, amp=False)
    sc16, cls16, reg16, xs16, g16, stats16 = _oracle_run(cfg, sd, nc, xh, amp=True)
    if size == 64 and batch == 2:
        gold = np.load(os.path.join(GOLDEN, f"train_{case}.npz"))
        sc_g, *_ = _oracle_run(cfg, sd, nc, x, amp=False)
        np.testing.assert_allclose(sc_g, float(gold["scalar"]), rtol=1e-4)        # the oracle run IS the golden run
    # ---- HIP
    model = model.to(DEV).train()
    out, featmaps = model(x.to(DEV).half())
    stems, scores, distri = out
    scalar = (scores * scores).sum() + distri.square().mean()
    S = _backward_at_a_scale_that_fits(model, scalar)     # loss scale: activation gradients are fp16
    rep = dict(case=case, size=size, batch=batch)
    for name, got, r32, r16 in (("cls_scores", scores.detach().cpu(), cls32, cls16), ("reg_distri", distri.detach().cpu(), reg32, reg16)):
        e_hip, e_ref = _rel_l2(got, r32), _rel_l2(r16, r32)
        rep[name] = dict(hip=e_hip, fp16_floor=e_ref)
        assert e_hip <= 2 * e_ref + 2e-3, f"{name}: HIP {e_hip:.3e} vs fp16 noise floor {e_ref:.3e}"
    for f, r in zip(list(stems), xs32):
        assert f.shape == r.shape
    # running statistics after one step (momentum 0.03, unbiased variance)
    msd = model.state_dict()
    worst_stat = floor_stat = 0.0
    for p, (rm, rv) in stats32.items():
        for j, (key, ref) in enumerate(((p + ".running_mean", rm), (p + ".running_var", rv))):
            den = max(1.0, float(ref.abs().max()))
            worst_stat = max(worst_stat, float((msd[key].cpu() - ref.detach()).abs().max()) / den)
            floor_stat = max(floor_stat, float((stats16[p][j].detach() - ref.detach()).abs().max()) / den)
    rep["running_stats_worst"] = dict(hip=worst_stat, fp16_floor=floor_stat)
    assert worst_stat < 2 * floor_stat + 1e-3, f"running statistics deviate by {worst_stat:.3e} (fp16 floor {floor_stat:.3e})"
    k0 = "backbone.stem.rbr_dense.bn.num_batches_tracked"
    assert int(msd[k0]) == int(sd[k0]) + 1
    # every parameter gradient
    named = dict(model.named_parameters())
    errs, floors, bad = {}, {}, []
    for k, ref in g32.items():
        if k == "detect.proj" or k.startswith("detect.proj_conv") or float(ref.norm()) == 0.0:
            continue
        got = named[k].grad.detach().float().cpu() / S
        errs[k], floors[k] = _rel_l2(got, ref), _rel_l2(g16[k], ref)
        if floors[k] > 1.0:
            # the fp32 value is rounding noise around a mathematically zero gradient (a ConvTranspose bias in front of a
            # 1x1 conv + BatchNorm: the batch mean removes it exactly) - nothing to compare
            del errs[k], floors[k]
            continue
        if not (errs[k] <= 2 * floors[k] + 2e-3):      # (written so that a NaN / inf error is a failure, not a silent pass)
            bad.append((k, errs[k], floors[k]))
    worst = max(errs, key=errs.get)
    rep.update(n_params=len(errs), grad_worst=dict(name=worst, hip=errs[worst], fp16_floor=floors[worst]),
               grad_median=dict(hip=float(np.median([v for v in errs.values() if v == v])),
                                fp16_floor=float(np.median([v for v in floors.values() if v == v]))))
    out_dir = os.path.join(os.path.dirname(GOLDEN), "..", "gpurun_out")
    os.makedirs(out_dir, exist_ok=True)
    with open(os.path.join(out_dir, f"train_grad_report_{case}_{size}_b{batch}.json"), "w") as f:
        json.dump(dict(summary=rep, errs=errs, floors=floors), f, indent=1)
    print(json.dumps(rep))
    assert not bad, f"{len(bad)} parameter gradients above twice the fp16 noise floor, e.g. {bad[:5]}"


# ------------------------------------------------------------------------------------------------------------------
# Block-level training graphs: the wiring of every block type (branches, accumulation, dilation, concat slices,
# flipped / transposed weight images) against torch autograd of the oracle's functional statement.  A block is only a few
# layers deep, so fp16 storage keeps everything within ~1e-2 of fp32 and a wiring error cannot hide in rounding noise.
def _block_case(kind):
    from yolov6_amd.layers import common as L
    if kind == "repvgg_s1":
        return L.RepVGGBlock(32, 32), [(2, 32, 12, 20)], lambda o, xs: o.block(xs[0], "m", 1), "repvgg"
    if kind == "repvgg_s2":
        return L.RepVGGBlock(16, 32, stride=2), [(2, 16, 12, 20)], lambda o, xs: o.block(xs[0], "m", 2), "repvgg"
    if kind == "repvgg_widen":
        return L.RepVGGBlock(16, 48), [(3, 16, 9, 11)], lambda o, xs: o.block(xs[0], "m", 1), "repvgg"
    if kind == "convbnsilu3":
        return L.ConvBNSiLU(24, 40, 3, 1), [(2, 24, 10, 10)], lambda o, xs: o.convbn(xs[0], "m", "silu"), "repvgg"
    if kind == "convbnrelu1":
        return L.ConvBNReLU(64, 24, 1, 1), [(2, 64, 7, 9)], lambda o, xs: o.convbn(xs[0], "m", "relu"), "repvgg"
    if kind == "convbnrelu3s2":
        return L.ConvBNReLU(16, 16, 3, 2), [(2, 16, 8, 12)], lambda o, xs: o.convbn(xs[0], "m", "relu", stride=2), "repvgg"
    if kind == "repblock":
        return L.RepBlock(16, 32, n=3), [(2, 16, 10, 14)], lambda o, xs: o.repblock(xs[0], "m", 3), "repvgg"
    if kind == "simsppf":
        return L.SimSPPF(32, 32), [(2, 32, 9, 9)], lambda o, xs: o.sppf(xs[0], "m.sppf", "relu"), "repvgg"
    if kind == "simcspsppf":
        return L.SimCSPSPPF(32, 32), [(2, 32, 8, 8)], lambda o, xs: o.cspsppf(xs[0], "m.cspsppf", "relu"), "repvgg"
    if kind == "transpose":
        return L.Transpose(32, 32), [(2, 32, 5, 7)], lambda o, xs: o.transpose(xs[0], "m"), "repvgg"
    if kind == "bifusion":
        return (L.BiFusion([32, 16], 16), [(2, 16, 4, 6), (2, 32, 8, 12), (2, 16, 16, 24)],
                lambda o, xs: o.bifusion(xs, "m"), "repvgg")
    if kind == "bottlerep":
        return L.BottleRep(16, 16, basic_block=L.RepVGGBlock, weight=True), [(2, 16, 8, 8)], lambda o, xs: o.bottlerep(xs[0], "m"), "repvgg"
    if kind == "bepc3":
        return L.BepC3(32, 32, n=2, block=L.RepVGGBlock), [(2, 32, 8, 8)], lambda o, xs: o.bepc3(xs[0], "m", 2), "repvgg"
    if kind == "qarep_s1":        # first version: raw 1x1 and identity branches, BatchNorm after the sum (common.py:337-343)
        return L.QARepVGGBlock(32, 32), [(2, 32, 12, 20)], lambda o, xs: o.block(xs[0], "m", 1), "qarepvgg"
    if kind == "qarepv2_s1":      # + the 3x3 average-pool branch (common.py:412-419): what configs/qarepvgg/* train with
        return L.QARepVGGBlockV2(32, 32), [(2, 32, 12, 20)], lambda o, xs: o.block(xs[0], "m", 1), "qarepvggv2"
    if kind == "qarepv2_s2":
        return L.QARepVGGBlockV2(16, 32, stride=2), [(2, 16, 12, 20)], lambda o, xs: o.block(xs[0], "m", 2), "qarepvggv2"
    if kind == "qarepv2_widen":
        return L.QARepVGGBlockV2(16, 48), [(3, 16, 9, 11)], lambda o, xs: o.block(xs[0], "m", 1), "qarepvggv2"
    if kind == "bottlerep3":
        return L.BottleRep3(16, 16, basic_block=L.RepVGGBlock, weight=True), [(2, 16, 8, 8)], lambda o, xs: o.bottlerep3(xs[0], "m"), "repvgg"
    if kind == "mbla":            # three branches: n = 4 -> n_list [0, 1, 2] (common.py:653-692)
        return L.MBLABlock(32, 32, n=4, block=L.RepVGGBlock), [(2, 32, 8, 8)], lambda o, xs: o.mbla(xs[0], "m", 4), "repvgg"
    if kind == "mbla_silu":       # two branches, ConvBNSiLU body (the yolov6*_mbla configs)
        return L.MBLABlock(32, 64, n=2, block=L.ConvBNSiLU), [(2, 32, 8, 8)], lambda o, xs: o.mbla(xs[0], "m", 2), "conv_silu"
    if kind == "bepc3_silu":
        return L.BepC3(32, 32, n=2, block=L.ConvBNSiLU), [(2, 32, 8, 8)], lambda o, xs: o.bepc3(xs[0], "m", 2), "conv_silu"
    raise KeyError(kind)


BLOCKS = ["repvgg_s1", "repvgg_s2", "repvgg_widen", "convbnsilu3", "convbnrelu1", "convbnrelu3s2", "repblock", "simsppf", "simcspsppf",
          "transpose", "bifusion", "bottlerep", "bepc3", "bepc3_silu", "qarep_s1", "qarepv2_s1", "qarepv2_s2", "qarepv2_widen",
          "bottlerep3", "mbla", "mbla_silu"]


@pytest.mark.parametrize("case,policy", [("tiny", "asap"), ("tiny", "alap"), ("s_qa_tiny", "asap"), ("s_mbla_tiny", "asap")])
def test_training_forward_two_stream_schedule_bit_identical(case, policy):
    """The training-form forward with its branch ops on the plan's side stream (train_engine.schedule_forward, Y6_TRAIN_FWD_STREAMS=2):
    same kernels, same data - head outputs, stem feature maps, every BatchNorm's running statistics and every parameter
    gradient of the step equal the one-stream run bit for bit (the sums of these kernels are order-fixed, so equality is the bar)."""
    import copy
    from oracle import synth
    from yolov6_amd import schedule as Sch
    cfg, meta, sd, model = _tiny_train_model(case)
    x = synth.synth_images(2, 64, seed=31).to(DEV).half()
    runs = {}
    for mode in ("one", "two"):
        m = copy.deepcopy(model).to(DEV).train()
        m(x)                                                       # builds the graph (and takes one step of running statistics)
        graph = next(iter(m.__dict__["_y6_train_graphs"].values()))
        if mode == "two":
            info = graph.fwd_plan.schedule(costs=Sch.train_costs(graph.fwd_log), accesses=[Sch.train_op_access(e) for e in graph.fwd_log],
                                           policy=policy)
            assert info is not None and len(info["side_ops"]) > graph.fwd_plan.num_ops // 5
        else:
            graph.fwd_plan.clear_schedule()
        outs = []
        for step in range(3):
            for p in m.parameters():
                p.grad = None
            (stems, scores, distri), _ = m(x)
            ((scores * scores).sum() * 64.0 + distri.square().mean() * 64.0).backward()
            outs.append((scores.detach().clone(), distri.detach().clone()))
        torch.cuda.synchronize()
        runs[mode] = (outs, [t.clone() for t in stems], {k: v.clone() for k, v in m.state_dict().items()},
                      {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None})
    (o1, f1, s1, g1), (o2, f2, s2, g2) = runs["one"], runs["two"]
    for (a, b), (c, d) in zip(o1, o2):
        assert torch.equal(a, c) and torch.equal(b, d), f"{case}/{policy}: head outputs differ"
    for a, b in zip(f1, f2):
        assert torch.equal(a, b)
    assert s1.keys() == s2.keys() and g1.keys() == g2.keys() and len(g1) > 20
    for k in s1:
        assert torch.equal(s1[k], s2[k]), f"{case}/{policy}: {k} differs after three steps"
    for k in g1:      # (the backward plan is the same in both runs; equal forward activations -> equal gradients up to the order of any atomic sums)
        den = float(g1[k].float().abs().max()) + 1e-12
        assert float((g1[k].float() - g2[k].float()).abs().max()) <= 1e-5 * den, f"{case}/{policy}: gradient of {k} differs"


@pytest.mark.parametrize("kind", BLOCKS)
def test_block_training_graph_vs_autograd(kind):
    from oracle import synth
    from oracle.model_oracle import TrainOracle
    from yolov6_amd.configs import tiny_config
    from yolov6_amd.train_engine import ModuleTrainGraph
    from yolov6_amd.utils.torch_utils import initialize_weights
    torch.manual_seed(0)
    module, shapes, fn, mode = _block_case(kind)
    initialize_weights(module)                       # eps 1e-3, momentum 0.03 as in the model
    sd_m = synth.synth_state_dict(module.state_dict(), seed=4)
    if "alpha" in sd_m:
        sd_m["alpha"] = torch.tensor([0.8])
    module.load_state_dict(sd_m)
    g = torch.Generator().manual_seed(1)
    xs = [(torch.randn(s, generator=g)).half().float() for s in shapes]
    # ---- torch autograd of the oracle statement (fp32)
    cfg = tiny_config()
    cfg["training_mode"] = mode
    sd = {"m." + k: v for k, v in sd_m.items()}
    params = {k: v.clone().float().requires_grad_(True) for k, v in sd.items() if v.dtype.is_floating_point and "running_" not in k}
    orc = TrainOracle(cfg, sd, 80)
    orc.a.mode = mode
    orc.a.silu_body = mode == "conv_silu"
    orc.sd = {k: (params[k] if k in params else v.float()) for k, v in sd.items()}
    xv = [x.clone().requires_grad_(True) for x in xs]
    out_ref = fn(orc, xv)
    dout = torch.randn(out_ref.shape, generator=g).half().float()
    out_ref.backward(dout)
    # the same statement with fp16-rounded activations (straight-through): the noise floor of an fp16 pipeline.  ReLU masks
    # and max-pool winners that flip under rounding re-route gradient, which dominates cancelling sums (BN biases).
    p16 = {k: v.clone().float().requires_grad_(True) for k, v in sd.items() if v.dtype.is_floating_point and "running_" not in k}
    o16 = TrainOracle(cfg, sd, 80, amp_fp16=True)
    o16.a.mode, o16.a.silu_body = mode, mode == "conv_silu"
    o16.sd = {k: (p16[k] if k in p16 else v.float()) for k, v in sd.items()}
    x16 = [x.clone().requires_grad_(True) for x in xs]
    fn(o16, x16).backward(dout)
    # ---- HIP
    module = module.to(DEV).train()
    graph = ModuleTrainGraph(module, [x.to(DEV).half() for x in xs])
    out = graph.forward()[0]
    dxs = graph.backward([dout.to(DEV)])
    torch.cuda.synchronize()
    e_out = _rel_l2(out.cpu(), out_ref.detach())
    assert e_out < 5e-3, f"{kind}: forward deviates by {e_out:.3e}"
    named = dict(module.named_parameters())
    worst = ("", 0.0, 0.0)
    bad = []

    def judge(name, got, ref, ref16):
        nonlocal worst
        floor = _rel_l2(ref16, ref)
        if floor > 1.0:
            return                                    # mathematically zero gradient (a bias in front of conv + BatchNorm)
        err = _rel_l2(got, ref)
        if err > worst[1]:
            worst = (name, err, floor)
        if not (err <= 3 * floor + 3e-3):      # (NaN is a failure)
            bad.append((name, err, floor))
    for k, p in params.items():
        if p.grad is not None:
            judge(k, named[k[2:]].grad.detach().float().cpu(), p.grad, p16[k].grad)
    for i, (dx, xr, xr16) in enumerate(zip(dxs, xv, x16)):
        assert dx is not None, f"{kind}: no gradient reached input {i}"
        judge(f"dx{i}", dx.cpu(), xr.grad, xr16.grad)
    # running statistics
    msd = module.state_dict()
    for p_, (rm, rv) in orc.new_stats.items():
        assert float((msd[p_[2:] + ".running_mean"].cpu() - rm.detach()).abs().max()) < 3e-3 * max(1.0, float(rm.abs().max()))
        assert float((msd[p_[2:] + ".running_var"].cpu() - rv.detach()).abs().max()) < 3e-3 * max(1.0, float(rv.abs().max()))
    print(f"{kind}: forward {e_out:.2e}, worst gradient {worst[0]} {worst[1]:.2e} (fp16 floor {worst[2]:.2e})")
    assert not bad, f"{kind}: gradients above 3x the fp16 noise floor: {bad[:4]}"


def test_full_training_steps_loss_backward_sgd():
    """model(x) -> ComputeLoss -> scaler.scale(loss).backward() -> fused SGD, three steps on a fixed batch: the loss gradient
    lands in the graph's own head-gradient buffers (no copy), gradients are finite, parameters move, the loss goes down, and
    one step equals torch.optim.SGD applied to the same arena gradients."""
    from oracle import synth
    from yolov6_amd.models.losses.loss import ComputeLoss
    from yolov6_amd.solver import FusedSGD, LossScaler, param_groups
    cfg, meta, sd, model = _tiny_train_model("tiny")
    model = model.to(DEV).train()
    x = synth.synth_images(4, 128, seed=3).to(DEV).half()
    g = torch.Generator().manual_seed(0)
    n = 12
    targets = torch.cat([torch.randint(0, 4, (n, 1), generator=g).float(), torch.randint(0, 80, (n, 1), generator=g).float(),
                         torch.rand((n, 2), generator=g) * 0.6 + 0.2, torch.rand((n, 2), generator=g) * 0.3 + 0.1], 1).to(DEV)
    h = cfg.model.head
    crit = ComputeLoss(num_classes=80, ori_img_size=128, warmup_epoch=0, use_dfl=h.use_dfl, reg_max=h.reg_max, iou_type=h.iou_type)
    (feats, scores, distri), _ = model(x)
    graph = scores._y6_graph
    arena = graph.arena
    opt = FusedSGD(model, arena, lr=0.02, momentum=0.9, weight_decay=5e-4)
    scaler = LossScaler(DEV, init_scale=1024.0)
    losses = []
    for i in range(4):
        opt.zero_grad()
        (feats, scores, distri), _ = model(x)
        loss, items = crit((feats, scores, distri), targets, 10, i, 128, 128)
        scaler.scale_loss(loss).backward()
        assert torch.isfinite(arena.grad).all()
        assert float(arena.grad.abs().max()) > 0
        if i == 0:
            # reference optimizer on the same gradients: torch.optim.SGD with the reference's three groups
            before = arena.data.clone()
            g_bnw, g_w, g_b = param_groups(model)
            ref_p = {id(p): p.detach().clone() for p in arena.params}
            grads = {id(p): p.grad.detach().clone() / 1024.0 for p in arena.params}
        opt.step(scaler)
        scaler.update()
        if i == 0:
            for k, (ps, wd) in enumerate(((g_bnw, 0.0), (g_w, 5e-4), (g_b, 0.0))):
                for p in ps:
                    if id(p) not in ref_p:
                        continue
                    d = grads[id(p)] + wd * ref_p[id(p)]
                    want = ref_p[id(p)] - 0.02 * (d + 0.9 * d)          # first Nesterov step: buf = d
                    assert torch.allclose(p.detach(), want, rtol=1e-4, atol=1e-6)
            assert not torch.equal(before, arena.data)
        losses.append(float(loss))
    print("losses", losses, "scale", float(scaler.scale))
    assert losses[-1] < losses[0], losses
    assert float(scaler.scale) == 1024.0           # no overflow happened


# ------------------------------------------------------------------ fuse_ab (anchor-based auxiliary branch)
@pytest.mark.parametrize("case", ["giou", "siou"])
def test_fuseab_loss_and_gradient_vs_reference(case):
    """loss_fuseab.ComputeLoss on the HIP path == the unmodified reference (value, items, gradients)."""
    from yolov6_amd.models.losses.loss_fuseab import ComputeLoss
    from yolov6_amd.utils import synth
    g = np.load(os.path.join(GOLDEN, f"lossab_{case}.npz"))
    m = json.loads(str(g["meta"]))
    inp = synth.synth_loss_inputs_ab(m["B"], m["feat_sizes"], m["strides"], m["C"], seed=m["seed"])
    crit = ComputeLoss(fpn_strides=m["strides"], num_classes=m["C"], ori_img_size=inp["img"], warmup_epoch=0, use_dfl=False, reg_max=0,
                       iou_type=m["iou_type"])
    feats = [torch.zeros(m["B"], 1, h, w, device=DEV) for h, w in m["feat_sizes"]]
    ps = inp["pred_scores"].to(DEV).requires_grad_(True)
    pd = inp["pred_distri"].to(DEV).requires_grad_(True)
    loss, items = crit((feats, ps, pd), inp["targets"].to(DEV), 10, 1, inp["img"], inp["img"])
    loss.backward()
    torch.cuda.synchronize()
    np.testing.assert_allclose(float(loss), float(g["loss"]), rtol=1e-4)
    np.testing.assert_allclose(items.cpu().numpy(), g["items"], rtol=1e-4, atol=1e-6)
    for got, name in ((ps.grad, "dscores"), (pd.grad, "ddistri")):
        ref = g[name].astype(np.float64)
        err = float(np.abs(got.cpu().numpy().astype(np.float64) - ref).max()) / max(float(np.abs(ref).max()), 1e-12)
        assert err < 1e-4, f"{case}: {name} deviates by {err:.3e} of its max"


def test_fuseab_training_graph_vs_oracle():
    """Model(fuse_ab=True) in training mode: the four head outputs (effidehead_fuseab.py:139) and every parameter gradient
    against TrainOracle.head_train_fuseab (pinned to the reference's goldens), same noise-floor criterion as the base model."""
    from oracle import synth
    from oracle.model_oracle import TrainOracle
    from tests.helpers import case_config, synth_sd_from_keys
    from yolov6_amd.models.yolo import build_model
    with open(os.path.join(GOLDEN, "keys_tiny_fuseab.json")) as f:
        meta = json.load(f)
    cfg, _ = case_config("tiny")
    sd = synth_sd_from_keys(meta["train"])
    x = synth.synth_images(4, 192, seed=21)
    xh = x.half().float()

    def run(amp):
        params = {k: v.clone().float().requires_grad_(True) for k, v in sd.items() if v.dtype.is_floating_point and "running_" not in k}
        orc = TrainOracle(cfg, sd, meta["num_classes"], amp_fp16=amp)
        orc.sd = {k: (params[k] if k in params else v.float()) for k, v in sd.items()}
        (xs, cab, rab, caf, raf), _ = orc.forward_train_fuseab(xh, meta["anchors_init"])
        ((cab * cab).sum() + rab.square().mean() + (caf * caf).sum() + raf.square().mean()).backward()
        return [t.detach() for t in (cab, rab, caf, raf)], {k: p.grad.detach() for k, p in params.items() if p.grad is not None}
    o32, g32 = run(False)
    o16, g16 = run(True)
    model = build_model(cfg, meta["num_classes"], "cpu", fuse_ab=True)
    model.load_state_dict(sd)
    model = model.to(DEV).train()
    (stems, cab, rab, caf, raf), _ = model(x.to(DEV).half())
    S = _backward_at_a_scale_that_fits(model, (cab * cab).sum() + rab.square().mean() + (caf * caf).sum() + raf.square().mean())
    torch.cuda.synchronize()
    for name, got, r32, r16 in zip(("cls_ab", "reg_ab", "cls_af", "reg_af"), (cab, rab, caf, raf), o32, o16):
        e, fl = _rel_l2(got.detach().cpu(), r32), _rel_l2(r16, r32)
        assert e <= 2 * fl + 2e-3, f"{name}: HIP {e:.3e} vs fp16 noise floor {fl:.3e}"
    named = dict(model.named_parameters())
    bad = []
    for k, ref in g32.items():
        if k == "detect.proj" or k.startswith("detect.proj_conv") or float(ref.norm()) == 0.0:
            continue
        fl = _rel_l2(g16[k], ref)
        if fl > 1.0:
            continue
        e = _rel_l2(named[k].grad.detach().float().cpu() / S, ref)
        if not (e <= 2 * fl + 2e-3):      # (NaN is a failure)
            bad.append((k, e, fl))
    assert not bad, f"{len(bad)} gradients above twice the fp16 floor, e.g. {bad[:4]}"
    ab = [k for k in g32 if "_ab" in k]
    assert len(ab) == 12 and all(float(named[k].grad.abs().max()) > 0 for k in ab)


def test_distill_ns_training_graph_vs_oracle():
    """Model(distill_ns=True) in training mode on the HIP path (heads/effidehead_distill_ns.py:80-103): the three head outputs and
    every parameter gradient against TrainOracle.forward_train_distill_ns (pinned to the reference's training-mode golden on the
    CPU), at the fp16 noise floor; then one self-distillation step: loss_distill_ns.ComputeLoss with a teacher's outputs drives
    the native backward plan."""
    from oracle import synth
    from oracle.model_oracle import TrainOracle
    from tests.helpers import case_config, synth_sd_from_keys
    from yolov6_amd.models.losses.loss_distill_ns import ComputeLoss
    from yolov6_amd.models.yolo import build_model
    with open(os.path.join(GOLDEN, "keys_tiny_distill_ns.json")) as f:
        meta = json.load(f)
    cfg, _ = case_config("tiny")
    nc = meta["num_classes"]
    sd = synth_sd_from_keys(meta["train"])
    x = synth.synth_images(4, 192, seed=21)
    xh = x.half().float()

    def run(amp):
        params = {k: v.clone().float().requires_grad_(True) for k, v in sd.items() if v.dtype.is_floating_point and "running_" not in k}
        orc = TrainOracle(cfg, sd, nc, amp_fp16=amp)
        orc.sd = {k: (params[k] if k in params else v.float()) for k, v in sd.items()}
        (xs, c, d, l), _ = orc.forward_train_distill_ns(xh)
        ((c * c).sum() + d.square().mean() + l.square().mean()).backward()
        return [t.detach() for t in (c, d, l)], {k: p.grad.detach() for k, p in params.items() if p.grad is not None}
    o32, g32 = run(False)
    o16, g16 = run(True)
    model = build_model(cfg, nc, "cpu", distill_ns=True)
    model.load_state_dict(sd)
    model = model.to(DEV).train()
    (stems, c, d, l), _ = model(x.to(DEV).half())
    S = _backward_at_a_scale_that_fits(model, (c * c).sum() + d.square().mean() + l.square().mean())
    torch.cuda.synchronize()
    for name, got, r32, r16 in zip(("cls_scores", "reg_distri", "reg_lrtb"), (c, d, l), o32, o16):
        e, fl = _rel_l2(got.detach().cpu(), r32), _rel_l2(r16, r32)
        assert e <= 2 * fl + 2e-3, f"{name}: HIP {e:.3e} vs fp16 noise floor {fl:.3e}"
    named = dict(model.named_parameters())
    bad = []
    for k, ref in g32.items():
        if k == "detect.proj" or k.startswith("detect.proj_conv") or float(ref.norm()) == 0.0:
            continue
        fl = _rel_l2(g16[k], ref)
        if fl > 1.0:
            continue
        e = _rel_l2(named[k].grad.detach().float().cpu() / S, ref)
        if not (e <= 2 * fl + 2e-3):      # (NaN is a failure)
            bad.append((k, e, fl))
    assert not bad, f"{len(bad)} gradients above twice the fp16 floor, e.g. {bad[:4]}"
    both = [k for k in g32 if "reg_preds" in k]
    assert len(both) == 12 and all(float(named[k].grad.abs().max()) > 0 for k in both)       # reg_preds AND reg_preds_dist, 3 levels x (w, b)
    # ---- one self-distillation step (core/engine.py:153-160): the teacher's outputs are plain tensors
    h = cfg.model.head
    crit = ComputeLoss(num_classes=nc, ori_img_size=192, warmup_epoch=0, use_dfl=h.use_dfl, reg_max=h.reg_max, iou_type=h.iou_type)
    arena = c._y6_graph.arena
    arena.zero_grad()
    xs_in = x.to(DEV).half()
    with torch.no_grad():
        t_out, t_feats = model(xs_in)
        # a teacher is an ordinary model: (feats, cls_scores, reg_distri) - loss_distill_ns.py:76 takes t_outputs[-2], [-1]
        t_out = tuple(t.clone() if isinstance(t, torch.Tensor) else t for t in t_out[:3])
    outs, s_feats = model(xs_in.flip(0).contiguous())
    targets = _seam_targets(8)
    targets[:, 1] = targets[:, 1] % nc
    loss, items = crit(outs, t_out, s_feats, t_feats, targets, 5, 100, 20.0, 1, 192, 192)
    (loss * 64.0).backward()
    torch.cuda.synchronize()
    assert torch.isfinite(loss) and items.shape == (4,) and torch.isfinite(arena.grad).all()
    assert float(named["detect.reg_preds.0.weight"].grad.abs().max()) > 0 and float(named["backbone.stem.rbr_dense.conv.weight"].grad.abs().max()) > 0
    # ---- channel-wise feature distillation through the graph's gradient inlets: with the head's gradients zeroed, the backbone
    # still receives the feature term's gradient, and it equals autograd's through the same neck maps
    g_plain = arena.grad.clone()
    model2 = build_model(cfg, nc, "cpu", distill_ns=True)
    model2.load_state_dict(sd)
    model2.distill_feat = True
    model2 = model2.to(DEV).train()
    crit_f = ComputeLoss(num_classes=nc, ori_img_size=192, warmup_epoch=0, use_dfl=h.use_dfl, reg_max=h.reg_max, iou_type=h.iou_type,
                         distill_feat=True)
    outs2, s_feats2 = model2(xs_in.flip(0).contiguous())
    graph2 = outs2[1]._y6_graph
    assert graph2.feat_inlets is not None and len(graph2.feat_inlets) == 3
    graph2.arena.zero_grad()
    t_feats_t = [f.float() * 0.5 + 0.1 for f in s_feats2]                      # a "teacher" with different maps
    loss_f, items_f = crit_f(outs2, t_out, s_feats2, t_feats_t, targets, 5, 100, 20.0, 1, 192, 192)
    (loss_f * 64.0).backward()
    torch.cuda.synchronize()
    assert float(items_f[3]) > 0 and torch.isfinite(graph2.arena.grad).all()
    # the inlets hold 64 x d (w_cwd decay d_cw) / d feature map: compare with autograd of the same statement on the same maps
    import math
    decay = ((1 - math.cos(5 * math.pi / 100)) / 2) * (0.01 - 1) + 1
    for inlet, sfm, tfm in zip(graph2.feat_inlets, s_feats2, t_feats_t):
        sl = sfm.float().clone().requires_grad_(True)
        N_, C_, H_, W_ = sl.shape
        ref = torch.nn.functional.kl_div(torch.log_softmax(sl.view(N_, C_, -1), 2), torch.log_softmax(tfm.view(N_, C_, -1), 2), reduction="sum",
                                         log_target=True) / (N_ * C_)
        (ref * 10.0 * decay * 64.0).backward()
        got = inlet.to_nhwc_tensor().permute(0, 3, 1, 2).float()
        den = float(sl.grad.abs().max())
        assert float((got - sl.grad).abs().max()) <= 2e-3 * den + 1e-6, "gradient inlet of a neck map"
    dd = (graph2.arena.grad - g_plain).abs().max()
    assert float(dd) > 0


# ------------------------------------------------------------------ the reference trainer's step on the HIP model (seam)
def _seam_targets(n=12):
    g = torch.Generator().manual_seed(0)
    return torch.cat([torch.randint(0, 4, (n, 1), generator=g).float(), torch.randint(0, 80, (n, 1), generator=g).float(),
                      torch.rand((n, 2), generator=g) * 0.6 + 0.2, torch.rand((n, 2), generator=g) * 0.3 + 0.1], 1).to(DEV)


def test_reference_shaped_training_step_equals_fused_path():
    """The step yolov6/core/engine.py:142-176 runs, on the HIP model: `torch.cuda.amp.autocast`, `torch.cuda.amp.GradScaler`,
    `torch.optim.SGD` with the three parameter groups of yolov6/solver/build.py:12-21 (Nesterov, weight decay on conv weights
    only), fp32 `/255` images - three steps - against the fused path (FusedSGD + LossScaler) on a second copy of the same
    model fed the same gradients: the parameters must agree to optimizer rounding after every step."""
    from oracle import synth
    from yolov6_amd.models.losses.loss import ComputeLoss
    from yolov6_amd.solver import FusedSGD, LossScaler, param_groups
    cfg, meta, sd, model_a = _tiny_train_model("tiny")
    _, _, _, model_b = _tiny_train_model("tiny")
    model_a, model_b = model_a.to(DEV).train(), model_b.to(DEV).train()
    img_u8 = (synth.synth_images(4, 128, seed=3) * 255).round().clamp(0, 255)
    x32 = (img_u8.to(DEV).float() / 255)                       # engine.py:407-410 prepro_data: .float() / 255
    targets = _seam_targets()
    h = cfg.model.head
    lr, mom, wd, scale0 = 0.02, 0.9, 5e-4, 1024.0

    def crit():
        return ComputeLoss(num_classes=80, ori_img_size=128, warmup_epoch=0, use_dfl=h.use_dfl, reg_max=h.reg_max, iou_type=h.iou_type)
    # --- A: the reference's objects
    g_bnw, g_w, g_b = param_groups(model_a)                     # build.py:12-21
    opt_a = torch.optim.SGD(g_bnw, lr=lr, momentum=mom, nesterov=True)
    opt_a.add_param_group({'params': g_w, 'weight_decay': wd})
    opt_a.add_param_group({'params': g_b})
    scaler_a = torch.cuda.amp.GradScaler(init_scale=scale0)
    crit_a = crit()
    # --- B: the fused path, fed THE SAME gradients step by step (the loss is discontinuous in the parameters through the
    # assigner's top-k, so two free-running trajectories part ways at the first flipped assignment; what is compared here
    # is the optimizer side: unscale + inf check + three-group Nesterov SGD with its momentum buffers over three steps)
    losses_a = []
    (f, s, d), _ = model_b(x32)
    arena_b = s._y6_graph.arena
    opt_b = FusedSGD(model_b, arena_b, lr=lr, momentum=mom, weight_decay=wd)
    scaler_b = LossScaler(DEV, init_scale=scale0)
    arena_a = None
    worst = moved = 0.0
    sd0 = {k: v.to(DEV) for k, v in sd.items()}
    for i in range(3):
        with torch.cuda.amp.autocast(enabled=True):
            preds, s_featmaps = model_a(x32)
            total_loss, loss_items = crit_a(preds, targets, 10, i, 128, 128)
        scaler_a.scale(total_loss).backward()
        arena_a = model_a.__dict__["_y6_arena"]
        assert torch.isfinite(arena_a.grad).all() and float(arena_a.grad.abs().max()) > 0
        arena_b.grad.copy_(arena_a.grad)                         # same layout: same architecture, same registration order
        scaler_a.step(opt_a)
        scaler_a.update()
        opt_a.zero_grad()                                        # set_to_none=True by default: `.grad` views are re-attached
        opt_b.step(scaler_b)
        scaler_b.update()
        losses_a.append(float(total_loss))
        for (na, pa), (nb, pb) in zip(model_a.named_parameters(), model_b.named_parameters()):
            assert na == nb
            worst = max(worst, float((pa - pb).abs().max() / pb.abs().max().clamp(min=1e-6)))
            moved = max(moved, float((pb - sd0[nb]).abs().max()))
    torch.cuda.synchronize()
    assert float(scaler_a.get_scale()) == scale0 and float(scaler_b.scale) == scale0      # no step was skipped on either side
    print("reference-shaped vs fused: worst relative parameter difference", worst, "largest parameter move", moved, losses_a)
    assert losses_a[-1] < losses_a[0], losses_a
    assert moved > 1e-4
    assert worst <= 1e-5, worst
    # the parameters of A are still views of its arena (torch.optim.SGD updated the arena in place)
    arena_a = model_a.__dict__["_y6_arena"]
    assert all(p.data_ptr() == arena_a.data.data_ptr() + 4 * arena_a.offset_of(p) for p in arena_a.params)


def test_checkpoint_paths_after_training_steps(tmp_path):
    """What the reference does at the end of every epoch (core/engine.py:192-203): `deepcopy(de_parallel(model)).half()`,
    `torch.save({'model': ..., 'ema': ...})`, and evaluating the model in eval mode between training epochs - after train-mode
    forwards / backwards have populated the native state (`_y6_train_graphs`, `_y6_arena`, ctypes plan handles).  Also the
    staleness rules: eval after a natively-updated step must see the NEW parameters; `.half()` on the live model must not
    leave a training graph pointing at the old storage."""
    import copy
    from oracle import synth
    from yolov6_amd.models.losses.loss import ComputeLoss
    from yolov6_amd.solver import ArenaEMA, FusedSGD, LossScaler
    cfg, meta, sd, model = _tiny_train_model("tiny")
    model = model.to(DEV)
    x = synth.synth_images(2, 64, seed=5).to(DEV).half()
    targets = _seam_targets(6)
    targets[:, 0] = targets[:, 0] % 2
    h = cfg.model.head
    crit = ComputeLoss(num_classes=80, ori_img_size=64, warmup_epoch=0, use_dfl=h.use_dfl, reg_max=h.reg_max, iou_type=h.iou_type)
    model.eval()
    det0 = model(x)[0].clone()                                   # caches an eval plan with the initial weights
    assert torch.isfinite(det0).all()
    model.train()
    (f, s, d), _ = model(x)
    arena = s._y6_graph.arena
    # (a small step: with lr 0.05 three steps move these random weights by O(1) while the running statistics have followed the
    #  batch by 9 % only - eval mode, which normalises with the RUNNING statistics, then overflows fp16 on any implementation)
    opt, scaler, ema = FusedSGD(model, arena, lr=0.0002), LossScaler(DEV, init_scale=256.0), ArenaEMA(model, arena)
    x_first = x.clone()
    for i in range(2):
        opt.zero_grad()
        (f, s, d), _ = model(x if i == 0 else x.flip(0).contiguous())
        loss, _ = crit((f, s, d), targets, 10, i, 64, 64)
        scaler.scale_loss(loss).backward()
        opt.step(scaler)
        scaler.update()
        ema.update()
        assert torch.isfinite(loss).all() and torch.isfinite(arena.grad).all() and torch.isfinite(arena.data).all(), (i, float(loss))
        bad = [n for n, b in model.named_buffers() if not torch.isfinite(b.float()).all()]
        assert not bad, (i, bad[:5])
    assert torch.equal(x, x_first)                               # the caller's first batch is not the graph's staging buffer
    # eval between epochs: same shape as the cached plan, parameters changed by native kernels only
    model.eval()
    det1 = model(x)[0].clone()
    assert not torch.equal(det0, det1), "eval after a fused SGD step served the stale plan (packed weights of the initial model)"
    # ... and it is the eval forward of the parameters / running statistics the native kernels left behind: the fp16-emulating
    # oracle on the model's CURRENT state_dict (bounds of tests/test_gpu_model.py)
    from oracle.model_oracle import Oracle
    sd1 = {k: v.detach().float().cpu().clone() for k, v in model.state_dict().items()}
    with torch.no_grad():
        ref1, _ = Oracle(cfg, sd1, 80, emulate_fp16=True).forward(x.float().cpu())
        ref32, _ = Oracle(cfg, sd1, 80, emulate_fp16=False).forward(x.float().cpu())
    assert torch.isfinite(ref1).all(), "the test's own configuration overflows fp16 in eval mode"
    assert torch.isfinite(det1).all(), [int(v) for v in (~torch.isfinite(det1)).nonzero()[0]]
    rel = lambda a, b: (a - b).abs() / b.abs().clamp(min=1.0)
    e1, fl = rel(det1.float().cpu(), ref32), rel(ref1, ref32)
    # two training steps on random weights leave running statistics that do not match the activations any more: eval-mode
    # activations grow by orders of magnitude and fp16 storage noise with them.  The scale is the reference's own fp16
    # deviation from its fp32 result on this state_dict (the bar of tests/test_gpu_model.py).
    print("eval after native training steps, error vs the fp32 oracle on the new state_dict: HIP scores", float(e1[..., 5:].max()), "all",
          float(e1.max()), "| fp16-emulating oracle scores", float(fl[..., 5:].max()), "all", float(fl.max()),
          "| largest |det|", float(ref32.abs().max()))
    assert float(e1[..., 5:].max()) <= 2.0 * float(fl[..., 5:].max()) + 1e-3, (float(e1[..., 5:].max()), float(fl[..., 5:].max()))
    assert float(e1.max()) <= 3.0 * float(fl.max()) + 1e-2, (float(e1.max()), float(fl.max()))
    fresh = copy.deepcopy(model)
    assert not any(k.startswith("_y6_") for m in fresh.modules() for k in m.__dict__)
    assert torch.allclose(fresh(x)[0], det1, atol=2e-3, rtol=2e-3)          # (the autotuner may pick other kernel variants)
    # checkpoint
    ck = copy.deepcopy(model).half()
    assert all(p.dtype == torch.float16 and p._base is None for p in ck.parameters())       # plain tensors, not arena views
    em = ema.ema_module(model).half()
    path = str(tmp_path / "ckpt.pt")
    torch.save({'model': ck, 'ema': em, 'updates': ema.updates}, path)
    back = torch.load(path, map_location=DEV, weights_only=False)
    for (n1, p1), (n2, p2) in zip(ck.named_parameters(), back['model'].named_parameters()):
        assert n1 == n2 and torch.equal(p1, p2)
    assert back['ema'].float().eval()(x.float())[0].shape == det1.shape
    # the LIVE model too (torch.save of a model holding native state must not choke on ctypes handles)
    torch.save(model, str(tmp_path / "live.pt"))
    # .half() on the live model moves every parameter: the next training forward must rebuild, not reuse stale pointers
    model.train()
    model.float()
    (f2, s2, d2), _ = model(x)
    g2 = s2._y6_graph
    assert g2.arena is not arena
    assert all(p.data_ptr() == g2.arena.data.data_ptr() + 4 * g2.arena.offset_of(p) for p in g2.arena.params)
    torch.cuda.synchronize()
    assert torch.isfinite(s2).all()


def test_two_fresh_processes_train_to_the_same_bits():
    """Reproducibility of the training path (reference yolov6/core/engine.py:142-176 runs the same step on every rank): two
    fresh processes, the same seeds, ten steps of YOLOv6-N 192^2 b4 - forward, TAL, loss, backward, SGD, loss scale.  With the
    kernel variants derived from the layer shapes (the default of the training plans; an autotuned plan picks by timing, per
    process) and no order-dependent float atomics on the path (SPPF pool backward: 25 conflict-free rounds), every loss must
    agree bit for bit and the variant tables must hash the same."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    runs = []
    for _ in range(2):
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--mode", "train", "--model", "yolov6n", "--size", "192",
                            "--batch", "4", "--steps", "4", "--warmup", "6"], capture_output=True, text=True, timeout=600, cwd=root)
        assert r.returncode == 0, r.stderr[-2000:]
        runs.append(json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]))
    a, b = runs
    assert a["variants"]["chosen_by"] == "layer shape"
    assert a["variants"] == b["variants"], (a["variants"], b["variants"])
    assert len(a["loss"]["bits"]) == 7
    assert a["loss"]["bits"] == b["loss"]["bits"], (a["loss"]["bits"], b["loss"]["bits"])
