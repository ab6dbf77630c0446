"""Training-mode FORWARD of the conv + BatchNorm blocks on the HIP path (first step).

    conv_module_train_forward(m, x)   ConvModule in .train() mode: conv -> BatchNorm(batch statistics) -> act
                                      (reference yolov6/layers/common.py:44-49)
    repvgg_train_forward(m, x)        RepVGGBlock training form: ReLU(bn(conv3x3) + bn(conv1x1) + bn_id(x))
                                      (reference common.py:250-255)

The convolutions are the inference kernels (no bias, no activation), the statistics come from y6_bn_stats and the
normalise + branch-sum + activation is ONE y6_bn_apply pass.  Like torch's BatchNorm2d in training mode, the modules'
running_mean / running_var / num_batches_tracked are updated in place (momentum from the module, unbiased variance in
the running estimate).  FORWARD ONLY: no autograd graph is recorded, so these functions refuse to run with gradients
enabled (the backward pass is the next row of DESIGN.md §9); use them under torch.no_grad().
"""
import ctypes as C

import torch

from .. import _lib
from ..engine import ACT_BY_NAME, NCHWInput, PlanBuilder, TRef


def _require_no_grad(x):
    if torch.is_grad_enabled():
        raise NotImplementedError(
            "yolov6_amd: the training-mode forward on the HIP path records no autograd graph yet (forward only, "
            "DESIGN.md §9) - call it under torch.no_grad()")
    _lib.require_gpu_tensor(x, "input")


def _run_convs(x, specs):
    """One plan: NCHW x -> NHWC view, then every (weight, stride) conv of `specs` without bias / activation."""
    pb = PlanBuilder(x.device)
    xr = pb.as_nhwc(NCHWInput(x.contiguous()))
    outs = [pb.conv(xr, w, None, stride=s, act=None) for w, s in specs]
    plan = pb.finalize(None, autotune=False)
    plan.run()
    return plan, xr, outs


def _bn_stats(t: TRef):
    lib = _lib.load()
    dev = t.buf.device
    mean = torch.empty(t.C, dtype=torch.float32, device=dev)
    var = torch.empty(t.C, dtype=torch.float32, device=dev)
    ws = torch.empty(int(lib.y6_bn_stats_workspace_bytes(t.C)), dtype=torch.uint8, device=dev)
    ct = t.ct()
    _lib.check(lib.y6_bn_stats(C.byref(ct), C.c_void_p(mean.data_ptr()), C.c_void_p(var.data_ptr()),
                               C.c_void_p(ws.data_ptr()), ws.numel(), _lib.current_stream_ptr()), "bn_stats")
    return mean, var


def _scale_shift_and_update(bn, mean, var, n):
    """Per-channel affine of a training-mode BatchNorm2d and the in-place update of its running estimates."""
    eps = bn.eps
    gamma = bn.weight.detach().float() if bn.weight is not None else torch.ones_like(mean)
    beta = bn.bias.detach().float() if bn.bias is not None else torch.zeros_like(mean)
    scale = gamma / torch.sqrt(var + eps)
    shift = beta - mean * scale
    if bn.track_running_stats and bn.running_mean is not None:
        bn.num_batches_tracked += 1
        m = bn.momentum if bn.momentum is not None else 1.0 / float(bn.num_batches_tracked)
        unbiased = var * (n / max(n - 1, 1))
        bn.running_mean.mul_(1 - m).add_(mean.to(bn.running_mean.dtype), alpha=m)
        bn.running_var.mul_(1 - m).add_(unbiased.to(bn.running_var.dtype), alpha=m)
    return scale.contiguous(), shift.contiguous()


def _bn_apply(branches, act, like: TRef):
    """branches: [(TRef, scale, shift)]; returns the NCHW result in fp16."""
    lib = _lib.load()
    dev = like.buf.device
    out = torch.empty((like.B, like.H, like.W, like.C), dtype=torch.float16, device=dev)
    d = _lib.BnApplyDesc()
    d.n = len(branches)
    for i, (t, sc, sh) in enumerate(branches):
        d.x[i] = t.ct()
        d.scale[i] = sc.data_ptr()
        d.shift[i] = sh.data_ptr()
    d.out = TRef(out, like.B, like.H, like.W, like.C, like.C, 0).ct()
    d.act = ACT_BY_NAME[act]
    _lib.check(lib.y6_bn_apply(C.byref(d), _lib.current_stream_ptr()), "bn_apply")
    return out.permute(0, 3, 1, 2).contiguous()


def conv_module_train_forward(m, x):
    """ConvModule.forward in training mode (common.py:44-49), values only."""
    _require_no_grad(x)
    if not hasattr(m, "bn"):
        raise RuntimeError("conv_module_train_forward: the BatchNorm of this ConvModule has been fused away")
    plan, _, (y,) = _run_convs(x, [(m.conv.weight, m.conv.stride[0])])
    mean, var = _bn_stats(y)
    sc, sh = _scale_shift_and_update(m.bn, mean, var, y.B * y.H * y.W)
    return _bn_apply([(y, sc, sh)], m.activation_type, y)


def repvgg_train_forward(m, x):
    """RepVGGBlock.forward of the un-fused block in training mode (common.py:250-255), values only."""
    _require_no_grad(x)
    if getattr(m, "deploy", False) or not hasattr(m, "rbr_dense"):
        raise RuntimeError("repvgg_train_forward: the block is in deploy form")
    s = m.rbr_dense.conv.stride[0]
    plan, xr, (d3, d1) = _run_convs(x, [(m.rbr_dense.conv.weight, s), (m.rbr_1x1.conv.weight, s)])
    n = d3.B * d3.H * d3.W
    branches = []
    for t, bn in ((d3, m.rbr_dense.bn), (d1, m.rbr_1x1.bn)):
        mean, var = _bn_stats(t)
        sc, sh = _scale_shift_and_update(bn, mean, var, n)
        branches.append((t, sc, sh))
    if m.rbr_identity is not None:
        mean, var = _bn_stats(xr)
        sc, sh = _scale_shift_and_update(m.rbr_identity, mean, var, xr.B * xr.H * xr.W)
        branches.append((xr, sc, sh))
    return _bn_apply(branches, "relu", d3)
