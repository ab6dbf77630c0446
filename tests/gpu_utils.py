"""Helpers for the -m gpu tests: run single ops through the same PlanBuilder/C-ABI path the
models use, and compute the fp32 reference of a fused conv on the CPU."""
import numpy as np
import torch
import torch.nn.functional as F

from yolov6_amd import _lib
from yolov6_amd.engine import PlanBuilder, TRef

DEV = "cuda:0"


# conv kernel variants written after the last GPU visit of a round: the hard tests skip them (supports() answers no), the
# isolated probe of tests/test_gpu_families.py runs the same tests with Y6_TEST_UNSEEN=1 in a process of its own
UNSEEN_VARIANTS = set()          # round 3, visit r03a: variants 33-37 passed the op tests on hardware


def variant_names():
    lib = _lib.load()
    return [lib.y6_conv_variant_name(i).decode() for i in range(lib.y6_conv_variants())]


def rand_nhwc(B, H, W, C, cstride=None, coff=0, seed=0, scale=1.0):
    """Random fp16 NHWC buffer [B,H,W,cstride] on the GPU and the TRef of channels [coff, coff+C)."""
    g = torch.Generator().manual_seed(seed)
    cstride = cstride or C
    buf = ((torch.rand((B, H, W, cstride), generator=g) * 2 - 1) * scale).half().to(DEV)
    return TRef(buf, B, H, W, C, cstride, coff)


def nhwc_to_nchw_f32(ref: TRef):
    return ref.to_nhwc_tensor().float().cpu().permute(0, 3, 1, 2).contiguous()


def act_fn(y, act):
    return {None: lambda t: t, "relu": F.relu, "silu": F.silu, "hardswish": F.hardswish}[act](y)


def q16(t):
    return t.half().float()


def conv_reference(x_nchw, w, b, stride, act, post=None, res=None, alpha=None):
    """fp32 CPU statement of the fused op with fp16-rounded parameters (what model.half() holds), rounded to fp16 at the op
    boundaries of the reference's half-precision graph (conv | BatchNorm | activation | residual add are separate fp16
    ops there): the fused HIP epilogue rounds at the same places (yolov6_amd/csrc/common.hpp::y6_round_f16)."""
    k = w.shape[-1]
    y = F.conv2d(x_nchw, w.half().float(), None if b is None else b.half().float(), stride=stride, padding=k // 2)
    if post is not None:
        y = q16(y) * post[0].half().float().view(1, -1, 1, 1) + post[1].half().float().view(1, -1, 1, 1)
    if act in ("silu", "hardswish"):
        y = q16(y)                       # ReLU commutes with the rounding
    y = act_fn(y, act)
    if res is not None:
        a = 1.0 if alpha is None else float(alpha.half().float())
        y = q16(y) + q16(a * res)
    return y


def op_tolerance(act, with_res=False):
    """Bound of |hip - ref| / max(1, |ref|) for ONE fused op against the statement above.  Both sides round to fp16 at the
    same places; fp32 accumulation-order differences flip isolated roundings by one ulp (2^-10 relative, < 1e-3 - the
    north_star's bar, met by conv + bias (+ ReLU)).  A flipped fp16 INPUT of SiLU / hardswish moves the output by up to
    one input-ulp, which is two ulps of an output that sits one binade lower (silu(2.1) = 1.87): 2 ulp = 1.96e-3; the
    residual add rounds once more: 3 ulp."""
    ulp = 2.0 ** -10
    n = 1 + (1 if act in ("silu", "hardswish") else 0) + (1 if with_res else 0)
    return 1e-3 if n == 1 else n * ulp * 1.002


def run_conv(x: TRef, w, b, stride, act, variant, out=None, post=None, res=None, alpha=None):
    pb = PlanBuilder(DEV)
    pb.force_variant = variant
    o = pb.conv(x, w, b, stride=stride, act=act, out=out, post=post, res=res, res_alpha=alpha)
    plan = pb.finalize(o, autotune=False)
    plan.run()
    torch.cuda.synchronize()
    return o, plan


def max_rel(a, b):
    a, b = a.double(), b.double()
    return float(((a - b).abs() / b.abs().clamp(min=1.0)).max())


def supports(x: TRef, w, stride, variant, out_c=None):
    """Ask the library whether `variant` can run this conv."""
    import ctypes as C
    import os
    lib = _lib.load()
    if lib.y6_conv_variant_name(variant).decode() in UNSEEN_VARIANTS and os.environ.get("Y6_TEST_UNSEEN") != "1":
        return False
    Cout, Cin, K, _ = w.shape
    pad = K // 2
    Ho, Wo = (x.H + 2 * pad - K) // stride + 1, (x.W + 2 * pad - K) // stride + 1
    d = _lib.ConvDesc()
    d.inp = x.ct()
    d.out = _lib.Tensor(C.c_void_p(x.buf.data_ptr()), x.B, Ho, Wo, Cout, Cout, 0)
    d.w_packed = C.c_void_p(x.buf.data_ptr())
    d.w_oihw = C.c_void_p(x.buf.data_ptr())
    d.ksize, d.stride = K, stride
    return bool(lib.y6_conv_variant_supports(C.byref(d), variant))
