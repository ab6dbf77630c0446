"""Training-step builder: lowers the module tree in TRAIN form (batch-statistics BatchNorm, un-fused RepVGG branches,
Detect training branch) into TWO native plans of HIP kernels - forward and backward - over NHWC fp16 activations, fp32
master parameters and an fp32 gradient arena.

Replaces, for the reference's training step (yolov6/core/engine.py:142-176):
    preds = model(images)                      -> TrainGraph.forward      (one native plan)
    scaler.scale(total_loss).backward()        -> TrainGraph.backward     (one native plan; parameter gradients are
                                                                           accumulated straight into `p.grad`, which are
                                                                           views of ONE flat arena - the DDP all-reduce
                                                                           and the fused SGD run over that arena)
Leaf modules (ConvModule, RepVGGBlock, Transpose, SPPF pools, Detect) call the builder below from their ordinary
`lower()`; the composite modules (RepBlock, BepC3, BiFusion, backbones, necks) are shared with the inference lowering.

Per conv the backward emits (see yolov6_amd/csrc/wgrad.hip, train.hip):
    data gradient     the forward MFMA conv kernel on the flipped / transposed weights (stride 2: over the zero-inserted
                      output gradient, which the BatchNorm backward writes directly in dilated form)
    weight gradient   channel-major transposes of x and dy + the tap-table MFMA GEMM, fp32 atomics into the arena
"""
import os
import ctypes as C
from dataclasses import dataclass, field
from typing import List, Optional

import torch
import torch.nn as nn

from . import _lib
from ._lib import ACT_BY_NAME, Y6_F16, Y6_F32
from .engine import NCHWInput, Plan, TRef, _dtype_tag, _null_tensor


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _rup(v, m):
    return (v + m - 1) // m * m


# ---------------------------------------------------------------------------------------------------- parameter arena
class ParamArena:
    """All trainable parameters of a model as views of ONE flat fp32 tensor, their gradients as views of another.

    `p.data` / `p.grad` are re-pointed (values preserved), so state_dict / optimizers / GradScaler keep working on the
    ordinary Parameter objects while the DDP all-reduce and the fused SGD see two contiguous arrays.  Parameters are laid
    out in REVERSE registration order: the backward pass finishes gradients roughly in that order, so leading chunks of
    the arena can be all-reduced while the rest of the backward still runs."""

    def __init__(self, model: nn.Module, device=None):
        params = [p for p in model.parameters() if p.requires_grad]
        seen, uniq = set(), []
        for p in reversed(params):
            if id(p) not in seen:
                seen.add(id(p))
                uniq.append(p)
        self.params = uniq
        device = device or uniq[0].device
        offs, n = [], 0
        for p in uniq:
            offs.append(n)
            n += _rup(p.numel(), 4)              # 16-byte aligned views (float4 loads of BN vectors)
        self.offsets = offs
        self.numel = n
        self.data = torch.zeros(n, dtype=torch.float32, device=device)
        self.grad = torch.zeros(n, dtype=torch.float32, device=device)
        for p, o in zip(uniq, offs):
            v = self.data[o:o + p.numel()].view(p.shape)
            v.copy_(p.data.to(device=device, dtype=torch.float32))
            p.data = v
            p.grad = self.grad[o:o + p.numel()].view(p.shape)
        self.index = {id(p): i for i, p in enumerate(uniq)}

    def offset_of(self, p):
        return self.offsets[self.index[id(p)]]

    def grad_ptr(self, p):
        return C.c_void_p(self.grad.data_ptr() + 4 * self.offset_of(p))

    def data_ptr(self, p):
        return C.c_void_p(self.data.data_ptr() + 4 * self.offset_of(p))

    def zero_grad(self):
        self.grad.zero_()

    def reattach(self):
        """`optimizer.zero_grad(set_to_none=True)` drops the views: point `.grad` back at the arena."""
        for p, o in zip(self.params, self.offsets):
            if p.grad is None or p.grad.data_ptr() != self.grad.data_ptr() + 4 * o:
                p.grad = self.grad[o:o + p.numel()].view(p.shape)


# ---------------------------------------------------------------------------------------------------- records
@dataclass
class BnStats:
    module: Optional[nn.BatchNorm2d]
    scale: torch.Tensor
    shift: torch.Tensor
    mean: torch.Tensor
    invstd: torch.Tensor
    c0: int = 0               # first channel of the BatchNorm these vectors belong to (channel slices: MBLABlock's cv1)

    def slice(self, c0: int, n: int) -> "BnStats":
        """The statistics of channels [c0, c0 + n): views of the same vectors (BatchNorm is per channel)."""
        assert c0 % 4 == 0 and n % 4 == 0, "16-byte aligned statistic slices"
        return BnStats(self.module, self.scale[c0:c0 + n], self.shift[c0:c0 + n], self.mean[c0:c0 + n], self.invstd[c0:c0 + n],
                       self.c0 + c0)


@dataclass
class ConvRec:
    x: object                 # TRef or NCHWInput (stem)
    y: TRef
    weight: nn.Parameter
    bias: Optional[nn.Parameter]
    k: int
    stride: int
    dy: Optional[TRef] = None     # gradient wrt y as the backward sees it (dilated for stride 2)
    dy_dil: int = 1
    cpad: int = 0                 # channels of the (8-padded) dy view


class TrainBuilder:
    """PlanBuilder-compatible facade (as_nhwc / new_buffer / sppf_pool / convt2x2 / head ops) that records a tape."""
    is_train = True

    def __init__(self, device, arena: ParamArena):
        self.lib = _lib.load()
        self.device = torch.device(device)
        self.arena = arena
        self.fwd = C.c_void_p(self.lib.y6_plan_create())
        self.bwd = C.c_void_p(self.lib.y6_plan_create())
        self.keep: List[torch.Tensor] = []
        self.tape = []
        self.pack_jobs = []          # (src_ptr, dst tensor, kind, Cout, Cin, K)
        self.gbuf = {}               # forward buffer data_ptr -> gradient buffer
        self.gspans = {}             # gradient buffer data_ptr -> list[(c0, c1)] already written
        self.inputs: List[torch.Tensor] = []
        self.head_outputs = []       # (output tensor, buffer its gradient is read from), in the order the model returns them
        self.bwd_marks = []          # (bwd op index after a closure, params finalised by it)
        self.fwd_flops = 0.0
        self.bwd_flops = 0.0
        self.n_fwd_ops = 0
        self.n_bwd_ops = 0
        # partial sums of the weight-gradient slices: one workspace shared by every wgrad op (they run in stream order)
        self.wgrad_ws = torch.empty(256 << 20, dtype=torch.uint8, device=self.device) if self.device.type == "cuda" else \
            torch.empty(64 << 20, dtype=torch.uint8)
        self.keep.append(self.wgrad_ws)
        self.trace = {}              # name -> activation view (filled by the composite modules; tools/train_trace.py)
        # what every op of the forward / backward plan reads and writes, index == op index in the plan: the teacher-forced
        # per-op replay of tests/train_replay.py walks these (test infrastructure only; nothing in the step reads them)
        self.fwd_log: List[dict] = []
        self.bwd_log: List[dict] = []
        self.convs: List[ConvRec] = []     # every conv of the forward (the backward shares wgrad operand planes between them)
        self.planes = {}                   # (view key, sy, sx, oy, ox, R, Q, C) -> transposed copy already in the backward plan
        self.share_planes = os.environ.get("Y6_NO_SHARED_PLANES") is None
        self.wgrad_nhwc = os.environ.get("Y6_WGRAD_PLANES") is None     # A/B: weight gradients of stride-1 convs from NHWC (LDS transpose reads)
        # round 6: every stride-1 conv reads NHWC (csrc/wgrad_flat.hip takes the narrow maps the row-ring kernel lost to the
        # plane-fed one; the library routes per shape); a minimum width > 0 sends narrower maps back to the plane-fed kernel (A/B)
        self.wgrad_nhwc_minw3 = int(os.environ.get("Y6_WGRAD_NHWC_MINW3", "0"))
        self.wgrad_nhwc_minw1 = int(os.environ.get("Y6_WGRAD_NHWC_MINW1", "0"))
        self.wgrad_flat_s2 = os.environ.get("Y6_WGRAD_FLAT_S2", "1") != "0"      # A/B: stride-2 convs back on the plane-fed kernel
        # round 6: the stem block's weight gradients (3x3 s2 + 1x1 s2 over the NCHW image) as one pass of csrc/wgrad_stem.hip
        self.wgrad_stem = os.environ.get("Y6_WGRAD_STEM", "1") != "0"            # A/B: back on the plane-fed kernel + its transposes
        self.stem_recs: List[ConvRec] = []
        # round 6: the data gradient of the stride-2 convs from the COMPACT gradient (csrc/dgrad_s2.hip: four parity classes, the
        # 1x1 branch of a RepVGG block in the same launch) instead of the stride-1 conv over a zero-inserted one
        self.dgrad_s2 = os.environ.get("Y6_DGRAD_S2", "1") != "0"              # A/B: zero-inserted dy + the stride-1 kernels

    # ------------------------------------------------------------------ memory
    def new_buffer(self, B, H, W, C_, zero=False) -> TRef:
        t = (torch.zeros if zero else torch.empty)((B, H, W, C_), dtype=torch.float16, device=self.device)
        self.keep.append(t)
        return TRef(t, B, H, W, C_, C_, 0)

    def f32(self, n, zero=False):
        t = (torch.zeros if zero else torch.empty)(_rup(n, 4), dtype=torch.float32, device=self.device)
        self.keep.append(t)
        return t

    def bytes_(self, n):
        t = torch.zeros(_rup(n, 16), dtype=torch.uint8, device=self.device)
        self.keep.append(t)
        return t

    def as_nhwc(self, x):
        if isinstance(x, TRef):
            return x
        raise RuntimeError("yolov6_amd: in training form only the stem reads the caller's NCHW tensor")

    def input(self, t: torch.Tensor) -> TRef:
        """An NCHW tensor entering the graph as an NHWC fp16 activation (block-level graphs: ModuleTrainGraph)."""
        B, C_, H, W = t.shape
        out = self.new_buffer(B, H, W, C_)
        ct = out.ct()
        self._f(self.lib.y6_plan_add_nchw2nhwc(self.fwd, C.c_void_p(t.data_ptr()), _dtype_tag(t), C.byref(ct)), "plan_add_nchw2nhwc", x=t, out=out)
        self.inputs.append(t)
        return out

    # ------------------------------------------------------------------ gradient bookkeeping
    def grad(self, t: TRef) -> TRef:
        key = t.buf.data_ptr()
        g = self.gbuf.get(key)
        if g is None:
            g = torch.empty_like(t.buf)
            self.keep.append(g)
            self.gbuf[key] = g
        return TRef(g, t.B, t.H, t.W, t.C, t.cstride, t.coff)

    def grad_mode(self, g: TRef) -> int:
        """0 if this is the first write to the span (overwrite), 1 if the op must accumulate.  Marks it written."""
        spans = self.gspans.setdefault(g.buf.data_ptr(), [])
        lo, hi = g.coff, g.coff + g.C
        for (a, b) in spans:
            if a <= lo and hi <= b:           # inside a span that was already written (e.g. by the conv over a concat)
                return 1
            if a < hi and lo < b:
                raise RuntimeError("yolov6_amd: gradient spans overlap partially (unsupported concat pattern)")
        spans.append((lo, hi))
        return 0

    def grad_ready(self, g: TRef):
        spans = sorted(self.gspans.get(g.buf.data_ptr(), []))
        pos = g.coff
        for a, b in spans:
            if a <= pos < b:
                pos = b
        if pos < g.coff + g.C:
            raise RuntimeError("yolov6_amd: a gradient is read before every consumer wrote it (backward order bug)")

    # ------------------------------------------------------------------ plan plumbing
    def _f(self, rc, what, **log):
        _lib.check(rc, what)
        self.n_fwd_ops += 1
        self.fwd_log.append(dict(kind=what.replace("plan_add_", ""), **log))

    # weight-gradient work: ordered behind everything before it, feeds only the optimizer step -> the plan's side stream
    # (include/yolov6_hip.h y6_plan_mark_side).  Their inputs are complete when they are emitted (grad_ready asserts it for
    # gradients; activations and operand planes do not change during the backward), every buffer of the graph is its own
    # allocation, and the workspaces they share (wgrad_ws) are shared among side ops only.
    _SIDE_OPS = ("plan_add_wgrad_transpose", "plan_add_wgrad", "plan_add_wgrad_stem", "plan_add_channel_sum")

    def _b(self, rc, what, **log):
        _lib.check(rc, what)
        if what in self._SIDE_OPS:
            _lib.check(self.lib.y6_plan_mark_side(self.bwd), "plan_mark_side")
        self.n_bwd_ops += 1
        entry = dict(kind=what.replace("plan_add_", ""), side=what in self._SIDE_OPS, **log)
        self.bwd_log.append(entry)
        # the side-stream contract, checked as the plan is built (schedule.side_conflicts: a main-stream op must not write what an
        # earlier side op reads, nor touch what it writes): a future in-place backward op or a reused buffer fails HERE, not as a
        # race under load.  Ops whose access list this file does not know are skipped (tests/test_host_cpu.py requires all known).
        if not entry["side"]:
            from . import schedule as _S
            bad = _S.side_conflicts(self.bwd_log, self.arena, only_last=True)
            if bad:
                i, j, why = bad[0]
                raise RuntimeError(f"yolov6_amd: backward op {j} ({self.bwd_log[j]['kind']}) {why} of side-stream op {i} "
                                   f"({self.bwd_log[i]['kind']}): the weight-gradient side stream contract is broken")

    def _add_pack(self, src_ptr, kind, Cout, Cin, K):
        n = int(self.lib.y6_pack_job_elems(kind, Cout, Cin, K))
        if kind == 4:
            dst = torch.zeros(n, dtype=torch.float32, device=self.device)
        else:
            dst = torch.empty(n, dtype=torch.float16, device=self.device)
        self.keep.append(dst)
        self.pack_jobs.append((src_ptr, dst, kind, Cout, Cin, K, n))
        return dst

    def _conv_op(self, plan, x: TRef, out: TRef, packed, k, stride, bias_ptr=None, res: Optional[TRef] = None, log=None):
        d = _lib.ConvDesc()
        d.inp, d.out = x.ct(), out.ct()
        d.w_packed = _ptr(packed)
        d.w_oihw = None
        d.bias = bias_ptr
        d.post_scale = d.post_shift = None
        d.res = res.ct() if res is not None else _null_tensor()
        d.res_alpha = None
        d.ksize, d.stride, d.act, d.variant = k, stride, 0, -1
        rc = self.lib.y6_plan_add_conv(plan, C.byref(d))
        flops = 2.0 * out.B * out.H * out.W * out.C * x.C * k * k
        log = dict(log or {}, x=x, out=out, k=k, stride=stride, acc=res is not None)
        if plan is self.fwd:
            self._f(rc, "plan_add_conv", **log)
            self.fwd_flops += flops
        else:
            self._b(rc, "plan_add_conv", **log)
            self.bwd_flops += flops

    # ------------------------------------------------------------------ forward ops
    def conv(self, x, weight: nn.Parameter, stride: int, bias: Optional[nn.Parameter] = None) -> TRef:
        """Raw convolution y = conv(x, W) (+ bias for the prediction convs); W is the fp32 master parameter."""
        Cout, Cin, K, _ = weight.shape
        if K not in (1, 3) or stride not in (1, 2):
            raise NotImplementedError("yolov6_amd: training convs are 1x1 / 3x3, stride 1 / 2")
        wptr = self.arena.data_ptr(weight)
        if isinstance(x, NCHWInput):
            return self._stem_conv(x, weight, stride)
        pad = K // 2
        Ho, Wo = (x.H + 2 * pad - K) // stride + 1, (x.W + 2 * pad - K) // stride + 1
        y = self.new_buffer(x.B, Ho, Wo, Cout)
        packed = self._add_pack(wptr, 0, Cout, Cin, K)
        bptr = self.arena.data_ptr(bias) if bias is not None else None
        if K == 1 and stride == 2:      # the MFMA 1x1 kernels are stride-1 GEMMs: sample x[2y, 2x] first
            xs = self.new_buffer(x.B, Ho, Wo, Cin)
            ca, cb = x.ct(), xs.ct()
            self._f(self.lib.y6_plan_add_subsample2(self.fwd, C.byref(ca), C.byref(cb)), "plan_add_subsample2", x=x, out=xs)
            self._conv_op(self.fwd, xs, y, packed, 1, 1, bias_ptr=bptr, log=dict(role="fwd", weight=weight, bias=bias))
        else:
            self._conv_op(self.fwd, x, y, packed, K, stride, bias_ptr=bptr, log=dict(role="fwd", weight=weight, bias=bias))
        rec = ConvRec(x, y, weight, bias, K, stride)
        y._conv = rec
        self.convs.append(rec)
        self.tape.append(lambda: self._conv_backward(rec))
        return y

    def _stem_conv(self, x: NCHWInput, weight, stride):
        t = x.t
        _lib.require_gpu_tensor(t, "input") if self.device.type == "cuda" else None
        if not t.is_contiguous():
            raise RuntimeError("yolov6_amd: input must be a contiguous NCHW tensor")
        B, Cin, H, W = t.shape
        Cout, _, K, _ = weight.shape
        if stride != 2 or Cin > 4:
            raise NotImplementedError("yolov6_amd: the NCHW boundary is read by the stride-2 stem only")
        Ho, Wo = (H + 2 - 3) // 2 + 1, (W + 2 - 3) // 2 + 1
        y = self.new_buffer(B, Ho, Wo, Cout)
        if t not in self.inputs:
            self.inputs.append(t)
        wptr = self.arena.data_ptr(weight)
        if K == 1:     # the 1x1 stride-2 branch of the stem block: a 3x3 kernel with only the centre tap set
            w33 = self._add_pack(wptr, 4, Cout, Cin, 1)
            wsrc = C.c_void_p(w33.data_ptr())
        else:
            wsrc = wptr
        d = _lib.StemDesc()
        d.in_nchw, d.in_dtype = C.c_void_p(t.data_ptr()), _dtype_tag(t)
        d.B, d.Cin, d.H, d.W = B, Cin, H, W
        d.out = y.ct()
        d.w_oihw_f32 = wsrc
        d.bias = d.post_scale = d.post_shift = None
        d.act = 0
        self._f(self.lib.y6_plan_add_stem(self.fwd, C.byref(d)), "plan_add_stem", x=t, out=y, weight=weight, k=K)
        self.fwd_flops += 2.0 * B * Ho * Wo * Cout * Cin * K * K
        rec = ConvRec(x, y, weight, None, K, 2)
        y._conv = rec
        self.stem_recs.append(rec)
        self.tape.append(lambda: self._conv_backward(rec))
        return y

    def _bn_desc(self, d, y: TRef, bn: nn.BatchNorm2d) -> BnStats:
        Cn = y.C
        st = BnStats(bn, self.f32(Cn), self.f32(Cn), self.f32(Cn), self.f32(Cn))
        ws = self.bytes_(int(self.lib.y6_bn_stats_workspace_bytes_for(Cn, y.B * y.H * y.W)))   # sized for this tensor
        d.x = y.ct()
        d.gamma = self.arena.data_ptr(bn.weight) if bn.weight is not None else None
        d.beta = self.arena.data_ptr(bn.bias) if bn.bias is not None else None
        track = bn.track_running_stats and bn.running_mean is not None
        d.running_mean = _ptr(bn.running_mean) if track else None
        d.running_var = _ptr(bn.running_var) if track else None
        d.num_batches_tracked = _ptr(bn.num_batches_tracked) if track else None
        d.momentum = float(bn.momentum if bn.momentum is not None else 0.1)
        d.eps = float(bn.eps)
        d.scale, d.shift, d.mean, d.invstd = (_ptr(t) for t in (st.scale, st.shift, st.mean, st.invstd))
        d.workspace, d.workspace_bytes = _ptr(ws), ws.numel()
        d.workspace_clean = 1          # bytes_() hands out zeroed memory and nothing else touches it: one launch per BatchNorm
        return st

    def bn(self, y: TRef, bn: nn.BatchNorm2d) -> BnStats:
        d = _lib.BnTrainDesc()
        st = self._bn_desc(d, y, bn)
        self._f(self.lib.y6_plan_add_bn_train_stats(self.fwd, C.byref(d)), "plan_add_bn_train_stats", x=y, bn=bn, stats=st)
        return st

    def bn_multi(self, pairs) -> List[BnStats]:
        """The statistics of up to three (tensor, BatchNorm) pairs of ONE shape as one plan op (one pair of launches instead of one
        per tensor; same bits - include/yolov6_hip.h y6_bn_train_stats_multi): the branches of a RepVGG block.  Falls back to one op
        per tensor for a single pair, for mixed shapes, with Y6_BN_MULTI=0 (A/B) and under the opt-in two-stream forward schedule
        (whose point is to run the 1x1 / identity statistics beside the 3x3 conv)."""
        pairs = list(pairs)
        shape = {(y.B, y.H, y.W, y.C) for y, _ in pairs}
        if (len(pairs) < 2 or len(pairs) > 3 or len(shape) != 1 or os.environ.get("Y6_BN_MULTI", "1") == "0"
                or os.environ.get("Y6_TRAIN_FWD_STREAMS", "1") == "2"):
            return [self.bn(y, bn) for y, bn in pairs]
        md = _lib.BnTrainMultiDesc()
        md.n = len(pairs)
        sts = [self._bn_desc(md.d[i], y, bn) for i, (y, bn) in enumerate(pairs)]
        self._f(self.lib.y6_plan_add_bn_train_stats_multi(self.fwd, C.byref(md)), "plan_add_bn_train_stats_multi",
                items=[(y, bn, st) for (y, bn), st in zip(pairs, sts)])
        return sts

    def bnact(self, branches, act, out: Optional[TRef] = None, res: Optional[TRef] = None,
              alpha: Optional[nn.Parameter] = None) -> TRef:
        """out = act(sum_b x_b*scale_b + shift_b) [+ alpha*res];  branches: [(TRef, BnStats | None)]."""
        ref = branches[0][0]
        if out is None:
            out = self.new_buffer(ref.B, ref.H, ref.W, ref.C)
        d = _lib.BnActDesc()
        d.n = len(branches)
        for i, (t, st) in enumerate(branches):
            d.x[i] = t.ct()
            d.scale[i] = st.scale.data_ptr() if st is not None else None
            d.shift[i] = st.shift.data_ptr() if st is not None else None
        d.res = res.ct() if res is not None else _null_tensor()
        d.res_alpha = self.arena.data_ptr(alpha) if alpha is not None else None
        d.out = out.ct()
        d.act = ACT_BY_NAME[act]
        self._f(self.lib.y6_plan_add_bnact_forward(self.fwd, C.byref(d)), "plan_add_bnact_forward", branches=list(branches), act=act,
                out=out, res=res, alpha=alpha)
        self.tape.append(lambda: self._bnact_backward(d, branches, out, res, alpha))
        return out

    def avgpool3(self, x: TRef, with_identity: bool) -> TRef:
        """[x +] AvgPool2d(3, 1, 1)(x): the raw identity + average-pool branches of QARepVGGBlockV2's training form
        (common.py:416-419) as one tensor.  The pooling is its own adjoint: the backward is the same op on the gradient."""
        out = self.new_buffer(x.B, x.H, x.W, x.C)
        ca, cb = x.ct(), out.ct()
        self._f(self.lib.y6_plan_add_avgpool3(self.fwd, C.byref(ca), C.byref(cb), int(with_identity), 0), "plan_add_avgpool3", x=x, out=out,
                with_identity=bool(with_identity))

        def bwd():
            gout = self.grad(out)
            self.grad_ready(gout)
            gx = self.grad(x)
            acc = self.grad_mode(gx)
            ga, gb = gout.ct(), gx.ct()
            self._b(self.lib.y6_plan_add_avgpool3(self.bwd, C.byref(ga), C.byref(gb), int(with_identity), acc), "plan_add_avgpool3", x=gout, out=gx,
                    with_identity=bool(with_identity), acc=int(acc))
        self.tape.append(bwd)
        return out

    def grad_inlet(self, t: TRef) -> TRef:
        """An external gradient source for activation `t` (the neck feature maps under channel-wise feature distillation,
        loss_distill.py:222-246): returns a zero-initialised NHWC fp16 buffer; the backward plan STARTS by copying it into the
        gradient of `t` (call this after every forward consumer of `t` has been lowered), so whoever fills it before
        `loss.backward()` reaches the native plan adds that gradient."""
        ext = self.new_buffer(t.B, t.H, t.W, t.C, zero=True)

        def bwd():
            gx = self.grad(t)
            acc = self.grad_mode(gx)
            ca, cb = ext.ct(), gx.ct()
            self._b(self.lib.y6_plan_add_tensor_add(self.bwd, C.byref(ca), C.byref(cb), acc), "plan_add_tensor_add", x=ext, out=gx, acc=int(acc))
        self.tape.append(bwd)
        return ext

    def sppf_pool(self, x: TRef, y1: TRef, y2: TRef, y3: TRef):
        cts = [t.ct() for t in (x, y1, y2, y3)]
        self._f(self.lib.y6_plan_add_sppf(self.fwd, *[C.byref(c) for c in cts]), "plan_add_sppf", x=x, outs=[y1, y2, y3])
        self.tape.append(lambda: self._sppf_backward(x, y1, y2, y3))

    def convt2x2(self, x: TRef, weight: nn.Parameter, bias: nn.Parameter, out: Optional[TRef] = None) -> TRef:
        Cin, Cout = weight.shape[0], weight.shape[1]
        if out is None:
            out = self.new_buffer(x.B, 2 * x.H, 2 * x.W, Cout)
        packed = self._add_pack(self.arena.data_ptr(weight), 2, Cout, Cin, 2)
        d = _lib.ConvTDesc()
        d.inp, d.out = x.ct(), out.ct()
        d.w_packed = _ptr(packed)
        d.bias = self.arena.data_ptr(bias)
        self._f(self.lib.y6_plan_add_convt(self.fwd, C.byref(d)), "plan_add_convt", x=x, out=out, weight=weight, bias=bias)
        self.fwd_flops += 2.0 * out.B * out.H * out.W * Cout * Cin
        self.tape.append(lambda: self._convt_backward(x, out, weight, bias))
        return out

    def head_pack(self, cls: Optional[List[TRef]], reg: List[TRef], nc: int, nreg: int):
        """Detect's training branch tail: per level sigmoid(cls) and reg, flattened and concatenated -> scores [B,A,nc], distri
        [B,A,nreg] (fp32).  cls None / nc 0: a regression-only pack (the plain-distance output of the distillation head)."""
        only_reg = cls is None
        B = reg[0].B
        A = sum(r.H * r.W for r in reg)
        scores = None if only_reg else torch.zeros((B, A, nc), dtype=torch.float32, device=self.device)
        dscores = None if only_reg else torch.zeros((B, A, nc), dtype=torch.float32, device=self.device)
        distri = torch.zeros((B, A, nreg), dtype=torch.float32, device=self.device)
        ddistri = torch.zeros((B, A, nreg), dtype=torch.float32, device=self.device)
        self.keep += [t for t in (scores, dscores, distri, ddistri) if t is not None]
        if not only_reg:
            self.scores, self.distri, self.dscores, self.ddistri = scores, distri, dscores, ddistri
        d = _lib.HeadPackDesc()
        d.n_levels = len(reg)
        for i, r in enumerate(reg):
            d.reg[i] = r.ct()
            if not only_reg:
                d.cls[i] = cls[i].ct()
        d.scores, d.distri = _ptr(scores), _ptr(distri)
        d.nc, d.nreg = (0 if only_reg else nc), nreg
        self._f(self.lib.y6_plan_add_head_pack(self.fwd, C.byref(d)), "plan_add_head_pack", cls=[] if only_reg else list(cls), reg=list(reg),
                scores=scores, distri=distri)
        self.head_outputs += ([] if only_reg else [(scores, dscores)]) + [(distri, ddistri)]
        cls_l = [] if only_reg else list(cls)

        def bwd():
            g = _lib.HeadPackDesc()
            g.n_levels = len(reg)
            for i in range(len(reg)):
                for t, slot in ([(cls_l[i], g.cls)] if cls_l else []) + [(reg[i], g.reg)]:
                    rec = t._conv
                    cp = _rup(t.C, 8)                       # conv inputs need 8-channel views: pad channels stay zero
                    buf = self.new_buffer(t.B, t.H, t.W, cp, zero=True)
                    rec.dy, rec.dy_dil, rec.cpad = buf, 1, cp
                    slot[i] = TRef(buf.buf, t.B, t.H, t.W, t.C, cp, 0).ct()
            g.scores = _ptr(scores)
            g.dscores, g.ddistri = _ptr(dscores), _ptr(ddistri)
            g.nc, g.nreg = (0 if only_reg else nc), nreg
            self._b(self.lib.y6_plan_add_head_unpack_backward(self.bwd, C.byref(g)), "plan_add_head_unpack_backward",
                    dcls=[c._conv.dy for c in cls_l], dreg=[r._conv.dy for r in reg], nc=[c.C for c in cls_l], nreg=[r.C for r in reg],
                    scores=scores, dscores=dscores, ddistri=ddistri)
        self.tape.append(bwd)
        return scores, distri

    def head_pack_ab(self, cls: List[TRef], reg: List[TRef], nc: int, na: int, anchors_init: torch.Tensor):
        """Anchor-based auxiliary branch of the fuse_ab head (effidehead_fuseab.py:110-124)."""
        B = cls[0].B
        A = sum(na * c.H * c.W for c in cls)
        scores = torch.zeros((B, A, nc), dtype=torch.float32, device=self.device)
        distri = torch.zeros((B, A, 4), dtype=torch.float32, device=self.device)
        dscores, ddistri = torch.zeros_like(scores), torch.zeros_like(distri)
        anc = anchors_init.detach().float().cpu().reshape(len(cls), na, 2)

        def desc(cls_t, reg_t):
            d = _lib.HeadAbDesc()
            d.n_levels, d.nc, d.na = len(cls), nc, na
            for i in range(len(cls)):
                d.cls[i], d.reg[i] = cls_t[i].ct(), reg_t[i].ct()
                d.reg_fwd[i] = reg[i].ct()
                for k in range(na):
                    d.anchors[(i * 3 + k) * 2], d.anchors[(i * 3 + k) * 2 + 1] = float(anc[i, k, 0]), float(anc[i, k, 1])
            d.scores, d.distri = _ptr(scores), _ptr(distri)
            d.dscores, d.ddistri = _ptr(dscores), _ptr(ddistri)
            return d
        self._f(self.lib.y6_plan_add_head_ab_pack(self.fwd, C.byref(desc(cls, reg))), "plan_add_head_ab_pack", cls=list(cls), reg=list(reg),
                scores=scores, distri=distri, na=na, anchors=anc)
        self.head_outputs += [(scores, dscores), (distri, ddistri)]

        def bwd():
            gc, gr = [], []
            for t, out in [(c, gc) for c in cls] + [(r, gr) for r in reg]:
                rec = t._conv
                cp = _rup(t.C, 8)
                buf = self.new_buffer(t.B, t.H, t.W, cp, zero=True)
                rec.dy, rec.dy_dil, rec.cpad = buf, 1, cp
                out.append(TRef(buf.buf, t.B, t.H, t.W, t.C, cp, 0))
            self._b(self.lib.y6_plan_add_head_ab_unpack_backward(self.bwd, C.byref(desc(gc, gr))), "plan_add_head_ab_unpack_backward",
                    dcls=gc, dreg=gr, reg_fwd=list(reg), scores=scores, dscores=dscores, ddistri=ddistri, na=na, anchors=anc)
        self.tape.append(bwd)
        return scores, distri

    # ------------------------------------------------------------------ backward emitters
    def _bnact_backward(self, fwd_desc, branches, out: TRef, res, alpha):
        n = len(branches)
        g = _lib.BnActBwdDesc()
        g.fwd = fwd_desc
        gout = self.grad(out)
        self.grad_ready(gout)
        g.dout = gout.ct()
        finals = []
        dx_log = []
        for i, (t, st) in enumerate(branches):
            if st is not None:
                g.mean[i], g.invstd[i] = st.mean.data_ptr(), st.invstd.data_ptr()
                bn = st.module
                if bn.weight is not None:
                    g.gamma[i] = self.arena.data_ptr(bn.weight).value + 4 * st.c0
                    g.dgamma[i] = self.arena.grad_ptr(bn.weight).value + 4 * st.c0
                    finals.append(bn.weight)
                if bn.bias is not None:
                    g.dbeta[i] = self.arena.grad_ptr(bn.bias).value + 4 * st.c0
                    finals.append(bn.bias)
            rec = getattr(t, "_conv", None)
            if rec is not None:                      # a conv output: its gradient lives in a private buffer
                if rec.stride == 2 and not isinstance(rec.x, NCHWInput) and not self._dgrad_s2_route(rec):
                    # (the stem has no data gradient, csrc/dgrad_s2.hip reads the compact one: compact dy for both)
                    xin = rec.x
                    Hd, Wd = (xin.shape[2], xin.shape[3]) if isinstance(xin, NCHWInput) else (xin.H, xin.W)
                    if Hd % 2 or Wd % 2:
                        raise NotImplementedError("yolov6_amd: stride-2 convs need even input sizes in training form")
                    dy = self.new_buffer(t.B, Hd, Wd, t.C, zero=True)     # zero-inserted: only (2y, 2x) is ever written
                    rec.dy, rec.dy_dil = dy, 2
                else:
                    dy = self.new_buffer(t.B, t.H, t.W, t.C)
                    rec.dy, rec.dy_dil = dy, 1
                rec.cpad = t.C
                g.dx[i] = dy.ct()
                g.dx_dil[i] = rec.dy_dil
                g.dx_acc[i] = 0
                dx_log.append((dy, rec.dy_dil, 0))
            else:                                    # an activation that is also read elsewhere (RepVGG identity, raw branches)
                gx = self.grad(t)
                g.dx[i] = gx.ct()
                g.dx_dil[i] = 1
                g.dx_acc[i] = self.grad_mode(gx)
                dx_log.append((gx, 1, int(g.dx_acc[i])))
        dres_log = None
        if res is not None:
            gr = self.grad(res)
            g.dres = gr.ct()
            g.dres_acc = self.grad_mode(gr)
            dres_log = (gr, int(g.dres_acc))
            if alpha is not None:
                g.dalpha = self.arena.grad_ptr(alpha).value
                finals.append(alpha)
        ws = self.bytes_(int(self.lib.y6_bnact_bwd_workspace_bytes_for(out.C, out.B * out.H * out.W)))
        g.workspace, g.workspace_bytes = _ptr(ws), ws.numel()
        g.workspace_clean = 1
        self._b(self.lib.y6_plan_add_bnact_backward(self.bwd, C.byref(g)), "plan_add_bnact_backward", branches=list(branches),
                act=[k for k, v in ACT_BY_NAME.items() if v == fwd_desc.act][0], out=out, res=res, alpha=alpha, dout=gout, dx=dx_log, dres=dres_log)
        self.bwd_marks.append((self.n_bwd_ops, finals))

    def _transpose(self, view: Optional[TRef], sy, sx, oy, ox, R, Q, Cn, B, nchw_t=None) -> torch.Tensor:
        """Channel-major sampling dst[c][b][r][q] = src(b, r*sy+oy, q*sx+ox, c) into a private fp16 buffer.  A plane that an
        earlier conv of the backward already made of the same view (RepVGG's 3x3 and 1x1 branch read one x; the head's cls and
        reg convs one stem output) is handed out again: activations do not change during the backward."""
        key = None
        if nchw_t is None and self.share_planes:
            key = (view.buf.data_ptr(), view.B, view.H, view.W, view.C, view.cstride, view.coff, sy, sx, oy, ox, R, Q, Cn)
            if key in self.planes:
                return self.planes[key]
        dst = torch.empty(Cn * B * R * Q, dtype=torch.float16, device=self.device)
        self.keep.append(dst)
        d = _lib.WgradTDesc()
        if nchw_t is not None:
            Bn, Cc, Hn, Wn = nchw_t.shape
            d.src = _lib.Tensor(C.c_void_p(nchw_t.data_ptr()), Bn, Hn, Wn, Cc, Cc, 0)
            d.nchw, d.src_dtype = 1, _dtype_tag(nchw_t)
        else:
            d.src = view.ct()
            d.nchw, d.src_dtype = 0, Y6_F16
        d.sy, d.sx, d.oy, d.ox, d.R, d.Q = sy, sx, oy, ox, R, Q
        d.dst = dst.data_ptr()
        self._b(self.lib.y6_plan_add_wgrad_transpose(self.bwd, C.byref(d)), "plan_add_wgrad_transpose", aux=True,
                src=(nchw_t if nchw_t is not None else view), dst=dst)
        if key is not None:
            self.planes[key] = dst
        return dst

    def _reads_same(self, rec: ConvRec, k: int, stride: int) -> bool:
        """Does another conv (k, stride) of the forward read exactly the view rec.x?"""
        x = rec.x
        if isinstance(x, NCHWInput):
            return False
        for o in self.convs:
            if o is rec or o.k != k or o.stride != stride or isinstance(o.x, NCHWInput):
                continue
            ox = o.x
            if (ox.buf.data_ptr(), ox.B, ox.H, ox.W, ox.C, ox.cstride, ox.coff) == (x.buf.data_ptr(), x.B, x.H, x.W, x.C, x.cstride, x.coff):
                return True
        return False

    def _wgrad(self, mode, a, planes, M, N, B, Q, rows, T, out_ptr, flops, a_ch=None, b_ch=None, log=None):
        w = _lib.WgradDesc()
        w.mode = mode
        w.a = a.data_ptr()
        w.M, w.N, w.B, w.Q, w.rows, w.a_rows = M, N, B, Q, rows, rows
        w.a_channels, w.plane_channels = a_ch or M, b_ch or N
        for i, (t, prow, drow) in enumerate(planes):
            w.plane[i] = t.data_ptr()
            w.plane_rows[i] = prow
            w.drow[i] = drow
        w.out = out_ptr
        w.sm, w.sn, w.st = N * T, T, 1
        w.flops = flops
        w.workspace, w.workspace_bytes = self.wgrad_ws.data_ptr(), self.wgrad_ws.numel()
        self._b(self.lib.y6_plan_add_wgrad(self.bwd, C.byref(w)), "plan_add_wgrad", mode=mode, a=a, planes=[t for t, _, _ in planes],
                ws=self.wgrad_ws, **(log or {}))
        self.bwd_flops += flops

    def _conv_backward(self, rec: ConvRec):
        """dW (+ db) and dx of one conv, given rec.dy."""
        x, y, K, s = rec.x, rec.y, rec.k, rec.stride
        if rec.dy is None:       # consumed directly by something that wrote grad(y)
            gy = self.grad(y)
            self.grad_ready(gy)
            rec.dy, rec.dy_dil, rec.cpad = gy, 1, y.C
        dy = rec.dy
        Cout, Cin = rec.weight.shape[0], rec.weight.shape[1]
        B, Ho, Wo = y.B, y.H, y.W
        Q = _rup(Wo, 16)
        is_stem = isinstance(x, NCHWInput)
        xt = x.t if is_stem else None
        xv = None if is_stem else x
        # A operand: dy, channel-major (a dilated gradient is sampled back with stride 2)
        dyv = TRef(dy.buf, dy.B, dy.H, dy.W, rec.cpad or dy.C, dy.cstride, dy.coff)
        flops = 2.0 * Cout * Cin * K * K * B * Ho * Wo
        wlog = dict(weight=rec.weight, x=(xt if is_stem else xv), dy=dyv, dil=rec.dy_dil, k=K, stride=s, cout=Cout)
        nhwc_s1 = s == 1 and rec.dy_dil == 1 and Wo >= (self.wgrad_nhwc_minw3 if K == 3 else self.wgrad_nhwc_minw1)
        nhwc_s2 = s == 2 and self.wgrad_flat_s2           # round 6: the flat-index kernel's parity-plane form (dy compact or zero-inserted)
        if not is_stem and self.wgrad_nhwc and (nhwc_s1 or nhwc_s2):
            # the weight gradient reads x and dy as they lie (NHWC) - no transposed copies
            w = _lib.WgradNhwcDesc()
            w.ksize, w.dy, w.x, w.M, w.N = K, dyv.ct(), xv.ct(), Cout, Cin
            w.stride = s
            w.out = self.arena.grad_ptr(rec.weight)
            w.sm, w.sn, w.st = Cin * K * K, K * K, 1
            w.flops = flops
            w.workspace, w.workspace_bytes = self.wgrad_ws.data_ptr(), self.wgrad_ws.numel()
            if self.lib.y6_wgrad_nhwc_supported(C.byref(w)):
                self._b(self.lib.y6_plan_add_wgrad_nhwc(self.bwd, C.byref(w)), "plan_add_wgrad", mode=(_lib.WG_3X3S1 if K == 3 else _lib.WG_1X1),
                        nhwc=True, ws=self.wgrad_ws, **wlog)
                self.bwd_flops += flops
                self._conv_backward_rest(rec, dyv, is_stem)
                return
        if is_stem and self._stem_wgrad(rec, dyv):
            self._conv_backward_rest(rec, dyv, is_stem)
            return
        a = self._transpose(dyv, rec.dy_dil, rec.dy_dil, 0, 0, Ho, Q, dyv.C, B)
        if s == 1 and K == 3:
            mode = _lib.WG_3X3S1
            p = self._transpose(xv, 1, 1, -1, 0, Ho + 2, Q, Cin, B, xt)      # one zero row above and below
            planes = [(p, Ho + 2, ky) for ky in range(3)]
        elif s == 1 and K == 1:
            mode = _lib.WG_1X1
            if self.share_planes and self._reads_same(rec, 3, 1):
                # a 3x3 stride-1 conv reads the same view: its plane (one zero row above and below) serves with a row offset
                planes = [(self._transpose(xv, 1, 1, -1, 0, Ho + 2, Q, Cin, B, xt), Ho + 2, 1)]
            else:
                planes = [(self._transpose(xv, 1, 1, 0, 0, Ho, Q, Cin, B, xt), Ho, 0)]
        elif s == 2 and K == 3:
            mode = _lib.WG_3X3S2
            pe = [self._transpose(xv, 2, 2, 0, cp, Ho, Q, Cin, B, xt) for cp in (0, 1)]          # even rows  x[2r][2q+cp]
            po = [self._transpose(xv, 2, 2, -1, cp, Ho + 1, Q, Cin, B, xt) for cp in (0, 1)]     # odd rows   x[2r-1][2q+cp]
            planes = []
            for ky in range(3):
                for cp in range(2):
                    planes.append((pe[cp], Ho, 0) if ky == 1 else (po[cp], Ho + 1, 0 if ky == 0 else 1))
        else:                      # 1x1 stride 2: x[2y, 2x]
            mode = _lib.WG_1X1
            planes = [(self._transpose(xv, 2, 2, 0, 0, Ho, Q, Cin, B, xt), Ho, 0)]
        self._wgrad(mode, a, planes, Cout, Cin, B, Q, Ho, K * K, self.arena.grad_ptr(rec.weight), flops, a_ch=dyv.C, b_ch=Cin, log=wlog)
        self._conv_backward_rest(rec, dyv, is_stem)

    def _s2_mate(self, rec: ConvRec, k: int) -> Optional[ConvRec]:
        """The stride-2 conv with kernel size k that reads exactly rec.x (the other branch of a RepVGG stride-2 block)."""
        x = rec.x
        key = (x.buf.data_ptr(), x.B, x.H, x.W, x.C, x.cstride, x.coff)
        for o in self.convs:
            if o is rec or o.k != k or o.stride != 2 or isinstance(o.x, NCHWInput) or o.bias is not None:
                continue
            ox = o.x
            if (ox.buf.data_ptr(), ox.B, ox.H, ox.W, ox.C, ox.cstride, ox.coff) == key:
                return o
        return None

    def _dgrad_s2_route(self, rec: ConvRec) -> bool:
        """Does csrc/dgrad_s2.hip compute this stride-2 conv's data gradient (then its dy stays compact)?  A 3x3 conv whose
        channel counts are multiples of 32 over an even-sized map; a 1x1 conv only as the partner of such a 3x3 conv over the same
        input with the same output width (its gradient is a tenth tap of the partner's launch)."""
        if not self.dgrad_s2 or rec.stride != 2 or isinstance(rec.x, NCHWInput) or rec.bias is not None:
            return False
        x, y = rec.x, rec.y
        if x.H % 2 or x.W % 2 or x.C % 32 or y.C % 32 or x.cstride % 8 or x.coff % 8 or rec.weight.shape[0] != y.C:
            return False
        if rec.k == 3:
            return True
        m = self._s2_mate(rec, 3) if rec.k == 1 else None
        return m is not None and m.y.C == y.C and self._dgrad_s2_route(m)

    def _dgrad_s2_op(self, r3: ConvRec, r1: Optional[ConvRec]):
        """dx of a stride-2 3x3 conv (+ its 1x1 partner) as one op of csrc/dgrad_s2.hip."""
        def view(r):
            return TRef(r.dy.buf, r.dy.B, r.dy.H, r.dy.W, r.cpad or r.dy.C, r.dy.cstride, r.dy.coff)
        Cout, Cin = r3.weight.shape[0], r3.weight.shape[1]
        gx = self.grad(r3.x)
        acc = self.grad_mode(gx)
        d = _lib.DgradS2Desc()
        v3 = view(r3)
        v1 = view(r1) if r1 is not None else None
        d.dy3 = v3.ct()
        d.dy1 = v1.ct() if v1 is not None else _null_tensor()
        d.dx = gx.ct()
        d.w3_packed = _ptr(self._add_pack(self.arena.data_ptr(r3.weight), 1, Cout, Cin, 3))
        d.w1_packed = _ptr(self._add_pack(self.arena.data_ptr(r1.weight), 1, Cout, Cin, 1)) if r1 is not None else None
        d.accumulate = int(acc)
        self._b(self.lib.y6_plan_add_dgrad_s2(self.bwd, C.byref(d)), "plan_add_dgrad_s2", dys=[v3, v1], out=gx, acc=acc,
                weights=[r3.weight, r1.weight if r1 is not None else None])
        self.bwd_flops += 2.0 * Cout * Cin * (9 + (1 if r1 is not None else 0)) * v3.B * v3.H * v3.W

    def _stem_wgrad(self, rec: ConvRec, dyv: TRef) -> bool:
        """The weight gradient(s) of the convs that read the NCHW image `rec.x.t` (the stem block: RepVGGBlock(3 -> C, k3 s2) in train
        form, efficientrep.py:28-41 / common.py:250-255) as ONE op of csrc/wgrad_stem.hip: the 3x3 conv and - when the block has one and
        its gradient is at hand - the 1x1 conv of the same image in the same pass.  Returns False when the plane-fed route must serve
        (A/B switch, fp32 / uint8 images, odd sizes, a lone 1x1 conv)."""
        if getattr(rec, "_stem_done", False):
            return True                      # written by its partner's op
        if not self.wgrad_stem or rec.stride != 2 or rec.dy_dil != 1:
            return False
        xt = rec.x.t
        mates = [r for r in self.stem_recs if r is not rec and r.x.t is xt and r.stride == 2 and r.k != rec.k and r.dy is not None
                 and r.dy_dil == 1 and not getattr(r, "_stem_done", False) and r.weight.shape[0] == rec.weight.shape[0]]
        r3 = rec if rec.k == 3 else (mates[0] if mates else None)
        r1 = rec if rec.k == 1 else (mates[0] if mates else None)
        if r3 is None or r3.k != 3 or (r1 is not None and r1.k != 1):
            return False

        def view(r):
            return TRef(r.dy.buf, r.dy.B, r.dy.H, r.dy.W, r.cpad or r.dy.C, r.dy.cstride, r.dy.coff)
        B, Cin, H, W = xt.shape
        Cout = r3.weight.shape[0]
        d = _lib.WgradStemDesc()
        d.x, d.in_dtype = xt.data_ptr(), _dtype_tag(xt)
        d.B, d.Cin, d.H, d.W, d.Cout = B, Cin, H, W, Cout
        v3 = view(r3)
        v1 = view(r1) if r1 is not None else None
        d.dy3 = v3.ct()
        d.dy1 = v1.ct() if v1 is not None else _null_tensor()
        d.out3 = self.arena.grad_ptr(r3.weight)
        d.out1 = self.arena.grad_ptr(r1.weight) if r1 is not None else None
        d.workspace, d.workspace_bytes = self.wgrad_ws.data_ptr(), self.wgrad_ws.numel()
        if not xt.is_contiguous() or not self.lib.y6_wgrad_stem_supported(C.byref(d)):
            return False
        self._b(self.lib.y6_plan_add_wgrad_stem(self.bwd, C.byref(d)), "plan_add_wgrad_stem", x=xt, dys=[v3, v1],
                weights=[r3.weight, r1.weight if r1 is not None else None], ws=self.wgrad_ws, cout=Cout)
        self.bwd_flops += 2.0 * Cout * Cin * (9 + (1 if r1 is not None else 0)) * B * v3.H * v3.W
        for r in (r3, r1):
            if r is not None and r is not rec:
                r._stem_done = True
        return True

    def _conv_backward_rest(self, rec: ConvRec, dyv: TRef, is_stem: bool):
        """Bias gradient and data gradient of one conv (after its weight gradient)."""
        x, y, K, s, dy = rec.x, rec.y, rec.k, rec.stride, rec.dy
        Cout, Cin = rec.weight.shape[0], rec.weight.shape[1]
        finals = [rec.weight]
        if rec.bias is not None:
            ws = self.bytes_(16 * _rup(max(y.C, 1), 8))
            ct = TRef(dy.buf, dy.B, dy.H, dy.W, y.C, dy.cstride, dy.coff).ct()
            self._b(self.lib.y6_plan_add_channel_sum(self.bwd, C.byref(ct), self.arena.grad_ptr(rec.bias), _ptr(ws), ws.numel()),
                    "plan_add_channel_sum", x=TRef(dy.buf, dy.B, dy.H, dy.W, y.C, dy.cstride, dy.coff), param=rec.bias, ws=ws)
            finals.append(rec.bias)
        # data gradient: the forward conv kernel on the flipped / transposed weights, stride 1 over the (dilated) dy -
        # or, for the stride-2 convs csrc/dgrad_s2.hip takes, one launch per block over the compact dy: emitted by whichever of
        # the two branches runs its backward LAST (both gradients are at hand then)
        if not is_stem and rec.dy_dil == 1 and self._dgrad_s2_route(rec):
            mate = self._s2_mate(rec, 1 if K == 3 else 3)
            if mate is not None and not self._dgrad_s2_route(mate):
                mate = None
            rec._s2_wgrad_done = True
            if mate is None or getattr(mate, "_s2_wgrad_done", False):
                r3, r1 = (rec, mate) if K == 3 else (mate, rec)
                self._dgrad_s2_op(r3, r1)
        elif not is_stem:
            gx = self.grad(x)
            acc = self.grad_mode(gx)
            # padded prediction-conv gradients (68 -> 72 channels): the packed image is zero beyond Cout and the
            # 32-channel chunk counts of 68 and 72 agree
            assert (dyv.C + 31) // 32 == (Cout + 31) // 32
            packed = self._add_pack(self.arena.data_ptr(rec.weight), 1, Cout, Cin, K)
            self._conv_op(self.bwd, dyv, gx, packed, K, 1, res=gx if acc else None,
                          log=dict(role="dgrad", weight=rec.weight, dil=rec.dy_dil, fwd_stride=s, fwd_k=K))
        self.bwd_marks.append((self.n_bwd_ops, finals))

    def _sppf_backward(self, x, y1, y2, y3):
        d = _lib.SppfBwdDesc()
        d.x, d.y1, d.y2 = x.ct(), y1.ct(), y2.ct()
        g1, g2, g3, gx = self.grad(y1), self.grad(y2), self.grad(y3), self.grad(x)
        for g in (g1, g2, g3, gx):
            self.grad_ready(g)
        d.dy1, d.dy2, d.dy3, d.dx = g1.ct(), g2.ct(), g3.ct(), gx.ct()
        d.dx_acc = 1
        self._b(self.lib.y6_plan_add_sppf_backward(self.bwd, C.byref(d)), "plan_add_sppf_backward", x=x, ys=[y1, y2], dys=[g1, g2, g3], dx=gx)

    def _convt_backward(self, x: TRef, out: TRef, weight, bias):
        Cin, Cout = weight.shape[0], weight.shape[1]
        gout = self.grad(out)
        self.grad_ready(gout)
        B, H, W = x.B, x.H, x.W
        Q = _rup(W, 16)
        ws = self.bytes_(16 * _rup(Cout, 8))
        ct = gout.ct()
        self._b(self.lib.y6_plan_add_channel_sum(self.bwd, C.byref(ct), self.arena.grad_ptr(bias), _ptr(ws), ws.numel()),
                "plan_add_channel_sum", x=gout, param=bias, ws=ws)
        a = self._transpose(x, 1, 1, 0, 0, H, Q, Cin, B)
        planes = [(self._transpose(gout, 2, 2, sub >> 1, sub & 1, H, Q, Cout, B), H, 0) for sub in range(4)]
        self._wgrad(_lib.WG_CONVT, a, planes, Cin, Cout, B, Q, H, 4, self.arena.grad_ptr(weight), 2.0 * Cin * Cout * 4 * B * H * W,
                    log=dict(weight=weight, x=x, dy=gout, dil=1, k=2, stride=2, cout=Cout, convt=True))
        # dx = 1x1 conv over space-to-depth(dout) with W'[ci][sub*Cout + co]
        s2d = self.new_buffer(B, H, W, 4 * Cout)
        ca, cb = gout.ct(), s2d.ct()
        self._b(self.lib.y6_plan_add_space_to_depth2(self.bwd, C.byref(ca), C.byref(cb)), "plan_add_space_to_depth2", x=gout, out=s2d)
        gx = self.grad(x)
        acc = self.grad_mode(gx)
        packed = self._add_pack(self.arena.data_ptr(weight), 3, Cout, Cin, 2)
        self._conv_op(self.bwd, s2d, gx, packed, 1, 1, res=gx if acc else None, log=dict(role="convt_dgrad", weight=weight, dy=gout))
        self.bwd_marks.append((self.n_bwd_ops, [weight, bias]))

    # ------------------------------------------------------------------ finish
    def finalize(self):
        """Emit the backward plan from the tape (reverse order) and build the per-step weight-packing table."""
        for closure in reversed(self.tape):
            closure()
        jobs = (_lib.PackJob * len(self.pack_jobs))()
        first = 0
        for j, (src, dst, kind, Cout, Cin, K, n) in enumerate(self.pack_jobs):
            jobs[j].src = src
            jobs[j].dst = dst.data_ptr()
            jobs[j].kind, jobs[j].Cout, jobs[j].Cin, jobs[j].K = kind, Cout, Cin, K
            jobs[j].first = first
            first += n
        raw = bytes(jobs)
        self.jobs_dev = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(self.device)
        self.keep.append(self.jobs_dev)
        self.pack = C.c_void_p(self.lib.y6_plan_create())
        d = _lib.PackBatchDesc()
        d.jobs, d.njobs, d.total = self.jobs_dev.data_ptr(), len(self.pack_jobs), first
        _lib.check(self.lib.y6_plan_add_pack_batch(self.pack, C.byref(d)), "plan_add_pack_batch")
        plans = [Plan(h, self.keep, None, ()) for h in (self.pack, self.fwd, self.bwd)]
        self.pack = self.fwd = self.bwd = None
        return plans


# ---------------------------------------------------------------------------------------------------- the graph
class _LazyNCHW(list):
    """NHWC plan buffers presented as the NCHW tensors the reference returns, converted on first access."""

    def __init__(self, refs, dtype):
        super().__init__()
        self._refs, self._dtype = refs, dtype

    def _fill(self):
        if self._refs is not None:
            refs, self._refs = self._refs, None
            for r in refs:
                super().append(r.to_nhwc_tensor().permute(0, 3, 1, 2).contiguous().to(self._dtype))

    def __len__(self):
        return len(self._refs) if self._refs is not None else super().__len__()

    def feat_sizes(self):
        """(H, W) per level without converting anything (ComputeLoss only needs the sizes, loss.py:63-69)."""
        if self._refs is not None:
            return [torch.Size((r.H, r.W)) for r in self._refs]
        return [t.shape[2:] for t in list.__iter__(self)]

    def __iter__(self):
        self._fill()
        return super().__iter__()

    def __getitem__(self, i):
        self._fill()
        return super().__getitem__(i)


def schedule_forward(plan, fwd_log):
    """Two-stream schedule of a training-form forward plan (yolov6_amd/schedule.py): a RepVGG block's 1x1 / identity branches
    (their convs and BatchNorm statistics passes, reference yolov6/layers/common.py:250-255) are independent of its 3x3 branch
    until the branch sum, BepC3's two 1x1 inputs of each other (:634-650) - on one stream they queue behind each other.
    Opt-in (Y6_TRAIN_FWD_STREAMS=2) until measured.  Returns the schedule summary or None."""
    if os.environ.get("Y6_TRAIN_FWD_STREAMS", "1") != "2" or plan.num_ops != len(fwd_log):
        return None
    from . import schedule as S
    return plan.schedule(costs=S.train_costs(fwd_log), accesses=[S.train_op_access(e) for e in fwd_log],
                         policy=os.environ.get("Y6_TRAIN_FWD_POLICY", "asap"))


class TrainGraph:
    """Forward + backward native plans of one model for one input shape."""

    def __init__(self, model, x: torch.Tensor):
        arena = model.__dict__.get("_y6_arena")
        if arena is None:
            arena = model.__dict__["_y6_arena"] = ParamArena(model, x.device)
        self.arena = arena
        self.model = model
        # a private staging tensor: the stem's plan op holds its address, and the caller's first batch must not be
        # overwritten by later steps (prefetch queues, plotting)
        self.input = torch.empty_like(x, memory_format=torch.contiguous_format).copy_(x)
        tb = TrainBuilder(x.device, arena)
        stems, necks, heads = model.lower_train(tb, NCHWInput(self.input))
        # channel-wise feature distillation (`model.distill_feat = True` before the first training forward): gradient inlets
        # on the neck feature maps, filled by loss_distill.ComputeLoss
        self.feat_inlets = [tb.grad_inlet(r) for r in necks] if getattr(model, "distill_feat", False) else None
        self.pack_plan, self.fwd_plan, self.bwd_plan = tb.finalize()
        self.fwd_sched = schedule_forward(self.fwd_plan, tb.fwd_log)
        self.tb = tb
        self.stem_refs, self.neck_refs = stems, necks
        # head outputs in the order the model returns them, each with the buffer the backward plan reads its gradient from
        by_ptr = {o.data_ptr(): g for o, g in tb.head_outputs}
        self.outputs = list(heads)
        self.grad_inputs = [by_ptr[o.data_ptr()] for o in self.outputs]
        self.scores, self.distri = self.outputs[-2], self.outputs[-1]            # the anchor-free pair
        self.dscores, self.ddistri = self.grad_inputs[-2], self.grad_inputs[-1]
        self.fwd_flops, self.bwd_flops = tb.fwd_flops, tb.bwd_flops
        self.bwd_marks = tb.bwd_marks
        self.fwd_log, self.bwd_log = tb.fwd_log, tb.bwd_log
        self.n_bwd_ops = self.bwd_plan.num_ops
        self.anchor = torch.zeros((), dtype=torch.float32, device=x.device, requires_grad=True)
        self.signature = model_signature(model)

    def forward(self, x):
        self.input.copy_(x)
        self.pack_plan.run()
        self.fwd_plan.run()
        return tuple(self.outputs)

    def grad_buffer_of(self, out):
        """The buffer the backward plan reads d loss / d `out` from (ComputeLoss writes there directly)."""
        for o, g in zip(self.outputs, self.grad_inputs):
            if o.data_ptr() == out.data_ptr():
                return g
        return None

    def backward(self, grads=None, first=0, last=None):
        """Run backward ops [first, last) (default: all).  `grads`: gradients wrt the head outputs, in output order
        (None entries mean zero); tensors that are not the graph's own gradient buffers are copied in."""
        if grads is not None:
            for g, buf in zip(grads, self.grad_inputs):
                if g is None:
                    buf.zero_()
                elif g.data_ptr() != buf.data_ptr():
                    buf.copy_(g)
        if first == 0:
            p0 = self.arena.params[0]
            if p0.grad is None:                 # optimizer.zero_grad(set_to_none=True): same as zeroing
                self.arena.zero_grad()
                self.arena.reattach()
        self.bwd_plan.run_range(first, self.n_bwd_ops if last is None else last)


class ModuleTrainGraph:
    """Training graph of ONE block (any HipModule) on NCHW inputs: forward, then backward from caller-supplied output
    gradients.  Test harness for the wiring of every block type against autograd (tests/test_gpu_training.py); the model
    path is TrainGraph."""

    def __init__(self, module, inputs):
        self.arena = ParamArena(module, inputs[0].device)
        tb = TrainBuilder(inputs[0].device, self.arena)
        self.inputs = [t.contiguous() for t in inputs]
        self.in_refs = [tb.input(t) for t in self.inputs]
        outs = module.lower(tb, self.in_refs[0] if len(self.in_refs) == 1 else list(self.in_refs))
        self.out_refs = [outs] if isinstance(outs, TRef) else list(outs)
        self.dout_refs = []
        for o in self.out_refs:
            g = tb.grad(o)
            tb.grad_mode(g)                   # written by the caller (set_output_grads) before the backward plan runs
            self.dout_refs.append(g)
        self.pack_plan, self.fwd_plan, self.bwd_plan = tb.finalize()
        self.fwd_sched = schedule_forward(self.fwd_plan, tb.fwd_log)
        self.tb = tb

    def forward(self):
        self.pack_plan.run()
        self.fwd_plan.run()
        return [r.to_nhwc_tensor().permute(0, 3, 1, 2).float() for r in self.out_refs]

    def backward(self, douts):
        for g, d in zip(self.dout_refs, douts):
            g.to_nhwc_tensor().copy_(d.permute(0, 2, 3, 1).to(torch.float16))
        self.bwd_plan.run()
        res = []
        for r in self.in_refs:
            g = self.tb.grad(r)
            res.append(g.to_nhwc_tensor().permute(0, 3, 1, 2).float() if self.tb.gspans.get(g.buf.data_ptr()) else None)
        return res


def model_signature(model):
    """Addresses of every parameter and buffer: the training plans hold them as raw pointers (arena slots, BatchNorm
    running statistics), so a graph is only valid while they stay where they were (`.half()`, `.to()`, `load_state_dict`
    with `assign=True`, `switch_to_deploy` on a sub-module move them without telling the root module)."""
    return tuple(t.data_ptr() for t in model.parameters()) + tuple(t.data_ptr() for t in model.buffers())


class _TrainStepFn(torch.autograd.Function):
    """Autograd bridge: `loss.backward()` reaches the native backward plan through this node.  Parameter gradients are
    accumulated by the kernels straight into `p.grad` (arena views).

    Nothing needs to flow back through autograd for that - but torch's `DistributedDataParallel` (what the reference's
    trainer wraps the model in, core/engine.py:466) learns that a gradient is ready from a hook on each parameter's
    AccumulateGrad node, which only runs if autograd delivers a gradient to the parameter.  When a process group is up and
    no native reducer (parallel.GradReducer) owns the exchange, the parameters are therefore inputs of this node and it
    returns a broadcast ZERO for each: AccumulateGrad adds it to the arena view the kernels just filled (values
    unchanged, infs / NaNs stay), the DDP hook fires, DDP all-reduces and writes the averages back in place - into the
    arena, so torch.optim.SGD and FusedSGD both see them (tests/test_dist_cpu.py wraps the module in torch DDP)."""

    @staticmethod
    def forward(ctx, graph, x, anchor, *params):
        ctx.graph = graph
        ctx.n_params = len(params)
        return tuple(o.view_as(o) for o in graph.forward(x))

    @staticmethod
    def backward(ctx, *grads):
        g = ctx.graph
        grads = [None if t is None else t.contiguous() for t in grads]
        hook = g.model.__dict__.get("_y6_backward_hook")
        if hook is not None:
            hook(g, grads)                       # e.g. GradReducer: segmented backward + all-reduce
        else:
            g.backward(grads)
        if ctx.n_params:
            g.arena.reattach()                   # `.grad` must BE the arena view before AccumulateGrad adds the zero to it
            z = g.arena.grad.new_zeros(())
            return (None, None, None) + tuple(z.expand(p.shape) for p in g.arena.params)
        return None, None, None


def _autograd_visible_params(model, graph):
    """The parameters to thread through autograd (see _TrainStepFn): all of them when this process is one rank of a
    process group and the gradient exchange is not done natively, else none."""
    if model.__dict__.get("_y6_backward_hook") is not None:
        return ()
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        return tuple(graph.arena.params)
    return ()


def run_train_graph(model, g, x):
    """One forward of a training graph under autograd (shared by Model.forward and the CPU tests' stand-in graph)."""
    if not torch.is_grad_enabled():
        return g.forward(x)
    return _TrainStepFn.apply(g, x, g.anchor, *_autograd_visible_params(model, g))


def train_forward(model, x):
    """Model.forward in training mode (reference yolov6/models/yolo.py:33-41 with Detect's training branch):
    returns [(head stem features, cls_scores [B,A,nc], reg_distri [B,A,nreg]), neck feature maps]."""
    from .layers.common import bump_native_generation
    key = (tuple(x.shape), x.dtype, x.device)
    graphs = model.__dict__.setdefault("_y6_train_graphs", {})
    g = graphs.get(key)
    if g is not None and g.signature != model_signature(model):
        # parameters / buffers moved under the graph (see model_signature): its plans point at stale memory
        graphs.clear()
        model.__dict__.pop("_y6_arena", None)
        g = None
    if g is None:
        graphs.clear()
        g = graphs[key] = TrainGraph(model, x)
    bump_native_generation()                 # the forward plan updates the BatchNorm running statistics in place
    outs = run_train_graph(model, g, x)
    for t in outs:
        t._y6_graph = g                      # lets ComputeLoss write its gradients straight into the graph's buffers
    # base head: (feats, cls_scores, reg_distri); fuse_ab head: (feats, cls_ab, reg_ab, cls_af, reg_af) - effidehead_fuseab.py:139
    necks = _LazyNCHW(g.neck_refs, x.dtype)
    necks._y6_graph = g                      # channel-wise feature distillation finds the graph's gradient inlets here
    return [(_LazyNCHW(g.stem_refs, x.dtype),) + tuple(outs), necks]
