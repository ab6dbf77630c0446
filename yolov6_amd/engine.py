"""Plan builder: lowers the module tree (yolov6_amd.layers / .models) into a native
`y6_plan` (yolov6_amd/csrc/plan.hip) of HIP kernel launches over NHWC fp16 buffers.

torch provides device memory (buffers, packed weights) and the stream; all compute goes
through the C ABI in include/yolov6_hip.h.
"""
import ctypes as C
from dataclasses import dataclass
from typing import List, Optional

import torch

from . import _lib
from ._lib import ACT_BY_NAME, Y6_F16, Y6_F32, Y6_U8


@dataclass
class TRef:
    """View [B,H,W,C] of an NHWC fp16 buffer: channels [coff, coff+C) of `cstride`."""
    buf: torch.Tensor
    B: int
    H: int
    W: int
    C: int
    cstride: int
    coff: int = 0

    def slice(self, c0: int, n: int) -> "TRef":
        assert 0 <= c0 and c0 + n <= self.C
        return TRef(self.buf, self.B, self.H, self.W, n, self.cstride, self.coff + c0)

    def ct(self) -> _lib.Tensor:
        return _lib.Tensor(C.c_void_p(self.buf.data_ptr()), self.B, self.H, self.W, self.C, self.cstride, self.coff)

    def to_nhwc_tensor(self) -> torch.Tensor:
        """torch view [B,H,W,C] (for tests/debug)."""
        return self.buf.view(self.B, self.H, self.W, self.cstride)[..., self.coff:self.coff + self.C]


@dataclass
class NCHWInput:
    """The caller's NCHW image/feature tensor (fp16 or fp32), not yet in the NHWC domain."""
    t: torch.Tensor

    @property
    def shape(self):
        return tuple(self.t.shape)


def _dtype_tag(t: torch.Tensor) -> int:
    if t.dtype == torch.float16:
        return Y6_F16
    if t.dtype == torch.float32:
        return Y6_F32
    if t.dtype == torch.uint8:
        return Y6_U8            # the stem only: pixels enter as imgs.half() / 255 (core/evaler.py:121-123)
    raise RuntimeError(f"yolov6_amd: unsupported dtype {t.dtype} (fp16 / fp32, uint8 images at the stem)")


def _null_tensor() -> _lib.Tensor:
    return _lib.Tensor(None, 0, 0, 0, 0, 0, 0)


class Plan:
    """A finalized native plan; run() enqueues every kernel on the current stream."""

    def __init__(self, handle, keep, outputs, inputs=()):
        self._lib = _lib.load()
        self._h = handle
        self._keep = keep
        self.outputs = outputs
        self.inputs = list(inputs)   # caller tensors the first ops read (rebindable)
        self.captured = False

    def bind_inputs(self, tensors):
        """Zero-copy: point the plan at new input tensors of the same shape/dtype."""
        for i, (old, new) in enumerate(zip(self.inputs, tensors)):
            if new.data_ptr() == old.data_ptr():
                continue
            if new.shape != old.shape or new.dtype != old.dtype or not new.is_contiguous():
                raise RuntimeError("yolov6_amd: rebinding needs a contiguous tensor of the compiled shape/dtype")
            if self.captured:   # a captured graph baked the address: stage through the bound tensor
                old.copy_(new)
                continue
            # by position (the i-th boundary-reading op), never by matching the old address: a caller that swaps
            # two inputs would otherwise end with both ops on the same tensor
            rc = self._lib.y6_plan_rebind_input(self._h, i, C.c_void_p(new.data_ptr()))
            if rc < 0:
                _lib.check(rc, "plan_rebind_input")
            self.inputs[i] = new

    def rebind_output(self, new: torch.Tensor):
        """Zero-copy results: the op that writes the plan's (single-tensor) output writes `new` from now on."""
        old = self.outputs
        if not isinstance(old, torch.Tensor):
            raise RuntimeError("yolov6_amd: rebind_output needs a plan with one output tensor")
        if new.data_ptr() == old.data_ptr():
            return
        if new.shape != old.shape or new.dtype != old.dtype or not new.is_contiguous():
            raise RuntimeError("yolov6_amd: rebinding needs a contiguous tensor of the compiled shape/dtype")
        n = self._lib.y6_plan_rebind_output(self._h, C.c_void_p(old.data_ptr()), C.c_void_p(new.data_ptr()))
        if n < 0:
            _lib.check(n, "plan_rebind_output")
        if n == 0:
            raise RuntimeError("yolov6_amd: no op of this plan writes its output tensor (rebind_output)")
        self.outputs = new            # (the caller keeps `new` alive: Model.forward holds its ring of result tensors)
        self.captured = False

    def run(self):
        # every launch of the plan counts: utils/nms.py takes the head tail's candidates only for the plan's LATEST run
        self._run_seq = getattr(self, "_run_seq", 0) + 1
        _lib.check(self._lib.y6_plan_run(self._h, _lib.current_stream_ptr()), "plan_run")
        return self.outputs

    def attach_nms(self, conf_thres=0.25, classes=None, multi_label=False):
        """Let the plan's fused head tail select the NMS candidates of its rows while they sit in LDS (include/yolov6_hip.h
        y6_nms_sink: the arithmetic of y6_nms's own first stage, reference yolov6/utils/nms.py:48, :69-84).  Returns a
        token to pass to utils.nms.nms_raw(..., candidates=token) with the SAME conf_thres / classes / multi_label - that call
        then starts at the sort and does not re-read the prediction tensor for the selection - or None when the plan has no
        fused head tail that can serve it (the caller just keeps calling nms_raw without the token).  attach_nms(None)
        detaches."""
        sink = _lib.NmsSink()
        if conf_thres is None:
            n = self._lib.y6_plan_set_nms_sink(self._h, C.byref(sink))
            self._nms_token = None
            return None
        out = self.outputs[0] if isinstance(self.outputs, (tuple, list)) else self.outputs
        if not isinstance(out, torch.Tensor) or out.dim() != 3:
            return None
        B, A, no = out.shape
        nc = no - 5
        ml = bool(multi_label) and nc > 1
        nbytes = self._lib.y6_nms_workspace_bytes(B, A, nc, int(ml))
        ws = torch.empty(nbytes, dtype=torch.uint8, device=out.device)
        cls_t = torch.as_tensor(list(classes), dtype=torch.int32, device=out.device) if classes is not None else None
        sink.workspace, sink.workspace_bytes = C.c_void_p(ws.data_ptr()), nbytes
        sink.conf_thres = float(conf_thres)
        sink.classes = C.c_void_p(cls_t.data_ptr()) if cls_t is not None else None
        sink.n_classes = int(cls_t.numel()) if cls_t is not None else 0
        sink.multi_label = int(ml)
        n = self._lib.y6_plan_set_nms_sink(self._h, C.byref(sink))
        if n < 0:
            _lib.check(n, "plan_set_nms_sink")
        if n == 0:
            self._nms_token = None
            return None
        self.captured = False
        self._nms_token = dict(workspace=ws, classes_t=cls_t, conf_thres=float(conf_thres), multi_label=ml,
                               classes=None if classes is None else tuple(int(c) for c in classes), shape=(B, A, no), plan=self)
        return self._nms_token

    def run_range(self, first: int, last: int):
        """Eager launch of ops [first, last) only (per-layer parity tests)."""
        self._run_seq = getattr(self, "_run_seq", 0) + 1
        _lib.check(self._lib.y6_plan_run_range(self._h, _lib.current_stream_ptr(), first, last), "plan_run_range")

    def autotune(self, iters: int = 3):
        _lib.check(self._lib.y6_plan_autotune(self._h, _lib.current_stream_ptr(), iters), "plan_autotune")

    def copy_variants_from(self, other: "Plan"):
        """Run the conv kernels `other` runs (a plan of the same module for the same input shapes: no second tuning)."""
        _lib.check(self._lib.y6_plan_copy_variants(self._h, other._h), "plan_copy_variants")

    def capture(self):
        """Capture into a hipGraph (needs a non-default stream current)."""
        _lib.check(self._lib.y6_plan_capture(self._h, _lib.current_stream_ptr()), "plan_capture")
        self.captured = True

    def schedule(self, costs=None, profile_iters: int = 3, policy=None, margin=None, accesses=None):
        """Two-stream schedule of run() (yolov6_amd/schedule.py): ops off the critical path of the forward go to the plan's
        side stream, each as early as its inputs allow.  `costs`: per-op times (any unit); default: the ops timed one by one
        on this device.  Returns a summary dict, or None when the plan has nothing to overlap / holds ops without view
        information (int8 twins, calibration): run() then stays on one stream."""
        from . import schedule as S
        n = self.num_ops
        if accesses is None:
            log = getattr(self, "op_log", None)
            if not log or len(log) != n:
                return None
            accesses = [S.op_access(e) for e in log]
        acc = list(accesses)
        if len(acc) != n or any(a is None for a in acc):
            return None
        deps = S.dependences(acc)
        if S.is_chain(deps):             # (single-op and straight-line plans: nothing to overlap, nothing to time)
            return None
        if costs is None:
            costs = [r["ms"] for r in self.profile(profile_iters)]
        import os
        policy = policy or os.environ.get("Y6_SCHED_POLICY", "alap")
        margin = float(margin if margin is not None else os.environ.get("Y6_SCHED_MARGIN", "2.0"))
        try:
            res = S.build_schedule(deps, costs, policy=policy, margin=margin)
            if res is None:
                return None
            order, stream, edges = res
            S.check_schedule(deps, order, stream, edges)    # independent re-statement of what the executor guarantees
        except AssertionError as e:                          # an optimisation must not take the model down: one stream, loudly
            import warnings
            warnings.warn(f"yolov6_amd: two-stream schedule rejected ({e}); the plan runs on one stream")
            return None
        co = (C.c_int32 * n)(*order)
        cs = (C.c_int32 * n)(*stream)
        flat = [v for e in edges for v in e]
        ce = (C.c_int32 * max(1, len(flat)))(*flat)
        _lib.check(self._lib.y6_plan_set_schedule(self._h, co, cs, n, ce, len(edges)), "plan_set_schedule")
        self.sched = dict(policy=policy, margin=margin, order=list(order), stream=list(stream), edges=list(edges), costs=list(costs),
                          side_ops=[i for i in range(n) if stream[i]], side_cost=sum(costs[i] for i in range(n) if stream[i]),
                          total_cost=sum(costs))
        return self.sched

    def clear_schedule(self):
        _lib.check(self._lib.y6_plan_set_schedule(self._h, None, None, 0, None, 0), "plan_set_schedule")
        self.sched = None

    def timing_begin(self, slots: int):
        _lib.check(self._lib.y6_plan_timing_begin(self._h, slots), "plan_timing_begin")

    def run_timed(self):
        """run() with a hipEvent between consecutive ops (live per-kernel timing)."""
        _lib.check(self._lib.y6_plan_run_timed(self._h, _lib.current_stream_ptr()), "plan_run_timed")
        return self.outputs

    def timing_read(self):
        """After a device sync: list of dict(op, kind, variant, ksize, stride, ms, flops, bytes) with ms = mean per run."""
        n = self.num_ops
        ms = (C.c_float * n)()
        used = self._lib.y6_plan_timing_read(self._h, ms, n)
        if used < 0:
            _lib.check(used, "plan_timing_read")
        names = {1: "conv", 2: "convt", 3: "stem", 4: "sppf", 5: "decode", 6: "nchw2nhwc", 7: "nhwc2nchw"}
        rows = []
        for i in range(n):
            kind, var, ks, st = (C.c_int32() for _ in range(4))
            fl, by = C.c_double(), C.c_double()
            _lib.check(self._lib.y6_plan_op_info(self._h, i, C.byref(kind), C.byref(var), C.byref(ks), C.byref(st),
                                                 C.byref(fl), C.byref(by)), "plan_op_info")
            vname = self._lib.y6_conv_variant_name(var.value).decode() if var.value >= 0 else ""
            kname = _lib.TOP_NAMES.get(ks.value, "generic") if kind.value == 8 else names.get(kind.value, "?")
            rows.append(dict(op=i, kind=kname, variant=vname, ksize=ks.value if kind.value != 8 else 0, stride=st.value,
                             ms=float(ms[i]) / max(used, 1), flops=fl.value, bytes=by.value))
        return rows

    @property
    def num_ops(self) -> int:
        return self._lib.y6_plan_num_ops(self._h)

    def side_pending(self) -> int:
        """Side-stream ops the current stream has not been ordered behind (0 after every run / run_range)."""
        return int(self._lib.y6_plan_side_pending(self._h))

    def variant_table(self):
        """[(op index, conv kernel variant name)] of the plan's conv ops: what an autotuned plan chose by timing, what a plan
        that was not autotuned derives from the layer shapes (csrc/conv_misc.hip default_variant: the same in every process)."""
        out = []
        for i in range(self.num_ops):
            kind, var, ks, st = (C.c_int32() for _ in range(4))
            fl, by = C.c_double(), C.c_double()
            _lib.check(self._lib.y6_plan_op_info(self._h, i, C.byref(kind), C.byref(var), C.byref(ks), C.byref(st),
                                                 C.byref(fl), C.byref(by)), "plan_op_info")
            if kind.value == 1:
                out.append((i, self._lib.y6_conv_variant_name(var.value).decode() if var.value >= 0 else "shape-derived"))
        return out

    def variant_hash(self) -> str:
        import hashlib
        return hashlib.sha256(repr(self.variant_table()).encode()).hexdigest()[:16]

    def profile(self, iters: int = 5):
        n = self.num_ops
        ms = (C.c_float * n)()
        kind = (C.c_int32 * n)()
        var = (C.c_int32 * n)()
        fl = (C.c_double * n)()
        by = (C.c_double * n)()
        rc = self._lib.y6_plan_profile(self._h, _lib.current_stream_ptr(), iters, ms, kind, var, fl, by, n)
        if rc < 0:
            _lib.check(rc, "plan_profile")
        names = {1: "conv", 2: "convt", 3: "stem", 4: "sppf", 5: "decode", 6: "nchw2nhwc", 7: "nhwc2nchw"}
        rows = []
        for i in range(n):
            vname = self._lib.y6_conv_variant_name(var[i]).decode() if var[i] >= 0 else ""
            rows.append(dict(op=i, kind=names.get(kind[i], str(kind[i])), variant=vname, ms=float(ms[i]),
                             flops=float(fl[i]), bytes=float(by[i])))
        return rows

    def __del__(self):
        try:
            if self._h:
                self._lib.y6_plan_destroy(self._h)
                self._h = None
        except Exception:
            pass


class PlanBuilder:
    def __init__(self, device, quant=None):
        self.lib = _lib.load()
        self.device = torch.device(device)
        self.quant = quant       # yolov6_amd.quant.QuantState or None: calibration pass / int8 lowering
        self._no_quant = 0       # depth of `with pb.no_quant():` (detection head: kept in fp16)
        if self.device.type != "cuda":
            raise RuntimeError("yolov6_amd: the HIP hot path needs a ROCm device; there is no CPU fallback")
        self.h = C.c_void_p(self.lib.y6_plan_create())
        self.keep: List[torch.Tensor] = []
        self.force_variant = -1  # tests: force one conv kernel variant for every conv
        self.inputs: List[torch.Tensor] = []
        self.conv_log = []       # (Cin, Cout, k, s, H, W) per conv, for reporting
        self.op_log = []         # one dict per plan op, in plan order: what it reads / writes and its parameters
                                 # (tests replay single ops against the oracle with these)
        self.buf_ids = {}        # data_ptr of an activation buffer -> allocation index (stable between two lowerings)
        self.fp16_reads = []     # views read outside the plan (lazy feature maps): their fp16 form must exist
        self.twins = {}          # buffer index -> (int8 tensor [B,H,W,cstride], amax): int8 twins written by producers
        # producer -> 3x3 stride-2 fusion (csrc/conv_fused.hip): a 1x1 conv / the image conv whose output the lowering declared
        # single-use (hint_single_use(): BiFusion's cv2, the backbone's stem) is held back for one call; if the very next op is
        # a 3x3 stride-2 conv reading exactly that tensor, the pair becomes ONE op and the intermediate tensor is never written
        # (a read of it by any later op raises).  A/B switch: Y6_NO_FUSE_S2.
        self._pending = None
        self._single_use = False # set by hint_single_use(): the NEXT conv's output has exactly one consumer, the op after it
        self._elided = set()     # data_ptr of buffers whose producer was fused away
        import os
        self._fuse_s2 = not os.environ.get("Y6_NO_FUSE_S2")
        # widths at which `1x1 -> 3x3 stride 2` runs as the fused kernel.  Round 6: with the whole-reduction 1x1 kernel (csrc/conv_pw.hip)
        # the 128-channel pair is faster as two launches (80x80 b32: 50 us against 66 fused, tools/fused_bench.py); the 64-channel pair
        # stays fused (60 against 90).  Y6_FUSE_PW_S2_128=1: A/B switch
        self._fuse_pw_widths = (64, 128) if os.environ.get("Y6_FUSE_PW_S2_128") else (64,)

    def __del__(self):
        h, self.h = getattr(self, "h", None), None
        if h:                    # a builder that was never finalized (scan pass of the int8 lowering) owns its plan
            self.lib.y6_plan_destroy(h)

    # ---------------------------------------------------------------- memory
    def new_buffer(self, B, H, W, C_) -> TRef:
        t = torch.empty((B, H, W, C_), dtype=torch.float16, device=self.device)
        self.keep.append(t)
        self.buf_ids[t.data_ptr()] = len(self.buf_ids)
        return TRef(t, B, H, W, C_, C_, 0)

    def buf_id(self, ref: "TRef"):
        return self.buf_ids.get(ref.buf.data_ptr())

    def keep_fp16(self, refs):
        """Declare views that are read outside the plan (Model's lazily converted feature maps)."""
        refs = list(refs)
        if self._pending is not None and any(isinstance(r, TRef) and r.buf.data_ptr() == self._pending["out"].buf.data_ptr() for r in refs):
            self._flush()
        self._live(*refs)
        self.fp16_reads += refs

    def _twin(self, ref: "TRef", amax: float) -> "TRef":
        """The int8 twin view of `ref` (allocated with the fp16 buffer's geometry on first use)."""
        bid = self.buf_id(ref)
        if bid not in self.twins:
            t = torch.zeros((ref.B, ref.H, ref.W, ref.cstride), dtype=torch.int8, device=self.device)
            self.keep.append(t)
            self.twins[bid] = (t, float(amax))
        t, a = self.twins[bid]
        assert a == float(amax), "one activation scale per int8 twin buffer"
        return TRef(t, ref.B, ref.H, ref.W, ref.C, ref.cstride, ref.coff)

    def _f32(self, t: Optional[torch.Tensor], fp16_round=True):
        if t is None:
            return None
        t = t.detach().to(self.device, torch.float32)
        if fp16_round:  # the reference runs model.half(): biases / affine terms are fp16 values
            t = t.half().float()
        t = t.contiguous()
        self.keep.append(t)
        return t

    @staticmethod
    def _ptr(t: Optional[torch.Tensor]):
        return C.c_void_p(t.data_ptr()) if t is not None else None

    # ---------------------------------------------------------------- held-back producers
    def hint_single_use(self):
        """The output of the NEXT conv() call is read by exactly one op: the one lowered right after it.  (Only the module
        that lowers both ops knows; the builder cannot see future readers of a tensor.)"""
        self._single_use = True

    def _flush(self):
        """Add a held-back producer op as the plain op it is."""
        pend, self._pending = self._pending, None
        if pend is None:
            return
        if pend["kind"] == "pw":
            _lib.check(self.lib.y6_plan_add_conv(self.h, C.byref(pend["desc"])), "plan_add_conv")
        else:
            _lib.check(self.lib.y6_plan_add_stem(self.h, C.byref(pend["desc"])), "plan_add_stem")
        self.op_log.append(pend["entry"])

    def _live(self, *refs):
        for r in refs:
            if isinstance(r, TRef) and r.buf.data_ptr() in self._elided:
                raise RuntimeError("yolov6_amd: an op reads a tensor whose producer was fused into its consumer "
                                   "(set Y6_NO_FUSE_S2=1 and report the graph)")

    def _fuse_with_pending(self, x, K, stride, post, res):
        """The held-back producer, if THIS conv (3x3 stride 2, plain epilogue) reads exactly its output; else None."""
        pend = self._pending
        if pend is None or not isinstance(x, TRef) or K != 3 or stride != 2 or post is not None or res is not None:
            return None
        o = pend["out"]
        if x.buf.data_ptr() != o.buf.data_ptr() or x.coff != 0 or x.C != o.C or x.cstride != o.cstride:
            return None
        return pend

    # ---------------------------------------------------------------- inputs / outputs
    def as_nhwc(self, x) -> TRef:
        """NCHWInput -> TRef through the layout adapter kernel (TRef passes through)."""
        if isinstance(x, TRef):
            return x
        t = x.t
        _lib.require_gpu_tensor(t, "input")
        if not t.is_contiguous():
            raise RuntimeError("yolov6_amd: input must be a contiguous NCHW tensor")
        B, C_, H, W = t.shape
        self._flush()
        out = self.new_buffer(B, H, W, C_)
        self.inputs.append(t)
        ct = out.ct()
        _lib.check(self.lib.y6_plan_add_nchw2nhwc(self.h, C.c_void_p(t.data_ptr()), _dtype_tag(t), C.byref(ct)),
                   "plan_add_nchw2nhwc")
        self.op_log.append(dict(kind="nchw2nhwc", x=t, out=out))
        return out

    def to_nchw(self, x: TRef, dtype=torch.float16) -> torch.Tensor:
        self._flush()
        self._live(x)
        out = torch.empty((x.B, x.C, x.H, x.W), dtype=dtype, device=self.device)
        self.keep.append(out)
        ct = x.ct()
        _lib.check(self.lib.y6_plan_add_nhwc2nchw(self.h, C.byref(ct), C.c_void_p(out.data_ptr()), _dtype_tag(out)),
                   "plan_add_nhwc2nchw")
        self.op_log.append(dict(kind="nhwc2nchw", x=x, out=out))
        return out

    # ---------------------------------------------------------------- ops
    def no_quant(self):
        """Context: convs lowered inside stay fp16 under an int8 lowering (the detection head, like the `skip` list of
        the reference's QAT config configs/repopt/yolov6s_opt_qat.py:70-76)."""
        pb = self

        class _Ctx:
            def __enter__(self):
                pb._no_quant += 1

            def __exit__(self, *exc):
                pb._no_quant -= 1
        return _Ctx()

    def conv(self, x, weight, bias, stride=1, act=None, out: Optional[TRef] = None, post=None,
             res: Optional[TRef] = None, res_alpha: Optional[torch.Tensor] = None) -> TRef:
        """conv(k in {1,3}, pad=k//2) + bias (+post affine) + act (+alpha*res). weight: OIHW."""
        Cout, Cin, K, K2 = weight.shape
        assert K == K2
        reads_image = isinstance(x, NCHWInput)      # the network's first conv stays fp16 (its input is 8-bit pixels already)
        single_use, self._single_use = self._single_use, False
        fuse_src = None if reads_image else self._fuse_with_pending(x, K, stride, post, res)
        if fuse_src is None:
            self._flush()
        self._live(x, res)
        if reads_image:
            if K == 3 and stride == 2 and x.shape[1] <= 4 and Cout in (8, 16, 32, 48, 64) and res is None:
                return self._stem(x, weight, bias, act, out, post, single_use)
            x = self.as_nhwc(x)
        if x.C != Cin:
            raise RuntimeError(f"yolov6_amd: conv expects {Cin} input channels, got {x.C}")
        pad = K // 2
        Ho = (x.H + 2 * pad - K) // stride + 1
        Wo = (x.W + 2 * pad - K) // stride + 1
        fresh_out = out is None                      # this call allocates the output: nobody else holds a view of it yet
        if out is None:
            out = self.new_buffer(x.B, Ho, Wo, Cout)
        assert (out.B, out.H, out.W, out.C) == (x.B, Ho, Wo, Cout), "conv output slice has the wrong shape"
        if self.quant is not None and not self._no_quant and not reads_image:
            self._flush()
            fuse_src = None
            idx = self.quant.next_index(dict(cin=Cin, cout=Cout, k=K, stride=stride))
            if self.quant.mode == "calibrate":
                xt = x.ct()
                _lib.check(self.lib.y6_plan_add_absmax(self.h, C.byref(xt), C.c_void_p(self.quant.slot_ptr(idx, self.device))),
                           "plan_add_absmax")
                self.op_log.append(dict(kind="absmax", x=x, index=idx))
            else:
                return self._conv_i8(x, weight, bias, stride, act, out, post, res, res_alpha, self.quant.amax_of(idx))
        w32 = weight.detach().to(self.device, torch.float32).contiguous()
        w16 = w32.half().contiguous()
        n = self.lib.y6_packed_weight_elems(Cout, Cin, K)
        packed = torch.empty(n, dtype=torch.float16, device=self.device)
        _lib.check(self.lib.y6_pack_conv_weight(self._ptr(w16), Y6_F16, Cout, Cin, K, self._ptr(packed),
                                                _lib.current_stream_ptr()), "pack_conv_weight")
        self.keep += [w16, packed]
        b32 = self._f32(bias)
        ps = self._f32(post[0]) if post is not None else None
        pt = self._f32(post[1]) if post is not None else None
        ra = self._f32(res_alpha.reshape(1), fp16_round=True) if res_alpha is not None else None
        d = _lib.ConvDesc()
        d.inp, d.out = x.ct(), out.ct()
        d.w_packed, d.w_oihw = self._ptr(packed), self._ptr(w16)
        d.bias, d.post_scale, d.post_shift = self._ptr(b32), self._ptr(ps), self._ptr(pt)
        d.res = res.ct() if res is not None else _null_tensor()
        d.res_alpha = self._ptr(ra)
        d.ksize, d.stride, d.act, d.variant = K, stride, ACT_BY_NAME[act], self.force_variant
        self.conv_log.append((Cin, Cout, K, stride, x.H, x.W))
        entry = dict(kind="conv", x=x, out=out, w=w32, b=bias, stride=stride, act=act, post=post, res=res, alpha=res_alpha)
        if fuse_src is not None and self._add_fused_pair(fuse_src, d, entry):
            return out
        self._flush()                                # (a held-back producer whose consumer the fused kernel did not take)
        # a 1x1 conv into a tensor of its own may turn out to be the producer of a 3x3 stride-2 conv: hold it back one call
        if (self._fuse_s2 and single_use and K == 1 and stride == 1 and fresh_out and post is None and res is None and self.quant is None
                and self.force_variant < 0 and Cin == Cout and Cout in self._fuse_pw_widths):
            self._pending = dict(kind="pw", desc=d, entry=entry, out=out)
            return out
        _lib.check(self.lib.y6_plan_add_conv(self.h, C.byref(d)), "plan_add_conv")
        self.op_log.append(entry)
        return out

    def _add_fused_pair(self, pend, d_s2, entry_s2) -> bool:
        """Add `held-back producer -> this 3x3 stride-2 conv` as one fused op, if the kernel takes the pair."""
        if pend["kind"] == "pw":
            fd = _lib.PwS2Desc()
            fd.pw, fd.s2 = pend["desc"], d_s2
            if not self.lib.y6_fused_pw_s2_supported(C.byref(fd)):
                return False
            _lib.check(self.lib.y6_plan_add_pw_s2(self.h, C.byref(fd)), "plan_add_pw_s2")
            kind = "pw_s2"
        else:
            fd = _lib.StemS2Desc()
            fd.stem, fd.s2 = pend["desc"], d_s2
            if not self.lib.y6_fused_stem_s2_supported(C.byref(fd)):
                return False
            _lib.check(self.lib.y6_plan_add_stem_s2(self.h, C.byref(fd)), "plan_add_stem_s2")
            kind = "stem_s2"
        self._pending = None
        self._elided.add(pend["out"].buf.data_ptr())
        pe = pend["entry"]
        self.op_log.append(dict(kind=kind, x=pe["x"], mid=pend["out"], out=entry_s2["out"], producer=pe, s2=entry_s2))
        return True

    def _conv_i8(self, x: TRef, weight, bias, stride, act, out: TRef, post, res, res_alpha, amax: float) -> TRef:
        """int8 conv (include/yolov6_hip.h y6_conv_i8_desc): weights quantised here per output channel, the fp16
        input quantised by the kernel with the calibrated `amax`."""
        from .quant import quantize_weight, dequant_vector
        self._flush()
        self._live(x, res)
        Cout, Cin, K, _ = weight.shape
        wq, s_w = quantize_weight(weight)                       # CPU: int8 OIHW, fp32 [Cout]
        wq_d = wq.to(self.device).contiguous()
        packed = torch.empty(self.lib.y6_packed_weight_i8_bytes(Cout, Cin, K), dtype=torch.int8, device=self.device)
        _lib.check(self.lib.y6_pack_conv_weight_i8(self._ptr(wq_d), Cout, Cin, K, self._ptr(packed), _lib.current_stream_ptr()),
                   "pack_conv_weight_i8")
        dq = dequant_vector(amax, s_w).to(self.device).contiguous()
        self.keep += [wq_d, packed, dq]
        d = _lib.ConvI8Desc()
        c = d.conv
        c.inp, c.out = x.ct(), out.ct()
        c.w_packed, c.w_oihw = self._ptr(packed), None
        c.bias = self._ptr(self._f32(bias, fp16_round=False))
        c.post_scale = self._ptr(self._f32(post[0])) if post is not None else None
        c.post_shift = self._ptr(self._f32(post[1])) if post is not None else None
        c.res = res.ct() if res is not None else _null_tensor()
        c.res_alpha = self._ptr(self._f32(res_alpha.reshape(1), fp16_round=True)) if res_alpha is not None else None
        c.ksize, c.stride, c.act, c.variant = K, stride, ACT_BY_NAME[act], 0
        d.dequant = self._ptr(dq)
        d.in_amax = float(amax)
        d.q_in, d.q_out, d.q_out_amax, d.acc_out = _null_tensor(), _null_tensor(), 0.0, None
        # int8 twins (yolov6_amd.quant.plan_twins decided them in a scan lowering): read the producer's int8 copy of the
        # input instead of quantising fp16 on load; write an int8 copy of the output for its quantised consumers and
        # drop the fp16 store when nothing reads it
        q_in = q_out = None
        q_out_amax, has_out = 0.0, True
        dec = getattr(self.quant, "decisions", None)
        if dec:
            di = dec.get(self.buf_id(x))
            if di is not None and di["twin"]:
                q_in = self._twin(x, di["amax"])
                assert di["amax"] == float(amax)
                d.q_in = q_in.ct()
            do = dec.get(self.buf_id(out))
            if do is not None and do["twin"]:
                q_out, q_out_amax = self._twin(out, do["amax"]), do["amax"]
                d.q_out, d.q_out_amax = q_out.ct(), q_out_amax
                if not do["fp16"]:
                    has_out = False
                    c.out = _lib.Tensor(None, out.B, out.H, out.W, out.C, out.cstride, out.coff)
        _lib.check(self.lib.y6_plan_add_conv_i8(self.h, C.byref(d)), "plan_add_conv_i8")
        self.conv_log.append((Cin, Cout, K, stride, x.H, x.W))
        self.op_log.append(dict(kind="conv_i8", x=x, out=out, w=weight.detach().float().cpu(), b=bias, stride=stride, act=act,
                                post=post, res=res, alpha=res_alpha, amax=float(amax), wq=wq, s_w=s_w, dequant=dq,
                                q_in=q_in, q_out=q_out, q_out_amax=q_out_amax, has_out=has_out))
        return out

    def _stem(self, x: NCHWInput, weight, bias, act, out, post, single_use=False) -> TRef:
        t = x.t
        _lib.require_gpu_tensor(t, "input")
        if not t.is_contiguous():
            raise RuntimeError("yolov6_amd: input must be a contiguous NCHW tensor")
        B, Cin, H, W = t.shape
        Cout = weight.shape[0]
        Ho, Wo = (H + 2 - 3) // 2 + 1, (W + 2 - 3) // 2 + 1
        fresh_out = out is None
        if out is None:
            out = self.new_buffer(B, Ho, Wo, Cout)
        # fp16-rounded weights, as model.half() would hold them
        w32 = weight.detach().to(self.device, torch.float32).half().float().contiguous()
        self.keep += [w32]
        self.inputs.append(t)
        d = _lib.StemDesc()
        d.in_nchw, d.in_dtype = C.c_void_p(t.data_ptr()), _dtype_tag(t)
        d.B, d.Cin, d.H, d.W = B, Cin, H, W
        d.out = out.ct()
        d.w_oihw_f32 = self._ptr(w32)
        d.bias = self._ptr(self._f32(bias))
        d.post_scale = self._ptr(self._f32(post[0])) if post is not None else None
        d.post_shift = self._ptr(self._f32(post[1])) if post is not None else None
        d.act = ACT_BY_NAME[act]
        d.q_out, d.q_out_amax = _null_tensor(), 0.0
        entry = dict(kind="stem", x=t, out=out, w=w32, b=bias, stride=2, act=act, post=post, res=None, alpha=None)
        if self.quant is not None and self.quant.mode == "int8" and not self._no_quant:
            # int8 plans (round 6): the image conv stays fp16 (its input is 8-bit pixels already) but can leave the int8 twin of
            # its output for the quantised conv behind it - that conv then reads 1/2 of the bytes on the register-fed stride-2
            # int8 kernel instead of quantising 210 MB of fp16 on the per-tap kernel (S-QA 640^2 b32: 205 us, 10 % of the step)
            probe = _lib.Tensor(None, out.B, out.H, out.W, out.C, (out.cstride + 15) // 16 * 16, out.coff)
            d.q_out = probe
            entry["twin_ok"] = bool(self.lib.y6_stem_twin_supported(C.byref(d))) and out.cstride % 16 == 0 and out.coff % 16 == 0
            d.q_out = _null_tensor()
            dec = getattr(self.quant, "decisions", None)
            do = dec.get(self.buf_id(out)) if dec else None
            if do is not None and do["twin"] and entry["twin_ok"]:
                q_out = self._twin(out, do["amax"])
                d.q_out, d.q_out_amax = q_out.ct(), float(do["amax"])
                entry.update(q_out=q_out, q_out_amax=float(do["amax"]), has_out=bool(do["fp16"]))
                if not do["fp16"]:
                    d.out = _lib.Tensor(None, out.B, out.H, out.W, out.C, out.cstride, out.coff)
        if (self._fuse_s2 and single_use and fresh_out and post is None and self.quant is None and self.force_variant < 0
                and Cin == 3 and Cout == 32):
            self._pending = dict(kind="stem", desc=d, entry=entry, out=out)      # (conv() flushed before calling us)
            return out
        _lib.check(self.lib.y6_plan_add_stem(self.h, C.byref(d)), "plan_add_stem")
        self.op_log.append(entry)
        return out

    def convt2x2(self, x, weight, bias, out: Optional[TRef] = None) -> TRef:
        """ConvTranspose2d(k=2, s=2): weight IOHW [Cin, Cout, 2, 2]."""
        self._flush()
        x = self.as_nhwc(x)
        self._live(x)
        Cin, Cout = weight.shape[0], weight.shape[1]
        assert tuple(weight.shape[2:]) == (2, 2) and x.C == Cin
        if out is None:
            out = self.new_buffer(x.B, 2 * x.H, 2 * x.W, Cout)
        w16 = weight.detach().to(self.device, torch.float16).contiguous()
        n = self.lib.y6_packed_weight_elems(Cout, Cin, 1) * 4
        packed = torch.empty(n, dtype=torch.float16, device=self.device)
        _lib.check(self.lib.y6_pack_convt2x2_weight(self._ptr(w16), Y6_F16, Cin, Cout, self._ptr(packed),
                                                    _lib.current_stream_ptr()), "pack_convt2x2_weight")
        self.keep += [w16, packed]
        d = _lib.ConvTDesc()
        d.inp, d.out = x.ct(), out.ct()
        d.w_packed = self._ptr(packed)
        d.bias = self._ptr(self._f32(bias))
        _lib.check(self.lib.y6_plan_add_convt(self.h, C.byref(d)), "plan_add_convt")
        self.op_log.append(dict(kind="convt", x=x, out=out, w=weight, b=bias))
        return out

    def sppf_pool(self, x: TRef, y1: TRef, y2: TRef, y3: TRef):
        self._flush()
        self._live(x)
        # int8 plans (round 6): when the concat buffer the pools write has an int8 twin (quant.plan_twins: every other writer is an int8
        # conv, one consumer scale), the pools leave the twins of their slices too - max-pooling commutes with the quantiser
        dec = getattr(self.quant, "decisions", None) if self.quant is not None else None
        do = dec.get(self.buf_id(y1)) if dec else None
        twin_ok = all(t.cstride % 8 == 0 and t.coff % 8 == 0 for t in (y1, y2, y3))
        q_outs, amax = None, 0.0
        if do is not None and do["twin"] and twin_ok and all(self.buf_id(t) == self.buf_id(y1) for t in (y2, y3)):
            amax = float(do["amax"])
            q_outs = [self._twin(t, amax) for t in (y1, y2, y3)]
            d = _lib.SppfQDesc()
            d.x, d.y1, d.y2, d.y3 = x.ct(), y1.ct(), y2.ct(), y3.ct()
            d.q1, d.q2, d.q3 = (q.ct() for q in q_outs)
            d.q_amax = amax
            _lib.check(self.lib.y6_plan_add_sppf_q(self.h, C.byref(d)), "plan_add_sppf_q")
        else:
            cts = [t.ct() for t in (x, y1, y2, y3)]
            _lib.check(self.lib.y6_plan_add_sppf(self.h, *[C.byref(c) for c in cts]), "plan_add_sppf")
        self.op_log.append(dict(kind="sppf", x=x, outs=[y1, y2, y3], q_outs=q_outs, q_amax=amax, twin_ok=twin_ok))

    def head_decode(self, cls: List[TRef], reg: List[TRef], strides, use_dfl, reg_max, proj, nc,
                    grid_cell_offset=0.5) -> torch.Tensor:
        self._flush()
        self._live(*cls, *reg)
        B = cls[0].B
        A = sum(c.H * c.W for c in cls)
        out = torch.empty((B, A, 5 + nc), dtype=torch.float32, device=self.device)
        self.keep.append(out)
        d = _lib.DecodeDesc()
        d.n_levels = len(cls)
        for i, (c, r) in enumerate(zip(cls, reg)):
            d.cls[i], d.reg[i] = c.ct(), r.ct()
            d.stride[i] = float(strides[i])
        d.use_dfl, d.reg_max = int(bool(use_dfl)), int(reg_max)
        pj = self._f32(proj, fp16_round=False) if use_dfl else None
        d.proj = self._ptr(pj)
        d.grid_cell_offset = grid_cell_offset
        d.out = C.c_void_p(out.data_ptr())
        d.nc = nc
        _lib.check(self.lib.y6_plan_add_decode(self.h, C.byref(d)), "plan_add_decode")
        self.op_log.append(dict(kind="decode", cls=list(cls), reg=list(reg), out=out, strides=list(strides), use_dfl=bool(use_dfl),
                                reg_max=int(reg_max), proj=proj, nc=nc))
        return out

    def _packed_1x1(self, weight):
        """Packed MFMA image of a [Cout, Cin, 1, 1] weight (fp16-rounded, as model.half() holds it)."""
        Cout, Cin = weight.shape[0], weight.shape[1]
        w16 = weight.detach().to(self.device, torch.float32).half().contiguous()
        packed = torch.empty(self.lib.y6_packed_weight_elems(Cout, Cin, 1), dtype=torch.float16, device=self.device)
        _lib.check(self.lib.y6_pack_conv_weight(self._ptr(w16), Y6_F16, Cout, Cin, 1, self._ptr(packed), _lib.current_stream_ptr()),
                   "pack_conv_weight")
        self.keep += [w16, packed]
        return packed

    def _pred_decode_desc(self, cls_feat, reg_feat, cls_preds, reg_preds, strides, use_dfl, reg_max, proj, nc, grid_cell_offset, out,
                          first_anchor=0, total_anchors=0):
        d = _lib.PredDecodeDesc()
        d.n_levels = len(cls_feat)
        for i, (c, r, (wc, bc), (wr, br)) in enumerate(zip(cls_feat, reg_feat, cls_preds, reg_preds)):
            d.cls_feat[i], d.reg_feat[i] = c.ct(), r.ct()
            d.w_cls[i], d.w_reg[i] = self._packed_1x1(wc).data_ptr(), self._packed_1x1(wr).data_ptr()
            zc = bc if bc is not None else torch.zeros(wc.shape[0])
            zr = br if br is not None else torch.zeros(wr.shape[0])
            d.b_cls[i], d.b_reg[i] = self._f32(zc).data_ptr(), self._f32(zr).data_ptr()
            d.stride[i] = float(strides[i])
        d.use_dfl, d.reg_max = int(bool(use_dfl)), int(reg_max)
        d.proj = self._ptr(self._f32(proj, fp16_round=False)) if use_dfl else None
        d.grid_cell_offset = grid_cell_offset
        d.out = C.c_void_p(out.data_ptr()) if out is not None else None
        d.nc = nc
        d.first_anchor, d.total_anchors = int(first_anchor), int(total_anchors)
        return d

    def head_pred_decode(self, cls_feat: List[TRef], reg_feat: List[TRef], cls_preds, reg_preds, strides, use_dfl, reg_max, proj,
                         nc, grid_cell_offset=0.5) -> Optional[torch.Tensor]:
        """The head's tail in ONE launch (include/yolov6_hip.h y6_pred_decode_desc): cls_pred / reg_pred 1x1 convs of every
        level + the decode.  cls_preds / reg_preds: per level (weight [Cout,Cin,1,1], bias).  Returns None when the fused
        kernel does not take the shape (the caller then lowers convs + head_decode)."""
        self._flush()
        self._live(*cls_feat, *reg_feat)
        nreg = 4 * ((int(reg_max) + 1) if use_dfl else 1)
        ok = all(w.shape[0] == nc and tuple(w.shape[2:]) == (1, 1) and w.shape[1] == c.C for (w, _), c in zip(cls_preds, cls_feat)) and \
            all(w.shape[0] == nreg and tuple(w.shape[2:]) == (1, 1) and w.shape[1] == r.C for (w, _), r in zip(reg_preds, reg_feat)) and \
            all(c.C % 16 == 0 and c.C == r.C and c.cstride % 8 == 0 and c.coff % 8 == 0 and r.cstride % 8 == 0 and r.coff % 8 == 0
                for c, r in zip(cls_feat, reg_feat)) and len(cls_feat) <= _lib.MAX_LEVELS and (nc + 5 + nreg) * 64 * 4 <= 64 * 1024
        if not ok:
            return None
        B = cls_feat[0].B
        A = sum(c.H * c.W for c in cls_feat)
        out = torch.empty((B, A, 5 + nc), dtype=torch.float32, device=self.device)
        d = self._pred_decode_desc(cls_feat, reg_feat, cls_preds, reg_preds, strides, use_dfl, reg_max, proj, nc, grid_cell_offset, out)
        if not self.lib.y6_head_pred_decode_supported(C.byref(d)):
            return None
        self.keep.append(out)
        cpu = lambda ps: [(w.detach().float().cpu(), None if b is None else b.detach().float().cpu()) for w, b in ps]   # noqa: E731
        _lib.check(self.lib.y6_plan_add_pred_decode(self.h, C.byref(d)), "plan_add_pred_decode")
        self.op_log.append(dict(kind="pred_decode", cls_feat=list(cls_feat), reg_feat=list(reg_feat),
                                cls_preds=cpu(cls_preds), reg_preds=cpu(reg_preds),
                                out=out, strides=list(strides), use_dfl=bool(use_dfl), reg_max=int(reg_max), proj=proj, nc=nc))
        return out

    # ---------------------------------------------------------------- finish
    def finalize(self, outputs, autotune=True, iters=3, variants_from: Optional[Plan] = None) -> Plan:
        self._flush()
        plan = Plan(self.h, self.keep, outputs, self.inputs)
        plan.op_log = self.op_log
        self.h = None
        if variants_from is not None:
            plan.copy_variants_from(variants_from)
        elif autotune and self.force_variant < 0:
            plan.autotune(iters)
        import os
        # two-stream schedule of run(): on since r03u (+2.0 % img/s same box, alternating runs, bit-identical results:
        # profiles/r03/bench_infer_r03u_*.json); Y6_SCHED_STREAMS=1 keeps every op on the caller's stream
        # (int8 plans too since r04a: twin-aware access lists, 12 259 -> 12 405 img/s same box, int8 GPU tests green with it on)
        if os.environ.get("Y6_SCHED_STREAMS", "2") == "2" and (self.quant is None or self.quant.mode == "int8"):
            plan.schedule()
        return plan
