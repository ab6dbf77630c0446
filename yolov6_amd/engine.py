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
        _lib.check(self._lib.y6_plan_run(self._h, _lib.current_stream_ptr()), "plan_run")
        return self.outputs

    def run_range(self, first: int, last: int):
        """Eager launch of ops [first, last) only (per-layer parity tests)."""
        _lib.check(self._lib.y6_plan_run_range(self._h, _lib.current_stream_ptr(), first, last), "plan_run_range")

    def autotune(self, iters: int = 3):
        _lib.check(self._lib.y6_plan_autotune(self._h, _lib.current_stream_ptr(), iters), "plan_autotune")

    def capture(self):
        """Capture into a hipGraph (needs a non-default stream current)."""
        _lib.check(self._lib.y6_plan_capture(self._h, _lib.current_stream_ptr()), "plan_capture")
        self.captured = True

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
        self.fp16_reads += list(refs)

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
        out = self.new_buffer(B, H, W, C_)
        self.inputs.append(t)
        ct = out.ct()
        _lib.check(self.lib.y6_plan_add_nchw2nhwc(self.h, C.c_void_p(t.data_ptr()), _dtype_tag(t), C.byref(ct)),
                   "plan_add_nchw2nhwc")
        self.op_log.append(dict(kind="nchw2nhwc", x=t, out=out))
        return out

    def to_nchw(self, x: TRef, dtype=torch.float16) -> torch.Tensor:
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
        if reads_image:
            if K == 3 and stride == 2 and x.shape[1] <= 4 and Cout in (8, 16, 32, 48, 64) and res is None:
                return self._stem(x, weight, bias, act, out, post)
            x = self.as_nhwc(x)
        if x.C != Cin:
            raise RuntimeError(f"yolov6_amd: conv expects {Cin} input channels, got {x.C}")
        pad = K // 2
        Ho = (x.H + 2 * pad - K) // stride + 1
        Wo = (x.W + 2 * pad - K) // stride + 1
        if out is None:
            out = self.new_buffer(x.B, Ho, Wo, Cout)
        assert (out.B, out.H, out.W, out.C) == (x.B, Ho, Wo, Cout), "conv output slice has the wrong shape"
        if self.quant is not None and not self._no_quant and not reads_image:
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
        _lib.check(self.lib.y6_plan_add_conv(self.h, C.byref(d)), "plan_add_conv")
        self.conv_log.append((Cin, Cout, K, stride, x.H, x.W))
        self.op_log.append(dict(kind="conv", x=x, out=out, w=w32, b=bias, stride=stride, act=act, post=post, res=res,
                                alpha=res_alpha))
        return out

    def _conv_i8(self, x: TRef, weight, bias, stride, act, out: TRef, post, res, res_alpha, amax: float) -> TRef:
        """int8 conv (include/yolov6_hip.h y6_conv_i8_desc): weights quantised here per output channel, the fp16
        input quantised by the kernel with the calibrated `amax`."""
        from .quant import quantize_weight, dequant_vector
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

    def _stem(self, x: NCHWInput, weight, bias, act, out, post) -> TRef:
        t = x.t
        _lib.require_gpu_tensor(t, "input")
        if not t.is_contiguous():
            raise RuntimeError("yolov6_amd: input must be a contiguous NCHW tensor")
        B, Cin, H, W = t.shape
        Cout = weight.shape[0]
        Ho, Wo = (H + 2 - 3) // 2 + 1, (W + 2 - 3) // 2 + 1
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
        _lib.check(self.lib.y6_plan_add_stem(self.h, C.byref(d)), "plan_add_stem")
        self.op_log.append(dict(kind="stem", x=t, out=out, w=w32, b=bias, stride=2, act=act, post=post, res=None, alpha=None))
        return out

    def convt2x2(self, x, weight, bias, out: Optional[TRef] = None) -> TRef:
        """ConvTranspose2d(k=2, s=2): weight IOHW [Cin, Cout, 2, 2]."""
        x = self.as_nhwc(x)
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
        cts = [t.ct() for t in (x, y1, y2, y3)]
        _lib.check(self.lib.y6_plan_add_sppf(self.h, *[C.byref(c) for c in cts]), "plan_add_sppf")
        self.op_log.append(dict(kind="sppf", x=x, outs=[y1, y2, y3]))

    def head_decode(self, cls: List[TRef], reg: List[TRef], strides, use_dfl, reg_max, proj, nc,
                    grid_cell_offset=0.5) -> torch.Tensor:
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

    # ---------------------------------------------------------------- finish
    def finalize(self, outputs, autotune=True, iters=3) -> Plan:
        plan = Plan(self.h, self.keep, outputs, self.inputs)
        plan.op_log = self.op_log
        self.h = None
        if autotune and self.force_variant < 0:
            plan.autotune(iters)
        return plan
