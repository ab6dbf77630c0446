"""int8 inference of a deploy-form model (BASELINE configs[4]: YOLOv6-S QARepVGG).

The reference's int8 numbers come from TensorRT engines built from PTQ / QAT models (deploy/TensorRT/onnx_to_trt.py:62-112,
tools/qat/qat_utils.py:61-146: per-channel 8-bit weights, per-tensor 8-bit activations, the detection head and `proj_conv`
skipped - configs/repopt/yolov6s_opt_qat.py:70-76).  None of that arithmetic is in its tree; this module is the same recipe
on the HIP int8 kernels (include/yolov6_hip.h `y6_conv_i8_desc` holds the exact quantisation rule):

    model = build_model(cfg, nc, device).eval().half();  fuse_model(model);  switch_to_deploy ...
    table = yolov6_amd.quant.calibrate(model, [batch0, batch1, ...])     # PTQ: max-calibration, on the device
    yolov6_amd.quant.quantize(model, table)                              # later forwards run the int8 plan
    yolov6_amd.quant.dequantize(model)                                   # back to fp16

Quantised: every conv of backbone and neck except the one reading the image.  Kept in fp16: that first conv, the
transposed convs, the whole detection head and the decode.
"""
import ctypes as C

import torch

from . import _lib


class QuantState:
    """What PlanBuilder consults while lowering: `calibrate` inserts an abs-max reduction in front of every quantisable
    conv, `int8` replaces those convs by int8 ones with the calibrated scales."""
    SLOTS = 4096

    def __init__(self, mode, amax=None, twins=True):
        assert mode in ("calibrate", "int8")
        self.mode = mode
        self.amax = None if amax is None else [float(a) for a in amax]
        self.twins = bool(twins)  # int8 lowering: let producers write int8 copies for their quantised consumers
        self.decisions = None     # buffer index -> dict(twin, amax, fp16), from plan_twins() over a scan lowering
        self.layers = []          # per quantisable conv, in lowering order: dict(cin, cout, k, stride)
        self._buf = None

    def begin_lowering(self):
        self.layers = []

    def next_index(self, info):
        self.layers.append(info)
        return len(self.layers) - 1

    def slot_ptr(self, idx, device):
        if self._buf is None:
            self._buf = torch.zeros(self.SLOTS, dtype=torch.float32, device=device)
        if idx >= self.SLOTS:
            raise RuntimeError("yolov6_amd.quant: more quantisable convs than calibration slots")
        return self._buf.data_ptr() + 4 * idx

    def amax_of(self, idx):
        if self.amax is None or idx >= len(self.amax):
            raise RuntimeError(f"yolov6_amd.quant: no calibrated scale for conv #{idx} (table has "
                               f"{0 if self.amax is None else len(self.amax)} entries) - run calibrate() on this model first")
        a = self.amax[idx]
        if not (a > 0.0):
            raise RuntimeError(f"yolov6_amd.quant: conv #{idx} saw an all-zero input during calibration")
        return a

    def read(self):
        return self._buf[:len(self.layers)].cpu().tolist()

    def key(self):
        return (self.mode, None if self.amax is None else tuple(self.amax), self.twins)


def _views(e):
    """(reads, writes) of one op-log entry as lists of TRef."""
    k = e["kind"]
    rd, wr = [], []
    if k in ("conv", "conv_i8"):
        rd = [e["x"]] + ([e["res"]] if e.get("res") is not None else [])
        wr = [e["out"]]
    elif k == "convt":
        rd, wr = [e["x"]], [e["out"]]
    elif k == "sppf":
        rd, wr = [e["x"]], list(e["outs"])
    elif k == "decode":
        rd = list(e["cls"]) + list(e["reg"])
    elif k == "pred_decode":
        rd = list(e["cls_feat"]) + list(e["reg_feat"])
    elif k == "pw_s2":
        rd, wr = [e["x"]], [e["out"]]
    elif k == "stem_s2":
        wr = [e["out"]]
    elif k in ("nhwc2nchw", "absmax"):
        rd = [e["x"]]
    elif k in ("stem", "nchw2nhwc"):
        wr = [e["out"]]
    else:
        raise NotImplementedError(f"yolov6_amd.quant: op kind {k} in an int8 lowering")
    return rd, wr


def plan_twins(pb):
    """Decide, from a scan lowering, which activation buffers get an int8 twin.

    A buffer gets one when every quantised conv reading it uses ONE scale (convs reading the same tensor - or a concat
    buffer as a whole - calibrate to the same amax) and every op writing it is an int8 conv (whose epilogue then emits
    the int8 copy for free).  Its fp16 form is still written if anything else reads it: fp16 ops (transposed convs,
    pools, the head, residual adds, the decode), the caller (feature maps), or an int8 conv with a different scale."""
    info = {}

    def slot(ref):
        bid = pb.buf_id(ref)
        if bid is None:
            return None
        s = info.setdefault(bid, dict(scales=set(), i8_reads=0, fp16_reads=0, i8_writes=0, other_writes=0, aligned=True))
        if ref.C % 16 or ref.coff % 16 or ref.cstride % 16:      # 16-byte pieces of the int8 view
            s["aligned"] = False
        return s

    for e in pb.op_log:
        rd, wr = _views(e)
        for j, r in enumerate(rd):
            s = slot(r)
            if s is None:
                continue
            if e["kind"] == "conv_i8" and j == 0:
                s["scales"].add(e["amax"])
                s["i8_reads"] += 1
            else:
                s["fp16_reads"] += 1
        for w in wr:
            s = slot(w)
            if s is not None:
                s["i8_writes" if e["kind"] == "conv_i8" else "other_writes"] += 1
    for r in pb.fp16_reads:
        s = slot(r)
        if s is not None:
            s["fp16_reads"] += 1
    dec = {}
    for bid, s in info.items():
        twin = len(s["scales"]) == 1 and s["other_writes"] == 0 and s["i8_writes"] > 0 and s["aligned"]
        dec[bid] = dict(twin=twin, amax=next(iter(s["scales"])) if twin else None, fp16=(not twin) or s["fp16_reads"] > 0)
    return dec


def quantize_weight(w):
    """Per-output-channel symmetric int8 (host, one time):  s_w[c] = max|w[c]| / 127,  w_q = clamp(rne(w / s_w[c]), +-127)."""
    w = w.detach().float().cpu()
    s_w = (w.abs().amax(dim=(1, 2, 3)).clamp_min(1e-12) / 127.0).contiguous()
    wq = torch.clamp(torch.round(w / s_w.view(-1, 1, 1, 1)), -127, 127).to(torch.int8).contiguous()
    return wq, s_w


def dequant_vector(amax, s_w):
    """[Cout] fp32:  (fp32(fp16(amax)) / 127) * s_w[c] - the scale of one int32 accumulator unit."""
    a16 = torch.tensor(float(amax), dtype=torch.float32).half().float()
    return ((a16 / 127.0) * s_w.float()).contiguous()


def calibrate(model, batches):
    """Max-calibration on the device: runs the fp16 plan of `model` over `batches` (NCHW image tensors on the GPU) with an
    abs-max reduction (`y6_absmax`) over the input of every quantisable conv.  Returns the table (one float per conv, in
    lowering order) to hand to `quantize`."""
    if model.training:
        raise RuntimeError("yolov6_amd.quant: calibrate a deploy-form model in .eval() mode")
    batches = list(batches)
    if not batches:
        raise ValueError("yolov6_amd.quant: no calibration batches")
    prev = model.__dict__.get("_y6_quant")
    st = QuantState("calibrate")
    model.__dict__["_y6_quant"] = st
    model.invalidate_plans()
    try:
        for x in batches:
            plan = model.compile(x, autotune=False)
            plan.run()
        torch.cuda.synchronize()
        table = st.read()
    finally:
        if prev is None:
            model.__dict__.pop("_y6_quant", None)
        else:
            model.__dict__["_y6_quant"] = prev
        model.invalidate_plans()
    return table


def quantize(model, table, twins=True):
    """Later forwards of `model` lower to the int8 plan with these activation scales.  twins=False keeps every
    activation in fp16 only (each int8 conv quantises its input while loading it)."""
    model.__dict__["_y6_quant"] = QuantState("int8", table, twins=twins)
    model.invalidate_plans()
    return model


def dequantize(model):
    model.__dict__.pop("_y6_quant", None)
    model.invalidate_plans()
    return model
