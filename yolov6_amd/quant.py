"""int8 inference of a deploy-form model (BASELINE configs[4]: YOLOv6-S QARepVGG).

The reference's int8 numbers come from TensorRT engines built from PTQ / QAT models (deploy/TensorRT/onnx_to_trt.py:62-112,
tools/qat/qat_utils.py:61-146: per-channel 8-bit weights, per-tensor 8-bit activations, the detection head and `proj_conv`
skipped - configs/repopt/yolov6s_opt_qat.py:70-76).  None of that arithmetic is in its tree; this module is the same recipe
on the HIP int8 kernels (include/yolov6_hip.h `y6_conv_i8_desc` holds the exact quantisation rule):

    model = build_model(cfg, nc, device).eval().half();  fuse_model(model);  switch_to_deploy ...
    table = yolov6_amd.quant.calibrate(model, [batch0, batch1, ...])     # PTQ: max-calibration, on the device; or the reference's
                                                                         # recipe: method="histogram", histogram_amax_method="entropy"
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
    the int8 copy for free) or the fp16 image conv in its tiled form (which can, round 6).  Its fp16 form is still written if anything else reads it: fp16 ops (transposed convs,
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
                # (the fp16 image conv can write the twin too: engine.PlanBuilder._stem, round 6)
                # ... and so can the SPPF pools (round 6: y6_sppf_pool_q; max-pooling commutes with the quantiser)
                can = e["kind"] == "conv_i8" or (e["kind"] == "stem" and e.get("twin_ok")) or (e["kind"] == "sppf" and e.get("twin_ok"))
                s["i8_writes" if can else "other_writes"] += 1
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


class HistogramCalibrator:
    """|x| histogram of one activation tensor over the calibration batches and the three ways the reference's PTQ recipe turns
    it into a clipping value (tools/qat/qat_utils.py:12-58 `collect_stats` / `compute_amax`, configs/repopt/yolov6s_opt_qat.py:
    63-69: calib_method 'histogram', histogram_amax_method 'entropy' | 'percentile' | 'mse', percentile 99.99, 4 batches).
    The arithmetic is `pytorch_quantization.calib.HistogramCalibrator`'s (NVIDIA's library, an un-vendored dependency of the
    reference - requirements of tools/qat; not installed here): restated from its published source, PARITY UNPINNED.
      collect:    2048 bins over [0, max] of the first batch; a later batch with a larger maximum appends bins of the SAME width.
      percentile: the bin edge where the cumulative histogram reaches p %.
      mse:        amax = the bin centre (from bin 128 on) whose 8-bit fake quantisation of all centres has the least
                  count-weighted squared error.
      entropy:    TensorRT's rule - for every candidate i >= 128: P = the first i bins with the outliers folded into the last,
                  Q = P merged into 128 levels and spread back over its non-empty bins; amax = the edge of the LAST i that
                  minimises KL(P || Q)."""

    def __init__(self, num_bins=2048):
        self.num_bins = int(num_bins)
        self.hist = None      # float64 numpy
        self.edges = None

    def collect(self, x: torch.Tensor):
        import numpy as np
        x = x.detach().float().abs().flatten()
        x_max = float(x.max())
        if self.hist is None:
            hi = x_max if x_max > 0 else 1.0
            h = torch.histc(x, bins=self.num_bins, min=0.0, max=hi)
            self.hist = h.double().cpu().numpy()
            self.edges = np.linspace(0.0, hi, self.num_bins + 1)
            return
        if x_max > self.edges[-1]:
            width = self.edges[1] - self.edges[0]
            extra = int(np.ceil((x_max - self.edges[-1]) / width))
            self.edges = np.concatenate([self.edges, self.edges[-1] + width * np.arange(1, extra + 1)])
            self.hist = np.concatenate([self.hist, np.zeros(extra)])
        h = torch.histc(x, bins=len(self.edges) - 1, min=0.0, max=float(self.edges[-1]))
        self.hist = self.hist + h.double().cpu().numpy()

    def compute_amax(self, method="entropy", percentile=99.99, num_bits=8, stride=1, start_bin=128):
        if self.hist is None:
            raise RuntimeError("yolov6_amd.quant: HistogramCalibrator.compute_amax before collect")
        if method == "percentile":
            return amax_percentile(self.hist, self.edges, percentile)
        if method == "mse":
            return amax_mse(self.hist, self.edges, num_bits, stride, start_bin)
        if method == "entropy":
            return amax_entropy(self.hist, self.edges, num_bits, stride, start_bin)
        raise ValueError(f"yolov6_amd.quant: unknown histogram_amax_method {method!r} (entropy | percentile | mse)")


def amax_percentile(hist, edges, percentile):
    import numpy as np
    if not 0 <= percentile <= 100:
        raise ValueError("Invalid percentile. Must be in range 0 <= percentile <= 100.")
    cdf = np.cumsum(hist / hist.sum())
    idx = int(np.searchsorted(cdf, percentile / 100.0))
    return float(edges[min(idx, len(edges) - 1)])


def amax_mse(hist, edges, num_bits=8, stride=1, start_bin=128):
    import numpy as np
    centers = (edges[1:] + edges[:-1]) / 2.0
    bound = float((1 << (num_bits - 1)) - 1)
    best, best_i = None, None
    for i in range(start_bin, len(centers), stride):
        amax = centers[i]
        scale = bound / amax
        q = np.clip(np.round(centers * scale), -bound, bound) / scale      # fake_tensor_quant of the bin centres
        mse = float((((q - centers) ** 2) * hist).mean())
        if best is None or mse < best:
            best, best_i = mse, i
    return float(centers[best_i if best_i is not None else len(centers) - 1])


def amax_entropy(hist, edges, num_bits=8, stride=1, start_bin=128):
    import numpy as np
    bins = np.array(hist, dtype=np.float64)
    bins[0] = bins[1]
    nlev = 1 << (num_bits - 1)                      # 128 magnitudes + sign
    n = len(bins)
    if n <= start_bin:
        return float(edges[-1])
    tail = np.concatenate([np.cumsum(bins[::-1])[::-1], [0.0]])      # tail[i] = sum(bins[i:])
    best, best_i = None, None
    for i in range(start_bin, n + 1, stride):
        p = bins[:i].copy()
        nz = p != 0
        level = np.minimum((np.arange(i) * nlev) // i, nlev - 1)     # np.digitize(range(i), linspace(0, i, nlev + 1)) - 1
        cnt = np.bincount(level[nz], weights=p[nz], minlength=nlev)
        num = np.bincount(level[nz], minlength=nlev)
        q = np.zeros(i)
        q[nz] = (cnt / np.maximum(num, 1))[level[nz]]
        p[i - 1] += tail[i]
        ps, qs = p.sum(), q.sum()
        if ps == 0 or qs == 0:
            continue
        p /= ps
        q /= qs
        m = p > 0
        if np.any(q[m] == 0):
            kl = np.inf
        else:
            kl = float(np.sum(p[m] * np.log(p[m] / q[m])))         # scipy.stats.entropy(p, q)
        if best is None or kl <= best:                               # ties: the LAST minimum (len - 1 - argmin(reversed))
            best, best_i = kl, i
    return float(edges[best_i if best_i is not None else n])


def calibrate(model, batches, method="max", histogram_amax_method="entropy", percentile=99.99):
    """Activation scales of every quantisable conv, in lowering order - the table to hand to `quantize`.
    method "max": abs-max reductions (`y6_absmax`) inserted into the fp16 plan, on the device (the reference's `MaxCalibrator`).
    method "histogram": the reference's PTQ recipe (configs/repopt/yolov6s_opt_qat.py:63-69): the fp16 plan runs over
    `batches` (NCHW image tensors on the GPU), the |x| histogram of every quantisable conv's input is collected on the device
    (HistogramCalibrator) and turned into amax by `histogram_amax_method` ('entropy' - the recipe's choice - 'percentile' with
    `percentile`, or 'mse')."""
    if model.training:
        raise RuntimeError("yolov6_amd.quant: calibrate a deploy-form model in .eval() mode")
    if method not in ("max", "histogram"):
        raise ValueError(f"yolov6_amd.quant: calib_method {method!r} (max | histogram)")
    batches = list(batches)
    if not batches:
        raise ValueError("yolov6_amd.quant: no calibration batches")
    prev = model.__dict__.get("_y6_quant")
    st = QuantState("calibrate")
    model.__dict__["_y6_quant"] = st
    model.invalidate_plans()
    try:
        calibs = {}
        for x in batches:
            plan = model.compile(x, autotune=False)
            plan.run()
            if method == "histogram":
                torch.cuda.synchronize()
                # every activation of an inference plan is its own allocation: the conv inputs are still there after the run
                for e in plan.op_log:
                    if e.get("kind") == "absmax":
                        calibs.setdefault(e["index"], HistogramCalibrator()).collect(e["x"].to_nhwc_tensor())
        torch.cuda.synchronize()
        table = st.read()
        if method == "histogram":
            if sorted(calibs) != list(range(len(table))):
                raise RuntimeError("yolov6_amd.quant: the calibration plan does not expose every quantisable conv's input")
            table = [calibs[i].compute_amax(histogram_amax_method, percentile) for i in range(len(table))]
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
