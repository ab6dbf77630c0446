"""ORACLE (test infrastructure - never imported by the product path).  PARITY UNPINNED BY CONSTRUCTION.

int8 inference of a deploy-form graph (BASELINE configs[4]: YOLOv6-S QARepVGG).  The reference tree
contains no int8 arithmetic: its int8 numbers come from TensorRT engines / NVIDIA pytorch_quantization fake-quant
(deploy/TensorRT/onnx_to_trt.py:62-112, tools/qat/qat_utils.py:61-146: per-channel 8-bit weights, per-tensor 8-bit
activations, head skipped - configs/repopt/yolov6s_opt_qat.py:70-76), neither of which is in the tree or installed.
This file therefore DEFINES the arithmetic the HIP int8 path is held to (the same rule is written in
include/yolov6_hip.h next to y6_conv_i8_desc):

  * the graph is the half-precision deploy graph (`Oracle(emulate_fp16=True)`: activations are fp16 tensors between ops);
  * quantised: every conv of backbone and neck EXCEPT the one that reads the image; fp16 as before: that first conv, the
    transposed convs, the whole detection head, the decode;
  * weights      symmetric per output channel from the deploy weights as the half-precision model holds them (fp16 values):
                 s_w[c] = max|w[c]| / 127,  w_q = clamp(rne(w / s_w[c]), -127, 127)
  * activations  symmetric per tensor, amax = largest |x| the conv's input saw during calibration:
                 a = fp16(amax), inv = fp16(127 / a),  x_q = clamp(rne(x * inv), -127, 127)
                 (x, inv fp16 values; the product is taken exactly and rounded ONCE, half to even)
  * accumulate   int8 x int8 -> int32, exact (evaluated here as a float64 convolution of integer-valued tensors)
  * epilogue     fp32, every operation rounded separately:  y = fp32(acc) * d[c] + bias[c],  d[c] = (fp32(a) / 127) * s_w[c];
                 then the fp16 graph's tail: (round to fp16, kept post-BN affine of QARepVGG common.py:338-339),
                 round to fp16, activation, round to fp16.

`int8_conv` is the single-op form (tests replay a native plan op by op with it); `Int8Oracle` walks the whole model.  The
table of activation scales is a list in call order of the quantised convs.
"""
import torch
import torch.nn.functional as F

from .model_oracle import Oracle


def quantize_sym(t, scale):
    """clamp(round_half_even(t / scale), -127, 127) as an integer-valued fp32 tensor (weights)."""
    return torch.clamp(torch.round(t / scale), -127, 127)


def quantize_weight(w):
    w = w.detach().float()
    s_w = w.abs().amax(dim=(1, 2, 3)).clamp_min(1e-12) / 127.0
    return quantize_sym(w, s_w.view(-1, 1, 1, 1)), s_w


def act_constants(amax):
    """(a, inv) as fp16-valued python floats."""
    a = torch.tensor(float(amax), dtype=torch.float32).half()
    inv = (torch.tensor(127.0, dtype=torch.float32) / a.float()).half()
    return float(a), float(inv)


def quantize_act(x, amax):
    """x: fp16-valued tensor.  Exact product in float64 (11-bit x 11-bit significands), one half-to-even rounding."""
    a, inv = act_constants(amax)
    xc = torch.clamp(x.double(), -a, a)
    return torch.round(xc * inv)          # |xc * inv| <= 127.07: never reaches +-128


def exact_int_conv(xq, wq, stride):
    """Convolution of integer-valued tensors (|x|, |w| <= 127), exact, returned as float64.

    Evaluated with fp32 convolutions over groups of <= 64 input channels: every partial sum of such a group is an integer of
    magnitude <= 9 * 64 * 127 * 127 = 9.3e6 < 2^24, hence exactly representable in fp32 whatever the summation order (FMA or
    not, blocked or not); the group results are added in float64.  (The plain float64 convolution this replaces gives the
    same integers - tests/test_oracle_cpu.py checks both against each other - but takes minutes per 640x640 image at full
    width; this form takes seconds, which is what lets the full-width int8 parity test live in the `-m gpu` suite.)"""
    k = wq.shape[-1]
    cin = xq.shape[1]
    step = 64 if k == 3 else 512             # 1x1: 512 * 16129 = 8.3e6 < 2^24
    acc = None
    for c0 in range(0, cin, step):
        part = F.conv2d(xq[:, c0:c0 + step].float(), wq[:, c0:c0 + step].float(), None, stride=stride, padding=k // 2).double()
        acc = part if acc is None else acc + part
    return acc


def int8_accumulate(x, w, stride, amax):
    xq = quantize_act(x, amax)
    wq, s_w = quantize_weight(w)
    return exact_int_conv(xq, wq, stride), s_w


def int8_conv(orc, x, w, b, stride, act, post, amax):
    """One quantised conv + its fused tail; `orc` supplies the fp16 rounding (`q`) and the activation."""
    acc, s_w = int8_accumulate(x, w, stride, amax)
    a, _ = act_constants(amax)
    d = (torch.tensor(a, dtype=torch.float32) / 127.0) * s_w.float()
    y = acc.float() * d.view(1, -1, 1, 1)
    if b is not None:
        y = y + b.detach().float().view(1, -1, 1, 1)
    if post is not None:
        y = orc.q(y) * orc.q(post[0]).view(1, -1, 1, 1) + orc.q(post[1]).view(1, -1, 1, 1)
    return orc.q(orc.act(orc.q(y), act)), acc


class Int8Oracle(Oracle):
    def __init__(self, cfg, sd_deploy, num_classes=80):
        super().__init__(cfg, sd_deploy, num_classes, emulate_fp16=True)
        self.amax = None          # list, call order of the quantised convs
        self.calibrating = False
        self._calls = 0           # conv_fused calls of the current forward (call 0 reads the image)
        self._qidx = 0
        self._in_head = False
        self.layers = []          # per quantised conv: dict(cin, cout, k, stride) - compared with the product's lowering order
        self.stats = []           # per quantised conv of the last forward: dict(acc_absmax)

    # ------------------------------------------------------------------ calibration
    def calibrate(self, batches):
        self.calibrating = True
        self.amax = []
        with torch.no_grad():
            for x in batches:
                self.forward(x)
        self.calibrating = False
        return list(self.amax)

    # ------------------------------------------------------------------ the quantised conv
    def conv_fused(self, x, w, b, stride, act, post=None):
        first = self._calls == 0
        self._calls += 1
        if self._in_head or first:
            return super().conv_fused(x, w, b, stride, act, post)
        i = self._qidx
        self._qidx += 1
        if i == len(self.layers):
            self.layers.append(dict(cin=w.shape[1], cout=w.shape[0], k=w.shape[-1], stride=stride))
        if self.calibrating:
            m = float(x.abs().max())
            if i == len(self.amax):
                self.amax.append(m)
            else:
                self.amax[i] = max(self.amax[i], m)
            return super().conv_fused(x, w, b, stride, act, post)
        if self.amax is None or i >= len(self.amax):
            raise RuntimeError("Int8Oracle: call calibrate() (or set .amax) before forward()")
        # the half-precision model holds fp16 weights and biases: they are what gets quantised (as the product does)
        y, acc = int8_conv(self, x, self.q(w), None if b is None else self.q(b), stride, act, post, self.amax[i])
        self.stats.append(dict(acc_absmax=float(acc.abs().max())))
        return y

    def head(self, feats):
        self._in_head = True
        try:
            return super().head(feats)
        finally:
            self._in_head = False

    def forward(self, x, train_form=False):
        assert not train_form, "the int8 graph is the deploy form"
        self._calls = 0
        self._qidx = 0
        self.layers = []
        self.stats = []
        return super().forward(self.q(x), train_form=False)
