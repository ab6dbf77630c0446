"""ORACLE (test infrastructure - never imported by the product path).  PARITY UNPINNED BY CONSTRUCTION.

int8 inference of the QARepVGG deploy graph (BASELINE configs[4]).  The reference tree contains no
int8 arithmetic: its int8 numbers come from TensorRT engines / NVIDIA pytorch_quantization fake-quant
(deploy/TensorRT/onnx_to_trt.py:62-112, tools/qat/qat_utils.py:61-146), neither of which is in the tree or
installed.  This file therefore DEFINES the quantisation the HIP int8 path will be held to:

  * weights      symmetric, per output channel:  s_w[c] = max|w[c]| / 127,  w_q = clamp(round_half_even(w / s_w), +-127)
  * activations  symmetric, per tensor (the input of every quantised conv):  s_x = amax / 127 with amax the largest
                 |x| seen over the calibration batches (4 synthetic batches, `calibrate`)
  * accumulate   int8 x int8 -> int32, exact (evaluated here as a float64 convolution of integer-valued tensors)
  * epilogue     fp32:  y = acc * (s_x * s_w[c]) + bias[c], then the kept post-BatchNorm affine of QARepVGG
                 (common.py:338-339) and the activation, exactly as the fp graph
  * not quantised  the transposed convolutions, the whole detection head (stems, cls/reg convs, preds, proj_conv -
                 the `skip` list of configs/repopt/yolov6s_opt_qat.py:70-76) and the decode

The layer order (hence the key of every activation scale) is the call order of Oracle.conv_fused, which is
deterministic for a given config.
"""
import torch
import torch.nn.functional as F

from .model_oracle import Oracle


def quantize_sym(t, scale):
    """clamp(round_half_even(t / scale), -127, 127) as an integer-valued fp32 tensor."""
    return torch.clamp(torch.round(t / scale), -127, 127)


class Int8Oracle(Oracle):
    def __init__(self, cfg, sd_deploy, num_classes=80):
        super().__init__(cfg, sd_deploy, num_classes, emulate_fp16=False)
        self.amax = {}            # conv index -> calibrated |x| max of its input
        self.calibrating = False
        self._idx = 0
        self._in_head = False
        self.stats = {}           # conv index -> dict(s_x, acc_absmax) of the last forward

    # ------------------------------------------------------------------ calibration
    def calibrate(self, batches):
        self.calibrating = True
        self.amax = {}
        with torch.no_grad():
            for x in batches:
                self._idx = 0
                super().forward(x)
        self.calibrating = False
        return dict(self.amax)

    # ------------------------------------------------------------------ the quantised conv
    def conv_fused(self, x, w, b, stride, act, post=None):
        i = self._idx
        self._idx += 1
        if self._in_head:
            return super().conv_fused(x, w, b, stride, act, post)
        if self.calibrating:
            self.amax[i] = max(self.amax.get(i, 0.0), float(x.abs().max()))
            return super().conv_fused(x, w, b, stride, act, post)
        if i not in self.amax:
            raise RuntimeError("Int8Oracle: call calibrate() before forward()")
        s_x = max(self.amax[i], 1e-12) / 127.0
        s_w = w.abs().amax(dim=(1, 2, 3)).clamp_min(1e-12) / 127.0
        xq = quantize_sym(x, s_x)
        wq = quantize_sym(w, s_w.view(-1, 1, 1, 1))
        acc = F.conv2d(xq.double(), wq.double(), None, stride=stride, padding=w.shape[-1] // 2)   # exact int32 values
        self.stats[i] = dict(s_x=s_x, acc_absmax=float(acc.abs().max()))
        y = acc.float() * (s_x * s_w).view(1, -1, 1, 1)
        if b is not None:
            y = y + b.view(1, -1, 1, 1)
        if post is not None:
            y = y * post[0].view(1, -1, 1, 1) + post[1].view(1, -1, 1, 1)
        return self.act(y, act)

    def head(self, feats):
        self._in_head = True
        try:
            return super().head(feats)
        finally:
            self._in_head = False

    def forward(self, x, train_form=False):
        assert not train_form, "the int8 graph is the deploy form"
        self._idx = 0
        self.stats = {}
        return super().forward(x, train_form=False)
