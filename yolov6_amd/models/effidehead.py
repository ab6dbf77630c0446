"""EffiDeHead (anchor-free decoupled head) on the HIP path.
Reference: yolov6/models/effidehead.py (Detect :10-139, build_effidehead_layer :142-293).

Per level: stem 1x1 SiLU -> {cls 3x3 SiLU -> 1x1 (nc)} , {reg 3x3 SiLU -> 1x1 (4*(reg_max+1))}
as five fused conv kernels, then ONE decode kernel for all levels (sigmoid, DFL, dist2bbox,
x stride, concat) that writes the reference's [B, A, 5+nc] fp32 tensor directly.
"""
import math
import os

import torch
import torch.nn as nn

from ..layers.common import ConvBNSiLU, HipModule


class Detect(HipModule):
    export = False

    def __init__(self, num_classes=80, num_layers=3, inplace=True, head_layers=None, use_dfl=True, reg_max=16):
        super().__init__()
        assert head_layers is not None
        self.nc = num_classes
        self.no = num_classes + 5
        self.nl = num_layers
        self.grid = [torch.zeros(1)] * num_layers
        self.prior_prob = 1e-2
        self.inplace = inplace
        self.stride = torch.tensor([8, 16, 32] if num_layers == 3 else [8, 16, 32, 64])
        self.use_dfl = use_dfl
        self.reg_max = reg_max
        self.proj_conv = nn.Conv2d(self.reg_max + 1, 1, 1, bias=False)
        self.grid_cell_offset = 0.5
        self.grid_cell_size = 5.0
        groups = {"stems": 0, "cls_convs": 1, "reg_convs": 2, "cls_preds": 3, "reg_preds": 4}
        for name, k in groups.items():
            setattr(self, name, nn.ModuleList(head_layers[i * 5 + k] for i in range(num_layers)))

    def initialize_biases(self):
        '''Reference: effidehead.py:49-69 (prior-probability bias, zero prediction weights, DFL proj).'''
        cls_bias = -math.log((1 - self.prior_prob) / self.prior_prob)
        for convs, value in ((self.cls_preds, cls_bias), (self.reg_preds, 1.0)):
            for conv in convs:
                conv.bias = nn.Parameter(torch.full_like(conv.bias.detach().view(-1), value), requires_grad=True)
                conv.weight = nn.Parameter(torch.zeros_like(conv.weight.detach()), requires_grad=True)
        self.proj = nn.Parameter(torch.linspace(0, self.reg_max, self.reg_max + 1), requires_grad=False)
        self.proj_conv.weight = nn.Parameter(self.proj.view([1, self.reg_max + 1, 1, 1]).clone().detach(),
                                             requires_grad=False)

    def lower_train(self, tb, x):
        """Training branch (effidehead.py:72-92): per level stem -> {cls conv, 1x1 pred}, {reg conv, 1x1 pred}; sigmoid on the
        class logits; levels flattened and concatenated to cls_scores [B,A,nc], reg_distri [B,A,4*(reg_max+1)] (fp32)."""
        stems, cls_out, reg_out = [], [], []
        for i in range(self.nl):
            f = self.stems[i].lower(tb, x[i])
            stems.append(f)
            c = self.cls_convs[i].lower(tb, f)
            r = self.reg_convs[i].lower(tb, f)
            cp, rp = self.cls_preds[i], self.reg_preds[i]
            cls_out.append(tb.conv(c, cp.weight, 1, bias=cp.bias))
            reg_out.append(tb.conv(r, rp.weight, 1, bias=rp.bias))
            tb.trace.update({f"detect.stem{i}": f, f"detect.cls_conv{i}": c, f"detect.reg_conv{i}": r,
                             f"detect.cls_logit{i}": cls_out[-1], f"detect.reg_raw{i}": reg_out[-1]})
        return stems, tb.head_pack(cls_out, reg_out, self.nc, self.reg_preds[0].out_channels)

    def _lower_cls_reg_convs(self, pb, i, f):
        """cls_conv and reg_conv of one level (effidehead.py:95-99) read the same stem output with the same geometry and
        activation: ONE launch over the concatenated output channels (twice the work items per launch on the small maps,
        three launches fewer per forward); the two results are channel slices of one buffer."""
        cc, rc = getattr(self.cls_convs[i], "block", None), getattr(self.reg_convs[i], "block", None)
        same = (cc is not None and rc is not None and hasattr(cc, "fused_weight_bias") and hasattr(rc, "fused_weight_bias")
                and cc.conv.kernel_size == rc.conv.kernel_size == (3, 3) and cc.conv.stride == rc.conv.stride == (1, 1)
                and cc.conv.padding == rc.conv.padding == (1, 1) and cc.conv.groups == rc.conv.groups == 1
                and cc.conv.dilation == rc.conv.dilation == (1, 1) and cc.conv.in_channels == rc.conv.in_channels
                and cc._activation_name() == rc._activation_name() and cc.conv.out_channels % 8 == 0)
        if not same or os.environ.get("Y6_HEAD_NO_MERGE"):      # (A/B switch)
            return self.cls_convs[i].lower(pb, f), self.reg_convs[i].lower(pb, f)
        (wc, bc), (wr, br) = cc.fused_weight_bias(), rc.fused_weight_bias()
        nc_, nr_ = wc.shape[0], wr.shape[0]
        zeros = lambda n, ref: torch.zeros(n, dtype=torch.float32, device=ref.device)
        bc = zeros(nc_, wc) if bc is None else bc
        br = zeros(nr_, wr) if br is None else br
        w = torch.cat([wc.float(), wr.float()], 0)
        b = torch.cat([bc.float(), br.float()], 0)
        f = pb.as_nhwc(f)
        both = pb.new_buffer(f.B, f.H, f.W, nc_ + nr_)
        pb.conv(f, w, b, stride=1, act=cc._activation_name(), out=both)
        return both.slice(0, nc_), both.slice(nc_, nr_)

    def _eval_use_dfl(self):
        """Does the EVAL branch decode DFL bins (effidehead.py:107-110)?  The distillation head never does."""
        return bool(self.use_dfl)

    def lower(self, pb, x, out=None):
        if self.training:
            raise NotImplementedError("yolov6_amd: Detect's training branch runs through Model.forward in train mode "
                                      "(whole-model training graph)")
        if self.export:
            raise NotImplementedError("yolov6_amd: export mode (ONNX tracing) is out of scope of the HIP path")
        use_dfl = self._eval_use_dfl()
        # the reference's eval branch projects with proj_conv.weight (effidehead.py:107-109), not with self.proj;
        # the bin count comes from the loaded weight, not from the constructor default
        proj = self.proj_conv.weight.detach().reshape(-1) if use_dfl else None
        reg_max = proj.numel() - 1 if use_dfl else self.reg_max
        strides = [float(s) for s in self.stride.tolist()]
        cfeat, rfeat = [], []
        with pb.no_quant():        # the head stays fp16 under an int8 lowering (yolov6_amd/quant.py)
            for i in range(self.nl):
                f = self.stems[i].lower(pb, x[i])
                c, r = self._lower_cls_reg_convs(pb, i, f)
                cfeat.append(pb.as_nhwc(c))
                rfeat.append(pb.as_nhwc(r))
            # cls_pred / reg_pred of every level + the decode as ONE launch (csrc/head_decode.hip head_pred_decode_kernel):
            # 7 launches and the [B,A,nc+4] fp16 logits' round trip through HBM less (A/B switch: Y6_HEAD_NO_FUSE)
            fuse = getattr(pb, "head_pred_decode", None)
            if fuse is not None and not os.environ.get("Y6_HEAD_NO_FUSE"):
                det = fuse(cfeat, rfeat, [(m.weight, m.bias) for m in self._eval_cls_preds()],
                           [(m.weight, m.bias) for m in self._eval_reg_preds()], strides, use_dfl, reg_max, proj, self.nc,
                           self.grid_cell_offset)
                if det is not None:
                    return det
            cls_out = [pb.conv(c, cp.weight, cp.bias, stride=1, act=None) for c, cp in zip(cfeat, self._eval_cls_preds())]
            reg_out = [pb.conv(r, rp.weight, rp.bias, stride=1, act=None) for r, rp in zip(rfeat, self._eval_reg_preds())]
        return pb.head_decode(cls_out, reg_out, strides, use_dfl, reg_max, proj, self.nc, self.grid_cell_offset)

    def _eval_cls_preds(self):
        return list(self.cls_preds)

    def _eval_reg_preds(self):
        return list(self.reg_preds)


def build_effidehead_layer(channels_list, num_anchors, num_classes, reg_max=16, num_layers=3):
    '''Five layers per level in the reference's order (stem, cls_conv, reg_conv, cls_pred, reg_pred);
    levels beyond the third are registered under the reference's names (effidehead.py:248-291).'''
    chx = [6, 8, 10] if num_layers == 3 else [8, 9, 10, 11]
    head_layers = nn.Sequential()
    for lvl, ci in enumerate(chx):
        c = channels_list[ci]
        made = (("stem", ConvBNSiLU(in_channels=c, out_channels=c, kernel_size=1, stride=1)),
                ("cls_conv", ConvBNSiLU(in_channels=c, out_channels=c, kernel_size=3, stride=1)),
                ("reg_conv", ConvBNSiLU(in_channels=c, out_channels=c, kernel_size=3, stride=1)),
                ("cls_pred", nn.Conv2d(in_channels=c, out_channels=num_classes * num_anchors, kernel_size=1)),
                ("reg_pred", nn.Conv2d(in_channels=c, out_channels=4 * (reg_max + num_anchors), kernel_size=1)))
        for k, (name, layer) in enumerate(made):
            # nn.Sequential(*layers) names children "0".."14"; add_module uses explicit names for level 3
            head_layers.add_module(str(lvl * 5 + k) if lvl < 3 else f"{name}{lvl}", layer)
    return head_layers
