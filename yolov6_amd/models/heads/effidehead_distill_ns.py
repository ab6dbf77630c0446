"""Efficient decoupled head for cost-free self-distillation of the N / S models (`Model(..., distill_ns=True)`).
Reference: yolov6/models/heads/effidehead_distill_ns.py (Detect :10-139, build_effidehead_layer :142-270).

Six layers per level: stem, cls_conv, reg_conv, cls_pred, reg_pred_dist (4 * 17 DFL bins - only the distillation loss reads
it) and reg_pred (4 plain distances).  The EVAL branch (:104-139) is the anchor-free decode WITHOUT DFL on `reg_preds` - on
the HIP path exactly the parent's lowering (stem / merged cls+reg convs / 1x1 preds / one decode kernel), so checkpoints of
distilled N / S models load and run; `reg_preds_dist` is carried for the state_dict ABI only.  The training branch (four
outputs, :81-103) and the distillation losses (models/losses/loss_distill*.py; oracle: oracle/loss_distill_oracle.py) are
not on the HIP path yet: `.train()` forwards raise.
"""
import math

import torch
import torch.nn as nn

from ...layers.common import ConvBNSiLU
from ..effidehead import Detect as _Detect


class Detect(_Detect):
    export = False

    def __init__(self, num_classes=80, num_layers=3, inplace=True, head_layers=None, use_dfl=True, reg_max=16):
        nn.Module.__init__(self)        # the parent's constructor groups five layers per level
        assert head_layers is not None
        self.nc = num_classes
        self.no = num_classes + 5
        self.nl = num_layers
        self.grid = [torch.zeros(1)] * num_layers
        self.prior_prob = 1e-2
        self.inplace = inplace
        self.stride = torch.tensor([8, 16, 32])
        self.use_dfl = use_dfl
        self.reg_max = reg_max
        self.proj_conv = nn.Conv2d(self.reg_max + 1, 1, 1, bias=False)
        self.grid_cell_offset = 0.5
        self.grid_cell_size = 5.0
        groups = {"stems": 0, "cls_convs": 1, "reg_convs": 2, "cls_preds": 3, "reg_preds_dist": 4, "reg_preds": 5}
        for name, k in groups.items():
            setattr(self, name, nn.ModuleList(head_layers[i * 6 + k] for i in range(num_layers)))

    def initialize_biases(self):
        '''Reference: effidehead_distill_ns.py:48-79.'''
        cls_bias = -math.log((1 - self.prior_prob) / self.prior_prob)
        for convs, value in ((self.cls_preds, cls_bias), (self.reg_preds_dist, 1.0), (self.reg_preds, 1.0)):
            for conv in convs:
                conv.bias = nn.Parameter(torch.full_like(conv.bias.detach().view(-1), value), requires_grad=True)
                conv.weight = nn.Parameter(torch.zeros_like(conv.weight.detach()), requires_grad=True)
        self.proj = nn.Parameter(torch.linspace(0, self.reg_max, self.reg_max + 1), requires_grad=False)
        self.proj_conv.weight = nn.Parameter(self.proj.view([1, self.reg_max + 1, 1, 1]).clone().detach(),
                                             requires_grad=False)

    def _eval_use_dfl(self):
        return False                     # :116-133: boxes come from reg_preds (l, t, r, b), no bin projection

    def lower_train(self, tb, x):
        """Training branch (effidehead_distill_ns.py:81-103): per level stem -> {cls conv, cls_pred}, {reg conv, reg_preds_dist AND
        reg_preds}; sigmoid on the class logits; levels flattened and concatenated to cls_scores [B,A,nc], reg_distri
        [B,A,4*(reg_max+1)] and reg_lrtb [B,A,4] (the fourth output loss_distill_ns.py consumes)."""
        stems, cls_out, dist_out, lrtb_out = [], [], [], []
        for i in range(self.nl):
            f = self.stems[i].lower(tb, x[i])
            stems.append(f)
            c = self.cls_convs[i].lower(tb, f)
            r = self.reg_convs[i].lower(tb, f)
            cp, dp, rp = self.cls_preds[i], self.reg_preds_dist[i], self.reg_preds[i]
            cls_out.append(tb.conv(c, cp.weight, 1, bias=cp.bias))
            dist_out.append(tb.conv(r, dp.weight, 1, bias=dp.bias))
            lrtb_out.append(tb.conv(r, rp.weight, 1, bias=rp.bias))
            tb.trace.update({f"detect.stem{i}": f, f"detect.cls_conv{i}": c, f"detect.reg_conv{i}": r,
                             f"detect.cls_logit{i}": cls_out[-1], f"detect.reg_raw{i}": dist_out[-1], f"detect.reg_lrtb{i}": lrtb_out[-1]})
        scores, distri = tb.head_pack(cls_out, dist_out, self.nc, self.reg_preds_dist[0].out_channels)
        _, lrtb = tb.head_pack(None, lrtb_out, 0, self.reg_preds[0].out_channels)
        return stems, (scores, distri, lrtb)


def build_effidehead_layer(channels_list, num_anchors, num_classes, reg_max=16):
    '''Six layers per level in the reference's order, as an unnamed nn.Sequential (keys "0" .. "17").'''
    layers = []
    for ci in (6, 8, 10):
        c = channels_list[ci]
        layers += [ConvBNSiLU(in_channels=c, out_channels=c, kernel_size=1, stride=1),
                   ConvBNSiLU(in_channels=c, out_channels=c, kernel_size=3, stride=1),
                   ConvBNSiLU(in_channels=c, out_channels=c, kernel_size=3, stride=1),
                   nn.Conv2d(in_channels=c, out_channels=num_classes * num_anchors, kernel_size=1),
                   nn.Conv2d(in_channels=c, out_channels=4 * (reg_max + num_anchors), kernel_size=1),
                   nn.Conv2d(in_channels=c, out_channels=4 * num_anchors, kernel_size=1)]
    return nn.Sequential(*layers)
