"""EffiDeHead with the anchor-based auxiliary branch ("fuse_ab", the published training recipe of YOLOv6 N/S/M/L).
Reference: yolov6/models/heads/effidehead_fuseab.py (Detect :10-195, build_effidehead_layer :198-).

Per level seven layers: stem 1x1, cls conv 3x3, reg conv 3x3, the anchor-free predictions (cls nc, reg 4*(reg_max+1)) and
the anchor-based ones (cls nc*na, reg 4*na, na = 3 anchors per cell).  Inference uses the anchor-free branch only
(:143-195), so the eval lowering is the plain EffiDeHead's; the training branch (:94-139) additionally returns
cls_score_list_ab [B, na*A, nc] and reg_dist_list_ab [B, na*A, 4] = (dx, dy, (2 sigmoid(w))^2 * anchor_w, ... anchor_h).
"""
import torch
import torch.nn as nn

from ...layers.common import ConvBNSiLU
from ..effidehead import Detect as _AnchorFreeDetect


class Detect(_AnchorFreeDetect):
    def __init__(self, num_classes=80, anchors=None, num_layers=3, inplace=True, head_layers=None, use_dfl=True, reg_max=16):
        assert head_layers is not None
        # the parent registers stems / cls_convs / reg_convs / cls_preds / reg_preds from a 5-per-level list
        five = [head_layers[i * 7 + k] for i in range(num_layers) for k in range(5)]
        super().__init__(num_classes, num_layers, inplace, head_layers=five, use_dfl=use_dfl, reg_max=reg_max)
        self.na = len(anchors[0]) // 2 if isinstance(anchors, (list, tuple)) else anchors
        self.anchors_init = (torch.tensor(anchors) / self.stride[:, None]).reshape(self.nl, self.na, 2)
        self.cls_preds_ab = nn.ModuleList(head_layers[i * 7 + 5] for i in range(num_layers))
        self.reg_preds_ab = nn.ModuleList(head_layers[i * 7 + 6] for i in range(num_layers))

    def initialize_biases(self):
        super().initialize_biases()
        import math
        cls_bias = -math.log((1 - self.prior_prob) / self.prior_prob)
        for convs, value in ((self.cls_preds_ab, cls_bias), (self.reg_preds_ab, 1.0)):
            for conv in convs:
                conv.bias = nn.Parameter(torch.full_like(conv.bias.detach().view(-1), value), requires_grad=True)
                conv.weight = nn.Parameter(torch.zeros_like(conv.weight.detach()), requires_grad=True)

    def lower_train(self, tb, x):
        stems, cls_af, reg_af, cls_ab, reg_ab = [], [], [], [], []
        for i in range(self.nl):
            f = self.stems[i].lower(tb, x[i])
            stems.append(f)
            c = self.cls_convs[i].lower(tb, f)
            r = self.reg_convs[i].lower(tb, f)
            cab, rab = self.cls_preds_ab[i], self.reg_preds_ab[i]
            cls_ab.append(tb.conv(c, cab.weight, 1, bias=cab.bias))
            reg_ab.append(tb.conv(r, rab.weight, 1, bias=rab.bias))
            cp, rp = self.cls_preds[i], self.reg_preds[i]
            cls_af.append(tb.conv(c, cp.weight, 1, bias=cp.bias))
            reg_af.append(tb.conv(r, rp.weight, 1, bias=rp.bias))
        scores_af, distri_af = tb.head_pack(cls_af, reg_af, self.nc, self.reg_preds[0].out_channels)
        scores_ab, distri_ab = tb.head_pack_ab(cls_ab, reg_ab, self.nc, self.na, self.anchors_init)
        return stems, (scores_ab, distri_ab, scores_af, distri_af)


def build_effidehead_layer(channels_list, num_anchors, num_classes, reg_max=16, num_layers=3):
    '''Seven layers per level in the reference's order; children are named "0".."20" (three levels) like nn.Sequential(*layers).'''
    chx = [6, 8, 10] if num_layers == 3 else [8, 9, 10, 11]
    layers = []
    for ci in chx:
        c = channels_list[ci]
        layers += [ConvBNSiLU(in_channels=c, out_channels=c, kernel_size=1, stride=1),
                   ConvBNSiLU(in_channels=c, out_channels=c, kernel_size=3, stride=1),
                   ConvBNSiLU(in_channels=c, out_channels=c, kernel_size=3, stride=1),
                   nn.Conv2d(in_channels=c, out_channels=num_classes, kernel_size=1),
                   nn.Conv2d(in_channels=c, out_channels=4 * (reg_max + 1), kernel_size=1),
                   nn.Conv2d(in_channels=c, out_channels=num_classes * num_anchors, kernel_size=1),
                   nn.Conv2d(in_channels=c, out_channels=4 * num_anchors, kernel_size=1)]
    return nn.Sequential(*layers)
