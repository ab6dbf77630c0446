"""ORACLE (test infrastructure - never imported by the product path).

Gradient of the training loss wrt the head outputs: a torch-autograd restatement (float64) of the differentiable
part of ComputeLoss.__call__ (reference yolov6/models/losses/loss.py):
    bbox_decode   :194-198  softmax over the DFL bins . proj, dist2bbox xyxy (utils/general.py:32-43)
    VarifocalLoss :201-211  weight = alpha p^gamma (1-y) + q y  (NOT detached: it carries gradient), BCE in fp32
    BboxLoss      :214-278  IOUloss (utils/figure_iou.py:53-95; CIoU's alpha under no_grad :76-77), DFL :267-278
    normalisation :168-169, :238-261 and weights :171-181
The label assignment is not differentiable (`.detach()` at :91-103); it is taken from the numpy oracle
(oracle/loss_oracle.compute_loss -> target_labels / target_bboxes / target_scores / fg_mask).
Pinned by tests/test_oracle_cpu.py against gradients back-propagated by the unmodified reference
(tests/golden/lossgrad_*.npz, written by tests/golden/gen_golden.py).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from . import loss_oracle


def iou_loss(b1, b2, iou_type, eps=1e-10):
    x1, y1, x2, y2 = b1.unbind(-1)
    tx1, ty1, tx2, ty2 = b2.unbind(-1)
    inter = (torch.min(x2, tx2) - torch.max(x1, tx1)).clamp(0) * (torch.min(y2, ty2) - torch.max(y1, ty1)).clamp(0)
    w1, h1 = x2 - x1, y2 - y1 + eps
    w2, h2 = tx2 - tx1, ty2 - ty1 + eps
    union = w1 * h1 + w2 * h2 - inter + eps
    iou = inter / union
    cw = torch.max(x2, tx2) - torch.min(x1, tx1)
    ch = torch.max(y2, ty2) - torch.min(y1, ty1)
    if iou_type == "giou":
        c_area = cw * ch + eps
        iou = iou - (c_area - union) / c_area
    elif iou_type in ("diou", "ciou"):
        c2 = cw ** 2 + ch ** 2 + eps
        rho2 = ((tx1 + tx2 - x1 - x2) ** 2 + (ty1 + ty2 - y1 - y2) ** 2) / 4
        if iou_type == "diou":
            iou = iou - rho2 / c2
        else:
            v = (4 / math.pi ** 2) * (torch.atan(w2 / h2) - torch.atan(w1 / h1)) ** 2
            with torch.no_grad():
                alpha = v / (v - iou + (1 + eps))
            iou = iou - (rho2 / c2 + v * alpha)
    elif iou_type == "siou":
        s_cw = (tx1 + tx2 - x1 - x2) * 0.5 + eps
        s_ch = (ty1 + ty2 - y1 - y2) * 0.5 + eps
        sigma = (s_cw ** 2 + s_ch ** 2) ** 0.5
        sa1, sa2 = s_cw.abs() / sigma, s_ch.abs() / sigma
        sa = torch.where(sa1 > 2 ** 0.5 / 2, sa2, sa1)
        angle = torch.cos(torch.arcsin(sa) * 2 - math.pi / 2)
        gamma = angle - 2
        dist = 2 - torch.exp(gamma * (s_cw / cw) ** 2) - torch.exp(gamma * (s_ch / ch) ** 2)
        ow = (w1 - w2).abs() / torch.max(w1, w2)
        oh = (h1 - h2).abs() / torch.max(h1, h2)
        shape = (1 - torch.exp(-ow)) ** 4 + (1 - torch.exp(-oh)) ** 4
        iou = iou - 0.5 * (dist + shape)
    return 1.0 - iou


def loss_and_grads(pred_scores, pred_distri, anchor_points_s, target_labels, target_bboxes_s, target_scores, fg_mask,
                   num_classes, use_dfl, reg_max, iou_type, loss_weight=None, dtype=torch.float64, ab=False):
    """target_bboxes_s: already divided by the stride (loss.py:154).  Returns (loss, items[iou,dfl,cls], dscores, ddistri)."""
    lw = loss_weight or {"class": 1.0, "iou": 2.5, "dfl": 0.5}
    ps = torch.as_tensor(np.asarray(pred_scores), dtype=dtype).clone().requires_grad_(True)
    pdist = torch.as_tensor(np.asarray(pred_distri), dtype=dtype).clone().requires_grad_(True)
    pts = torch.as_tensor(np.asarray(anchor_points_s), dtype=dtype)
    tl = torch.as_tensor(np.asarray(target_labels)).long()
    tb = torch.as_tensor(np.asarray(target_bboxes_s), dtype=dtype)
    ts = torch.as_tensor(np.asarray(target_scores), dtype=dtype)
    fg = torch.as_tensor(np.asarray(fg_mask)).bool()
    B, A, _ = ps.shape
    if ab:        # loss_fuseab.py:75-76: (dx, dy, w, h) around the anchor point; xywh2xyxy with x2 = x1 + w
        x1y1 = pdist[..., :2] + pts - pdist[..., 2:] * 0.5
        pb = torch.cat([x1y1, x1y1 + pdist[..., 2:]], -1)
    else:
        if use_dfl:
            prob = F.softmax(pdist.view(B, A, 4, reg_max + 1), dim=-1)
            dist = prob.matmul(torch.linspace(0, reg_max, reg_max + 1, dtype=dtype))
        else:
            dist = pdist
        pb = torch.cat([pts - dist[..., :2], pts + dist[..., 2:]], -1)
    tl = torch.where(fg, tl, torch.full_like(tl, num_classes))
    one_hot = F.one_hot(tl, num_classes + 1)[..., :-1].to(dtype)
    weight = 0.75 * ps.pow(2.0) * (1 - one_hot) + ts * one_hot
    loss_cls = (F.binary_cross_entropy(ps, ts, reduction="none") * weight).sum()
    ts_sum = ts.sum()
    if ts_sum > 1:
        loss_cls = loss_cls / ts_sum
    loss_iou = loss_dfl = pdist.sum() * 0.0
    if int(fg.sum()) > 0:
        w = ts.sum(-1)[fg].unsqueeze(-1)
        li = iou_loss(pb[fg], tb[fg], iou_type).unsqueeze(-1) * w
        loss_iou = li.sum() / ts_sum if ts_sum > 1 else li.sum()
        if use_dfl:
            ltrb = torch.cat([pts - tb[..., :2], tb[..., 2:] - pts], -1).clip(0, reg_max - 0.01)[fg]
            logits = pdist.view(B, A, 4, reg_max + 1)[fg]
            left = ltrb.long()
            right = left + 1
            wl = right.to(dtype) - ltrb
            wr = 1 - wl
            ce_l = F.cross_entropy(logits.reshape(-1, reg_max + 1), left.reshape(-1), reduction="none").view(left.shape)
            ce_r = F.cross_entropy(logits.reshape(-1, reg_max + 1), right.reshape(-1), reduction="none").view(left.shape)
            ld = (ce_l * wl + ce_r * wr).mean(-1, keepdim=True) * w
            loss_dfl = ld.sum() / ts_sum if ts_sum > 1 else ld.sum()
    loss = lw["class"] * loss_cls + lw["iou"] * loss_iou + lw["dfl"] * loss_dfl
    loss.backward()
    items = np.array([float(lw["iou"] * loss_iou), float(lw["dfl"] * loss_dfl), float(lw["class"] * loss_cls)])
    return float(loss), items, ps.grad.numpy(), pdist.grad.numpy()


def compute_loss_with_grads(feat_sizes, pred_scores, pred_distri, targets, epoch_num, batch_height, batch_width, **kw):
    """The whole ComputeLoss call: numpy oracle for the assignment, autograd restatement for the value and gradients."""
    r = loss_oracle.compute_loss(feat_sizes, pred_scores, pred_distri, targets, epoch_num, batch_height, batch_width, **kw)
    strides = kw.get("fpn_strides", (8, 16, 32))
    _, pts, _, st = loss_oracle.generate_anchors(feat_sizes, strides, kw.get("grid_cell_size", 5.0), kw.get("grid_cell_offset", 0.5),
                                                 rep=3 if kw.get("ab") else 1)
    out = loss_and_grads(pred_scores, pred_distri, pts / st, r["target_labels"], r["target_bboxes"], r["target_scores"], r["fg_mask"],
                         kw.get("num_classes", 80), kw.get("use_dfl", True), kw.get("reg_max", 16), kw.get("iou_type", "giou"),
                         kw.get("loss_weight"), ab=bool(kw.get("ab")))
    return dict(loss=out[0], loss_items=out[1], dscores=out[2], ddistri=out[3], assign=r)
