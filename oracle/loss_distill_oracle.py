"""ORACLE (test infrastructure - never imported by the product path).

numpy restatement of the FORWARD VALUE of the self-distillation loss, ComputeLoss.__call__ of reference
yolov6/models/losses/loss_distill.py:59-208 (the HIP path does not implement it yet - this file and
its goldens are the groundwork):

  the detection loss of loss.py on the student's outputs (assigner, VarifocalLoss, IoU loss, DFL) with this file's
  normalisation rules (:178-183: class loss / target_scores_sum when that is > 0; BboxLoss :283-330: / target_scores_sum
  unless it is exactly 0), plus
  d_loss_cls   :210-221  KL(softmax(teacher / T) || softmax(student / T)) summed over all anchors, x T^2
                         (the "logits" are the POST-sigmoid class scores of both heads, as the reference passes them)
  d_loss_dfl   :349-359  on the positive anchors: KL of the teacher's vs the student's softened DFL bin distributions, summed
                         over bins, MEAN over (positives x 4 sides) - a scalar - times each positive's weight, summed,
                         / target_scores_sum (BboxLoss.forward :317-326), x T^2
  d_loss_cw    :222-246  (distill_feat) channel-wise feature distillation over the three neck maps:
                         sum KL(softmax_hw(t) || softmax_hw(s)) / (N C) per level
  distill_weightdecay    :193-197  ((1 - cos(epoch pi / max_epoch)) / 2) (0.01 - 1) + 1 on the three distillation terms
  loss = class (cls + d_cls w_class) + iou iou + dfl (dfl + d_dfl w_dfl) + cwd d_cw          loss weights 1.0 / 2.5 / 0.5 / 10.0

`pred_lrtb` given = the N / S variant, reference yolov6/models/losses/loss_distill_ns.py:59-200 (the head of
heads/effidehead_distill_ns.py has a fourth training output, plain (l, t, r, b) distances from `reg_preds`): no ATSS warm-up
(:96-104), and the IoU loss is the SUM of the DFL-decoded boxes' and the plain-distance boxes' losses (BboxLoss :265-325).

Pinned to the unmodified reference: tests/golden/gen_golden.py `lossdistill` -> tests/golden/lossdistill_*.npz (values and the
gradients the reference back-propagates to the student's scores, DFL logits and feature maps).
"""
import math

import numpy as np

from . import atss_oracle, tal_oracle
from .loss_oracle import bbox_decode, df_loss, f32, generate_anchors, iou_loss, preprocess, varifocal_terms


def _softmax64(x, axis):
    x = np.asarray(x, np.float64)
    x = x - x.max(axis, keepdims=True)
    e = np.exp(x)
    return e / e.sum(axis, keepdims=True)


def _log_softmax64(x, axis):
    x = np.asarray(x, np.float64)
    x = x - x.max(axis, keepdims=True)
    return x - np.log(np.exp(x).sum(axis, keepdims=True))


def kl_sum(p_teacher, p_student):
    """F.kl_div(log(p_student), p_teacher, reduction='none') = xlogy(t, t) - t log s, elementwise (0 where t == 0)."""
    t = np.asarray(p_teacher, np.float64)
    s = np.asarray(p_student, np.float64)
    with np.errstate(divide="ignore", invalid="ignore"):
        return np.where(t > 0, t * (np.log(t) - np.log(s)), 0.0)


def distill_loss_cls(scores_student, scores_teacher, num_classes, temperature):
    """loss_distill.py:210-221."""
    ps = _softmax64(np.reshape(scores_student, (-1, num_classes)) / temperature, 1)
    pt = _softmax64(np.reshape(scores_teacher, (-1, num_classes)) / temperature, 1)
    return float(kl_sum(pt, ps).sum()) * temperature ** 2


def distill_loss_dfl(dist_student_pos, dist_teacher_pos, temperature, reg_max=16):
    """loss_distill.py:349-359: a SCALAR (sum over bins, mean over positives x 4 sides), x T^2."""
    ps = _softmax64(np.reshape(dist_student_pos, (-1, reg_max + 1)) / temperature, 1)
    pt = _softmax64(np.reshape(dist_teacher_pos, (-1, reg_max + 1)) / temperature, 1)
    return float(kl_sum(pt, ps).sum(1).mean()) * temperature ** 2


def distill_loss_cw(s_feats, t_feats, temperature=1.0):
    """loss_distill.py:222-246: KL over the H*W positions of every (image, channel) plane, three levels."""
    total = 0.0
    for s, t in zip(s_feats[:3], t_feats[:3]):
        N, C, H, W = s.shape
        ls = _log_softmax64(np.reshape(s, (N, C, H * W)) / temperature, 2)
        lt = _log_softmax64(np.reshape(t, (N, C, H * W)) / temperature, 2)
        total += float((np.exp(lt) * (lt - ls)).sum()) * temperature * temperature / (N * C)
    return total


def compute_loss_distill(feat_sizes, pred_scores, pred_distri, t_pred_scores, t_pred_distri, s_feats, t_feats, targets,
                         epoch_num, max_epoch, temperature, batch_height, batch_width, fpn_strides=(8, 16, 32),
                         grid_cell_size=5.0, grid_cell_offset=0.5, num_classes=80, warmup_epoch=0, use_dfl=True, reg_max=16,
                         iou_type="giou", loss_weight=None, distill_feat=False, distill_weight=None, pred_lrtb=None):
    lw = loss_weight or {"class": 1.0, "iou": 2.5, "dfl": 0.5, "cwd": 10.0}
    dw = distill_weight or {"class": 1.0, "dfl": 1.0}
    pred_scores = np.asarray(pred_scores, f32)
    pred_distri = np.asarray(pred_distri, f32)
    t_pred_scores = np.asarray(t_pred_scores, f32)
    t_pred_distri = np.asarray(t_pred_distri, f32)
    B = pred_scores.shape[0]
    anchors, anchor_points, n_list, stride_t = generate_anchors(feat_sizes, fpn_strides, grid_cell_size, grid_cell_offset)
    scale = np.asarray([batch_width, batch_height, batch_width, batch_height], f32)
    tg = preprocess(targets, B, scale)
    gt_labels, gt_bboxes = tg[:, :, :1], tg[:, :, 1:]
    mask_gt = (gt_bboxes.sum(-1, keepdims=True) > 0).astype(f32)
    anchor_points_s = anchor_points / stride_t
    pred_bboxes = bbox_decode(anchor_points_s, pred_distri, use_dfl, reg_max)
    if epoch_num < warmup_epoch and pred_lrtb is None:
        tl, tb, ts, fg = atss_oracle.assign(anchors, n_list, gt_labels, gt_bboxes, mask_gt, pred_bboxes * stride_t,
                                            topk=9, num_classes=num_classes)
    else:
        tl, tb, ts, fg = tal_oracle.assign(pred_scores, pred_bboxes * stride_t, anchor_points, gt_labels, gt_bboxes,
                                           mask_gt, topk=13, num_classes=num_classes, alpha=1.0, beta=6.0)
    tb = (tb / stride_t).astype(f32)
    fg = fg.astype(bool)
    tl = np.where(fg, tl, num_classes)
    one_hot = np.zeros(pred_scores.shape, f32)
    bi, ai = np.nonzero(fg)
    one_hot[bi, ai, tl[bi, ai].astype(np.int64)] = 1
    loss_cls = float(varifocal_terms(pred_scores, ts, one_hot).sum(dtype=np.float64))
    ts_sum = float(ts.sum(dtype=np.float64))
    if ts_sum > 0:                                    # :181-183 (loss.py divides only when the sum exceeds 1)
        loss_cls /= ts_sum
    loss_iou = loss_dfl = d_loss_dfl = 0.0
    if int(fg.sum()) > 0:
        w = ts.sum(-1, dtype=f32)[fg][:, None]
        loss_iou = float((iou_loss(pred_bboxes[fg], tb[fg], iou_type) * w).sum(dtype=np.float64))
        if pred_lrtb is not None:                     # loss_distill_ns.py:93, :284-292: dist2bbox(..., 'xyxy') of the plain distances
            d = np.asarray(pred_lrtb, f32)
            boxes_lrtb = np.concatenate([anchor_points_s[None] - d[..., :2], anchor_points_s[None] + d[..., 2:]], -1).astype(f32)
            loss_iou += float((iou_loss(boxes_lrtb[fg], tb[fg], iou_type) * w).sum(dtype=np.float64))
        if ts_sum != 0:
            loss_iou /= ts_sum
        if use_dfl:
            lt = anchor_points_s[None] - tb[..., :2]
            rb = tb[..., 2:] - anchor_points_s[None]
            ltrb = np.clip(np.concatenate([lt, rb], -1), 0, reg_max - 0.01).astype(f32)
            s_pos = pred_distri.reshape(B, -1, 4, reg_max + 1)[fg]
            t_pos = t_pred_distri.reshape(B, -1, 4, reg_max + 1)[fg]
            loss_dfl = float((df_loss(s_pos, ltrb[fg], reg_max) * w).sum(dtype=np.float64))
            d_loss_dfl = distill_loss_dfl(s_pos, t_pos, temperature, reg_max) * float(w.sum(dtype=np.float64))
            if ts_sum != 0:
                loss_dfl /= ts_sum
                d_loss_dfl /= ts_sum
    d_loss_cls = distill_loss_cls(pred_scores, t_pred_scores, num_classes, temperature)
    d_loss_cw = distill_loss_cw(s_feats, t_feats) if distill_feat else 0.0
    decay = ((1 - math.cos(epoch_num * math.pi / max_epoch)) / 2) * (0.01 - 1) + 1
    d_loss_dfl *= decay
    d_loss_cls *= decay
    d_loss_cw *= decay
    cls_all = loss_cls + d_loss_cls * dw["class"]
    dfl_all = loss_dfl + d_loss_dfl * dw["dfl"]
    loss = lw["class"] * cls_all + lw["iou"] * loss_iou + lw["dfl"] * dfl_all + lw["cwd"] * d_loss_cw
    return dict(loss=loss, loss_items=np.array([lw["iou"] * loss_iou, lw["dfl"] * dfl_all, lw["class"] * cls_all,
                                                lw["cwd"] * d_loss_cw], np.float64),
                d_loss_cls=d_loss_cls, d_loss_dfl=d_loss_dfl, d_loss_cw=d_loss_cw, decay=decay, target_scores_sum=ts_sum,
                num_pos=int(fg.sum()))
