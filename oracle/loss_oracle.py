"""ORACLE (test infrastructure - never imported by the product path).

numpy restatement of the FORWARD VALUE of ComputeLoss.__call__ (reference
yolov6/models/losses/loss.py:52-182) and its parts:
  preprocess      loss.py:184-192   target packing  [N,6]=(img, cls, cx,cy,w,h in 0..1) -> [B,G,5] (cls, xyxy px)
  bbox_decode     loss.py:194-198   (+ dist2bbox general.py:32-43)
  VarifocalLoss   loss.py:201-211   BCE(p, q) * (alpha p^gamma (1-y) + q y), alpha .75 gamma 2, fp32
  BboxLoss        loss.py:214-278   IoU loss (figure_iou.py:7-100, box_format xyxy, eps 1e-10) x sum(target_scores),
                                    DFL two-bin cross entropy (:267-278), both / target_scores_sum when that is > 1
  loss weights    class 1.0, iou 2.5, dfl 0.5
The assigners are the ones of oracle/atss_oracle.py (epoch < warmup_epoch) and oracle/tal_oracle.py.
All arithmetic fp32; the final sums are taken in float64 (the reference's fp32 tree reductions agree to ~1e-6
relative, which is the tolerance the golden check uses).  Checked against the unmodified reference
(tests/golden/gen_golden.py -> tests/golden/loss_*.npz).
"""
import math

import numpy as np

from . import atss_oracle, tal_oracle

f32 = np.float32


def preprocess(targets, batch_size, scale):
    """loss.py:184-192.  targets [N,6] -> [B,G,5] (label, x1,y1,x2,y2); padding rows are (-1, 0,0,0,0)."""
    lists = [[] for _ in range(batch_size)]
    for row in np.asarray(targets, dtype=np.float64).tolist():
        lists[int(row[0])].append(row[1:])
    max_len = max([len(l) for l in lists] + [0])
    out = np.zeros((batch_size, max_len, 5), f32)
    out[:, :, 0] = -1
    for b, l in enumerate(lists):
        if l:
            out[b, :len(l)] = np.asarray(l, f32)
    box = out[:, :, 1:5] * np.asarray(scale, f32)
    xyxy = np.empty_like(box)
    # xywh2xyxy as the reference writes it (general.py:52-58, in place): x2 = x1 + w, not x + w/2
    xyxy[..., 0] = box[..., 0] - box[..., 2] * f32(0.5)
    xyxy[..., 1] = box[..., 1] - box[..., 3] * f32(0.5)
    xyxy[..., 2] = xyxy[..., 0] + box[..., 2]
    xyxy[..., 3] = xyxy[..., 1] + box[..., 3]
    out[:, :, 1:5] = xyxy
    return out


def bbox_decode(anchor_points_s, pred_dist, use_dfl, reg_max):
    """loss.py:194-198: softmax over the reg_max+1 bins . linspace(0, reg_max), then dist2bbox (xyxy)."""
    pred_dist = pred_dist.astype(f32)
    if use_dfl:
        B, A, _ = pred_dist.shape
        d = pred_dist.reshape(B, A, 4, reg_max + 1)
        d = d - d.max(-1, keepdims=True)
        e = np.exp(d, dtype=f32)
        p = e / e.sum(-1, keepdims=True, dtype=f32)
        proj = np.linspace(0, reg_max, reg_max + 1, dtype=f32)
        pred_dist = (p * proj).sum(-1, dtype=f32)
    lt, rb = pred_dist[..., :2], pred_dist[..., 2:]
    return np.concatenate([anchor_points_s - lt, anchor_points_s + rb], -1).astype(f32)


def varifocal_terms(pred_score, gt_score, label, alpha=0.75, gamma=2.0):
    """loss.py:205-209 elementwise (before the sum).  torch's BCE clamps both logs at -100."""
    p = pred_score.astype(f32)
    q = gt_score.astype(f32)
    y = label.astype(f32)
    weight = f32(alpha) * (p ** f32(gamma)) * (f32(1) - y) + q * y
    with np.errstate(divide="ignore"):
        logp = np.maximum(np.log(p, dtype=f32), f32(-100))
        log1p = np.maximum(np.log(f32(1) - p, dtype=f32), f32(-100))
    bce = -(q * logp + (f32(1) - q) * log1p)
    return bce * weight


def iou_loss(box1, box2, iou_type="giou", eps=1e-10):
    """figure_iou.py:24-100 for box_format 'xyxy', same-shape [M,4] inputs; returns [M,1] (1 - IoU variant)."""
    e = f32(eps)
    b1 = box1.astype(f32)
    b2 = box2.astype(f32)
    b1_x1, b1_y1, b1_x2, b1_y2 = (b1[:, i:i + 1] for i in range(4))
    b2_x1, b2_y1, b2_x2, b2_y2 = (b2[:, i:i + 1] for i in range(4))
    inter = np.clip(np.minimum(b1_x2, b2_x2) - np.maximum(b1_x1, b2_x1), 0, None) * \
        np.clip(np.minimum(b1_y2, b2_y2) - np.maximum(b1_y1, b2_y1), 0, None)
    w1, h1 = b1_x2 - b1_x1, b1_y2 - b1_y1 + e
    w2, h2 = b2_x2 - b2_x1, b2_y2 - b2_y1 + e
    union = w1 * h1 + w2 * h2 - inter + e
    iou = inter / union
    cw = np.maximum(b1_x2, b2_x2) - np.minimum(b1_x1, b2_x1)
    ch = np.maximum(b1_y2, b2_y2) - np.minimum(b1_y1, b2_y1)
    if iou_type == "giou":
        c_area = cw * ch + e
        iou = iou - (c_area - union) / c_area
    elif iou_type in ("diou", "ciou"):
        c2 = cw ** 2 + ch ** 2 + e
        rho2 = ((b2_x1 + b2_x2 - b1_x1 - b1_x2) ** 2 + (b2_y1 + b2_y2 - b1_y1 - b1_y2) ** 2) / f32(4)
        if iou_type == "diou":
            iou = iou - rho2 / c2
        else:
            v = f32(4 / math.pi ** 2) * (np.arctan(w2 / h2) - np.arctan(w1 / h1)) ** 2
            alpha = v / (v - iou + (f32(1) + e))
            iou = iou - (rho2 / c2 + v * alpha)
    elif iou_type == "siou":
        s_cw = (b2_x1 + b2_x2 - b1_x1 - b1_x2) * f32(0.5) + e
        s_ch = (b2_y1 + b2_y2 - b1_y1 - b1_y2) * f32(0.5) + e
        sigma = np.sqrt(s_cw ** 2 + s_ch ** 2)
        sin_alpha_1 = np.abs(s_cw) / sigma
        sin_alpha_2 = np.abs(s_ch) / sigma
        threshold = f32(2 ** 0.5 / 2)
        sin_alpha = np.where(sin_alpha_1 > threshold, sin_alpha_2, sin_alpha_1)
        angle_cost = np.cos(np.arcsin(sin_alpha) * f32(2) - f32(math.pi / 2))
        rho_x = (s_cw / cw) ** 2
        rho_y = (s_ch / ch) ** 2
        gamma = angle_cost - f32(2)
        distance_cost = f32(2) - np.exp(gamma * rho_x) - np.exp(gamma * rho_y)
        omiga_w = np.abs(w1 - w2) / np.maximum(w1, w2)
        omiga_h = np.abs(h1 - h2) / np.maximum(h1, h2)
        shape_cost = (f32(1) - np.exp(-omiga_w)) ** 4 + (f32(1) - np.exp(-omiga_h)) ** 4
        iou = iou - f32(0.5) * (distance_cost + shape_cost)
    else:
        raise ValueError(iou_type)
    return (f32(1) - iou).astype(f32)


def df_loss(pred_dist, target, reg_max):
    """loss.py:267-278.  pred_dist [M,4,R+1] logits, target [M,4] in [0, R-0.01] -> [M,1]."""
    tl = target.astype(np.int64)
    tr = tl + 1
    wl = tr.astype(f32) - target.astype(f32)
    wr = f32(1) - wl
    x = pred_dist.astype(f32)
    m = x.max(-1, keepdims=True)
    lse = (np.log(np.exp(x - m, dtype=f32).sum(-1, dtype=f32), dtype=f32) + m[..., 0]).astype(f32)
    ce_l = lse - np.take_along_axis(x, tl[..., None], -1)[..., 0]
    ce_r = lse - np.take_along_axis(x, tr[..., None], -1)[..., 0]
    return ((ce_l * wl + ce_r * wr).mean(-1, keepdims=True, dtype=f32)).astype(f32)


def generate_anchors(feat_sizes, strides, grid_cell_size=5.0, grid_cell_offset=0.5, rep=1):
    """Train-form anchors (anchor_generator.py:35-63): anchors [A,4], points [A,2] px, per-level counts, stride [A,1].
    rep=3 is mode='ab' (:52-60): every level's grid repeated three times (anchor-major), as loss_fuseab.py:60-61 asks."""
    anchors, points, n_list, stride_t = [], [], [], []
    for (h, w), s in zip(feat_sizes, strides):
        cell_half = grid_cell_size * s * 0.5
        sx = (np.arange(w, dtype=f32) + f32(grid_cell_offset)) * f32(s)
        sy = (np.arange(h, dtype=f32) + f32(grid_cell_offset)) * f32(s)
        yy, xx = np.meshgrid(sy, sx, indexing="ij")
        a = np.stack([xx - f32(cell_half), yy - f32(cell_half), xx + f32(cell_half), yy + f32(cell_half)], -1)
        anchors.append(np.tile(a.reshape(-1, 4).astype(f32), (rep, 1)))
        points.append(np.tile(np.stack([xx, yy], -1).reshape(-1, 2).astype(f32), (rep, 1)))
        n_list.append(h * w * rep)
        stride_t.append(np.full((h * w * rep, 1), s, f32))
    return np.concatenate(anchors), np.concatenate(points), n_list, np.concatenate(stride_t)


def compute_loss(feat_sizes, pred_scores, pred_distri, targets, epoch_num, batch_height, batch_width,
                 fpn_strides=(8, 16, 32), grid_cell_size=5.0, grid_cell_offset=0.5, num_classes=80, warmup_epoch=4,
                 use_dfl=True, reg_max=16, iou_type="giou", loss_weight=None, ab=False):
    """loss.py:52-182.  Returns dict(loss, loss_items[iou, dfl, cls] (weighted), plus the intermediates).
    ab=True: the anchor-based variant, loss_fuseab.py:39-147 (anchors x3, boxes from (dx,dy,w,h), TAL topk 26, no warm-up)."""
    lw = loss_weight or {"class": 1.0, "iou": 2.5, "dfl": 0.5}
    pred_scores = np.asarray(pred_scores, f32)
    pred_distri = np.asarray(pred_distri, f32)
    B = pred_scores.shape[0]
    anchors, anchor_points, n_list, stride_t = generate_anchors(feat_sizes, fpn_strides, grid_cell_size, grid_cell_offset,
                                                                 rep=3 if ab else 1)
    scale = np.asarray([batch_width, batch_height, batch_width, batch_height], f32)
    tg = preprocess(targets, B, scale)
    gt_labels, gt_bboxes = tg[:, :, :1], tg[:, :, 1:]
    mask_gt = (gt_bboxes.sum(-1, keepdims=True) > 0).astype(f32)
    anchor_points_s = anchor_points / stride_t
    if ab:        # loss_fuseab.py:75-76 (+ general.py:52-58: x2 = x1 + w)
        cxy = pred_distri[..., :2] + anchor_points_s
        x1y1 = cxy - pred_distri[..., 2:] * f32(0.5)
        pred_bboxes = np.concatenate([x1y1, x1y1 + pred_distri[..., 2:]], -1).astype(f32)
    else:
        pred_bboxes = bbox_decode(anchor_points_s, pred_distri, use_dfl, reg_max)
    if epoch_num < warmup_epoch and not ab:
        tl, tb, ts, fg = atss_oracle.assign(anchors, n_list, gt_labels, gt_bboxes, mask_gt, pred_bboxes * stride_t,
                                            topk=9, num_classes=num_classes)
    else:
        tl, tb, ts, fg = tal_oracle.assign(pred_scores, pred_bboxes * stride_t, anchor_points, gt_labels, gt_bboxes,
                                           mask_gt, topk=26 if ab else 13, num_classes=num_classes, alpha=1.0, beta=6.0)
    tb = (tb / stride_t).astype(f32)
    fg = fg.astype(bool)
    tl = np.where(fg, tl, num_classes)
    one_hot = np.zeros(pred_scores.shape, f32)
    bi, ai = np.nonzero(fg)
    one_hot[bi, ai, tl[bi, ai].astype(np.int64)] = 1
    loss_cls = float(varifocal_terms(pred_scores, ts, one_hot).sum(dtype=np.float64))
    ts_sum = float(ts.sum(dtype=np.float64))
    if ts_sum > 1:
        loss_cls /= ts_sum
    num_pos = int(fg.sum())
    loss_iou = loss_dfl = 0.0
    if num_pos > 0:
        w = ts.sum(-1, dtype=f32)[fg][:, None]
        li = iou_loss(pred_bboxes[fg], tb[fg], iou_type) * w
        loss_iou = float(li.sum(dtype=np.float64))
        if ts_sum > 1:
            loss_iou /= ts_sum
        if use_dfl:
            lt = anchor_points_s[None] - tb[..., :2]
            rb = tb[..., 2:] - anchor_points_s[None]
            ltrb = np.clip(np.concatenate([lt, rb], -1), 0, reg_max - 0.01).astype(f32)   # bbox2dist general.py:45-49
            ld = df_loss(pred_distri.reshape(B, -1, 4, reg_max + 1)[fg], ltrb[fg], reg_max) * w
            loss_dfl = float(ld.sum(dtype=np.float64))
            if ts_sum > 1:
                loss_dfl /= ts_sum
    loss = lw["class"] * loss_cls + lw["iou"] * loss_iou + lw["dfl"] * loss_dfl
    return dict(loss=loss, loss_items=np.array([lw["iou"] * loss_iou, lw["dfl"] * loss_dfl, lw["class"] * loss_cls], f32),
                target_scores_sum=ts_sum, num_pos=num_pos, pred_bboxes=pred_bboxes, target_labels=tl,
                target_bboxes=tb, target_scores=ts, fg_mask=fg)
