"""ORACLE (test infrastructure - never imported by the product path).

numpy restatement of ATSSAssigner.forward (reference yolov6/assigners/atss_assigner.py:18-161),
bbox_overlaps (yolov6/assigners/iou2d_calculator.py:186-241, mode 'iou', eps 1e-6) and dist_calculator
(yolov6/assigners/assigner_utils.py:4-23), one image at a time, fp32.

Tie rule for the per-level top-k nearest anchors (torch.topk leaves it unspecified): smaller distance,
then lower anchor index.  Checked against outputs of the unmodified reference
(tests/golden/gen_golden.py -> tests/golden/atss_*.npz).
"""
import numpy as np

from .tal_oracle import iou_calculator, select_candidates_in_gts

f32 = np.float32


def bbox_overlaps(b1, b2, eps=1e-6):
    """[G,4] x [A,4] -> [G,A]"""
    b1, b2 = b1.astype(f32), b2.astype(f32)
    area1 = (b1[:, 2] - b1[:, 0]) * (b1[:, 3] - b1[:, 1])
    area2 = (b2[:, 2] - b2[:, 0]) * (b2[:, 3] - b2[:, 1])
    lt = np.maximum(b1[:, None, :2], b2[None, :, :2])
    rb = np.minimum(b1[:, None, 2:], b2[None, :, 2:])
    wh = np.clip(rb - lt, 0, None)
    overlap = wh[..., 0] * wh[..., 1]
    union = np.maximum(area1[:, None] + area2[None, :] - overlap, f32(eps))
    return (overlap / union).astype(f32)


def assign_image(anc, n_level, gt_labels, gt_bboxes, mask_gt, pd_bboxes, topk, num_classes):
    A, G = anc.shape[0], gt_bboxes.shape[0]
    overlaps = bbox_overlaps(gt_bboxes, anc)                                        # :56
    gc = np.stack([(gt_bboxes[:, 0] + gt_bboxes[:, 2]) / f32(2), (gt_bboxes[:, 1] + gt_bboxes[:, 3]) / f32(2)], 1)
    ac = np.stack([(anc[:, 0] + anc[:, 2]) / f32(2), (anc[:, 1] + anc[:, 3]) / f32(2)], 1).astype(f32)
    dist = np.sqrt(((gc[:, None, :] - ac[None, :, :]).astype(f32) ** 2).sum(-1, dtype=f32)).astype(f32)   # :59
    is_cand = np.zeros((G, A), f32)
    cand_idx = []
    start = 0
    for n in n_level:                                                               # :96-111
        k = min(topk, n)
        lvl = dist[:, start:start + n]
        order = np.lexsort((np.broadcast_to(np.arange(n), lvl.shape), lvl.astype(np.float64)), axis=-1)[:, :k]
        cand_idx.append(order + start)
        for g in range(G):
            if mask_gt[g] > 0:
                is_cand[g, order[g] + start] = 1.0
        start += n
    cand_idx = np.concatenate(cand_idx, 1)                                          # [G, sum k]
    cand_ov = np.where(is_cand > 0, overlaps, f32(0))                               # :123-124
    co = np.take_along_axis(cand_ov, cand_idx, 1).astype(f32)                       # :125-130
    mean = co.mean(-1, keepdims=True, dtype=f32)
    std = co.astype(np.float64).std(-1, ddof=1, keepdims=True).astype(f32) if co.shape[1] > 1 else np.full_like(mean, np.nan)
    thr = mean + std                                                                # :132-134
    with np.errstate(invalid="ignore"):
        is_pos = np.where(cand_ov > thr, is_cand, f32(0))                           # :65-67
    in_gts = select_candidates_in_gts(ac, gt_bboxes)                                # :69
    mask_pos = is_pos * in_gts * (mask_gt[:, None] > 0)                             # :70
    fg = mask_pos.sum(0)
    multi = fg > 1
    if multi.any():                                                                 # assigner_utils.py:59-65
        best = overlaps.argmax(0)
        onehot = np.zeros_like(mask_pos)
        onehot[best, np.arange(A)] = 1.0
        mask_pos = np.where(multi[None, :], onehot, mask_pos)
        fg = mask_pos.sum(0)
    tgt = mask_pos.argmax(0)
    pos = fg > 0
    labels = np.where(pos, gt_labels[tgt].astype(np.int64), num_classes)            # :146-149
    bboxes = gt_bboxes[tgt].astype(f32)                                             # :152-153
    scores = np.zeros((A, num_classes), f32)                                        # :156-157
    scores[np.nonzero(pos)[0], labels[pos]] = 1.0
    if pd_bboxes is not None:                                                       # :80-84
        ious = (iou_calculator(gt_bboxes, pd_bboxes) * mask_pos).max(0)
        scores = scores * ious[:, None].astype(f32)
    return labels, bboxes, scores, pos


def assign(anc_bboxes, n_level_bboxes, gt_labels, gt_bboxes, mask_gt, pd_bboxes, topk=9, num_classes=80):
    anc = np.asarray(anc_bboxes, f32)
    gt_bboxes = np.asarray(gt_bboxes, f32)
    B, G = gt_bboxes.shape[:2]
    A = anc.shape[0]
    if G == 0:                                                                      # :48-53
        return (np.full((B, A), num_classes, np.int64), np.zeros((B, A, 4), f32),
                np.zeros((B, A, num_classes), f32), np.zeros((B, A), bool))
    gl = np.asarray(gt_labels, f32).reshape(B, G)
    mg = np.asarray(mask_gt, f32).reshape(B, G)
    out = [assign_image(anc, list(n_level_bboxes), gl[b], gt_bboxes[b], mg[b],
                        None if pd_bboxes is None else np.asarray(pd_bboxes[b], f32), topk, num_classes)
           for b in range(B)]
    return tuple(np.stack([o[i] for o in out]) for i in range(4))
