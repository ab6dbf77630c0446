"""ORACLE (test infrastructure - never imported by the product path).

numpy restatement of TaskAlignedAssigner.forward (reference
yolov6/assigners/tal_assigner.py:22-181) and its helpers in
yolov6/assigners/assigner_utils.py:25-89, in fp32, one image at a time (the reference's own
per-image path for n_max_boxes > 100, tal_assigner.py:55-63, shows the batch split changes
nothing).

Tie rule for top-k (torch.topk leaves it unspecified): larger value first, then lower anchor
index.  Checked against outputs of the unmodified reference on tie-free inputs
(tests/golden/gen_golden.py -> tests/golden/tal_*.npz).
"""
import numpy as np

f32 = np.float32


def iou_calculator(box1, box2, eps=1e-9):
    """assigner_utils.py:69-89 ; box1 [G,4] (gt), box2 [A,4] (pd) -> [G,A]"""
    b1 = box1[:, None, :].astype(f32)
    b2 = box2[None, :, :].astype(f32)
    x1y1 = np.maximum(b1[..., :2], b2[..., :2])
    x2y2 = np.minimum(b1[..., 2:], b2[..., 2:])
    overlap = np.clip(x2y2 - x1y1, 0, None).prod(-1, dtype=f32)
    a1 = np.clip(b1[..., 2:] - b1[..., :2], 0, None).prod(-1, dtype=f32)
    a2 = np.clip(b2[..., 2:] - b2[..., :2], 0, None).prod(-1, dtype=f32)
    union = a1 + a2 - overlap + f32(eps)
    return (overlap / union).astype(f32)


def select_candidates_in_gts(xy, gt, eps=1e-9):
    """assigner_utils.py:25-44 ; xy [A,2], gt [G,4] -> [G,A] in {0,1}"""
    lt = xy[None, :, :] - gt[:, None, :2]
    rb = gt[:, None, 2:] - xy[None, :, :]
    return (np.concatenate([lt, rb], -1).min(-1) > f32(eps)).astype(f32)


def topk_mask(metrics, k, valid):
    """select_topk_candidates tal_assigner.py:143-158 for one image: metrics [G,A], valid [G] bool."""
    G, A = metrics.shape
    out = np.zeros((G, A), f32)
    idx_all = np.arange(A)
    for g in range(G):
        if not valid[g]:
            continue  # indices forced to 0 -> counted k times -> zeroed by the `> 1` filter (:152-156)
        order = np.lexsort((idx_all, -metrics[g].astype(np.float64)))  # value desc, index asc
        out[g, order[:min(k, A)]] = 1.0
    return out


def assign_image(pd_scores, pd_bboxes, anc_points, gt_labels, gt_bboxes, mask_gt, topk=13, alpha=1.0, beta=6.0,
                 eps=1e-9):
    """One image.  pd_scores [A,C], pd_bboxes [A,4], anc_points [A,2], gt_labels [G], gt_bboxes [G,4], mask_gt [G]."""
    A, C = pd_scores.shape
    G = gt_bboxes.shape[0]
    lab = gt_labels.astype(np.int64)
    bbox_scores = pd_scores[:, lab].T.astype(f32)                                   # :131-135  [G,A]
    overlaps = iou_calculator(gt_bboxes, pd_bboxes)                                 # :137
    sa = bbox_scores if alpha == 1.0 else np.power(bbox_scores, f32(alpha))
    align = (sa * np.power(overlaps, f32(beta))).astype(f32)                        # :138
    in_gts = select_candidates_in_gts(anc_points, gt_bboxes)                        # :114
    mtopk = topk_mask(align * in_gts, topk, mask_gt > 0)                            # :116-117
    mask_pos = mtopk * in_gts * (mask_gt[:, None] > 0)                              # :119
    fg = mask_pos.sum(0)                                                            # assigner_utils.py:58
    multi = fg > 1
    if multi.any():                                                                 # :59-65
        best = overlaps.argmax(0)
        onehot = np.zeros_like(mask_pos)
        onehot[best, np.arange(A)] = 1.0
        mask_pos = np.where(multi[None, :], onehot, mask_pos)
        fg = mask_pos.sum(0)
    tgt = mask_pos.argmax(0)                                                        # :66
    labels = lab[tgt].copy()                                                        # tal_assigner.py:166-168
    bboxes = gt_bboxes[tgt].astype(f32)                                             # :171
    labels[labels < 0] = 0                                                          # :174
    scores = np.zeros((A, C), f32)                                                  # :175-178
    pos = fg > 0
    scores[np.nonzero(pos)[0], labels[pos]] = 1.0
    align = align * mask_pos                                                        # :76
    pos_align = align.max(-1, keepdims=True)                                        # :77
    pos_ov = (overlaps * mask_pos).max(-1, keepdims=True)                           # :78
    norm = (align * pos_ov / (pos_align + f32(eps))).max(0)                         # :79
    scores = scores * norm[:, None].astype(f32)                                     # :80
    return labels, bboxes, scores.astype(f32), pos


def assign(pd_scores, pd_bboxes, anc_points, gt_labels, gt_bboxes, mask_gt, topk=13, num_classes=80, alpha=1.0,
           beta=6.0, eps=1e-9):
    """Batch API mirroring TaskAlignedAssigner.forward: arrays shaped like the reference's tensors."""
    pd_scores, pd_bboxes = np.asarray(pd_scores, f32), np.asarray(pd_bboxes, f32)
    anc_points, gt_bboxes = np.asarray(anc_points, f32), np.asarray(gt_bboxes, f32)
    B, A, C = pd_scores.shape
    G = gt_bboxes.shape[1]
    if G == 0:                                                                      # :48-53
        return (np.full((B, A), num_classes, np.int64), np.zeros((B, A, 4), f32), np.zeros((B, A, C), f32),
                np.zeros((B, A), bool))
    gl = np.asarray(gt_labels, f32).reshape(B, G)
    mg = np.asarray(mask_gt, f32).reshape(B, G)
    L, Bx, S, F = [], [], [], []
    for b in range(B):
        l, bx, s, f = assign_image(pd_scores[b], pd_bboxes[b], anc_points, gl[b], gt_bboxes[b], mg[b], topk, alpha,
                                   beta, eps)
        L.append(l); Bx.append(bx); S.append(s); F.append(f)
    return np.stack(L), np.stack(Bx), np.stack(S), np.stack(F)
