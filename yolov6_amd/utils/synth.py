"""Synthetic data: deterministic weights, images, head-shaped predictions and assigner inputs
(no datasets or checkpoints are reachable; bench.py and the tests need realistic shapes).

Deterministic, key-addressed weights: every state_dict entry is generated from
crc32(key) ^ seed, so the reference model (in tests/golden/gen_golden.py), the oracle and
the HIP model all receive bit-identical parameters without shipping checkpoints.

Why not the default init: Detect.initialize_biases zero-fills the prediction convs
(reference effidehead.py:51-65) and fresh BatchNorms are identity, so default-initialised
models prove nothing.
"""
import zlib

import numpy as np
import torch


def _rng(key, seed):
    return np.random.RandomState((zlib.crc32(key.encode()) ^ (seed * 2654435761)) & 0x7FFFFFFF)


def synth_tensor(key, shape, dtype, seed=0):
    r = _rng(key, seed)
    shape = tuple(shape)
    leaf = key.rsplit(".", 1)[-1]
    if leaf == "num_batches_tracked":
        return torch.zeros(shape, dtype=dtype)
    if leaf == "proj" or key.endswith("proj_conv.weight"):
        n = shape[0] if leaf == "proj" else shape[1]
        return torch.linspace(0, n - 1, n).reshape(shape).to(dtype)
    if leaf == "alpha":
        v = r.uniform(0.5, 1.5, size=shape)
    elif leaf == "running_var":
        v = r.uniform(0.5, 1.5, size=shape)
    elif leaf == "running_mean":
        v = r.normal(0.0, 0.1, size=shape)
    elif len(shape) == 4:  # conv / conv-transpose kernels: He-style scale keeps activations O(1)
        fan_in = shape[1] * shape[2] * shape[3]
        if "upsample_transpose" in key:
            fan_in = shape[0]
        if "_preds." in key:
            gain = 1.0
        elif ".rbr_" in key:      # three summed branches per RepVGG block: share the variance budget
            gain = 0.72
        else:
            gain = 1.3
        v = r.normal(0.0, gain / np.sqrt(fan_in), size=shape)
    elif leaf == "weight" and ".rbr_identity." in key:   # identity-branch BatchNorm gamma
        v = r.uniform(0.2, 0.5, size=shape)
    elif leaf == "weight" and key.split(".")[-2:-1] == ["bn"] and (key.split(".")[-3:-2] or [""])[0] not in ("block", "rbr_dense", "rbr_1x1"):
        v = r.uniform(0.25, 0.55, size=shape)            # QARepVGG post-BN gamma (raw identity adds variance)
    elif leaf == "weight":  # BatchNorm gamma
        v = r.uniform(0.5, 1.5, size=shape)
    elif leaf == "bias":
        if "cls_preds" in key:
            v = r.normal(-3.0, 0.5, size=shape)
        elif "reg_preds" in key:
            v = r.normal(1.0, 0.3, size=shape)
        else:
            v = r.normal(0.0, 0.1, size=shape)
    else:
        v = r.normal(0.0, 0.1, size=shape)
    return torch.from_numpy(np.asarray(v, dtype=np.float32)).to(dtype)


def synth_state_dict(template, seed=0):
    """template: {key: tensor} (any state_dict).  Returns new tensors with the same keys/shapes/dtypes."""
    return {k: synth_tensor(k, v.shape, v.dtype, seed) for k, v in template.items()}


def synth_images(batch, size, seed=0, channels=3):
    """Post-`/255` domain image batch, NCHW fp32 in [0,1)."""
    g = torch.Generator().manual_seed(seed)
    if isinstance(size, int):
        size = (size, size)
    return torch.rand((batch, channels, size[0], size[1]), generator=g)


def synth_predictions(B, A, nc, seed=0, frac=0.02, img=640.0):
    """Head-shaped predictions [B,A,5+nc] for NMS tests: xywh boxes clustered around a few
    centres (so NMS has overlaps to suppress), obj=1 like the real head, ~`frac` of class scores
    above 0.03.  Scores are distinct fp32 values (tie-free by construction)."""
    r = np.random.RandomState(seed + 17)
    ncent = max(4, A // 40)
    cx = r.uniform(0.1, 0.9, size=(B, ncent)) * img
    cy = r.uniform(0.1, 0.9, size=(B, ncent)) * img
    w0 = r.uniform(20, 200, size=(B, ncent))
    h0 = r.uniform(20, 200, size=(B, ncent))
    which = r.randint(0, ncent, size=(B, A))
    bi = np.arange(B)[:, None]
    x = cx[bi, which] + r.normal(0, 6, size=(B, A))
    y = cy[bi, which] + r.normal(0, 6, size=(B, A))
    w = w0[bi, which] * r.uniform(0.8, 1.25, size=(B, A))
    h = h0[bi, which] * r.uniform(0.8, 1.25, size=(B, A))
    scores = r.uniform(0, 0.03, size=(B, A, nc))
    hot = r.uniform(size=(B, A, nc)) < frac
    scores[hot] = r.uniform(0.03, 0.95, size=int(hot.sum()))
    pred = np.concatenate([x[..., None], y[..., None], w[..., None], h[..., None], np.ones((B, A, 1)), scores], -1)
    return torch.from_numpy(pred.astype(np.float32))


def synth_tal_inputs(B, feat_sizes, strides, C, G, seed=0, n_valid=None, img=None):
    """Assigner inputs shaped like ComputeLoss builds them (reference models/losses/loss.py:52-103):
    anchor points in pixels, predicted boxes centred near their anchors (so an anchor inside a gt
    has IoU > 0 with it), gt boxes large enough to hold >= topk anchors, padded with zeros."""
    r = np.random.RandomState(seed + 101)
    pts = []
    for (h, w), s in zip(feat_sizes, strides):
        gy, gx = np.meshgrid((np.arange(h) + 0.5) * s, (np.arange(w) + 0.5) * s, indexing="ij")
        pts.append(np.stack([gx, gy], -1).reshape(-1, 2))
    pts = np.concatenate(pts).astype(np.float32)
    A = pts.shape[0]
    if img is None:
        img = feat_sizes[0][0] * strides[0]
    half = r.uniform(8, 0.35 * img, size=(B, A, 2))
    ctr = pts[None] + r.normal(0, 2.0, size=(B, A, 2))
    pd_bboxes = np.concatenate([ctr - half, ctr + half], -1).astype(np.float32)
    pd_scores = r.uniform(0.01, 0.99, size=(B, A, C)).astype(np.float32)
    if n_valid is None:
        n_valid = [G] * B
    gt_bboxes = np.zeros((B, G, 4), np.float32)
    gt_labels = np.zeros((B, G, 1), np.float32)
    mask_gt = np.zeros((B, G, 1), np.float32)
    for b in range(B):
        for g in range(n_valid[b]):
            wh = r.uniform(0.25, 0.7, size=2) * img
            c = r.uniform(0.3, 0.7, size=2) * img
            x1y1 = np.clip(c - wh / 2, 0, img)
            x2y2 = np.clip(c + wh / 2, 0, img)
            gt_bboxes[b, g] = np.concatenate([x1y1, x2y2])
            gt_labels[b, g, 0] = r.randint(0, C)
            mask_gt[b, g, 0] = 1.0
    t = torch.from_numpy
    return dict(pd_scores=t(pd_scores), pd_bboxes=t(pd_bboxes), anc_points=t(pts), gt_labels=t(gt_labels),
                gt_bboxes=t(gt_bboxes), mask_gt=t(mask_gt))


def synth_loss_inputs(B, feat_sizes, strides, C, reg_max, use_dfl, seed=0, boxes_per_image=(1, 6), img=None):
    """Inputs of ComputeLoss.__call__ (reference models/losses/loss.py:52-60): the train-branch head outputs
    pred_scores [B,A,C] (post-sigmoid) and pred_distri [B,A,4*(reg_max+1)] (raw DFL logits; [B,A,4] distances in
    stride units without DFL), and targets [N,6] = (image, class, cx, cy, w, h in 0..1).  The distances are drawn so
    that a decoded box is a plausible box around its anchor; a few images get no target."""
    r = np.random.RandomState(seed + 211)
    A = sum(h * w for h, w in feat_sizes)
    if img is None:
        img = feat_sizes[0][0] * strides[0]
    pred_scores = r.uniform(0.005, 0.95, size=(B, A, C)).astype(np.float32)
    if use_dfl:
        pred_distri = r.normal(0.0, 1.5, size=(B, A, 4 * (reg_max + 1))).astype(np.float32)
        peak = r.randint(1, reg_max, size=(B, A, 4))
        bi, ai, si = np.meshgrid(np.arange(B), np.arange(A), np.arange(4), indexing="ij")
        pred_distri[bi, ai, si * (reg_max + 1) + peak] += 4.0
    else:
        pred_distri = r.uniform(0.5, 6.0, size=(B, A, 4)).astype(np.float32)
    rows = []
    for b in range(B):
        n = 0 if (B > 2 and b == B - 1) else r.randint(boxes_per_image[0], boxes_per_image[1] + 1)
        for _ in range(n):
            wh = r.uniform(0.2, 0.6, size=2)
            c = r.uniform(0.3, 0.7, size=2)
            rows.append([b, r.randint(0, C), c[0], c[1], wh[0], wh[1]])
    targets = np.asarray(rows, np.float32).reshape(-1, 6)
    t = torch.from_numpy
    return dict(pred_scores=t(pred_scores), pred_distri=t(pred_distri), targets=t(targets), img=int(img))


def synth_loss_inputs_ab(B, feat_sizes, strides, C, seed=0, boxes_per_image=(1, 6), img=None):
    """Inputs of the anchor-based ComputeLoss (reference models/losses/loss_fuseab.py:43-51): pred_scores [B,3A,C] and
    pred_distri [B,3A,4] = (dx, dy, w, h) in stride units around the anchor point (anchors ordered level, anchor, pixel),
    targets as in synth_loss_inputs."""
    base = synth_loss_inputs(B, feat_sizes, strides, C, 0, False, seed=seed, boxes_per_image=boxes_per_image, img=img)
    r = np.random.RandomState(seed + 977)
    A = 3 * sum(h * w for h, w in feat_sizes)
    pred_scores = r.uniform(0.005, 0.95, size=(B, A, C)).astype(np.float32)
    dxy = r.uniform(-1.0, 1.0, size=(B, A, 2))
    wh = r.uniform(1.0, 7.0, size=(B, A, 2))
    pred_distri = np.concatenate([dxy, wh], -1).astype(np.float32)
    return dict(pred_scores=torch.from_numpy(pred_scores), pred_distri=torch.from_numpy(pred_distri), targets=base["targets"],
                img=base["img"])
