"""ORACLE (test infrastructure - never imported by the product path).

numpy restatement of `non_max_suppression` (reference yolov6/utils/nms.py:31-105) and of
the `torchvision.ops.nms` it calls at :96.

torchvision is a third-party dependency of the reference (requirements.txt:5
`torchvision>=0.9.0`, un-pinned, not vendored, not installed here), so `nms()` below
restates its published algorithm: stable sort by score descending; greedy sweep; box j is
suppressed by a kept box i when  inter / (area_i + area_j - inter) > iou_threshold  with
inter = max(0, xx2-xx1) * max(0, yy2-yy1), all in fp32, no "+1", no epsilon; kept indices are
returned in descending-score order.  The same greedy / strict-">" form appears in-tree in
deploy/TensorRT/yolov6.cpp:122-155.  PARITY UNPINNED for this one call: the reference holds
no test vectors for it; everything around the call is pinned by running the
reference's own nms.py with this function injected (tests/golden/gen_golden.py).
CORROBORATED since round 4: oracle/build_ref.py compiles that in-tree C++ statement
(`nms_sorted_bboxes` + `intersection_area`, cut out of the reference file where it lies) into
oracle/_ref/libref_nms.so, and tests/test_oracle_ref.py checks that it and `nms()` keep the same
boxes on random, clustered and exactly-at-threshold inputs - a second implementation of the
rule written by the reference's authors, not the torchvision kernel itself.

Tie rule (upstream leaves it unspecified): equal scores keep their original order.
"""
import numpy as np

f32 = np.float32


def nms(boxes, scores, iou_threshold, max_keep=None):
    """boxes [n,4] fp32 xyxy, scores [n] fp32 -> kept indices (int64) in descending-score order.
    max_keep: stop after that many kept boxes - the greedy sweep never looks back, so the first max_keep entries are the
    same as those of the full result (the caller slices `[:max_det]`, nms.py:97-98); it only saves the oracle's time."""
    boxes = np.asarray(boxes, dtype=f32)
    scores = np.asarray(scores, dtype=f32)
    n = boxes.shape[0]
    if n == 0:
        return np.zeros((0,), np.int64)
    order = np.argsort(-scores, kind="stable")
    b = boxes[order]
    x1, y1, x2, y2 = b[:, 0], b[:, 1], b[:, 2], b[:, 3]
    areas = (x2 - x1) * (y2 - y1)
    thr = f32(iou_threshold)
    suppressed = np.zeros(n, bool)
    keep = []
    for i in range(n):
        if suppressed[i]:
            continue
        keep.append(order[i])
        if i + 1 == n or (max_keep is not None and len(keep) >= max_keep):
            break
        xx1 = np.maximum(x1[i], x1[i + 1:])
        yy1 = np.maximum(y1[i], y1[i + 1:])
        xx2 = np.minimum(x2[i], x2[i + 1:])
        yy2 = np.minimum(y2[i], y2[i + 1:])
        w = np.maximum(f32(0), xx2 - xx1)
        h = np.maximum(f32(0), yy2 - yy1)
        inter = w * h
        with np.errstate(divide="ignore", invalid="ignore"):
            ovr = inter / (areas[i] + areas[i + 1:] - inter)
        suppressed[i + 1:] |= ovr > thr
    return np.asarray(keep, np.int64)


def xywh2xyxy(x):
    """nms.py:21-28"""
    y = np.copy(x)
    y[:, 0] = x[:, 0] - x[:, 2] / f32(2)
    y[:, 1] = x[:, 1] - x[:, 3] / f32(2)
    y[:, 2] = x[:, 0] + x[:, 2] / f32(2)
    y[:, 3] = x[:, 1] + x[:, 3] / f32(2)
    return y


def non_max_suppression(prediction, conf_thres=0.25, iou_thres=0.45, classes=None, agnostic=False,
                        multi_label=False, max_det=300, return_index=False):
    """prediction [B,A,5+nc] fp32 -> list of [n,6] arrays (xyxy, conf, cls).  nms.py:31-105.
    With return_index, also the flat candidate ids (anchor*nc + cls) of the kept rows."""
    prediction = np.asarray(prediction, dtype=f32)
    nc = prediction.shape[2] - 5
    conf = f32(conf_thres)
    cand = np.logical_and(prediction[..., 4] > conf, prediction[..., 5:].max(-1) > conf)  # :48
    assert 0 <= conf_thres <= 1 and 0 <= iou_thres <= 1                                   # :50-51
    max_wh, max_nms = f32(4096), 30000                                                    # :54-55
    multi_label = multi_label and nc > 1                                                  # :57
    out, out_idx = [], []
    for img in range(prediction.shape[0]):                                                # :61
        anchors = np.nonzero(cand[img])[0]
        x = prediction[img][anchors].copy()
        if not x.shape[0]:
            out.append(np.zeros((0, 6), f32)); out_idx.append(np.zeros((0,), np.int64)); continue
        x[:, 5:] *= x[:, 4:5]                                                             # :69
        box = xywh2xyxy(x[:, :4])                                                         # :72
        if multi_label:                                                                   # :75-77
            bi, ci = np.nonzero(x[:, 5:] > conf)
            det = np.concatenate([box[bi], x[bi, ci + 5, None], ci[:, None].astype(f32)], 1)
            flat = anchors[bi] * nc + ci
        else:                                                                             # :79-80
            ci = x[:, 5:].argmax(1)
            cf = x[np.arange(x.shape[0]), 5 + ci]
            det = np.concatenate([box, cf[:, None], ci[:, None].astype(f32)], 1)
            sel = cf > conf
            det, flat = det[sel], (anchors * nc + ci)[sel]
        if classes is not None:                                                           # :83-84
            sel = np.isin(det[:, 5], np.asarray(classes, f32))
            det, flat = det[sel], flat[sel]
        n = det.shape[0]
        if not n:                                                                         # :88-89
            out.append(np.zeros((0, 6), f32)); out_idx.append(np.zeros((0,), np.int64)); continue
        if n > max_nms:                                                                   # :90-91
            o = np.argsort(-det[:, 4], kind="stable")[:max_nms]
            det, flat = det[o], flat[o]
        off = det[:, 5:6] * (f32(0) if agnostic else max_wh)                              # :94
        keep = nms(det[:, :4] + off, det[:, 4], iou_thres, max_keep=max_det)              # :95-96
        keep = keep[:max_det]                                                             # :97-98
        out.append(det[keep].astype(f32))
        out_idx.append(flat[keep].astype(np.int64))
    return (out, out_idx) if return_index else out
