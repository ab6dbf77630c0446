"""oracle/_ref (the reference's own C++ greedy NMS, deploy/TensorRT/yolov6.cpp:122-155, compiled by oracle/build_ref.py) against
oracle/nms_oracle.nms - the restatement of torchvision.ops.nms that non_max_suppression's parity rests on.  Two independent
statements of "descending score, keep a box unless a kept one overlaps it by more than the threshold" must pick the same boxes."""
import ctypes as C
import os

import numpy as np
import pytest

from oracle import build_ref, nms_oracle


def _lib():
    path = build_ref.build(verbose=False)
    if path is None or not os.path.exists(path):
        pytest.skip("oracle/_ref is not built (no reference checkout and no prebuilt library)")
    lib = C.CDLL(path)
    lib.ref_nms_sorted_bboxes.restype = C.c_int
    lib.ref_nms_sorted_bboxes.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_void_p]
    return lib


def _ref_keep(lib, xyxy, scores, thr):
    order = np.argsort(-scores, kind="stable")
    b = xyxy[order]
    xywh = np.ascontiguousarray(np.stack([b[:, 0], b[:, 1], b[:, 2] - b[:, 0], b[:, 3] - b[:, 1]], 1).astype(np.float32))
    out = np.zeros(len(b), np.int32)
    n = lib.ref_nms_sorted_bboxes(xywh.ctypes.data, len(b), C.c_float(thr), out.ctypes.data)
    return order[out[:n]]


@pytest.mark.parametrize("seed,n,thr", [(0, 300, 0.45), (1, 1200, 0.65), (2, 64, 0.3), (3, 2000, 0.5), (4, 1, 0.5)])
def test_reference_cpp_nms_equals_the_oracle(seed, n, thr):
    """Boxes on a quarter-pixel grid (corners, widths and areas exact in fp32: x + width == x2 holds, so the two box
    representations - cv::Rect x, y, w, h vs torchvision x1, y1, x2, y2 - describe the same numbers), distinct scores."""
    lib = _lib()
    g = np.random.default_rng(seed)
    x1 = g.integers(0, 2400, n) / 4.0
    y1 = g.integers(0, 2400, n) / 4.0
    w = g.integers(4, 800, n) / 4.0
    h = g.integers(4, 800, n) / 4.0
    xyxy = np.stack([x1, y1, x1 + w, y1 + h], 1).astype(np.float32)
    scores = g.permutation(n).astype(np.float32) / n        # tie-free: the two sorts cannot disagree
    keep_ref = _ref_keep(lib, xyxy, scores, thr)
    keep_orc = nms_oracle.nms(xyxy, scores, thr)
    assert np.array_equal(keep_ref.astype(np.int64), keep_orc.astype(np.int64))
    assert 0 < len(keep_orc) <= n


def test_reference_cpp_nms_equals_the_oracle_on_clustered_boxes():
    """Heavy overlap (many suppressions per kept box) and the boundary rule: IoU EXACTLY at the threshold is kept by both
    (strict `>`, yolov6.cpp:147 / torchvision)."""
    lib = _lib()
    g = np.random.default_rng(7)
    centers = g.integers(100, 500, (12, 2))
    rows = []
    for c in centers:
        for _ in range(40):
            d = g.integers(-12, 13, 2)
            s = g.integers(40, 72, 2)
            rows.append([c[0] + d[0], c[1] + d[1], c[0] + d[0] + s[0], c[1] + d[1] + s[1]])
    xyxy = np.asarray(rows, np.float32)
    scores = g.permutation(len(rows)).astype(np.float32)
    for thr in (0.3, 0.5, 0.7):
        assert np.array_equal(_ref_keep(lib, xyxy, scores, thr).astype(np.int64), nms_oracle.nms(xyxy, scores, thr).astype(np.int64))
    # two 4x4 boxes sharing a 2x4 strip: inter 8, union 24, IoU = 1/3 exactly representable? no - use inter 8, union 16: 0.5
    a = np.asarray([[0, 0, 4, 4], [0, 2, 4, 4 + 2 - 0]], np.float32)     # second: y 2..6 -> inter 4x2 = 8, union 16 + 16 - 8 = 24
    b = np.asarray([[0, 0, 4, 4], [2, 0, 6, 4]], np.float32)             # inter 2x4 = 8, union 24 -> 1/3 (rounded the same way in both)
    c = np.asarray([[0, 0, 4, 2], [0, 0, 4, 4]], np.float32)             # inter 8, union 8 + 16 - 8 = 16 -> exactly 0.5
    s = np.asarray([2.0, 1.0], np.float32)
    for boxes, thr in ((a, 1.0 / 3.0), (b, 1.0 / 3.0), (c, 0.5)):
        assert np.array_equal(_ref_keep(lib, boxes, s, np.float32(thr)).astype(np.int64), nms_oracle.nms(boxes, s, np.float32(thr)).astype(np.int64))
    assert len(nms_oracle.nms(c, s, 0.5)) == 2 and len(nms_oracle.nms(c, s, 0.4999)) == 1
