"""GPU: NMS and TAL through the C ABI - bit-exact kept indices / assignments against the oracle
and the reference-generated goldens; float payloads within 1e-6."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import nms_oracle, synth, tal_oracle
from tests.helpers import GOLDEN
from yolov6_amd.assigners import TaskAlignedAssigner
from yolov6_amd.utils.nms import nms_raw, non_max_suppression

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
NMS_CASES = ["eval_multilabel", "infer_single", "agnostic_classes", "max_det_cut", "over_max_nms", "empty"]


@pytest.mark.parametrize("case", NMS_CASES)
def test_nms_matches_reference_golden(case):
    g = np.load(os.path.join(GOLDEN, f"nms_{case}.npz"))
    meta = json.loads(str(g["meta"]))
    pred = synth.synth_predictions(meta["B"], meta["A"], meta["nc"], seed=meta["seed"], frac=meta["frac"])
    res = non_max_suppression(pred.to(DEV), **meta["kwargs"])
    counts = [int(r.shape[0]) for r in res]
    assert counts == g["counts"].tolist()
    dets = np.concatenate([r.cpu().numpy().reshape(-1, 6) for r in res], 0)
    assert np.array_equal(dets, g["dets"])            # same boxes, scores, classes, same order, bit for bit


def _check_vs_oracle(pred, **kw):
    dets, index, count = nms_raw(pred.to(DEV), **kw)
    torch.cuda.synchronize()
    exp, exp_idx = nms_oracle.non_max_suppression(pred.numpy(), return_index=True, **kw)
    for b in range(pred.shape[0]):
        n = int(count[b])
        assert n == exp[b].shape[0], (b, n, exp[b].shape[0])
        assert np.array_equal(index[b, :n].cpu().numpy().astype(np.int64), exp_idx[b]), f"image {b}: kept indices differ"
        assert np.array_equal(dets[b, :n].cpu().numpy(), exp[b])


def test_nms_full_size_batch():
    """BASELINE shape: 32 images x 8400 anchors x 80 classes, eval thresholds."""
    pred = synth.synth_predictions(32, 8400, 80, seed=11, frac=0.004)
    _check_vs_oracle(pred, conf_thres=0.03, iou_thres=0.65, multi_label=True, max_det=300)


def test_nms_edge_cases():
    # (a) all boxes identical -> exactly one survivor per class
    p = synth.synth_predictions(1, 64, 4, seed=1, frac=0.5)
    p[..., 0:4] = torch.tensor([100.0, 100.0, 50.0, 50.0])
    _check_vs_oracle(p, conf_thres=0.03, iou_thres=0.5, multi_label=True, max_det=300)
    # (b) objectness below threshold everywhere -> empty
    q = synth.synth_predictions(2, 64, 4, seed=2, frac=0.5)
    q[..., 4] = 0.01
    assert all(r.shape == (0, 6) for r in non_max_suppression(q.to(DEV), 0.25, 0.45))
    # (c) single class: multi_label is ignored (nms.py:57), class filter, agnostic
    r = synth.synth_predictions(2, 300, 1, seed=3, frac=0.5)
    _check_vs_oracle(r, conf_thres=0.1, iou_thres=0.45, multi_label=True, max_det=10)
    s = synth.synth_predictions(2, 300, 6, seed=4, frac=0.3)
    _check_vs_oracle(s, conf_thres=0.1, iou_thres=0.45, classes=[0, 5], agnostic=True, max_det=300)
    # (d) more than 16384 candidates (global-memory sort path) and more than max_nms
    t = synth.synth_predictions(2, 8400, 80, seed=5, frac=0.05)
    _check_vs_oracle(t, conf_thres=0.03, iou_thres=0.65, multi_label=True, max_det=300)
    # (f) objectness above 1 (not a sigmoid's): cls <= conf < cls * obj is then possible, and the row flag of nms.py:48 (max cls > conf)
    #     is no longer implied by the candidate - the selection's fast form must hand such blocks to the general one
    u = synth.synth_predictions(3, 700, 12, seed=6, frac=0.2)
    u[..., 4] = u[..., 4] * 1.8
    u[..., 5:] = u[..., 5:] * 0.7
    _check_vs_oracle(u, conf_thres=0.2, iou_thres=0.5, multi_label=True, max_det=300)
    _check_vs_oracle(u, conf_thres=0.2, iou_thres=0.5, multi_label=False, max_det=300)
    # (e) threshold asserts are the reference's (nms.py:50-51)
    with pytest.raises(AssertionError):
        non_max_suppression(s.to(DEV), conf_thres=1.5)


def test_nms_large_max_det():
    """max_det beyond the former 2048 limit (kept boxes stay in LDS: up to 8192): many survivors with a permissive IoU
    threshold; 8193 is refused loudly."""
    p = synth.synth_predictions(2, 8400, 20, seed=9, frac=0.05)
    _check_vs_oracle(p, conf_thres=0.03, iou_thres=0.9, multi_label=True, max_det=5000)
    _check_vs_oracle(p, conf_thres=0.03, iou_thres=0.95, multi_label=True, max_det=8192)
    with pytest.raises(RuntimeError):
        non_max_suppression(p.to(DEV), 0.03, 0.9, multi_label=True, max_det=8193)


def test_nms_idempotent_and_sorted():
    """Size-independent properties at the full size: output sorted by confidence; re-running NMS on the
    survivors (as a prediction tensor) keeps all of them."""
    pred = synth.synth_predictions(4, 8400, 80, seed=21, frac=0.01)
    res = non_max_suppression(pred.to(DEV), 0.03, 0.65, multi_label=True, max_det=300)
    for r in res:
        c = r[:, 4].cpu().numpy()
        assert np.all(c[:-1] >= c[1:])
        n = r.shape[0]
        again = torch.zeros((1, n, 85))
        xyxy = r[:, :4].cpu()
        again[0, :, 0:2] = (xyxy[:, :2] + xyxy[:, 2:]) / 2
        again[0, :, 2:4] = xyxy[:, 2:] - xyxy[:, :2]
        again[0, :, 4] = 1.0
        again[0, torch.arange(n), 5 + r[:, 5].long().cpu()] = r[:, 4].cpu()
        r2 = non_max_suppression(again.to(DEV), 0.03, 0.65, multi_label=True, max_det=300)[0]
        assert r2.shape[0] == n


TAL_CASES = ["basic", "padded", "many_gt", "topk26", "empty"]


def _run_tal(inp, topk, C):
    a = TaskAlignedAssigner(topk=topk, num_classes=C, alpha=1.0, beta=6.0)
    out = a(*(inp[k].to(DEV) for k in ("pd_scores", "pd_bboxes", "anc_points", "gt_labels", "gt_bboxes", "mask_gt")))
    torch.cuda.synchronize()
    return [o.cpu().numpy() for o in out]


@pytest.mark.parametrize("case", TAL_CASES)
def test_tal_matches_reference_golden(case):
    g = np.load(os.path.join(GOLDEN, f"tal_{case}.npz"))
    meta = json.loads(str(g["meta"]))
    inp = synth.synth_tal_inputs(meta["B"], [tuple(f) for f in meta["feat_sizes"]], meta["strides"], meta["C"],
                                 meta["G"], seed=meta["seed"], n_valid=meta["n_valid"])
    L, Bx, S, F = _run_tal(inp, meta["topk"], meta["C"])
    assert np.array_equal(F.astype(bool), g["fg"])                 # bit-exact assignment
    assert np.array_equal(L.astype(np.int64), g["labels"])
    assert np.array_equal(Bx, g["bboxes"])
    ref = np.zeros_like(S)
    idx = g["score_idx"]
    ref[idx[:, 0], idx[:, 1], idx[:, 2]] = g["score_val"]
    assert np.array_equal(S != 0, ref != 0)
    np.testing.assert_allclose(S, ref, rtol=2e-5, atol=1e-10)


def test_tal_training_size_vs_oracle():
    """A training-sized problem (8 images x 8400 anchors x 80 classes, 40 gts, ragged validity)."""
    fs, st = [(80, 80), (40, 40), (20, 20)], [8, 16, 32]
    nv = [40, 0, 17, 40, 3, 25, 1, 33]
    inp = synth.synth_tal_inputs(8, fs, st, 80, 40, seed=9, n_valid=nv, img=640)
    L, Bx, S, F = _run_tal(inp, 13, 80)
    eL, eB, eS, eF = tal_oracle.assign(*(inp[k].numpy() for k in ("pd_scores", "pd_bboxes", "anc_points", "gt_labels",
                                                                   "gt_bboxes", "mask_gt")), topk=13, num_classes=80)
    assert np.array_equal(F.astype(bool), eF)
    assert np.array_equal(L.astype(np.int64), eL)
    assert np.array_equal(Bx, eB)
    np.testing.assert_allclose(S, eS, rtol=2e-5, atol=1e-10)
    # properties: every fg anchor has exactly one non-zero class score; background rows are all zero
    nz = (S != 0).sum(-1)
    assert np.all(nz[F.astype(bool)] <= 1) and np.all(nz[~F.astype(bool)] == 0)


ATSS_CASES = ["basic", "padded_nopd", "p6", "empty"]


@pytest.mark.parametrize("case", ATSS_CASES)
def test_atss_matches_reference_golden(case):
    from yolov6_amd.assigners import ATSSAssigner
    g = np.load(os.path.join(GOLDEN, f"atss_{case}.npz"))
    meta = json.loads(str(g["meta"]))
    inp = synth.synth_tal_inputs(meta["B"], [tuple(f) for f in meta["feat_sizes"]], meta["strides"], meta["C"],
                                 meta["G"], seed=meta["seed"], n_valid=meta["n_valid"])
    a = ATSSAssigner(9, num_classes=meta["C"])
    out = a(torch.from_numpy(g["anchors"]).to(DEV), g["n_list"].tolist(), inp["gt_labels"].to(DEV),
            inp["gt_bboxes"].to(DEV), inp["mask_gt"].to(DEV), inp["pd_bboxes"].to(DEV) if meta["with_pd"] else None)
    torch.cuda.synchronize()
    L, Bx, S, F = [o.cpu().numpy() for o in out]
    assert np.array_equal(F.astype(bool), g["fg"])                 # bit-exact assignment
    assert np.array_equal(L.astype(np.int64), g["labels"])
    assert np.array_equal(Bx, g["bboxes"])
    ref = np.zeros_like(S)
    idx = g["score_idx"]
    ref[idx[:, 0], idx[:, 1], idx[:, 2]] = g["score_val"]
    assert np.array_equal(S != 0, ref != 0)
    np.testing.assert_allclose(S, ref, rtol=2e-5, atol=1e-10)


def test_atss_training_size_vs_oracle():
    from oracle import atss_oracle
    from yolov6_amd.assigners import ATSSAssigner, generate_anchors
    fs, st = [(80, 80), (40, 40), (20, 20)], [8, 16, 32]
    nv = [40, 0, 17, 40]
    inp = synth.synth_tal_inputs(4, fs, st, 80, 40, seed=19, n_valid=nv, img=640)
    feats = [torch.zeros(1, 1, h, w) for h, w in fs]
    anchors, _, n_list, _ = generate_anchors(feats, st, 5.0, 0.5, device="cpu", is_eval=False)
    out = ATSSAssigner(9, 80)(anchors.to(DEV), n_list, inp["gt_labels"].to(DEV), inp["gt_bboxes"].to(DEV),
                              inp["mask_gt"].to(DEV), inp["pd_bboxes"].to(DEV))
    torch.cuda.synchronize()
    L, Bx, S, F = [o.cpu().numpy() for o in out]
    eL, eB, eS, eF = atss_oracle.assign(anchors.numpy(), n_list, inp["gt_labels"].numpy(), inp["gt_bboxes"].numpy(),
                                        inp["mask_gt"].numpy(), inp["pd_bboxes"].numpy(), 9, 80)
    assert np.array_equal(F.astype(bool), eF) and np.array_equal(L.astype(np.int64), eL) and np.array_equal(Bx, eB)
    np.testing.assert_allclose(S, eS, rtol=2e-5, atol=1e-10)
