"""ComputeLoss forward value on the GPU (y6_bbox_decode + HIP assigner + y6_loss_forward through
yolov6_amd/models/losses/loss.py) against the reference-generated goldens and the CPU oracle."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import loss_oracle, synth
from tests.helpers import GOLDEN

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
CASES = sorted(f[len("loss_"):-len(".npz")] for f in os.listdir(GOLDEN) if f.startswith("loss_"))


def _run(m, inp, targets):
    from yolov6_amd.models.losses.loss import ComputeLoss
    crit = ComputeLoss(fpn_strides=m["strides"], num_classes=m["C"], ori_img_size=inp["img"], warmup_epoch=4,
                       use_dfl=m["use_dfl"], reg_max=m["reg_max"], iou_type=m["iou_type"])
    feats = [torch.zeros(m["B"], 1, h, w, device=DEV) for h, w in m["feat_sizes"]]
    loss, items = crit((feats, inp["pred_scores"].to(DEV), inp["pred_distri"].to(DEV)), targets.to(DEV), m["epoch"], 1,
                       inp["img"], inp["img"])
    torch.cuda.synchronize()
    return float(loss), items.cpu().numpy().astype(np.float64)


@pytest.mark.parametrize("case", CASES)
def test_loss_matches_reference_golden(case):
    g = np.load(os.path.join(GOLDEN, f"loss_{case}.npz"))
    m = json.loads(str(g["meta"]))
    inp = synth.synth_loss_inputs(m["B"], m["feat_sizes"], m["strides"], m["C"], m["reg_max"], m["use_dfl"], seed=m["seed"])
    targets = inp["targets"] if case != "no_targets" else inp["targets"][:0]
    loss, items = _run(m, inp, targets)
    # 1e-4: fp32 terms summed in a different order than torch's reduction, targets carried in fp32 (reference: fp64)
    np.testing.assert_allclose(loss, float(g["loss"]), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(items, g["items"], rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("iou_type,use_dfl,epoch", [("giou", True, 10), ("siou", False, 10), ("diou", True, 0)])
def test_loss_training_size_vs_oracle(iou_type, use_dfl, epoch):
    """B=4 x 8400 anchors x 80 classes (the S training shape at a quarter batch) against the CPU oracle."""
    m = dict(B=4, feat_sizes=[(80, 80), (40, 40), (20, 20)], strides=[8, 16, 32], C=80, reg_max=16, use_dfl=use_dfl,
             iou_type=iou_type, epoch=epoch)
    inp = synth.synth_loss_inputs(m["B"], m["feat_sizes"], m["strides"], m["C"], m["reg_max"], use_dfl, seed=7,
                                  boxes_per_image=(3, 12))
    loss, items = _run(m, inp, inp["targets"])
    ref = loss_oracle.compute_loss(m["feat_sizes"], inp["pred_scores"].numpy(), inp["pred_distri"].numpy(),
                                   inp["targets"].numpy(), epoch, inp["img"], inp["img"], fpn_strides=m["strides"],
                                   num_classes=m["C"], warmup_epoch=4, use_dfl=use_dfl, reg_max=m["reg_max"],
                                   iou_type=iou_type)
    np.testing.assert_allclose(loss, ref["loss"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(items, ref["loss_items"], rtol=1e-4, atol=1e-5)


def test_bbox_decode_matches_oracle():
    from yolov6_amd.models.losses.loss import ComputeLoss
    inp = synth.synth_loss_inputs(2, [(16, 16), (8, 8), (4, 4)], [8, 16, 32], 20, 16, True, seed=9)
    _, pts, _, st = loss_oracle.generate_anchors([(16, 16), (8, 8), (4, 4)], [8, 16, 32])
    pts_s = pts / st
    crit = ComputeLoss(num_classes=20)
    out = crit.bbox_decode(torch.from_numpy(pts_s).to(DEV), inp["pred_distri"].to(DEV)).cpu().numpy()
    ref = loss_oracle.bbox_decode(pts_s, inp["pred_distri"].numpy(), True, 16)
    np.testing.assert_allclose(out, ref, rtol=1e-5, atol=1e-5)


# ---------------------------------------------------------------- self-distillation losses
DISTILL_CASES = sorted(f[len("lossdistill_"):-len(".npz")] for f in os.listdir(GOLDEN) if f.startswith("lossdistill_"))
DISTILL_NS_CASES = sorted(f[len("lossdistillns_"):-len(".npz")] for f in os.listdir(GOLDEN) if f.startswith("lossdistillns_"))


def _distill_inputs(m, ns):
    """The inputs tests/golden/gen_golden.py fed the reference (student: seed, teacher: seed + 50, feature maps / plain distances
    from a torch generator seeded 1000 / 2000 + seed)."""
    B, fs, st, C, reg_max, seed = m["B"], [tuple(v) for v in m["feat_sizes"]], m["strides"], m["C"], m["reg_max"], m["seed"]
    inp = synth.synth_loss_inputs(B, fs, st, C, reg_max, True, seed=seed)
    tea = synth.synth_loss_inputs(B, fs, st, C, reg_max, True, seed=seed + 50)
    g = torch.Generator().manual_seed((2000 if ns else 1000) + seed)
    lrtb = None
    if ns:
        A = inp["pred_scores"].shape[1]
        lrtb = torch.rand((B, A, 4), generator=g) * 3.0 + 0.2
    chans = m["feat_channels"]
    s_feats = [torch.randn((B, c, h, w), generator=g) for c, (h, w) in zip(chans, fs)]
    t_feats = [torch.randn((B, c, h, w), generator=g) for c, (h, w) in zip(chans, fs)]
    return inp, tea, lrtb, s_feats, t_feats


def _check_distill(gold, m, ns, case):
    if ns:
        from yolov6_amd.models.losses.loss_distill_ns import ComputeLoss
    else:
        from yolov6_amd.models.losses.loss_distill import ComputeLoss
    inp, tea, lrtb, s_feats, t_feats = _distill_inputs(m, ns)
    targets = inp["targets"] if case != "no_targets" else inp["targets"][:0]
    crit = ComputeLoss(fpn_strides=m["strides"], num_classes=m["C"], ori_img_size=inp["img"], warmup_epoch=m["warmup_epoch"], use_dfl=True,
                       reg_max=m["reg_max"], iou_type=m["iou_type"], distill_feat=m["distill_feat"])
    feats = [torch.zeros(m["B"], 1, h, w, device=DEV) for h, w in m["feat_sizes"]]
    ps = inp["pred_scores"].to(DEV).requires_grad_(True)
    pd = inp["pred_distri"].to(DEV).requires_grad_(True)
    sf = [f.to(DEV).requires_grad_(True) for f in s_feats]
    tf = [f.to(DEV) for f in t_feats]
    outs = (feats, ps, pd)
    pl = None
    if ns:
        assert np.array_equal(lrtb.numpy(), gold["lrtb"]), "the test's inputs are not the golden's"
        pl = lrtb.to(DEV).requires_grad_(True)
        outs = (feats, ps, pd, pl)
    loss, items = crit(outs, (feats, tea["pred_scores"].to(DEV), tea["pred_distri"].to(DEV)), sf, tf, targets.to(DEV), m["epoch"],
                       m["max_epoch"], m["temperature"], 1, inp["img"], inp["img"])
    S = 4.0                                      # an incoming gradient (a loss scale) is applied inside the kernels
    (loss * S).backward()
    torch.cuda.synchronize()
    np.testing.assert_allclose(float(loss), float(gold["loss"]), rtol=2e-4, atol=1e-5)
    np.testing.assert_allclose(items.cpu().numpy().astype(np.float64), gold["items"], rtol=2e-4, atol=1e-5)
    checks = [(ps.grad, "dscores"), (pd.grad, "ddistri")] + ([(pl.grad, "dlrtb")] if ns else [])
    if m["distill_feat"] and not ns:
        checks += [(sf[i].grad, f"dfeat{i}") for i in range(3)]
    for got, name in checks:
        ref = gold[name].astype(np.float64) * S
        scale = max(float(np.abs(ref).max()), 1e-12)
        err = float(np.abs(got.cpu().numpy().astype(np.float64) - ref).max()) / scale
        assert err < 2e-4, f"{case}: {name} deviates by {err:.3e} of its max"


@pytest.mark.parametrize("case", DISTILL_CASES)
def test_distill_loss_and_gradients_vs_reference(case):
    """loss_distill.py's ComputeLoss on the HIP path: value, loss items and every gradient the unmodified reference back-propagates
    (student class scores, DFL logits, and - distill_feat cases - the three feature maps)."""
    g = np.load(os.path.join(GOLDEN, f"lossdistill_{case}.npz"))
    _check_distill(g, json.loads(str(g["meta"])), False, case)


@pytest.mark.parametrize("case", DISTILL_NS_CASES)
def test_distill_ns_loss_and_gradients_vs_reference(case):
    """loss_distill_ns.py (fourth student output: plain distances; their IoU loss is added; no ATSS warm-up)."""
    g = np.load(os.path.join(GOLDEN, f"lossdistillns_{case}.npz"))
    _check_distill(g, json.loads(str(g["meta"])), True, case)
