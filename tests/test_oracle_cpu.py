"""CPU: the oracle restatements against the golden vectors produced by the reference itself
(tests/golden/gen_golden.py).  These pin the oracle; the GPU tests then compare HIP vs oracle."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import nms_oracle, synth, tal_oracle
from oracle.model_oracle import Oracle, deploy_state_dict
from tests.helpers import GOLDEN, case_config, case_golden, rel_err, synth_sd_from_keys

MODEL_CASES = ["tiny", "n", "s", "s_qa_tiny", "l6_tiny", "m_tiny", "s_mbla_tiny", "n6", "m6_tiny", "t_pan", "s_csp_pan_tiny", "n6_pan", "n_base", "s_base_tiny", "s_qav1_tiny"]


@pytest.mark.parametrize("case", MODEL_CASES)
def test_model_oracle_matches_reference(case):
    cfg, meta = case_config(case)
    g = case_golden(case)
    sd = synth_sd_from_keys(meta["train"])
    x = synth.synth_images(meta["batch"], meta["size"], seed=1)
    nc = meta["num_classes"]
    with torch.no_grad():
        det_t, _ = Oracle(cfg, sd, nc).forward(x, train_form=True)
        det_d, feats = Oracle(cfg, sd, nc).forward(x, train_form=False)
    # fp32 CPU conv summation order differs between fused and unfused graphs: 1e-4 relative
    assert rel_err(det_t.numpy(), g["det_train"]) < 2e-4
    assert rel_err(det_d.numpy(), g["det_deploy"]) < 2e-4
    for i, f in enumerate(feats):
        assert rel_err(f.numpy(), g[f"feat{i}"]) < 2e-4


@pytest.mark.parametrize("case", MODEL_CASES)
def test_deploy_state_dict_keys_match_reference(case):
    cfg, meta = case_config(case)
    sd = synth_sd_from_keys(meta["train"])
    dep = deploy_state_dict(cfg, sd, meta["num_classes"])
    assert {k: list(v.shape) for k, v in dep.items()} == meta["deploy"]
    # and the deploy-form state dict evaluates to the same detections
    x = synth.synth_images(meta["batch"], meta["size"], seed=1)
    with torch.no_grad():
        det, _ = Oracle(cfg, dep, meta["num_classes"]).forward(x)
    assert rel_err(det.numpy(), case_golden(case)["det_deploy"]) < 2e-4


def test_fp16_emulation_stays_close():
    cfg, meta = case_config("tiny")
    sd = synth_sd_from_keys(meta["train"])
    x = synth.synth_images(meta["batch"], meta["size"], seed=1)
    with torch.no_grad():
        a, _ = Oracle(cfg, sd, 80).forward(x)
        b, _ = Oracle(cfg, sd, 80, emulate_fp16=True).forward(x)
    assert rel_err(b.numpy(), a.numpy()) < 2e-2


@pytest.mark.parametrize("case", MODEL_CASES)
def test_fp16_emulation_pinned_to_reference_half_goldens(case):
    """`Oracle(emulate_fp16=True)` - what every GPU model test compares with - against the REFERENCE'S OWN `model.half()` run
    on the CPU (tests/golden/model_<case>_half.npz, gen_golden.py::gen_models_half).  The emulation rounds to fp16 where the
    half model's op boundaries round (conv | BN | act | add | softmax | sigmoid) but accumulates each conv in fp32, as the
    GPU kernels do and as cuDNN / MIOpen do; torch's CPU half conv does not, so single elements differ by one fp16 ulp and -
    in the few deep, badly conditioned random-weight cases - those flips are amplified exactly like the half model's own
    deviation from fp32.  Bars: class scores within one fp16 ulp of a probability (2^-12), feature maps within two ulps at
    the map's top binade, boxes within one ulp of a pixel coordinate - or, where the reference's own fp16 noise on this
    input (fp32 oracle vs half golden) is larger than that, within 1.5x (scores) / 2x (boxes, maps) that noise."""
    cfg, meta = case_config(case)
    sd = synth_sd_from_keys(meta["train"])
    x = synth.synth_images(meta["batch"], meta["size"], seed=1).half().float()
    g = np.load(os.path.join(GOLDEN, f"model_{case}_half.npz"))
    with torch.no_grad():
        det, feats = Oracle(cfg, sd, meta["num_classes"], emulate_fp16=True).forward(x)
        det32, feats32 = Oracle(cfg, sd, meta["num_classes"], emulate_fp16=False).forward(x)
    half = g["det_half"]
    assert str(g["det_dtype"]) == "torch.float32"        # the reference's decode promotes to fp32 (anchor points / strides are fp32)
    e, fl = np.abs(det.numpy() - half), np.abs(det32.numpy() - half)
    ulp_p = 2.0 ** -12                                     # fp16 ulp of a sigmoid output in [0.25, 0.5); the largest scores sit there
    assert e[..., 5:].max() <= max(ulp_p * 1.001, 1.5 * fl[..., 5:].max()), (case, e[..., 5:].max(), fl[..., 5:].max())
    ulp_box = 2.0 ** -10 * 2.0 ** np.floor(np.log2(max(1.0, np.abs(half[..., :4]).max())))
    assert e[..., :4].max() <= max(ulp_box * 1.001, 2.0 * fl[..., :4].max()), (case, e[..., :4].max(), fl[..., :4].max())
    for i, (f, f32) in enumerate(zip(feats, feats32)):
        h = g[f"feat{i}_half"]
        top = max(1.0, float(np.abs(h).max()))
        ef, ff = float(np.abs(f.numpy() - h).max()) / top, float(np.abs(f32.numpy() - h).max()) / top
        assert ef <= max(2.0 * 2.0 ** -10, 2.0 * ff), (case, i, ef, ff)        # two ulps at the top binade: a flipped conv output re-rounded by the activation


NMS_CASES = ["eval_multilabel", "infer_single", "agnostic_classes", "max_det_cut", "over_max_nms", "empty"]


@pytest.mark.parametrize("case", NMS_CASES)
def test_nms_oracle_matches_reference(case):
    g = np.load(os.path.join(GOLDEN, f"nms_{case}.npz"))
    meta = json.loads(str(g["meta"]))
    pred = synth.synth_predictions(meta["B"], meta["A"], meta["nc"], seed=meta["seed"], frac=meta["frac"]).numpy()
    res = nms_oracle.non_max_suppression(pred, **meta["kwargs"])
    counts = np.array([r.shape[0] for r in res], np.int32)
    assert counts.tolist() == g["counts"].tolist()
    dets = np.concatenate([r.reshape(-1, 6) for r in res], 0)
    assert np.array_equal(dets, g["dets"])          # bit-exact, including order


def test_nms_known_answers():
    # two identical boxes: the later one is suppressed; IoU == thr is NOT suppressed (strict >)
    b = np.array([[0, 0, 10, 10], [0, 0, 10, 10], [0, 0, 10, 5]], np.float32)
    s = np.array([0.9, 0.8, 0.7], np.float32)
    assert nms_oracle.nms(b, s, 0.5).tolist() == [0, 2]         # duplicate 1 dies; IoU(0,2) = 0.5 is not > 0.5
    assert nms_oracle.nms(b[[0, 2]], s[[0, 2]], 0.5).tolist() == [0, 1]   # 0.5 > 0.5 is false
    assert nms_oracle.nms(b[[0, 2]], s[[0, 2]], 0.49).tolist() == [0]
    # stable order on equal scores
    s2 = np.array([0.5, 0.5, 0.5], np.float32)
    far = np.array([[0, 0, 1, 1], [5, 5, 6, 6], [9, 9, 10, 10]], np.float32)
    assert nms_oracle.nms(far, s2, 0.5).tolist() == [0, 1, 2]
    assert nms_oracle.nms(np.zeros((0, 4), np.float32), np.zeros((0,), np.float32), 0.5).shape == (0,)


TAL_CASES = ["basic", "padded", "many_gt", "topk26", "empty"]


@pytest.mark.parametrize("case", TAL_CASES)
def test_tal_oracle_matches_reference(case):
    g = np.load(os.path.join(GOLDEN, f"tal_{case}.npz"))
    meta = json.loads(str(g["meta"]))
    inp = synth.synth_tal_inputs(meta["B"], [tuple(f) for f in meta["feat_sizes"]], meta["strides"], meta["C"],
                                 meta["G"], seed=meta["seed"], n_valid=meta["n_valid"])
    L, Bx, S, F = tal_oracle.assign(*(inp[k].numpy() for k in ("pd_scores", "pd_bboxes", "anc_points", "gt_labels",
                                                                "gt_bboxes", "mask_gt")),
                                    topk=meta["topk"], num_classes=meta["C"])
    assert np.array_equal(F.astype(bool), g["fg"])
    assert np.array_equal(L, g["labels"])                      # bit-exact assignment
    assert np.array_equal(Bx, g["bboxes"])
    ref_scores = np.zeros_like(S)
    idx = g["score_idx"]
    ref_scores[idx[:, 0], idx[:, 1], idx[:, 2]] = g["score_val"]
    assert np.array_equal(S != 0, ref_scores != 0)
    np.testing.assert_allclose(S, ref_scores, rtol=2e-6, atol=1e-12)


ATSS_CASES = ["basic", "padded_nopd", "p6", "empty"]


def _atss_inputs(g):
    meta = json.loads(str(g["meta"]))
    inp = synth.synth_tal_inputs(meta["B"], [tuple(f) for f in meta["feat_sizes"]], meta["strides"], meta["C"],
                                 meta["G"], seed=meta["seed"], n_valid=meta["n_valid"])
    return meta, inp


@pytest.mark.parametrize("case", ATSS_CASES)
def test_atss_oracle_matches_reference(case):
    from oracle import atss_oracle
    from yolov6_amd.assigners import generate_anchors
    g = np.load(os.path.join(GOLDEN, f"atss_{case}.npz"))
    meta, inp = _atss_inputs(g)
    # the product's host-side generate_anchors equals the reference's (anchors stored in the golden)
    feats = [torch.zeros(1, 1, h, w) for h, w in meta["feat_sizes"]]
    anchors, _, n_list, _ = generate_anchors(feats, meta["strides"], 5.0, 0.5, device="cpu", is_eval=False)
    assert np.array_equal(anchors.numpy(), g["anchors"]) and list(n_list) == g["n_list"].tolist()
    L, Bx, S, F = atss_oracle.assign(g["anchors"], g["n_list"].tolist(), inp["gt_labels"].numpy(),
                                     inp["gt_bboxes"].numpy(), inp["mask_gt"].numpy(),
                                     inp["pd_bboxes"].numpy() if meta["with_pd"] else None, topk=9,
                                     num_classes=meta["C"])
    assert np.array_equal(F.astype(bool), g["fg"])
    assert np.array_equal(L, g["labels"])
    assert np.array_equal(Bx, g["bboxes"])
    ref = np.zeros_like(S)
    idx = g["score_idx"]
    ref[idx[:, 0], idx[:, 1], idx[:, 2]] = g["score_val"]
    assert np.array_equal(S != 0, ref != 0)
    np.testing.assert_allclose(S, ref, rtol=2e-6, atol=1e-12)


# ------------------------------------------------------------------ ComputeLoss forward value (a16)
LOSS_GOLDEN = sorted(f[len("loss_"):-len(".npz")] for f in os.listdir(GOLDEN) if f.startswith("loss_"))


@pytest.mark.parametrize("case", LOSS_GOLDEN)
def test_loss_oracle_matches_reference_golden(case):
    """oracle/loss_oracle.py vs the unmodified reference's ComputeLoss (tests/golden/gen_golden.py::gen_loss)."""
    from oracle import loss_oracle
    g = np.load(os.path.join(GOLDEN, f"loss_{case}.npz"))
    m = json.loads(str(g["meta"]))
    inp = synth.synth_loss_inputs(m["B"], m["feat_sizes"], m["strides"], m["C"], m["reg_max"], m["use_dfl"], seed=m["seed"])
    targets = inp["targets"].numpy()
    if case == "no_targets":
        targets = targets[:0]
    out = loss_oracle.compute_loss(m["feat_sizes"], inp["pred_scores"].numpy(), inp["pred_distri"].numpy(), targets,
                                   m["epoch"], inp["img"], inp["img"], fpn_strides=m["strides"], num_classes=m["C"],
                                   warmup_epoch=4, use_dfl=m["use_dfl"], reg_max=m["reg_max"], iou_type=m["iou_type"])
    # the reference carries the targets in float64 (loss.py:189 builds them from python floats); fp32 here: 2e-5
    np.testing.assert_allclose(out["loss"], float(g["loss"]), rtol=2e-5, atol=1e-6)
    np.testing.assert_allclose(out["loss_items"], g["items"], rtol=2e-5, atol=1e-6)


@pytest.mark.parametrize("case", LOSS_GOLDEN if "LOSS_GOLDEN" in globals() else
                         sorted(f[len("loss_"):-len(".npz")] for f in os.listdir(GOLDEN) if f.startswith("loss_")))
def test_loss_gradient_oracle_matches_reference_autograd(case):
    """oracle/loss_grad_oracle.py (autograd restatement, float64) vs the gradients the unmodified reference
    back-propagates through ComputeLoss (tests/golden/lossgrad_*.npz)."""
    from oracle import loss_grad_oracle
    g = np.load(os.path.join(GOLDEN, f"loss_{case}.npz"))
    gg = np.load(os.path.join(GOLDEN, f"lossgrad_{case}.npz"))
    m = json.loads(str(g["meta"]))
    inp = synth.synth_loss_inputs(m["B"], m["feat_sizes"], m["strides"], m["C"], m["reg_max"], m["use_dfl"], seed=m["seed"])
    targets = inp["targets"].numpy()
    if case == "no_targets":
        targets = targets[:0]
    out = loss_grad_oracle.compute_loss_with_grads(
        m["feat_sizes"], inp["pred_scores"].numpy(), inp["pred_distri"].numpy(), targets, m["epoch"], inp["img"], inp["img"],
        fpn_strides=m["strides"], num_classes=m["C"], warmup_epoch=4, use_dfl=m["use_dfl"], reg_max=m["reg_max"],
        iou_type=m["iou_type"])
    np.testing.assert_allclose(out["loss"], float(gg["loss"]), rtol=2e-5, atol=1e-6)
    for name in ("dscores", "ddistri"):
        ref = gg[name].astype(np.float64)
        scale = max(float(np.abs(ref).max()), 1e-12)
        err = float(np.abs(out[name] - ref).max()) / scale
        assert err < 2e-4, f"{case}: {name} deviates from the reference's autograd by {err:.3e} of its max"


# ------------------------------------------------------------------ training-mode forward (K15 groundwork)
TRAIN_GOLDEN = sorted(f[len("train_"):-len(".npz")] for f in os.listdir(GOLDEN) if f.startswith("train_"))


@pytest.mark.parametrize("case", TRAIN_GOLDEN)
def test_train_mode_forward_oracle_matches_reference_golden(case):
    """oracle/model_oracle.py::TrainOracle (batch-statistics BN, Detect training branch) vs the reference Model in
    .train() mode (tests/golden/gen_golden.py::gen_train_forward): outputs and the BatchNorm running statistics."""
    from oracle.model_oracle import TrainOracle
    cfg, meta = case_config(case)
    sd = synth_sd_from_keys(meta["train"])
    g = np.load(os.path.join(GOLDEN, f"train_{case}.npz"))
    x = synth.synth_images(max(meta["batch"], 2), meta["size"], seed=21)
    orc = TrainOracle(cfg, sd, meta["num_classes"])
    gprobes = [k[len("grad:"):] for k in g.files if k.startswith("grad:")]
    assert gprobes
    for q in gprobes:
        orc.sd[q].requires_grad_(True)
    (xs, cls_scores, reg_distri), feats = orc.forward_train(x)
    # backward through the oracle graph: the same scalar the golden script back-propagated through the reference
    scalar = (cls_scores * cls_scores).sum() + reg_distri.square().mean()
    scalar.backward()
    np.testing.assert_allclose(float(scalar), float(g["scalar"]), rtol=1e-4)
    for q in gprobes:
        ga, gb = orc.sd[q].grad.numpy(), g["grad:" + q]
        scale = max(1e-6, float(np.abs(gb).max()))
        assert float(np.abs(ga - gb).max()) / scale < 2e-2, f"gradient of {q}"   # same fp32 conditioning as the forward
    cls_scores, reg_distri = cls_scores.detach(), reg_distri.detach()
    xs, feats = [t.detach() for t in xs], [t.detach() for t in feats]
    # batch statistics over as few as 8 samples (2 images x 2x2 maps at the last level) divide by small variances, so
    # the graph is ill-conditioned in fp32: evaluating the SAME oracle graph in float64 moves m_tiny's reg outputs by
    # 2.5e-3 and sits 3.1e-3 from the reference's fp32 result.  5e-3 bounds that noise; a wrong branch or a wrong
    # statistic is orders of magnitude larger.
    assert rel_err(cls_scores.numpy(), g["cls_scores"]) < 1e-3
    assert rel_err(reg_distri.numpy(), g["reg_distri"]) < 5e-3
    assert rel_err(xs[0].numpy(), g["stem0"]) < 5e-3
    assert rel_err(feats[0].numpy(), g["feat0"]) < 5e-3
    probes = [k[:-len(".running_mean")] for k in g.files if k.endswith(".running_mean")]
    assert probes
    for q in probes:
        mean, var = orc.new_stats[q]
        assert rel_err(mean.detach().numpy(), g[q + ".running_mean"]) < 1e-4, q
        assert rel_err(var.detach().numpy(), g[q + ".running_var"]) < 1e-4, q


# ------------------------------------------------------------------ int8 QARepVGG oracle (a17: parity unpinned, spec only)
def test_int8_oracle_spec_properties():
    """The int8 oracle DEFINES the quantisation (no int8 exists in the reference tree): check what a definition can
    be checked for - exact integer accumulators inside int32, determinism, the rounding rules, sensible accuracy against
    the fp16 graph it is derived from."""
    from oracle.int8_oracle import Int8Oracle, quantize_sym, quantize_act, act_constants
    cfg, meta = case_config("s_qa_tiny")
    sd = deploy_state_dict(cfg, synth_sd_from_keys(meta["train"]), meta["num_classes"])
    cal = [synth.synth_images(2, meta["size"], seed=100 + i) for i in range(4)]
    x = synth.synth_images(meta["batch"], meta["size"], seed=1)
    q = Int8Oracle(cfg, sd, meta["num_classes"])
    amax = q.calibrate(cal)
    assert len(amax) > 20 and all(v > 0 for v in amax)
    with torch.no_grad():
        d8, _ = q.forward(x)
        n_q = len(q.stats)
        d8b, _ = q.forward(x)
        d16, _ = Oracle(cfg, sd, meta["num_classes"], emulate_fp16=True).forward(x)
    assert torch.equal(d8, d8b)                                               # deterministic
    assert n_q == len(amax) == len(q.layers)                                  # every calibrated conv ran quantised
    assert all(s["acc_absmax"] < 2 ** 31 for s in q.stats)                    # fits the int32 accumulator
    assert q.layers[0]["cin"] >= 8                                            # the image-reading conv is not in the table
    # weights: round-half-even and clamping
    t = torch.tensor([0.5, 1.5, 2.5, -0.5, -1.5, 300.0, -300.0])
    assert quantize_sym(t, 1.0).tolist() == [0.0, 2.0, 2.0, -0.0, -2.0, 127.0, -127.0]
    # activations: fp16 constants, exact product, ONE half-to-even rounding, never +-128
    a, inv = act_constants(3.0)
    assert a == 3.0 and inv == float(torch.tensor(127.0 / 3.0).half())
    xs = torch.tensor([3.0, -3.0, 10.0, -10.0, 0.0, 1.5 / inv, 2.5 / inv]).half().float()
    got = quantize_act(xs, 3.0).tolist()
    assert got[:5] == [127.0, -127.0, 127.0, -127.0, 0.0] and all(abs(v) <= 127 for v in got)
    # accuracy against the fp16 deploy graph on this random-weight model: class scores within a few 1e-2
    e = float((d8[..., 5:] - d16[..., 5:]).abs().max())
    print(f"int8 vs fp16 class scores: max abs diff {e:.3e}")
    assert e < 0.1


# ------------------------------------------------------------------ fuse_ab
def test_fuseab_train_oracle_matches_reference_golden():
    """TrainOracle.head_train_fuseab vs the unmodified reference Model(fuse_ab=True) in training mode: the four head outputs
    and seven reference parameter gradients."""
    from oracle.model_oracle import TrainOracle
    with open(os.path.join(GOLDEN, "keys_tiny_fuseab.json")) as f:
        meta = json.load(f)
    g = np.load(os.path.join(GOLDEN, "fuseab_train_tiny.npz"))
    cfg, _ = case_config("tiny")
    sd = synth_sd_from_keys(meta["train"])
    params = {k: v.clone().float().requires_grad_(True) for k, v in sd.items() if v.dtype.is_floating_point and "running_" not in k}
    orc = TrainOracle(cfg, sd, meta["num_classes"])
    orc.sd = {k: (params[k] if k in params else v.float()) for k, v in sd.items()}
    x = synth.synth_images(max(meta["batch"], 2), meta["size"], seed=21)
    (xs, cab, rab, caf, raf), _ = orc.forward_train_fuseab(x, meta["anchors_init"])
    for name, t in (("cls_ab", cab), ("reg_ab", rab), ("cls_af", caf), ("reg_af", raf)):
        np.testing.assert_allclose(t.detach().numpy(), g[name], rtol=2e-4, atol=1e-4, err_msg=name)
    scalar = (cab * cab).sum() + rab.square().mean() + (caf * caf).sum() + raf.square().mean()
    scalar.backward()
    np.testing.assert_allclose(float(scalar), float(g["scalar"]), rtol=1e-5)
    for k in g.files:
        if k.startswith("grad:"):
            ref = g[k]
            got = params[k[5:]].grad.numpy()
            assert np.abs(got - ref).max() <= 2e-3 * max(np.abs(ref).max(), 1e-6), k


@pytest.mark.parametrize("case", ["giou", "siou"])
def test_fuseab_loss_oracles_match_reference(case):
    """oracle/loss_oracle.py (ab=True) and oracle/loss_grad_oracle.py vs the reference's loss_fuseab.ComputeLoss (value, items,
    gradients wrt pred_scores / pred_distri)."""
    from oracle import loss_grad_oracle
    g = np.load(os.path.join(GOLDEN, f"lossab_{case}.npz"))
    m = json.loads(str(g["meta"]))
    inp = synth.synth_loss_inputs_ab(m["B"], m["feat_sizes"], m["strides"], m["C"], seed=m["seed"])
    out = loss_grad_oracle.compute_loss_with_grads(
        m["feat_sizes"], inp["pred_scores"].numpy(), inp["pred_distri"].numpy(), inp["targets"].numpy(), 10, inp["img"], inp["img"],
        fpn_strides=m["strides"], num_classes=m["C"], warmup_epoch=0, use_dfl=False, reg_max=0, iou_type=m["iou_type"], ab=True)
    np.testing.assert_allclose(out["loss"], float(g["loss"]), rtol=2e-5)
    np.testing.assert_allclose(out["loss_items"], g["items"], rtol=2e-5, atol=1e-6)
    for name in ("dscores", "ddistri"):
        ref = g[name].astype(np.float64)
        err = float(np.abs(out[name] - ref).max()) / max(float(np.abs(ref).max()), 1e-12)
        assert err < 2e-4, f"{case}: {name} {err:.3e}"


def test_fuseab_state_dict_keys_match_reference():
    from yolov6_amd.models.yolo import build_model
    with open(os.path.join(GOLDEN, "keys_tiny_fuseab.json")) as f:
        meta = json.load(f)
    cfg, _ = case_config("tiny")
    model = build_model(cfg, meta["num_classes"], "cpu", fuse_ab=True)
    got = {k: list(v.shape) for k, v in model.state_dict().items()}
    assert list(got) == list(meta["train"]) and got == meta["train"]


# ------------------------------------------------------------------ self-distillation loss (f4 groundwork)
LOSSDISTILL_GOLDEN = sorted(f[len("lossdistill_"):-len(".npz")] for f in os.listdir(GOLDEN) if f.startswith("lossdistill_"))


def _distill_inputs(m):
    inp = synth.synth_loss_inputs(m["B"], m["feat_sizes"], m["strides"], m["C"], m["reg_max"], True, seed=m["seed"])
    tea = synth.synth_loss_inputs(m["B"], m["feat_sizes"], m["strides"], m["C"], m["reg_max"], True, seed=m["seed"] + 50)
    g = torch.Generator().manual_seed(1000 + m["seed"])
    s_feats = [torch.randn((m["B"], c, h, w), generator=g) for c, (h, w) in zip(m["feat_channels"], m["feat_sizes"])]
    t_feats = [torch.randn((m["B"], c, h, w), generator=g) for c, (h, w) in zip(m["feat_channels"], m["feat_sizes"])]
    return inp, tea, [f.numpy() for f in s_feats], [f.numpy() for f in t_feats]


@pytest.mark.parametrize("case", LOSSDISTILL_GOLDEN)
def test_distill_loss_oracle_matches_reference_golden(case):
    """oracle/loss_distill_oracle.py vs the unmodified reference's self-distillation ComputeLoss
    (models/losses/loss_distill.py; tests/golden/gen_golden.py::gen_loss_distill): loss value and the four loss items; the
    gradient golden of the feature maps against the closed form of the channel-wise term."""
    from oracle import loss_distill_oracle as ldo
    g = np.load(os.path.join(GOLDEN, f"lossdistill_{case}.npz"))
    m = json.loads(str(g["meta"]))
    inp, tea, s_feats, t_feats = _distill_inputs(m)
    targets = inp["targets"].numpy()
    if case == "no_targets":
        targets = targets[:0]
    out = ldo.compute_loss_distill(m["feat_sizes"], inp["pred_scores"].numpy(), inp["pred_distri"].numpy(),
                                   tea["pred_scores"].numpy(), tea["pred_distri"].numpy(), s_feats, t_feats, targets,
                                   m["epoch"], m["max_epoch"], m["temperature"], inp["img"], inp["img"],
                                   fpn_strides=m["strides"], num_classes=m["C"], warmup_epoch=m["warmup_epoch"],
                                   reg_max=m["reg_max"], iou_type=m["iou_type"], distill_feat=m["distill_feat"])
    # fp32 reductions of ~1e3-sized sums in the reference: 5e-5 relative
    np.testing.assert_allclose(out["loss"], float(g["loss"]), rtol=5e-5, atol=1e-5)
    np.testing.assert_allclose(out["loss_items"], g["items"], rtol=5e-5, atol=1e-5)
    assert (out["d_loss_cw"] > 0) == bool(m["distill_feat"])
    # d loss / d student feature map = cwd * decay * (softmax_hw(s) - softmax_hw(t)) / (N C)   (temperature 1)
    for i, (s, t) in enumerate(zip(s_feats, t_feats)):
        ref = g[f"dfeat{i}"].astype(np.float64)
        if not m["distill_feat"]:
            assert not ref.any()
            continue
        N, C_, H, W = s.shape
        ps = ldo._softmax64(s.reshape(N, C_, H * W), 2).reshape(s.shape)
        pt = ldo._softmax64(t.reshape(N, C_, H * W), 2).reshape(s.shape)
        want = 10.0 * out["decay"] * (ps - pt) / (N * C_)
        assert float(np.abs(want - ref).max()) < 2e-4 * max(float(np.abs(ref).max()), 1e-12)
    # the class-score gradient carries the KL term everywhere (also on images without targets)
    assert float(np.abs(g["dscores"]).min()) >= 0.0 and float(np.abs(g["dscores"]).max()) > 0.0


@pytest.mark.parametrize("case", ["tal_giou", "atss_warmup_feat", "siou_feat_late"])
def test_distill_ns_loss_oracle_matches_reference_golden(case):
    """The N / S variant (models/losses/loss_distill_ns.py; gen_golden.py::gen_loss_distill_ns): a fourth student output of plain
    (l, t, r, b) distances whose IoU loss is added to the DFL branch's, TAL from epoch 0 (`atss_warmup_feat` has warmup_epoch 4
    and epoch 1: this loss ignores it).  Value and items; the gradient golden of the distances is non-zero exactly on the
    positives of the assignment."""
    from oracle import loss_distill_oracle as ldo
    g = np.load(os.path.join(GOLDEN, f"lossdistillns_{case}.npz"))
    m = json.loads(str(g["meta"]))
    inp, tea, _, _ = _distill_inputs(m)
    gen = torch.Generator().manual_seed(2000 + m["seed"])
    A = inp["pred_scores"].shape[1]
    lrtb = (torch.rand((m["B"], A, 4), generator=gen) * 3.0 + 0.2).numpy()
    np.testing.assert_array_equal(lrtb, g["lrtb"])
    s_feats = [torch.randn((m["B"], c, h, w), generator=gen).numpy() for c, (h, w) in zip(m["feat_channels"], m["feat_sizes"])]
    t_feats = [torch.randn((m["B"], c, h, w), generator=gen).numpy() for c, (h, w) in zip(m["feat_channels"], m["feat_sizes"])]
    kw = dict(fpn_strides=m["strides"], num_classes=m["C"], warmup_epoch=m["warmup_epoch"], reg_max=m["reg_max"],
              iou_type=m["iou_type"], distill_feat=m["distill_feat"])
    args = (m["feat_sizes"], inp["pred_scores"].numpy(), inp["pred_distri"].numpy(), tea["pred_scores"].numpy(),
            tea["pred_distri"].numpy(), s_feats, t_feats, inp["targets"].numpy(), m["epoch"], m["max_epoch"], m["temperature"],
            inp["img"], inp["img"])
    out = ldo.compute_loss_distill(*args, pred_lrtb=lrtb, **kw)
    np.testing.assert_allclose(out["loss"], float(g["loss"]), rtol=5e-5, atol=1e-5)
    np.testing.assert_allclose(out["loss_items"], g["items"], rtol=5e-5, atol=1e-5)
    # the extra IoU term is what separates it from loss_distill.py on the same inputs (where the assigner is the same)
    if m["epoch"] >= m["warmup_epoch"]:
        base = ldo.compute_loss_distill(*args, **kw)
        assert out["loss_items"][0] > base["loss_items"][0]
        np.testing.assert_allclose(out["loss_items"][1:], base["loss_items"][1:], rtol=1e-12)
    touched = np.abs(g["dlrtb"]).sum(-1) > 0
    assert int(touched.sum()) == out["num_pos"]


def test_int8_oracle_exact_int_conv_equals_float64_conv():
    """oracle/int8_oracle.py::exact_int_conv (grouped fp32 convolutions, every partial sum < 2^24) against the float64
    convolution it replaces, on worst-case operands (all +-127: the largest partial sums) and on random codes."""
    import torch.nn.functional as F
    from oracle.int8_oracle import exact_int_conv
    g = torch.Generator().manual_seed(0)
    for cin, cout, k, s, hw in [(200, 24, 3, 1, 9), (64, 16, 3, 2, 12), (1024, 8, 1, 1, 5), (130, 8, 3, 1, 7)]:
        for worst in (True, False):
            if worst:
                x = torch.full((1, cin, hw, hw), 127.0)
                w = torch.full((cout, cin, k, k), -127.0)
            else:
                x = torch.randint(-127, 128, (2, cin, hw, hw), generator=g).float()
                w = torch.randint(-127, 128, (cout, cin, k, k), generator=g).float()
            want = F.conv2d(x.double(), w.double(), None, stride=s, padding=k // 2)
            got = exact_int_conv(x, w, s)
            assert got.dtype == torch.float64 and torch.equal(got, want), (cin, cout, k, s, worst)


# ---------------------------------------------------------------- letterbox (row f3): host logic pinned to the reference's own function
def _reference_letterbox_with_stub_cv2():
    """The reference's letterbox() source (yolov6/data/data_augment.py) executed with a stand-in `cv2` module: opencv is not
    installed here, so the two cv2 calls are served by the oracle's restatement - everything AROUND them (scale ratio, rounding
    of the new size, `auto` stride modulus, the top / bottom / left / right split) is the reference's own code."""
    import ast
    import types
    import numpy as np
    from oracle import letterbox_oracle as LO
    path = os.path.join(os.environ.get("Y6_REFERENCE_ROOT", "/root/reference"), "yolov6", "data", "data_augment.py")
    if not os.path.exists(path):
        pytest.skip("reference checkout not present")
    src = open(path).read()
    tree = ast.parse(src)
    fn = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "letterbox")
    cv2 = types.SimpleNamespace(INTER_LINEAR=1, BORDER_CONSTANT=0)
    cv2.resize = lambda im, dsize, interpolation=None: LO.resize_linear_u8(im, dsize[0], dsize[1])

    def copy_make_border(im, top, bottom, left, right, kind, value=None):
        out = np.empty((im.shape[0] + top + bottom, im.shape[1] + left + right, 3), np.uint8)
        out[...] = np.asarray(value, np.uint8)
        out[top:top + im.shape[0], left:left + im.shape[1]] = im
        return out
    cv2.copyMakeBorder = copy_make_border
    ns = {"np": np, "cv2": cv2}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), path, "exec"), ns)
    return ns["letterbox"]


@pytest.mark.parametrize("shape", [(480, 640), (1080, 1920), (375, 500), (640, 640), (1280, 1280), (333, 1000), (97, 61)])
@pytest.mark.parametrize("new_shape,auto,scaleup", [((640, 640), True, True), (640, False, True), ((1280, 1280), True, False), ([416], True, True)])
def test_letterbox_oracle_host_logic_equals_reference(shape, new_shape, auto, scaleup):
    import numpy as np
    from oracle import letterbox_oracle as LO
    ref_letterbox = _reference_letterbox_with_stub_cv2()
    rng = np.random.default_rng(5)
    im = rng.integers(0, 256, shape + (3,), dtype=np.uint8)
    exp, r_exp, off_exp = ref_letterbox(im.copy(), new_shape, (114, 114, 114), auto, scaleup, 32)
    got, r, off = LO.letterbox(im, new_shape, (114, 114, 114), auto, scaleup, 32)
    assert r == r_exp and tuple(off) == tuple(off_exp)
    assert got.shape == exp.shape and np.array_equal(got, exp)


def test_resize_linear_restatement_known_answers():
    """cv2.resize(INTER_LINEAR) facts that need no cv2 to state: half-pixel centres with clamped borders ([0, 100] -> 4 wide is
    [0, 25, 75, 100]), the identity, constant images stay constant at any scale, an exact 2 x 2 down-scale is the rounded mean
    of each 2 x 2 block (resize.cpp routes it to INTER_AREA's fast path)."""
    import numpy as np
    from oracle import letterbox_oracle as LO
    row = np.array([[[0, 0, 0], [100, 100, 100]]], np.uint8)
    assert LO.resize_linear_u8(row, 4, 1)[0, :, 0].tolist() == [0, 25, 75, 100]
    rng = np.random.default_rng(6)
    im = rng.integers(0, 256, (37, 53, 3), dtype=np.uint8)
    assert np.array_equal(LO.resize_linear_u8(im, 53, 37), im)
    const = np.full((20, 31, 3), 177, np.uint8)
    for w, h in ((64, 48), (9, 7), (31, 40)):
        assert (LO.resize_linear_u8(const, w, h) == 177).all()
    im2 = rng.integers(0, 256, (40, 60, 3), dtype=np.uint8)
    s = im2.astype(np.int32)
    assert np.array_equal(LO.resize_linear_u8(im2, 30, 20), ((s[0::2, 0::2] + s[0::2, 1::2] + s[1::2, 0::2] + s[1::2, 1::2] + 2) >> 2).astype(np.uint8))
    # monotone ramps stay monotone and inside the source range
    ramp = np.tile(np.arange(0, 250, 5, dtype=np.uint8)[None, :, None], (4, 1, 3))
    out = LO.resize_linear_u8(ramp, 123, 4)[0, :, 0].astype(int)
    assert (np.diff(out) >= 0).all() and out.min() >= 0 and out.max() <= 245


def test_distill_ns_train_oracle_matches_reference_golden():
    """TrainOracle.head_train_distill_ns vs the unmodified reference Model(distill_ns=True) in training mode: the three head outputs
    (class scores, DFL logits, plain distances) and seven reference parameter gradients."""
    from oracle.model_oracle import TrainOracle
    with open(os.path.join(GOLDEN, "keys_tiny_distill_ns.json")) as f:
        meta = json.load(f)
    g = np.load(os.path.join(GOLDEN, "distill_ns_train_tiny.npz"))
    cfg, _ = case_config("tiny")
    sd = synth_sd_from_keys(meta["train"])
    params = {k: v.clone().float().requires_grad_(True) for k, v in sd.items() if v.dtype.is_floating_point and "running_" not in k}
    orc = TrainOracle(cfg, sd, meta["num_classes"])
    orc.sd = {k: (params[k] if k in params else v.float()) for k, v in sd.items()}
    x = synth.synth_images(max(meta["batch"], 2), meta["size"], seed=21)
    (xs, cls_s, dist, lrtb), _ = orc.forward_train_distill_ns(x)
    for name, t in (("cls_scores", cls_s), ("reg_distri", dist), ("reg_lrtb", lrtb)):
        np.testing.assert_allclose(t.detach().numpy(), g[name], rtol=2e-4, atol=1e-4, err_msg=name)
    scalar = (cls_s * cls_s).sum() + dist.square().mean() + lrtb.square().mean()
    scalar.backward()
    np.testing.assert_allclose(float(scalar), float(g["scalar"]), rtol=1e-5)
    for k in g.files:
        if k.startswith("grad:"):
            ref = g[k]
            got = params[k[5:]].grad.numpy()
            assert np.abs(got - ref).max() <= 2e-3 * max(np.abs(ref).max(), 1e-6), k
