"""CPU: host logic of the product package - module/state_dict ABI, re-parameterisation
math, config loading, C-ABI symbol table, and the no-fallback contract."""
import ctypes
import os
import pickle
import re

import numpy as np
import pytest
import torch

from oracle import synth
from oracle.model_oracle import deploy_state_dict
from tests.helpers import case_config, case_meta
from yolov6_amd import _lib
from yolov6_amd.configs import Config, get_config, load_config
from yolov6_amd.layers import common
from yolov6_amd.models.yolo import Model, build_model
from yolov6_amd.utils.torch_utils import fuse_model, switch_to_deploy

CASES = ["tiny", "n", "s", "s_qa_tiny", "l6_tiny", "m_tiny", "s_mbla_tiny", "n6", "m6_tiny", "t_pan", "s_csp_pan_tiny", "n6_pan", "n_base", "s_base_tiny", "s_qav1_tiny"]


@pytest.mark.parametrize("case", CASES)
def test_state_dict_abi_matches_reference(case):
    """Same keys and shapes as the reference Model, in train form and after the deploy transform."""
    cfg, meta = case_config(case)
    m = Model(cfg, 3, meta["num_classes"]).eval()
    assert {k: list(v.shape) for k, v in m.state_dict().items()} == meta["train"]
    assert list(m.state_dict().keys()) == list(meta["train"].keys())      # registration order too
    switch_to_deploy(fuse_model(m))
    assert {k: list(v.shape) for k, v in m.state_dict().items()} == meta["deploy"]


@pytest.mark.parametrize("case", ["tiny", "s_qa_tiny", "m_tiny"])
def test_reparam_matches_oracle(case):
    """fuse_model + switch_to_deploy of the product modules == the oracle's state_dict transform."""
    cfg, meta = case_config(case)
    m = Model(cfg, 3, meta["num_classes"]).eval()
    sd = synth.synth_state_dict(m.state_dict(), 0)
    m.load_state_dict(sd)
    # plan-time (implicit) re-parameterisation of the un-fused block
    exp = deploy_state_dict(cfg, sd, meta["num_classes"])
    for name, mod in m.named_modules():
        if isinstance(mod, common.RepVGGBlock):
            w, b = mod._deploy_weight_bias()
            torch.testing.assert_close(w, exp[name + ".rbr_reparam.weight"], rtol=1e-5, atol=1e-6)
            torch.testing.assert_close(b, exp[name + ".rbr_reparam.bias"], rtol=1e-5, atol=1e-6)
    switch_to_deploy(fuse_model(m))
    got = m.state_dict()
    for k, v in exp.items():
        torch.testing.assert_close(got[k].float(), v.float(), rtol=1e-5, atol=1e-6, msg=k)


def test_builtin_configs_equal_reference_files():
    """When the reference tree is present (build container), the built-in dicts equal configs/*.py."""
    ref = "/root/reference/configs"
    if not os.path.isdir(ref):
        pytest.skip("reference tree not present (GPU box)")
    pairs = {"yolov6n": "yolov6n.py", "yolov6s": "yolov6s.py", "yolov6m": "yolov6m.py", "yolov6l": "yolov6l.py",
             "yolov6l6": "yolov6l6.py", "yolov6s_qa": "qarepvgg/yolov6s_qa.py", "yolov6s_mbla": "mbla/yolov6s_mbla.py",
             "yolov6m_mbla": "mbla/yolov6m_mbla.py", "yolov6l_mbla": "mbla/yolov6l_mbla.py", "yolov6x_mbla": "mbla/yolov6x_mbla.py",
             "yolov6n6": "yolov6n6.py", "yolov6s6": "yolov6s6.py", "yolov6m6": "yolov6m6.py",
             "yolov6n_qa": "qarepvgg/yolov6n_qa.py", "yolov6m_qa": "qarepvgg/yolov6m_qa.py",
             "yolov6n_base": "base/yolov6n_base.py", "yolov6s_base": "base/yolov6s_base.py", "yolov6m_base": "base/yolov6m_base.py",
             "yolov6l_base": "base/yolov6l_base.py", "yolov6t": "experiment/yolov6t.py", "yolov6s_csp": "experiment/yolov6s_csp_scaled.py"}
    for name, f in pairs.items():
        a, b = get_config(name), load_config(os.path.join(ref, f))
        assert a.training_mode == (b.get("training_mode") or "repvgg"), name      # tools/train.py:99-100 default
        for part in ("backbone", "neck"):
            for k, v in b.model[part].items():
                assert a.model[part].get(k) == v or (not a.model[part].get(k) and not v), (name, part, k)
        for k in ("num_layers", "use_dfl", "reg_max", "strides", "atss_warmup_epoch", "iou_type"):
            assert a.model.head.get(k) == b.model.head.get(k), (name, k)
        assert (a.model.depth_multiple, a.model.width_multiple) == (b.model.depth_multiple, b.model.width_multiple)


def test_config_attr_dict():
    c = Config(dict(model=dict(head=dict(num_layers=3)), x=1))
    assert c.model.head.num_layers == 3 and c["model"]["head"]["num_layers"] == 3
    assert c.model.get("missing") is None
    with pytest.raises(AttributeError):
        c.model.nope


def test_get_block_contract():
    assert common.get_block("repvgg") is common.RepVGGBlock
    assert common.get_block("conv_silu") is common.ConvBNSiLU
    with pytest.raises(NotImplementedError):
        common.get_block("bogus")
    with pytest.raises(AssertionError):
        common.RepVGGBlock(8, 8, kernel_size=5)


def test_header_and_library_symbols(hip_lib):
    """Every function declared in include/yolov6_hip.h is bound in _lib.SIGNATURES and exported by the .so."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = open(os.path.join(root, "include", "yolov6_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(y6_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    so = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(so, name), name
    assert hip_lib.y6_abi_version() == 1
    assert hip_lib.y6_conv_variants() >= 2
    assert hip_lib.y6_packed_weight_elems(80, 64, 1) == 4 * 2 * 1 * 1024


def test_no_cpu_fallback(hip_lib):
    """The product path must fail loudly on CPU tensors instead of computing with aten."""
    cfg, meta = case_config("tiny")
    m = build_model(cfg, 80, "cpu").eval()
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.zeros(1, 3, 64, 64))
    from yolov6_amd.utils.nms import non_max_suppression
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        non_max_suppression(torch.zeros(1, 10, 85))
    from yolov6_amd.assigners import TaskAlignedAssigner
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        TaskAlignedAssigner()(torch.zeros(1, 10, 80), torch.zeros(1, 10, 4), torch.zeros(10, 2), torch.zeros(1, 2, 1),
                              torch.zeros(1, 2, 4), torch.zeros(1, 2, 1))


def test_missing_library_is_loud(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(RuntimeError, match="is missing"):
        _lib.load()


def test_modules_pickle_without_plans():
    cfg, _ = case_config("tiny")
    m = Model(cfg, 3, 80).eval()
    m.__dict__["_y6_plans"] = {"k": object()}
    m2 = pickle.loads(pickle.dumps(m))
    assert "_y6_plans" not in m2.__dict__
    assert list(m2.state_dict()) == list(m.state_dict())


def test_install_as_yolov6_aliases():
    import sys
    import yolov6_amd
    saved = {k: v for k, v in sys.modules.items() if k == "yolov6" or k.startswith("yolov6.")}
    try:
        for k in saved:
            del sys.modules[k]
        yolov6_amd.install_as_yolov6()
        from yolov6.layers.common import RepVGGBlock
        from yolov6.models.yolo import Model as M2
        from yolov6.utils.nms import non_max_suppression  # noqa: F401
        assert RepVGGBlock is common.RepVGGBlock and M2 is Model
    finally:
        for k in [k for k in sys.modules if k == "yolov6" or k.startswith("yolov6.")]:
            del sys.modules[k]
        sys.modules.update(saved)


def test_anchor_generator_matches_oracle_grid():
    from yolov6_amd.assigners import generate_anchors
    feats = [torch.zeros(1, 8, 4, 6), torch.zeros(1, 8, 2, 3)]
    pts, st = generate_anchors(feats, [8, 16], is_eval=True)
    assert pts.shape == (30, 2) and st.shape == (30, 1)
    assert pts[0].tolist() == [0.5, 0.5] and pts[7].tolist() == [1.5, 1.5] and st[-1].item() == 16
    anchors, pts2, n, st2 = generate_anchors(feats, [8, 16], 5.0, 0.5, is_eval=False)
    assert n == [24, 6] and anchors.shape == (30, 4)
    assert pts2[0].tolist() == [4.0, 4.0] and anchors[0].tolist() == [-16.0, -16.0, 24.0, 24.0]


def test_compute_loss_target_packing_matches_oracle():
    """ComputeLoss.preprocess is host-side in the reference (loss.py:184-192) and here: same packing, same padding
    rows, same xywh2xyxy arithmetic (x2 = x1 + w) as the oracle restatement."""
    import numpy as np
    from oracle import loss_oracle
    from yolov6_amd.models.losses.loss import ComputeLoss
    from yolov6_amd.utils import synth
    inp = synth.synth_loss_inputs(4, [(8, 8), (4, 4), (2, 2)], [8, 16, 32], 20, 16, True, seed=5)
    crit = ComputeLoss(num_classes=20)
    scale = torch.tensor([64.0, 48.0, 64.0, 48.0])
    got = crit.preprocess(inp["targets"], 4, scale).numpy()
    ref = loss_oracle.preprocess(inp["targets"].numpy(), 4, scale.numpy())
    assert got.shape == ref.shape and got.shape[0] == 4
    assert np.array_equal(got, ref)
    # an image without targets keeps only padding rows (label -1, zero box) -> mask_gt 0
    assert (got[3, :, 0] == -1).all() and (got[3, :, 1:] == 0).all()
    # no targets at all: G = 0
    assert crit.preprocess(inp["targets"][:0], 4, scale).shape == (4, 0, 5)
    with pytest.raises(ValueError):
        ComputeLoss(iou_type="eiou")


# ---------------------------------------------------------------------------------------------------------------------
# int8 lowering: which activation buffers get a producer-written int8 twin (yolov6_amd/quant.py::plan_twins) - host logic
# ---------------------------------------------------------------------------------------------------------------------
class _FakeRef:
    def __init__(self, bid, C, cstride=None, coff=0):
        self.bid, self.C, self.cstride, self.coff = bid, C, (C if cstride is None else cstride), coff


class _FakePB:
    def __init__(self, op_log, fp16_reads=()):
        self.op_log, self.fp16_reads = op_log, list(fp16_reads)

    @staticmethod
    def buf_id(ref):
        return ref.bid


def test_int8_twin_planning_rules():
    from yolov6_amd.quant import plan_twins
    a, b, c, d, e = (_FakeRef(i, 64) for i in range(5))
    cat = _FakeRef(7, 128)                       # a concat buffer written in two slices
    cat_lo, cat_hi = _FakeRef(7, 64, 128, 0), _FakeRef(7, 64, 128, 64)
    odd = _FakeRef(9, 24)                        # 24 channels: not a whole number of 16-byte int8 pieces
    log = [
        dict(kind="stem", out=a),                                                    # fp16 producer: `a` cannot have a twin
        dict(kind="conv_i8", x=a, out=b, amax=2.0, res=None),
        dict(kind="conv_i8", x=b, out=c, amax=3.0, res=None),                        # b: one int8 reader, int8 writer -> twin, no fp16
        dict(kind="conv_i8", x=c, out=cat_lo, amax=1.5, res=None),
        dict(kind="conv_i8", x=c, out=cat_hi, amax=1.5, res=None),                   # c: two readers, ONE scale -> twin
        dict(kind="conv_i8", x=cat, out=d, amax=4.0, res=None),                      # cat: both writers int8, one reader scale -> twin
        dict(kind="convt", x=d, out=e),                                              # d: read by an fp16 op only -> no twin
        dict(kind="conv_i8", x=e, out=odd, amax=1.0, res=None),                      # e: written by an fp16 op -> no twin
        dict(kind="conv_i8", x=odd, out=_FakeRef(11, 64), amax=1.0, res=None),       # odd: misaligned -> no twin
        dict(kind="conv_i8", x=b, out=_FakeRef(12, 64), amax=3.0, res=c),            # c also read as an fp16 residual
    ]
    dec = plan_twins(_FakePB(log, fp16_reads=[d]))
    assert dec[0]["twin"] is False                                   # written by the stem
    assert dec[1]["twin"] and dec[1]["amax"] == 3.0 and dec[1]["fp16"] is False
    assert dec[2]["twin"] and dec[2]["amax"] == 1.5 and dec[2]["fp16"] is True     # the residual read keeps its fp16 form
    assert dec[7]["twin"] and dec[7]["amax"] == 4.0 and dec[7]["fp16"] is False
    assert dec[3]["twin"] is False and dec[3]["fp16"] is True        # d: fp16 reader (convT) and a lazy feature-map read
    assert dec[4]["twin"] is False
    assert dec[9]["twin"] is False
    # two int8 readers with DIFFERENT scales: no twin
    log2 = [dict(kind="conv_i8", x=a, out=b, amax=2.0, res=None), dict(kind="conv_i8", x=b, out=c, amax=3.0, res=None),
            dict(kind="conv_i8", x=b, out=d, amax=3.5, res=None)]
    assert plan_twins(_FakePB(log2))[1]["twin"] is False


def test_pmc_kernel_classification():
    """tools/pmc_traffic.py maps rocprofv3 kernel names to the bench's kernel classes (roofline.traffic depends on it)."""
    import importlib.util, os, sys
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "pmc_traffic.py")
    src = open(path).read()
    ns = {}
    exec(src[src.index("def classify"):src.index("res = {}")], {"re": __import__("re")}, ns)
    classify = ns["classify"]
    ns_ = "void (anonymous namespace)::"
    assert classify(ns_ + "conv3x3_dma_kernel<2, 2, 4, 2, 2, 1, 16, false, 1>((anonymous namespace)::ConvKArgs)") == "conv3x3s1"
    assert classify(ns_ + "conv3x3_dma_kernel<2, 1, 8, 2, 2, 1, 16, false, 2>((anonymous namespace)::ConvKArgs)") == "conv3x3s2"
    assert classify(ns_ + "conv3x3_dma_kernel<2, 1, 8, 2, 2, 1, 16, false, 2, false>((anonymous namespace)::ConvKArgs)") == "conv3x3s2"
    assert classify(ns_ + "conv3x3_dma_kernel<2, 2, 8, 2, 2, 1, 32, false, 1, true>((anonymous namespace)::ConvKArgs)") == "conv3x3s1"
    assert classify(ns_ + "conv3x3_dma_kernel<2, 2, 8, 2, 2, 1, 32, true, 1, false>((anonymous namespace)::ConvKArgs)") == "conv3x3s1"
    assert classify(ns_ + "conv_mfma_pipe_kernel<2, 2, 2, 4, 1, 2>((anonymous namespace)::ConvKArgs)") == "conv3x3s1"
    assert classify(ns_ + "conv_mfma_kernel<4, 1, 3, 2>((anonymous namespace)::ConvKArgs)") == "conv3x3s2"
    assert classify(ns_ + "conv_mfma_kernel<2, 1, 1, 1>((anonymous namespace)::ConvKArgs)") == "conv1x1s1"
    assert classify(ns_ + "conv1x1_stream_kernel<1, 4>((anonymous namespace)::ConvKArgs)") == "conv1x1s1"
    assert classify("(anonymous namespace)::nms_sweep_kernel<1024>(float const*, int)") == "nms"


def test_distill_ns_head_state_dict_abi_and_eval_oracle():
    """Model(distill_ns=True) (heads/effidehead_distill_ns.py): parameter names / shapes / order equal the reference's, and the
    eval branch (plain distances from reg_preds, no DFL) is the oracle's non-DFL head on the same weights, equal to the
    reference's eval output (tests/golden/model_tiny_distill_ns.npz)."""
    import copy
    import json
    import numpy as np
    from oracle import synth
    from oracle.model_oracle import Oracle
    from tests.helpers import GOLDEN, synth_sd_from_keys
    from yolov6_amd.configs import tiny_config
    from yolov6_amd.models.yolo import build_model
    with open(os.path.join(GOLDEN, "keys_tiny_distill_ns.json")) as f:
        meta = json.load(f)
    cfg = tiny_config()
    m = build_model(cfg, meta["num_classes"], "cpu", distill_ns=True)
    got = {k: list(v.shape) for k, v in m.state_dict().items()}
    assert list(got.keys()) == list(meta["train"].keys())
    assert got == meta["train"]
    assert m.detect._eval_use_dfl() is False
    sd = synth_sd_from_keys(meta["train"])
    m.load_state_dict(sd)
    ocfg = copy.deepcopy(cfg)
    ocfg.model.head.use_dfl = False          # the distillation head's eval decode never projects DFL bins
    x = synth.synth_images(meta["batch"], meta["size"], seed=1)
    with torch.no_grad():
        det, _ = Oracle(ocfg, sd, meta["num_classes"]).forward(x, train_form=True)
    g = np.load(os.path.join(GOLDEN, "model_tiny_distill_ns.npz"))["det_train"]
    assert float(np.abs(det.numpy() - g).max() / max(1.0, float(np.abs(g).max()))) < 1e-4
    # the training branch exists on the HIP path now (tests/test_gpu_training.py::test_distill_ns_training_graph_vs_oracle): it
    # returns three head outputs - class scores, DFL logits, plain distances - from the SAME reg_conv features
    assert callable(m.detect.lower_train) and len(m.detect.reg_preds_dist) == len(m.detect.reg_preds) == 3


WIRING_CASES = CASES + ["tiny_distill_ns", "tiny_fuseab_eval"]


@pytest.mark.parametrize("case", WIRING_CASES)
def test_lowering_wiring_matches_reference_golden(case):
    """Every model family's `lower()` graph, executed op by op on the CPU by tests/mock_plan.py (fp32 torch ops on the same
    buffer / channel-slice views the real builder hands out), reproduces the REFERENCE's deploy-form output: concat-free
    slice wiring, re-parameterised weights, epilogue order, decode arguments.  The HIP kernels are not involved."""
    import copy
    import json
    import numpy as np
    from tests.helpers import GOLDEN, case_golden, rel_err, synth_sd_from_keys
    from tests.mock_plan import MockBuilder
    from yolov6_amd.configs import tiny_config
    from yolov6_amd.engine import NCHWInput
    from yolov6_amd.utils.torch_utils import fuse_model, switch_to_deploy
    special = case in ("tiny_distill_ns", "tiny_fuseab_eval")     # heads selected by a Model() flag: train-form goldens only
    if special:
        with open(os.path.join(GOLDEN, "keys_tiny_distill_ns.json" if case == "tiny_distill_ns" else "keys_tiny_fuseab.json")) as f:
            meta = json.load(f)
        m = build_model(tiny_config(), meta["num_classes"], "cpu", distill_ns=case == "tiny_distill_ns",
                        fuse_ab=case == "tiny_fuseab_eval").eval()
        gold, gfeats = np.load(os.path.join(GOLDEN, f"model_{case}.npz"))["det_train"], None
    else:
        cfg, meta = case_config(case)
        m = build_model(cfg, meta["num_classes"], "cpu").eval()
        g = case_golden(case)
        gold, gfeats = g["det_deploy"], [g[k] for k in sorted(k for k in g.files if k.startswith("feat"))]
    m.load_state_dict(synth_sd_from_keys(meta["train"]))
    if not special:
        m.detect.proj_conv.weight.data = m.detect.proj.view(1, -1, 1, 1).clone()
    switch_to_deploy(fuse_model(m))
    x = synth.synth_images(meta["batch"], meta["size"], seed=1)
    pb = MockBuilder()
    with torch.no_grad():
        det = m.lower(pb, NCHWInput(x))
    assert rel_err(det.numpy(), gold) < 2e-4
    if gfeats is not None:
        feats = [pb.to_nchw(r).numpy() for r in m._featrefs]
        assert len(feats) == len(gfeats)
        for f, gf in zip(feats, gfeats):
            assert rel_err(f, gf) < 2e-4
    kinds = [e["kind"] for e in pb.op_log]
    assert kinds.count("decode") + kinds.count("pred_decode") == 1 and kinds.count("conv") > 20
    # the same op sequence under the fp16-emulating oracle's per-op arithmetic (tests/plan_replay.py - what the GPU parity tests
    # compare each HIP op with) must give the oracle's own whole-model forward: the lowering's op boundaries ARE the oracle's
    # rounding points.  Bit-exact here: the oracle reads the model's own deploy-form state_dict, so both multiply the same folded weights.
    import types
    from oracle.model_oracle import Oracle
    from tests.plan_replay import OracleChain
    ocfg = copy.deepcopy(tiny_config() if special else cfg)
    if special:
        ocfg.model.head.use_dfl = False
    orc = Oracle(ocfg, m.state_dict(), meta["num_classes"], emulate_fp16=True)
    xq = x.half().float()
    with torch.no_grad():
        ref, _ = orc.forward(xq)
        chain = OracleChain(types.SimpleNamespace(op_log=pb.op_log), orc)
        chain.run(teacher_force=False)
    sync = float(((chain.final - ref).abs() / ref.abs().clamp(min=1.0)).max())
    assert sync == 0.0, f"{case}: lowered op chain vs Oracle.forward {sync:.3e}"


@pytest.mark.parametrize("case", WIRING_CASES)
def test_two_stream_schedule_enforces_every_dependence(case):
    """yolov6_amd/schedule.py on the lowering graph of every model family (CPU mock of the plan builder): the ops off the
    critical path go to the side stream, the enqueue order is topological, and stream FIFO + the cross-stream event edges
    cover every RAW / WAR / WAW dependence between the ops' tensor views (checked by an independent closure walk)."""
    import json
    from tests.helpers import GOLDEN, synth_sd_from_keys
    from tests.mock_plan import MockBuilder
    from yolov6_amd import schedule as S
    from yolov6_amd.configs import tiny_config
    from yolov6_amd.engine import NCHWInput
    from yolov6_amd.utils.torch_utils import fuse_model, switch_to_deploy
    special = case in ("tiny_distill_ns", "tiny_fuseab_eval")
    if special:
        with open(os.path.join(GOLDEN, "keys_tiny_distill_ns.json" if case == "tiny_distill_ns" else "keys_tiny_fuseab.json")) as f:
            meta = json.load(f)
        m = build_model(tiny_config(), meta["num_classes"], "cpu", distill_ns=case == "tiny_distill_ns",
                        fuse_ab=case == "tiny_fuseab_eval").eval()
    else:
        cfg, meta = case_config(case)
        m = build_model(cfg, meta["num_classes"], "cpu").eval()
    m.load_state_dict(synth_sd_from_keys(meta["train"]))
    switch_to_deploy(fuse_model(m))
    pb = MockBuilder()
    with torch.no_grad():
        m.lower(pb, NCHWInput(synth.synth_images(1, meta["size"], seed=1)))
    acc = [S.op_access(e) for e in pb.op_log]
    assert all(a is not None for a in acc)
    deps = S.dependences(acc)
    n = len(deps)
    # a conv reads what the op before it in its chain wrote; the decode waits for every head level
    # (the mock logs the image conv as an adapter op + a stem op reading the image itself: ops 0 / 1 have no producer)
    assert all(deps[j] for j in range(2, n)) and len(deps[-1]) >= len(m.detect.stems)
    # costs as the device would see them: proportional to the MACs of a conv, a floor for everything else
    cost = []
    for e in pb.op_log:
        if e["kind"] in ("conv", "stem"):
            co, ci, k, _ = e["w"].shape
            cost.append(5.0 + co * ci * k * k * e["out"].H * e["out"].W * 1e-6)
        else:
            cost.append(5.0)
    res = S.build_schedule(deps, cost)
    assert res is not None, "every YOLOv6 family has lateral / head branches off the critical path"
    order, stream, edges = res
    S.check_schedule(deps, order, stream, edges)
    pos = {op: i for i, op in enumerate(order)}
    side = [i for i in range(n) if stream[i]]
    main = [i for i in range(n) if not stream[i]]
    assert [i for i in order if not stream[i]] == main, "main-stream ops keep plan order"
    assert stream[0] == 0 and stream[n - 1] == 0 and 2 <= len(side) < n // 2
    # hoisting: some side op is enqueued EARLIER than its plan position (it overlaps the chain instead of following it)
    assert any(pos[i] < i for i in side)
    # every edge crosses streams, and nothing waits for a later op
    assert all(stream[a] != stream[b] and pos[a] < pos[b] for a, b in edges)
    # uniform costs (what a build without a device profile would use) must give a valid schedule too
    r2 = S.build_schedule(deps, None)
    if r2 is not None:
        S.check_schedule(deps, *r2)


@pytest.mark.parametrize("case,flags", [("tiny", {}), ("s_qa_tiny", {}), ("s_mbla_tiny", {}), ("tiny", {"fuse_ab": True}),
                                        ("tiny", {"distill_ns": True}), ("m_tiny", {})])
def test_training_forward_schedule_enforces_every_dependence(case, flags):
    """The training-form forward (train_engine.TrainBuilder lowers it without a GPU: plans only record launches) under
    schedule.train_op_access / build_schedule: every RepVGG block's 1x1 branch leaves the chain, BatchNorm statistics are
    ordered in front of the branch sum that reads them, running statistics are written by one op each, and stream FIFO + the
    event edges cover every dependence."""
    from yolov6_amd import schedule as S
    from yolov6_amd.engine import NCHWInput
    from yolov6_amd.train_engine import ParamArena, TrainBuilder
    cfg, meta = case_config(case)
    m = build_model(cfg, meta["num_classes"], "cpu", **flags).train()
    x = synth.synth_images(2, 64, seed=1)
    tb = TrainBuilder(x.device, ParamArena(m, x.device))
    m.lower_train(tb, NCHWInput(x))
    log = tb.fwd_log
    acc = [S.train_op_access(e) for e in log]
    assert all(a is not None for a in acc), sorted({e["kind"] for e, a in zip(log, acc) if a is None})
    deps = S.dependences(acc)
    kinds = [e["kind"] for e in log]
    for j, e in enumerate(log):
        if e["kind"] == "bnact_forward":     # the branch sum waits for the statistics op of every normalised branch
            stats_ops = [i for i in deps[j] if kinds[i].startswith("bn_train_stats")]    # (a block's branches may share one op)
            assert sum(len(log[i]["items"]) if kinds[i] == "bn_train_stats_multi" else 1 for i in stats_ops) == \
                sum(1 for _, st in e["branches"] if st is not None)
        if e["kind"].startswith("bn_train_stats"):    # ... which waits for the op(s) that wrote the tensor(s) it reduces - and for nothing else
            assert deps[j] and all(kinds[i] in ("conv", "stem", "bnact_forward", "nchw2nhwc", "subsample2", "convt", "avgpool3", "sppf")
                                   for i in deps[j])       # (more than one producer: statistics over a concat buffer)
    for policy in ("asap", "alap"):
        res = S.build_schedule(deps, S.train_costs(log), policy=policy)
        assert res is not None
        order, stream, edges = res
        S.check_schedule(deps, order, stream, edges)
        side = [i for i in range(len(log)) if stream[i]]
        assert any(kinds[i] == "conv" and log[i]["k"] == 1 for i in side) and any(kinds[i].startswith("bn_train_stats") for i in side)
        assert stream[0] == 0 and kinds[-1] in ("head_pack", "head_ab_pack") and len(side) > len(log) // 5


def test_schedule_access_lists_follow_int8_twins():
    """An int8 conv may read its producer's int8 twin instead of the fp16 view and write a twin for its consumers: the
    schedule's dependence analysis must see both buffers (engine.PlanBuilder._conv_i8 logs them as q_in / q_out)."""
    from yolov6_amd import schedule as S
    from yolov6_amd.engine import TRef
    f16 = [torch.zeros(1, 4, 4, 16, dtype=torch.float16) for _ in range(3)]
    i8 = [torch.zeros(1, 4, 4, 16, dtype=torch.int8) for _ in range(3)]
    ref = lambda t: TRef(t, 1, 4, 4, 16, 16, 0)   # noqa: E731
    log = [dict(kind="conv_i8", x=ref(f16[0]), out=ref(f16[1]), q_in=None, q_out=ref(i8[1]), res=None, has_out=False),
           dict(kind="conv_i8", x=ref(f16[1]), out=ref(f16[2]), q_in=ref(i8[1]), q_out=None, res=None, has_out=True),
           dict(kind="conv_i8", x=ref(f16[0]), out=ref(f16[1]), q_in=None, q_out=ref(i8[1]), res=None, has_out=True)]
    deps = S.dependences([S.op_access(e) for e in log])
    assert deps == [[], [0], [0, 1]]            # RAW through the twin; WAW + WAR of the rewrite
    assert S.op_access(dict(kind="absmax", x=ref(f16[0]), index=0)) is None


def test_two_stream_schedule_random_dags():
    """Property check of schedule.build_schedule on random dependence graphs: valid for every graph, and a broken schedule (an
    edge removed) is caught by check_schedule."""
    import random
    from yolov6_amd import schedule as S
    rng = random.Random(7)
    caught = 0
    for trial in range(300):
        n = rng.randint(2, 40)
        deps = [sorted(set(rng.sample(range(j), min(j, rng.randint(0, 3))))) if j else [] for j in range(n)]
        cost = [rng.uniform(1.0, 50.0) for _ in range(n)]
        res = S.build_schedule(deps, cost)
        if S.is_chain(deps) and all(deps[j] for j in range(1, n)):
            assert res is None                      # one straight line: nothing leaves the critical path
        if res is None:
            continue
        order, stream, edges = res
        S.check_schedule(deps, order, stream, edges)
        if edges:
            cut = edges[:-1] if trial % 2 else edges[1:]
            try:
                S.check_schedule(deps, order, stream, cut)
            except AssertionError:
                caught += 1
    assert caught > 50       # (an edge is only dropped by build_schedule when another wait already implies it: most cuts break)


def test_every_public_struct_has_a_size_checked_mirror(hip_lib):
    """include/yolov6_hip.h <-> yolov6_amd/_lib.py: every `typedef struct {...} y6_*;` of the header has a ctypes mirror, and each
    mirror's size equals the C compiler's (y6_abi_sizeof; _lib.load() refuses to return a library whose layouts differ)."""
    import ctypes as C
    import re
    from yolov6_amd import _lib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(root, "include", "yolov6_hip.h")) as f:
        names = re.findall(r"^\} (y6_[a-z0-9_]+);", f.read(), re.M)
    assert len(names) >= 27 and sorted(names) == sorted(_lib.STRUCTS)
    for n, t in _lib.STRUCTS.items():
        assert int(hip_lib.y6_abi_sizeof(n.encode())) == C.sizeof(t) > 0, n
    assert int(hip_lib.y6_abi_sizeof(b"y6_no_such_struct")) == 0


def test_native_state_never_pickled_or_deepcopied():
    """after a train-mode forward the model's __dict__ holds `_y6_train_graphs` / `_y6_arena` /
    `_y6_backward_hook` (ctypes handles inside); the reference's epoch-end path `deepcopy(model).half()` + torch.save must not
    see them.  Every `_y6_*` key is native state and is dropped by HipModule.__getstate__; parameters that are views of a
    (here: stand-in) arena come out of the copy as plain tensors."""
    import copy
    import ctypes
    import io
    import torch
    from yolov6_amd.configs import tiny_config
    from yolov6_amd.models.yolo import build_model
    from yolov6_amd.train_engine import ParamArena
    m = build_model(tiny_config(), 4, "cpu")
    arena = ParamArena(m, "cpu")
    m.__dict__["_y6_arena"] = arena
    m.__dict__["_y6_train_graphs"] = {"k": ctypes.c_void_p(1234)}
    m.__dict__["_y6_backward_hook"] = lambda g, grads: None
    m.__dict__["_y6_plans"] = {"k": ctypes.c_void_p(1)}
    m.backbone.__dict__["_y6_plans"] = {"k": ctypes.c_void_p(2)}
    c = copy.deepcopy(m).half()
    assert not any(k.startswith("_y6_") for mod in c.modules() for k in mod.__dict__)
    assert all(p._base is None and p.dtype == torch.float16 for p in c.parameters())
    buf = io.BytesIO()
    torch.save({"model": c, "live": m}, buf)
    buf.seek(0)
    back = torch.load(buf, weights_only=False)
    for (n1, p1), (n2, p2) in zip(m.named_parameters(), back["live"].named_parameters()):
        assert n1 == n2 and torch.equal(p1, p2)
    # the live model still owns its native state
    assert m.__dict__["_y6_arena"] is arena
    # ... until something moves the parameters
    m.half()
    assert "_y6_arena" not in m.__dict__ and "_y6_train_graphs" not in m.__dict__


def test_result_ring_never_overwrites_a_held_result():
    """`Model.forward` hands out its detections without a copy (`_ResultRing`, models/yolo.py): the reference returns an
    independent tensor per call (yolo.py:37-47), so a result - or a view of one - that the caller still holds must never
    be the next run's output; results the caller dropped are written again (no allocation in the steady state)."""
    from yolov6_amd.models.yolo import _ResultRing

    class FakePlan:
        def __init__(self):
            self.outputs = torch.zeros(3)

        def rebind_output(self, t):
            self.outputs = t

    plan = FakePlan()
    ring = _ResultRing(plan.outputs, 2)

    def call():
        plan.rebind_output(ring.next(plan.outputs))
        return plan.outputs

    held = [call() for _ in range(5)]                   # TTA / results collected over several batches
    assert len({id(t) for t in held}) == 5
    view = call()[0:1]
    later = [call() for _ in range(4)]
    assert all(t is not view._base for t in later)      # a view keeps its base out of the rotation
    del held, view, later
    ids = set()
    for _ in range(8):
        t = call()
        ids.add(id(t))
        del t
    assert len(ids) == 2                                # dropped results: the two slots alternate


def _device_kernel_notes(lib_path):
    """(kernel name -> metadata dict) of every gfx950 code object embedded in the shared library (llvm-readelf --notes)."""
    import re
    import struct
    import subprocess
    import tempfile
    readelf = "/opt/rocm/lib/llvm/bin/llvm-readelf"
    blob = open(lib_path, "rb").read()
    out = {}
    pos = blob.find(b"\x7fELF", 1)
    while pos >= 0:
        e_shoff, = struct.unpack_from("<Q", blob, pos + 0x28)
        e_shentsize, e_shnum = struct.unpack_from("<HH", blob, pos + 0x3A)
        e_machine, = struct.unpack_from("<H", blob, pos + 0x12)
        size = e_shoff + e_shentsize * e_shnum
        if e_machine == 224 and size > 0:   # EM_AMDGPU
            with tempfile.NamedTemporaryFile(suffix=".co") as f:
                f.write(blob[pos:pos + size])
                f.flush()
                txt = subprocess.run([readelf, "--notes", f.name], capture_output=True, text=True).stdout
            cur = None
            for line in txt.splitlines():
                m = re.match(r"\s*-?\s*\.(\w+):\s*(.*)$", line)
                if not m:
                    continue
                k, v = m.group(1), m.group(2).strip()
                if k == "name" and v.startswith("_Z"):
                    cur = out.setdefault(v, {})
                elif cur is not None and k in ("vgpr_spill_count", "sgpr_spill_count", "vgpr_count", "private_segment_fixed_size"):
                    cur[k] = int(v)
        pos = blob.find(b"\x7fELF", pos + 4)
    return out


def test_int8_conv_variant_heuristic_over_the_sqa_layer_shapes(hip_lib):
    """y6_conv2d_i8_variant is host arithmetic (nothing launched): which kernel each int8 conv shape of YOLOv6-S-QA at 640 x 640,
    batch 32 gets, and that every precondition of the register-fed kernels (conv_wreg.hip, int8 form: it has the fast epilogue
    only) sends a conv back to the kernels that can do it.  Pointers are only tested for NULL."""
    import ctypes as C

    def desc(B, H, W, Cin, Cout, k, stride, q_in=True, out=True, q_out=True, res=False, acc=False, out_cstride=None, q_out_coff=0):
        d = _lib.ConvI8Desc()
        Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
        fake = 1 << 20      # 16-byte aligned, never dereferenced
        d.conv.inp = _lib.Tensor(fake, B, H, W, Cin, Cin, 0)
        d.conv.out = _lib.Tensor(fake if out else None, B, Ho, Wo, Cout, out_cstride or Cout, 0)
        d.conv.res = _lib.Tensor(fake if res else None, B, Ho, Wo, Cout, Cout, 0)
        d.conv.w_packed = fake
        d.conv.ksize, d.conv.stride, d.conv.variant = k, stride, 0
        d.dequant = fake
        d.in_amax = 1.0
        d.q_in = _lib.Tensor(fake if q_in else None, B, H, W, Cin, Cin, 0)
        d.q_out = _lib.Tensor(fake if q_out else None, B, Ho, Wo, Cout, Cout + q_out_coff, q_out_coff)
        d.q_out_amax = 1.0
        d.acc_out = fake if acc else None
        return d

    v = lambda *a, **k: hip_lib.y6_conv2d_i8_variant(C.byref(desc(*a, **k)))
    # the backbone of S-QA: (B, H, W, Cin, Cout, k, stride) -> variant
    assert v(32, 80, 80, 128, 128, 3, 1) == 10        # 200-pixel items fill the 512 resident blocks: 7 fragments per wave
    assert v(32, 40, 40, 256, 256, 3, 1) == 10
    assert v(32, 20, 20, 512, 512, 3, 1) == 11        # 4 fragments per wave
    assert v(32, 160, 160, 64, 128, 3, 2) == 12
    assert v(32, 80, 80, 128, 256, 3, 2) == 12
    assert v(32, 160, 160, 64, 64, 3, 2) == 13        # 64-cout blocks, stride 2
    assert v(32, 320, 320, 32, 64, 3, 2) == 13        # 32 input channels: half a stage
    assert v(32, 160, 160, 64, 64, 3, 1) in (8, 9)    # 64 couts at stride 1: the LDS-DMA kernels (their register-fed form lost)
    assert v(32, 80, 80, 128, 128, 1, 1) == 14         # 1x1 over the twin: the whole-reduction kernel (conv_pw.hip, round 6), 128-cout blocks
    assert v(32, 80, 80, 192, 64, 1, 1) == 15          # ... 64-cout blocks
    assert v(32, 20, 20, 1024, 256, 1, 1) == 14        # the SPPF concat
    assert v(32, 80, 80, 128, 128, 1, 1, q_in=False) in (1, 2, 3)   # quantise-on-load: per-tap tiles
    assert v(32, 80, 80, 128, 128, 1, 1, res=True) in (1, 2, 3)     # residual
    assert v(32, 80, 80, 128, 128, 1, 1, acc=True) in (1, 2, 3)     # accumulator dump
    assert v(32, 80, 80, 96, 128, 1, 1) in (1, 2, 3)                # 96 input channels
    assert v(32, 80, 80, 128, 96, 1, 1) in (1, 2, 3)                # 96 couts: not whole 64-cout blocks
    assert v(32, 80, 80, 128, 128, 1, 1, out_cstride=132) in (1, 2, 3)
    # preconditions: each one alone sends the conv elsewhere
    base = (32, 80, 80, 128, 128, 3, 1)
    assert v(*base, q_in=False) in (4, 5)             # quantise-on-load: per-tap
    assert v(*base, res=True) in (8, 9)               # residual: the LDS-DMA kernels' general epilogue
    assert v(*base, acc=True) in (8, 9)               # accumulator dump
    assert v(*base, out_cstride=132) in (8, 9)        # fp16 view not 16-byte aligned per pixel
    assert v(*base, q_out_coff=2) in (8, 9)           # int8 twin not 4-byte aligned
    assert v(*base, out=False) == 10                  # twin only is fine
    assert v(32, 81, 81, 64, 128, 3, 2) in (1, 2, 3)  # odd map at stride 2
    assert v(32, 80, 80, 96, 128, 3, 1) in (8,)       # 96 input channels: not whole 64-channel stages
    d = desc(*base)
    d.conv.variant = 5
    assert hip_lib.y6_conv2d_i8_variant(C.byref(d)) == 5   # a forced variant is returned as is
    assert hip_lib.y6_conv2d_i8_variant(None) == 0


def _conv_desc_for_geometry(B, H, W, Cin, Cout, k, stride):
    d = _lib.ConvDesc()
    fake = 1 << 20      # 16-byte aligned, never dereferenced
    Ho, Wo = (H + 2 * (k // 2) - k) // stride + 1, (W + 2 * (k // 2) - k) // stride + 1
    d.inp = _lib.Tensor(fake, B, H, W, Cin, Cin, 0)
    d.out = _lib.Tensor(fake, B, Ho, Wo, Cout, Cout, 0)
    d.res = _lib.Tensor(None, 0, 0, 0, 0, 0, 0)
    d.w_packed, d.w_oihw = fake, fake
    d.ksize, d.stride, d.variant = k, stride, -1
    return d, Ho, Wo


def _check_geometry(hip_lib, shape):
    """Invariants of one conv shape's launch geometry under EVERY kernel variant that claims to support it."""
    import ctypes as C
    B, H, W, Cin, Cout, k, stride = shape
    d, Ho, Wo = _conv_desc_for_geometry(*shape)
    g = _lib.ConvGeometry()
    took = 0
    for i in range(1, hip_lib.y6_conv_variants()):
        if not hip_lib.y6_conv_variant_supports(C.byref(d), i):
            assert hip_lib.y6_conv_launch_geometry(C.byref(d), i, C.byref(g)) != 0
            continue
        name = hip_lib.y6_conv_variant_name(i).decode()
        tag = f"{name} on {shape}"
        assert hip_lib.y6_conv_launch_geometry(C.byref(d), i, C.byref(g)) == 0, tag
        took += 1
        assert g.lds_bytes <= g.lds_limit, f"{tag}: {g.lds_bytes} bytes of LDS"
        assert g.tile_h >= 1 and g.tile_w >= 1 and g.tile_h * g.tile_w <= g.block_pixels, tag
        if k == 1:       # a GEMM over flattened pixels: one image of one row
            npix = B * Ho * Wo
            assert g.tile_h == 1 and (g.tiles_x - 1) * g.tile_w < npix <= g.tiles_x * g.tile_w and g.tiles_y == 1, tag
            tiles = g.tiles_x
        else:
            assert (g.tiles_x - 1) * g.tile_w < Wo <= g.tiles_x * g.tile_w, tag
            assert (g.tiles_y - 1) * g.tile_h < Ho <= g.tiles_y * g.tile_h, tag
            assert g.halo_h == (g.tile_h - 1) * stride + k and g.halo_w == (g.tile_w - 1) * stride + k, tag
            tiles = B * g.tiles_x * g.tiles_y
        assert g.cout_blocks >= 1 and g.items >= tiles * g.cout_blocks, tag
        if g.cout_blocks > 1:
            assert g.items % (8 * g.cout_blocks) == 0, tag     # ids of one tile's cout blocks share id % 8 (XCD)
        if name.startswith("wreg"):
            assert 0 < g.halo_pieces <= g.halo_pieces_max, tag
            assert g.row_pitch >= g.halo_w, tag                                     # a halo row fits its LDS row
            assert g.halo_pieces * 64 >= 5 * g.halo_h * g.row_pitch, tag            # 5 16-byte slots per halo pixel
            assert g.lds_bytes >= 2 * g.halo_pieces * 1024, tag                      # two stage images
            assert Cin % 32 == 0 and Cout % 128 == 0, tag
    return took


def test_conv_launch_geometry_on_the_benchmarked_layers(hip_lib):
    """y6_conv_launch_geometry is host arithmetic (nothing launched): every 3x3 / 1x1 layer shape of YOLOv6-S at 640 x 640 b32 and
    YOLOv6-L6 at 1280 x 1280 b8, under every kernel variant that takes it."""
    shapes = []
    for B, size, chans in ((32, 640, (32, 64, 128, 256, 512)), (8, 1280, (64, 128, 256, 512, 768, 1024))):
        for lvl, c in enumerate(chans):
            hw = size // (2 << lvl)
            shapes += [(B, hw, hw, c, c, 3, 1), (B, hw, hw, c, c, 1, 1)]
            if lvl + 1 < len(chans):
                shapes.append((B, hw, hw, c, chans[lvl + 1], 3, 2))
    for sh in shapes:
        assert _check_geometry(hip_lib, sh) >= 1, sh
    # the register-fed kernels are the ones the headline runs on: whole rounds of the persistent walk on the large maps
    import ctypes as C
    d, _, _ = _conv_desc_for_geometry(32, 40, 40, 256, 256, 3, 1)
    g = _lib.ConvGeometry()
    p7 = [i for i in range(hip_lib.y6_conv_variants()) if hip_lib.y6_conv_variant_name(i) == b"wreg_p7"][0]
    assert hip_lib.y6_conv_launch_geometry(C.byref(d), p7, C.byref(g)) == 0
    assert g.tile_h * g.tile_w == 200 and g.items == 512      # 5 x 40 / 10 x 20 tiles: one item per resident block


def test_conv_launch_geometry_property(hip_lib):
    """Random conv shapes (odd maps, thin maps, one-pixel maps, ragged channel counts, large batches): whatever a variant claims to
    support, its tile must cover the map, fit its pixel slots and its LDS, and - for the register-fed kernels - its halo must fit
    the requests a stage can issue.  A violated invariant here is a memory fault or a refused launch on a user's shape."""
    from hypothesis import given, settings, strategies as st

    chan = st.sampled_from([8, 16, 24, 32, 48, 64, 96, 128, 192, 256, 320, 384, 512, 768, 1024])
    dim = st.one_of(st.integers(1, 40), st.sampled_from([64, 80, 96, 160, 320, 333, 640]))

    @settings(max_examples=250, deadline=None, derandomize=True)
    @given(B=st.sampled_from([1, 2, 3, 8, 32, 64]), H=dim, W=dim, Cin=chan, Cout=chan, k=st.sampled_from([1, 3]), stride=st.sampled_from([1, 2]))
    def run(B, H, W, Cin, Cout, k, stride):
        if k == 1 and stride == 2:
            return
        if B * H * W * max(Cin, Cout) >= 1 << 30:      # the library's tensor-size limits are tested elsewhere
            return
        _check_geometry(hip_lib, (B, H, W, Cin, Cout, k, stride))

    run()


def test_inflight_ticket_expires_when_its_slot_is_reused():
    """pipeline.Ticket.result(): rows are independent copies by default; a ticket whose slot a later submit() took raises instead
    of returning the later batch's rows (host logic only: a stand-in runner and event)."""
    from yolov6_amd.pipeline import Ticket

    class Ev:
        waited = 0

        def synchronize(self):
            Ev.waited += 1

    class Runner:
        generation = [3, 7]
        plans = [None, None]

    dets = torch.arange(2 * 4 * 6, dtype=torch.float32).reshape(2, 4, 6)
    count = torch.tensor([2, 0], dtype=torch.int32)
    t = Ticket(None, dets, None, count, Ev(), Runner, 1, 7)
    rows, counts = t.result()
    assert counts == [2, 0] and [tuple(r.shape) for r in rows] == [(2, 6), (0, 6)] and Ev.waited == 1
    dets[0, 0, 0] = -1.0
    assert float(rows[0][0, 0]) == 0.0                                  # a copy
    views, _ = t.result(copy=False)
    assert float(views[0][0, 0]) == -1.0                                # a view
    Runner.generation[1] = 8                                            # the slot was handed to a later batch
    with pytest.raises(RuntimeError, match="slot was reused"):
        t.result()
    assert Ticket(None, dets, None, count, Ev()).result()[1] == [2, 0]  # a bare ticket (no runner) has nothing to check


def test_wreg_kernels_keep_their_asm_loaded_registers():
    """conv_wreg.hip loads weight and pixel fragments by inline asm and awaits them by hand-counted s_waitcnt: hipcc does not know
    those registers are still in flight, so a spilled one (scratch store of a value that has not landed) is a WRONG RESULT, not a
    slowdown.  Every instantiation in the shipped library must be free of VGPR spills and scratch."""
    import os
    if not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-readelf"):
        import pytest
        pytest.skip("llvm-readelf not present")
    from yolov6_amd import _lib
    notes = _device_kernel_notes(_lib.LIB_PATH)
    wreg = {k: v for k, v in notes.items() if "conv3x3_wreg_kernel" in k}
    assert len(wreg) >= 4, sorted(notes)[:5]
    for name, meta in wreg.items():
        assert meta.get("vgpr_spill_count", -1) == 0, (name, meta)
        assert meta.get("private_segment_fixed_size", -1) == 0, (name, meta)


def _device_function_sizes(lib_path):
    """(mangled kernel name -> bytes of machine code) from the symbol tables of the gfx950 code objects embedded in the library."""
    import struct
    import subprocess
    import tempfile
    blob = open(lib_path, "rb").read()
    out = {}
    pos = blob.find(b"\x7fELF", 1)
    while pos >= 0:
        e_shoff, = struct.unpack_from("<Q", blob, pos + 0x28)
        e_shentsize, e_shnum = struct.unpack_from("<HH", blob, pos + 0x3A)
        e_machine, = struct.unpack_from("<H", blob, pos + 0x12)
        size = e_shoff + e_shentsize * e_shnum
        if e_machine == 224 and size > 0:
            with tempfile.NamedTemporaryFile(suffix=".co") as f:
                f.write(blob[pos:pos + size])
                f.flush()
                txt = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "-s", "--wide", f.name], capture_output=True, text=True).stdout
            for line in txt.splitlines():
                p = line.split()
                if len(p) >= 8 and p[3] == "FUNC":
                    out[p[7]] = int(p[2])
        pos = blob.find(b"\x7fELF", pos + 4)
    return out


def test_specialised_epilogue_kernels_stay_small():
    """code that runs once per work item is executed at instruction-fetch latency whenever the function's lines have
    left the 64 KB instruction cache (a conv launch lost 20-35 us to it after two or three other kernels).  The forms that compile one
    epilogue alone - conv3x3_wreg_kernel<..., EPI = 1 | 2 | 3> and the activation-specialised per-tap 1x1 kernels - are what keeps a
    step's kernel functions resident together: their machine code must stay a fraction of the general forms' (a change that inlines
    a second epilogue into them shows up here, on the CPU, not as 2 % on a GPU box)."""
    import os
    if not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-readelf"):
        pytest.skip("llvm-readelf not present")
    from yolov6_amd import _lib
    sizes = _device_function_sizes(_lib.LIB_PATH)
    wreg = {k: v for k, v in sizes.items() if "conv3x3_wreg_kernel" in k}
    special = {k: v for k, v in wreg.items() if k.endswith(("Lb0ELi1EEEvNS_9ConvKArgsE", "Lb0ELi2EEEvNS_9ConvKArgsE", "Lb0ELi3EEEvNS_9ConvKArgsE"))}
    general = {k: v for k, v in wreg.items() if k.endswith("Lb0ELi0EEEvNS_9ConvKArgsE")}
    assert len(special) >= 18 and len(general) >= 6, (len(special), len(general), sorted(wreg)[:3])
    assert max(special.values()) <= 32 * 1024, max(special.items(), key=lambda kv: kv[1])
    assert min(general.values()) >= 2 * max(special.values())          # (what they replace: 89-180 KB)
    one = {k: v for k, v in sizes.items() if "conv_mfma_kernelILi2ELi1ELi1ELi1E" in k}      # <2, 1, 1, 1, ACT>: the 1x1 c2p1 tile
    relu = [v for k, v in one.items() if "ELi1EEEvNS" in k[len("_ZN12_GLOBAL__N_116conv_mfma_kernelILi2ELi1ELi1ELi1"):]]
    gen = [v for k, v in one.items() if "ELin1EEEvNS" in k]
    assert relu and gen and relu[0] * 2 < gen[0], one


@pytest.mark.parametrize("case,flags", [("tiny", {}), ("s_qa_tiny", {}), ("s_mbla_tiny", {}), ("tiny", {"fuse_ab": True}), ("m_tiny", {})])
def test_backward_side_stream_contract_holds_on_every_training_graph(case, flags):
    """The backward plan runs its weight-gradient work (operand transposes, weight-gradient GEMMs, bias sums) on a side stream
    that is only ordered behind EARLIER ops and joins at the end of a run (csrc/plan.hip y6_plan_mark_side).  For every side op
    and every later main-stream op: the main op must not write what the side op reads nor touch what it writes
    (schedule.side_conflicts over TrainBuilder.bwd_log: gradient buffers, operand planes, workspaces, arena gradient slices).
    Every backward op kind must have an access list - an unknown kind would silently escape the check."""
    from yolov6_amd import schedule as S
    from yolov6_amd.engine import NCHWInput
    from yolov6_amd.train_engine import ParamArena, TrainBuilder
    cfg, meta = case_config(case)
    m = build_model(cfg, meta["num_classes"], "cpu", **flags).train()
    x = synth.synth_images(2, 64, seed=1)
    arena = ParamArena(m, x.device)
    tb = TrainBuilder(x.device, arena)
    m.lower_train(tb, NCHWInput(x))
    tb.finalize()
    log = tb.bwd_log
    assert len(log) > 50 and any(e["side"] for e in log) and any(not e["side"] for e in log)
    unknown = sorted({e["kind"] for e in log if S.train_bwd_access(e, arena) is None})
    assert not unknown, unknown
    assert S.side_conflicts(log, arena) == []
    # the checker does see a violation: a main-stream op that overwrites the gradient a pending transpose / GEMM still reads,
    # and one that works in place on what it wrote
    i = next(k for k, e in enumerate(log) if e["kind"] == "wgrad_transpose" and hasattr(e["src"], "buf"))
    rogue = dict(kind="tensor_add", side=False, x=log[i]["src"], out=log[i]["src"], acc=0)
    bad = S.side_conflicts(log[:i + 1] + [rogue], arena)
    # (round 6: the NHWC-fed weight gradients read the same activation without a transposed copy, so earlier side ops may be hit too)
    assert bad and any(b[0] == i and b[2] == "writes an input" for b in bad)
    j = next(k for k, e in enumerate(log) if e["kind"] == "wgrad")
    from yolov6_amd.engine import TRef
    g = arena.grad
    rogue2 = dict(kind="channel_sum", side=False, x=log[i]["src"], param=log[j]["weight"], ws=torch.zeros(16, dtype=torch.uint8))
    bad2 = S.side_conflicts(log[:j + 1] + [rogue2], arena)
    assert any(b[0] == j and b[2] == "touches an output" for b in bad2), bad2


def test_histogram_calibration_rules():
    """yolov6_amd.quant.HistogramCalibrator: the reference's PTQ recipe (tools/qat/qat_utils.py:12-58, histogram / entropy,
    configs/repopt/yolov6s_opt_qat.py:63-69) on known distributions: the histogram grows by whole bins of the first batch's
    width, percentile 100 is the range, a long thin tail is clipped by all three rules (below the abs-max) and a uniform
    distribution is not (entropy / mse keep ~ the whole range), the rules are ordered sensibly."""
    from yolov6_amd import quant as Q
    g = torch.Generator().manual_seed(0)
    c = Q.HistogramCalibrator()
    a = torch.randn(200000, generator=g)
    c.collect(a)
    w = c.edges[1] - c.edges[0]
    assert len(c.hist) == 2048 and abs(c.hist.sum() - 200000) < 1
    b = torch.cat([torch.randn(200000, generator=g), torch.tensor([37.5, -41.0])])     # two far outliers in the second batch
    c.collect(b)
    assert len(c.hist) > 2048 and abs((c.edges[1] - c.edges[0]) - w) < 1e-12 and c.edges[-1] >= 41.0
    assert abs(c.hist.sum() - 400002) < 1
    amax = float(torch.cat([a, b]).abs().max())
    p100 = c.compute_amax("percentile", 100.0)
    p9999 = c.compute_amax("percentile", 99.99)
    ent = c.compute_amax("entropy")
    mse = c.compute_amax("mse")
    assert p100 >= amax - w and 3.0 < p9999 < 5.0          # 99.99 % of |N(0,1)| lies below 3.9
    # (mse weighs the two outliers' squared clipping error against 400 000 values' rounding noise: it lands between the two)
    assert 2.0 < ent < 8.0 and ent < amax / 4 and ent < mse < amax - 5.0
    u = Q.HistogramCalibrator()
    u.collect(torch.rand(400000, generator=g) * 6.0)
    assert u.compute_amax("entropy") > 5.0 and u.compute_amax("mse") > 5.0 and u.compute_amax("percentile", 99.99) > 5.9
    with pytest.raises(ValueError):
        c.compute_amax("median")


def test_bench_supervisor_reruns_a_child_killed_by_a_signal_once(tmp_path):
    """bench.py measures in a child process; a child that dies of a signal (the HIP runtime aborts the process on a GPU memory
    fault) is re-run ONCE and the JSON line reports it; a second death is final (the exit status of a signal death, no line)."""
    import json
    import subprocess
    import sys
    bench = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py")
    env = dict(os.environ, Y6_BENCH_MOCK="1")
    ok = subprocess.run([sys.executable, bench, "--gpus", "1", "--steps", "2", "--warmup", "1"], env=env, capture_output=True, text=True, timeout=300)
    assert ok.returncode == 0, ok.stderr[-2000:]
    d = json.loads([ln for ln in ok.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["supervisor"]["attempts"] == 1 and d["supervisor"]["failed_attempts"] == []
    flag = str(tmp_path / "aborted_once")
    r = subprocess.run([sys.executable, bench, "--gpus", "1", "--steps", "2", "--warmup", "1"], env=dict(env, Y6_BENCH_TEST_ABORT=f"once:{flag}"),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["supervisor"]["attempts"] == 2 and d["supervisor"]["failed_attempts"] == [{"attempt": 1, "signal": 6}]
    assert "killed by signal 6" in r.stderr
    r = subprocess.run([sys.executable, bench, "--gpus", "1", "--steps", "2", "--warmup", "1"], env=dict(env, Y6_BENCH_TEST_ABORT="always"),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 134 and not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    # --no-supervisor: the measurement runs in the launched process itself
    r = subprocess.run([sys.executable, bench, "--gpus", "1", "--steps", "2", "--warmup", "1", "--no-supervisor"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "supervisor" not in json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])


def test_autotune_resolution_default_is_reproducible_and_env_reaches_every_plan(monkeypatch, tmp_path):
    """`model(x)` / compile() / new_plan() without an argument take their kernels from the layer shapes (the same bits in every
    process); Y6_AUTOTUNE=1 opts the default in, autotune=True asks explicitly, Y6_AUTOTUNE=0 forces shape-derived
    kernels for EVERY plan - new_plan() / InflightRunner included.  Host logic only (a CPU tensor stops compile())."""
    import inspect
    from yolov6_amd.layers import common
    from yolov6_amd.layers.common import HipModule, resolve_autotune
    monkeypatch.delenv("Y6_AUTOTUNE", raising=False)
    assert resolve_autotune(None) is False and resolve_autotune(False) is False and resolve_autotune(True) is True
    monkeypatch.setenv("Y6_AUTOTUNE", "1")
    assert resolve_autotune(None) is True and resolve_autotune(False) is False and resolve_autotune(True) is True
    monkeypatch.setenv("Y6_AUTOTUNE", "0")
    assert resolve_autotune(None) is False and resolve_autotune(True) is False
    src = inspect.getsource(HipModule.compile)
    assert src.index("resolve_autotune") < src.index("sig = ("), "the switch must be applied before the cache signature is formed"
    # every plan is lowered by _lower_plan: compile() and new_plan() both go through the resolution there
    assert "resolve_autotune" in inspect.getsource(HipModule._lower_plan)
    assert inspect.signature(HipModule.compile).parameters["autotune"].default is None
    assert inspect.signature(HipModule.new_plan).parameters["autotune"].default is None
    m = HipModule()
    with pytest.raises(RuntimeError, match="needs ROCm tensors"):
        m.compile(torch.zeros(1, 3, 8, 8))
    with pytest.raises(RuntimeError, match="needs ROCm tensors"):
        m.new_plan(torch.zeros(1, 3, 8, 8))
    # the on-disk table: a caller-chosen file wins, "" disables
    monkeypatch.setattr(common, "_cache_path_set", [False])
    monkeypatch.setenv("Y6_AUTOTUNE_CACHE", str(tmp_path / "t.txt"))
    assert common.autotune_cache_path() == str(tmp_path / "t.txt")
    monkeypatch.setattr(common, "_cache_path_set", [False])
    monkeypatch.setenv("Y6_AUTOTUNE_CACHE", "")
    assert common.autotune_cache_path() is None


def test_plan_is_current_and_inflight_guard_see_native_updates():
    """Native kernels (fused SGD, running-statistics update, EMA) bump a process-wide generation instead of `tensor._version`;
    `plan_is_current` and InflightRunner.submit's cheap check compare it."""
    import inspect
    from types import SimpleNamespace
    from yolov6_amd import pipeline
    from yolov6_amd.layers import common
    from yolov6_amd.layers.common import HipModule, _params_version
    m = HipModule()
    m.w = torch.nn.Parameter(torch.zeros(3))
    plan = SimpleNamespace(params_version=_params_version(m), quant_key=None,
                           generations=(common._NATIVE_GENERATION[0], common._STRUCTURE_GENERATION[0]))
    assert m.plan_is_current(plan)
    common.bump_native_generation()
    assert not m.plan_is_current(plan), "a native parameter update must make the plan stale"
    plan.generations = (common._NATIVE_GENERATION[0], common._STRUCTURE_GENERATION[0])
    assert m.plan_is_current(plan)
    m.invalidate_plans()
    assert not m.plan_is_current(plan), "invalidate_plans() / _apply must make the plan stale"
    src = inspect.getsource(pipeline.InflightRunner.submit)
    assert "_NATIVE_GENERATION" in src and "_STRUCTURE_GENERATION" in src and "_holders" in src


def test_result_ring_slots_count_views_in_every_grad_mode():
    """The result ring hands out a slot again only when nobody holds it or a view of it; inference tensors do not count views, so
    slots are allocated outside inference mode and every slot is probed itself."""
    from yolov6_amd.models.yolo import _ResultRing
    like = torch.empty(2, 4)
    for ctx in (torch.enable_grad, torch.no_grad, torch.inference_mode):
        with ctx():
            ring = _ResultRing(like, 2)
            assert all(not t.is_inference() and _ResultRing.counts_views(t) for t in ring.slots)
            a = ring.next()
            boxes = a[..., :2]
            ida = id(a)
            del a
            b = ring.next()
            c = ring.next()           # position of `a` again: still held through `boxes` -> a fresh tensor, never the old one
            assert id(c) != ida and c.data_ptr() != boxes.data_ptr()
            assert not c.is_inference()
    with torch.inference_mode():
        assert not _ResultRing.counts_views(torch.empty(3))      # what the per-slot probe exists for


def test_wgrad_flat_routing_and_geometry_properties():
    """The weight-gradient planning queries of the C ABI (host arithmetic, no GPU): which kernel a conv's weight gradient gets
    (y6_wgrad_nhwc_route: 1 flat-index kernel, 2 row ring, 0 neither = the plane-fed kernel stays) and the flat kernel's geometry
    (y6_wgrad_flat_geometry), over the YOLOv6 training shapes and random ones.  Invariants: the padded flat index covers the batch,
    the chunks tile it, the slices tile the chunks, the stages fit the LDS (160 KiB double-buffered; 80 KiB single-buffered at
    stride 2), the partial tiles fit the workspace."""
    import ctypes as C
    import random
    from yolov6_amd import _lib
    lib = _lib.load()

    def desc(K, cin, cout, B, H, W, stride=1, ws=256 << 20, dilated=False):
        d = _lib.WgradNhwcDesc()
        Ho, Wo = ((H - 1) // 2 + 1, (W - 1) // 2 + 1) if stride == 2 else (H, W)
        dh, dw = (H, W) if dilated else (Ho, Wo)
        d.ksize, d.M, d.N, d.stride = K, cout, cin, stride
        d.dy = _lib.Tensor(C.c_void_p(4096), B, dh, dw, cout, cout, 0)
        d.x = _lib.Tensor(C.c_void_p(8192), B, H, W, cin, cin, 0)
        d.out = 16384                       # (never dereferenced: host arithmetic only)
        d.sm, d.sn, d.st = cin * K * K, K * K, 1
        d.workspace, d.workspace_bytes = C.c_void_p(1 << 20), ws
        return d, Ho, Wo

    # the routing of the YOLOv6-S b64 layers (measured: profiles/r06/wgrad_bench_r06i.json)
    want = {(3, 256, 256, 40): 1, (3, 512, 512, 20): 1, (3, 128, 128, 80): 1, (1, 256, 256, 40): 1, (3, 64, 64, 160): 2, (3, 64, 64, 80): 2,
            (3, 128, 128, 40): 1, (1, 64, 64, 160): 2}
    for (K, cin, cout, hw), route in want.items():
        d, _, _ = desc(K, cin, cout, 64, hw, hw)
        assert lib.y6_wgrad_nhwc_route(C.byref(d)) == route, (K, cin, cout, hw)
    # stride 2: the flat kernel or nothing (the row ring has no stride-2 form); outputs wider than 48 keep the plane-fed kernel
    d, _, _ = desc(3, 128, 256, 64, 80, 80, stride=2, dilated=True)
    assert lib.y6_wgrad_nhwc_route(C.byref(d)) == 1
    d, _, _ = desc(3, 64, 128, 64, 160, 160, stride=2)
    assert lib.y6_wgrad_nhwc_route(C.byref(d)) == 0
    d, _, _ = desc(1, 64, 128, 64, 160, 160, stride=2)
    assert lib.y6_wgrad_nhwc_route(C.byref(d)) == 1

    rnd = random.Random(7)
    seen = 0
    for _ in range(400):
        K = rnd.choice((1, 3))
        stride = rnd.choice((1, 1, 2))
        cin, cout = 8 * rnd.randint(1, 64), 8 * rnd.randint(1, 64)
        B, H, W = rnd.randint(1, 64), 2 * rnd.randint(1, 60), 2 * rnd.randint(1, 60)
        ws = rnd.choice((4 << 20, 64 << 20, 256 << 20))
        d, Ho, Wo = desc(K, cin, cout, B, H, W, stride=stride, ws=ws, dilated=rnd.random() < 0.5)
        g = _lib.WgradFlatGeom()
        rc = lib.y6_wgrad_flat_geometry(C.byref(d), C.byref(g))
        if rc != 0:
            continue
        seen += 1
        T = K * K
        if K == 3:
            assert g.row_pitch == Wo + 1 and g.plane == (Ho + 1) * (Wo + 1)
        else:
            assert g.row_pitch == Wo and g.plane == Ho * Wo
        assert g.flat_positions == B * g.plane < (1 << 24)
        assert g.chunk == 128 and (g.chunks - 1) * g.chunk < g.flat_positions <= g.chunks * g.chunk
        assert g.tile_m in (64, 128) and g.tile_n in (32, 64) and g.tiles == -(-cout // g.tile_m) * -(-cin // g.tile_n)
        assert 1 <= g.slices and (g.slices - 1) * g.chunks_per_slice < g.chunks <= g.slices * g.chunks_per_slice
        assert g.slices * g.tiles <= 512 + g.tiles                      # one round of blocks over the chip
        assert g.partial_bytes == g.slices * T * cout * cin * 4 <= ws
        if K == 3 and stride == 2:
            assert g.stages == 1 and g.lds_bytes <= 80 * 1024 and Wo <= 48 and g.tile_n == 32
            assert g.x_positions >= g.chunk + g.row_pitch + 1
        else:
            assert g.stages == 2 and g.lds_bytes <= 160 * 1024 and g.tile_n == 64
            assert g.x_positions >= (g.chunk + 2 * g.row_pitch + 24 if K == 3 else g.chunk)
        assert g.x_positions % 16 == 0
    assert seen > 100


def test_torch_library_seam_registers_the_stateless_operators():
    """yolov6_amd/torch_ops.py (`TORCH_LIBRARY(yolov6_hip, m)`): the stateless operators are dispatcher-visible custom
    ops with schemas and fake kernels (shape inference without a GPU); there is no CPU kernel - a CPU tensor raises a RuntimeError
    (NotImplementedError is one), never a fallback."""
    import yolov6_amd.torch_ops  # noqa: F401
    from torch._subclasses.fake_tensor import FakeTensorMode
    ns = torch.ops.yolov6_hip
    for name in ("nms_batched", "tal_assign", "atss_assign", "conv2d_bias_act"):
        assert hasattr(ns, name), name
    assert "(Tensor, Tensor, Tensor)" in str(ns.nms_batched.default._schema)
    with FakeTensorMode():
        d, i, c = ns.nms_batched(torch.empty(2, 100, 85, device="cuda"), 0.03, 0.65, None, False, True, 300)
        assert d.shape == (2, 300, 6) and i.dtype == torch.int32 and c.shape == (2,)
        lab, box, sc, fg = ns.tal_assign(torch.empty(2, 50, 80, device="cuda"), torch.empty(2, 50, 4, device="cuda"), torch.empty(50, 2, device="cuda"),
                                         torch.empty(2, 7, 1, device="cuda"), torch.empty(2, 7, 4, device="cuda"), torch.empty(2, 7, 1, device="cuda"),
                                         13, 1.0, 6.0, 1e-9)
        assert lab.shape == (2, 50) and lab.dtype == torch.int64 and sc.shape == (2, 50, 80) and fg.dtype == torch.bool
        y = ns.conv2d_bias_act(torch.empty(2, 16, 20, 24, device="cuda", dtype=torch.float16), torch.empty(32, 16, 3, 3), None, "relu", 2)
        assert y.shape == (2, 32, 10, 12) and y.dtype == torch.float16
    with pytest.raises(RuntimeError):
        ns.nms_batched(torch.zeros(1, 10, 85), 0.03, 0.65, None, False, True, 300)
