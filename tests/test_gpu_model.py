"""GPU: whole-model parity.  The HIP plan computes in fp16 storage / fp32 accumulation, like the
reference's `model.half()` on a GPU.  Per-op parity (tests/test_gpu_ops.py) holds the north_star's
1e-3.  End to end, two correct fp16 pipelines differ by accumulated fp16 rounding (one ulp per stored
activation, ~40 layers deep), so the whole-model bar is stated against the reference's OWN fp16
error, measured with the same metric rel = max|a-b| / max(1,|b|):
  * class scores:  |HIP - fp16-emulating oracle| <= 1e-3 absolute,
  * everything (boxes in pixels included):  err(HIP, fp32 golden from the reference) <=
    2 x err(fp16-emulating oracle, fp32 golden) + 1e-3  - i.e. the HIP path is as close to the
    reference's fp32 result as the reference's half-precision path is."""
import numpy as np
import pytest
import torch

from oracle import synth
from oracle.model_oracle import Oracle
from tests.helpers import case_config, case_golden, rel_err, synth_sd_from_keys
from yolov6_amd.layers import common
from yolov6_amd.models.yolo import build_model
from yolov6_amd.utils.torch_utils import fuse_model, switch_to_deploy

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
CASES = ["tiny", "n", "s", "s_qa_tiny", "l6_tiny", "m_tiny", "s_mbla_tiny"]


def _build(case, deploy, half=True):
    cfg, meta = case_config(case)
    m = build_model(cfg, meta["num_classes"], "cpu").eval()
    sd = synth_sd_from_keys(meta["train"])
    m.load_state_dict(sd)
    m.detect.proj_conv.weight.data = m.detect.proj.view(1, -1, 1, 1).clone()
    if deploy:
        switch_to_deploy(fuse_model(m))
    m = m.to(DEV)
    return cfg, meta, sd, (m.half() if half else m)


@pytest.mark.parametrize("case", CASES)
def test_model_vs_oracle_and_golden(case):
    cfg, meta, sd, m = _build(case, deploy=True)
    x = synth.synth_images(meta["batch"], meta["size"], seed=1)
    det, feats = m(x.to(DEV).half())
    torch.cuda.synchronize()
    assert det.dtype == torch.float32
    with torch.no_grad():
        ref16, rfeats = Oracle(cfg, sd, meta["num_classes"], emulate_fp16=True).forward(x.half().float())
    d = det.cpu().numpy()
    e_scores = float(np.abs(d[..., 5:] - ref16.numpy()[..., 5:]).max())
    assert e_scores < 1e-3, f"{case}: class scores HIP vs fp16-emulating oracle {e_scores:.3e}"
    assert np.array_equal(d[..., 4], np.ones_like(d[..., 4]))
    g = case_golden(case)
    e_hip = rel_err(d, g["det_deploy"])
    e_ref16 = rel_err(ref16.numpy(), g["det_deploy"])
    print(f"{case}: err(HIP,fp32 ref)={e_hip:.3e} err(fp16 ref,fp32 ref)={e_ref16:.3e} "
          f"err(HIP,fp16 ref)={rel_err(d, ref16.numpy()):.3e}")
    assert e_hip <= 2.0 * e_ref16 + 1e-3, f"{case}: HIP {e_hip:.3e} vs reference fp16 path {e_ref16:.3e}"
    feats = list(feats)
    assert len(feats) == len(rfeats)
    for f, r in zip(feats, rfeats):
        assert f.shape == r.shape
        assert rel_err(f.float().cpu().numpy(), r.numpy()) < 5e-3   # fp16 feature maps, ~30 layers deep


@pytest.mark.parametrize("case,size", [("n", 640), ("l6_tiny", 1280)])
def test_full_resolution_b1_vs_oracle_with_nms(case, size):
    """BASELINE configs[0] / configs[3] at their real resolutions (batch 1): YOLOv6-N at 640x640 and the P6
    graph (four levels, DFL head) at 1280x1280.  Full-size maps exercise ragged tiles on 160x160 ... 20x20 levels and
    every persistent-kernel tail; the detections then go through NMS with the inference thresholds of
    tools/infer.py (conf 0.4, IoU 0.45, max_det 1000) and must keep exactly the boxes the oracle NMS keeps."""
    from oracle import nms_oracle
    from yolov6_amd.utils.nms import non_max_suppression
    cfg, meta, sd, m = _build(case, deploy=True)
    x = synth.synth_images(1, size, seed=11)
    det, _ = m(x.to(DEV).half())
    torch.cuda.synchronize()
    with torch.no_grad():
        ref16, _ = Oracle(cfg, sd, meta["num_classes"], emulate_fp16=True).forward(x.half().float())
    d = det.cpu().numpy()
    r = ref16.numpy()
    assert d.shape == r.shape
    e_scores = float(np.abs(d[..., 5:] - r[..., 5:]).max())
    e_all = rel_err(d, r)
    print(f"{case}@{size}: scores {e_scores:.3e} all {e_all:.3e}")
    assert e_scores < 1e-3, f"{case}@{size}: class scores HIP vs fp16-emulating oracle {e_scores:.3e}"
    # boxes are pixels (up to 1280) through a DFL softmax on fp16 logits x stride 64: two correct fp16 pipelines differ by
    # a few 1e-2 in this metric (the reference's own half path is 7e-3 ... 4e-2 from its fp32 result on the golden
    # cases); measured here 3.3e-2 (P6 @1280) and below 1e-2 (N @640)
    px = float(np.abs(d[..., :4] - r[..., :4]).max())
    print(f"{case}@{size}: boxes deviate by at most {px:.3f} px")
    # bound = measured on MI355X (round 1: 3.3e-2 for the P6 graph, < 1e-2 for N) + 20 %
    bound = {"n": 1.2e-2, "l6_tiny": 4.0e-2}[case]
    assert e_all < bound, f"{case}@{size}: boxes (pixels) HIP vs fp16-emulating oracle {e_all:.3e}"
    # NMS on the HIP detections: device result == oracle NMS of the SAME tensor, bit for bit
    thr = float(np.quantile(d[0, :, 5:].max(-1), 0.97))     # random weights: take the top 3 % of anchors as candidates
    out = non_max_suppression(det, conf_thres=thr, iou_thres=0.45, max_det=1000)
    ref = nms_oracle.non_max_suppression(d, thr, 0.45, max_det=1000)
    assert out[0].shape[0] == ref[0].shape[0] > 0
    assert np.array_equal(out[0].cpu().numpy(), ref[0].astype(np.float32))


@pytest.mark.parametrize("case", ["tiny", "s_qa_tiny"])
def test_train_form_eval_equals_deploy(case):
    """Un-fused multi-branch modules in eval mode are re-parameterised at plan-build time."""
    # keep the parameters fp32 on both sides: the plan folds in fp32 and rounds the packed weights to
    # fp16 once, exactly like deploy-then-half() (a .half() train-form model rounds each branch first)
    cfg, meta, sd, m_dep = _build(case, deploy=True, half=False)
    _, _, _, m_train = _build(case, deploy=False, half=False)
    x = synth.synth_images(meta["batch"], meta["size"], seed=2).to(DEV).half()
    a, _ = m_dep(x)
    b, _ = m_train(x)
    # fold order differs by fp32 rounding before the fp16 pack: isolated fp16 flips of a regression distance (one ulp at
    # 8..16 is 2^-7) reach the box columns multiplied by the stride
    assert rel_err(a.cpu().numpy(), b.cpu().numpy()) < 1e-2


def test_rebind_and_repeat():
    """Model.forward hands out its detections without a copy (yolov6_amd/models/yolo.py `_ResultRing`: `output_buffers` result
    tensors per plan, the decode op re-pointed per call).  A result is never overwritten while the caller holds it (the
    reference returns independent tensors, yolo.py:37-47); once dropped, its memory is written again; `output_buffers = 0`
    restores clone-per-call."""
    cfg, meta, sd, m = _build("tiny", deploy=True)
    x1 = synth.synth_images(2, 64, seed=5).to(DEV).half()
    x2 = synth.synth_images(2, 64, seed=6).to(DEV).half()
    a1, _ = m(x1)
    c1 = a1.clone()
    a2, _ = m(x2)
    torch.cuda.synchronize()
    assert a1.data_ptr() != a2.data_ptr()
    assert torch.equal(a1, c1)           # the previous result survived the next call
    assert not torch.equal(a1, a2)
    c2 = a2.clone()
    a3, _ = m(x1.clone())
    assert torch.equal(a3, c1)           # deterministic, and the plan followed the new input pointer
    assert torch.equal(a2, c2) and torch.equal(a1, c1)
    assert a3.data_ptr() not in (a1.data_ptr(), a2.data_ptr())     # both earlier results are still held: neither is re-used
    held = [m(x)[0] for x in (x1, x2, x1, x2, x1)]                  # results collected over several batches (TTA, deferred NMS)
    torch.cuda.synchronize()
    assert len({t.data_ptr() for t in held}) == 5
    assert all(torch.equal(t, c) for t, c in zip(held, (c1, c2, c1, c2, c1)))
    view = m(x2)[0][:, :10]              # only a view survives: its base must not be written either
    for _ in range(4):
        m(x1)
    torch.cuda.synchronize()
    assert torch.equal(view, c2[:, :10])
    del a1, a2, a3, held, view
    ptrs = set()
    for _ in range(6):                   # the usual caller drops the result: two tensors alternate, nothing is allocated
        t = m(x1)[0]
        ptrs.add(t.data_ptr())
        del t
    assert len(ptrs) == 2
    m.output_buffers = 0
    b1, _ = m(x1)
    b2, _ = m(x2)
    b3, _ = m(x2)
    assert len({b1.data_ptr(), b2.data_ptr(), b3.data_ptr()}) == 3 and torch.equal(b1, c1) and torch.equal(b2, c2)


def test_graph_capture_matches_eager():
    cfg, meta, sd, m = _build("tiny", deploy=True)
    x = synth.synth_images(2, 64, seed=7).to(DEV).half()
    plan = m.compile(x)
    eager = plan.run().clone()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        plan.capture()
        out = plan.run()
    s.synchronize()
    assert torch.equal(out, eager)


@pytest.mark.parametrize("case", ["tiny"])
def test_multistream_graph_matches_eager(case, monkeypatch):
    """Y6_GRAPH_STREAMS=2 (experimental path): independent branches (head levels, cls/reg, neck laterals) are captured on two streams;
    the dependence analysis must keep every RAW/WAR/WAW edge - replays are bit-identical to the eager run."""
    cfg, meta, sd, m = _build(case, deploy=True)
    x = synth.synth_images(meta["batch"], meta["size"], seed=8).to(DEV).half()
    plan = m.compile(x)
    eager = plan.run().clone()
    monkeypatch.setenv("Y6_GRAPH_STREAMS", "2")
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        plan.capture()
        outs = [plan.run().clone() for _ in range(5)]
    s.synchronize()
    for o in outs:
        assert torch.equal(o, eager)


@pytest.mark.parametrize("policy", ["asap", "alap"])
@pytest.mark.parametrize("case", ["tiny", "s", "l6_tiny", "m_tiny"])
def test_two_stream_schedule_matches_one_stream(case, policy):
    """Plan.schedule() (yolov6_amd/schedule.py + y6_plan_set_schedule): the ops off the critical path run on the plan's side stream,
    ordered by events.  The same kernels on the same data: every result - detections and the neck feature maps - must be
    bit-identical to the one-stream run, also when runs are enqueued back to back on alternating inputs (a run's side
    stream must not overtake the previous run's readers, nor the next run's writers this run's)."""
    cfg, meta, sd, m = _build(case, deploy=True)
    xs = [synth.synth_images(meta["batch"], meta["size"], seed=20 + i).to(DEV).half() for i in range(3)]
    plan = m.compile(xs[0])
    plan.clear_schedule()                   # (compile() schedules by default since r03u)
    want = []
    for x in xs:
        plan = m.compile(x)
        want.append(plan.run().clone())
    torch.cuda.synchronize()
    info = plan.schedule(policy=policy)
    assert info is not None and info["side_ops"] and len(info["order"]) == plan.num_ops
    assert 0.0 < info["side_cost"] < 0.5 * info["total_cost"]
    got = []
    for rep in range(4):
        for x in xs:
            plan = m.compile(x)             # same plan, input rebound
            got.append(plan.run().clone())
    torch.cuda.synchronize()
    for i, g in enumerate(got):
        assert torch.equal(g, want[i % len(xs)]), f"{case}/{policy}: scheduled run {i} differs from the one-stream run"
    det, feats = m(xs[1])                   # the reference-signature API over the scheduled plan (lazy feature maps read after the join)
    torch.cuda.synchronize()
    assert torch.equal(det, want[1])
    plan.clear_schedule()
    assert torch.equal(m.compile(xs[2]).run(), want[2])


def test_single_block_in_train_mode_is_refused_not_faked():
    """Batch-statistics BatchNorm runs through the whole-model training graph (tests/test_gpu_training.py); a lone block
    in .train() mode has no standalone forward and must say so instead of silently using running statistics."""
    blk = common.RepVGGBlock(16, 16).to(DEV).half().train()
    with pytest.raises(NotImplementedError):
        blk(torch.zeros(1, 16, 8, 8, device=DEV, dtype=torch.float16))


def test_block_level_forward_matches_oracle():
    """Blocks keep the reference's NCHW forward contract on their own."""
    torch.manual_seed(0)
    blk = common.RepVGGBlock(32, 32).eval()
    sd = synth.synth_state_dict(blk.state_dict(), 3)
    blk.load_state_dict(sd)
    for mod in blk.modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.eps = 1e-3
    x = synth.synth_images(2, 24, seed=8, channels=32)
    # torch statement of common.py:250-255 on CPU
    import torch.nn.functional as F

    def bn(t, p):
        s = sd[p + ".weight"] / torch.sqrt(sd[p + ".running_var"] + 1e-3)
        return t * s.view(1, -1, 1, 1) + (sd[p + ".bias"] - sd[p + ".running_mean"] * s).view(1, -1, 1, 1)
    xh = x.half().float()
    ref = F.relu(bn(F.conv2d(xh, sd["rbr_dense.conv.weight"], padding=1), "rbr_dense.bn") +
                 bn(F.conv2d(xh, sd["rbr_1x1.conv.weight"]), "rbr_1x1.bn") + bn(xh, "rbr_identity"))
    out = blk.to(DEV).half()(x.to(DEV).half())
    assert out.shape == ref.shape and out.dtype == torch.float16
    assert rel_err(out.float().cpu().numpy(), ref.numpy()) < 3e-3   # unfused fp32 vs fused fp16 weights


def test_uint8_images_equal_half_div255(monkeypatch):
    """Input boundary: uint8 pixels fed straight to the stem == `imgs.half(); imgs /= 255` (core/evaler.py:121-123)
    followed by the fp16 path, bit for bit - the conversion pass and the fp16 copy of the batch disappear.  (Two plans are compared:
    both take their kernels from the layer shapes, Y6_AUTOTUNE=0 - two separately TIMED plans may pick different kernels, i.e.
    different fp32 summation orders, and then differ in the low bits for a reason that has nothing to do with the input path.)"""
    monkeypatch.setenv("Y6_AUTOTUNE", "0")
    cfg, meta, sd, m = _build("tiny", deploy=True)
    g = torch.Generator().manual_seed(4)
    u8 = torch.randint(0, 256, (2, 3, 64, 64), generator=g, dtype=torch.uint8).to(DEV)
    ref, _ = m(u8.half() / 255)
    got, _ = m(u8)
    assert torch.equal(got, ref)
    # the non-vectorised stem kernels too (width not a multiple of 8 / unaligned rows)
    from yolov6_amd.engine import NCHWInput, PlanBuilder
    w = torch.randn((16, 3, 3, 3), generator=g) * 0.2
    b = torch.randn((16,), generator=g) * 0.1
    u8b = torch.randint(0, 256, (2, 3, 18, 22), generator=g, dtype=torch.uint8).to(DEV)
    outs = []
    for x in (u8b, (u8b.half() / 255).contiguous()):
        pb = PlanBuilder(DEV)
        o = pb.conv(NCHWInput(x), w, b, stride=2, act="relu")
        pb.finalize(o, autotune=False).run()
        torch.cuda.synchronize()
        outs.append(o.to_nhwc_tensor().clone())
    assert torch.equal(outs[0], outs[1])


def test_replaced_parameter_is_noticed_without_invalidate_plans(monkeypatch):
    """`m.bias = nn.Parameter(...)` (what the reference's re-parameterisation and many user scripts do) registers a NEW tensor:
    the version counters of the old ones do not move, so the per-call fast path of HipModule.compile must also check that every
    captured parameter / buffer is still the registered object.  In-place edits (`copy_`, optimizers) were
    already seen through the version counters.  (The last comparison is between two separately built plans: shape-derived kernels,
    Y6_AUTOTUNE=0, so that equal weights mean equal bits - under the guard allocator, whose every free synchronises the device, two
    timed plans of this model picked different kernels, round 5.)"""
    monkeypatch.setenv("Y6_AUTOTUNE", "0")
    cfg, meta, sd, m = _build("tiny", deploy=True)
    x = synth.synth_images(2, 64, seed=5).to(DEV).half()
    a = m(x)[0].clone()
    b = m(x)[0].clone()
    assert torch.equal(a, b)
    conv = m.detect.cls_preds[0]
    conv.bias = torch.nn.Parameter(conv.bias.detach() + 2.0)           # a different Parameter object
    c = m(x)[0].clone()
    assert not torch.equal(a[..., 5:], c[..., 5:]), "the plan kept running on the replaced parameter"
    with torch.no_grad():
        conv.bias.sub_(2.0)                                             # in place: the version counter moves
    d = m(x)[0].clone()
    assert torch.equal(a, d)
