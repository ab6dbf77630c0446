"""GPU: a checkpoint pickled by the UNMODIFIED reference (tests/golden/ckpt_tiny_reference.pt, written by
`Y6_PROBE_NO_EMA=1 python tests/dropin_probe.py write /root/reference <path>` in the build container) loads through the
`yolov6.*` import paths onto the HIP path and reproduces the reference's own deploy-form output (tests/golden/model_tiny.npz,
same synthetic weights and input)."""
import os
import sys

import numpy as np
import pytest
import torch

from oracle import synth
from tests.helpers import GOLDEN, case_golden, rel_err

pytestmark = pytest.mark.gpu


def test_reference_checkpoint_runs_on_hip_path():
    import yolov6_amd
    saved = {k: v for k, v in sys.modules.items() if k == "yolov6" or k.startswith("yolov6.")}
    try:
        for k in saved:
            del sys.modules[k]
        yolov6_amd.install_as_yolov6()
        from yolov6.layers.common import DetectBackend, RepVGGBlock
        from yolov6.utils.nms import non_max_suppression
        be = DetectBackend(os.path.join(GOLDEN, "ckpt_tiny_reference.pt"), device=torch.device("cuda:0"))
        model = be.model
        assert type(model).__module__ == "yolov6_amd.models.yolo"
        for layer in model.modules():                      # core/evaler.py:70-73
            if isinstance(layer, RepVGGBlock):
                layer.switch_to_deploy()
        model.half()
        x = synth.synth_images(2, 64, seed=1).to("cuda:0").half()
        det = be(x)
        torch.cuda.synchronize()
        g = case_golden("tiny")
        # the checkpoint stores fp16 parameters (engine.py:192 `.half()`), folded after rounding: same bar as
        # tests/test_gpu_model.py (reference fp16 path vs its fp32 result is 7e-3 ... 4e-2 on these cases)
        e = rel_err(det.cpu().numpy(), g["det_deploy"])
        print(f"reference checkpoint on the HIP path: err vs reference fp32 deploy output {e:.3e}")
        assert float(np.abs(det.cpu().numpy()[..., 5:] - g["det_deploy"][..., 5:]).max()) < 3e-3
        assert e < 5e-2
        out = non_max_suppression(det, 0.03, 0.65, multi_label=True, max_det=300)
        assert len(out) == 2 and out[0].shape[1] == 6
    finally:
        for k in [k for k in sys.modules if k == "yolov6" or k.startswith("yolov6.")]:
            del sys.modules[k]
        sys.modules.update(saved)


@pytest.mark.gpu
def test_torch_library_ops_equal_the_module_mirrors_and_trace_under_torch_compile():
    """`torch.ops.yolov6_hip.*` (yolov6_amd/torch_ops.py) call the same kernels as the module mirrors: bit-equal results; a function
    that calls them is traceable by torch.compile (the op stays an opaque call; backend aot_eager: no code generation involved)."""
    import numpy as np
    import torch.nn.functional as F
    import yolov6_amd.torch_ops  # noqa: F401
    from oracle import synth
    from yolov6_amd.assigners import ATSSAssigner, TaskAlignedAssigner, generate_anchors
    from yolov6_amd.utils.nms import nms_raw
    from yolov6_amd.utils.synth import synth_predictions
    dev = "cuda:0"
    ns = torch.ops.yolov6_hip
    pred = synth_predictions(3, 2100, 80, seed=3).to(dev)
    want = nms_raw(pred, 0.03, 0.65, None, False, True, 300)
    got = ns.nms_batched(pred, 0.03, 0.65, None, False, True, 300)
    torch.cuda.synchronize()
    assert all(torch.equal(a, b) for a, b in zip(got, want))

    def f(p):
        d, i, c = ns.nms_batched(p * 1.0, 0.03, 0.65, None, False, True, 300)
        return d.sum(-1), c
    cf = torch.compile(f, backend="aot_eager", fullgraph=True)
    a, b = cf(pred), f(pred)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])

    fs, st = [(20, 20), (10, 10), (5, 5)], [8, 16, 32]
    inp = synth.synth_tal_inputs(2, fs, st, 80, 9, seed=5, n_valid=[9, 4], img=160)
    keys = ("pd_scores", "pd_bboxes", "anc_points", "gt_labels", "gt_bboxes", "mask_gt")
    want = TaskAlignedAssigner(13, 80, 1.0, 6.0)(*(inp[k].to(dev) for k in keys))
    got = ns.tal_assign(*(inp[k].to(dev) for k in keys), 13, 1.0, 6.0, 1e-9)
    assert all(torch.equal(a, b) for a, b in zip(got, want))
    anchors, _, n_list, _ = generate_anchors([torch.zeros(1, 1, h, w) for h, w in fs], st, 5.0, 0.5, device="cpu", is_eval=False)
    args = (anchors.to(dev), n_list, inp["gt_labels"].to(dev), inp["gt_bboxes"].to(dev), inp["mask_gt"].to(dev), inp["pd_bboxes"].to(dev))
    want = ATSSAssigner(9, 80)(*args)
    got = ns.atss_assign(*args, 9, 80)
    assert all(torch.equal(a, b) for a, b in zip(got, want))

    g = torch.Generator().manual_seed(1)
    x = (torch.rand((2, 32, 24, 40), generator=g) - 0.5).half()
    w = (torch.rand((64, 32, 3, 3), generator=g) - 0.5) * 0.2
    bias = torch.rand((64,), generator=g) - 0.5
    y = ns.conv2d_bias_act(x.to(dev), w.to(dev), bias.to(dev), "relu", 1)
    ref = F.relu(F.conv2d(x.float(), w.half().float(), bias.half().float(), padding=1))
    err = float((y.float().cpu() - ref).abs().max()) / max(1.0, float(ref.abs().max()))
    assert y.shape == ref.shape and err <= 2e-3, err


def test_nms_takes_the_forwards_candidates_only_when_they_are_this_results():
    """The drop-in `non_max_suppression(model(x)[0], ...)` (evaler.py:128-132) arms the plan's fused head tail with the thresholds of
    the first call and takes the candidates later forwards select (utils/nms.py `_speculated_candidates`) - but only for the very
    tensor the LAST forward returned, unmodified, with the same thresholds.  Every other case must take the full path, and both
    paths return the same detections bit for bit."""
    from tests.test_gpu_model import _build
    from yolov6_amd.utils import nms as N
    cfg, meta, sd, m = _build("tiny", deploy=True)
    with torch.no_grad():
        m.detect.cls_preds[0].bias.add_(3.0)           # enough candidates above the threshold to make the comparison mean something
    m = m.eval()
    x1 = synth.synth_images(meta["batch"], meta["size"], seed=1).to("cuda:0").half()
    x2 = synth.synth_images(meta["batch"], meta["size"], seed=2).to("cuda:0").half()
    kw = dict(conf_thres=0.03, iou_thres=0.65, multi_label=True, max_det=300)

    def full(det, **k):      # the full path: a clone is not the tensor the forward returned
        return N.non_max_suppression(det.clone(), **(k or kw))

    def same(a, b):
        return len(a) == len(b) and all(torch.equal(u, v) for u, v in zip(a, b))

    taken = []
    orig = N._speculated_candidates

    def spy(*a, **k):
        t = orig(*a, **k)
        taken.append(t is not None)
        return t
    N._speculated_candidates = spy
    try:
        d = m(x1)[0]
        first = N.non_max_suppression(d, **kw)                       # first call: full path, arms the plan
        assert taken[-1] is False and sum(len(t) for t in first) > 0
        d = m(x1)[0]
        fast = N.non_max_suppression(d, **kw)                        # now the forward selected the candidates
        assert taken[-1] is True and same(fast, first) and same(fast, full(d))
        again = N.non_max_suppression(d, **kw)                       # the workspace was consumed: a second call re-reads the tensor
        assert taken[-1] is False and same(again, first)
        # an older result: the plan's workspace holds the candidates of the LATER run
        d1 = m(x1)[0]
        d2 = m(x2)[0]
        r1 = N.non_max_suppression(d1, **kw)
        assert taken[-1] is False and same(r1, first)
        r2 = N.non_max_suppression(d2, **kw)
        assert taken[-1] is True and same(r2, full(d2)) and not same(r2, first)
        # written in place after the forward
        d = m(x1)[0]
        d[..., 5:] *= 0.5
        r = N.non_max_suppression(d, **kw)
        assert taken[-1] is False and same(r, full(d))
        # other thresholds: full path (and the plan is re-armed with them)
        d = m(x1)[0]
        k2 = dict(conf_thres=0.25, iou_thres=0.45, multi_label=False, max_det=100)
        r = N.non_max_suppression(d, **k2)
        assert taken[-1] is False and same(r, full(d, **k2))
        d = m(x2)[0]
        r = N.non_max_suppression(d, **k2)
        assert taken[-1] is True and same(r, full(d, **k2))
        # the plan launched again behind the forward's back (the plan API on the same plan): its workspace is no longer this result's
        d = m(x2)[0]
        m.compile(x2).run()
        torch.cuda.synchronize()
        r = N.non_max_suppression(d, **k2)
        assert taken[-1] is False and same(r, full(d, **k2))
        # a view / a clone is never matched
        d = m(x1)[0]
        r = N.non_max_suppression(d[:], **k2)
        assert taken[-1] is False and same(r, full(d, **k2))
    finally:
        N._speculated_candidates = orig
