"""Run as a script by tests/test_gpu_families.py, one case per process:

    python tests/family_probe.py n6 | m6_tiny | tiny_distill_ns | t_pan | s_csp_pan_tiny | n6_pan | n_base | s_base_tiny | s_qav1_tiny | tiny_fuseab_eval

Whole-model parity of the model families added after the round's last GPU visit (EfficientRep6 + RepBiFPANNeck6, the M6 CSP
graph, the self-distillation head's eval branch, the v2.0 PAN necks): (a) every op of the plan teacher-forced against the fp16-emulating oracle within its
per-op bound (tests/gpu_utils.py::op_tolerance), (b) end to end within twice the reference's own fp16-vs-fp32 deviation on the
same input.  A separate process so that a device fault in one configuration cannot take the rest of the GPU suite with it;
summary + per-op table land in gpurun_out/families/."""
import copy
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from oracle import synth                                           # noqa: E402
from oracle.model_oracle import Oracle                             # noqa: E402
from tests.gpu_utils import op_tolerance                           # noqa: E402
from tests.helpers import GOLDEN, case_config, case_golden, rel_err, synth_sd_from_keys      # noqa: E402
from tests.plan_replay import OracleChain                          # noqa: E402
from yolov6_amd.configs import tiny_config                         # noqa: E402
from yolov6_amd.models.yolo import build_model                     # noqa: E402
from yolov6_amd.utils.torch_utils import fuse_model, switch_to_deploy                         # noqa: E402

DEV = "cuda:0"


def main(case):
    special = case in ("tiny_distill_ns", "tiny_fuseab_eval")
    if special:
        with open(os.path.join(GOLDEN, "keys_tiny_distill_ns.json" if case == "tiny_distill_ns" else "keys_tiny_fuseab.json")) as f:
            meta = json.load(f)
        cfg = tiny_config()
        m = build_model(cfg, meta["num_classes"], "cpu", distill_ns=case == "tiny_distill_ns", fuse_ab=case == "tiny_fuseab_eval").eval()
        gold = np.load(os.path.join(GOLDEN, f"model_{case}.npz"))["det_train"]
        ocfg = copy.deepcopy(cfg)
        ocfg.model.head.use_dfl = False      # both eval branches decode plain distances from reg_preds
    else:
        cfg, meta = case_config(case)
        m = build_model(cfg, meta["num_classes"], "cpu").eval()
        gold = case_golden(case)["det_deploy"]
        ocfg = cfg
    sd = synth_sd_from_keys(meta["train"])
    m.load_state_dict(sd)
    if not special:
        m.detect.proj_conv.weight.data = m.detect.proj.view(1, -1, 1, 1).clone()
    switch_to_deploy(fuse_model(m))
    m = m.to(DEV).half()
    x = synth.synth_images(meta["batch"], meta["size"], seed=1)
    xh = x.to(DEV).half()
    plan = m.compile(xh)
    det = plan.run().clone()
    feats = [r.to_nhwc_tensor().permute(0, 3, 1, 2).float().cpu().numpy() for r in m._featrefs]
    torch.cuda.synchronize()
    orc = Oracle(ocfg, sd, meta["num_classes"], emulate_fp16=True)
    with torch.no_grad():
        ref16, rfeats = orc.forward(x.half().float())
        ref32, rfeats32 = Oracle(ocfg, sd, meta["num_classes"], emulate_fp16=False).forward(x.half().float())
        # (a) every op of the plan alone, on the fp16-emulating oracle's activation of the previous layer (teacher-forced): the
        # parity statement proper - free-running fp16 pipelines of a deep random-weight network drift apart by amplified
        # rounding flips no matter who computes them (the reference's own half() model is that far from its fp32 result)
        rows = OracleChain(plan, orc).run(teacher_force=True)
    bad = []
    for r in rows:
        dsc = r["desc"]
        act = "silu" if " silu" in dsc else ("hardswish" if "hardswish" in dsc else "relu")
        r["tol"] = op_tolerance(act, with_res="+res" in dsc)
        if not (r["err"] <= r["tol"]):      # (NaN is a failure)
            bad.append(r)
    d = det.cpu().numpy()
    r16, r32 = ref16.numpy(), ref32.numpy()
    e_scores = float(np.abs(d[..., 5:] - r16[..., 5:]).max())
    floor_scores = float(np.abs(r16[..., 5:] - r32[..., 5:]).max())
    e_hip, e_ref16 = rel_err(d, gold), rel_err(r16, gold)
    e_feats = [rel_err(f, r.numpy()) for f, r in zip(feats, rfeats)]
    floor_feats = [rel_err(r.numpy(), r3.numpy()) for r, r3 in zip(rfeats, rfeats32)]
    worst = max(rows, key=lambda r: r["err"])
    rep = dict(case=case, ops=len(rows), per_op_max=worst["err"], per_op_worst=worst["desc"], per_op_above_tol=len(bad),
               e_scores=e_scores, floor_scores_ref16_vs_ref32=floor_scores, e_hip=e_hip, e_ref16=e_ref16,
               e_hip_vs_ref16=rel_err(d, r16), e_feats=e_feats, floor_feats_ref16_vs_ref32=floor_feats)
    print(json.dumps(rep))
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "families")
    os.makedirs(out_dir, exist_ok=True)
    with open(os.path.join(out_dir, f"family_{case}.json"), "w") as f:
        json.dump(dict(summary=rep, rows=rows), f, indent=1)
    assert d.shape == gold.shape
    assert not bad, f"{len(bad)} ops above their bound teacher-forced, e.g. {bad[:3]}"
    # (b) end to end, free-running, on the scale of the reference's own fp16-vs-fp32 deviation on this input
    assert e_scores <= 2.0 * floor_scores + 1e-3, rep
    assert e_hip <= 2.0 * e_ref16 + 1e-3, rep
    for e, fl in zip(e_feats, floor_feats):
        assert e <= 2.0 * fl + 5e-3, rep
    print("FAMILY_PROBE_OK")


if __name__ == "__main__":
    main(sys.argv[1])
