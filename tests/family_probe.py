"""Run as a script by tests/test_gpu_families.py, one case per process:

    python tests/family_probe.py n6 | m6_tiny | tiny_distill_ns | t_pan | s_csp_pan_tiny | n6_pan | n_base | s_base_tiny | s_qav1_tiny | tiny_fuseab_eval

Whole-model parity of the model families added after the round's last GPU visit (EfficientRep6 + RepBiFPANNeck6, the M6 CSP
graph, the self-distillation head's eval branch, the v2.0 PAN necks): same bar as tests/test_gpu_model.py::test_model_vs_oracle_and_golden.  A
separate process so that a device fault in a not-yet-seen configuration cannot take the rest of the GPU suite with it."""
import copy
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from oracle import synth                                           # noqa: E402
from oracle.model_oracle import Oracle                             # noqa: E402
from tests.helpers import GOLDEN, case_config, case_golden, rel_err, synth_sd_from_keys      # noqa: E402
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
    det, feats = m(x.to(DEV).half())
    torch.cuda.synchronize()
    with torch.no_grad():
        ref16, rfeats = Oracle(ocfg, sd, meta["num_classes"], emulate_fp16=True).forward(x.half().float())
    d = det.cpu().numpy()
    e_scores = float(np.abs(d[..., 5:] - ref16.numpy()[..., 5:]).max())
    e_hip, e_ref16 = rel_err(d, gold), rel_err(ref16.numpy(), gold)
    print(json.dumps(dict(case=case, e_scores=e_scores, e_hip=e_hip, e_ref16=e_ref16, e_hip_vs_ref16=rel_err(d, ref16.numpy()))))
    assert d.shape == gold.shape
    assert e_scores < 1e-3
    assert e_hip <= 2.0 * e_ref16 + 1e-3
    for f, r in zip(list(feats), rfeats):
        assert rel_err(f.float().cpu().numpy(), r.numpy()) < 5e-3
    print("FAMILY_PROBE_OK")


if __name__ == "__main__":
    main(sys.argv[1])
