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
