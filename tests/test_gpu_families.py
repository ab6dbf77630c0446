"""GPU: whole-model parity of every model family beyond the five golden cases of tests/test_gpu_model.py (EfficientRep6 +
RepBiFPANNeck6, the M6 CSP graph, the self-distillation / fuse_ab heads' eval branches, the v2.0 PAN necks, `conv_relu` base
models, QARepVGG v1), one process per case (tests/family_probe.py): every op teacher-forced against the fp16-emulating oracle
within its per-op bound, end to end within twice the reference's own fp16-vs-fp32 deviation.  Hard tests since round 3 (in
round 2 three of them - m6_tiny, s_csp_pan_tiny, s_base_tiny - missed a flat 1e-3 end-to-end bar that the reference's own
`model.half()` misses by the same amount on these deep random-weight cases: tests/golden/model_*_half.npz)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _save_log(name, r):
    """stdout / stderr of the probe process under gpurun_out/families/ (merged back from the GPU box; tools/gpu_round.sh copies
    them into the round's profiles/ directory), so that a failure can be told apart: device fault, tolerance miss, host error."""
    d = os.path.join(ROOT, "gpurun_out", "families")
    os.makedirs(d, exist_ok=True)
    with open(os.path.join(d, name), "w") as f:
        f.write(f"returncode {r.returncode}\n---- stdout ----\n{r.stdout}\n---- stderr ----\n{r.stderr}\n")


@pytest.mark.parametrize("case", ["n6", "m6_tiny", "tiny_distill_ns", "t_pan", "s_csp_pan_tiny", "n6_pan",
                                  "n_base", "s_base_tiny", "s_qav1_tiny", "tiny_fuseab_eval"])
def test_new_family_in_subprocess(case):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "family_probe.py"), case], cwd=ROOT, capture_output=True,
                       text=True, timeout=300)
    _save_log(f"family_{case}.log", r)
    print(r.stdout[-2000:], r.stderr[-2000:])
    assert r.returncode == 0 and "FAMILY_PROBE_OK" in r.stdout
