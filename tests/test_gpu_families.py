"""GPU: model families whose host side and oracle were pinned on the CPU (tests/test_host_cpu.py, tests/test_oracle_cpu.py:
cases n6, m6_tiny, tiny_distill_ns, t_pan, s_csp_pan_tiny, n6_pan, n_base, s_base_tiny, s_qav1_tiny) AFTER this round's last GPU visit.  Their lowerings only compose ops the GPU suite already
covers (the L6 wiring with RepBlock stages; the N / S head's non-DFL decode), but the combination has not been seen on
hardware: each case runs in its own process (tests/family_probe.py) and is reported as xfail/xpass, not as a hard failure,
until a GPU visit has seen it green."""
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


@pytest.mark.xfail(strict=False, reason="added after the round's last GPU visit; promote to a hard test once seen green")
@pytest.mark.parametrize("case", ["n6", "m6_tiny", "tiny_distill_ns", "t_pan", "s_csp_pan_tiny", "n6_pan",
                                  "n_base", "s_base_tiny", "s_qav1_tiny", "tiny_fuseab_eval"])
def test_new_family_in_subprocess(case):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "family_probe.py"), case], cwd=ROOT, capture_output=True,
                       text=True, timeout=300)
    _save_log(f"family_{case}.log", r)
    print(r.stdout[-2000:], r.stderr[-2000:])
    assert r.returncode == 0 and "FAMILY_PROBE_OK" in r.stdout


@pytest.mark.xfail(strict=False, reason="kernel variants written after the round's last GPU visit (tests/gpu_utils.py::UNSEEN_VARIANTS)")
def test_unseen_conv_variants_in_subprocess():
    """The conv-op tests once more with the not-yet-measured variants included (dma8_c4p1: 128 couts x 256 pixels on eight waves; dmar8 / dmarw8_c2p2: tap images resident in LDS; dma_c2p4: 512-pixel blocks on four waves; dma8s2_c4p1: stride 2, 128 couts)."""
    env = dict(os.environ, Y6_TEST_UNSEEN="1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_ops.py"), "-x", "-q", "-m", "gpu", "-p",
                        "no:cacheprovider", "-k", "conv_all_variants or conv_dma or tap_geometry or epilogue_variants or not_transposed"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    _save_log("family_unseen_variants.log", r)
    print(r.stdout[-3000:], r.stderr[-1500:])
    assert r.returncode == 0
