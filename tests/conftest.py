import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # Y6_GUARD_ALLOC=end|start: every torch device allocation of this pytest process sits flush against an unmapped guard range
    # (tests/native/guard_alloc.cpp) - an out-of-bounds access of any kernel faults.  tests/test_gpu_tight_alloc.py runs the op
    # suites this way in a child process; must happen before the first device allocation.
    mode = os.environ.get("Y6_GUARD_ALLOC")
    if mode:
        from tests.tight_probe import install
        install(mode)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def hip_lib():
    """The built shared library (built on demand; hipcc cross-compiles without a GPU)."""
    import __graft_entry__ as ge
    ge.build()
    from yolov6_amd import _lib
    return _lib.load()
