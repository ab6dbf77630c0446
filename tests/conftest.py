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


def pytest_unconfigure(config):
    # proof for the parent test that this process really allocated through the guard allocator (its own stderr banner is swallowed
    # by pytest's capture): "<mode> <allocations served>" into the file named by Y6_GUARD_ALLOC_MARK
    mode, mark = os.environ.get("Y6_GUARD_ALLOC"), os.environ.get("Y6_GUARD_ALLOC_MARK")
    if mode and mark:
        import ctypes
        from tests.native import build as gb
        lib = ctypes.CDLL(gb.LIB)              # the library torch already loaded: the same counters
        lib.guard_total_blocks.restype = ctypes.c_long
        with open(mark, "w") as f:
            f.write(f"{mode} {int(lib.guard_total_blocks())}\n")


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
