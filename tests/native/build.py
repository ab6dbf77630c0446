#!/usr/bin/env python3
"""Build tests/native/libguard_alloc.so (the guard-page device allocator of the tight-allocation GPU tests) with hipcc.
Host code only: it links the HIP runtime, no device kernels.  Test infrastructure, never loaded by the product path."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "libguard_alloc.so")
SRC = os.path.join(HERE, "guard_alloc.cpp")


def build(verbose=True):
    if os.path.exists(LIB) and os.path.getmtime(LIB) >= os.path.getmtime(SRC):
        return LIB
    cc = os.environ.get("HIPCC") or shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    cmd = [cc, "-O1", "-std=c++17", "-fPIC", "-shared", "-x", "hip", "--offload-arch=gfx950", SRC, "-o", LIB]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"guard_alloc build failed:\n{' '.join(cmd)}\n{r.stdout}\n{r.stderr}")
    if verbose:
        print(f"[build] linked {LIB}")
    return LIB


if __name__ == "__main__":
    print(build())
