#!/usr/bin/env python3
"""Build libyolov6_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

    python yolov6_amd/csrc/build.py [--force] [--jobs N]

Objects are cached under yolov6_amd/csrc/build/ keyed on source + flag hashes; the shared
library lands in yolov6_amd/lib/ (git-ignored, travels to the GPU box with the snapshot).
"""
import argparse
import concurrent.futures as cf
import hashlib
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
OUT_DIR = os.path.join(os.path.dirname(HERE), "lib")
OBJ_DIR = os.path.join(HERE, "build")
LIB = os.path.join(OUT_DIR, "libyolov6_hip.so")

COMMON = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
          "-Wno-unused-variable", "-I" + os.path.join(ROOT, "include")]
# per-file extra flags: index parity needs the reference's unfused fp32 arithmetic
SOURCES = {
    "conv_mfma.hip": [],
    "conv_dma.hip": [],
    "conv_wreg.hip": [],
    "conv_pw.hip": [],
    "conv_misc.hip": [],
    "conv_fused.hip": [],
    "preproc.hip": ["-ffp-contract=off"],
    "head_decode.hip": [],
    "nms.hip": ["-ffp-contract=off"],
    "tal.hip": ["-ffp-contract=off"],
    "loss.hip": ["-ffp-contract=off"],
    "train.hip": [],
    "wgrad.hip": [],
    "wgrad_flat.hip": [],
    "wgrad_stem.hip": [],
    "dgrad_s2.hip": [],
    "quant.hip": [],
    "plan.hip": [],
}
HEADERS = ["common.hpp", "conv_common.hpp", "plan_internal.hpp", "stem_piece.hpp", "nms_cand.hpp", os.path.join(ROOT, "include", "yolov6_hip.h")]


def hipcc():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (set HIPCC)")


def digest(paths, extra):
    h = hashlib.sha256()
    for p in paths:
        with open(p if os.path.isabs(p) else os.path.join(HERE, p), "rb") as f:
            h.update(f.read())
    h.update(repr(extra).encode())
    return h.hexdigest()[:16]


def compile_one(cc, src, flags, force):
    tag = digest([src] + HEADERS, COMMON + flags)
    obj = os.path.join(OBJ_DIR, f"{os.path.splitext(src)[0]}.{tag}.o")
    if os.path.exists(obj) and not force:
        return obj, False
    cmd = [cc] + COMMON + flags + ["-c", os.path.join(HERE, src), "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{' '.join(cmd)}\n{r.stdout}\n{r.stderr}")
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
    return obj, True


def build(force=False, jobs=None, verbose=True):
    os.makedirs(OBJ_DIR, exist_ok=True)
    os.makedirs(OUT_DIR, exist_ok=True)
    cc = hipcc()
    jobs = jobs or min(len(SOURCES), os.cpu_count() or 4)
    objs, rebuilt = [], False
    with cf.ThreadPoolExecutor(max_workers=jobs) as ex:
        futs = {ex.submit(compile_one, cc, s, f, force): s for s, f in SOURCES.items()}
        for fut in cf.as_completed(futs):
            obj, did = fut.result()
            objs.append(obj)
            rebuilt |= did
            if verbose and did:
                print(f"[build] compiled {futs[fut]}")
    if rebuilt or force or not os.path.exists(LIB):
        cmd = [cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + sorted(objs)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{' '.join(cmd)}\n{r.stdout}\n{r.stderr}")
        if verbose:
            print(f"[build] linked {LIB}")
    # drop stale objects
    keep = set(objs)
    for f in os.listdir(OBJ_DIR):
        p = os.path.join(OBJ_DIR, f)
        if p.endswith(".o") and p not in keep:
            os.remove(p)
    return LIB


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--force", action="store_true")
    ap.add_argument("--jobs", type=int, default=None)
    a = ap.parse_args()
    print(build(force=a.force, jobs=a.jobs))
