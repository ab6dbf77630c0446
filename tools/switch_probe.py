#!/usr/bin/env python3
"""Does the first launch after a kernel switch cost extra?  One plan with a fixed chain of 3x3 convs on a
128-channel 80x80 b32 tensor (every op reads the previous op's output); op i uses the kernel variant given by a
pattern string.  Per-op hipEvent times (mean of 20 passes) are printed next to the same chain with one kernel only.

    python tools/switch_probe.py            # default patterns
"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from yolov6_amd import _lib
from yolov6_amd.engine import PlanBuilder, TRef

lib = _lib.load()
names = [lib.y6_conv_variant_name(i).decode() for i in range(lib.y6_conv_variants())]
dev = "cuda:0"
B, H, W, C = 32, 80, 80, 128
A, Bv, Cv = names.index("pipe_c2p2"), names.index("mfma_c2p1"), names.index("mfma_c4p1")


def run(pattern, passes=20):
    x = torch.randn((B, H, W, C), device=dev).half()
    w = torch.randn((C, C, 3, 3)) / (C * 9) ** 0.5
    b = torch.zeros((C,))
    pb = PlanBuilder(dev)
    t = TRef(x, B, H, W, C, C, 0)
    for v in pattern:
        pb.force_variant = v
        t = pb.conv(t, w, b, stride=1, act="relu")
    plan = pb.finalize(None, autotune=False)
    for _ in range(3):
        plan.run()
    torch.cuda.synchronize()
    plan.timing_begin(passes)
    for _ in range(passes):
        plan.run_timed()
    torch.cuda.synchronize()
    return [round(r["ms"] * 1000, 1) for r in plan.timing_read()]


for label, pat in (("AAAAAAAA", [A] * 8), ("BBBBBBBB", [Bv] * 8), ("AABBAABB", [A, A, Bv, Bv, A, A, Bv, Bv]),
                   ("ABABABAB", [A, Bv] * 4), ("ABCABCAB", [A, Bv, Cv, A, Bv, Cv, A, Bv])):
    print(label, "us per op:", run(pat), flush=True)
print("A = pipe_c2p2, B = mfma_c2p1, C = mfma_c4p1")
