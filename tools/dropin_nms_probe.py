#!/usr/bin/env python
"""Is the speculated candidate path of utils/nms.py taken in the reference-shaped loop, and what does it buy?
   usage: python tools/dropin_nms_probe.py   (prints GPU time of model(x) + non_max_suppression per step, event-timed, and the loop's wall time)"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    from yolov6_amd.utils import nms as N
    import argparse
    a = argparse.Namespace(model="yolov6s", batch=32, size=640)
    _, _, model, x = bench.build_model_and_input(a, "cuda:0")
    bench.calibrate_head_bias(model, x)
    taken = []
    orig = N._speculated_candidates

    def spy(*a, **k):
        t = orig(*a, **k)
        taken.append(t is not None)
        return t
    N._speculated_candidates = spy
    for _ in range(5):
        d, _ = model(x)
        N.non_max_suppression(d, 0.03, 0.65, multi_label=True, max_det=300)
    torch.cuda.synchronize()
    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    fw = nm = 0.0
    n = 50
    t0 = time.perf_counter()
    for _ in range(n):
        e0.record()
        d, _ = model(x)
        e1.record()
        N.non_max_suppression(d, 0.03, 0.65, multi_label=True, max_det=300)
        e2.record()
        torch.cuda.synchronize()
        fw += e0.elapsed_time(e1)
        nm += e1.elapsed_time(e2)
    wall = (time.perf_counter() - t0) / n * 1e3
    print(f"sink={os.environ.get('Y6_DROPIN_SINK', '1')} taken {sum(taken)}/{len(taken)}  forward {fw / n:.3f} ms  nms {nm / n:.3f} ms  wall {wall:.3f} ms/step")


if __name__ == "__main__":
    main()
