#!/usr/bin/env python3
"""In-plan probe of the first-of-run penalty: ops of the real YOLOv6-S b32 plan replayed in chosen orders with events around one op."""
import json, os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench

class A: model="yolov6s"; batch=32; size=640
dev = torch.device("cuda:0")
cfg, sd, model, x = bench.build_model_and_input(A, dev)
os.environ["Y6_SCHED_STREAMS"] = "1"
plan = model.compile(x, autotune=False)
plan.run(); torch.cuda.synchronize()
vt = dict(plan.variant_table())
n = plan.num_ops

def t_last(seq, reps=15):
    """run the (first,last) ranges of seq in order; time the LAST one"""
    ts = []
    for _ in range(reps):
        for a, b in seq[:-1]:
            plan.run_range(a, b)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); plan.run_range(*seq[-1]); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return round(statistics.median(ts), 1)

res = {"variants": {k: v for k, v in vt.items() if k in (16, 17, 19, 24, 30, 31, 32, 33, 45, 46)}}
whole = [(0, n)]
R = lambda i: (i, i + 1)
res["op31 in the full plan order (ops 0..30 before it)"] = t_last([(0, 31), R(31)])
res["op32 in the full plan order"] = t_last([(0, 32), R(32)])
res["op31 twice: second"] = t_last([(0, 31), R(31), R(31)])
res["op31 right behind op30 only (after a full run)"] = t_last([(0, n), R(30), R(31)])
res["op31 right behind op32 (same kernel, other layer)"] = t_last([(0, n), R(32), R(31)])
res["op32 right behind op30 (the 1x1)"] = t_last([(0, n), R(30), R(32)])
res["op31 behind ops 18..30"] = t_last([(0, n), (18, 31), R(31)])
res["op31 behind ops 25..30"] = t_last([(0, n), (25, 31), R(31)])
res["op31 behind ops 28..30"] = t_last([(0, n), (28, 31), R(31)])
res["op31 behind op 29 (fused pw_s2) + 30"] = t_last([(0, n), (29, 31), R(31)])
res["op31 alone after a full run"] = t_last([(0, n), R(31)])
res["op45 in plan order"] = t_last([(0, 45), R(45)])
res["op45 behind op44 only"] = t_last([(0, n), R(44), R(45)])
res["op45 behind ops 35..44"] = t_last([(0, n), (35, 45), R(45)])
res["op19 in plan order"] = t_last([(0, 19), R(19)])
res["op24 in plan order"] = t_last([(0, 24), R(24)])
print(json.dumps(res, indent=1))
