#!/usr/bin/env python3
"""What resets the first-use penalty of a kernel function?  YOLOv6-S b32, shape-derived plan, one stream; the LAST range of every
sequence is timed (median of 9)."""
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
n = plan.num_ops
big_a = torch.empty(512 * 1024 * 1024, dtype=torch.uint8, device=dev); big_b = torch.empty_like(big_a)
R = lambda i: (i, i + 1)

def t_last(seq, reps=9):
    ts = []
    for _ in range(reps):
        for item in seq[:-1]:
            if item == "copy":
                big_b.copy_(big_a)
            elif item == "sync":
                torch.cuda.synchronize()
            else:
                plan.run_range(*item)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); plan.run_range(*seq[-1]); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return round(statistics.median(ts), 1)

res = {}
res["a  ops0-3 | op4"] = t_last([(0, 4), R(4)])
res["a' ops0-4 | op5"] = t_last([(0, 5), R(5)])
res["b  ops0-3 | op5 (same kernel+shape, other layer)"] = t_last([(0, 4), R(5)])
res["c  ops0-3 | op9 (same kernel, 40x40 map)"] = t_last([(0, 4), R(9)])
res["d  op14, ops0-3 | op4"] = t_last([R(14), (0, 4), R(4)])
res["e  op14, ops1-3 | op4"] = t_last([R(14), (1, 4), R(4)])
res["f  op14, ops2-3 | op4"] = t_last([R(14), (2, 4), R(4)])
res["g  op14, op3 | op4"] = t_last([R(14), R(3), R(4)])
res["g' op5, op3 | op4"] = t_last([R(5), R(3), R(4)])
res["h  ops0-3, op9 | op4"] = t_last([(0, 4), R(9), R(4)])
res["i  ops0-3, copy | op4"] = t_last([(0, 4), "copy", R(4)])
res["j  op5, copy, copy | op4"] = t_last([R(5), "copy", "copy", R(4)])
res["k  op5, ops18-30 | op4"] = t_last([R(5), (18, 31), R(4)])
res["l  op5, ops18-30 x3 | op4"] = t_last([R(5), (18, 31), (18, 31), (18, 31), R(4)])
res["m  op5, op3 x6 | op4"] = t_last([R(5)] + [R(3)] * 6 + [R(4)])
res["n  full step | op4 (no op3 in front)"] = t_last([(0, n), R(4)])
res["o  full step, op3 | op4"] = t_last([(0, n), R(3), R(4)])
res["p  full step, sync, op3 | op4"] = t_last([(0, n), "sync", R(3), R(4)])
res["q  ops0-39 | op40"] = t_last([(0, 40), R(40)])
res["r  ops0-39, op41 | op40"] = t_last([(0, 40), R(41), R(40)])
res["s  op41, ops35-39 | op40"] = t_last([R(41), (35, 40), R(40)])
res["t  op41, op39 | op40"] = t_last([R(41), R(39), R(40)])
res["u  ops0-54 | op55"] = t_last([(0, 55), R(55)])
res["v  op55 | op55"] = t_last([R(55), R(55)])
res["w  op9, op54 | op55"] = t_last([R(9), R(54), R(55)])
print(json.dumps(res, indent=1))
