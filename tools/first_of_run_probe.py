#!/usr/bin/env python3
"""Can a tiny launch of the same kernel function pay the first-use penalty off the critical path?"""
import json, os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from yolov6_amd import _lib
from yolov6_amd.engine import PlanBuilder, TRef

class A: model="yolov6s"; batch=32; size=640
dev = torch.device("cuda:0")
cfg, sd, model, x = bench.build_model_and_input(A, dev)
os.environ["Y6_SCHED_STREAMS"] = "1"
plan = model.compile(x, autotune=False)
plan.run(); torch.cuda.synchronize()
lib = _lib.load()
names = [lib.y6_conv_variant_name(i).decode() for i in range(lib.y6_conv_variants())]

def mini(cin, cout, H, B, variant):
    xx = torch.randn((B, H, H, cin)).half().to(dev)
    w = torch.randn((cout, cin, 3, 3)) / (cin * 9) ** 0.5
    pb = PlanBuilder(dev); pb.force_variant = names.index(variant)
    pb.conv(TRef(xx, B, H, H, cin, cin, 0), w, torch.zeros(cout), stride=1, act="relu")
    p = pb.finalize(None, autotune=False); p.run(); torch.cuda.synchronize()
    return p

small = mini(128, 128, 8, 1, "wreg_p7")       # 1 item
mid = mini(128, 128, 40, 4, "wreg_p7")        # 32 items
full1 = mini(32, 128, 40, 32, "wreg_p7")      # 256 items, one 32-channel stage: every CU runs the loop once (~6 us of work)
side = torch.cuda.Stream()
R = lambda i: (i, i + 1)

def t_last(seq, reps=9):
    ts = []
    for _ in range(reps):
        for item in seq[:-1]:
            if isinstance(item, tuple):
                plan.run_range(*item)
            elif isinstance(item, list):        # [plan] -> on the side stream, concurrently with what follows
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    item[0].run()
            else:
                item.run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); plan.run_range(*seq[-1]); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return round(statistics.median(ts), 1)

def t_self(p, pre, reps=9):
    ts = []
    for _ in range(reps):
        for item in pre:
            plan.run_range(*item)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); p.run(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return round(statistics.median(ts), 1)

res = {}
res["baseline ops0-3 | op4"] = t_last([(0, 4), R(4)])
res["ops0-3, small | op4"] = t_last([(0, 4), small, R(4)])
res["ops0-3, mid | op4"] = t_last([(0, 4), mid, R(4)])
res["ops0-3, full1 | op4"] = t_last([(0, 4), full1, R(4)])
res["ops0-2, small, op3 | op4"] = t_last([(0, 3), small, R(3), R(4)])
res["ops0-2, full1, op3 | op4"] = t_last([(0, 3), full1, R(3), R(4)])
res["ops0-2, [full1 on side stream], op3 | op4"] = t_last([(0, 3), [full1], R(3), R(4)])
res["ops0-2, [small on side stream], op3 | op4"] = t_last([(0, 3), [small], R(3), R(4)])
res["cost of small behind ops0-3"] = t_self(small, [(0, 4)])
res["cost of mid behind ops0-3"] = t_self(mid, [(0, 4)])
res["cost of full1 behind ops0-3"] = t_self(full1, [(0, 4)])
res["cost of full1 behind itself"] = t_self(full1, [])
res["op3 behind ops0-2"] = t_last([(0, 3), R(3)])
res["op3 behind op8 (same kernel)"] = t_last([(0, 3), R(8), R(3)])
print(json.dumps(res, indent=1))
