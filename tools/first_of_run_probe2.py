#!/usr/bin/env python3
"""Is the first-of-run penalty of the per-op tables in the un-instrumented step?  YOLOv6-S b32 plan, one stream:
  T_full   one whole plan.run_range(0, n), events around it only (median of 20)
  S_chain  sum over ops of [ops 0..i-1 enqueued in one go, then op i between two events]
  S_timed  sum over ops of plan.run_timed() (an event after every op - what bench.py's breakdown uses)"""
import json, os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench

class A: model="yolov6s"; batch=32; size=640
dev = torch.device("cuda:0")
cfg, sd, model, x = bench.build_model_and_input(A, dev)
os.environ["Y6_SCHED_STREAMS"] = "1"
tune = len(sys.argv) > 1 and sys.argv[1] == "tune"
plan = model.compile(x, autotune=tune)
plan.run(); torch.cuda.synchronize()
n = plan.num_ops

def ev():
    return torch.cuda.Event(enable_timing=True)

def t_full(reps=20):
    ts = []
    for _ in range(reps):
        e0, e1 = ev(), ev()
        e0.record(); plan.run_range(0, n); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return statistics.median(ts)

def t_pipelined(k=10):
    e0, e1 = ev(), ev()
    e0.record()
    for _ in range(k):
        plan.run_range(0, n)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / k

def t_chain(i, reps=7):
    ts = []
    for _ in range(reps):
        if i:
            plan.run_range(0, i)
        e0, e1 = ev(), ev()
        e0.record(); plan.run_range(i, i + 1); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return statistics.median(ts)

res = {"autotuned": tune, "ops": n}
res["T_full_us"] = round(t_full(), 1)
res["T_pipelined_us_per_step"] = round(t_pipelined(), 1)
chain = [round(t_chain(i), 1) for i in range(n)]
res["S_chain_us"] = round(sum(chain), 1)
plan.timing_begin(8)
for _ in range(8):
    plan.run_timed()
torch.cuda.synchronize()
rows = plan.timing_read()
timed = [round(r["ms"] * 1e3, 1) for r in rows]
res["S_timed_us"] = round(sum(timed), 1)
res["per_op"] = [{"op": i, "kind": rows[i]["kind"], "variant": rows[i]["variant"], "chain_us": chain[i], "timed_us": timed[i]} for i in range(n)]
print(json.dumps(res))
