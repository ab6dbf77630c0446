#!/usr/bin/env python3
"""How much of a step's time is launch ramps?  The bench's step (plan.run + nms_raw on one b32 batch) with 1, 2 or 3 steps in
flight: step i runs on stream i % n with its own plan (own activation buffers, same weights values).  Throughput only - every
step still does its full forward + NMS.   python tools/inflight_probe.py [steps]"""
import argparse, copy, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
args = argparse.Namespace(model="yolov6s", batch=32, size=640)
from yolov6_amd.utils.nms import nms_raw
dev = "cuda:0"
cfg, sd, model, x = bench.build_model_and_input(args, dev)
bench.calibrate_head_bias(model, x)
models = [model, copy.deepcopy(model), copy.deepcopy(model)]
plans, toks = [], []
for m in models:
    p = m.compile(x, autotune=True)
    plans.append(p)
    toks.append(p.attach_nms(bench.CONF, None, True))
for n in (1, 2, 3, 1, 2):
    streams = [torch.cuda.Stream() for _ in range(n)]
    def step(i):
        k = i % n
        with torch.cuda.stream(streams[k]):
            det = plans[k].run()
            return nms_raw(det, bench.CONF, bench.IOU, multi_label=True, max_det=bench.MAX_DET, candidates=toks[k])
    for i in range(12):
        out = step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        out = step(i)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(json.dumps(dict(inflight=n, steps=steps, ms_per_step=round(dt / steps * 1e3, 4), img_s=round(32 * steps / dt, 1),
                          kept=float(out[2].float().mean()))), flush=True)
