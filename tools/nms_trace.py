#!/usr/bin/env python3
"""s_memtime timeline of block 0 / thread 0 of the NMS sweep on the bench's own decode output (debug tool)."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
buf = torch.zeros(512, dtype=torch.int64, device="cuda:0")
os.environ["Y6_NMS_TRACE"] = str(buf.data_ptr())
import bench
from yolov6_amd.utils.nms import nms_raw
class A: pass
a = A(); a.model = "yolov6s"; a.size = 640; a.batch = 32; a.no_autotune = True; a.int8 = False
cfg, sd, model, x = bench.build_model_and_input(a, "cuda:0")
bench.calibrate_head_bias(model, x)
det = model.compile(x, autotune=False).run()
for _ in range(3):
    out = nms_raw(det, 0.03, 0.65, multi_label=True, max_det=300)
torch.cuda.synchronize()
t = buf.cpu().view(256, 2).tolist()
prev = t[0][0]
ev = []
for ts, tag in t:
    if tag == 0: break
    ev.append((int(tag) & 0xff, int(tag) >> 8, int(ts - prev))); prev = ts
print("events", len(ev), "total cycles", sum(e[2] for e in ev))
print(ev[:120])
names = {2: "window boxes built", 3: "survivors of earlier windows", 10: "batch gathered (arg = size)", 11: "column masks", 12: "fixed point + publish (arg = kept)", 13: "later candidates tested", 20: "outputs"}
acc = {}
for tag, arg, d in ev:
    acc.setdefault(tag, []).append(d)
for tag, v in sorted(acc.items()):
    print("  %-40s n=%3d mean %8.0f sum %9d" % (names.get(tag, tag), len(v), sum(v) / len(v), sum(v)))
