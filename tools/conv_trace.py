#!/usr/bin/env python3
"""s_memtime timeline of block 0 / wave 0 of the persistent conv kernel (debug tool).
   python tools/conv_trace.py 256,256,3,1,40,40,32 10"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
buf = torch.zeros(512, dtype=torch.int64, device="cuda:0")
os.environ["Y6_CONV_TRACE"] = str(buf.data_ptr())
from yolov6_amd.engine import PlanBuilder, TRef
spec, variant = sys.argv[1], int(sys.argv[2])
cin, cout, k, s, H, W, B = (int(v) for v in spec.split(","))
x = torch.randn((B, H, W, cin), device="cuda:0").half()
w = torch.randn((cout, cin, k, k)) / (cin * k * k) ** 0.5
pb = PlanBuilder("cuda:0"); pb.force_variant = variant
pb.conv(TRef(x, B, H, W, cin, cin, 0), w, torch.zeros(cout), stride=s, act="relu")
plan = pb.finalize(None, autotune=False)
for _ in range(3):
    plan.run()
torch.cuda.synchronize()
t = buf.cpu().view(256, 2).tolist()
names = {1: "start", 2: "prologue done", 10: "prefetch issued", 11: "mfma issued", 12: "lds reads done", 13: "prefetch landed",
         14: "barrier", 15: "halo published", 20: "epilogue issued"}
prev = t[0][0]
acc = {}
for ts, tag in t:
    if tag == 0: break
    d = ts - prev; prev = ts
    acc.setdefault(tag, []).append(d)
print(spec, "variant", variant, "events", sum(len(v) for v in acc.values()))
for tag, v in sorted(acc.items()):
    print("  -> %-18s n=%3d mean %8.0f  min %8.0f max %8.0f ticks" % (names.get(tag, tag), len(v), sum(v) / len(v), min(v), max(v)))
print("  first 40 deltas:", [(int(tag), int(b - a)) for (a, _), (b, tag) in zip(t[:40], t[1:41]) if tag])
