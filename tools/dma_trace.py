#!/usr/bin/env python3
"""s_memtime timeline of block 0 / thread 0 of the LDS-DMA conv kernel (needs the trace build:
   python tools/build_probe_libs.py --dma 1;  Y6_LIB_PATH=tools/_build/libyolov6_hip_dmaprobe1.so python tools/dma_trace.py 128,128,3,1,80,80,32 dma_c2p2)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
buf = torch.zeros(4096, dtype=torch.int64, device="cuda:0")
os.environ["Y6_CONV_TRACE"] = str(buf.data_ptr())
from yolov6_amd import _lib
from yolov6_amd.engine import PlanBuilder, TRef
lib = _lib.load()
names = [lib.y6_conv_variant_name(i).decode() for i in range(lib.y6_conv_variants())]
spec, vname = sys.argv[1], sys.argv[2]
variant = names.index(vname)
cin, cout, k, s, H, W, B = (int(v) for v in spec.split(","))
x = torch.randn((B, H, W, cin), device="cuda:0").half()
if os.environ.get("Y6_TRACE_DATA") == "relu":
    x = x.clamp(min=0)
if os.environ.get("Y6_TRACE_DATA") == "zeros":
    x = x * 0
w = torch.randn((cout, cin, k, k)) / (cin * k * k) ** 0.5
pb = PlanBuilder("cuda:0"); pb.force_variant = variant
pb.conv(TRef(x, B, H, W, cin, cin, 0), w, torch.zeros(cout), stride=s, act="relu")
plan = pb.finalize(None, autotune=False)
for _ in range(3):
    plan.run()
torch.cuda.synchronize()
import time
buf.zero_(); torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
ev[0].record(); plan.run(); ev[1].record(); torch.cuda.synchronize()
print(f"one launch between two events: {ev[0].elapsed_time(ev[1]) * 1e3:.1f} us")
allb = buf.cpu()
t = allb[:512].view(256, 2).tolist()
blk = allb[1024:3072].view(1024, 2)
blk = blk[blk[:, 0] > 0]
if blk.shape[0]:
    t0 = int(blk[:, 0].min())
    st, en = (blk[:, 0] - t0).float() / 100.0, (blk[:, 1] - t0).float() / 100.0
    life = en - st
    q = lambda v, f: float(v.sort().values[int(f * (len(v) - 1))])
    print(f"{blk.shape[0]} blocks (us on the 100 MHz counter): start min/median/max {q(st, 0):.2f}/{q(st, .5):.2f}/{q(st, 1):.2f}, end min/median/max {q(en, 0):.2f}/{q(en, .5):.2f}/{q(en, 1):.2f}, "
          f"lifetime min/median/max {q(life, 0):.2f}/{q(life, .5):.2f}/{q(life, 1):.2f}")
    print("   end-time histogram (us):", torch.histc(en, bins=10, min=float(en.min()), max=float(en.max())).int().tolist(), f"from {float(en.min()):.1f} to {float(en.max()):.1f}")
if vname.startswith("wreg"):   # conv_wreg.hip (python tools/build_probe_libs.py --wreg 1)
    tags = {1: "kernel start", 2: "prologue done", 9: "stage top", 10: "barrier passed", 11: "next stage's halo requested", 12: "18 units issued",
            20: "epilogue: args, output pixels, bias", 21: "epilogue: fragments stored", 22: "last loads landed (kernel end)"}
else:
  tags = {1: "kernel start", 2: "prologue done", 9: "chunk top (after prev taps/epilogue)", 10: "barrier passed", 11: "taps issued", 19: "chunk loop done",
        20: "epilogue issued", 21: "args reloaded + output pixels", 22: "cout fragment 0 stored", 23: "last item's epilogue units issued (kernel end)"}
real = [ts for ts, tag in t if tag in (90, 91)]
t = [e for e in t if e[1] not in (90, 91)]
if len(real) == 2:
    cyc = [ts for ts, tag in t if tag in (1, 22)]
    if len(cyc) == 2:
        us = (real[1] - real[0]) / 100.0
        print(f"block 0 lived {cyc[1] - cyc[0]} shader cycles in {us:.2f} us (100 MHz counter): {(cyc[1] - cyc[0]) / us / 1e3:.3f} GHz")
prev = t[0][0]
acc = {}
for ts, tag in t:
    if tag == 0: break
    acc.setdefault(tag, []).append(ts - prev); prev = ts
n = sum(len(v) for v in acc.values())
print(spec, vname, "events", n, "span", (prev - t[0][0]))
for tag, v in sorted(acc.items()):
    print("  -> %-40s n=%3d mean %8.0f  min %8.0f max %8.0f" % (tags.get(tag, tag), len(v), sum(v) / len(v), min(v), max(v)))
print("  first 60:", [(int(tag), int(b - a)) for (a, _), (b, tag) in zip(t[:60], t[1:61]) if tag])
