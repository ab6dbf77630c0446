"""Time the EffiDeHead eval epilogue (head_decode) alone at the bench shape: b32, 640x640, 80 classes."""
import os, sys, torch
sys.path.insert(0, os.getcwd())
from yolov6_amd.engine import PlanBuilder
sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import gpu_utils as G
B, nc = 32, 80
sizes, strides = [(80, 80), (40, 40), (20, 20)], [8.0, 16.0, 32.0]
cls = [G.rand_nhwc(B, h, w, nc, seed=i, scale=4.0) for i, (h, w) in enumerate(sizes)]
reg = [G.rand_nhwc(B, h, w, 4, seed=9 + i, scale=3.0) for i, (h, w) in enumerate(sizes)]
pb = PlanBuilder("cuda:0")
out = pb.head_decode(cls, reg, strides, False, 16, torch.linspace(0, 16, 17), nc)
plan = pb.finalize(out, autotune=False)
for _ in range(3): plan.run()
torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): plan.run()
e1.record(); torch.cuda.synchronize(); ms = e0.elapsed_time(e1) / 20
by = B * 8400 * (85 * 4 + 84 * 2)
print("decode", "flat" if os.environ.get("Y6_DECODE_FLAT") else "tiled", round(ms * 1000, 1), "us", round(by / ms / 1e6, 1), "GB/s")
