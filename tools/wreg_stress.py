#!/usr/bin/env python3
"""Determinism stress of a conv variant: the same launch N times on the same operands must give the same bits every time
(a race in the kernel's own synchronisation shows up as a rare mismatch).  Optionally beside a memory-hungry kernel on a second stream.
   python tools/wreg_stress.py 128,128,3,1,80,80,32 wreg_p4 [iters] [--noise]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from yolov6_amd import _lib
from yolov6_amd.engine import PlanBuilder, TRef
lib = _lib.load()
names = [lib.y6_conv_variant_name(i).decode() for i in range(lib.y6_conv_variants())]
spec, vname = sys.argv[1], sys.argv[2]
iters = int(sys.argv[3]) if len(sys.argv) > 3 and not sys.argv[3].startswith("-") else 300
noise = "--noise" in sys.argv
cin, cout, k, s, H, W, B = (int(v) for v in spec.split(","))
torch.manual_seed(0)
x = torch.randn((B, H, W, cin), device="cuda:0").half().clamp(min=0)
w = torch.randn((cout, cin, k, k)) / (cin * k * k) ** 0.5
b = torch.randn(cout) * 0.1
pb = PlanBuilder("cuda:0"); pb.force_variant = names.index(vname)
o = pb.conv(TRef(x, B, H, W, cin, cin, 0), w, b, stride=s, act="relu")
plan = pb.finalize(None, autotune=False)
plan.run(); torch.cuda.synchronize()
first = o.to_nhwc_tensor().clone()
side = torch.cuda.Stream()
big = torch.empty(256 << 20, dtype=torch.uint8, device="cuda:0") if noise else None
bad = 0
worst = 0.0
for it in range(iters):
    o.buf.fill_(7.0)
    if noise:
        with torch.cuda.stream(side):
            big.add_(1)            # 512 MB of HBM traffic beside the conv
    plan.run()
    torch.cuda.synchronize()
    got = o.to_nhwc_tensor()
    if not torch.equal(got, first):
        bad += 1
        d = (got.float() - first.float()).abs()
        worst = max(worst, float(d.max()))
        if bad <= 3:
            idx = (d > 0).nonzero()
            print(f"   iter {it}: {idx.shape[0]} elements differ, max {float(d.max()):.3e}; images {sorted(set(idx[:, 0].tolist()))[:8]} rows {sorted(set(idx[:, 1].tolist()))[:12]} cols {sorted(set(idx[:, 2].tolist()))[:12]} channels {len(set(idx[:, 3].tolist()))}")
print(f"{spec} {vname} noise={noise} lib={os.environ.get('Y6_LIB_PATH', 'product')}: {bad} of {iters} runs differ from the first (worst {worst:.3e})")
