#!/usr/bin/env python3
"""Where a conv variant's output deviates from a torch fp32 GPU convolution of the same fp16 operands (debugging aid):
   python tools/wreg_debug.py 128,128,3,1,80,80,32 wreg_p4 [repeat]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as F
from yolov6_amd import _lib
from yolov6_amd.engine import PlanBuilder, TRef
lib = _lib.load()
names = [lib.y6_conv_variant_name(i).decode() for i in range(lib.y6_conv_variants())]
spec, vname = sys.argv[1], sys.argv[2]
rep = int(sys.argv[3]) if len(sys.argv) > 3 else 3
cin, cout, k, s, H, W, B = (int(v) for v in spec.split(","))
torch.manual_seed(0)
x = torch.randn((B, H, W, cin), device="cuda:0").half()
w = (torch.randn((cout, cin, k, k)) / (cin * k * k) ** 0.5).half().float()
b = torch.randn(cout) * 0.1
ref = F.relu(F.conv2d(x.float().permute(0, 3, 1, 2), w.cuda(), b.cuda(), stride=s, padding=k // 2)).permute(0, 2, 3, 1)
pb = PlanBuilder("cuda:0"); pb.force_variant = names.index(vname)
o = pb.conv(TRef(x, B, H, W, cin, cin, 0), w, b, stride=s, act="relu")
plan = pb.finalize(None, autotune=False)
for r in range(rep):
    o.buf.zero_()
    plan.run(); torch.cuda.synchronize()
    got = o.to_nhwc_tensor().float()
    err = (got - ref).abs() / ref.abs().clamp(min=1.0)
    bad = err > 2e-3
    print(f"{spec} {vname} run {r}: max err {float(err.max()):.3e}, bad elements {int(bad.sum())} of {bad.numel()}")
    if bad.any():
        idx = bad.nonzero()
        for name, d in (("image", 0), ("row", 1), ("col", 2), ("channel", 3)):
            u, c = idx[:, d].unique(return_counts=True)
            print(f"   bad by {name}: {len(u)} distinct; first {[(int(a), int(n)) for a, n in zip(u[:24], c[:24])]}")
        i = idx[0].tolist()
        print("   first bad", i, "got", float(got[tuple(i)]), "want", float(ref[tuple(i)]))
