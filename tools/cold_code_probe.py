#!/usr/bin/env python3
"""Why does the FIRST 3x3 conv of a run of equal layers cost ~25 us more than the others (profiles/r05/per_layer_infer_r05g.md)?
One layer (128 -> 128 @40x40, b32, 15.1 GFLOP) on one kernel, timed with events in different contexts:
  a  back to back                                     (code, weights, input warm)
  b  behind a 1 GB device copy                        (everything evicted from L2 / MALL)
  d  behind the copy + a rewrite of the layer's input (input warm; code + weights cold)
  f  behind the copy + the SAME kernel on another layer's weights (code warm; weights, input cold)
  g  behind the copy + rewrite of the input + the same kernel on another layer (code + input warm; weights cold)
  h  behind 8 launches of ANOTHER big conv kernel (~250 us)
"""
import json, os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from yolov6_amd import _lib
from yolov6_amd.engine import PlanBuilder, TRef

dev = "cuda:0"
lib = _lib.load()
names = [lib.y6_conv_variant_name(i).decode() for i in range(lib.y6_conv_variants())]


def layer(cin, cout, H, B, variant, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn((B, H, H, cin), generator=g).clamp(min=0).half().to(dev)
    w = torch.randn((cout, cin, 3, 3), generator=g) / (cin * 9) ** 0.5
    b = torch.randn((cout,), generator=g) * 0.1
    pb = PlanBuilder(dev)
    pb.force_variant = names.index(variant)
    pb.conv(TRef(x, B, H, H, cin, cin, 0), w, b, stride=1, act="relu")
    plan = pb.finalize(None, autotune=False)
    plan.run()
    torch.cuda.synchronize()
    return plan, x


V = sys.argv[1] if len(sys.argv) > 1 else "wreg_p7"
P1, x1 = layer(128, 128, 40, 32, V, 1)
P1b, _ = layer(128, 128, 40, 32, V, 2)          # same kernel, other weights / buffers
P2, _ = layer(64, 64, 80, 32, "dma_c2p2", 3)     # another big kernel
big_a = torch.empty(512 * 1024 * 1024, dtype=torch.uint8, device=dev)
big_b = torch.empty_like(big_a)


def timed(pre, n=20):
    ts = []
    for _ in range(n):
        pre()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        P1.run()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return round(statistics.median(ts), 2), round(min(ts), 2)


evict = lambda: big_b.copy_(big_a)
touch = lambda: x1.add_(0)
res = {"variant": V}
res["a_back_to_back"] = timed(lambda: P1.run())
res["b_behind_1GB_copy"] = timed(evict)
res["d_copy_then_rewrite_input"] = timed(lambda: (evict(), touch()))
res["f_copy_then_same_kernel_other_layer"] = timed(lambda: (evict(), P1b.run()))
res["g_copy_rewrite_input_same_kernel_other_layer"] = timed(lambda: (evict(), touch(), P1b.run()))
res["h_behind_8x_other_big_kernel"] = timed(lambda: [P2.run() for _ in range(8)])
res["i_8x_other_kernel_then_rewrite_input"] = timed(lambda: ([P2.run() for _ in range(8)], touch()))
print(json.dumps(res), flush=True)
