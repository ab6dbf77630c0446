"""Time the fused producer -> stride-2 ops alone at the bench shapes (YOLOv6-S b32 640x640):  python tools/fused_bench.py
Y6_FUSED_PROBE=<mask> disables phases (timing probes, wrong results): 1 producer, 2 consumer MFMA loop, 4 output stores, 8 input loads."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from yolov6_amd.engine import PlanBuilder, NCHWInput, TRef

dev = "cuda:0"
g = torch.Generator().manual_seed(0)


def w(co, ci, k):
    return torch.randn((co, ci, k, k), generator=g) * (1.0 / (ci * k * k) ** 0.5), torch.randn(co, generator=g) * 0.1


def timed(plan, iters=50):
    for _ in range(5):
        plan.run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        plan.run()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1000.0


res = {"probe": int(os.environ.get("Y6_FUSED_PROBE", "0"))}
B = 32
for fuse in (True, False):
    tag = "fused" if fuse else "two_ops"
    # stem (3 -> 32) + 3x3 s2 (32 -> 64) on 640x640
    pb = PlanBuilder(dev); pb._fuse_s2 = fuse; pb._fuse_pw_widths = (64, 128)
    img = torch.rand((B, 3, 640, 640), generator=g).half().to(dev)
    pb.hint_single_use()
    t = pb.conv(NCHWInput(img), *w(32, 3, 3), stride=2, act="relu")
    o = pb.conv(t, *w(64, 32, 3), stride=2, act="relu")
    res[f"stem_s2_{tag}_us"] = round(timed(pb.finalize(o, autotune=False)), 1)
    for C_, H in ((64, 160), (128, 80)):
        pb = PlanBuilder(dev); pb._fuse_s2 = fuse; pb._fuse_pw_widths = (64, 128)
        x = TRef(torch.randn((B, H, H, C_), generator=g).half().to(dev), B, H, H, C_, C_, 0)
        pb.hint_single_use()
        t = pb.conv(x, *w(C_, C_, 1), stride=1, act="relu")
        o = pb.conv(t, *w(C_, C_, 3), stride=2, act="relu")
        res[f"pw_s2_{C_}_{tag}_us"] = round(timed(pb.finalize(o, autotune=not fuse)), 1)
    if res["probe"]:
        break
print(json.dumps(res))
