#!/usr/bin/env python3
"""Where does the non-finite gradient of the tiny model's stem 1x1 branch come from?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from oracle import synth
from yolov6_amd.configs import tiny_config
from yolov6_amd.models.yolo import build_model
cfg = tiny_config()
model = build_model(cfg, 80, "cpu")
model.load_state_dict(synth.synth_state_dict(model.state_dict(), seed=0))
model = model.to("cuda:0").train()
x = synth.synth_images(2, 64, seed=3).to("cuda:0").half()
out, _ = model(x)
stems, scores, distri = out
((scores * scores).sum() + distri.square().mean()).mul(256.0).backward()
torch.cuda.synchronize()
g = next(iter(model.__dict__["_y6_train_graphs"].values()))
named = dict(model.named_parameters())
w = named["backbone.stem.rbr_1x1.conv.weight"]
print("grad finite:", bool(torch.isfinite(w.grad).all()), w.grad.flatten()[:8].tolist(), "shape", tuple(w.shape))
for i, e in enumerate(g.bwd_log):
    if e["kind"] == "wgrad" and e.get("weight") is w:
        print("wgrad op", i, {k: (tuple(v.shape) if hasattr(v, "shape") else v) for k, v in e.items() if k in ("mode", "k", "stride", "cout", "nhwc", "dil")})
        a = e.get("a"); pls = e.get("planes")
        if a is not None:
            print("  a plane numel", a.numel(), "finite", bool(torch.isfinite(a.float()).all()), "absmax", float(a.float().abs().max()))
            for j, p in enumerate(pls):
                print("  B plane", j, "numel", p.numel(), "finite", bool(torch.isfinite(p.float()).all()), "absmax", float(torch.nan_to_num(p.float()).abs().max()))
        dy = e["dy"]
        t = dy.to_nhwc_tensor().float()
        print("  dy", tuple(t.shape), "finite", bool(torch.isfinite(t).all()), "absmax", float(t.abs().max()))
        # re-run just this op (and its transposes) after zeroing the gradient, to see if it reproduces
        gview = w.grad
        for rep in range(2):
            gview.zero_()
            lo = i
            while lo > 0 and g.bwd_log[lo - 1]["kind"] == "wgrad_transpose":
                lo -= 1
            g.bwd_plan.run_range(lo, i + 1)
            torch.cuda.synchronize()
            print("  rerun", rep, "ops", lo, i, "finite", bool(torch.isfinite(gview).all()), gview.flatten()[:4].tolist())
