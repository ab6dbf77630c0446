#!/usr/bin/env python3
"""Per-stage deviation of the HIP training-form forward from the fp32 TrainOracle (debug / DESIGN numbers).
    python tools/train_trace.py [size] [batch] [case]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from oracle import synth
from oracle.model_oracle import TrainOracle
from tests.helpers import case_config, synth_sd_from_keys
from yolov6_amd.models.yolo import build_model

size = int(sys.argv[1]) if len(sys.argv) > 1 else 64
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 2
case = sys.argv[3] if len(sys.argv) > 3 else "tiny"
cfg, meta = case_config(case)
model = build_model(cfg, meta["num_classes"], "cpu")
sd = synth_sd_from_keys(meta["train"])
model.load_state_dict(sd)
x = synth.synth_images(batch, size, seed=21)
orc = TrainOracle(cfg, sd, meta["num_classes"])
a = orc.a
with torch.no_grad():
    ref = {}
    t = orc.block(x.half().float(), "backbone.stem", 2)
    ref["backbone.stem"] = t
    last = 6 if a.p6 else 5
    outs = []
    for k in range(2, last + 1):
        p = f"backbone.ERBlock_{k}"
        t = orc.block(t, p + ".0", 2)
        ref[p + ".0"] = t
        t = orc.stage(t, p + ".1", a.n[k - 1])
        ref[p + ".1"] = t
        if k == last:
            t = orc.merge(t, p + ".2")
            ref[p + ".2"] = t
        outs.append(t)
    first = 2 if (a.fuse_P2 or a.backbone == "CSPBepBackbone_P6") else 3
    feats = orc.neck(outs[first - 2:])
    for i, f in enumerate(feats):
        s_ = orc.convbn(f, f"detect.stems.{i}", "silu")
        ref[f"detect.stem{i}"] = s_
        c = orc.convbn(s_, f"detect.cls_convs.{i}", "silu")
        r = orc.convbn(s_, f"detect.reg_convs.{i}", "silu")
        ref[f"detect.cls_conv{i}"], ref[f"detect.reg_conv{i}"] = c, r
        import torch.nn.functional as F
        ref[f"detect.cls_logit{i}"] = F.conv2d(c, orc.sd[f"detect.cls_preds.{i}.weight"], orc.sd[f"detect.cls_preds.{i}.bias"])
        ref[f"detect.reg_raw{i}"] = F.conv2d(r, orc.sd[f"detect.reg_preds.{i}.weight"], orc.sd[f"detect.reg_preds.{i}.bias"])
model = model.to("cuda:0").train()
with torch.no_grad():
    out, _ = model(x.to("cuda:0").half())
torch.cuda.synchronize()
g = next(iter(model.__dict__["_y6_train_graphs"].values()))
for name, tref in g.tb.trace.items():
    if name not in ref:
        continue
    got = tref.to_nhwc_tensor().float().cpu().permute(0, 3, 1, 2)
    r = ref[name]
    err = float((got - r).abs().max())
    print(f"{name:28s} shape {tuple(r.shape)}  max|ref| {float(r.abs().max()):8.3f}  max err {err:.3e}  rel {err / max(1.0, float(r.abs().max())):.3e}")
