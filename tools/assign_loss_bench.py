#!/usr/bin/env python3
"""Training-side hot path (label assignment + loss value) at the reference's training shape: B images x 8400 anchors
x 80 classes, mosaic-like target counts.  Prints one JSON line: ms per call of ComputeLoss (bbox_decode + TAL or ATSS +
loss terms) and of the assigner alone, the algorithmic HBM bytes (scores read twice, target scores written once and
read twice, boxes), and the CPU oracle on a bounded sample of the same inputs.

    python tools/assign_loss_bench.py [--batch 64] [--iters 20] [--cpu-batch 2]
"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--cpu-batch", type=int, default=2)
a = ap.parse_args()

from yolov6_amd.models.losses.loss import ComputeLoss
from yolov6_amd.utils import synth

FS, ST, C = [(80, 80), (40, 40), (20, 20)], [8, 16, 32], 80
dev = "cuda:0"
inp = synth.synth_loss_inputs(a.batch, FS, ST, C, 16, False, seed=3, boxes_per_image=(8, 60))
feats = [torch.zeros(a.batch, 1, h, w, device=dev) for h, w in FS]
ps, pd, tg = inp["pred_scores"].to(dev), inp["pred_distri"].to(dev), inp["targets"].to(dev)
res = {}
for label, epoch in (("tal", 10), ("atss", 0)):
    crit = ComputeLoss(fpn_strides=ST, num_classes=C, ori_img_size=640, warmup_epoch=4, use_dfl=False, reg_max=16, iou_type="giou")
    for _ in range(3):
        loss, items = crit((feats, ps, pd), tg, epoch, 1, 640, 640)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(a.iters):
        loss, items = crit((feats, ps, pd), tg, epoch, 1, 640, 640)
    torch.cuda.synchronize()
    res[label] = dict(ms_per_call=round((time.perf_counter() - t0) / a.iters * 1e3, 3), loss=round(float(loss), 4))
A = 8400
G = int(np.bincount(inp["targets"][:, 0].numpy().astype(int), minlength=a.batch).max())
alg = a.batch * A * (C * 4 * 2 + C * 4 * 3 + 16 * 3 + 8 + 1)     # scores x2, target scores w + 2r, boxes, labels, fg
out = dict(workload=f"ComputeLoss value: B{a.batch} x A{A} x C{C}, up to {G} gt/img, no DFL, giou", tal=res["tal"], atss=res["atss"],
           algorithmic_bytes=alg, tal_gbs=round(alg / (res["tal"]["ms_per_call"] * 1e-3) / 1e9, 1),
           note="ms_per_call includes the host-side target packing of loss.py:184-192 (numpy, as in the reference)")
if a.cpu_batch > 0:
    from oracle import loss_oracle
    nb = a.cpu_batch
    keep = inp["targets"][:, 0] < nb
    t0 = time.perf_counter()
    ref = loss_oracle.compute_loss(FS, inp["pred_scores"][:nb].numpy(), inp["pred_distri"][:nb].numpy(), inp["targets"][keep].numpy(),
                                   10, 640, 640, fpn_strides=ST, num_classes=C, warmup_epoch=4, use_dfl=False, iou_type="giou")
    dt = time.perf_counter() - t0
    out["cpu_baseline"] = dict(kind="port", sample=f"{nb} images of the same batch, numpy oracle (TAL path), 1 core",
                               ms_per_image=round(dt / nb * 1e3, 1), cores=1)
print(json.dumps(out), flush=True)
