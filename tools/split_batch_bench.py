#!/usr/bin/env python3
"""A/B: the headline step (YOLOv6-S 640x640 b32 fp16, forward + NMS) as ONE plan over 32 images on one stream, against the same
32 images as 2 / 4 micro-batches, each with its own plan (own activation buffers) on its own HIP stream.

Why it might pay: the conv kernels are persistent launches sized to fill the chip; the last, partial round of a launch's walk
leaves most CUs idle (17-22 % on the 60-GFLOP layers) and the 21 small layers have at most one work item per block.  Kernels
of an independent micro-batch on another stream can run in those holes.  Why it might not: two persistent launches compete for
the same CU slots (the side-stream NMS experiment of visit r03q: +1 % only).

    python tools/split_batch_bench.py [--steps 200] [--splits 1 2 4] [--out file.json]
"""
import argparse
import copy
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--splits", type=int, nargs="+", default=[1, 2, 4])
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--size", type=int, default=640)
    ap.add_argument("--model", default="yolov6s")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    from yolov6_amd.utils.nms import nms_raw
    dev = torch.device("cuda:0")
    cfg, sd, model, x = bench.build_model_and_input(a, dev)
    bench.calibrate_head_bias(model, x)
    res = []
    ref = None
    for ns in a.splits:
        xs = [c.contiguous() for c in x.chunk(ns)]
        models = [model] + [copy.deepcopy(model) for _ in range(ns - 1)]
        plans = [m.compile(xi, autotune=True) for m, xi in zip(models, xs)]
        streams = [torch.cuda.Stream() for _ in range(ns)]
        outs = [None] * ns

        def step():
            for i in range(ns):
                with torch.cuda.stream(streams[i]):
                    det = plans[i].run()
                    outs[i] = (det, nms_raw(det, bench.CONF, bench.IOU, multi_label=True, max_det=bench.MAX_DET))

        torch.cuda.synchronize()
        for _ in range(a.warmup):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            step()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        det = torch.cat([o[0] for o in outs]).float()
        count = torch.cat([o[1][2] for o in outs])
        index = torch.cat([o[1][1] for o in outs])
        row = dict(splits=ns, images_per_plan=a.batch // ns, ms_per_step=round(el / a.steps * 1e3, 4),
                   images_per_sec=round(a.batch * a.steps / el, 1), kept_mean=float(count.float().mean()))
        if ref is None:
            ref = (det.clone(), count.clone(), index.clone())
        else:
            row["max_abs_det_diff_vs_one_plan"] = float((det - ref[0]).abs().max())
            row["nms_counts_equal"] = bool(torch.equal(count, ref[1]))
            row["nms_indices_equal"] = bool(torch.equal(index, ref[2]))
        res.append(row)
        print(json.dumps(row), flush=True)
        del plans, models, streams
    if a.out:
        with open(a.out, "w") as f:
            json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
