#!/usr/bin/env python3
"""Micro-benchmark of single conv layers per kernel variant (hipEvent timing through the plan
executor).  Designed to run under rocprofv3 (--kernel-trace / --pmc) as well:

    python tools/conv_bench.py --layers 256,256,3,1,40,40,32 64,64,3,1,160,160,32 --variants 1 2 5 6 --iters 10
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from yolov6_amd import _lib  # noqa: E402
from yolov6_amd.engine import PlanBuilder, TRef  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layers", nargs="+", default=["256,256,3,1,40,40,32"])
    ap.add_argument("--variants", nargs="*", default=None, help="variant names (wreg_p7 ...) or indices")
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--out", default=None)
    ap.add_argument("--data", default="rand", choices=["rand", "relu", "zeros"],
                    help="input activations: N(0,1), the same clamped at 0 (half zeros, as behind a ReLU), all zeros - the chip clocks to its power budget")
    a = ap.parse_args()
    lib = _lib.load()
    names = [lib.y6_conv_variant_name(i).decode() for i in range(lib.y6_conv_variants())]
    dev = "cuda:0"
    res = []
    for spec in a.layers:
        cin, cout, k, s, H, W, B = (int(v) for v in spec.split(","))
        x = torch.randn((B, H, W, cin), device=dev).half()
        if a.data == "relu":
            x = x.clamp(min=0)
        elif a.data == "zeros":
            x = x * 0
        w = torch.randn((cout, cin, k, k)) / (cin * k * k) ** 0.5
        b = torch.randn((cout,)) * 0.1
        xr = TRef(x, B, H, W, cin, cin, 0)
        chosen = [names.index(v) if v in names else int(v) for v in a.variants] if a.variants else range(1, len(names))
        for v in chosen:
            pb = PlanBuilder(dev)
            pb.force_variant = v
            try:
                pb.conv(xr, w, b, stride=s, act="relu")
                plan = pb.finalize(None, autotune=False)
                plan.run()
                torch.cuda.synchronize()
            except RuntimeError as e:
                if "does not support" in str(e) or "unsupported" in str(e):
                    continue
                raise
            ms = plan.profile(a.iters)[0]["ms"]
            fl = 2.0 * B * (H // s) * (W // s) * cout * cin * k * k
            row = dict(layer=spec, variant=names[v], ms=round(ms, 5), tflops=round(fl / ms / 1e9, 1), data=a.data)
            res.append(row)
            print(json.dumps(row), flush=True)
    if a.out:
        with open(a.out, "w") as f:
            json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
