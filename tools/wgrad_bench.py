"""Times the weight-gradient kernels on the YOLOv6-S b64 layer shapes: the flat-index block-tiled kernel (wgrad_flat.hip, round 6),
the row-ring NHWC kernel (Y6_WGRAD_FLAT=0) and the plane-fed kernel (y6_wgrad) plus the transposes it needs.
Usage: python tools/wgrad_bench.py [out.json]"""
import ctypes as C
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from yolov6_amd import _lib  # noqa: E402
from yolov6_amd.engine import TRef  # noqa: E402

DEV = torch.device("cuda:0")
lib = _lib.load()
ws = torch.empty(256 << 20, dtype=torch.uint8, device=DEV)


def stream():
    return _lib.current_stream_ptr()


def timeit(fn, n=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3   # us


def rup(v, m):
    return (v + m - 1) // m * m


def transpose(src, sy, sx, oy, ox, R, Q):
    dst = torch.empty(src.C * src.B * R * Q, dtype=torch.float16, device=DEV)
    d = _lib.WgradTDesc()
    d.src = src.ct()
    d.sy, d.sx, d.oy, d.ox, d.R, d.Q = sy, sx, oy, ox, R, Q
    d.dst = dst.data_ptr()
    return dst, lambda: _lib.check(lib.y6_wgrad_transpose(C.byref(d), stream()), "t")


SHAPES = [(64, 64, 160, 64), (128, 128, 80, 64), (256, 256, 40, 64), (512, 512, 20, 64), (64, 64, 80, 64), (128, 128, 40, 64),
          (256, 256, 20, 64), (256, 128, 40, 64), (64, 80, 80, 64)]
rows = []
for (cin, cout, hw, B) in SHAPES:
    x = TRef(torch.randn((B, hw, hw, cin), device=DEV).half(), B, hw, hw, cin, cin, 0)
    cpad = rup(cout, 8)
    dy = TRef(torch.randn((B, hw, hw, cpad), device=DEV).half(), B, hw, hw, cpad, cpad, 0)
    for K in (3, 1):
        T = K * K
        out = torch.zeros(cout * cin * T, dtype=torch.float32, device=DEV)
        w = _lib.WgradNhwcDesc()
        w.ksize, w.dy, w.x, w.M, w.N = K, dy.ct(), x.ct(), cout, cin
        w.out = out.data_ptr()
        w.sm, w.sn, w.st = cin * T, T, 1
        w.workspace, w.workspace_bytes = ws.data_ptr(), ws.numel()
        flops = 2.0 * cout * cin * T * B * hw * hw
        r = dict(cin=cin, cout=cout, hw=hw, B=B, k=K, gflop=round(flops / 1e9, 1))
        for name, env in (("flat", "1"), ("ring", "0")):
            os.environ["Y6_WGRAD_FLAT"] = env
            if lib.y6_wgrad_nhwc_supported(C.byref(w)) == 1:
                r[f"{name}_us"] = timeit(lambda: _lib.check(lib.y6_wgrad_nhwc(C.byref(w), stream()), "wgrad_nhwc"))
                r[f"{name}_tflops"] = flops / r[f"{name}_us"] / 1e6
        os.environ.pop("Y6_WGRAD_FLAT", None)
        if cpad == cout:
            Q = rup(hw, 16)
            a, ta = transpose(dy, 1, 1, 0, 0, hw, Q)
            if K == 3:
                p, tp = transpose(x, 1, 1, -1, 0, hw + 2, Q)
                planes = [(p, hw + 2, ky) for ky in range(3)]
                mode = _lib.WG_3X3S1
            else:
                p, tp = transpose(x, 1, 1, 0, 0, hw, Q)
                planes = [(p, hw, 0)]
                mode = _lib.WG_1X1
            ta(), tp()
            wd = _lib.WgradDesc()
            wd.mode, wd.a = mode, a.data_ptr()
            wd.M, wd.N, wd.B, wd.Q, wd.rows, wd.a_rows = cout, cin, B, Q, hw, hw
            wd.a_channels, wd.plane_channels = cout, cin
            for i, (t, prow, drow) in enumerate(planes):
                wd.plane[i], wd.plane_rows[i], wd.drow[i] = t.data_ptr(), prow, drow
            wd.out = out.data_ptr()
            wd.sm, wd.sn, wd.st = cin * T, T, 1
            wd.workspace, wd.workspace_bytes = ws.data_ptr(), ws.numel()
            r["planes_us"] = timeit(lambda: _lib.check(lib.y6_wgrad(C.byref(wd), stream()), "wgrad"))
            r["transpose_a_us"] = timeit(ta)
            r["transpose_x_us"] = timeit(tp)
            r["planes_tflops"] = flops / r["planes_us"] / 1e6
        rows.append(r)
        print({k: (round(v, 1) if isinstance(v, float) else v) for k, v in r.items()}, flush=True)
if len(sys.argv) > 1:
    json.dump(rows, open(sys.argv[1], "w"), indent=1)
