"""Where the host time of the reference-signature step goes: cProfile over `model(x)` + `non_max_suppression` (YOLOv6-S 640^2 b32).
The GPU idles from the moment the host learns the NMS counts until the first launch of the next forward, so every microsecond of
Python on that stretch is a microsecond of step time.  Usage: python tools/dropin_profile.py [steps]"""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from yolov6_amd.utils.nms import non_max_suppression  # noqa: E402


class A:
    model, batch, size = "yolov6s", 32, 640


dev = torch.device("cuda:0")
cfg, sd, model, x = bench.build_model_and_input(A, dev)
bench.calibrate_head_bias(model, x)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
for _ in range(5):
    non_max_suppression(model(x)[0], 0.03, 0.65, multi_label=True, max_det=300)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    non_max_suppression(model(x)[0], 0.03, 0.65, multi_label=True, max_det=300)
torch.cuda.synchronize()
print("ms per step", (time.perf_counter() - t0) / n * 1e3)
# host time with the GPU out of the picture: the same calls, profiled
pr = cProfile.Profile()
pr.enable()
for _ in range(n):
    non_max_suppression(model(x)[0], 0.03, 0.65, multi_label=True, max_det=300)
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(18)
