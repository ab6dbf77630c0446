#!/usr/bin/env python3
"""Calibration only (not part of the product or of bench.py): what does this MI355X box deliver for
  (a) a plain device copy (achievable HBM bandwidth),
  (b) PyTorch-ROCm / MIOpen fp16 channels_last conv2d on the hot layer shapes,
  (c) the reference-equivalent eager forward (oracle graph, torch ops, fp16, channels_last) of YOLOv6-S b32?
This is how the reference itself would run on MI355X (PyTorch + MIOpen), i.e. the number to beat."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as F


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    dev = "cuda:0"
    out = {}
    # (a) copy bandwidth
    x = torch.empty(288 * 1024 * 1024 // 2, dtype=torch.float16, device=dev)
    y = torch.empty_like(x)
    ms = timeit(lambda: y.copy_(x))
    out["copy_288MB"] = dict(ms=round(ms, 4), gbs=round(2 * x.numel() * 2 / ms / 1e6, 1))
    # (b) MIOpen conv on layer shapes (Cin, Cout, k, s, H, W, B)
    layers = [(64, 64, 3, 1, 160, 160, 32), (128, 128, 3, 1, 80, 80, 32), (256, 256, 3, 1, 40, 40, 32),
              (512, 512, 3, 1, 20, 20, 32), (64, 128, 3, 2, 160, 160, 32), (64, 64, 1, 1, 160, 160, 32),
              (3, 32, 3, 2, 640, 640, 32)]
    torch.backends.cudnn.benchmark = True
    convs = []
    for (ci, co, k, s, H, W, B) in layers:
        xi = torch.randn(B, ci, H, W, device=dev).half().contiguous(memory_format=torch.channels_last)
        w = torch.randn(co, ci, k, k, device=dev).half().contiguous(memory_format=torch.channels_last)
        b = torch.randn(co, device=dev).half()
        try:
            ms = timeit(lambda: F.relu(F.conv2d(xi, w, b, stride=s, padding=k // 2)))
            fl = 2.0 * B * (H // s) * (W // s) * co * ci * k * k
            convs.append(dict(layer=(ci, co, k, s, H, W, B), ms=round(ms, 4), tflops=round(fl / ms / 1e9, 1)))
        except Exception as e:  # noqa
            convs.append(dict(layer=(ci, co, k, s, H, W, B), error=str(e)[:100]))
        print(convs[-1], flush=True)
    out["miopen_conv_relu"] = convs
    # (c) reference-equivalent eager forward
    from oracle.model_oracle import Oracle, deploy_state_dict
    from yolov6_amd.configs import get_config
    from yolov6_amd.models.yolo import build_model
    from yolov6_amd.utils import synth
    cfg = get_config("yolov6s")
    m = build_model(cfg, 80, "cpu").eval()
    sd = deploy_state_dict(cfg, synth.synth_state_dict(m.state_dict(), 0), 80)
    sd = {k: v.to(dev).half() for k, v in sd.items()}
    orc = Oracle(cfg, {}, 80)
    orc.sd = sd                       # keep fp16 device tensors (Oracle.__init__ would upcast)
    orc.q = lambda t: t               # native fp16 math
    x32 = synth.synth_images(32, 640, seed=0).to(dev).half().contiguous(memory_format=torch.channels_last)

    def fwd():
        with torch.no_grad():
            return orc.forward_device(x32)
    try:
        ms = timeit(fwd, iters=10, warm=3)
        out["eager_fp16_channels_last_forward_b32"] = dict(ms=round(ms, 3), img_s=round(32 / ms * 1e3, 1))
    except Exception as e:  # noqa
        out["eager_fp16_channels_last_forward_b32"] = dict(error=repr(e)[:300])
    print(json.dumps(out), flush=True)
    if len(sys.argv) > 1:
        json.dump(out, open(sys.argv[1], "w"), indent=1)


if __name__ == "__main__":
    main()
