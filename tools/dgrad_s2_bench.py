#!/usr/bin/env python
"""Micro-bench of csrc/dgrad_s2.hip on the four stride-2 blocks of YOLOv6-S at b64 (and the neck's 3x3-only downsamples).
   usage: python tools/dgrad_s2_bench.py   (env Y6_DGRAD_S2_WAVES)"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from yolov6_amd import _lib  # noqa: E402
from yolov6_amd.engine import TRef, _null_tensor  # noqa: E402

SHAPES = [(64, 32, 160, True), (128, 64, 80, True), (256, 128, 40, True), (512, 256, 20, True), (64, 64, 80, False), (128, 128, 40, False)]


def main():
    lib = _lib.load()
    B = 64
    for M, N, Ho, has1 in SHAPES:
        dy3 = torch.randn((B, Ho, Ho, M), device="cuda:0").half()
        dy1 = torch.randn((B, Ho, Ho, M), device="cuda:0").half()
        dx = torch.zeros((B, 2 * Ho, 2 * Ho, N), device="cuda:0").half()
        w3 = torch.randn(int(lib.y6_pack_job_elems(1, M, N, 3)), device="cuda:0").half()
        w1 = torch.randn(int(lib.y6_pack_job_elems(1, M, N, 1)), device="cuda:0").half()
        d = _lib.DgradS2Desc()
        d.dy3 = TRef(dy3, B, Ho, Ho, M, M, 0).ct()
        d.dy1 = TRef(dy1, B, Ho, Ho, M, M, 0).ct() if has1 else _null_tensor()
        d.dx = TRef(dx, B, 2 * Ho, 2 * Ho, N, N, 0).ct()
        d.w3_packed, d.w1_packed, d.accumulate = w3.data_ptr(), (w1.data_ptr() if has1 else None), 1
        s = _lib.current_stream_ptr()
        for _ in range(3):
            _lib.check(lib.y6_dgrad_s2(C.byref(d), s), "dgrad_s2")
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 20
        e0.record()
        for _ in range(n):
            lib.y6_dgrad_s2(C.byref(d), s)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / n * 1e3
        fl = 2.0 * B * Ho * Ho * M * N * (10 if has1 else 9)
        by = 2.0 * B * Ho * Ho * M * (2 if has1 else 1) + 2.0 * 4 * B * Ho * Ho * N * 2
        print(f"M{M} N{N} {Ho}x{Ho} k1={int(has1)}: {us:7.1f} us  {fl / us / 1e6:6.0f} TF/s  {by / us / 1e6:5.2f} TB/s", flush=True)


if __name__ == "__main__":
    main()
