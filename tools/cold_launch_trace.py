#!/usr/bin/env python3
"""Where do the +25 us of a cold kernel function go?  Probe build of conv_wreg.hip (tools/build_probe_libs.py --wreg 1; Y6_LIB_PATH):
every block writes its start / end on the 100 MHz counter, block 0 / thread 0 a shader-clock timeline.  YOLOv6-S b32, shape-derived
plan; op 3 (wregs2_p3) then op 4 (wreg_p7), each into its own trace buffer - once COLD (ops 0-2 in front) and once WARM (op 5 in front)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench

class A: model="yolov6s"; batch=32; size=640
dev = torch.device("cuda:0")
cfg, sd, model, x = bench.build_model_and_input(A, dev)
os.environ["Y6_SCHED_STREAMS"] = "1"
plan = model.compile(x, autotune=False)
plan.run(); torch.cuda.synchronize()
bufA = torch.zeros(4096, dtype=torch.int64, device=dev)
bufB = torch.zeros(4096, dtype=torch.int64, device=dev)

def traced(pre):
    out = []
    for rep in range(4):
        os.environ.pop("Y6_CONV_TRACE", None)
        for r in pre:
            plan.run_range(*r)
        bufA.zero_(); bufB.zero_()
        os.environ["Y6_CONV_TRACE"] = str(bufA.data_ptr()); plan.run_range(3, 4)
        os.environ["Y6_CONV_TRACE"] = str(bufB.data_ptr())
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); plan.run_range(4, 5); e1.record(); torch.cuda.synchronize()
        os.environ.pop("Y6_CONV_TRACE", None)
        a, b = bufA.cpu(), bufB.cpu()
        ba = a[1024:3072].view(1024, 2); ba = ba[ba[:, 0] > 0]
        bb = b[1024:3072].view(1024, 2); bb = bb[bb[:, 0] > 0]
        prev_end = int(ba[:, 1].max())
        st = (bb[:, 0] - prev_end).float() / 100.0
        en = (bb[:, 1] - prev_end).float() / 100.0
        life = en - st
        q = lambda v, f: round(float(v.sort().values[int(f * (len(v) - 1))]), 2)
        tl = b[:512].view(256, 2).tolist()
        tl = [(ts, tag) for ts, tag in tl if tag not in (0, 90, 91)]
        deltas = [(int(t2), int(a2 - a1)) for (a1, _), (a2, t2) in zip(tl[:40], tl[1:41])]
        out.append(dict(event_us=round(e0.elapsed_time(e1) * 1e3, 1), blocks=int(bb.shape[0]),
                        start_after_prev_end_us=dict(min=q(st, 0), p10=q(st, .1), median=q(st, .5), p90=q(st, .9), max=q(st, 1)),
                        end_after_prev_end_us=dict(min=q(en, 0), median=q(en, .5), max=q(en, 1)),
                        lifetime_us=dict(min=q(life, 0), median=q(life, .5), max=q(life, 1)),
                        block0_cycles_between_tags=deltas))
    return out[1:]

res = {"cold (ops 0-2, op 3 | op 4)": traced([(0, 3)]), "warm (op 5, op 3 | op 4)": traced([(5, 6)])}
print(json.dumps(res))
