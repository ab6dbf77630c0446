#!/usr/bin/env python3
"""Per-class and per-kernel summary of a rocprofv3 --kernel-trace --stats run of bench.py: kernel_stats.csv -> JSON with, per kernel
name, calls / average ns / share, and per kernel class (the classes of bench.py's `breakdown`) the time per step.
    python tools/rocprof_classes.py <dir with *kernel_stats.csv> <steps profiled incl. warm-up> > summary.json"""
import csv, glob, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

d, steps = sys.argv[1], float(sys.argv[2])


def classify(name):
    import re
    if "conv1x1_stream_kernel" in name or "conv_pw_kernel" in name:   # (conv_pw_kernel also runs the two ConvTranspose2d layers: two launches of the class per step)
        return "conv1x1s1"
    m = re.search(r"conv3x3_wreg_kernel<\d+, \d+, \d+, (\d)", name)
    if m:
        return "conv3x3s%s" % m.group(1)
    if "conv3x3_dma_kernel" in name:
        m = re.search(r"conv3x3_dma_kernel<(?:[^<>]*?, )?(?:true|false), (\d)(?:, (?:true|false))?>", name)
        return "conv3x3s%s" % (m.group(1) if m else "1")
    m = re.search(r"conv_mfma_pipe_kernel<\d+, \d+, \d+, \d+, (\d)", name)
    if m:
        return "conv3x3s%s" % m.group(1)
    m = re.search(r"conv_mfma_kernel<\d+, \d+, (\d), (\d)(?:, -?\d+)?>", name)   # (<CF, PF, KS, ST[, ACT]>)
    if m:
        return "conv%sx%ss%s" % (m.group(1), m.group(1), m.group(2))
    for key, cls in (("fused_pw_s2", "pw_s2"), ("fused_stem_s2", "stem_s2"), ("stem_", "stem"), ("pred_decode", "pred_decode"), ("head_decode", "decode"),
                     ("nms_", "nms"), ("sppf", "sppf"), ("convt", "convt")):
        if key in name:
            return cls
    return "other"


f = sorted(glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True))[0]
rows = list(csv.DictReader(open(f)))
kern, cls = [], {}
for r in rows:
    name, calls, tot, avg = r["Name"], int(r["Calls"]), float(r["TotalDurationNs"]), float(r["AverageNs"])
    c = classify(name)
    kern.append({"kernel": name[:160], "class": c, "calls": calls, "avg_us": round(avg / 1e3, 2), "total_ms": round(tot / 1e6, 3)})
    e = cls.setdefault(c, {"calls": 0, "total_ms": 0.0})
    e["calls"] += calls
    e["total_ms"] += tot / 1e6
for c, e in cls.items():
    e["ms_per_step"] = round(e["total_ms"] / steps, 4)
    e["launches_per_step"] = round(e["calls"] / steps, 2)
    e["total_ms"] = round(e["total_ms"], 3)
print(json.dumps({"source": os.path.relpath(f), "steps_profiled": steps, "classes": dict(sorted(cls.items())),
                  "kernels": sorted(kern, key=lambda k: -k["total_ms"])[:60]}, indent=1))
