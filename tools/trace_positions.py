#!/usr/bin/env python3
"""Per-position timeline of a step from a rocprofv3 --kernel-trace CSV of `bench.py --inflight 1` with a Y6_SCHED_STREAMS=1 plan (one
stream, plan order): the dispatches are cut into steps at every stem / fused stem kernel; for each position inside a step: kernel,
mean duration, mean gap from the previous kernel's end to this one's start (launch + dependency overhead), over the last N steps.
    python tools/trace_positions.py <dir> [steps] > positions.json"""
import csv, glob, json, os, sys
d = sys.argv[1]
last = int(sys.argv[2]) if len(sys.argv) > 2 else 20
f = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True))[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
steps, cur = [], None
for r in rows:
    n = r["Kernel_Name"]
    if "fused_stem_s2_kernel" in n or ("stem_mfma" in n and cur is not None and len(cur) > 40):
        if cur:
            steps.append(cur)
        cur = []
    if cur is not None:
        cur.append(r)
if cur:
    steps.append(cur)
from collections import Counter
L = Counter(len(s) for s in steps).most_common(1)[0][0]
steps = [s for s in steps if len(s) == L][-last:]
out = []
for i in range(L):
    durs = [(int(s[i]["End_Timestamp"]) - int(s[i]["Start_Timestamp"])) / 1e3 for s in steps]
    gaps = [(int(s[i]["Start_Timestamp"]) - int(s[i - 1]["End_Timestamp"])) / 1e3 for s in steps] if i else [0.0]
    name = steps[0][i]["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:70]
    out.append({"pos": i, "kernel": name, "dur_us": round(sum(durs) / len(durs), 2), "gap_us": round(sum(gaps) / len(gaps), 2),
                "lds": steps[0][i].get("LDS_Block_Size"), "grid": steps[0][i].get("Grid_Size"), "wg": steps[0][i].get("Workgroup_Size")})
tot_d, tot_g = sum(o["dur_us"] for o in out), sum(o["gap_us"] for o in out)
print(json.dumps({"steps_used": len(steps), "kernels_per_step": L, "sum_dur_us": round(tot_d, 1), "sum_gap_us": round(tot_g, 1), "positions": out}, indent=0))
