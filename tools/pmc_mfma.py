"""Aggregate an SQ counter pass (tools/gpu_pmc_mfma.sh) into per-dispatch MFMA-pipe use.

Units per MI355X_MICROARCH.md (per-instruction constants): SQ_VALU_MFMA_BUSY_CYCLES counts shader cycles summed over SIMDs
(32 per v_mfma_f32_32x32x16_f16 on the issuing SIMD); SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed
over waves; SQ_BUSY_CYCLES is per shader engine (summed over the chip's SEs / XCDs).  The figure that needs no unit guess is
mfma_busy / (n_simd_used x kernel_cycles) where kernel_cycles comes from the dispatch's own duration x an assumed clock; it is
reported next to the unit-free ratio mfma_busy / (4 x wave_cycles_per_resident_wave_slot)."""
import csv, glob, json, os, sys
from collections import defaultdict

out = sys.argv[1]
rows = defaultdict(dict)
meta = {}
for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = (r.get("Dispatch_Id") or r.get("Dispatch_ID"), r["Kernel_Name"])
        rows[k][r["Counter_Name"]] = rows[k].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        meta[k] = dict(grid=int(r.get("Grid_Size", 0) or 0), wg=int(r.get("Workgroup_Size", 0) or 0), vgpr=r.get("VGPR_Count"),
                       agpr=r.get("Accum_VGPR_Count"), sgpr=r.get("SGPR_Count"), lds=r.get("LDS_Block_Size"))
dur = {}
for f in glob.glob(os.path.join(out, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        dur[(r.get("Dispatch_Id"), r["Kernel_Name"])] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
agg = defaultdict(list)
for k, c in rows.items():
    if "conv" not in k[1]:
        continue
    agg[(k[1], meta[k]["grid"])].append((c, meta[k], dur.get(k)))
res = []
for (name, grid), lst in sorted(agg.items()):
    n = len(lst)
    c = {key: sum(x[0].get(key, 0.0) for x in lst) / n for key in lst[0][0]}
    m = lst[0][1]
    d = [x[2] for x in lst if x[2]]
    row = dict(kernel=name[:110], grid_threads=grid, wg=m["wg"], vgpr=m["vgpr"], agpr=m["agpr"], lds=m["lds"], dispatches=n,
               counters={k: round(v, 1) for k, v in c.items()})
    if d:
        row["ns_profiled"] = round(sum(d) / len(d))
    mf, wc = c.get("SQ_VALU_MFMA_BUSY_CYCLES"), c.get("SQ_WAVE_CYCLES")
    if mf and d:
        ns = sum(d) / len(d)
        # chip-wide matrix-pipe cycles available in the dispatch: 1024 SIMDs x duration x clock
        for ghz in (2.4, 2.0):
            row[f"mfma_busy_frac_at_{ghz}GHz"] = round(mf / (1024 * ns * ghz), 4)
    if mf and wc:
        row["mfma_busy_per_wave_quadcycle"] = round(mf / wc, 4)
    res.append(row)
print(json.dumps(dict(source="rocprofv3 --kernel-trace --pmc SQ_* over tools/conv_bench.py (one pass)", rows=res), indent=1))
