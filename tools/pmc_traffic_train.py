"""Aggregate the FETCH_SIZE / WRITE_SIZE passes written by tools/gpu_pmc_traffic_train.sh (bench.py --mode train) into HBM
bytes per launch, per kernel (template arguments stripped) and per roofline class:
  mfma = the kernels bench.py --mode train's `roofline` covers (forward / data-gradient convs, stem, weight-gradient GEMM),
  hbm  = the memory-bound glue (BatchNorm statistics / apply / backward, operand transposes, pools, head, loss, optimizer).
Corrections as tools/pmc_traffic.py (MI355X_MICROARCH.md, HBM section): FETCH_SIZE x2 on gfx950, WRITE_SIZE as reported."""
import csv, glob, json, os, re, sys

out = sys.argv[1]
MFMA = ("conv3x3_dma_kernel", "conv_mfma_pipe_kernel", "conv_mfma_kernel", "conv_mfma_persist_kernel", "conv1x1_stream_kernel",
        "stem_", "wgrad_kernel", "conv3x3_wreg_kernel", "wgrad_lds_kernel", "wgrad_flat_kernel", "wgrad_flat_s2_kernel", "conv_pw_kernel",
        "wgrad_stem_kernel", "dgrad_s2_kernel")


def base(name):
    name = re.sub(r"^void ", "", name)
    name = name.replace("(anonymous namespace)::", "")
    return re.split(r"[<(]", name, maxsplit=1)[0].strip()


per = {}
for counter in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(os.path.join(out, counter, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            c = per.setdefault(base(r["Kernel_Name"]), {"FETCH_SIZE": [0.0, 0], "WRITE_SIZE": [0.0, 0]})
            c[counter][0] += float(r["Counter_Value"])
            c[counter][1] += 1
kernels, classes = {}, {"mfma": [0.0, 0], "hbm": [0.0, 0]}
for k, c in sorted(per.items()):
    nf, nw = c["FETCH_SIZE"][1], c["WRITE_SIZE"][1]
    fetch = c["FETCH_SIZE"][0] * 1024 * 2
    write = c["WRITE_SIZE"][0] * 1024
    n = max(nf, nw, 1)
    kernels[k] = {"launches_sampled": n, "fetch_bytes_total_corrected_x2": round(fetch), "write_bytes_total": round(write),
                  "hbm_bytes_per_launch": round(fetch / max(nf, 1) + write / max(nw, 1))}
    cls = "mfma" if any(k.startswith(m) for m in MFMA) else "hbm"
    classes[cls][0] += fetch / max(nf, 1) * n + write / max(nw, 1) * n
    classes[cls][1] += n
print(json.dumps({"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), python bench.py --mode train --steps 2 --warmup 1 --no-autotune",
                  "corrections": "FETCH_SIZE x2 (gfx950: 128-B requests tallied as 64 B), WRITE_SIZE as reported (uncalibrated); units KB x 1024",
                  "steps_profiled": "all launches of the run (plan build + 3 steps + 3 instrumented steps)",
                  "classes": {k: {"launches_sampled": v[1], "hbm_bytes_total": round(v[0]),
                                  "hbm_bytes_per_launch": round(v[0] / max(v[1], 1))} for k, v in classes.items()},
                  "kernels": kernels}, indent=1))
