"""Aggregate the FETCH_SIZE / WRITE_SIZE passes written by tools/gpu_pmc_traffic.sh into per-kernel-class HBM
bytes per launch.  Corrections per MI355X_MICROARCH.md (HBM section): rocprofv3 reports both in KiB-like units of
1024 B... (FETCH_SIZE = TCC_EA0_RDREQ x 64 B expressed in KB) and on gfx950 FETCH_SIZE counts 128-byte requests
as 64 B, so wide coalesced reads are DOUBLED; WRITE_SIZE is taken as reported (uncalibrated)."""
import csv, glob, json, os, re, sys

out = sys.argv[1]


def classify(name):
    if "conv1x1_stream_kernel" in name or "conv_pw_kernel" in name:
        return "conv1x1s1"
    m = re.search(r"conv3x3_wreg_kernel<\d+, \d+, \d+, (\d)[,>]", name)   # conv_wreg.hip: <PF, WC, WP, stride>
    if m:
        return "conv3x3s%s" % m.group(1)
    if "conv3x3_dma_kernel" in name:     # conv_dma.hip: <..., I8, stride[, resident weights]> (builds before the flag end at the stride)
        m = re.search(r"conv3x3_dma_kernel<(?:[^<>]*?, )?(?:true|false), (\d)(?:, (?:true|false))?>", name)
        return "conv3x3s%s" % (m.group(1) if m else "1")
    m = re.search(r"conv_mfma_pipe_kernel<\d+, \d+, \d+, \d+, (\d)", name)
    if m:
        return "conv3x3s%s" % m.group(1)
    if "conv_mfma_pipe_kernel" in name:
        return "conv3x3s1"
    m = re.search(r"conv_mfma_persist_kernel<\d+, \d+, (\d)>", name)
    if m:
        return "conv3x3s%s" % m.group(1)
    m = re.search(r"conv_mfma_kernel<\d+, \d+, (\d), (\d)(?:, -?\d+)?>", name)   # (<CF, PF, KS, ST[, ACT]>)
    if m:
        return "conv%sx%ss%s" % (m.group(1), m.group(1), m.group(2))
    for key, cls in (("stem_", "stem"), ("head_decode", "decode"), ("nms_", "nms"), ("sppf", "sppf")):
        if key in name:
            return cls
    return "other"


res = {}
for counter in ("FETCH_SIZE", "WRITE_SIZE"):
    files = glob.glob(os.path.join(out, counter, "**", "*counter_collection.csv"), recursive=True)
    for f in files:
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            c = res.setdefault(classify(r["Kernel_Name"]), {"FETCH_SIZE": [0.0, 0], "WRITE_SIZE": [0.0, 0]})
            c[counter][0] += float(r["Counter_Value"])
            c[counter][1] += 1
summary = {}
for cls, c in sorted(res.items()):
    nf, nw = c["FETCH_SIZE"][1], c["WRITE_SIZE"][1]
    fetch_kb = c["FETCH_SIZE"][0] / nf if nf else 0.0
    write_kb = c["WRITE_SIZE"][0] / nw if nw else 0.0
    summary[cls] = {
        "launches_sampled": nf,
        "fetch_bytes_per_launch_raw": round(fetch_kb * 1024),
        "fetch_bytes_per_launch_corrected_x2": round(fetch_kb * 1024 * 2),
        "write_bytes_per_launch": round(write_kb * 1024),
        "hbm_bytes_per_launch": round(fetch_kb * 1024 * 2 + write_kb * 1024),
    }
print(json.dumps({"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), python bench.py --steps 2 --warmup 1",
                  "corrections": "FETCH_SIZE x2 (gfx950: 128-B requests tallied as 64 B), WRITE_SIZE as reported (uncalibrated); units KB x 1024",
                  "classes": summary}, indent=1))
