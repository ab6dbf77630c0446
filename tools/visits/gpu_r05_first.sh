#!/usr/bin/env bash
# Round 5, first visit (prepared at the end of round 4, after the GPU budget was spent): where every benchmarked configuration
# stands on the round's first box - per-op tables of the headline, the int8, the L6 and the training plans, the rocprofv3 kernel
# summary of the default line - so that the round's work starts from same-box numbers.  ~4 minutes of box time.
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/${1:-r05a}; mkdir -p "$OUT"
T0=$(date +%s)
lap() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
line() {  # name, bench args...
  local name=$1; shift
  timeout -k 5 200 python bench.py "$@" --profile-out "$OUT/ops_$name.json" > "$OUT/bench_$name.json" 2> "$OUT/bench_$name.err"
  python - <<PY
import json
try:
    d=json.load(open("$OUT/bench_$name.json"))
    seq=d.get("sequential") or {}
    print("$name", d["value"], d["unit"], d["ms_per_step"], "ms; one at a time", seq.get("value"), "frac", d["roofline"]["frac"])
except Exception as e: print("$name: no result", e); print(open("$OUT/bench_$name.err").read()[-600:])
PY
}
line default                                                   # the driver's line (cpu_baseline + train sub-object included)
lap default
line qa_int8 --model yolov6s_qa --int8 --no-cpu-baseline --no-train-sub --dropin-steps 0
line qa_fp16 --model yolov6s_qa --no-cpu-baseline --no-train-sub --dropin-steps 0
line l6 --model yolov6l6 --size 1280 --batch 8 --no-cpu-baseline --no-train-sub --dropin-steps 0
lap "int8 / fp16 S-QA / L6"
line train --mode train --no-cpu-baseline
lap train
R=$PWD
( cd /tmp && timeout -k 5 120 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$OUT/rocprof_default" -o b -- python "$R/bench.py" --no-cpu-baseline --no-train-sub --dropin-steps 0 --windows 1 --steps 50 --no-verify > "$R/$OUT/rocprof_default.json" 2> "$R/$OUT/rocprof_default.err" )
find "$OUT/rocprof_default" -name "*kernel_trace.csv" -delete 2>/dev/null
lap "rocprof rc=$?"
echo done
