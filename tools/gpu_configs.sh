#!/usr/bin/env bash
# The other benchmarked configurations on one box: S-QA int8 / fp16 (configs[4]), L6 1280 b8 (configs[3]), the training step (configs[2]).
#   usage: tools/gpu_configs.sh <tag>
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/${1:-cfg}; mkdir -p "$OUT"
line() {  # name, bench args...
  local name=$1; shift
  timeout -k 5 240 python3 bench.py "$@" --profile-out "$OUT/ops_$name.json" > "$OUT/bench_$name.json" 2> "$OUT/bench_$name.err"
  python3 - <<PY
import json
try:
    d=json.load(open("$OUT/bench_$name.json")); seq=d.get("sequential") or {}
    print("$name", d["value"], d["unit"], d["ms_per_step"], "ms; one at a time", seq.get("value"), "frac", d["roofline"]["frac"], "attempts", d["supervisor"]["attempts"])
except Exception as e: print("$name: no result", e); print(open("$OUT/bench_$name.err").read()[-600:])
PY
}
line qa_int8 --model yolov6s_qa --int8 --no-cpu-baseline --no-train-sub --dropin-steps 0
line qa_fp16 --model yolov6s_qa --no-cpu-baseline --no-train-sub --dropin-steps 0
line l6 --model yolov6l6 --size 1280 --batch 8 --no-cpu-baseline --no-train-sub --dropin-steps 0
line train --mode train
