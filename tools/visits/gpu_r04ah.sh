#!/usr/bin/env bash
# Round 4, visit ah: with two steps in flight, does the in-plan two-stream schedule still pay?  + L6 / int8 lines with the new default
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/${1:-r04ah}; mkdir -p "$OUT"
run() {  # name, env/args
  local name=$1; shift
  env "$@" timeout 300 python bench.py --no-cpu-baseline --no-train-sub --dropin-steps 0 $EXTRA > "$OUT/bench_$name.json" 2> "$OUT/bench_$name.err"
  python - <<PY
import json
try:
    d=json.load(open("$OUT/bench_$name.json"))
    print("$name", d["value"], d["ms_per_step"], "seq", d["sequential"]["value"], d["roofline"]["frac"], d.get("schedule") and d["schedule"]["side_ops"])
except Exception as e: print("$name: no result", e); print(open("$OUT/bench_$name.err").read()[-600:])
PY
}
EXTRA=""
run sched2_a Y6_DUMMY=1
run sched1_a Y6_SCHED_STREAMS=1
run sched2_b Y6_DUMMY=1
run sched1_b Y6_SCHED_STREAMS=1
EXTRA="--model yolov6l6 --size 1280 --batch 8"; run l6 Y6_DUMMY=1
EXTRA="--model yolov6s_qa --int8"; run qa_int8 Y6_DUMMY=1
EXTRA="--model yolov6s_qa"; run qa_fp16 Y6_DUMMY=1
echo done
