#!/usr/bin/env bash
# Round 3, visit w (the round's last GPU seconds): one more same-box A/B of the inference plan's two-stream schedule.
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/r03w; mkdir -p $OUT
run() {
  local name=$1; shift
  env "$@" Y6_AUTOTUNE_CACHE="$PWD/$OUT/autotune.cache" timeout 40 python bench.py --no-cpu-baseline --dropin-steps 0 > $OUT/bench_$name.json 2> $OUT/bench_$name.err
  python - <<PY
import json
try:
    d=json.load(open("$OUT/bench_$name.json")); print("$name", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["forward"]["ms"], d.get("schedule"))
except Exception as e: print("no result", e)
PY
}
run two1 Y6_SCHED_STREAMS=2
run one1 Y6_SCHED_STREAMS=1
run two2 Y6_SCHED_STREAMS=2
run one2 Y6_SCHED_STREAMS=1
echo done
