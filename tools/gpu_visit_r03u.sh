#!/usr/bin/env bash
# Round 3, visit u (the last GPU minutes of the round): the two-stream schedule of the inference plan (yolov6_amd/schedule.py,
# y6_plan_set_schedule) - bit-identity tests, then the headline bench A/B (one stream / alap with three margins / asap), sharing
# one autotune cache; then the model / drop-in / family tests with the schedule on; last a fresh rocprofv3 of the training step.
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/r03u; mkdir -p $OUT
T0=$(date +%s)
lap() { echo "-- $1 done at +$(( $(date +%s) - T0 )) s"; }
timeout 150 python -m pytest tests/test_gpu_model.py -m gpu -q --tb=short --timeout 120 -p no:cacheprovider -k two_stream > $OUT/pytest_schedule.log 2>&1
echo "pytest schedule rc=$?"; tail -5 $OUT/pytest_schedule.log | cut -c1-300; grep -E "Error|assert|FAILED" $OUT/pytest_schedule.log | head -8 | cut -c1-300; lap tests
run() {  # name, env...
  local name=$1; shift
  env "$@" Y6_AUTOTUNE_CACHE="$PWD/$OUT/autotune.cache" timeout 90 python bench.py --no-cpu-baseline --dropin-steps 20 > $OUT/bench_$name.json 2> $OUT/bench_$name.err
  echo "bench $name rc=$?"; tail -1 $OUT/bench_$name.err | cut -c1-200
  python - <<PY
import json
try:
    d=json.load(open("$OUT/bench_$name.json"))
    print("$name", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["forward"]["ms"], d["nms"]["ms"], d["dropin_api"]["value"], d["self_check"], d.get("schedule"))
except Exception as e: print("no result", e)
PY
}
run one1 Y6_SCHED_STREAMS=1
run alap2 Y6_SCHED_STREAMS=2 Y6_SCHED_POLICY=alap Y6_SCHED_MARGIN=2.0
run asap Y6_SCHED_STREAMS=2 Y6_SCHED_POLICY=asap
run alap1 Y6_SCHED_STREAMS=2 Y6_SCHED_POLICY=alap Y6_SCHED_MARGIN=1.0
run alap4 Y6_SCHED_STREAMS=2 Y6_SCHED_POLICY=alap Y6_SCHED_MARGIN=4.0
run one2 Y6_SCHED_STREAMS=1
run alap2b Y6_SCHED_STREAMS=2 Y6_SCHED_POLICY=alap Y6_SCHED_MARGIN=2.0
lap benches
Y6_SCHED_STREAMS=2 timeout 120 python -m pytest tests/test_gpu_model.py tests/test_gpu_dropin.py -m gpu -q --tb=short --timeout 100 -p no:cacheprovider -x > $OUT/pytest_model_sched_on.log 2>&1
echo "pytest model+dropin (schedule on) rc=$?"; tail -3 $OUT/pytest_model_sched_on.log | cut -c1-300; lap "model tests"
( cd /tmp && timeout 70 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/$OUT/prof_train" -o train -- python "$OLDPWD/bench.py" --mode train --steps 5 --warmup 2 --no-autotune > "$OLDPWD/$OUT/prof_train.json" 2> "$OLDPWD/$OUT/prof_train.err" )
echo "trainprof rc=$?"; f=$(find "$OUT/prof_train" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -8 "$f" | cut -c1-200
find "$OUT/prof_train" -name "*kernel_trace.csv" -size +20M -delete; lap trainprof
Y6_SCHED_STREAMS=2 timeout 100 python -m pytest tests/test_gpu_families.py -m gpu -q --tb=short --timeout 90 -p no:cacheprovider -x > $OUT/pytest_families_sched_on.log 2>&1
echo "pytest families (schedule on) rc=$?"; tail -3 $OUT/pytest_families_sched_on.log | cut -c1-300; lap families
echo done
