#!/usr/bin/env bash
# Round 3, visit v (what is left of the round's GPU budget): the headline bench at HEAD (two-stream schedule on by default), then the
# two-stream schedule of the TRAINING forward (Y6_TRAIN_FWD_STREAMS=2): bit-identity tests, training bench A/B, the training tests with it on.
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/r03v; mkdir -p $OUT
T0=$(date +%s)
lap() { echo "-- $1 done at +$(( $(date +%s) - T0 )) s"; }
Y6_AUTOTUNE_CACHE="$PWD/$OUT/autotune.cache" timeout 120 python bench.py --profile-out $OUT/bench_ops.json > $OUT/bench.json 2> $OUT/bench.err
echo "bench rc=$?"; tail -1 $OUT/bench.err | cut -c1-200
python - <<PY
import json
try:
    d=json.load(open("$OUT/bench.json"))
    print("headline", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["forward"]["ms"], d["nms"]["ms"], d["dropin_api"]["value"], d["self_check"], d.get("schedule"), d["cpu_baseline"]["value"])
except Exception as e: print("no result", e)
PY
lap headline
timeout 120 python -m pytest tests/test_gpu_training.py -m gpu -q --tb=short --timeout 100 -p no:cacheprovider -k two_stream > $OUT/pytest_train_schedule.log 2>&1
echo "pytest train schedule rc=$?"; tail -4 $OUT/pytest_train_schedule.log | cut -c1-300; grep -E "Error|assert|FAILED" $OUT/pytest_train_schedule.log | head -8 | cut -c1-300; lap tests
train() {  # name, env...
  local name=$1; shift
  env "$@" timeout 120 python bench.py --mode train > $OUT/bench_train_$name.json 2> $OUT/bench_train_$name.err
  echo "train $name rc=$?"; tail -1 $OUT/bench_train_$name.err | cut -c1-160
  python - <<PY
import json
try:
    d=json.load(open("$OUT/bench_train_$name.json"))
    print("$name", d["value"], d["ms_per_step"], d["loss"], d["memory_gb"])
except Exception as e: print("no result", e)
PY
}
train one1 Y6_TRAIN_FWD_STREAMS=1
train asap1 Y6_TRAIN_FWD_STREAMS=2 Y6_TRAIN_FWD_POLICY=asap
train alap1 Y6_TRAIN_FWD_STREAMS=2 Y6_TRAIN_FWD_POLICY=alap
train one2 Y6_TRAIN_FWD_STREAMS=1
train asap2 Y6_TRAIN_FWD_STREAMS=2 Y6_TRAIN_FWD_POLICY=asap
lap "train A/B"
Y6_TRAIN_FWD_STREAMS=2 timeout 150 python -m pytest tests/test_gpu_training.py -m gpu -q --tb=short --timeout 120 -p no:cacheprovider > $OUT/pytest_training_fwdsched_on.log 2>&1
echo "pytest training (forward schedule on) rc=$?"; tail -3 $OUT/pytest_training_fwdsched_on.log | cut -c1-300; lap "training tests"
echo done
