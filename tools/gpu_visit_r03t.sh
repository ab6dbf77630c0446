#!/usr/bin/env bash
# Round 3, visit t: weight-gradient work of the backward plan on a side stream (Y6_SIDE_STREAM=1): training tests with it on,
# then the training bench A/B, alternating.
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/r03t; mkdir -p $OUT
Y6_SIDE_STREAM=1 timeout 900 python -m pytest tests/test_gpu_training.py tests/test_gpu_loss.py -m gpu -q --tb=short --timeout 600 -p no:cacheprovider > $OUT/pytest_training_side.log 2>&1
rc=$?; echo "pytest training (side stream) rc=$rc"; tail -4 $OUT/pytest_training_side.log | cut -c1-300
if [ $rc -ne 0 ]; then grep -E "Error|assert|FAILED" $OUT/pytest_training_side.log | head -20 | cut -c1-300; fi
run() {  # name, env...
  local name=$1; shift
  env "$@" timeout 600 python bench.py --mode train > $OUT/bench_train_$name.json 2> $OUT/bench_train_$name.err
  echo "train $name rc=$?"; tail -1 $OUT/bench_train_$name.err | cut -c1-200
  python - <<PY
import json
try:
    d=json.load(open("$OUT/bench_train_$name.json"))
    print("$name", d["value"], d["ms_per_step"], d["loss"], d["memory_gb"])
except Exception as e: print("no result", e)
PY
}
run one1 Y6_SIDE_STREAM=0
run side1 Y6_SIDE_STREAM=1
run one2 Y6_SIDE_STREAM=0
run side2 Y6_SIDE_STREAM=1
echo done
