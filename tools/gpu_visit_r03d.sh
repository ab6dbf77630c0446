#!/usr/bin/env bash
# Round 3, visit d: the fused head tail (pred convs + decode in one launch): op test, model / parity tests, same-box A/B.
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/r03d; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q --tb=short --timeout 600 -p no:cacheprovider -k "head_pred_decode or head_decode" -s > $OUT/pytest_head.log 2>&1
echo "pytest head rc=$?" | tee -a $OUT/pytest_head.log; tail -15 $OUT/pytest_head.log
timeout 1500 python -m pytest tests/test_gpu_model.py tests/test_gpu_parity_bench.py tests/test_gpu_training.py tests/test_gpu_train_parity.py tests/test_gpu_dropin.py \
  -m gpu -q --tb=short --timeout 900 -p no:cacheprovider -k "not block_training_graph" > $OUT/pytest_model.log 2>&1
echo "pytest model rc=$?" | tee -a $OUT/pytest_model.log; tail -25 $OUT/pytest_model.log
for mode in fused unfused; do
  if [ $mode = unfused ]; then export Y6_HEAD_NO_FUSE=1; else unset Y6_HEAD_NO_FUSE; fi
  timeout 600 python bench.py --no-cpu-baseline --dropin-steps 0 --profile-out $OUT/bench_ops_$mode.json > $OUT/bench_$mode.json 2> $OUT/bench_$mode.err
  echo "bench $mode rc=$?"; cut -c1-200 $OUT/bench_$mode.json; python - <<PY
import json
d=json.load(open("$OUT/bench_$mode.json"))
print("$mode", d["value"], d["ms_per_step"], {k:(v["ms"],v["launches"]) for k,v in d["breakdown"].items()})
PY
done
