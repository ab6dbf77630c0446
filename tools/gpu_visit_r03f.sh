#!/usr/bin/env bash
# Round 3, visit f: fused pairs after the fixes (LDS-resident stem weights, window prefetch, two pixel fragments per wave, three-deep
# weight ring); BatchNorm backward v2 (8 channels per thread); tests, same-box A/B of both.
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/r03f; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q --tb=short --timeout 600 -p no:cacheprovider -k "fused or stem" > $OUT/pytest_fused.log 2>&1
echo "pytest fused rc=$?" | tee -a $OUT/pytest_fused.log; tail -6 $OUT/pytest_fused.log
timeout 1500 python -m pytest tests/test_gpu_training.py tests/test_gpu_train_ops.py -m gpu -q --tb=short --timeout 900 -p no:cacheprovider > $OUT/pytest_train.log 2>&1
echo "pytest train rc=$?" | tee -a $OUT/pytest_train.log; tail -12 $OUT/pytest_train.log | cut -c1-300
timeout 1500 python -m pytest tests/test_gpu_model.py tests/test_gpu_parity_bench.py tests/test_gpu_dropin.py tests/test_gpu_train_parity.py tests/test_gpu_families.py \
  -m gpu -q --tb=short --timeout 900 -p no:cacheprovider -s > $OUT/pytest_model.log 2>&1
echo "pytest model rc=$?" | tee -a $OUT/pytest_model.log; grep -v "^{" $OUT/pytest_model.log | tail -25 | cut -c1-400
for mode in fused nofuse_s2; do
  unset Y6_HEAD_NO_FUSE Y6_NO_FUSE_S2
  [ $mode = nofuse_s2 ] && export Y6_NO_FUSE_S2=1
  timeout 600 python bench.py --no-cpu-baseline --dropin-steps 0 --profile-out $OUT/bench_ops_$mode.json > $OUT/bench_$mode.json 2> $OUT/bench_$mode.err
  echo "bench $mode rc=$?"; tail -2 $OUT/bench_$mode.err | cut -c1-300; python - <<PY
import json
try:
    d=json.load(open("$OUT/bench_$mode.json"))
    print("$mode", d["value"], d["ms_per_step"], {k:(v["ms"],v["launches"]) for k,v in d["breakdown"].items()})
except Exception as e: print("$mode: no result", e)
PY
done
unset Y6_NO_FUSE_S2
for mode in v2 v1; do
  unset Y6_BNACT_BWD_V1
  [ $mode = v1 ] && export Y6_BNACT_BWD_V1=1
  timeout 900 python bench.py --mode train --profile-out $OUT/train_ops_$mode.json > $OUT/bench_train_$mode.json 2> $OUT/bench_train_$mode.err
  echo "train $mode rc=$?"; tail -2 $OUT/bench_train_$mode.err | cut -c1-300; python - <<PY
import json
try:
    d=json.load(open("$OUT/bench_train_$mode.json"))
    print("train $mode", d["value"], d["ms_per_step"], {k:(v["ms"],v["launches"],v["gbs"]) for k,v in d["breakdown"].items() if "bnact" in k}, d["loss"])
except Exception as e: print("$mode: no result", e)
PY
done
