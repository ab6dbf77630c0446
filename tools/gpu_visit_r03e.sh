#!/usr/bin/env bash
# Round 3, visit e: fused producer -> stride-2 pairs (conv_fused.hip): op tests, model / parity tests, same-box A/B.
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/r03e; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q --tb=short --timeout 600 -p no:cacheprovider -k "fused or stem" -s > $OUT/pytest_fused.log 2>&1
echo "pytest fused rc=$?" | tee -a $OUT/pytest_fused.log; grep -v "^fused\|^\.$" $OUT/pytest_fused.log | tail -25; grep "^fused\|^\.fused" $OUT/pytest_fused.log | sort | uniq -c | sort -rn | head -12
timeout 1500 python -m pytest tests/test_gpu_model.py tests/test_gpu_parity_bench.py tests/test_gpu_dropin.py "tests/test_gpu_training.py::test_checkpoint_paths_after_training_steps" tests/test_gpu_train_parity.py \
  -m gpu -q --tb=short --timeout 900 -p no:cacheprovider -s > $OUT/pytest_model.log 2>&1
echo "pytest model rc=$?" | tee -a $OUT/pytest_model.log; grep -v "^{" $OUT/pytest_model.log | tail -30 | cut -c1-400
for mode in fused nofuse_s2 nofuse_all; do
  unset Y6_HEAD_NO_FUSE Y6_NO_FUSE_S2
  [ $mode = nofuse_s2 ] && export Y6_NO_FUSE_S2=1
  [ $mode = nofuse_all ] && export Y6_NO_FUSE_S2=1 Y6_HEAD_NO_FUSE=1
  timeout 600 python bench.py --no-cpu-baseline --dropin-steps 0 --profile-out $OUT/bench_ops_$mode.json > $OUT/bench_$mode.json 2> $OUT/bench_$mode.err
  echo "bench $mode rc=$?"; tail -2 $OUT/bench_$mode.err | cut -c1-300; python - <<PY
import json
try:
    d=json.load(open("$OUT/bench_$mode.json"))
    print("$mode", d["value"], d["ms_per_step"], {k:(v["ms"],v["launches"]) for k,v in d["breakdown"].items()})
except Exception as e: print("$mode: no result", e)
PY
done
