#!/usr/bin/env bash
# Round 3, visit h: fused pairs with the block-level staged epilogue + register-resident stem-pair weights; tests + same-box A/B + rocprof.
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/r03h; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_preproc.py -m gpu -q --tb=short --timeout 600 -p no:cacheprovider -k "fused or stem or preproc or process_image or letterbox" > $OUT/pytest_fused.log 2>&1
echo "pytest fused rc=$?" | tee -a $OUT/pytest_fused.log; tail -6 $OUT/pytest_fused.log | cut -c1-300
timeout 1500 python -m pytest tests/test_gpu_model.py tests/test_gpu_parity_bench.py tests/test_gpu_dropin.py "tests/test_gpu_training.py::test_checkpoint_paths_after_training_steps" \
  -m gpu -q --tb=short --timeout 900 -p no:cacheprovider -s > $OUT/pytest_model.log 2>&1
echo "pytest model rc=$?" | tee -a $OUT/pytest_model.log; grep -v "^{" $OUT/pytest_model.log | tail -12 | cut -c1-400
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
