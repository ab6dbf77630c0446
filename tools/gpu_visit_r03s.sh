#!/usr/bin/env bash
# Round 3, visit s: BatchNorm statistics / bnact backward sums without atomics (block partials + ordered second-level sums):
# the training-side tests (incl. the many-block reproducibility case and the config-3 per-op parity), then the training bench
# A/B against the atomic form (Y6_BN_ATOMICS=1), alternating.
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/r03s; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_training.py tests/test_gpu_train_ops.py -m gpu -q --tb=short --timeout 600 -p no:cacheprovider -x > $OUT/pytest_training.log 2>&1
rc=$?; echo "pytest training rc=$rc"; tail -4 $OUT/pytest_training.log | cut -c1-300
if [ $rc -ne 0 ]; then grep -E "Error|assert|FAILED" $OUT/pytest_training.log | head -20 | cut -c1-300; fi
run() {  # name, env...
  local name=$1; shift
  env "$@" timeout 600 python bench.py --mode train --profile-out $OUT/train_ops_$name.json > $OUT/bench_train_$name.json 2> $OUT/bench_train_$name.err
  echo "train $name rc=$?"; tail -1 $OUT/bench_train_$name.err | cut -c1-200
  python - <<PY
import json
try:
    d=json.load(open("$OUT/bench_train_$name.json"))
    b=d["breakdown"]
    print("$name", d["value"], d["ms_per_step"], {k:(round(v["ms"],2),v["launches"]) for k,v in b.items() if v["ms"]>0.5}, d["loss"], d["memory_gb"])
except Exception as e: print("no result", e)
PY
}
run part1 Y6_DUMMY=1
run atomics1 Y6_BN_ATOMICS=1
run part2 Y6_DUMMY=1
run atomics2 Y6_BN_ATOMICS=1
timeout 900 python -m pytest tests/test_gpu_train_parity.py -m gpu -q --tb=short --timeout 800 -p no:cacheprovider > $OUT/pytest_train_parity.log 2>&1
echo "pytest train parity rc=$?"; tail -3 $OUT/pytest_train_parity.log | cut -c1-300
echo done
