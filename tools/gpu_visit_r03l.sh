#!/usr/bin/env bash
# Round 3, visit l: single-launch BatchNorm statistics / bnact backward reduction (no memset, no finalize launches), wgrad operand
# planes shared between convs that read the same view; training tests + train bench A/B.
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/r03l; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_training.py tests/test_gpu_train_ops.py tests/test_gpu_train_parity.py -m gpu -q --tb=short --timeout 900 -p no:cacheprovider > $OUT/pytest_train.log 2>&1
echo "pytest train rc=$?" | tee -a $OUT/pytest_train.log; tail -12 $OUT/pytest_train.log | cut -c1-300
run() {  # name, env...
  local name=$1; shift
  env "$@" timeout 600 python bench.py --mode train --profile-out $OUT/train_ops_$name.json > $OUT/bench_train_$name.json 2> $OUT/bench_train_$name.err
  echo "train $name rc=$?"; tail -1 $OUT/bench_train_$name.err | cut -c1-200
  python - <<PY
import json
try:
    d=json.load(open("$OUT/bench_train_$name.json"))
    b=d["breakdown"]
    print("$name", d["value"], d["ms_per_step"], {k:(round(v["ms"],2),v["launches"]) for k,v in b.items() if v["ms"]>0.5}, d["loss"])
except Exception as e: print("no result", e)
PY
}
run new Y6_DUMMY=1
run old Y6_BN_THREE_LAUNCH=1 Y6_NO_SHARED_PLANES=1
run items4k Y6_WGRAD_ITEMS=4096
run items16k Y6_WGRAD_ITEMS=16384
echo done
