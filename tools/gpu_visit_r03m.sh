#!/usr/bin/env bash
# Round 3, visit m: weight gradients of the stride-1 convs read from NHWC (LDS-DMA rows + ds_read_b64_tr_b16): instruction probe,
# kernel tests, training tests, train bench A/B against the plane-fed kernel.
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/r03m; mkdir -p $OUT
timeout 120 python tools/tr16_probe.py $OUT/tr16_probe.json > $OUT/tr16_probe.log 2>&1; echo "probe rc=$?"; head -c 1500 $OUT/tr16_probe.log; echo
timeout 600 python -m pytest tests/test_gpu_training.py -m gpu -q --tb=short --timeout 300 -p no:cacheprovider -k "wgrad" -s > $OUT/pytest_wgrad.log 2>&1
rc=$?; echo "pytest wgrad rc=$rc"; grep -E "wgrad_nhwc|per-tap|passed|failed" $OUT/pytest_wgrad.log | cut -c1-300 | head -40
if [ $rc -ne 0 ]; then tail -30 $OUT/pytest_wgrad.log | cut -c1-300; echo "stopping: kernel tests failed"; exit 0; fi
timeout 1200 python -m pytest tests/test_gpu_training.py tests/test_gpu_train_parity.py -m gpu -q --tb=short --timeout 900 -p no:cacheprovider -k "not wgrad" > $OUT/pytest_train.log 2>&1
echo "pytest train rc=$?" | tee -a $OUT/pytest_train.log; tail -8 $OUT/pytest_train.log | cut -c1-300
run() {  # name, env...
  local name=$1; shift
  env "$@" timeout 600 python bench.py --mode train --profile-out $OUT/train_ops_$name.json > $OUT/bench_train_$name.json 2> $OUT/bench_train_$name.err
  echo "train $name rc=$?"; tail -1 $OUT/bench_train_$name.err | cut -c1-200
  python - <<PY
import json
try:
    d=json.load(open("$OUT/bench_train_$name.json"))
    b=d["breakdown"]
    print("$name", d["value"], d["ms_per_step"], {k:(round(v["ms"],2),v["launches"],round(v["tflops"],1)) for k,v in b.items() if v["ms"]>0.5}, d["loss"])
except Exception as e: print("no result", e)
PY
}
run nhwc Y6_DUMMY=1
run planes Y6_WGRAD_PLANES=1
echo done
