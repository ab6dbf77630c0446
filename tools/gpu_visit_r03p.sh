#!/usr/bin/env bash
# Round 3, visit p: NHWC-fed weight gradient v4 (+ k halves combined in LDS, incremental row / slot counters): kernel tests, kernel bench with
# timing probes against the plane-fed kernel, train bench.
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/r03p; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_training.py -m gpu -q --tb=short --timeout 300 -p no:cacheprovider -k "wgrad" -s > $OUT/pytest_wgrad.log 2>&1
rc=$?; echo "pytest wgrad rc=$rc"; grep -E "wgrad_nhwc|per-tap|passed|failed" $OUT/pytest_wgrad.log | cut -c1-300 | head -40
if [ $rc -ne 0 ]; then tail -30 $OUT/pytest_wgrad.log | cut -c1-300; echo "stopping: kernel tests failed"; exit 0; fi
timeout 600 python tools/wgrad_bench.py $OUT/wgrad_bench.json > $OUT/wgrad_bench.log 2>&1; echo "wgrad_bench rc=$?"; cut -c1-420 $OUT/wgrad_bench.log | tail -12
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
run nhwc_w64 Y6_DUMMY=1
run nhwc_w32 Y6_WGRAD_NHWC_MINW3=32 Y6_WGRAD_NHWC_MINW1=32
echo done
