#!/usr/bin/env bash
# Round 4, visit aj: the int8 register-fed kernels are the default now - the int8 tests with nothing set, the per-op tables of the
# S-QA int8 plan with and without them (what round 5 starts from), the test files visit ai did not run (training, losses, NMS /
# assigners, drop-in, families, preprocessing, training parity) on the relinked library, rocprofv3 kernel stats of the int8 line.
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/${1:-r04aj}; mkdir -p "$OUT"
T0=$(date +%s)
lap() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
bench() {  # name, env
  local name=$1; shift
  env "$@" timeout -k 5 100 python bench.py --model yolov6s_qa --int8 --no-cpu-baseline --no-train-sub --dropin-steps 0 --windows 2 --profile-out "$OUT/ops_$name.json" > "$OUT/bench_$name.json" 2> "$OUT/bench_$name.err"
  python - <<PY
import json
try:
    d=json.load(open("$OUT/bench_$name.json"))
    print("$name", d["value"], d["ms_per_step"], "seq", d["sequential"]["value"], "frac", d["roofline"]["frac"])
except Exception as e: print("$name: no result", e); print(open("$OUT/bench_$name.err").read()[-800:])
PY
}
bench qa_int8_default Y6_DUMMY=1
bench qa_int8_nowreg Y6_I8_WREG=0
lap benches
timeout -k 5 90 python -m pytest tests/test_gpu_int8.py -q -m gpu > "$OUT/pytest_int8_default.log" 2>&1
lap "int8 tests, defaults: rc=$? $(tail -1 "$OUT/pytest_int8_default.log")"
timeout -k 5 40 python -m pytest tests/test_gpu_ops.py -q -m gpu -k wreg > "$OUT/pytest_wreg_ops.log" 2>&1
lap "wreg op tests: rc=$? $(tail -1 "$OUT/pytest_wreg_ops.log")"
timeout -k 5 250 python -m pytest tests/test_gpu_training.py tests/test_gpu_loss.py tests/test_gpu_nms_tal.py tests/test_gpu_dropin.py tests/test_gpu_families.py tests/test_gpu_preproc.py tests/test_gpu_train_parity.py -q -m gpu --durations=8 > "$OUT/pytest_rest.log" 2>&1
lap "remaining test files: rc=$? $(tail -1 "$OUT/pytest_rest.log")"
R=$PWD
( cd /tmp && timeout -k 5 90 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$OUT/rocprof_int8" -o b -- python "$R/bench.py" --model yolov6s_qa --int8 --no-cpu-baseline --no-train-sub --dropin-steps 0 --windows 1 --steps 50 --no-verify > "$R/$OUT/rocprof_int8.json" 2> "$R/$OUT/rocprof_int8.err" )
lap "rocprof rc=$?"
find "$OUT/rocprof_int8" -name "*kernel_trace.csv" -delete 2>/dev/null
echo done
