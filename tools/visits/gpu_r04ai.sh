#!/usr/bin/env bash
# Round 4, visit ai (the round's last): the int8 form of the register-fed 3x3 kernels (conv_wreg.hip I8; int8 variants 10 / 11 / 12,
# opt-in by Y6_I8_WREG=1) - parity (op tests, the S-QA model tests, the full-width teacher-forced test) with the switch ON, the
# S-QA int8 bench line with and without it - then the regression check of everything the template change touches with the
# switch at its default, smoke() and the default bench line.
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/${1:-r04ai}; mkdir -p "$OUT"
T0=$(date +%s)
lap() { echo "[$(( $(date +%s) - T0 )) s] $*"; }

Y6_I8_WREG=1 timeout -k 5 230 python -m pytest tests/test_gpu_int8.py -q -m gpu --durations=6 > "$OUT/pytest_int8_wreg_on.log" 2>&1
lap "int8 tests, switch on: rc=$? $(tail -1 "$OUT/pytest_int8_wreg_on.log")"

bench() {  # name, env
  local name=$1; shift
  env "$@" timeout -k 5 100 python bench.py --model yolov6s_qa --int8 --no-cpu-baseline --no-train-sub --dropin-steps 0 --windows 2 > "$OUT/bench_$name.json" 2> "$OUT/bench_$name.err"
  python - <<PY
import json
try:
    d=json.load(open("$OUT/bench_$name.json"))
    b={k:round(v["ms"],4) for k,v in d.get("breakdown",{}).items() if isinstance(v,dict) and "ms" in v}
    print("$name", d["value"], d["ms_per_step"], "seq", d["sequential"]["value"], "frac", d["roofline"]["frac"], b)
except Exception as e: print("$name: no result", e); print(open("$OUT/bench_$name.err").read()[-800:])
PY
}
bench qa_int8_wreg Y6_I8_WREG=1
lap bench1
bench qa_int8_base Y6_I8_WREG=0
lap bench2

timeout -k 5 260 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py tests/test_gpu_parity_bench.py tests/test_gpu_int8.py -q -m gpu --durations=8 > "$OUT/pytest_subset_default.log" 2>&1
lap "conv / model / parity / int8 tests, defaults: rc=$? $(tail -1 "$OUT/pytest_subset_default.log")"

timeout -k 5 60 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1
lap "smoke rc=$? $(tail -1 "$OUT/smoke.log")"
timeout -k 5 200 python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
lap "bench rc=$? $(head -c 300 "$OUT/bench_default.json")"
echo done
