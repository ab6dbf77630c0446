#!/usr/bin/env bash
# Round 4, visit ak (last GPU minutes): the 64-cout int8 forms of the register-fed kernel (int8 variants 13 / 14, also Cin = 32),
# opt-in by Y6_I8_WREG2=1 - the whole int8 test file with the switch on (op tests against the oracle + the model tests with the
# heuristic picking them), the S-QA int8 line with and without, per-op tables, rocprofv3 kernel stats of the line with them.
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/${1:-r04ak}; mkdir -p "$OUT"
T0=$(date +%s)
lap() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
Y6_I8_WREG2=1 timeout -k 5 80 python -m pytest tests/test_gpu_int8.py -q -m gpu > "$OUT/pytest_int8_wreg2_on.log" 2>&1
lap "int8 tests, 64-cout forms on: rc=$? $(tail -1 "$OUT/pytest_int8_wreg2_on.log")"
grep -E "^(FAILED|ERROR)|Error|assert" "$OUT/pytest_int8_wreg2_on.log" | head -12
bench() {  # name, env
  local name=$1; shift
  env "$@" timeout -k 5 60 python bench.py --model yolov6s_qa --int8 --no-cpu-baseline --no-train-sub --dropin-steps 0 --windows 2 --profile-out "$OUT/ops_$name.json" > "$OUT/bench_$name.json" 2> "$OUT/bench_$name.err"
  python - <<PY
import json
try:
    d=json.load(open("$OUT/bench_$name.json"))
    print("$name", d["value"], d["ms_per_step"], "seq", d["sequential"]["value"], "frac", d["roofline"]["frac"], "self_check", d.get("self_check"))
    r=json.load(open("$OUT/ops_$name.json"))["rows"]
    print("   ops 1-3, 40-46 (us):", [round(x["ms"]*1e3,1) for x in r[1:4]], [round(x["ms"]*1e3,1) for x in r[40:47]])
except Exception as e: print("$name: no result", e); print(open("$OUT/bench_$name.err").read()[-800:])
PY
}
bench qa_int8_wreg2 Y6_I8_WREG2=1
bench qa_int8_default Y6_I8_WREG2=0
bench qa_int8_wreg2_b Y6_I8_WREG2=1
lap benches
R=$PWD
( cd /tmp && Y6_I8_WREG2=1 timeout -k 5 60 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$OUT/rocprof_int8" -o b -- python "$R/bench.py" --model yolov6s_qa --int8 --no-cpu-baseline --no-train-sub --dropin-steps 0 --windows 1 --steps 50 --no-verify > "$R/$OUT/rocprof_int8.json" 2> "$R/$OUT/rocprof_int8.err" )
lap "rocprof rc=$?"
find "$OUT/rocprof_int8" -name "*kernel_trace.csv" -delete 2>/dev/null
echo done
