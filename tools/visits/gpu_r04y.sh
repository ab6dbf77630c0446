#!/usr/bin/env bash
# Round 4, visit y: bench.py with N steps in flight (default 2) against one at a time, same box; the bench self-check and parity test
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/${1:-r04y}; mkdir -p "$OUT"
run() {  # name, args...
  local name=$1; shift
  timeout 300 python bench.py --no-cpu-baseline --no-train-sub --dropin-steps 0 "$@" > "$OUT/bench_$name.json" 2> "$OUT/bench_$name.err"
  python - <<PY
import json
try:
    d=json.load(open("$OUT/bench_$name.json"))
    print("$name", d["value"], d["ms_per_step"], "inflight", d["inflight"], "seq", d["sequential"], d["windows"], d["roofline"]["frac"], {k: round(v["ms"], 3) for k, v in d["breakdown"].items()}, d.get("self_check"))
except Exception as e: print("$name: no result", e); print(open("$OUT/bench_$name.err").read()[-1500:])
PY
}
run fly2a
run fly1a --inflight 1
run fly2b
run fly3 --inflight 3
run fly1b --inflight 1
run driver20 --steps 20 --warmup 5
timeout 600 python -m pytest tests/test_gpu_parity_bench.py -m gpu -q --tb=short --timeout 600 -p no:cacheprovider -k "bench_step or candidates" 2>&1 | tail -3
echo done
