#!/usr/bin/env bash
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/${1:-r04z}; mkdir -p "$OUT"
timeout 600 python -m pytest tests/test_gpu_parity_bench.py -m gpu -q --tb=short --timeout 600 -p no:cacheprovider -k "inflight or bench_step or candidates" 2>&1 | tail -8
timeout 300 python bench.py --no-cpu-baseline --no-train-sub --dropin-steps 0 > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
python - <<PY
import json
try:
    d=json.load(open("$OUT/bench_default.json"))
    print("default", d["value"], d["ms_per_step"], "inflight", d["inflight"], "seq", d["sequential"]["value"], d["windows"], d["roofline"]["frac"], d.get("self_check"))
except Exception as e: print("no result", e); print(open("$OUT/bench_default.err").read()[-1500:])
PY
echo done
