#!/usr/bin/env bash
# Round 4, visit aa: after the prune (25 conv variants of 45, the chunk-granular persistent kernel and the unused pipe / LDS-DMA forms
# gone, name-based candidate set): conv / int8 / model / train GPU tests, default bench line
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/${1:-r04aa}; mkdir -p "$OUT"
timeout 2400 python -m pytest tests/test_gpu_ops.py tests/test_gpu_int8.py tests/test_gpu_model.py tests/test_gpu_parity_bench.py tests/test_gpu_families.py -m gpu -q --tb=short --timeout 900 -p no:cacheprovider -x > "$OUT/pytest.log" 2>&1
echo "pytest rc=$?"; tail -6 "$OUT/pytest.log" | cut -c1-300
timeout 600 python bench.py --no-cpu-baseline --dropin-steps 0 > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
python - <<PY
import json
try:
    d=json.load(open("$OUT/bench_default.json"))
    print("default", d["value"], d["ms_per_step"], "seq", d["sequential"]["value"], d["windows"]["spread_pct"], d["roofline"]["frac"], d["roofline"]["kernel"][-120:], {k: round(v["ms"], 3) for k, v in d["breakdown"].items()}, d.get("self_check"), (d.get("train") or {}).get("ms_per_step"))
except Exception as e: print("no result", e); print(open("$OUT/bench_default.err").read()[-1500:])
PY
timeout 200 python bench.py --model yolov6s_qa --int8 --no-cpu-baseline --no-train-sub --dropin-steps 0 > "$OUT/bench_int8.json" 2> "$OUT/bench_int8.err"
python - <<PY
import json
try:
    d=json.load(open("$OUT/bench_int8.json")); print("int8", d["value"], d["ms_per_step"], "seq", d["sequential"]["value"])
except Exception as e: print("int8: no result", e); print(open("$OUT/bench_int8.err").read()[-800:])
PY
echo done
