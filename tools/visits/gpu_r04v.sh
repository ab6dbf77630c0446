#!/usr/bin/env bash
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/${1:-r04v}; mkdir -p "$OUT"
timeout 900 python -m pytest tests/test_gpu_int8.py -m gpu -q --tb=short --timeout 600 -p no:cacheprovider -k "histogram or calibration_layers" -s > "$OUT/pytest_int8.log" 2>&1
echo "pytest rc=$?"; grep -E "int8 vs fp16|passed|failed|Error|assert" "$OUT/pytest_int8.log" | tail -12 | cut -c1-400
for name in fp16 int8; do
  extra=""; [ $name = int8 ] && extra="--int8"
  timeout 200 python bench.py --model yolov6s_qa $extra --no-cpu-baseline --no-train-sub --dropin-steps 0 > "$OUT/bench_qa_$name.json" 2> "$OUT/bench_qa_$name.err"
  python - <<PY
import json
try:
    d=json.load(open("$OUT/bench_qa_$name.json")); print("$name", d["value"], d["ms_per_step"], d["roofline"]["frac"], {k: round(v["ms"], 3) for k, v in d["breakdown"].items()})
except Exception as e: print("$name: no result", e)
PY
done
timeout 300 python bench.py --model yolov6l6 --size 1280 --batch 8 --no-cpu-baseline --no-train-sub --dropin-steps 0 > "$OUT/bench_l6.json" 2> "$OUT/bench_l6.err"
python - <<PY
import json
try:
    d=json.load(open("$OUT/bench_l6.json")); print("l6", d["value"], d["ms_per_step"], d["roofline"]["frac"], {k: round(v["ms"], 3) for k, v in d["breakdown"].items()}, d["roofline"]["kernel"][-200:])
except Exception as e: print("l6: no result", e)
PY
echo done
