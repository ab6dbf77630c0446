#!/usr/bin/env bash
# Round 4, visit q: NMS candidates from the decode launch (y6_nms_sink): parity, headline A/B
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/${1:-r04q}; mkdir -p "$OUT"
timeout 900 python -m pytest tests/test_gpu_parity_bench.py tests/test_gpu_nms_tal.py -m gpu -q --tb=short --timeout 600 -p no:cacheprovider -k "candidates or nms" > "$OUT/pytest_nms.log" 2>&1
echo "pytest nms rc=$?"; tail -12 "$OUT/pytest_nms.log" | cut -c1-500
run() {  # name, args...
  local name=$1; shift
  timeout 120 python bench.py --no-cpu-baseline --dropin-steps 0 "$@" > "$OUT/bench_$name.json" 2> "$OUT/bench_$name.err"
  python - <<PY
import json
try:
    d=json.load(open("$OUT/bench_$name.json"))
    print("$name", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["forward"]["ms"], {k: round(v["ms"], 3) for k, v in d["breakdown"].items()}, d["nms"], d.get("self_check"))
except Exception as e: print("$name: no result", e)
PY
}
run nofuse1 --no-fuse-candidates
run fuse1
run nofuse2 --no-fuse-candidates
run fuse2
echo done
