#!/usr/bin/env bash
# Round 4, visit p: conv_wreg at stride 2 (wregs2_p3 / wregs2_p4): parity, layer table against the LDS-DMA / per-tap stride-2 kernels, headline
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/${1:-r04p}; mkdir -p "$OUT"
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q --tb=short --timeout 400 -p no:cacheprovider -k "wreg" > "$OUT/pytest_wreg.log" 2>&1
echo "pytest wreg rc=$?"; tail -15 "$OUT/pytest_wreg.log" | cut -c1-400
L="64,128,3,2,160,160,32 128,256,3,2,80,80,32 256,512,3,2,40,40,32 128,128,3,2,40,40,32 256,256,3,2,20,20,32"
timeout 300 python tools/conv_bench.py --data relu --layers $L --iters 50 --out "$OUT/conv_bench_s2.json" 2>&1 | grep -v amdgpu | cut -c1-200
run() {  # name, env...
  local name=$1; shift
  env "$@" timeout 120 python bench.py --no-cpu-baseline --dropin-steps 0 --profile-out "$OUT/ops_$name.json" > "$OUT/bench_$name.json" 2> "$OUT/bench_$name.err"
  python - <<PY
import json
try:
    d=json.load(open("$OUT/bench_$name.json")); r=json.load(open("$OUT/ops_$name.json"))["rows"]
    print("$name", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["forward"]["ms"], {k: round(v["ms"], 3) for k, v in d["breakdown"].items()}, d.get("self_check"))
    print("   s2:", " ".join(f"{x['op']}:{x['variant']}:{x['ms']*1e3:.0f}" for x in r if x["kind"] == "conv" and x["ksize"] == 3 and x["stride"] == 2))
except Exception as e: print("$name: no result", e)
PY
}
EX="7,8,9,12,13,14,15,16,17,18,19,20,21,24,28,29,30,34,36,42"
run nos2a Y6_AUTOTUNE_EXCLUDE=$EX,43,44
run s2a Y6_AUTOTUNE_EXCLUDE=$EX
run nos2b Y6_AUTOTUNE_EXCLUDE=$EX,43,44
run s2b Y6_AUTOTUNE_EXCLUDE=$EX
echo done
