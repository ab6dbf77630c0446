#!/usr/bin/env bash
# Round 4, visit i: the safe vmcnt protocol as the product: stress + parity, its price against the unsafe counted waits, timeline,
# the headline with and without wreg (one stream / two streams)
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/${1:-r04i}; mkdir -p "$OUT"
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q --tb=short --timeout 400 -p no:cacheprovider -k "wreg or conv_all_variants" > "$OUT/pytest_wreg.log" 2>&1
echo "pytest wreg rc=$?"; tail -5 "$OUT/pytest_wreg.log" | cut -c1-400
L="128,128,3,1,80,80,32 256,256,3,1,40,40,32 512,512,3,1,20,20,32 128,128,3,1,40,40,32"
for lib in "" tools/_build/libyolov6_hip_wregprobe9.so; do
  echo "== lib ${lib:-product}"
  Y6_LIB_PATH=$lib timeout 200 python tools/conv_bench.py --data relu --layers $L --variants 39 40 --iters 20 2>&1 | grep -v amdgpu | cut -c1-200
done
Y6_TRACE_DATA=relu Y6_LIB_PATH=tools/_build/libyolov6_hip_wregprobe1.so timeout 100 python tools/dma_trace.py 256,256,3,1,40,40,32 wreg_p7 2>&1 | grep -v amdgpu | cut -c1-1200
NOWREG="7,8,9,12,13,14,15,16,17,18,19,20,21,24,28,29,30,34,36,38,39,40,41,42"
run() {  # name, env...
  local name=$1; shift
  env "$@" timeout 120 python bench.py --no-cpu-baseline --dropin-steps 0 --profile-out "$OUT/ops_$name.json" > "$OUT/bench_$name.json" 2> "$OUT/bench_$name.err"
  python - <<PY
import json
try:
    d=json.load(open("$OUT/bench_$name.json")); r=json.load(open("$OUT/ops_$name.json"))["rows"]
    print("$name", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["forward"]["ms"], {k: round(v["ms"], 3) for k, v in d["breakdown"].items()}, d.get("self_check"), d.get("schedule"))
    print("   3x3:", " ".join(f"{x['op']}:{x['variant']}:{x['ms']*1e3:.0f}" for x in r if x["kind"] == "conv" and x["ksize"] == 3 and x["stride"] == 1))
except Exception as e: print("$name: no result", e)
PY
}
run old1 Y6_AUTOTUNE_EXCLUDE=$NOWREG
run wreg1 Y6_DUMMY=1
run wreg1_1stream Y6_SCHED_STREAMS=1
run old1_1stream Y6_SCHED_STREAMS=1 Y6_AUTOTUNE_EXCLUDE=$NOWREG
run wreg2 Y6_DUMMY=1
echo done
