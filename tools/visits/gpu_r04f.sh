#!/usr/bin/env bash
# Round 4, visit f: is the 3x3 conv clock / power bound?  (shader cycles against the 100 MHz counter; random / half-zero / zero data),
# then the headline with and without the register-fed kernels (same box, alternating) and the model-level GPU tests with them in the plan
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/${1:-r04f}; mkdir -p "$OUT"
for data in rand relu zeros; do
  for spec in "256,256,3,1,40,40,32 wreg_p7" "256,256,3,1,40,40,32 dma8_c4p1"; do
    set -- $spec
    if [[ $2 == wreg* ]]; then LIBP=tools/_build/libyolov6_hip_wregprobe1.so; else LIBP=""; fi
    [ -n "$LIBP" ] && Y6_TRACE_DATA=$data Y6_LIB_PATH=$LIBP timeout 100 python tools/dma_trace.py $1 $2 2>&1 | grep -E "lived|span" | sed "s/^/$data $2: /"
  done
  timeout 200 python tools/conv_bench.py --data $data --layers 256,256,3,1,40,40,32 128,128,3,1,80,80,32 --variants 25 33 39 --iters 20 --out "$OUT/conv_bench_$data.json" 2>&1 | grep -v amdgpu | cut -c1-200
done
NOWREG="7,8,9,12,13,14,15,16,17,18,19,20,21,24,28,29,30,34,36,38,39,40,41,42"
run() {  # name, env...
  local name=$1; shift
  env "$@" timeout 120 python bench.py --no-cpu-baseline --dropin-steps 0 --profile-out "$OUT/ops_$name.json" > "$OUT/bench_$name.json" 2> "$OUT/bench_$name.err"
  python - <<PY
import json
try:
    d=json.load(open("$OUT/bench_$name.json")); r=json.load(open("$OUT/ops_$name.json"))["rows"]
    print("$name", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["forward"]["ms"], {k: round(v["ms"], 3) for k, v in d["breakdown"].items()}, d.get("self_check"))
    print("   3x3:", " ".join(f"{x['op']}:{x['variant']}:{x['ms']*1e3:.0f}" for x in r if x["kind"] == "conv" and x["ksize"] == 3 and x["stride"] == 1))
except Exception as e: print("$name: no result", e)
PY
}
run old1 Y6_AUTOTUNE_EXCLUDE=$NOWREG
run wreg1 Y6_DUMMY=1
run old2 Y6_AUTOTUNE_EXCLUDE=$NOWREG
run wreg2 Y6_DUMMY=1
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py tests/test_gpu_parity_bench.py tests/test_gpu_dropin.py -m gpu -q --tb=short --timeout 300 -p no:cacheprovider -x > "$OUT/pytest_model.log" 2>&1
echo "pytest ops+model+parity+dropin rc=$?"; tail -6 "$OUT/pytest_model.log" | cut -c1-400
echo done
