#!/usr/bin/env bash
# Round 4, visit n: same-box A/B of the library at 6db1b23 (compiler-scheduled fragment reads, register-direct stores) against HEAD
# (hand-counted fragment reads; Y6_WREG_EPI=0 register-direct / 1 LDS-transposed stores), p5 / p6 tiles; parity + stress first
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/${1:-r04n}; mkdir -p "$OUT"
for epi in 1 0; do
  Y6_WREG_EPI=$epi timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q --tb=short --timeout 400 -p no:cacheprovider -k "wreg or conv_all_variants or epilogue" > "$OUT/pytest_wreg_epi$epi.log" 2>&1
  echo "pytest wreg epi$epi rc=$?"; tail -4 "$OUT/pytest_wreg_epi$epi.log" | cut -c1-600
done
L="128,128,3,1,80,80,32 256,256,3,1,40,40,32 512,512,3,1,20,20,32 128,128,3,1,40,40,32 256,256,3,1,20,20,32 64,128,3,1,80,80,32 128,256,3,1,40,40,32 256,512,3,1,20,20,32"
echo "== old lib"; Y6_LIB_PATH=tools/_build/libyolov6_hip_6db1b23.so timeout 200 python tools/conv_bench.py --data relu --layers $L --variants 39 40 --iters 20 --out "$OUT/conv_bench_old.json" 2>&1 | grep -v amdgpu | cut -c1-200
for epi in 0 1; do
  echo "== new lib epi $epi"
  Y6_WREG_EPI=$epi timeout 200 python tools/conv_bench.py --data relu --layers $L --variants 38 39 40 41 --iters 20 --out "$OUT/conv_bench_epi$epi.json" 2>&1 | grep -v amdgpu | cut -c1-200
done
for epi in 0 1; do
 for spec in "256,256,3,1,40,40,32 wreg_p7" "128,128,3,1,40,40,32 wreg_p4"; do
  set -- $spec
  echo "== epi $epi $1 $2" >> "$OUT/block_times.txt"
  Y6_WREG_EPI=$epi Y6_TRACE_DATA=relu Y6_LIB_PATH=tools/_build/libyolov6_hip_wregprobe1.so timeout 100 python tools/dma_trace.py $1 $2 2>&1 | grep -v amdgpu | cut -c1-900 >> "$OUT/block_times.txt"
 done
done
grep -E "==|blocks|lived|one launch|histogram|epilogue|units|stage|halo|barrier|prologue" "$OUT/block_times.txt" | cut -c1-330
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
run old1 Y6_LIB_PATH=tools/_build/libyolov6_hip_6db1b23.so
run epi1 Y6_WREG_EPI=1
run epi0 Y6_WREG_EPI=0
run old2 Y6_LIB_PATH=tools/_build/libyolov6_hip_6db1b23.so
run epi1b Y6_WREG_EPI=1
for sp in "128,128,3,1,80,80,32 wreg_p7" "256,256,3,1,40,40,32 wreg_p4" "128,128,3,1,40,40,32 wreg_p5" "256,256,3,1,20,20,32 wreg_p6"; do timeout 120 python tools/wreg_stress.py $sp 200 --noise 2>&1 | grep -v amdgpu | tail -1 | cut -c1-300 | tee -a "$OUT/wreg_stress.log"; done
echo done
