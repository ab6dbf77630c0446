#!/usr/bin/env bash
# Round 4, visit o: clean same-box A/B of the hand-counted fragment reads: library at 6db1b23 vs HEAD with the SAME candidate set
# (p5 / p6 excluded), two runs each, interleaved; then HEAD with p5 / p6 allowed
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/${1:-r04o}; mkdir -p "$OUT"
EX="7,8,9,12,13,14,15,16,17,18,19,20,21,24,28,29,30,34,36,38,41,42"
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
run old1 Y6_LIB_PATH=tools/_build/libyolov6_hip_6db1b23.so Y6_AUTOTUNE_EXCLUDE=$EX
run new1 Y6_AUTOTUNE_EXCLUDE=$EX
run old2 Y6_LIB_PATH=tools/_build/libyolov6_hip_6db1b23.so Y6_AUTOTUNE_EXCLUDE=$EX
run new2 Y6_AUTOTUNE_EXCLUDE=$EX
run new_p56a Y6_DUMMY=1
run new_p56b Y6_DUMMY=1
L="128,128,3,1,80,80,32 256,256,3,1,40,40,32 512,512,3,1,20,20,32 128,128,3,1,40,40,32 256,256,3,1,20,20,32"
for rep in 1 2; do
echo "== old lib rep $rep"; Y6_LIB_PATH=tools/_build/libyolov6_hip_6db1b23.so timeout 200 python tools/conv_bench.py --data relu --layers $L --variants 39 40 --iters 200 --out "$OUT/conv_bench_old$rep.json" 2>&1 | grep -v amdgpu | cut -c1-200
echo "== new lib rep $rep"; timeout 200 python tools/conv_bench.py --data relu --layers $L --variants 39 40 --iters 200 --out "$OUT/conv_bench_new$rep.json" 2>&1 | grep -v amdgpu | cut -c1-200
done
echo done
