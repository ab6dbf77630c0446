#!/usr/bin/env bash
# Round 4, visit l: wave-priority modes of conv_wreg (Y6_WREG_PRIO 0..3): all-block timelines + effective clock, layer table, headline
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/${1:-r04l}; mkdir -p "$OUT"
for p in 0 1 2 3; do
 for spec in "256,256,3,1,40,40,32 wreg_p7" "128,128,3,1,80,80,32 wreg_p7" "128,128,3,1,40,40,32 wreg_p4" "512,512,3,1,20,20,32 wreg_p4"; do
  set -- $spec
  echo "== prio $p $1 $2" >> "$OUT/block_times.txt"
  Y6_WREG_PRIO=$p Y6_TRACE_DATA=relu Y6_LIB_PATH=tools/_build/libyolov6_hip_wregprobe1.so timeout 100 python tools/dma_trace.py $1 $2 2>&1 | grep -v amdgpu | cut -c1-900 >> "$OUT/block_times.txt"
 done
done
grep -E "==|blocks|lived|one launch" "$OUT/block_times.txt" | cut -c1-330
L="128,128,3,1,80,80,32 256,256,3,1,40,40,32 512,512,3,1,20,20,32 128,128,3,1,40,40,32 256,256,3,1,20,20,32 256,128,3,1,40,40,32 64,64,3,1,160,160,32"
for p in 0 1 2 3; do
  echo "== prio $p"
  Y6_WREG_PRIO=$p timeout 200 python tools/conv_bench.py --data relu --layers $L --variants 39 40 --iters 20 --out "$OUT/conv_bench_prio$p.json" 2>&1 | grep -v amdgpu | cut -c1-200
done
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
run prio0 Y6_WREG_PRIO=0
run prio1 Y6_WREG_PRIO=1
run prio2 Y6_WREG_PRIO=2
run prio3 Y6_WREG_PRIO=3
run prio0b Y6_WREG_PRIO=0
echo done
