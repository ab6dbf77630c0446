#!/usr/bin/env bash
# Round 4, first GPU visit: settle the gated candidates of round 3 (parity, then same-box A/B), time the fixed cost of a
# small-map 3x3 launch (batch sweep: intercept vs slope) and trace one (s_memtime timeline of block 0).
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/${1:-r04a}; mkdir -p "$OUT"
T0=$(date +%s); lap() { echo "-- $1 done at +$(( $(date +%s) - T0 )) s"; }
# 1. parity of every candidate at once (a failure names the test)
Y6_ENABLE_CANDIDATES=all timeout 420 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py tests/test_gpu_dropin.py tests/test_gpu_parity_bench.py -m gpu -q --tb=short --timeout 200 -p no:cacheprovider -x > "$OUT/pytest_candidates_all.log" 2>&1
echo "pytest candidates(all) rc=$?"; tail -5 "$OUT/pytest_candidates_all.log" | cut -c1-300; lap "candidate parity"
# 2. layer tables
Y6_ENABLE_CANDIDATES=stg3 timeout 150 python tools/conv_bench.py --layers 128,128,3,1,80,80,32 256,256,3,1,40,40,32 512,512,3,1,20,20,32 128,128,3,1,40,40,32 256,256,3,1,20,20,32 64,64,3,1,80,80,32 --variants 25 26 27 33 40 --iters 20 --out "$OUT/conv_bench_stg3.json" > "$OUT/conv_bench_stg3.log" 2>&1
grep -v amdgpu "$OUT/conv_bench_stg3.log" | tail -32 | cut -c1-200; lap "stg3 table"
L="512,256,1,1,20,20,32 256,256,1,1,20,20,32 1024,256,1,1,20,20,32 512,512,1,1,20,20,32 512,128,1,1,20,20,32 384,128,1,1,40,40,32 256,64,1,1,40,40,32 192,64,1,1,80,80,32 128,128,1,1,40,40,32"
Y6_ENABLE_CANDIDATES=kres timeout 200 python tools/conv_bench.py --layers $L --variants 2 3 22 23 38 39 --iters 20 --out "$OUT/conv_bench_1x1.json" > "$OUT/conv_bench_1x1.log" 2>&1
grep -v amdgpu "$OUT/conv_bench_1x1.log" | tail -56 | cut -c1-200; lap "1x1 table"
# 3. fixed cost of a small-map launch: the same layer at batch 4..128 (rounds 0.11 .. 3.5 of the 8-wave form)
SW=""; for b in 4 8 16 32 64 128; do SW="$SW 128,128,3,1,40,40,$b 256,256,3,1,20,20,$b 64,64,3,1,80,80,$b"; done
timeout 200 python tools/conv_bench.py --layers $SW --variants 26 33 --iters 30 --out "$OUT/conv_bench_batch_sweep.json" > "$OUT/conv_bench_batch_sweep.log" 2>&1
grep -v amdgpu "$OUT/conv_bench_batch_sweep.log" | tail -40 | cut -c1-200; lap "batch sweep"
# 4. timelines
for spec in "128,128,3,1,40,40,32 dma8_c4p1" "256,256,3,1,20,20,32 dma_c2p1" "64,64,3,1,80,80,32 dma_c2p1" "256,256,3,1,40,40,32 dma8_c4p1"; do
  set -- $spec
  Y6_LIB_PATH=tools/_build/libyolov6_hip_dmaprobe1.so timeout 100 python tools/dma_trace.py $1 $2 > "$OUT/trace_${2}_$(echo $1 | tr , _).txt" 2>&1
  grep -v amdgpu "$OUT/trace_${2}_$(echo $1 | tr , _).txt" | cut -c1-1500
done; lap "traces"
# 5. headline A/B
run() {  # name, env...
  local name=$1; shift
  env "$@" timeout 100 python bench.py --no-cpu-baseline --dropin-steps 0 --profile-out "$OUT/ops_$name.json" > "$OUT/bench_$name.json" 2> "$OUT/bench_$name.err"
  python - <<PY
import json
try:
    d=json.load(open("$OUT/bench_$name.json")); r=json.load(open("$OUT/ops_$name.json"))["rows"]
    print("$name", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["forward"]["ms"], {k: round(v["ms"], 3) for k, v in d["breakdown"].items()}, d.get("schedule"))
except Exception as e: print("$name: no result", e)
PY
}
run base1
run kres1 Y6_ENABLE_CANDIDATES=kres
run sppf1 Y6_ENABLE_CANDIDATES=sppf
run levels1 Y6_ENABLE_CANDIDATES=levels
run stg3 Y6_ENABLE_CANDIDATES=stg3
run split1 Y6_ENABLE_CANDIDATES=split
run split6400 Y6_ENABLE_CANDIDATES=split Y6_SPLIT_MAX_HW=6400
run all1 Y6_ENABLE_CANDIDATES=all
run base2
run all2 Y6_ENABLE_CANDIDATES=all
lap "headline A/B"
Y6_ENABLE_CANDIDATES=i8sched timeout 200 python -m pytest tests/test_gpu_int8.py -m gpu -q --tb=short --timeout 150 -p no:cacheprovider > "$OUT/pytest_int8_sched.log" 2>&1
echo "pytest int8 (scheduled) rc=$?"; tail -3 "$OUT/pytest_int8_sched.log" | cut -c1-300
for n in i8_one i8_sched; do
  case $n in *sched*) E="Y6_ENABLE_CANDIDATES=i8sched";; *) E="Y6_DUMMY=1";; esac
  env $E timeout 120 python bench.py --model yolov6s_qa --int8 --no-cpu-baseline --dropin-steps 0 > "$OUT/bench_$n.json" 2> "$OUT/bench_$n.err"
  python -c "import json; d=json.load(open('$OUT/bench_$n.json')); print('$n', d['value'], d['ms_per_step'], d['self_check'], d.get('schedule'))" 2>/dev/null || echo "$n: no result"
done
lap "int8 schedule"
echo done
