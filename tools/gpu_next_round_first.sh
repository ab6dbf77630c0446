#!/usr/bin/env bash
# First GPU visit of the round after round 2 (one box, ~12 min):
#   1. the probes / tests of everything written after round 2's last visit (model families, dma8_c4p1) - reported, not fatal
#   2. fill pricing: LDS-DMA requests + barriers only, all / halo only / weights only (probe builds 2 / 7 / 8), four layers
#   3. same-box A/B of the candidate set: default / + dma8_c4p1 (33) / + resident-weight forms (34, 35) / + dma_c2p4 (36) / + stride-2 dma8s2_c4p1 (37) / + all five
# Before the visit, in the build container:  python tools/build_probe_libs.py --dma 2 7 8   (tools/_build/ travels with the snapshot)
set -u
OUT=gpurun_out/${1:-next1}; mkdir -p "$OUT"; export TMPDIR=/tmp
for n in 2 7 8; do
  [ -f tools/_build/libyolov6_hip_dmaprobe$n.so ] || { echo "probe lib $n missing: building on the box"; python tools/build_probe_libs.py --dma $n > "$OUT/build_probe_$n.log" 2>&1; }
done
timeout 900 python -m pytest tests/test_gpu_families.py -q -m gpu -rxX -p no:cacheprovider > "$OUT/pytest_families.log" 2>&1
tail -8 "$OUT/pytest_families.log"
L="64,64,3,1,160,160,32 64,64,3,1,80,80,32 64,128,3,1,80,80,32 128,128,3,1,80,80,32 256,256,3,1,40,40,32 512,512,3,1,20,20,32 128,128,3,1,40,40,32"
( echo base; timeout 300 python tools/conv_bench.py --layers $L --variants 25 26 28 33 34 35 36 --iters 20 ) > "$OUT/probe_base.log" 2>&1
( echo "base s2"; timeout 300 python tools/conv_bench.py --layers 64,128,3,2,160,160,32 128,256,3,2,80,80,32 256,512,3,2,40,40,32 128,128,3,2,40,40,32 --variants 2 3 31 32 37 --iters 20 ) > "$OUT/probe_base_s2.log" 2>&1
for n in 2 7 8; do
  ( echo probe $n; Y6_LIB_PATH=tools/_build/libyolov6_hip_dmaprobe$n.so timeout 300 python tools/conv_bench.py --layers $L --variants 25 26 --iters 20 ) > "$OUT/probe_$n.log" 2>&1
done
grep -h "probe\|base\|ms" "$OUT"/probe_*.log | grep -v amdgpu
BASE="7,8,9,12,13,14,15,16,17,18,19,20,21,24,28,29,30"
for cfg in default:$BASE,33,34,35,36,37 all:$BASE; do
  name=${cfg%%:*}; ex=${cfg#*:}
  Y6_AUTOTUNE_EXCLUDE="$ex" Y6_AUTOTUNE_LOG="$OUT/autotune_$name.log" timeout 600 python bench.py --steps 200 --no-cpu-baseline --dropin-steps 0 --profile-out "$OUT/ops_$name.json" > "$OUT/bench_$name.json" 2> "$OUT/bench_$name.err"
  python - <<PY
import json
d=json.load(open("$OUT/bench_$name.json")); r=json.load(open("$OUT/ops_$name.json"))["rows"]
print("$name", d["value"], d["ms_per_step"], "3x3s1", round(d["breakdown"]["conv3x3s1"]["ms"],3))
print("   ", " ".join(f"{x['op']}:{x['variant']}:{x['ms']*1e3:.0f}" for x in r if x["ksize"] == 3))
PY
done
