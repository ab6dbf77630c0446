#!/usr/bin/env bash
# Round 4, visit e: wreg v3 (no hoisted-and-spilled invariants, bias in registers per item): parity, table, timeline
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 Y6_ENABLE_CANDIDATES=wreg
OUT=gpurun_out/${1:-r04e}; mkdir -p "$OUT"
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q --tb=short --timeout 300 -p no:cacheprovider -k "wreg or conv_all_variants or tap_geometry or not_transposed" > "$OUT/pytest_wreg.log" 2>&1
echo "pytest wreg rc=$?"; tail -8 "$OUT/pytest_wreg.log" | cut -c1-400
L="64,64,3,1,160,160,32 128,128,3,1,80,80,32 256,256,3,1,40,40,32 512,512,3,1,20,20,32 128,128,3,1,40,40,32 256,256,3,1,20,20,32 64,64,3,1,80,80,32 256,128,3,1,40,40,32"
timeout 300 python tools/conv_bench.py --layers $L --variants 25 26 33 38 39 40 41 42 --iters 20 --out "$OUT/conv_bench_wreg.json" 2>&1 | grep -v amdgpu | cut -c1-200
for spec in "256,256,3,1,40,40,32 wreg_p7" "128,128,3,1,40,40,32 wreg_p4" "128,128,3,1,80,80,32 wreg_p7" "64,64,3,1,160,160,32 wreg2_p7"; do
  set -- $spec
  Y6_LIB_PATH=tools/_build/libyolov6_hip_wregprobe1.so timeout 100 python tools/dma_trace.py $1 $2 > "$OUT/trace_${2}_$(echo $1 | tr , _).txt" 2>&1
  grep -v amdgpu "$OUT/trace_${2}_$(echo $1 | tr , _).txt" | cut -c1-1200
done
echo done
