#!/usr/bin/env bash
# Round 4, visit k: the ping-pong form (one 8-wave block, two groups alternating matrix / request phases): parity + stress, layer table
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/${1:-r04k}; mkdir -p "$OUT"
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q --tb=short --timeout 400 -p no:cacheprovider -k "wreg or conv_all_variants" > "$OUT/pytest_wreg.log" 2>&1
echo "pytest wreg rc=$?"; tail -8 "$OUT/pytest_wreg.log" | cut -c1-600
L="128,128,3,1,80,80,32 256,256,3,1,40,40,32 512,512,3,1,20,20,32 128,128,3,1,40,40,32 256,256,3,1,20,20,32 256,128,3,1,40,40,32"
timeout 300 python tools/conv_bench.py --data relu --layers $L --variants 33 39 40 43 44 45 --iters 20 --out "$OUT/conv_bench_pp.json" 2>&1 | grep -v amdgpu | cut -c1-200
echo done
