#!/usr/bin/env bash
# One visit for the 1x1 kernel of conv_pw.hip: parity of every variant, then the fourteen 1x1 shapes of YOLOv6-S b32 per kernel form.
#   usage: tools/gpu_pw.sh <tag>
set -u
TAG=${1:-pw}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_ops.py -q --tb=short --timeout 600 -p no:cacheprovider -k "conv_all_variants" > "$OUT/pytest_conv.log" 2>&1
echo "pytest rc=$?"; tail -15 "$OUT/pytest_conv.log"
timeout 900 python tools/conv_bench.py --iters 20 --out "$OUT/conv_bench_1x1.json" \
  --variants mfma_c2p1 mfma_c4p1 stream1x1_c1 stream1x1_c2 pw_c4p2 pw_c2p2 \
  --layers 512,256,1,1,20,20,32 256,256,1,1,20,20,32 1024,256,1,1,20,20,32 512,512,1,1,20,20,32 512,128,1,1,20,20,32 \
           256,128,1,1,40,40,32 384,128,1,1,40,40,32 128,64,1,1,40,40,32 128,128,1,1,40,40,32 \
           128,64,1,1,80,80,32 192,64,1,1,80,80,32 64,64,1,1,80,80,32 > "$OUT/conv_bench_1x1.log" 2>&1
echo "bench rc=$?"; cat "$OUT/conv_bench_1x1.log" | tail -80
