#!/usr/bin/env bash
# MFMA-busy / wave-cycle counters of the 3x3 stride-1 LDS-DMA kernels (one SQ pass; never combined with tracing domains).
#   usage: tools/gpu_pmc_mfma.sh <tag> ["<layers>"] ["<variants>"]
set -u
TAG=${1:-pmc_mfma}
LAYERS=${2:-"64,64,3,1,160,160,32 128,128,3,1,80,80,32 256,256,3,1,40,40,32 512,512,3,1,20,20,32 128,128,3,1,40,40,32 64,64,3,1,80,80,32"}
VARIANTS=${3:-"25 26 28"}
R=$PWD; OUT=$R/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_WAVES -d "$OUT/sq" -o p -- python $R/tools/conv_bench.py --layers $LAYERS --variants $VARIANTS --iters 3 > "$OUT/sq.log" 2>&1 )
echo "pmc rc=$?"
find "$OUT" -name "*kernel_trace.csv" -size +5M -delete
python $R/tools/pmc_mfma.py "$OUT" > "$OUT/pmc_mfma.json" 2> "$OUT/pmc_mfma.err"; head -60 "$OUT/pmc_mfma.json"
