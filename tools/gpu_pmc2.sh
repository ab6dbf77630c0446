#!/usr/bin/env bash
# LDS-conflict counters with 16-wide vs 32-wide tile rows
set -u
OUT=$PWD/gpurun_out/${1:-pmc02}; mkdir -p "$OUT"; export TMPDIR=/tmp; cd /tmp
L="128,128,3,1,80,80,32 64,64,3,1,160,160,32"
for tw in 0 1; do
  CMD="python $OLDPWD/tools/conv_bench.py --layers $L --variants 2 5 9 --iters 3"
  Y6_CONV_TW32=$tw timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d "$OUT/tw$tw" -o p -- $CMD > "$OUT/tw$tw.log" 2>&1
  Y6_CONV_TW32=$tw python $OLDPWD/tools/conv_bench.py --layers $L --variants 1 2 4 5 7 8 9 --iters 10 2>/dev/null | sed "s/^/tw32=$tw /"
done
find "$OUT" -name "*kernel_trace.csv" -delete
