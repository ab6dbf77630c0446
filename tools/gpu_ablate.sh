#!/usr/bin/env bash
# Ablation of the persistent conv kernel: which part of a chunk costs the time?
set -u
OUT=gpurun_out/${1:-abl01}; mkdir -p $OUT
L="256,256,3,1,40,40,32 64,64,3,1,160,160,32"
for m in 0 1 2 3 4 8 12 16 7 15 31; do
  echo "== ablate $m"
  Y6_CONV_ABLATE=$m python tools/conv_bench.py --layers $L --variants 7 8 9 10 --iters 10 2>/dev/null | sed "s/^/abl=$m /"
done > $OUT/ablate.log 2>&1
cat $OUT/ablate.log
