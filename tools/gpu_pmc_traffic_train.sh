#!/usr/bin/env bash
# HBM traffic of the training step's kernels: separate rocprofv3 --pmc passes for FETCH_SIZE and WRITE_SIZE over
# `bench.py --mode train` (no autotune trials: heuristic variants, so only the kernels of the step run).
#   usage: tools/gpu_pmc_traffic_train.sh <tag> [batch]
set -u
TAG=${1:-pmc_train}
BATCH=${2:-64}
OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
R=$PWD
for c in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout -k 5 420 rocprofv3 --kernel-trace --output-format csv --pmc $c -d "$OUT/$c" -o t -- python "$R/bench.py" --no-supervisor --mode train --batch $BATCH --steps 2 --warmup 1 --no-autotune > "$OUT/$c.json" 2> "$OUT/$c.err" )
  echo "$c rc=$?"
  find "$OUT/$c" -name "*kernel_trace.csv" -delete
done
python tools/pmc_traffic_train.py "$OUT" > "$OUT/pmc_traffic_train.json"; head -30 "$OUT/pmc_traffic_train.json"
