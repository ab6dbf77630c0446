#!/usr/bin/env bash
# HBM traffic of the bench's kernels: separate rocprofv3 --pmc passes for FETCH_SIZE and WRITE_SIZE
# (MI355X_MICROARCH.md: they do not fit one pass), tuning choices taken from a cache file so that only
# the chosen kernels run.   usage: tools/gpu_pmc_traffic.sh <tag> [autotune.cache]
set -u
TAG=${1:-pmc}
OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"
CACHE=${2:-$OUT/autotune.cache}
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
R=$PWD
if [ ! -s "$CACHE" ]; then
  Y6_AUTOTUNE_CACHE="$CACHE" timeout -k 5 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-train-sub --windows 1 --dropin-steps 0 --no-verify --no-supervisor > "$OUT/tune_bench.json" 2> "$OUT/tune.err"
fi
for c in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && Y6_AUTOTUNE_CACHE="$CACHE" timeout -k 5 300 rocprofv3 --kernel-trace --output-format csv --pmc $c -d "$OUT/$c" -o b -- python "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-train-sub --windows 1 --dropin-steps 0 --no-verify --no-supervisor > "$OUT/$c.json" 2> "$OUT/$c.err" )
  echo "$c rc=$?"
  find "$OUT/$c" -name "*kernel_trace.csv" -delete
done
python tools/pmc_traffic.py "$OUT" > "$OUT/pmc_traffic.json"; cat "$OUT/pmc_traffic.json" | head -40
