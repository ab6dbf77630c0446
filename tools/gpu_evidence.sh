#!/usr/bin/env bash
# Row (d) evidence for the committed library: the driver's line with a per-op table, the rocprofv3 kernel summary of the SAME plan
# (autotune choices from a cache file, so that only the chosen kernels appear), PMC traffic, the one-rank torchrun / RCCL launch.
#   usage: tools/gpu_evidence.sh <tag> [what...]   what in {line,prof,pmc,dist,guardtrain} (default: all)
set -u
TAG=${1:-r05g}; shift || true
WHAT=${*:-line prof pmc dist guardtrain}
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
R=$PWD; OUT=$R/gpurun_out/$TAG; mkdir -p "$OUT"
CACHE=$OUT/autotune.cache
T0=$(date +%s); lap() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
has() { [[ " $WHAT " == *" $1 "* ]]; }
if has line; then
  rm -f "$CACHE"
  Y6_AUTOTUNE_CACHE="$CACHE" timeout -k 5 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 --profile-out "$OUT/ops_driver.json" > "$OUT/bench_driver.json" 2> "$OUT/bench_driver.err"
  echo "driver line rc=$?"; cut -c1-300 "$OUT/bench_driver.json"
  Y6_AUTOTUNE_CACHE="$CACHE" timeout -k 5 600 python3 bench.py --profile-out "$OUT/ops_default.json" --no-train-sub --no-config-subs > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
  echo "default line (200 steps, cached variants) rc=$?"; cut -c1-300 "$OUT/bench_default.json"
  lap line
fi
if has prof; then
  ( cd /tmp && Y6_AUTOTUNE_CACHE="$CACHE" timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/rocprof" -o b -- python3 "$R/bench.py" --no-supervisor --inflight 1 --steps 50 --warmup 5 --windows 1 --no-cpu-baseline --no-train-sub --no-config-subs --dropin-steps 0 --no-verify > "$OUT/rocprof_bench.json" 2> "$OUT/rocprof.err" )
  echo "rocprof rc=$?"; find "$OUT/rocprof" -name "*kernel_trace.csv" -delete
  python3 tools/rocprof_classes.py "$OUT/rocprof" 61 > "$OUT/rocprof_classes.json" 2>> "$OUT/rocprof.err"; head -30 "$OUT/rocprof_classes.json"
  lap prof
fi
if has pmc; then
  for c in FETCH_SIZE WRITE_SIZE; do
    ( cd /tmp && Y6_AUTOTUNE_CACHE="$CACHE" timeout -k 5 300 rocprofv3 --kernel-trace --output-format csv --pmc $c -d "$OUT/$c" -o b -- python3 "$R/bench.py" --no-supervisor --inflight 1 --steps 2 --warmup 1 --no-cpu-baseline --no-train-sub --no-config-subs --windows 1 --dropin-steps 0 --no-verify > "$OUT/$c.json" 2> "$OUT/$c.err" )
    echo "$c rc=$?"; find "$OUT/$c" -name "*kernel_trace.csv" -delete
  done
  python3 tools/pmc_traffic.py "$OUT" > "$OUT/pmc_traffic.json"; head -50 "$OUT/pmc_traffic.json"
  lap pmc
fi
if has dist; then
  # the driver's N > 1 launch form with one rank, the process group forced up: RCCL communicator, barriers, MAX reduce, in-flight default
  Y6_FORCE_DIST=1 timeout -k 5 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-train-sub --no-config-subs > "$OUT/dist_infer.json" 2> "$OUT/dist_infer.err"
  echo "torchrun + RCCL, infer rc=$?"; cut -c1-200 "$OUT/dist_infer.json"
  Y6_FORCE_DIST=1 timeout -k 5 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29612 bench.py --mode train --gpus 1 --steps 10 --warmup 3 > "$OUT/dist_train.json" 2> "$OUT/dist_train.err"
  echo "torchrun + RCCL, train rc=$?"; cut -c1-200 "$OUT/dist_train.json"
  lap dist
fi
if has guardtrain; then
  for mode in end start; do
    timeout -k 5 600 python3 tests/tight_probe.py --mode $mode bench.py --mode train --steps 3 --warmup 2 > "$OUT/guard_train_$mode.json" 2> "$OUT/guard_train_$mode.err"
    echo "train step under the guard allocator ($mode) rc=$?"; grep -E "Memory access|Error" "$OUT/guard_train_$mode.err" | tail -2
  done
  lap guardtrain
fi
du -sh "$OUT"
