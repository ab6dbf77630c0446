#!/usr/bin/env bash
# PMC counter passes over the conv micro-benchmark (separate rocprofv3 runs per counter group, as
# MI355X_MICROARCH.md prescribes; never combined with sys/hip/hsa tracing).
#   usage: tools/gpu_pmc.sh <tag> "<layers...>" "<variants...>"
set -u
TAG=${1:-pmc01}
LAYERS=${2:-"256,256,3,1,40,40,32 64,64,3,1,160,160,32"}
VARIANTS=${3:-"1 2 5 6 11"}
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
rocprofv3 -L > "$OUT/counters_list.txt" 2>&1 || true
CMD="python $OLDPWD/tools/conv_bench.py --layers $LAYERS --variants $VARIANTS --iters 3"
run() { # name counters...
  local name=$1; shift
  timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc "$@" -d "$OUT/$name" -o p -- $CMD > "$OUT/$name.log" 2>&1
  echo "$name rc=$?"
}
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU
run sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM
run fetch FETCH_SIZE
run write WRITE_SIZE
run tcc TCC_HIT_sum TCC_MISS_sum
python $OLDPWD/tools/conv_bench.py --layers $LAYERS --iters 10 --out "$OUT/conv_bench.json" > "$OUT/conv_bench.log" 2>&1
find "$OUT" -name "*kernel_trace.csv" -size +5M -delete
du -sh "$OUT"; ls "$OUT"; for f in $(find "$OUT" -name "*counter_collection.csv" | head -8); do echo "$f: $(wc -l < $f) rows"; done
