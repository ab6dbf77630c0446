#!/usr/bin/env bash
# Round 3, visit r: (1) pixel-major 16-channel halo image with register-resident fragment addresses vs the planar default
# (kernel tests on the probe library, then the headline bench with each library, alternating); (2) the step as 1 / 2 / 4
# micro-batch plans on as many streams; (3) fresh PMC traffic of the shipped plan.
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/r03r; mkdir -p $OUT
PIX=$PWD/tools/_build/libyolov6_hip_dmapixmajor.so
Y6_LIB_PATH=$PIX timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q --tb=short --timeout 300 -p no:cacheprovider -k "conv_all_variants or dma or tap_geometry or epilogue" > $OUT/pytest_pixmajor.log 2>&1
echo "pytest pixmajor rc=$?"; tail -3 $OUT/pytest_pixmajor.log | cut -c1-300
ab() {  # name, lib
  local name=$1 lib=$2
  Y6_LIB_PATH=$lib Y6_AUTOTUNE_LOG=$OUT/autotune_$name.log timeout 600 python bench.py --steps 100 --no-cpu-baseline --dropin-steps 0 --profile-out $OUT/ops_$name.json > $OUT/bench_$name.json 2> $OUT/bench_$name.err
  echo "ab $name rc=$?"; tail -1 $OUT/bench_$name.err | cut -c1-200
  python - <<PY
import json
try:
    d=json.load(open("$OUT/bench_$name.json"))
    print("$name", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["ms_per_step"], d["roofline"]["kernel"][-80:], d["self_check"])
except Exception as e: print("no result", e)
PY
}
DEF=$PWD/yolov6_amd/lib/libyolov6_hip.so
ab planar1 $DEF
ab pixmajor1 $PIX
ab planar2 $DEF
ab pixmajor2 $PIX
Y6_AUTOTUNE_CACHE=$PWD/$OUT/split.cache timeout 900 python tools/split_batch_bench.py --out $OUT/split_batch.json > $OUT/split_batch.log 2>&1
echo "split rc=$?"; grep -E "^\{" $OUT/split_batch.log | cut -c1-400; tail -2 $OUT/split_batch.log | cut -c1-300
cp profiles/r03/autotune_r03q.cache $OUT/autotune.cache
timeout 900 bash tools/gpu_pmc_traffic.sh r03r $PWD/$OUT/autotune.cache 2>&1 | cut -c1-300 | tail -45
echo done
