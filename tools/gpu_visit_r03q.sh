#!/usr/bin/env bash
# Round 3, visit q: state of HEAD - smoke, the whole -m gpu suite, headline bench + rocprofv3 kernel stats of the same command,
# A/B of the two-stream step (NMS of batch k beside the forward of batch k+1), training bench.
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/r03q; mkdir -p $OUT
tools/gpu_round.sh r03q smoke tests bench prof 2>&1 | cut -c1-600 | tail -120
ab() {  # name, args...
  local name=$1; shift
  Y6_AUTOTUNE_CACHE="$PWD/$OUT/autotune.cache" timeout 600 python bench.py --no-cpu-baseline --dropin-steps 0 "$@" > $OUT/bench_ab_$name.json 2> $OUT/bench_ab_$name.err
  echo "ab $name rc=$?"; tail -1 $OUT/bench_ab_$name.err | cut -c1-200
  python - <<PY
import json
try:
    d=json.load(open("$OUT/bench_ab_$name.json"))
    print("$name", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["ms_per_step"], d["nms"], d["self_check"], {k:round(v["ms"],3) for k,v in d["breakdown"].items()})
except Exception as e: print("no result", e)
PY
}
ab same1 --nms-stream same
ab side1 --nms-stream side
ab same2 --nms-stream same
ab side2 --nms-stream side
timeout 900 python bench.py --mode train --profile-out $OUT/train_ops.json > $OUT/bench_train.json 2> $OUT/bench_train.err
echo "train rc=$?"; tail -1 $OUT/bench_train.err | cut -c1-200; cut -c1-1500 $OUT/bench_train.json
echo done
