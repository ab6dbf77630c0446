#!/usr/bin/env bash
# Round 3, visit t (the round's last: one box, ordered by what must not be lost if the budget clamp cuts the visit short):
#   1 smoke + headline bench (200 steps) + the driver's form of it (--steps 20 --warmup 5), same tuning choices
#   2 the multi-GPU launch path on one GPU: bench.py under torch.distributed.run with a forced one-rank RCCL group
#     (Y6_FORCE_DIST=1: communicator, barriers, MAX reduce; training: the chunked gradient all-reduce on the side stream)
#   3 training bench A/B: weight-gradient work of the backward plan on a side stream (Y6_SIDE_STREAM=1)
#   4 configs[3] (L6 1280^2 b8) and configs[4] (S-QA int8 vs its fp16 plan) on this round's kernels
#   5 rocprofv3 kernel stats of the headline command
#   6 the whole -m gpu suite at HEAD; then the training tests with the side stream on
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
TAG=r03t
OUT=gpurun_out/$TAG; mkdir -p $OUT
T0=$(date +%s)
lap() { echo "-- $1 done at +$(( $(date +%s) - T0 )) s"; }
show() {  # file, python expression over d
  python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print(eval(sys.argv[2]))
except Exception as e:
    print("no result:", sys.argv[1], e)
PY
}
INF='(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["ms_per_step"], d["forward"]["ms"], d["nms"]["ms"], (d.get("dropin_api") or {}).get("value"), d["self_check"])'
TRN='(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline_hbm"]["frac"], d["loss"], d["memory_gb"], {k: (round(v["ms"], 2), v["launches"]) for k, v in d["breakdown"].items() if v["ms"] > 0.5})'

# ---- 1
tools/gpu_round.sh $TAG smoke bench 2>&1 | cut -c1-400 | tail -12
show $OUT/bench.json "$INF"; lap "smoke + headline"
Y6_AUTOTUNE_CACHE="$PWD/$OUT/autotune.cache" timeout 240 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_driver_form.json 2> $OUT/bench_driver_form.err
echo "driver-form rc=$?"; show $OUT/bench_driver_form.json "$INF"; lap "driver form"

# ---- 2
Y6_FORCE_DIST=1 Y6_AUTOTUNE_CACHE="$PWD/$OUT/autotune.cache" timeout 240 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 \
  --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --dropin-steps 0 \
  > $OUT/bench_torchrun_rccl1.json 2> $OUT/bench_torchrun_rccl1.err
echo "torchrun infer rc=$?"; tail -2 $OUT/bench_torchrun_rccl1.err | cut -c1-200; show $OUT/bench_torchrun_rccl1.json "$INF"
Y6_FORCE_DIST=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29542 \
  bench.py --mode train --gpus 1 --steps 10 --warmup 3 > $OUT/bench_train_torchrun_rccl1.json 2> $OUT/bench_train_torchrun_rccl1.err
echo "torchrun train rc=$?"; tail -2 $OUT/bench_train_torchrun_rccl1.err | cut -c1-200; show $OUT/bench_train_torchrun_rccl1.json "$TRN"; lap "torchrun"

# ---- 3
train() {  # name, env...
  local name=$1; shift
  env "$@" timeout 300 python bench.py --mode train --profile-out $OUT/train_ops_$name.json > $OUT/bench_train_$name.json 2> $OUT/bench_train_$name.err
  echo "train $name rc=$?"; tail -1 $OUT/bench_train_$name.err | cut -c1-200; show $OUT/bench_train_$name.json "$TRN"
}
train one1 Y6_SIDE_STREAM=0
train side1 Y6_SIDE_STREAM=1
lap "train A/B"

# ---- 4
timeout 300 python bench.py --model yolov6l6 --size 1280 --batch 8 --steps 50 --warmup 5 --no-cpu-baseline --dropin-steps 10 > $OUT/bench_l6.json 2> $OUT/bench_l6.err
echo "l6 rc=$?"; tail -1 $OUT/bench_l6.err | cut -c1-200; show $OUT/bench_l6.json "$INF"
timeout 240 python bench.py --model yolov6s_qa --no-cpu-baseline --dropin-steps 0 > $OUT/bench_qa_fp16.json 2> $OUT/bench_qa_fp16.err
echo "qa fp16 rc=$?"; show $OUT/bench_qa_fp16.json "$INF"
timeout 240 python bench.py --model yolov6s_qa --int8 --no-cpu-baseline --dropin-steps 0 > $OUT/bench_qa_int8.json 2> $OUT/bench_qa_int8.err
echo "qa int8 rc=$?"; tail -1 $OUT/bench_qa_int8.err | cut -c1-200; show $OUT/bench_qa_int8.json "$INF"; lap "l6 + int8"

# ---- 5
tools/gpu_round.sh $TAG prof 2>&1 | cut -c1-300 | tail -14; lap "rocprofv3"

# ---- 6
timeout 900 python -m pytest tests -m gpu -q --tb=short --timeout 600 -p no:cacheprovider -x > $OUT/pytest_gpu.log 2>&1
echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log; tail -6 $OUT/pytest_gpu.log | cut -c1-300; lap "pytest"
Y6_SIDE_STREAM=1 timeout 500 python -m pytest tests/test_gpu_training.py tests/test_gpu_train_parity.py tests/test_gpu_loss.py -m gpu -q --tb=short --timeout 300 -p no:cacheprovider > $OUT/pytest_training_side.log 2>&1
echo "pytest training (side stream) rc=$?" | tee -a $OUT/pytest_training_side.log; tail -4 $OUT/pytest_training_side.log | cut -c1-300; lap "pytest side"
echo done
