#!/usr/bin/env bash
# Round 3, visit g: residual operand read as 8-byte pieces (conv epilogue), letterbox on the device; op / training tests + train bench.
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/r03g; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_ops.py tests/test_gpu_preproc.py tests/test_gpu_training.py tests/test_gpu_train_ops.py tests/test_gpu_int8.py -m gpu -q --tb=short --timeout 900 -p no:cacheprovider > $OUT/pytest_ops.log 2>&1
echo "pytest ops rc=$?" | tee -a $OUT/pytest_ops.log; tail -12 $OUT/pytest_ops.log | cut -c1-300
timeout 900 python bench.py --mode train --profile-out $OUT/train_ops.json > $OUT/bench_train.json 2> $OUT/bench_train.err
echo "train rc=$?"; tail -2 $OUT/bench_train.err | cut -c1-300; python - <<PY
import json
try:
    d=json.load(open("$OUT/bench_train.json"))
    print("train", d["value"], d["ms_per_step"], {k:(v["ms"],v["launches"],v["tflops"],v["gbs"]) for k,v in d["breakdown"].items()}, d["loss"])
except Exception as e: print("no result", e)
PY
timeout 600 python bench.py --model yolov6m --no-cpu-baseline --dropin-steps 0 --steps 100 > $OUT/bench_m.json 2> $OUT/bench_m.err
echo "yolov6m rc=$?"; tail -2 $OUT/bench_m.err | cut -c1-300; cut -c1-300 $OUT/bench_m.json
