#!/usr/bin/env bash
# Round 3, visit c: re-check the tests that failed / were added after visit b, then the training-step evidence
# (bench line with both roofline classes, rocprofv3 kernel stats, PMC traffic) and the inference headline.
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/r03c; mkdir -p $OUT
timeout 1200 python -m pytest tests/test_gpu_training.py tests/test_gpu_train_parity.py tests/test_gpu_dropin.py "tests/test_gpu_model.py::test_rebind_and_repeat" \
  -m gpu -q --tb=short --timeout 600 -p no:cacheprovider -k "reference_shaped or checkpoint_paths or teacher_forced or reference_checkpoint or rebind" -s > $OUT/pytest_targeted.log 2>&1
echo "pytest rc=$?" | tee -a $OUT/pytest_targeted.log; tail -25 $OUT/pytest_targeted.log
tools/gpu_round.sh r03c bench train trainprof
tools/gpu_pmc_traffic_train.sh r03c_pmc_train
