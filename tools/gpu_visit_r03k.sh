#!/usr/bin/env bash
# Round 3, visit k: self-distillation losses + the distillation head's training branch, MBLA / QARepVGG teacher-forced parity.
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/r03k; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_loss.py tests/test_gpu_training.py tests/test_gpu_train_parity.py tests/test_gpu_train_ops.py -m gpu -q --tb=short --timeout 900 -p no:cacheprovider \
  -k "distill or loss or teacher_forced or block_training_graph or head_pack or fuseab or training_graph" > $OUT/pytest.log 2>&1
echo "pytest rc=$?" | tee -a $OUT/pytest.log; tail -40 $OUT/pytest.log | cut -c1-500
