#!/usr/bin/env bash
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/${1:-r04ac}; mkdir -p "$OUT"
for i in 1 2 3; do
timeout 600 python -m pytest tests/test_gpu_training.py -m gpu -q --tb=line --timeout 600 -p no:cacheprovider -k "training_graph_forward_backward_vs_oracle" 2>&1 | tail -4 | cut -c1-400
done
echo done
