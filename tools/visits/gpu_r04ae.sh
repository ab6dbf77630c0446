#!/usr/bin/env bash
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 300 python tools/stem_wgrad_debug.py 2>&1 | grep -v amdgpu | tail -20
timeout 600 python -m pytest tests/test_gpu_training.py -m gpu -q --tb=line --timeout 600 -p no:cacheprovider -k "training_graph_forward_backward_vs_oracle" 2>&1 | tail -4 | cut -c1-500
echo done
