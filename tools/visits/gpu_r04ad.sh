#!/usr/bin/env bash
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/${1:-r04ad}; mkdir -p "$OUT"
for lib in tools/_build/libyolov6_hip_0ac64ea.so ""; do
echo "== lib ${lib:-HEAD}"
Y6_LIB_PATH=$lib timeout 600 python -m pytest tests/test_gpu_training.py -m gpu -q --tb=line --timeout 600 -p no:cacheprovider -k "training_graph_forward_backward_vs_oracle and tiny-64" 2>&1 | tail -3 | cut -c1-300
python - <<PY
import json
d=json.load(open("gpurun_out/train_grad_report_tiny_64_b2.json"))
print(d["summary"]["grad_worst"], d["summary"]["grad_median"])
e=d["errs"]; print({k:e[k] for k in e if "stem" in k})
PY
done
echo done
