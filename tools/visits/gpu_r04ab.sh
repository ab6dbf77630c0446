#!/usr/bin/env bash
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/${1:-r04ab}; mkdir -p "$OUT"
timeout 3000 python -m pytest tests -m gpu -q --tb=short --timeout 900 -p no:cacheprovider > "$OUT/pytest_gpu.log" 2>&1
echo "pytest gpu rc=$?"; tail -8 "$OUT/pytest_gpu.log" | cut -c1-300
echo done
