#!/usr/bin/env bash
# Round 4, visit s: y6_nms stage by stage, library at 6db1b23 against HEAD, same box
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/${1:-r04s}; mkdir -p "$OUT"
for lib in tools/_build/libyolov6_hip_6db1b23.so ""; do
  echo "== lib ${lib:-HEAD}" | tee -a "$OUT/nms_stages.txt"
  Y6_LIB_PATH=$lib timeout 300 python tools/nms_bench.py 0.02 2>&1 | grep -v amdgpu | tee -a "$OUT/nms_stages.txt"
done
echo done
