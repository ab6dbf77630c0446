#!/usr/bin/env bash
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/${1:-r04x}; mkdir -p "$OUT"
timeout 400 python tools/inflight_probe.py 300 2>&1 | grep -v amdgpu | tee "$OUT/inflight.txt"
echo done
