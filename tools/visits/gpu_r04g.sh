#!/usr/bin/env bash
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
for spec in "128,128,3,1,80,80,32 wreg_p4" "128,128,3,1,80,80,32 wreg_p7" "512,512,3,1,20,20,32 wreg_p4" "512,512,3,1,20,20,32 wreg_p7" "256,256,3,1,40,40,32 wreg_p7"; do
  set -- $spec
  timeout 120 python tools/wreg_debug.py $1 $2 2>&1 | grep -v amdgpu | cut -c1-700
done
