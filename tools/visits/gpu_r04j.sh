#!/usr/bin/env bash
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
for spec in "256,256,3,1,40,40,32 wreg_p7" "128,128,3,1,80,80,32 wreg_p7" "128,128,3,1,40,40,32 wreg_p4" "512,512,3,1,20,20,32 wreg_p4"; do
  set -- $spec
  Y6_TRACE_DATA=relu Y6_LIB_PATH=tools/_build/libyolov6_hip_wregprobe1.so timeout 100 python tools/dma_trace.py $1 $2 2>&1 | grep -E "blocks|histogram|lived|one launch|span"
done
