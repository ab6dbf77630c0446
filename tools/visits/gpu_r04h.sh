#!/usr/bin/env bash
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
for lib in "" tools/_build/libyolov6_hip_wregprobe8.so; do
 for spec in "128,128,3,1,80,80,32 wreg_p4" "128,128,3,1,80,80,32 wreg_p7" "512,512,3,1,20,20,32 wreg_p4" "256,256,3,1,40,40,32 wreg_p7"; do
  set -- $spec
  Y6_LIB_PATH=$lib timeout 200 python tools/wreg_stress.py $1 $2 400 2>&1 | grep -v amdgpu | cut -c1-400
  Y6_LIB_PATH=$lib timeout 200 python tools/wreg_stress.py $1 $2 300 --noise 2>&1 | grep -v amdgpu | cut -c1-400
 done
done
timeout 100 python tools/wreg_stress.py 128,128,3,1,80,80,32 dma_c2p2 300 --noise 2>&1 | grep -v amdgpu | cut -c1-400
