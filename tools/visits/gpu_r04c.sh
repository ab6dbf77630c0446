#!/usr/bin/env bash
# Round 4, visit c: where the register-fed 3x3 kernel spends its time: s_memtime timeline + ceiling probes (tools/build_probe_libs.py --wreg)
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 Y6_ENABLE_CANDIDATES=wreg
OUT=gpurun_out/${1:-r04c}; mkdir -p "$OUT"
for spec in "256,256,3,1,40,40,32 wreg_p7" "128,128,3,1,40,40,32 wreg_p4" "128,128,3,1,80,80,32 wreg_p7"; do
  set -- $spec
  Y6_LIB_PATH=tools/_build/libyolov6_hip_wregprobe1.so timeout 100 python tools/dma_trace.py $1 $2 > "$OUT/trace_${2}_$(echo $1 | tr , _).txt" 2>&1
  grep -v amdgpu "$OUT/trace_${2}_$(echo $1 | tr , _).txt" | cut -c1-1800
done
L="256,256,3,1,40,40,32 128,128,3,1,80,80,32 128,128,3,1,40,40,32 512,512,3,1,20,20,32"
for n in 0 2 3 4 5 6 7; do
  if [ $n = 0 ]; then LIBP=""; else LIBP="tools/_build/libyolov6_hip_wregprobe$n.so"; fi
  echo "== probe $n"
  Y6_LIB_PATH=$LIBP timeout 200 python tools/conv_bench.py --layers $L --variants 39 40 --iters 20 --out "$OUT/conv_bench_probe$n.json" 2>&1 | grep -v amdgpu | cut -c1-160
done
echo done
