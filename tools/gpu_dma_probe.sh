#!/usr/bin/env bash
# LDS-DMA conv kernel: s_memtime trace + ceiling probes on three layer shapes (one GPU visit).
set -u
OUT=gpurun_out/${1:-dmaprobe}
mkdir -p "$OUT"
export TMPDIR=/tmp
L=${LAYERS:-"64,64,3,1,160,160,32 128,128,3,1,80,80,32 512,512,3,1,20,20,32 256,256,3,1,20,20,32"}
for spec in $L; do
  for v in ${TRACE_V:-dma_c2p2 dma_c2p1 dma8_c2p2}; do
    Y6_LIB_PATH=tools/_build/libyolov6_hip_dmaprobe1.so timeout 120 python tools/dma_trace.py $spec $v 2>&1 | grep -v amdgpu.ids
  done
done > "$OUT/trace.log" 2>&1
cat "$OUT/trace.log" | cut -c1-900
echo "== probes (ms per layer; base first)"
V="${PROBE_V:-24 25 26}"
( echo base; timeout 300 python tools/conv_bench.py --layers $L --variants $V --iters 20 ) > "$OUT/probe_base.log" 2>&1
for n in ${PROBES:-2 7 8 3 4 5 6}; do
  ( echo probe $n; Y6_LIB_PATH=tools/_build/libyolov6_hip_dmaprobe$n.so timeout 300 python tools/conv_bench.py --layers $L --variants $V --iters 20 ) > "$OUT/probe_$n.log" 2>&1
done
grep -h "probe\|base\|ms" "$OUT"/probe_*.log | grep -v amdgpu
