#!/usr/bin/env bash
# 16-channel-chunk DMA kernels: planar halo image (default build) vs pixel-major swizzled (python tools/build_probe_libs.py --dma-pixmajor)
set -u
OUT=gpurun_out/${1:-ablayout}; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_int8.py -q --tb=short --timeout 300 -p no:cacheprovider -x > "$OUT/pytest.log" 2>&1
echo "pytest rc=$?"; tail -4 "$OUT/pytest.log"
for m in planar swz planar swz; do
  if [ $m = swz ]; then export Y6_LIB_PATH=tools/_build/libyolov6_hip_dmapixmajor.so; else unset Y6_LIB_PATH; fi
  timeout 600 python bench.py --steps 200 --no-cpu-baseline --dropin-steps 0 > "$OUT/bench_$m.json" 2> "$OUT/bench_$m.err"
  python - <<PY
import json
d=json.load(open("$OUT/bench_$m.json"))
print("$m", d["value"], d["ms_per_step"], d["roofline"]["achieved"], d["roofline"]["frac"], {k:(round(v["ms"],3),v["launches"]) for k,v in d["breakdown"].items() if k in ("conv3x3s1","nms")}, d["roofline"]["kernel"][-60:])
PY
done
