#!/usr/bin/env bash
# LDS-DMA conv kernels: correctness tests, then the headline bench with the autotune log (per-layer ms of every candidate).
#   usage: tools/gpu_dma_ab.sh [tag]
set -u
TAG=${1:-dma}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_int8.py -q --tb=short --timeout 300 -p no:cacheprovider -x \
  -k "dma or conv_all_variants or tap_geometry or layout or epilogue_variants or i8 or int8" > "$OUT/pytest.log" 2>&1
echo "pytest rc=$?"; tail -30 "$OUT/pytest.log"
rm -f "$OUT/autotune.log"
Y6_AUTOTUNE_LOG="$OUT/autotune.log" timeout 600 python bench.py --steps 200 --no-cpu-baseline --dropin-steps 0 --profile-out "$OUT/bench_ops.json" > "$OUT/bench.json" 2> "$OUT/bench.err"
echo "bench rc=$?"; tail -3 "$OUT/bench.err"; cut -c1-300 "$OUT/bench.json"; python - <<PY
import json
d=json.load(open("$OUT/bench.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["achieved"], d["roofline"]["frac"], d["roofline"]["kernel"][-200:])
print({k:(round(v["ms"],3),v["launches"]) for k,v in d["breakdown"].items()})
PY
