#!/usr/bin/env bash
# Round 3, visit j: pw pairs with the three-deep input ring; QARepVGG / MBLA training graphs (block tests + teacher-forced per-op);
# the whole GPU suite; bench.
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/r03j; mkdir -p $OUT
timeout 300 python tools/fused_bench.py 2>/dev/null | tail -1 | tee $OUT/fused_bench.log
timeout 2400 python -m pytest tests -m gpu -q --tb=short --timeout 900 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1
echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log; tail -30 $OUT/pytest_gpu.log | cut -c1-400
timeout 600 python bench.py --no-cpu-baseline --dropin-steps 20 --profile-out $OUT/bench_ops.json > $OUT/bench.json 2> $OUT/bench.err
echo "bench rc=$?"; python - <<PY
import json
d=json.load(open("$OUT/bench.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], {k:(v["ms"],v["launches"]) for k,v in d["breakdown"].items()}, d["dropin_api"])
PY
