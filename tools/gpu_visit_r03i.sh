#!/usr/bin/env bash
# Round 3, visit i: where the fused pairs spend their time (phase probes), QARepVGG training form tests, bench.
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/r03i; mkdir -p $OUT
for p in 0 1 2 4 8 3 7 15; do
  Y6_FUSED_PROBE=$p timeout 300 python tools/fused_bench.py 2>/dev/null | tail -1 | tee -a $OUT/fused_probes.log
done
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q --tb=short --timeout 600 -p no:cacheprovider -k "fused or stem" > $OUT/pytest_fused.log 2>&1
echo "pytest fused rc=$?" | tee -a $OUT/pytest_fused.log; tail -4 $OUT/pytest_fused.log | cut -c1-300
timeout 1500 python -m pytest tests/test_gpu_training.py -m gpu -q --tb=short --timeout 900 -p no:cacheprovider -k "block_training_graph or training_graph_forward_backward" -s > $OUT/pytest_qa_train.log 2>&1
echo "pytest qa train rc=$?" | tee -a $OUT/pytest_qa_train.log; grep -v "^{" $OUT/pytest_qa_train.log | tail -30 | cut -c1-300
timeout 600 python bench.py --no-cpu-baseline --dropin-steps 0 --profile-out $OUT/bench_ops.json > $OUT/bench.json 2> $OUT/bench.err
echo "bench rc=$?"; python - <<PY
import json
d=json.load(open("$OUT/bench.json"))
print(d["value"], d["ms_per_step"], {k:(v["ms"],v["launches"]) for k,v in d["breakdown"].items()})
PY
