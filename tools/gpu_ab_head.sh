#!/usr/bin/env bash
# one visit: NMS / conv / model tests, then the headline bench with the head's cls/reg convs merged vs separate
set -u
OUT=gpurun_out/${1:-abhead}; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_nms_tal.py tests/test_gpu_ops.py tests/test_gpu_model.py tests/test_gpu_parity_bench.py -q --tb=short --timeout 600 -p no:cacheprovider -x > "$OUT/pytest.log" 2>&1
echo "pytest rc=$?"; tail -8 "$OUT/pytest.log"
for m in merged separate merged separate; do
  if [ $m = separate ]; then export Y6_HEAD_NO_MERGE=1; else unset Y6_HEAD_NO_MERGE; fi
  timeout 600 python bench.py --steps 200 --no-cpu-baseline --dropin-steps 0 --profile-out "$OUT/ops_$m.json" > "$OUT/bench_$m.json" 2> "$OUT/bench_$m.err"
  python - <<PY
import json
d=json.load(open("$OUT/bench_$m.json"))
print("$m", d["value"], d["ms_per_step"], d["roofline"]["achieved"], {k:(round(v["ms"],3),v["launches"]) for k,v in d["breakdown"].items()})
PY
done
