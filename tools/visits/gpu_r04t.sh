#!/usr/bin/env bash
# Round 4, visit t: reproducible training (shape-derived variants, deterministic pool backward): the two-process test, the SPPF /
# training tests, training step with shape-derived vs timed variants, inference with shape-derived variants
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/${1:-r04t}; mkdir -p "$OUT"
timeout 1200 python -m pytest tests/test_gpu_training.py -m gpu -q --tb=short --timeout 900 -p no:cacheprovider -k "same_bits or sppf or full_training_steps or training_graph_forward" > "$OUT/pytest_train.log" 2>&1
echo "pytest train rc=$?"; tail -12 "$OUT/pytest_train.log" | cut -c1-600
for name in shape1 shape2; do
  timeout 300 python bench.py --mode train --steps 10 --warmup 3 > "$OUT/train_$name.json" 2> "$OUT/train_$name.err"
  python - <<PY
import json
try:
    d=json.load(open("$OUT/train_$name.json")); print("$name", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["loss"]["first"], d["loss"]["last"], d["variants"], d["loss"]["bits"][:2])
except Exception as e: print("$name: no result", e)
PY
done
timeout 300 python bench.py --mode train --steps 10 --warmup 3 --train-autotune > "$OUT/train_tuned.json" 2> "$OUT/train_tuned.err"
python - <<PY
import json
try:
    d=json.load(open("$OUT/train_tuned.json")); print("tuned", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["loss"]["first"], d["loss"]["last"], d["variants"])
except Exception as e: print("tuned: no result", e)
PY
for name in infer_tuned infer_shape; do
  extra=""; [ $name = infer_shape ] && extra="--no-autotune"
  timeout 120 python bench.py --no-cpu-baseline --dropin-steps 0 $extra > "$OUT/bench_$name.json" 2> "$OUT/bench_$name.err"
  python - <<PY
import json
try:
    d=json.load(open("$OUT/bench_$name.json")); print("$name", d["value"], d["ms_per_step"], d["roofline"]["frac"], {k: round(v["ms"], 3) for k, v in d["breakdown"].items()})
except Exception as e: print("$name: no result", e)
PY
done
echo done
