#!/usr/bin/env bash
# Round 4, visit ag: smoke() against the reference goldens, the tests touched since the last full run, default bench line
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/${1:-r04ag}; mkdir -p "$OUT"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu | tail -3 | cut -c1-400
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_training.py -m gpu -q --tb=short --timeout 600 -p no:cacheprovider -k "replaced_parameter or rebind or bnact or bn_train or full_training_steps or same_bits or training_graph_forward" 2>&1 | tail -5 | cut -c1-400
timeout 900 python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"; echo "bench rc=$?"
python - <<PY
import json
try:
    d=json.load(open("$OUT/bench_default.json"))
    print("default", d["value"], d["ms_per_step"], "seq", d["sequential"]["value"], d["windows"]["spread_pct"], d["roofline"]["frac"], d.get("self_check"), (d.get("train") or {}).get("ms_per_step"), d["cpu_baseline"]["value"])
except Exception as e: print("no result", e); print(open("$OUT/bench_default.err").read()[-1500:])
PY
echo done
