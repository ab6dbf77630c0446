#!/usr/bin/env bash
# Full validation visit: smoke, the whole -m gpu suite, then the training A/B of the specialised epilogues.
#   usage: tools/gpu_full.sh <tag>
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/${1:-full}; mkdir -p $OUT
T0=$(date +%s); lap() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
timeout 300 python3 __graft_entry__.py > /dev/null 2>&1; timeout 300 python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
lap smoke
timeout 1500 python3 -m pytest tests -m gpu -q --tb=short --timeout 900 -p no:cacheprovider --durations 12 > $OUT/pytest_gpu.log 2>&1
echo "pytest rc=$?"; tail -25 $OUT/pytest_gpu.log
lap tests
for rep in 1 2; do
  timeout 200 python3 bench.py --no-supervisor --mode train --steps 15 --warmup 4 > $OUT/train_special$rep.json 2> $OUT/train_special$rep.err
  Y6_WREG_GENERAL_EPI=1 Y6_CONV_GENERAL_EPI=1 timeout 200 python3 bench.py --no-supervisor --mode train --steps 15 --warmup 4 > $OUT/train_general$rep.json 2> $OUT/train_general$rep.err
done
python3 - <<PY
import json
for n in ("special1","general1","special2","general2"):
    try:
        d=json.load(open("$OUT/train_%s.json"%n)); print("train", n, d["value"], d["ms_per_step"], d["roofline"]["frac"], d["loss"]["first"], d["loss"]["last"])
    except Exception as e: print(n, "failed", e)
PY
lap train
