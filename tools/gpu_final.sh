#!/usr/bin/env bash
# Last visit of a round: a short soak of the driver's command, the whole -m gpu suite, then - LAST, on the library the suite just
# validated - the driver's exact command.   usage: tools/gpu_final.sh <tag> [soak runs]
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/${1:-final}; mkdir -p $OUT
T0=$(date +%s); lap() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
bash tools/gpu_soak.sh ${1:-final}_soak ${2:-10} | tail -$(( ${2:-10} + 1 ))
lap soak
timeout 300 python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 1500 python3 -m pytest tests -m gpu -q --tb=short --timeout 900 -p no:cacheprovider --durations 8 > $OUT/pytest_gpu.log 2>&1
echo "pytest rc=$?"; tail -14 $OUT/pytest_gpu.log | cut -c1-200
lap tests
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err
echo "driver command rc=$?"; cut -c1-420 $OUT/bench_driver.json
python3 -c "
import json; d=json.load(open('$OUT/bench_driver.json')); print('value', d['value'], 'sequential', d['sequential']['value'], 'frac', d['roofline']['frac'], 'cpu', d['cpu_baseline']['value'], d['cpu_baseline']['kind'], 'train', d['train'].get('value'), d['train'].get('ms_per_step'), 'supervisor', d['supervisor'], 'self_check', d['self_check'])"
lap "driver command"
