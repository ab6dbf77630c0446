#!/usr/bin/env bash
# Round 4, visit u: state of HEAD: smoke, the whole GPU suite, the default bench line (3 windows, cpu_baseline, train sub-object),
# rocprofv3 kernel stats of the same command, HBM traffic of the kernel classes (PMC passes)
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/${1:-r04u}; mkdir -p "$OUT"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1; echo "smoke rc=$?"; tail -2 "$OUT/smoke.log" | cut -c1-300
timeout 2400 python -m pytest tests -m gpu -q --tb=short --timeout 900 -p no:cacheprovider -x > "$OUT/pytest_gpu.log" 2>&1
echo "pytest gpu rc=$?"; tail -8 "$OUT/pytest_gpu.log" | cut -c1-400
timeout 900 python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"; echo "bench rc=$?"
python - <<PY
import json
try:
    d=json.load(open("$OUT/bench_default.json"))
    print("default", d["value"], d["ms_per_step"], d["windows"], d["roofline"]["frac"], {k: round(v["ms"], 3) for k, v in d["breakdown"].items()}, d["cpu_baseline"], d.get("train"))
except Exception as e: print("default: no result", e)
PY
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/$OUT/rocprof" -o infer -- python "$OLDPWD/bench.py" --steps 20 --warmup 5 --windows 1 --no-cpu-baseline --no-train-sub --dropin-steps 0 --no-verify > "$OLDPWD/$OUT/rocprof_bench.json" 2> "$OLDPWD/$OUT/rocprof.err" ); echo "rocprof rc=$?"
find "$OUT/rocprof" -name "*kernel_stats.csv" | head -2
find "$OUT/rocprof" -name "*kernel_trace.csv" -size +20M -delete
bash tools/gpu_pmc_traffic.sh ${1:-r04u}/pmc 2>&1 | tail -30
echo done
