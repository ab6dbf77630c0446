#!/usr/bin/env bash
# Round 4, visit w: cache policy of conv_wreg's output stores (Y6_WREG_STORE = 0 | 16 sc1 | 2 nt | 18 nt sc1 | 17 sc0 sc1): layer table, headline
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/${1:-r04w}; mkdir -p "$OUT"
L="128,128,3,1,80,80,32 256,256,3,1,40,40,32 512,512,3,1,20,20,32 128,128,3,1,40,40,32 256,256,3,1,20,20,32"
for p in 0 16 2 18 17; do
  echo "== store policy $p"
  Y6_WREG_STORE=$p timeout 200 python tools/conv_bench.py --data relu --layers $L --variants 39 40 --iters 100 --out "$OUT/conv_bench_st$p.json" 2>&1 | grep -v amdgpu | cut -c1-200
done
run() {  # name, env...
  local name=$1; shift
  env "$@" timeout 120 python bench.py --no-cpu-baseline --no-train-sub --windows 1 --dropin-steps 0 --profile-out "$OUT/ops_$name.json" > "$OUT/bench_$name.json" 2> "$OUT/bench_$name.err"
  python - <<PY
import json
try:
    d=json.load(open("$OUT/bench_$name.json")); r=json.load(open("$OUT/ops_$name.json"))["rows"]
    print("$name", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["forward"]["ms"], {k: round(v["ms"], 3) for k, v in d["breakdown"].items()}, d.get("self_check"))
except Exception as e: print("$name: no result", e)
PY
}
for p in 0 16 2 18 17 0; do run st${p}_$RANDOM Y6_WREG_STORE=$p; done
echo done
