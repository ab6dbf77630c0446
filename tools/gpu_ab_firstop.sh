#!/usr/bin/env bash
# first-launch penalty of a kernel at the start of a stage: same box, three candidate sets
set -u
OUT=gpurun_out/${1:-firstop}; mkdir -p "$OUT"; export TMPDIR=/tmp
BASE="7,8,9,12,13,14,15,16,17,18,19,20,21,24,30"
for cfg in default:$BASE hc16:$BASE,28,29 old:$BASE,25,26,27,28,29; do
  name=${cfg%%:*}; ex=${cfg#*:}
  Y6_AUTOTUNE_EXCLUDE="$ex" timeout 600 python bench.py --steps 200 --no-cpu-baseline --dropin-steps 0 --profile-out "$OUT/ops_$name.json" > "$OUT/bench_$name.json" 2> "$OUT/bench_$name.err"
  python - <<PY
import json
d=json.load(open("$OUT/bench_$name.json")); r=json.load(open("$OUT/ops_$name.json"))["rows"]
print("$name", d["value"], d["ms_per_step"], "3x3s1", round(d["breakdown"]["conv3x3s1"]["ms"],3))
print("   ", " ".join(f"{x['op']}:{x['variant']}:{x['ms']*1e3:.0f}" for x in r if x["op"] in (2,3,5,6,10,11,17,18,20,25,33,34,43,44,48,49,53,54)))
PY
done
