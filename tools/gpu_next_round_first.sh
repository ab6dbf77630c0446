#!/usr/bin/env bash
# First GPU visit of the round after round 3 (one box, 8-9 minutes of box time):
#   1. the K-resident 1x1 kernel (kres1x1_c2 / kres1x1_c1, csrc/conv_mfma.hip; written after round 3's last GPU visit, selectable
#      only with Y6_ENABLE_CANDIDATES=1): parity on its own shapes, then its time against every other 1x1 kernel on the layers it
#      was written for (the CSP-SPPF / neck 1x1s of YOLOv6-S at b32: 14-30 us each today)
#      + the packed-maximum form of the SPPF pool kernel (same switch; 27.7 us per launch today, VALU-bound on unpacked fp16 compares)
#      + one head-tail op per level instead of one for all (token `levels`): levels 0 / 1 decode beside the neck's small-map stretch
#      + small-map convs lowered as two ops over the batch halves (token `split`): the 21 layers with at most one work item per CU
#        become two independent chains the schedule runs side by side (the targeted form of the micro-batch A/B of r03r: +2.7 %)
#   2. same-box A/B of the headline with each candidate alone and all together, one-stream and scheduled
#   3. the training-forward schedule once more (r03v: no gain; the `alap` run ended on a different loss - if that repeats, find
#      the undeclared dependence before anybody turns it on)
# Outcome -> DESIGN.md 6 / 9; if (1) is green and (2) gains: drop the Y6_ENABLE_CANDIDATES gate in y6_conv_mfma_supports and add
# the test shapes of tests/test_gpu_ops.py (behind the same variable today) to CONV_SHAPES.
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/${1:-r04a}; mkdir -p "$OUT"
T0=$(date +%s); lap() { echo "-- $1 done at +$(( $(date +%s) - T0 )) s"; }
Y6_ENABLE_CANDIDATES=kres timeout 200 python -m pytest tests/test_gpu_ops.py -m gpu -q --tb=short --timeout 150 -p no:cacheprovider -k conv_all_variants > "$OUT/pytest_kres.log" 2>&1
echo "pytest kres rc=$?"; tail -4 "$OUT/pytest_kres.log" | cut -c1-300; grep -E "variant kres|Error|FAILED" "$OUT/pytest_kres.log" | head -8 | cut -c1-300; lap "kres parity"
# the packed-maximum form of the SPPF pool kernel (same switch): bit-exact pool tests, then the model tests that run a whole SPPF
Y6_ENABLE_CANDIDATES=sppf timeout 200 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -m gpu -q --tb=short --timeout 150 -p no:cacheprovider -k "sppf or pool or model_vs_oracle" > "$OUT/pytest_sppf_pk.log" 2>&1
echo "pytest sppf (packed max) rc=$?"; tail -3 "$OUT/pytest_sppf_pk.log" | cut -c1-300; lap "sppf parity"
# one head-tail op per level (y6_pred_decode_desc.first_anchor / total_anchors; the schedule then decodes levels 0 / 1 early)
Y6_ENABLE_CANDIDATES=levels timeout 200 python -m pytest tests/test_gpu_model.py tests/test_gpu_dropin.py -m gpu -q --tb=short --timeout 150 -p no:cacheprovider -k "model_vs_oracle or two_stream or rebind or full_resolution or dropin" > "$OUT/pytest_levels.log" 2>&1
echo "pytest per-level head tail rc=$?"; tail -3 "$OUT/pytest_levels.log" | cut -c1-300; lap "levels parity"
# small-map convs lowered per batch half (token `split`): two independent chains through the 20x20 / 40x40 stretches
Y6_ENABLE_CANDIDATES=split timeout 250 python -m pytest tests/test_gpu_model.py tests/test_gpu_dropin.py tests/test_gpu_parity_bench.py -m gpu -q --tb=short --timeout 200 -p no:cacheprovider > "$OUT/pytest_split.log" 2>&1
echo "pytest batch-half lowering rc=$?"; tail -3 "$OUT/pytest_split.log" | cut -c1-300; lap "split parity"
# the three-stage form of dma8_c4p1 (variant 40, token `stg3`): the dma parity tests, then its time against the two-stage form on its layers
Y6_ENABLE_CANDIDATES=stg3 timeout 200 python -m pytest tests/test_gpu_ops.py -m gpu -q --tb=short --timeout 150 -p no:cacheprovider -k "dma or conv_all_variants" > "$OUT/pytest_stg3.log" 2>&1
echo "pytest stg3 rc=$?"; tail -3 "$OUT/pytest_stg3.log" | cut -c1-300
Y6_ENABLE_CANDIDATES=stg3 timeout 150 python tools/conv_bench.py --layers 128,128,3,1,80,80,32 256,256,3,1,40,40,32 512,512,3,1,20,20,32 128,128,3,1,40,40,32 256,256,3,1,20,20,32 --variants 26 33 40 --iters 20 --out "$OUT/conv_bench_stg3.json" > "$OUT/conv_bench_stg3.log" 2>&1
grep -v amdgpu "$OUT/conv_bench_stg3.log" | tail -16 | cut -c1-200; lap "stg3"
# variants: 1-6 per-tap, 22/23 streaming, 38/39 K-resident
L="512,256,1,1,20,20,32 256,256,1,1,20,20,32 1024,256,1,1,20,20,32 512,512,1,1,20,20,32 512,128,1,1,20,20,32 384,128,1,1,40,40,32 256,64,1,1,40,40,32 192,64,1,1,80,80,32 128,128,1,1,40,40,32"
Y6_ENABLE_CANDIDATES=kres timeout 200 python tools/conv_bench.py --layers $L --variants 1 2 3 4 5 6 22 23 38 39 --iters 20 --out "$OUT/conv_bench_1x1.json" > "$OUT/conv_bench_1x1.log" 2>&1
grep -v amdgpu "$OUT/conv_bench_1x1.log" | tail -60 | cut -c1-200; lap "1x1 layer table"
run() {  # name, env...
  local name=$1; shift
  env "$@" timeout 90 python bench.py --no-cpu-baseline --dropin-steps 0 --profile-out "$OUT/ops_$name.json" > "$OUT/bench_$name.json" 2> "$OUT/bench_$name.err"
  python - <<PY
import json
try:
    d=json.load(open("$OUT/bench_$name.json")); r=json.load(open("$OUT/ops_$name.json"))["rows"]
    print("$name", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["forward"]["ms"], {k: round(v["ms"], 3) for k, v in d["breakdown"].items()}, d.get("schedule"))
    print("   1x1:", " ".join(f"{x['op']}:{x['variant']}:{x['ms']*1e3:.0f}" for x in r if x["kind"] == "conv" and x["ksize"] == 1))
except Exception as e: print("no result", e)
PY
}
run base1
run kres1 Y6_ENABLE_CANDIDATES=kres
run sppf1 Y6_ENABLE_CANDIDATES=sppf
run levels1 Y6_ENABLE_CANDIDATES=levels
run stg3 Y6_ENABLE_CANDIDATES=stg3
run split1 Y6_ENABLE_CANDIDATES=split
run split400 Y6_ENABLE_CANDIDATES=split Y6_SPLIT_MAX_HW=400
run split6400 Y6_ENABLE_CANDIDATES=split Y6_SPLIT_MAX_HW=6400      # + the 80x80 layers: their last partial round (3.1 -> 4) fills with the other half's blocks
run split25600 Y6_ENABLE_CANDIDATES=split Y6_SPLIT_MAX_HW=25600    # every conv: micro-batching inside one plan
run split_levels Y6_ENABLE_CANDIDATES=split,levels
run all1 Y6_ENABLE_CANDIDATES=all
run base2
run all2 Y6_ENABLE_CANDIDATES=all
run base_1stream Y6_SCHED_STREAMS=1
run all_1stream Y6_SCHED_STREAMS=1 Y6_ENABLE_CANDIDATES=all
lap "headline A/B"
# int8 plan (configs[4]) under the two-stream schedule (twin-aware access lists, token `i8sched`): parity first
Y6_ENABLE_CANDIDATES=i8sched timeout 200 python -m pytest tests/test_gpu_int8.py -m gpu -q --tb=short --timeout 150 -p no:cacheprovider > "$OUT/pytest_int8_sched.log" 2>&1
echo "pytest int8 (scheduled) rc=$?"; tail -3 "$OUT/pytest_int8_sched.log" | cut -c1-300
for n in i8_one i8_sched i8_one2 i8_sched2; do
  case $n in *sched*) E="Y6_ENABLE_CANDIDATES=i8sched";; *) E="Y6_DUMMY=1";; esac
  env $E timeout 120 python bench.py --model yolov6s_qa --int8 --no-cpu-baseline --dropin-steps 0 > "$OUT/bench_$n.json" 2> "$OUT/bench_$n.err"
  python -c "import json; d=json.load(open('$OUT/bench_$n.json')); print('$n', d['value'], d['ms_per_step'], d['self_check'], d.get('schedule'))" 2>/dev/null || echo "$n: no result"
done
lap "int8 schedule"
for n in one asap alap one2 alap2; do
  case $n in one*) E="Y6_TRAIN_FWD_STREAMS=1";; asap*) E="Y6_TRAIN_FWD_STREAMS=2 Y6_TRAIN_FWD_POLICY=asap";; *) E="Y6_TRAIN_FWD_STREAMS=2 Y6_TRAIN_FWD_POLICY=alap";; esac
  # one tuning cache for all five runs: identical kernel choices, so any difference in the losses is the schedule's
  env $E Y6_AUTOTUNE_CACHE="$PWD/$OUT/autotune_train.cache" timeout 120 python bench.py --mode train > "$OUT/bench_train_$n.json" 2> "$OUT/bench_train_$n.err"
  python -c "import json; d=json.load(open('$OUT/bench_train_$n.json')); print('train $n', d['value'], d['ms_per_step'], d['loss'])" 2>/dev/null || echo "train $n: no result"
done
lap "training forward schedule"
echo done
