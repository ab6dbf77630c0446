#!/usr/bin/env bash
# One GPU-box visit: smoke, -m gpu tests, bench (+ autotune log, per-op table), rocprofv3 kernel stats.
# Everything of interest lands under gpurun_out/ (merged back by gpurun).
#   usage: tools/gpu_round.sh [tag] [what...]    what in {smoke,tests,bench,prof,pmc,train,trainprof,l6,int8} (default: smoke tests bench prof)
set -u
TAG=${1:-r01}
shift || true
WHAT=${*:-smoke tests bench prof}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
export HSA_ENABLE_IPC_MODE_LEGACY=0
has() { [[ " $WHAT " == *" $1 "* ]]; }

{
  echo "== device"; rocminfo 2>/dev/null | grep -E "Marketing Name|gfx9|Compute Unit" | head -6
  python -c "import torch;print('torch',torch.__version__,torch.cuda.is_available(),torch.cuda.get_device_name(0) if torch.cuda.is_available() else '')"
  nproc
} > "$OUT/device.txt" 2>&1

if has smoke; then
  echo "== smoke"; timeout 600 python __graft_entry__.py smoke > "$OUT/smoke.log" 2>&1; echo "smoke rc=$?" | tee -a "$OUT/smoke.log"; tail -3 "$OUT/smoke.log"
fi
if has tests; then
  echo "== tests"; timeout 2400 python -m pytest tests -m gpu -q --tb=short --timeout 600 -p no:cacheprovider > "$OUT/pytest_gpu.log" 2>&1
  echo "pytest rc=$?" | tee -a "$OUT/pytest_gpu.log"; tail -40 "$OUT/pytest_gpu.log"
fi
if has bench; then
  echo "== bench"; rm -f "$OUT/autotune.log" "$OUT/autotune.cache"
  Y6_AUTOTUNE_CACHE="$PWD/$OUT/autotune.cache" Y6_AUTOTUNE_LOG="$OUT/autotune.log" timeout 1200 python bench.py --profile-out "$OUT/bench_ops.json" > "$OUT/bench.json" 2> "$OUT/bench.err"
  echo "bench rc=$?"; tail -2 "$OUT/bench.err"; cat "$OUT/bench.json"
fi
if has prof; then
  echo "== rocprofv3 kernel stats"
  # the bench leg's tuning choices are reused, so the stats hold the chosen kernels only
  ( cd /tmp && Y6_AUTOTUNE_CACHE="$OLDPWD/$OUT/autotune.cache" timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/$OUT/prof" -o bench -- python "$OLDPWD/bench.py" --steps 8 --warmup 2 --no-cpu-baseline --dropin-steps 0 --no-verify > "$OLDPWD/$OUT/prof_bench.json" 2> "$OLDPWD/$OUT/prof.err" )
  echo "prof rc=$?"; find "$OUT/prof" -name "*kernel_stats*" | head -3
  f=$(find "$OUT/prof" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -25 "$f"
  # keep the merge-back small: drop the raw trace, keep stats
  find "$OUT/prof" -name "*kernel_trace.csv" -size +20M -delete
fi
if has train; then
  echo "== train bench (configs[2])"
  timeout 1200 python bench.py --mode train --profile-out "$OUT/train_ops.json" > "$OUT/bench_train.json" 2> "$OUT/bench_train.err"
  echo "train bench rc=$?"; tail -2 "$OUT/bench_train.err"; cut -c1-900 "$OUT/bench_train.json"
fi
if has trainprof; then
  echo "== rocprofv3 kernel stats of the training step"
  ( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/$OUT/prof_train" -o train -- python "$OLDPWD/bench.py" --mode train --steps 5 --warmup 2 --no-autotune > "$OLDPWD/$OUT/prof_train.json" 2> "$OLDPWD/$OUT/prof_train.err" )
  echo "trainprof rc=$?"; f=$(find "$OUT/prof_train" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -16 "$f"
  find "$OUT/prof_train" -name "*kernel_trace.csv" -size +20M -delete
fi
if has l6; then
  echo "== L6 1280 b8 (configs[3])"
  timeout 1200 python bench.py --model yolov6l6 --size 1280 --batch 8 --steps 50 --warmup 5 --no-cpu-baseline --dropin-steps 10 > "$OUT/bench_l6.json" 2> "$OUT/bench_l6.err"
  echo "l6 rc=$?"; tail -2 "$OUT/bench_l6.err"; cut -c1-700 "$OUT/bench_l6.json"
fi
if has int8; then
  echo "== int8 (configs[4]): tests, then fp16 and int8 plans of the same S-qa model in one visit"
  timeout 900 python -m pytest tests/test_gpu_int8.py -q --tb=short --timeout 600 -p no:cacheprovider -s > "$OUT/pytest_int8.log" 2>&1
  echo "pytest int8 rc=$?" | tee -a "$OUT/pytest_int8.log"; tail -25 "$OUT/pytest_int8.log"
  timeout 900 python bench.py --model yolov6s_qa --no-cpu-baseline --dropin-steps 0 --profile-out "$OUT/bench_qa_fp16_ops.json" > "$OUT/bench_qa_fp16.json" 2> "$OUT/bench_qa_fp16.err"
  echo "qa fp16 rc=$?"; tail -2 "$OUT/bench_qa_fp16.err"; cut -c1-400 "$OUT/bench_qa_fp16.json"
  timeout 900 python bench.py --model yolov6s_qa --int8 --no-cpu-baseline --dropin-steps 0 --profile-out "$OUT/bench_qa_int8_ops.json" > "$OUT/bench_qa_int8.json" 2> "$OUT/bench_qa_int8.err"
  echo "qa int8 rc=$?"; tail -2 "$OUT/bench_qa_int8.err"; cut -c1-1500 "$OUT/bench_qa_int8.json"
fi
if has pmc; then
  echo "== rocprofv3 pmc (separate passes)"
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d "$OLDPWD/$OUT/pmc_fetch" -o b -- python "$OLDPWD/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --dropin-steps 0 --no-verify > /dev/null 2> "$OLDPWD/$OUT/pmc_fetch.err" )
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d "$OLDPWD/$OUT/pmc_write" -o b -- python "$OLDPWD/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --dropin-steps 0 --no-verify > /dev/null 2> "$OLDPWD/$OUT/pmc_write.err" )
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY -d "$OLDPWD/$OUT/pmc_sq" -o b -- python "$OLDPWD/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --dropin-steps 0 --no-verify > /dev/null 2> "$OLDPWD/$OUT/pmc_sq.err" )
  ls "$OUT"/pmc_*
fi
du -sh "$OUT"
