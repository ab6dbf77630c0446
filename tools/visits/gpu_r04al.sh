#!/usr/bin/env bash
# Round 4, visit al: the last GPU seconds of the round on the FINAL library (rebuilt after visit ak: int8 variant 13 renumbered, its
# stride-1 sibling deleted, y6_conv2d_i8_variant exported) - the register-fed int8 op tests, then as much of the int8 file as fits.
set -u
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/${1:-r04al}; mkdir -p "$OUT"
timeout -k 3 30 python -m pytest tests/test_gpu_int8.py -q -m gpu -k "wreg" > "$OUT/pytest_int8_wreg_final.log" 2>&1
echo "wreg op tests: rc=$? $(tail -1 "$OUT/pytest_int8_wreg_final.log")"
timeout -k 3 50 python -m pytest tests/test_gpu_int8.py -q -m gpu -k "not wreg" > "$OUT/pytest_int8_rest_final.log" 2>&1
echo "rest of the int8 file: rc=$? $(tail -1 "$OUT/pytest_int8_rest_final.log")"
