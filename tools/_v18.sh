export TMPDIR=/tmp
mkdir -p gpurun_out/r05q
Y6_LIB_PATH=$PWD/tools/_probe/libyolov6_hip_wregprobe1.so timeout 120 python3 tools/cold_launch_trace.py 2>/dev/null | tail -1 > gpurun_out/r05q/cold_trace.json
python3 - <<PY
import json
d=json.load(open("gpurun_out/r05q/cold_trace.json"))
for k,v in d.items():
    print(k)
    for r in v:
        print("  ", {kk:r[kk] for kk in ("event_us","blocks","start_after_prev_end_us","end_after_prev_end_us","lifetime_us")})
        print("     ", r["block0_cycles_between_tags"][:24])
PY
