run() { python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', d['value'], d['ms_per_step'], {k:v['ms'] for k,v in d['breakdown'].items()}, d['roofline']['kernel'][60:])"; }
run all
Y6_AUTOTUNE_EXCLUDE=15,16,17,18,19 run no_pipe8
Y6_AUTOTUNE_EXCLUDE=15,16,18,19 run no_scratch_no_s2
Y6_AUTOTUNE_EXCLUDE=7,8,9,12,13,14,15,16,17,18,19 run few
run all_again
