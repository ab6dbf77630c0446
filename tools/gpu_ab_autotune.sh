# autotune timing A/B on one box: in-context (default) vs burst (Y6_AUTOTUNE_BURST=1)
run() { python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', d['value'], d['ms_per_step'], {k:v['ms'] for k,v in d['breakdown'].items() if k.startswith('conv')}, d['roofline']['kernel'][60:])"; }
run context
Y6_AUTOTUNE_BURST=1 run burst
run context_again
Y6_AUTOTUNE_BURST=1 run burst_again
