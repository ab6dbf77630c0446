# event-granularity A/B on one box: per-op events vs per-class events (same tuning choices via a cache file)
run() { python bench.py --steps 30 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', d['value'], d['ms_per_step'], 'fwd', d['forward']['ms'], {k:v['ms'] for k,v in d['breakdown'].items()})"; }
export Y6_AUTOTUNE_CACHE=/tmp/ab_events.cache
run tune_op
run op
Y6_TIMED_EVENTS=class run class
run op_again
Y6_TIMED_EVENTS=class run class_again
