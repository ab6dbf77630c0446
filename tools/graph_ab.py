"""A/B on one box: eager plan.run(), eager run_timed (per-op / per-class events) and hipGraph replay of the same plan,
forward only, same tuning.  Prints ms per forward for each mode."""
import os, sys, time, torch
sys.path.insert(0, os.getcwd())
import bench
class A: pass
args = A(); args.batch, args.size, args.model = 32, 640, "yolov6s"
dev = torch.device("cuda:0")
cfg, sd, model, x = bench.build_model_and_input(args, dev)
bench.calibrate_head_bias(model, x)
plan = model.compile(x, autotune=True)
def timeit(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("eager run        %.4f ms" % timeit(plan.run))
plan.timing_begin(40)
print("eager run_timed  %.4f ms (%s events)" % (timeit(plan.run_timed), os.environ.get("Y6_TIMED_EVENTS", "op")))
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for ns in ("1", "2"):
        os.environ["Y6_GRAPH_STREAMS"] = ns
        plan.capture()
        print("graph replay, %s capture stream(s)  %.4f ms" % (ns, timeit(plan.run)))
