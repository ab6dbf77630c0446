import torch, sys, os
sys.path.insert(0, os.getcwd())
from yolov6_amd.engine import PlanBuilder, NCHWInput
x = torch.rand(32,3,640,640, device="cuda").half()
w = torch.randn(32,3,3,3)*0.2; b = torch.randn(32)*0.1
pb = PlanBuilder("cuda:0"); o = pb.conv(NCHWInput(x), w, b, stride=2, act="relu"); plan = pb.finalize(o, autotune=False)
for _ in range(3): plan.run()
torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): plan.run()
e1.record(); torch.cuda.synchronize(); ms=e0.elapsed_time(e1)/20
print("stem", "no_v4" if os.environ.get("Y6_STEM_NO_V4") else "v4", round(ms*1000,1), "us", round(289e6/ms/1e6,1), "GB/s")
