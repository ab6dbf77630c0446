import os, sys, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import test_gpu_model as T
from yolov6_amd.utils import synth
for case in ("tiny", "l6_tiny", "s", "m_tiny"):
    for ns in ("2",):
        cfg, meta, sd, m = T._build(case, deploy=True)
        x = synth.synth_images(meta["batch"], meta["size"], seed=8).to("cuda:0").half()
        plan = m.compile(x)
        eager = plan.run().clone()
        os.environ["Y6_GRAPH_STREAMS"] = ns
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            plan.capture()
            ok = all(torch.equal(plan.run().clone(), eager) for _ in range(5))
        s.synchronize()
        print(case, ns, "streams:", "identical" if ok else "MISMATCH", flush=True)
