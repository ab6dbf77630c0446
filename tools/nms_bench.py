"""Time y6_nms on the bench-shaped prediction tensor (b32, 8400 anchors, 80 classes), stage by stage:
each stage count runs in its own process (Y6_NMS_STOP_AFTER is read once by the library)."""
import os, subprocess, sys
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    import torch
    sys.path.insert(0, os.getcwd())
    from yolov6_amd.utils.nms import nms_raw
    from yolov6_amd.utils import synth
    frac = float(sys.argv[2])
    pred = synth.synth_predictions(32, 8400, 80, seed=0, frac=frac).cuda()
    for _ in range(3): out = nms_raw(pred, 0.03, 0.65, multi_label=True, max_det=300)
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): out = nms_raw(pred, 0.03, 0.65, multi_label=True, max_det=300)
    e1.record(); torch.cuda.synchronize()
    print("nms frac %g stages<=%s: %.1f us  kept mean %.1f" % (frac, os.environ.get("Y6_NMS_STOP_AFTER", "all"), e0.elapsed_time(e1) / 20 * 1000, out[2].float().mean().item()), flush=True)
else:
    frac = sys.argv[1] if len(sys.argv) > 1 else "0.02"
    for stop in ("0", "1", "2", "3", "4"):
        env = dict(os.environ, Y6_NMS_STOP_AFTER=stop)
        subprocess.run([sys.executable, __file__, "--child", frac], env=env, timeout=120, stderr=subprocess.DEVNULL)
