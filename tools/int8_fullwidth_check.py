#!/usr/bin/env python3
"""int8 plan of the FULL-WIDTH YOLOv6-S-QA model at 640x640 against oracle/int8_oracle.py (run on a GPU box; the oracle's
float64 convolutions of a full-width model take a minute or two per image on the host cores, which is why this is a tool and not
part of the `-m gpu` suite):

    python tools/int8_fullwidth_check.py [batch=1] [size=640]

Prints the device-vs-oracle calibration table deviation, the end-to-end class-score / box deviation of the int8 plan from the
int8 oracle (same scales), the oracle's own quantisation error against the fp16 graph, and the NMS agreement."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from oracle import nms_oracle, synth
from oracle.int8_oracle import Int8Oracle
from oracle.model_oracle import Oracle
from yolov6_amd import quant
from yolov6_amd.configs import get_config
from yolov6_amd.models.yolo import build_model
from yolov6_amd.utils.nms import non_max_suppression
from yolov6_amd.utils.torch_utils import fuse_model, switch_to_deploy

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 1
size = int(sys.argv[2]) if len(sys.argv) > 2 else 640
dev = "cuda:0"
cfg = get_config("yolov6s_qa")
model = build_model(cfg, 80, "cpu").eval()
model.load_state_dict(synth.synth_state_dict(model.state_dict(), seed=0))
switch_to_deploy(fuse_model(model))
model = model.to(dev).half()
sd = {k: v.detach().float().cpu() for k, v in model.state_dict().items()}     # the fp16 values the int8 path quantises
cal = [synth.synth_images(batch, size, seed=100 + i) for i in range(2)]
x = synth.synth_images(batch, size, seed=1)
orc = Int8Oracle(cfg, sd, 80)
table_ref = orc.calibrate(cal)
table = quant.calibrate(model, [c.to(dev).half() for c in cal])
print("calibration: %d scales, device vs oracle max rel %.3e" % (len(table), max(abs(a - b) / b for a, b in zip(table, table_ref))))
quant.quantize(model, table_ref)
det = model(x.to(dev).half())[0]
torch.cuda.synchronize()
with torch.no_grad():
    ref, _ = orc.forward(x)
    d16, _ = Oracle(cfg, sd, 80, emulate_fp16=True).forward(x)
d = (det.float().cpu() - ref).abs()
print("int8 HIP vs int8 oracle: class scores %.3e, boxes %.3e px (p99.9 %.3e)" % (float(d[..., 5:].max()), float(d[..., :4].max()),
      float(np.quantile(d[..., :4].numpy(), 0.999))))
print("quantisation error of the oracle itself vs the fp16 graph: class scores %.3e, boxes %.3e px" % (
      float((ref[..., 5:] - d16[..., 5:]).abs().max()), float((ref[..., :4] - d16[..., :4]).abs().max())))
thr = float(np.quantile(ref[0, :, 5:].max(-1).values.numpy(), 0.97))
out = non_max_suppression(det, conf_thres=thr, iou_thres=0.65, max_det=300)
want = nms_oracle.non_max_suppression(det.float().cpu().numpy(), thr, 0.65, max_det=300)
print("NMS of the int8 detections: device == oracle:", all(np.array_equal(o.cpu().numpy(), w.astype(np.float32)) for o, w in zip(out, want)))
