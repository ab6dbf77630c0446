"""CPU baseline of the bench, `kind: "reference"`: the UNMODIFIED reference modules (read-only, imported from /root/reference)
timed on the host cores - BASELINE.md 2 / SURVEY 8d: `yolov6/models/yolo.py:136` build_model's Model, `utils/torch_utils.py:85`
fuse_model, `layers/common.py:302` switch_to_deploy, `utils/nms.py:31-105` non_max_suppression, `assigners/tal_assigner.py`.

TEST / BASELINE INFRASTRUCTURE, like everything under oracle/: only bench.py's `cpu_baseline` leg calls it, and only when the
reference checkout exists (this build container; a GPU box has none and times the oracle port instead, `kind: "port"`).
torchvision is not installed: the one `torchvision.ops.nms` call of the reference's NMS is served by oracle/nms_oracle.nms (numpy),
everything around it is the reference's own code (the same arrangement as tests/golden/gen_golden.py)."""
import os
import statistics
import sys
import time
import types

REF = "/root/reference"


def available():
    return os.path.isdir(os.path.join(REF, "yolov6", "models"))


def _install_stubs():
    import torch
    from oracle import nms_oracle
    cv2 = types.ModuleType("cv2")
    cv2.setNumThreads = lambda n: None
    sys.modules.setdefault("cv2", cv2)
    tv = types.ModuleType("torchvision")
    ops = types.ModuleType("torchvision.ops")

    def nms(boxes, scores, iou_threshold):
        keep = nms_oracle.nms(boxes.detach().cpu().numpy(), scores.detach().cpu().numpy(), float(iou_threshold))
        return torch.from_numpy(keep).to(boxes.device)

    ops.nms = nms
    tv.ops = ops
    sys.modules.setdefault("torchvision", tv)
    sys.modules.setdefault("torchvision.ops", ops)


def build_reference_model(model_name, sd_train, shift, num_classes=80):
    """The reference's Model from the reference's own config file, the bench's synthetic weights, eval + deploy form."""
    import torch
    from yolov6_amd.configs import load_config
    if REF not in sys.path:
        sys.path.insert(0, REF)
    _install_stubs()
    from yolov6.layers.common import RepVGGBlock
    from yolov6.models.yolo import Model
    from yolov6.utils.torch_utils import fuse_model
    cfile = {"yolov6s": "configs/yolov6s.py", "yolov6n": "configs/yolov6n.py", "yolov6l6": "configs/yolov6l6.py",
             "yolov6s_qa": "configs/qarepvgg/yolov6s_qa.py"}[model_name]
    cfg = load_config(os.path.join(REF, cfile))
    model = Model(cfg, channels=3, num_classes=num_classes).eval()
    model.load_state_dict(sd_train)
    with torch.no_grad():
        fuse_model(model)
        for m in model.modules():
            if isinstance(m, RepVGGBlock):
                m.switch_to_deploy()
        for conv in model.detect.cls_preds:
            conv.bias.add_(shift)
    return model


def time_reference(model_name, size, n, sd_train, shift, conf, iou, max_det, passes=3):
    """1 warm-up + `passes` timed passes of: forward of n images (fp32, deploy form), the reference's non_max_suppression with the
    eval thresholds, the reference's TaskAlignedAssigner on n images (8400 anchors, up to 40 boxes).  -> dict for the bench line."""
    import numpy as np
    import torch
    from oracle import synth
    model = build_reference_model(model_name, sd_train, shift)
    from yolov6.assigners.tal_assigner import TaskAlignedAssigner
    from yolov6.utils.nms import non_max_suppression
    x = synth.synth_images(n, size, seed=0)
    fs, st = [(size // s, size // s) for s in (8, 16, 32)], [8, 16, 32]
    g = np.random.default_rng(0)
    t = synth.synth_tal_inputs(n, fs, st, 80, 40, seed=9, n_valid=[int(v) for v in g.integers(1, 41, n)], img=size)
    tal = TaskAlignedAssigner(topk=13, num_classes=80, alpha=1.0, beta=6.0)
    fwd, nms, talt = [], [], []
    with torch.no_grad():
        for rep in range(passes + 1):
            t0 = time.perf_counter()
            det = model(x)[0]
            t1 = time.perf_counter()
            non_max_suppression(det, conf, iou, multi_label=True, max_det=max_det)
            t2 = time.perf_counter()
            tal(t["pd_scores"], t["pd_bboxes"], t["anc_points"], t["gt_labels"], t["gt_bboxes"], t["mask_gt"])
            t3 = time.perf_counter()
            if rep:
                fwd.append(t1 - t0)
                nms.append(t2 - t1)
                talt.append(t3 - t2)
    step = [a + b for a, b in zip(fwd, nms)]
    med = statistics.median
    return dict(value=round(n / med(step), 3), unit="images/sec", cores=torch.get_num_threads(), kind="reference",
                value_best=round(n / min(step), 3),
                forward_s={"min": round(min(fwd), 3), "median": round(med(fwd), 3)},
                nms_s={"min": round(min(nms), 3), "median": round(med(nms), 3)},
                tal_s={"min": round(min(talt), 3), "median": round(med(talt), 3), "images": n, "anchors": 8400, "max_boxes": 40},
                sample=f"{n} images {size}x{size}: the reference's own Model (fp32, torch-CPU / MKL-DNN, deploy form) + its non_max_suppression "
                       f"(torchvision.ops.nms served by the numpy oracle) - value = images / median (forward + NMS); its TaskAlignedAssigner "
                       f"on {n} images; 1 warm-up + {passes} timed passes",
                torch=torch.__version__)
