#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the UNMODIFIED reference (read-only, from
/root/reference) on CPU.  Run in the build container only - the GPU box has no reference:

    python tests/golden/gen_golden.py

What is pinned
  model_<name>.npz  reference Model (yolov6/models/yolo.py) built from the reference's own
                    configs/*.py, weights from oracle/synth.py, eval forward in train form and
                    after fuse_model + switch_to_deploy (detections + neck feature maps)
  model_<name>_half.npz  the same deploy-form model after the reference's `model.half()`, run on CPU in fp16
  keys_<name>.json  the reference state_dict keys/shapes (train form and deploy form)
  nms_<case>.npz    reference yolov6/utils/nms.py::non_max_suppression.  torchvision is absent,
                    so `torchvision.ops.nms` is served by oracle/nms_oracle.py::nms - the
                    candidate / multi-label / class-offset / max_det logic around it is the
                    reference's own code (PARITY UNPINNED for the torchvision call itself).
  tal_<case>.npz    reference yolov6/assigners/tal_assigner.py::TaskAlignedAssigner on CPU
  atss_<case>.npz   reference yolov6/assigners/atss_assigner.py::ATSSAssigner on CPU (anchors from the
                    reference's generate_anchors)
  train_<case>.npz  reference Model in TRAINING mode (batch-stat BN, Detect train branch): head outputs, BN running stats
  loss_<case>.npz   reference yolov6/models/losses/loss.py::ComputeLoss forward value (loss, loss_items) on CPU
Inputs are regenerated from seeds by oracle/synth.py, so only outputs are stored.
"""
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)

from oracle import nms_oracle, synth  # noqa: E402
from yolov6_amd.configs import load_config  # noqa: E402

torch.set_num_threads(8)


def install_stubs():
    """cv2 / torchvision are not installed; the reference imports them at module top (nms.py:9-11)."""
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


MODEL_CASES = {
    # name: (reference config file, overrides, image size, batch, num_classes)
    "tiny": ("configs/yolov6s.py", dict(width_multiple=0.125, depth_multiple=0.17), 64, 2, 80),
    "n": ("configs/yolov6n.py", {}, 64, 1, 80),
    "s": ("configs/yolov6s.py", {}, 64, 2, 80),
    "s_qa_tiny": ("configs/qarepvgg/yolov6s_qa.py", dict(width_multiple=0.125, depth_multiple=0.17), 64, 2, 80),
    "l6_tiny": ("configs/yolov6l6.py", dict(width_multiple=0.125, depth_multiple=0.34), 128, 1, 80),
    "m_tiny": ("configs/yolov6m.py", dict(width_multiple=0.125, depth_multiple=0.34), 64, 1, 20),
    # MBLABlock stages (conv_silu): width 0.25 keeps the branch width a multiple of 8; depth 0.5 gives both branch shapes
    # (n_list [0, 1] and [0, 1, 2])
    "s_mbla_tiny": ("configs/mbla/yolov6s_mbla.py", dict(width_multiple=0.25, depth_multiple=0.5), 64, 1, 20),
    # the other 1280-pixel families: EfficientRep6 + RepBiFPANNeck6 (N6 / S6), CSPBepBackbone_P6 with csp_e 2/3 (M6)
    "n6": ("configs/yolov6n6.py", {}, 128, 1, 80),
    "m6_tiny": ("configs/yolov6m6.py", dict(width_multiple=0.125, depth_multiple=0.34), 128, 1, 20),
    # the uni-directional PAN necks of the v2.0 models (dotted keys override nested config entries)
    "t_pan": ("configs/experiment/yolov6t.py", {}, 64, 1, 80),
    "s_csp_pan_tiny": ("configs/experiment/yolov6s_csp_scaled.py", dict(width_multiple=0.125, depth_multiple=0.34), 64, 1, 20),
    "n6_pan": ("configs/yolov6n6.py", {"neck.type": "RepPANNeck6", "backbone.fuse_P2": False}, 128, 1, 80),
    # conv_relu models (configs/base) and the first QARepVGG block version (no config selects it; get_block does)
    "n_base": ("configs/base/yolov6n_base.py", {}, 64, 1, 80),
    "s_base_tiny": ("configs/base/yolov6s_base.py", dict(width_multiple=0.125, depth_multiple=0.34), 64, 1, 20),
    "s_qav1_tiny": ("configs/qarepvgg/yolov6s_qa.py", dict(width_multiple=0.125, depth_multiple=0.17, training_mode="qarepvgg"), 64, 2, 80),
    # (CSPRepPANNeck_P6 has no backbone in the reference that feeds it four maps: CSPBepBackbone_P6 always returns five)
}


def ref_config(path, overrides):
    cfg = load_config(os.path.join(REF, path))
    for k, v in overrides.items():
        if k == "training_mode":
            cfg.training_mode = v
            continue
        node = cfg.model
        *path, leaf = k.split(".")
        for part in path:
            node = node[part]
        node[leaf] = v
    return cfg


def gen_models():
    from yolov6.layers.common import RepVGGBlock
    from yolov6.models.yolo import Model
    from yolov6.utils.torch_utils import fuse_model

    only = os.environ.get("GOLDEN_ONLY")      # e.g. GOLDEN_ONLY=s_mbla_tiny: (re)generate one model case
    for name, (cfile, over, size, batch, nc) in MODEL_CASES.items():
        if only and name != only:
            continue
        cfg = ref_config(cfile, over)
        torch.manual_seed(0)
        model = Model(cfg, channels=3, num_classes=nc).eval()
        sd = synth.synth_state_dict(model.state_dict(), seed=0)
        model.load_state_dict(sd)
        model.detect.proj_conv.weight.data = model.detect.proj.view(1, -1, 1, 1).clone()
        x = synth.synth_images(batch, size, seed=1)
        keys_train = {k: list(v.shape) for k, v in model.state_dict().items()}
        with torch.no_grad():
            det_train, feats_train = model(x)
            fuse_model(model)
            for m in model.modules():
                if isinstance(m, RepVGGBlock):
                    m.switch_to_deploy()
            det_dep, feats_dep = model(x)
        keys_dep = {k: list(v.shape) for k, v in model.state_dict().items()}
        out = dict(det_train=det_train.numpy(), det_deploy=det_dep.numpy())
        for i, f in enumerate(feats_dep):
            out[f"feat{i}"] = f.numpy()
        np.savez_compressed(os.path.join(HERE, f"model_{name}.npz"), **out)
        with open(os.path.join(HERE, f"keys_{name}.json"), "w") as f:
            json.dump(dict(config=cfile, overrides=over, size=size, batch=batch, num_classes=nc,
                           training_mode=cfg.training_mode, train=keys_train, deploy=keys_dep), f)
        d = float((det_train - det_dep).abs().max())
        print(f"model_{name}: det {tuple(det_dep.shape)} train-vs-deploy max diff {d:.3e} "
              f"params {sum(p.numel() for p in model.parameters()) / 1e6:.3f}M keys {len(keys_train)}/{len(keys_dep)}")


def gen_models_half():
    """The reference's OWN half-precision path on the CPU: the deploy-form model after `model.half()` on `x.half()`
    (what tools/eval.py --half / core/evaler.py:86-88 run on the GPU).  Pins the rounding points of
    oracle.model_oracle.Oracle(emulate_fp16=True), which every GPU parity test compares with.
    Separate files (model_<name>_half.npz) so that the fp32 goldens stay byte-identical."""
    from yolov6.layers.common import RepVGGBlock
    from yolov6.models.yolo import Model
    from yolov6.utils.torch_utils import fuse_model

    only = os.environ.get("GOLDEN_ONLY")
    for name, (cfile, over, size, batch, nc) in MODEL_CASES.items():
        if only and name != only:
            continue
        cfg = ref_config(cfile, over)
        torch.manual_seed(0)
        model = Model(cfg, channels=3, num_classes=nc).eval()
        model.load_state_dict(synth.synth_state_dict(model.state_dict(), seed=0))
        model.detect.proj_conv.weight.data = model.detect.proj.view(1, -1, 1, 1).clone()
        x = synth.synth_images(batch, size, seed=1)
        with torch.no_grad():
            fuse_model(model)
            for m in model.modules():
                if isinstance(m, RepVGGBlock):
                    m.switch_to_deploy()
            det32, _ = model(x)
            model.half()
            det, feats = model(x.half())
        out = dict(det_half=det.float().numpy(), det_dtype=str(det.dtype))
        for i, f in enumerate(feats):
            out[f"feat{i}_half"] = f.float().numpy()
        np.savez_compressed(os.path.join(HERE, f"model_{name}_half.npz"), **out)
        d = (det.float() - det32).abs()
        print(f"model_{name}_half: det {tuple(det.shape)} {det.dtype}; half-vs-fp32 scores {float(d[..., 5:].max()):.3e} boxes {float(d[..., :4].max()):.3e}")


NMS_CASES = {
    # name: (B, A, nc, seed, frac, kwargs)
    "eval_multilabel": (3, 600, 20, 0, 0.03, dict(conf_thres=0.03, iou_thres=0.65, multi_label=True, max_det=300)),
    "infer_single": (2, 900, 80, 1, 0.01, dict(conf_thres=0.4, iou_thres=0.45, max_det=1000)),
    "agnostic_classes": (2, 500, 20, 2, 0.05, dict(conf_thres=0.25, iou_thres=0.5, agnostic=True, classes=[1, 3, 7],
                                                   multi_label=True, max_det=50)),
    "max_det_cut": (1, 2000, 10, 3, 0.2, dict(conf_thres=0.03, iou_thres=0.9, multi_label=True, max_det=20)),
    "over_max_nms": (1, 8400, 80, 4, 0.06, dict(conf_thres=0.03, iou_thres=0.65, multi_label=True, max_det=300)),
    "empty": (2, 300, 20, 5, 0.0, dict(conf_thres=0.5, iou_thres=0.45)),
}


def gen_nms():
    from yolov6.utils.nms import non_max_suppression
    for name, (B, A, nc, seed, frac, kw) in NMS_CASES.items():
        pred = synth.synth_predictions(B, A, nc, seed=seed, frac=frac)
        with torch.no_grad():
            res = non_max_suppression(pred.clone(), **kw)
        counts = np.array([r.shape[0] for r in res], np.int32)
        dets = np.concatenate([r.numpy().reshape(-1, 6) for r in res], 0).astype(np.float32)
        np.savez_compressed(os.path.join(HERE, f"nms_{name}.npz"), counts=counts, dets=dets,
                            meta=json.dumps(dict(B=B, A=A, nc=nc, seed=seed, frac=frac, kwargs=kw)))
        print(f"nms_{name}: counts {counts.tolist()}")


TAL_CASES = {
    # name: (B, feat sizes, strides, C, G, n_valid, seed, topk)
    "basic": (2, [(20, 20), (10, 10), (5, 5)], [8, 16, 32], 20, 6, None, 0, 13),
    "padded": (3, [(20, 20), (10, 10), (5, 5)], [8, 16, 32], 20, 8, [8, 3, 0], 1, 13),
    "many_gt": (2, [(16, 16), (8, 8), (4, 4)], [8, 16, 32], 10, 101, None, 2, 13),   # per-image path (:55)
    "topk26": (1, [(20, 20), (10, 10), (5, 5)], [8, 16, 32], 80, 5, None, 3, 26),
    "empty": (2, [(8, 8), (4, 4), (2, 2)], [8, 16, 32], 20, 0, None, 4, 13),
}


def gen_tal():
    from yolov6.assigners.tal_assigner import TaskAlignedAssigner
    for name, (B, fs, st, C, G, nv, seed, topk) in TAL_CASES.items():
        inp = synth.synth_tal_inputs(B, fs, st, C, G, seed=seed, n_valid=nv)
        assigner = TaskAlignedAssigner(topk=topk, num_classes=C, alpha=1.0, beta=6.0)
        with torch.no_grad():
            tl, tb, ts, fg = assigner(inp["pd_scores"], inp["pd_bboxes"], inp["anc_points"], inp["gt_labels"],
                                      inp["gt_bboxes"], inp["mask_gt"])
        ts = ts.numpy().astype(np.float32)
        nz = np.nonzero(ts)
        np.savez_compressed(os.path.join(HERE, f"tal_{name}.npz"), labels=tl.numpy().astype(np.int64),
                            bboxes=tb.numpy().astype(np.float32), fg=fg.numpy().astype(bool),
                            score_idx=np.stack(nz, 1).astype(np.int32), score_val=ts[nz],
                            meta=json.dumps(dict(B=B, feat_sizes=fs, strides=st, C=C, G=G, n_valid=nv, seed=seed,
                                                 topk=topk)))
        print(f"tal_{name}: fg {int(fg.sum())} nonzero scores {len(nz[0])}")


ATSS_CASES = {
    # name: (B, feat sizes, strides, C, G, n_valid, seed, with_pd)
    "basic": (2, [(20, 20), (10, 10), (5, 5)], [8, 16, 32], 20, 6, None, 10, True),
    "padded_nopd": (3, [(20, 20), (10, 10), (5, 5)], [8, 16, 32], 20, 8, [8, 3, 0], 11, False),
    "p6": (1, [(32, 32), (16, 16), (8, 8), (4, 4)], [8, 16, 32, 64], 80, 12, None, 12, True),   # every level >= topk anchors (the reference itself breaks below that, atss_assigner.py:104)
    "empty": (2, [(8, 8), (4, 4), (2, 2)], [8, 16, 32], 20, 0, None, 13, True),
}


def atss_anchors(fs, st):
    """Train-form anchors exactly as the reference builds them (anchor_generator.py:35-63)."""
    from yolov6.assigners.anchor_generator import generate_anchors
    feats = [torch.zeros(1, 1, h, w) for h, w in fs]
    anchors, _, n_list, _ = generate_anchors(feats, st, 5.0, 0.5, device="cpu", is_eval=False)
    return anchors, n_list


def gen_atss():
    from yolov6.assigners.atss_assigner import ATSSAssigner
    for name, (B, fs, st, C, G, nv, seed, with_pd) in ATSS_CASES.items():
        inp = synth.synth_tal_inputs(B, fs, st, C, G, seed=seed, n_valid=nv)
        anchors, n_list = atss_anchors(fs, st)
        assigner = ATSSAssigner(9, num_classes=C)
        with torch.no_grad():
            tl, tb, ts, fg = assigner(anchors, n_list, inp["gt_labels"], inp["gt_bboxes"], inp["mask_gt"],
                                      inp["pd_bboxes"] if with_pd else None)
        ts = ts.numpy().astype(np.float32)
        nz = np.nonzero(ts)
        np.savez_compressed(os.path.join(HERE, f"atss_{name}.npz"), labels=tl.numpy().astype(np.int64),
                            bboxes=tb.numpy().astype(np.float32), fg=fg.numpy().astype(bool),
                            score_idx=np.stack(nz, 1).astype(np.int32), score_val=ts[nz],
                            anchors=anchors.numpy().astype(np.float32), n_list=np.array(n_list, np.int32),
                            meta=json.dumps(dict(B=B, feat_sizes=fs, strides=st, C=C, G=G, n_valid=nv, seed=seed,
                                                 with_pd=with_pd)))
        print(f"atss_{name}: fg {int(fg.sum())} nonzero scores {len(nz[0])}")


TRAIN_CASES = ["tiny", "s_qa_tiny", "m_tiny"]
TRAIN_GRAD_PROBES = ["backbone.stem.rbr_dense.conv.weight", "backbone.ERBlock_3.0.rbr_dense.bn.weight",
                     "detect.cls_preds.1.bias", "neck.reduce_layer0.block.conv.weight"]
TRAIN_BN_PROBES = ["backbone.stem.rbr_dense.bn", "backbone.ERBlock_3.0.rbr_dense.bn", "detect.stems.0.block.bn", "neck.reduce_layer0.block.bn"]


def gen_train_forward():
    """reference Model in TRAINING mode on CPU (batch-statistics BN, Detect training branch effidehead.py:72-92):
    head outputs, one neck feature map and the running statistics a few BatchNorms hold after the forward."""
    from yolov6.models.yolo import Model
    for name in TRAIN_CASES:
        cfile, over, size, batch, nc = MODEL_CASES[name]
        cfg = ref_config(cfile, over)
        torch.manual_seed(0)
        model = Model(cfg, channels=3, num_classes=nc)
        sd = synth.synth_state_dict(model.state_dict(), seed=0)
        model.load_state_dict(sd)
        model.train()
        x = synth.synth_images(max(batch, 2), size, seed=21)
        (xs, cls_scores, reg_distri), featmaps = model(x)
        # a scalar of the head outputs, back-propagated by the reference's autograd: gradient goldens for three
        # parameters at different depths (first conv, a mid-backbone BN weight, a head pred bias)
        scalar = (cls_scores * cls_scores).sum() + reg_distri.square().mean()
        model.zero_grad()
        scalar.backward()
        params = dict(model.named_parameters())
        grads = {q: params[q].grad.detach().numpy().copy() for q in TRAIN_GRAD_PROBES if q in params and params[q].grad is not None}
        after = model.state_dict()
        out = dict(cls_scores=cls_scores.detach().numpy(), reg_distri=reg_distri.detach().numpy(), stem0=xs[0].detach().numpy(),
                   feat0=featmaps[0].detach().numpy(), scalar=np.float64(float(scalar)))
        for q, gq in grads.items():
            out["grad:" + q] = gq
        for q in TRAIN_BN_PROBES:
            if q + ".running_mean" in after:
                out[q + ".running_mean"] = after[q + ".running_mean"].numpy()
                out[q + ".running_var"] = after[q + ".running_var"].numpy()
        np.savez_compressed(os.path.join(HERE, f"train_{name}.npz"), **out)
        print(f"train_{name}: cls {tuple(cls_scores.shape)} reg {tuple(reg_distri.shape)} "
              f"probes {[q for q in TRAIN_BN_PROBES if q + '.running_mean' in after]}")


LOSS_CASES = {
    # name: (B, feat sizes, strides, C, reg_max, use_dfl, iou_type, epoch (vs warmup 4), seed)
    "tal_giou_dfl": (3, [(16, 16), (8, 8), (4, 4)], [8, 16, 32], 20, 16, True, "giou", 10, 0),
    "tal_siou_nodfl": (2, [(16, 16), (8, 8), (4, 4)], [8, 16, 32], 80, 16, False, "siou", 10, 1),   # yolov6n/s setting
    "atss_warmup": (3, [(16, 16), (8, 8), (4, 4)], [8, 16, 32], 20, 16, True, "giou", 0, 2),
    "ciou": (2, [(16, 16), (8, 8), (4, 4)], [8, 16, 32], 10, 16, True, "ciou", 10, 3),
    "no_targets": (2, [(8, 8), (4, 4), (2, 2)], [8, 16, 32], 20, 16, True, "giou", 10, 4),
}


def gen_loss():
    """reference ComputeLoss (models/losses/loss.py) on CPU; `.cuda()` of its two param-less modules is patched
    to a no-op (no GPU here) - nothing else is touched."""
    import torch.nn as nn
    nn.Module.cuda = lambda self, device=None: self
    from yolov6.models.losses.loss import ComputeLoss
    for name, (B, fs, st, C, reg_max, use_dfl, iou_type, epoch, seed) in LOSS_CASES.items():
        inp = synth.synth_loss_inputs(B, fs, st, C, reg_max, use_dfl, seed=seed)
        if name == "no_targets":
            inp["targets"] = inp["targets"][:0]
        crit = ComputeLoss(fpn_strides=st, num_classes=C, ori_img_size=inp["img"], warmup_epoch=4, use_dfl=use_dfl,
                           reg_max=reg_max, iou_type=iou_type)
        feats = [torch.zeros(B, 1, h, w) for h, w in fs]
        with torch.no_grad():
            loss, items = crit((feats, inp["pred_scores"].clone(), inp["pred_distri"].clone()), inp["targets"].clone(),
                               epoch, 1, inp["img"], inp["img"])
        np.savez_compressed(os.path.join(HERE, f"loss_{name}.npz"), loss=np.float64(float(loss)),
                            items=items.numpy().astype(np.float64),
                            meta=json.dumps(dict(B=B, feat_sizes=fs, strides=st, C=C, reg_max=reg_max, use_dfl=use_dfl,
                                                 iou_type=iou_type, epoch=epoch, seed=seed)))
        print(f"loss_{name}: loss {float(loss):.6f} items {items.numpy().tolist()}")
        # the same call with autograd: d loss / d pred_scores, d loss / d pred_distri as the reference back-propagates them
        ps = inp["pred_scores"].clone().requires_grad_(True)
        pd = inp["pred_distri"].clone().requires_grad_(True)
        loss2, _ = crit((feats, ps, pd), inp["targets"].clone(), epoch, 1, inp["img"], inp["img"])
        loss2.backward()
        np.savez_compressed(os.path.join(HERE, f"lossgrad_{name}.npz"), loss=np.float64(float(loss2)),
                            dscores=ps.grad.numpy(), ddistri=pd.grad.numpy())


LOSSDISTILL_CASES = {
    # name: (B, feat sizes, C, iou_type, epoch, max_epoch, warmup_epoch, temperature, distill_feat, seed)
    "tal_giou": (3, [(16, 16), (8, 8), (4, 4)], 20, "giou", 10, 100, 0, 20.0, False, 0),
    "atss_warmup_feat": (2, [(16, 16), (8, 8), (4, 4)], 20, "giou", 1, 50, 4, 20.0, True, 1),
    "siou_feat_late": (2, [(16, 16), (8, 8), (4, 4)], 80, "siou", 90, 100, 0, 10.0, True, 2),
    "no_targets": (2, [(8, 8), (4, 4), (2, 2)], 20, "giou", 10, 100, 0, 20.0, True, 3),
}


def gen_loss_distill():
    """reference self-distillation ComputeLoss (models/losses/loss_distill.py) on CPU (`.cuda()` of its two param-less
    modules patched to a no-op): loss value, loss items, and the gradients it back-propagates to the STUDENT's class scores,
    DFL logits and the three feature maps.  Teacher outputs / feature maps are a second synthetic draw."""
    import torch.nn as nn
    nn.Module.cuda = lambda self, device=None: self
    from yolov6.models.losses.loss_distill import ComputeLoss
    st, reg_max = [8, 16, 32], 16
    for name, (B, fs, C, iou_type, epoch, max_epoch, warmup, temp, dfeat, seed) in LOSSDISTILL_CASES.items():
        inp = synth.synth_loss_inputs(B, fs, st, C, reg_max, True, seed=seed)
        tea = synth.synth_loss_inputs(B, fs, st, C, reg_max, True, seed=seed + 50)
        if name == "no_targets":
            inp["targets"] = inp["targets"][:0]
        g = torch.Generator().manual_seed(1000 + seed)
        chans = [16, 24, 32]
        s_feats = [torch.randn((B, c, h, w), generator=g) for c, (h, w) in zip(chans, fs)]
        t_feats = [torch.randn((B, c, h, w), generator=g) for c, (h, w) in zip(chans, fs)]
        crit = ComputeLoss(fpn_strides=st, num_classes=C, ori_img_size=inp["img"], warmup_epoch=warmup, use_dfl=True,
                           reg_max=reg_max, iou_type=iou_type, distill_feat=dfeat)
        feats = [torch.zeros(B, 1, h, w) for h, w in fs]
        ps = inp["pred_scores"].clone().requires_grad_(True)
        pd = inp["pred_distri"].clone().requires_grad_(True)
        sf = [f.clone().requires_grad_(True) for f in s_feats]
        loss, items = crit((feats, ps, pd), (feats, tea["pred_scores"].clone(), tea["pred_distri"].clone()), sf, t_feats,
                           inp["targets"].clone(), epoch, max_epoch, temp, 1, inp["img"], inp["img"])
        loss.backward()
        out = dict(loss=np.float64(float(loss)), items=items.numpy().astype(np.float64), dscores=ps.grad.numpy(),
                   ddistri=pd.grad.numpy(),
                   meta=json.dumps(dict(B=B, feat_sizes=fs, strides=st, C=C, reg_max=reg_max, iou_type=iou_type, epoch=epoch,
                                        max_epoch=max_epoch, warmup_epoch=warmup, temperature=temp, distill_feat=dfeat,
                                        seed=seed, feat_channels=chans)))
        for i, f in enumerate(sf):
            out[f"dfeat{i}"] = (f.grad if f.grad is not None else torch.zeros_like(f)).numpy()
        np.savez_compressed(os.path.join(HERE, f"lossdistill_{name}.npz"), **out)
        print(f"lossdistill_{name}: loss {float(loss):.6f} items {items.numpy().tolist()}")


def gen_fuseab_eval():
    """Model(fuse_ab=True) in EVAL mode (effidehead_fuseab.py: the anchor-free branch alone): train-form weights, eval output."""
    from yolov6.models.yolo import Model
    cfile, over, size, batch, nc = MODEL_CASES["tiny"]
    cfg = ref_config(cfile, over)
    torch.manual_seed(0)
    model = Model(cfg, channels=3, num_classes=nc, fuse_ab=True)
    model.load_state_dict(synth.synth_state_dict(model.state_dict(), seed=0))
    model.eval()
    x = synth.synth_images(batch, size, seed=1)
    with torch.no_grad():
        det, feats = model(x)
    np.savez_compressed(os.path.join(HERE, "model_tiny_fuseab_eval.npz"), det_train=det.numpy())
    print(f"model_tiny_fuseab_eval: det {tuple(det.shape)}")


def gen_loss_distill_ns():
    """reference ComputeLoss of models/losses/loss_distill_ns.py (the N / S self-distillation loss: a fourth student output,
    plain (l, t, r, b) distances): value, items, gradients to the student's scores, DFL logits and distances."""
    import torch.nn as nn
    nn.Module.cuda = lambda self, device=None: self
    from yolov6.models.losses.loss_distill_ns import ComputeLoss
    st, reg_max = [8, 16, 32], 16
    for name in ("tal_giou", "atss_warmup_feat", "siou_feat_late"):
        B, fs, C, iou_type, epoch, max_epoch, warmup, temp, dfeat, seed = LOSSDISTILL_CASES[name]
        inp = synth.synth_loss_inputs(B, fs, st, C, reg_max, True, seed=seed)
        tea = synth.synth_loss_inputs(B, fs, st, C, reg_max, True, seed=seed + 50)
        g = torch.Generator().manual_seed(2000 + seed)
        A = inp["pred_scores"].shape[1]
        lrtb = (torch.rand((B, A, 4), generator=g) * 3.0 + 0.2)
        chans = [16, 24, 32]
        s_feats = [torch.randn((B, c, h, w), generator=g) for c, (h, w) in zip(chans, fs)]
        t_feats = [torch.randn((B, c, h, w), generator=g) for c, (h, w) in zip(chans, fs)]
        crit = ComputeLoss(fpn_strides=st, num_classes=C, ori_img_size=inp["img"], warmup_epoch=warmup, use_dfl=True,
                           reg_max=reg_max, iou_type=iou_type, distill_feat=dfeat)
        feats = [torch.zeros(B, 1, h, w) for h, w in fs]
        ps = inp["pred_scores"].clone().requires_grad_(True)
        pd = inp["pred_distri"].clone().requires_grad_(True)
        pl = lrtb.clone().requires_grad_(True)
        loss, items = crit((feats, ps, pd, pl), (feats, tea["pred_scores"].clone(), tea["pred_distri"].clone()), s_feats, t_feats,
                           inp["targets"].clone(), epoch, max_epoch, temp, 1, inp["img"], inp["img"])
        loss.backward()
        np.savez_compressed(os.path.join(HERE, f"lossdistillns_{name}.npz"), loss=np.float64(float(loss)),
                            items=items.numpy().astype(np.float64), dscores=ps.grad.numpy(), ddistri=pd.grad.numpy(),
                            dlrtb=pl.grad.numpy(), lrtb=lrtb.numpy(),
                            meta=json.dumps(dict(B=B, feat_sizes=fs, strides=st, C=C, reg_max=reg_max, iou_type=iou_type,
                                                 epoch=epoch, max_epoch=max_epoch, warmup_epoch=warmup, temperature=temp,
                                                 distill_feat=dfeat, seed=seed, feat_channels=chans)))
        print(f"lossdistillns_{name}: loss {float(loss):.6f} items {items.numpy().tolist()}")


def gen_distill_ns():
    """Model(..., distill_ns=True) (heads/effidehead_distill_ns.py): state_dict keys and the EVAL output of the tiny S graph."""
    from yolov6.models.yolo import Model
    cfile, over, size, batch, nc = MODEL_CASES["tiny"]
    cfg = ref_config(cfile, over)
    torch.manual_seed(0)
    model = Model(cfg, channels=3, num_classes=nc, distill_ns=True).eval()
    sd = synth.synth_state_dict(model.state_dict(), seed=0)
    model.load_state_dict(sd)
    x = synth.synth_images(batch, size, seed=1)
    with torch.no_grad():
        det, feats = model(x)
    with open(os.path.join(HERE, "keys_tiny_distill_ns.json"), "w") as f:
        json.dump(dict(config=cfile, overrides=over, size=size, batch=batch, num_classes=nc, training_mode=cfg.training_mode,
                       train={k: list(v.shape) for k, v in model.state_dict().items()}), f)
    np.savez_compressed(os.path.join(HERE, "model_tiny_distill_ns.npz"), det_train=det.numpy())
    print(f"model_tiny_distill_ns: det {tuple(det.shape)} keys {len(model.state_dict())}")


DISTILL_NS_GRAD_PROBES = ["backbone.stem.rbr_dense.conv.weight", "detect.reg_convs.1.block.conv.weight", "detect.reg_preds.0.weight",
                          "detect.reg_preds.2.bias", "detect.reg_preds_dist.1.weight", "detect.reg_preds_dist.0.bias", "detect.cls_preds.2.bias"]


def gen_distill_ns_train():
    """Model(..., distill_ns=True) in TRAINING mode (heads/effidehead_distill_ns.py:80-103): the three head outputs - class scores,
    DFL logits, plain (l, t, r, b) distances - and reference parameter gradients of a scalar of all three."""
    from yolov6.models.yolo import Model
    cfile, over, size, batch, nc = MODEL_CASES["tiny"]
    cfg = ref_config(cfile, over)
    torch.manual_seed(0)
    model = Model(cfg, channels=3, num_classes=nc, distill_ns=True)
    sd = synth.synth_state_dict(model.state_dict(), seed=0)
    model.load_state_dict(sd)
    model.train()
    x = synth.synth_images(max(batch, 2), size, seed=21)
    (xs, cls_scores, reg_distri, reg_lrtb), featmaps = model(x)
    scalar = (cls_scores * cls_scores).sum() + reg_distri.square().mean() + reg_lrtb.square().mean()
    model.zero_grad()
    scalar.backward()
    params = dict(model.named_parameters())
    out = dict(cls_scores=cls_scores.detach().numpy(), reg_distri=reg_distri.detach().numpy(), reg_lrtb=reg_lrtb.detach().numpy(),
               scalar=np.float64(float(scalar)))
    for q in DISTILL_NS_GRAD_PROBES:
        out["grad:" + q] = params[q].grad.detach().numpy().copy()
    np.savez_compressed(os.path.join(HERE, "distill_ns_train_tiny.npz"), **out)
    print(f"distill_ns_train_tiny: cls {tuple(cls_scores.shape)} distri {tuple(reg_distri.shape)} lrtb {tuple(reg_lrtb.shape)} scalar {float(scalar):.4f}")


FUSEAB_GRAD_PROBES = ["backbone.stem.rbr_dense.conv.weight", "neck.Rep_p3.conv1.rbr_1x1.bn.weight", "detect.cls_convs.1.block.conv.weight",
                      "detect.cls_preds_ab.0.weight", "detect.reg_preds_ab.1.bias", "detect.reg_preds_ab.2.weight", "detect.cls_preds.2.bias"]
LOSSAB_CASES = {
    # name: (B, feat sizes, strides, C, iou_type, seed)
    "giou": (3, [(16, 16), (8, 8), (4, 4)], [8, 16, 32], 20, "giou", 0),
    "siou": (2, [(16, 16), (8, 8), (4, 4)], [8, 16, 32], 80, "siou", 1),
}


def gen_fuseab():
    """fuse_ab: reference Model(fuse_ab=True) in TRAINING mode - the five head outputs of
    models/heads/effidehead_fuseab.py:139 - with reference gradients, the state_dict keys, and the anchor-based
    ComputeLoss (models/losses/loss_fuseab.py) value + gradients."""
    import torch.nn as nn
    from yolov6.models.yolo import Model
    cfile, over, size, batch, nc = MODEL_CASES["tiny"]
    cfg = ref_config(cfile, over)
    torch.manual_seed(0)
    model = Model(cfg, channels=3, num_classes=nc, fuse_ab=True)
    sd = synth.synth_state_dict(model.state_dict(), seed=0)
    model.load_state_dict(sd)
    with open(os.path.join(HERE, "keys_tiny_fuseab.json"), "w") as f:
        json.dump(dict(config=cfile, overrides=over, size=size, batch=batch, num_classes=nc, training_mode=cfg.training_mode,
                       train={k: list(v.shape) for k, v in model.state_dict().items()},
                       anchors_init=cfg.model.head.anchors_init), f)
    model.train()
    x = synth.synth_images(max(batch, 2), size, seed=21)
    (xs, cls_ab, reg_ab, cls_af, reg_af), featmaps = model(x)
    scalar = (cls_ab * cls_ab).sum() + reg_ab.square().mean() + (cls_af * cls_af).sum() + reg_af.square().mean()
    model.zero_grad()
    scalar.backward()
    params = dict(model.named_parameters())
    out = dict(cls_ab=cls_ab.detach().numpy(), reg_ab=reg_ab.detach().numpy(), cls_af=cls_af.detach().numpy(),
               reg_af=reg_af.detach().numpy(), scalar=np.float64(float(scalar)))
    for q in FUSEAB_GRAD_PROBES:
        out["grad:" + q] = params[q].grad.detach().numpy().copy()
    np.savez_compressed(os.path.join(HERE, "fuseab_train_tiny.npz"), **out)
    print(f"train_tiny_fuseab: cls_ab {tuple(cls_ab.shape)} reg_ab {tuple(reg_ab.shape)} scalar {float(scalar):.4f}")
    nn.Module.cuda = lambda self, device=None: self
    from yolov6.models.losses.loss_fuseab import ComputeLoss
    for name, (B, fs, st, C, iou_type, seed) in LOSSAB_CASES.items():
        inp = synth.synth_loss_inputs_ab(B, fs, st, C, seed=seed)
        crit = ComputeLoss(fpn_strides=st, num_classes=C, ori_img_size=inp["img"], warmup_epoch=0, use_dfl=False, reg_max=0,
                           iou_type=iou_type)
        feats = [torch.zeros(B, 1, h, w) for h, w in fs]
        ps = inp["pred_scores"].clone().requires_grad_(True)
        pd = inp["pred_distri"].clone().requires_grad_(True)
        # the reference adds the anchor points to pred_distri IN PLACE (loss_fuseab.py:75): feed a non-leaf copy
        loss, items = crit((feats, ps * 1.0, pd * 1.0), inp["targets"].clone(), 10, 1, inp["img"], inp["img"])
        loss.backward()
        np.savez_compressed(os.path.join(HERE, f"lossab_{name}.npz"), loss=np.float64(float(loss)), items=items.numpy().astype(np.float64),
                            dscores=ps.grad.numpy(), ddistri=pd.grad.numpy(),
                            meta=json.dumps(dict(B=B, feat_sizes=fs, strides=st, C=C, iou_type=iou_type, seed=seed)))
        print(f"lossab_{name}: loss {float(loss):.6f} items {items.numpy().tolist()}")


if __name__ == "__main__":
    install_stubs()
    which = sys.argv[1:] or ["models", "nms", "tal", "atss", "loss", "train", "fuseab", "lossdistill", "distill_ns"]
    if "fuseab" in which:
        gen_fuseab()
    if "lossdistill" in which:
        gen_loss_distill()
    if "distill_ns" in which:
        gen_distill_ns()
        gen_loss_distill_ns()
    if "distill_ns_train" in which or "distill_ns" in which:
        gen_distill_ns_train()
    if "fuseab_eval" in which or "fuseab" in which:
        gen_fuseab_eval()
    if "models" in which:
        gen_models()
    if "models_half" in which or "models" in which:
        gen_models_half()
    if "nms" in which:
        gen_nms()
    if "tal" in which:
        gen_tal()
    if "atss" in which:
        gen_atss()
    if "loss" in which:
        gen_loss()
    if "train" in which:
        gen_train_forward()
