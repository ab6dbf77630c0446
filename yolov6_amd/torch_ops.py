"""The `torch.library` seam SURVEY 8b sketched (`TORCH_LIBRARY(yolov6_hip, m)`), for the STATELESS operators of the hot path.

    import yolov6_amd.torch_ops                      # registers the namespace
    dets, index, count = torch.ops.yolov6_hip.nms_batched(pred, 0.03, 0.65, None, False, True, 300)
    labels, boxes, scores, fg = torch.ops.yolov6_hip.tal_assign(pd_scores, pd_bboxes, anc_points, gt_labels, gt_bboxes, mask_gt,
                                                                 13, 1.0, 6.0, 1e-9)
    y = torch.ops.yolov6_hip.conv2d_bias_act(x_nchw, weight, bias, "relu", 1)

Each op is a dispatcher-visible custom op (schema, CUDA-only implementation, fake / meta kernel for shape inference), so
`torch.compile` traces THROUGH callers of them (they stay opaque calls into libyolov6_hip.so), `torch.ops.yolov6_hip.*` works from
TorchScript-free C++ / Python callers alike, and errors surface as `RuntimeError` - what the reference's assigner fallback catches
(yolov6/models/losses/loss.py:105).  The implementations are the SAME ctypes bindings the module mirrors use (utils/nms.py,
assigners/*.py, engine.PlanBuilder): one code path, two doors.  What is NOT here: the whole-model plans (`Model.forward`) and the
training graph - they own buffers, packed weights and streams, which is state a functional op cannot carry; they stay behind
`HipModule`.

Replaces / mirrors: `non_max_suppression` (yolov6/utils/nms.py:31-105), `TaskAlignedAssigner.forward` (assigners/tal_assigner.py:22-106),
`ATSSAssigner.forward` (assigners/atss_assigner.py:22-93), `ConvModule.forward_fuse` / the RepVGG deploy conv (layers/common.py:51-54,
:247-248)."""
from typing import List, Optional, Tuple

import torch
from torch import Tensor

NAMESPACE = "yolov6_hip"
__all__ = ["nms_batched", "tal_assign", "atss_assign", "conv2d_bias_act"]


@torch.library.custom_op(f"{NAMESPACE}::nms_batched", mutates_args=(), device_types="cuda")
def nms_batched(pred: Tensor, conf_thres: float, iou_thres: float, classes: Optional[List[int]], agnostic: bool, multi_label: bool,
                max_det: int) -> Tuple[Tensor, Tensor, Tensor]:
    """(dets [B, max_det, 6] f32 xyxy/conf/cls, index [B, max_det] i32 = anchor * nc + cls, count [B] i32): the device-side result
    of `non_max_suppression`; rows past `count[b]` are 0 / -1."""
    from .utils.nms import nms_raw
    return nms_raw(pred, conf_thres, iou_thres, classes, agnostic, multi_label, max_det)


@nms_batched.register_fake
def _(pred, conf_thres, iou_thres, classes, agnostic, multi_label, max_det):
    B = pred.shape[0]
    return (pred.new_empty((B, max_det, 6), dtype=torch.float32), pred.new_empty((B, max_det), dtype=torch.int32),
            pred.new_empty((B,), dtype=torch.int32))


def _assign_fake(ref: Tensor, B: int, A: int, C: int):
    return (ref.new_empty((B, A), dtype=torch.int64), ref.new_empty((B, A, 4), dtype=torch.float32),
            ref.new_empty((B, A, C), dtype=torch.float32), ref.new_empty((B, A), dtype=torch.bool))


@torch.library.custom_op(f"{NAMESPACE}::tal_assign", mutates_args=(), device_types="cuda")
def tal_assign(pd_scores: Tensor, pd_bboxes: Tensor, anc_points: Tensor, gt_labels: Tensor, gt_bboxes: Tensor, mask_gt: Tensor,
               topk: int, alpha: float, beta: float, eps: float) -> Tuple[Tensor, Tensor, Tensor, Tensor]:
    """(target_labels [B, A] i64, target_bboxes [B, A, 4] f32, target_scores [B, A, C] f32, fg_mask [B, A] bool)."""
    from .assigners import TaskAlignedAssigner
    G = gt_bboxes.shape[1]
    out = TaskAlignedAssigner(topk=topk, num_classes=pd_scores.shape[-1], alpha=alpha, beta=beta, eps=eps)(
        pd_scores, pd_bboxes, anc_points, gt_labels, gt_bboxes, mask_gt)
    lab, box, sc, fg = out
    if G == 0:        # (the module mirrors the reference's float early-out; the op's schema is fixed)
        lab, fg = lab.long(), fg.bool()
    return lab, box.float(), sc.float(), fg


@tal_assign.register_fake
def _(pd_scores, pd_bboxes, anc_points, gt_labels, gt_bboxes, mask_gt, topk, alpha, beta, eps):
    B, A, C = pd_scores.shape
    return _assign_fake(pd_scores, B, A, C)


@torch.library.custom_op(f"{NAMESPACE}::atss_assign", mutates_args=(), device_types="cuda")
def atss_assign(anc_bboxes: Tensor, n_level_bboxes: List[int], gt_labels: Tensor, gt_bboxes: Tensor, mask_gt: Tensor,
                pd_bboxes: Optional[Tensor], topk: int, num_classes: int) -> Tuple[Tensor, Tensor, Tensor, Tensor]:
    from .assigners import ATSSAssigner
    lab, box, sc, fg = ATSSAssigner(topk=topk, num_classes=num_classes)(anc_bboxes, n_level_bboxes, gt_labels, gt_bboxes, mask_gt, pd_bboxes)
    return lab.long(), box.float(), sc.float(), fg.bool()


@atss_assign.register_fake
def _(anc_bboxes, n_level_bboxes, gt_labels, gt_bboxes, mask_gt, pd_bboxes, topk, num_classes):
    return _assign_fake(anc_bboxes, gt_bboxes.shape[0], anc_bboxes.shape[0], num_classes)


@torch.library.custom_op(f"{NAMESPACE}::conv2d_bias_act", mutates_args=(), device_types="cuda")
def conv2d_bias_act(x: Tensor, weight: Tensor, bias: Optional[Tensor], act: str, stride: int) -> Tensor:
    """conv (k in {1, 3}, pad k//2, stride 1 / 2) + bias + activation ("relu" | "silu" | "hardswish" | "none") of an NCHW fp16
    tensor as one fused HIP launch; NCHW fp16 out.  A functional op packs the weights on every call: the module mirrors (which cache
    a plan) are the fast path - this door is for callers that want the single operator."""
    from .engine import NCHWInput, PlanBuilder
    if act not in ("relu", "silu", "hardswish", "none"):
        raise RuntimeError(f"yolov6_hip::conv2d_bias_act: unknown activation {act!r}")
    pb = PlanBuilder(x.device)
    y = pb.conv(NCHWInput(x.contiguous()), weight, bias, stride=stride, act=None if act == "none" else act)
    out = pb.to_nchw(y, x.dtype if x.dtype in (torch.float16, torch.float32) else torch.float16)
    plan = pb.finalize(out, autotune=False)
    return plan.run().clone()


@conv2d_bias_act.register_fake
def _(x, weight, bias, act, stride):
    B, _, H, W = x.shape
    k = weight.shape[-1]
    Ho, Wo = (H + 2 * (k // 2) - k) // stride + 1, (W + 2 * (k // 2) - k) // stride + 1
    return x.new_empty((B, weight.shape[0], Ho, Wo))
