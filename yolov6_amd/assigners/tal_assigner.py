"""TaskAlignedAssigner with the reference's constructor / forward signature
(yolov6/assigners/tal_assigner.py:6-44), running as four HIP kernels
(yolov6_amd/csrc/tal.hip) with O(B*A) scratch instead of the reference's
[B, G, topk, A] int64 one-hot temp (tal_assigner.py:155).
"""
import ctypes as C

import torch
import torch.nn as nn

from .. import _lib


class TaskAlignedAssigner(nn.Module):
    def __init__(self, topk=13, num_classes=80, alpha=1.0, beta=6.0, eps=1e-9):
        super().__init__()
        self.topk = topk
        self.num_classes = num_classes
        self.bg_idx = num_classes
        self.alpha = alpha
        self.beta = beta
        self.eps = eps

    @torch.no_grad()
    def forward(self, pd_scores, pd_bboxes, anc_points, gt_labels, gt_bboxes, mask_gt):
        """
        Args:
            pd_scores (Tensor): shape(bs, num_total_anchors, num_classes)
            pd_bboxes (Tensor): shape(bs, num_total_anchors, 4)   xyxy
            anc_points (Tensor): shape(num_total_anchors, 2)
            gt_labels (Tensor): shape(bs, n_max_boxes, 1)
            gt_bboxes (Tensor): shape(bs, n_max_boxes, 4)
            mask_gt (Tensor): shape(bs, n_max_boxes, 1)
        Returns:
            target_labels (bs, A) int64, target_bboxes (bs, A, 4), target_scores (bs, A, C), fg_mask (bs, A) bool
        """
        lib = _lib.load()
        _lib.require_gpu_tensor(pd_scores, "pd_scores")
        dev = pd_scores.device
        f32 = lambda t: t.detach().to(dev, torch.float32).contiguous()
        ps, pb, ap = f32(pd_scores), f32(pd_bboxes), f32(anc_points)
        B, A, Cn = ps.shape
        G = gt_bboxes.size(1)
        gl = f32(gt_labels).reshape(B, G)
        gb = f32(gt_bboxes)
        mg = f32(mask_gt).reshape(B, G)
        t_labels = torch.empty((B, A), dtype=torch.int64, device=dev)
        t_bboxes = torch.empty((B, A, 4), dtype=torch.float32, device=dev)
        t_scores = torch.empty((B, A, Cn), dtype=torch.float32, device=dev)
        fg = torch.empty((B, A), dtype=torch.bool, device=dev)
        ws = torch.empty(max(int(lib.y6_tal_workspace_bytes(B, A, G)), 256), dtype=torch.uint8, device=dev)
        d = _lib.TalDesc()
        d.pd_scores, d.pd_bboxes, d.anc_points = (C.c_void_p(t.data_ptr()) for t in (ps, pb, ap))
        d.gt_labels, d.gt_bboxes, d.mask_gt = (C.c_void_p(t.data_ptr()) if G > 0 else None for t in (gl, gb, mg))
        d.B, d.A, d.C, d.G, d.topk = B, A, Cn, G, int(self.topk)
        d.alpha, d.beta, d.eps = float(self.alpha), float(self.beta), float(self.eps)
        d.target_labels, d.target_bboxes = C.c_void_p(t_labels.data_ptr()), C.c_void_p(t_bboxes.data_ptr())
        d.target_scores, d.fg_mask = C.c_void_p(t_scores.data_ptr()), C.c_void_p(fg.data_ptr())
        d.workspace, d.workspace_bytes = C.c_void_p(ws.data_ptr()), ws.numel()
        _lib.check(lib.y6_tal_assign(C.byref(d), _lib.current_stream_ptr()), "tal_assign")
        # the reference returns float tensors in pd dtype for boxes/scores and float labels only on the
        # G==0 early-out; keep fp32 / int64 (what the G>0 path yields for fp32 inputs)
        if G == 0:
            t_labels = t_labels.to(pd_scores.dtype)
            fg = fg.to(pd_scores.dtype)
        return t_labels, t_bboxes.to(pd_bboxes.dtype), t_scores.to(pd_scores.dtype), fg
