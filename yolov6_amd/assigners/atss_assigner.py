"""ATSSAssigner with the reference's constructor / forward signature
(yolov6/assigners/atss_assigner.py:7-40), running as three HIP kernels (yolov6_amd/csrc/tal.hip).
Used by the reference's ComputeLoss only while `epoch < atss_warmup_epoch` (loss.py:86-94)."""
import ctypes as C

import torch
import torch.nn as nn

from .. import _lib


class ATSSAssigner(nn.Module):
    '''Adaptive Training Sample Selection Assigner'''

    def __init__(self, topk=9, num_classes=80):
        super().__init__()
        self.topk = topk
        self.num_classes = num_classes
        self.bg_idx = num_classes

    @torch.no_grad()
    def forward(self, anc_bboxes, n_level_bboxes, gt_labels, gt_bboxes, mask_gt, pd_bboxes):
        """
        Args:
            anc_bboxes (Tensor): shape(num_total_anchors, 4)
            n_level_bboxes (List): anchors per pyramid level
            gt_labels (Tensor): shape(bs, n_max_boxes, 1)
            gt_bboxes (Tensor): shape(bs, n_max_boxes, 4)
            mask_gt (Tensor): shape(bs, n_max_boxes, 1)
            pd_bboxes (Tensor or None): shape(bs, num_total_anchors, 4) - soft labels when given
        Returns:
            target_labels (bs, A) int64, target_bboxes (bs, A, 4), target_scores (bs, A, C), fg_mask (bs, A) bool
        """
        lib = _lib.load()
        _lib.require_gpu_tensor(anc_bboxes, "anc_bboxes")
        dev = anc_bboxes.device
        f32 = lambda t: t.detach().to(dev, torch.float32).contiguous()
        anc = f32(anc_bboxes)
        A = anc.shape[0]
        B, G = gt_bboxes.size(0), gt_bboxes.size(1)
        Cn = self.num_classes
        gl, gb, mg = f32(gt_labels).reshape(B, G), f32(gt_bboxes), f32(mask_gt).reshape(B, G)
        pb = f32(pd_bboxes) if pd_bboxes is not None else None
        t_labels = torch.empty((B, A), dtype=torch.int64, device=dev)
        t_bboxes = torch.empty((B, A, 4), dtype=torch.float32, device=dev)
        t_scores = torch.empty((B, A, Cn), dtype=torch.float32, device=dev)
        fg = torch.empty((B, A), dtype=torch.bool, device=dev)
        ws = torch.empty(max(int(lib.y6_atss_workspace_bytes(B, A, G)), 256), dtype=torch.uint8, device=dev)
        d = _lib.AtssDesc()
        d.anc_bboxes = C.c_void_p(anc.data_ptr())
        levels = [int(n) for n in n_level_bboxes]
        if len(levels) > _lib.MAX_LEVELS:
            raise RuntimeError(f"yolov6_amd: at most {_lib.MAX_LEVELS} pyramid levels")
        for i, n in enumerate(levels):
            d.n_level_bboxes[i] = n
        d.n_levels = len(levels)
        d.gt_labels, d.gt_bboxes, d.mask_gt = (C.c_void_p(t.data_ptr()) if G > 0 else None for t in (gl, gb, mg))
        d.pd_bboxes = C.c_void_p(pb.data_ptr()) if pb is not None else None
        d.B, d.A, d.C, d.G, d.topk = B, A, Cn, G, int(self.topk)
        d.target_labels, d.target_bboxes = C.c_void_p(t_labels.data_ptr()), C.c_void_p(t_bboxes.data_ptr())
        d.target_scores, d.fg_mask = C.c_void_p(t_scores.data_ptr()), C.c_void_p(fg.data_ptr())
        d.workspace, d.workspace_bytes = C.c_void_p(ws.data_ptr()), ws.numel()
        _lib.check(lib.y6_atss_assign(C.byref(d), _lib.current_stream_ptr()), "atss_assign")
        if G == 0:   # the reference's early-out returns a float fg mask (atss_assigner.py:48-53)
            return t_labels, t_bboxes, t_scores, fg.float()
        return t_labels, t_bboxes, t_scores, fg
