"""ComputeLoss - mirror of reference yolov6/models/losses/loss.py:14-198 (constructor arguments, call signature,
return value), computing the FORWARD VALUE of the training loss on the GPU:

    anchors            generate_anchors (host, cached per feature-map size like loss.py:63-69)
    targets            preprocess (host, as the reference: loss.py:184-192)
    pred_bboxes        y6_bbox_decode                       (loss.py:194-198)
    label assignment   ATSSAssigner for epoch < warmup_epoch, TaskAlignedAssigner after (loss.py:83-103) - HIP
    loss terms         y6_loss_forward: VarifocalLoss, IoU loss, DFL loss, normalisation, weights (loss.py:154-182)

The result is the reference's pair (loss, loss_items[iou, dfl, cls]).  When the predictions require grad, `loss` carries an
autograd node whose backward runs y6_loss_backward (dual-number IoU gradient, DFL / VarifocalLoss closed forms) with the
incoming gradient - the GradScaler's loss scale - applied inside the kernels, and writes d loss / d pred_scores,
d loss / d pred_distri straight into the training graph's head-gradient buffers (yolov6_amd/train_engine.py), so
`scaler.scale(loss).backward()` (core/engine.py:173) drives the native backward plan.  There is no CPU path and no OOM
fallback to one (the reference's `except RuntimeError` branch, loss.py:105-152, exists because its assigner needs
O(B*G*A) temporaries; the HIP assigners do not).
"""
import ctypes as C

import torch

from ... import _lib
from ...assigners.anchor_generator import generate_anchors
from ...assigners.atss_assigner import ATSSAssigner
from ...assigners.tal_assigner import TaskAlignedAssigner


class _LossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred_scores, pred_distri, holder):
        ctx.holder = holder
        ctx.dtypes = (pred_scores.dtype, pred_distri.dtype)
        return holder["loss"].clone()

    @staticmethod
    def backward(ctx, gout):
        h = ctx.holder
        gs = gout.detach().to(torch.float32).contiguous()
        d = h["desc"]
        d.grad_scale = C.c_void_p(gs.data_ptr())
        _lib.check(_lib.load().y6_loss_backward(C.byref(d), _lib.current_stream_ptr()), "loss_backward")
        ds, dd = h["dscores"], h["ddistri"]
        return ds.to(ctx.dtypes[0]), dd.to(ctx.dtypes[1]), None


class ComputeLoss:
    """Loss computation func."""
    anchor_mode = "af"     # generate_anchors mode
    box_mode = 0           # 0: (l,t,r,b) distances / DFL logits;  1: (dx, dy, w, h) around the anchor point (loss_fuseab.py)
    formal_topk = 13

    def __init__(self, fpn_strides=[8, 16, 32], grid_cell_size=5.0, grid_cell_offset=0.5, num_classes=80,
                 ori_img_size=640, warmup_epoch=4, use_dfl=True, reg_max=16, iou_type='giou',
                 loss_weight={'class': 1.0, 'iou': 2.5, 'dfl': 0.5}):
        self.fpn_strides = fpn_strides
        self.cached_feat_sizes = [torch.Size([0, 0]) for _ in fpn_strides]
        self.cached_anchors = None
        self.grid_cell_size = grid_cell_size
        self.grid_cell_offset = grid_cell_offset
        self.num_classes = num_classes
        self.ori_img_size = ori_img_size
        self.warmup_epoch = warmup_epoch
        self.warmup_assigner = ATSSAssigner(9, num_classes=self.num_classes)
        self.formal_assigner = TaskAlignedAssigner(topk=self.formal_topk, num_classes=self.num_classes, alpha=1.0, beta=6.0)
        self.use_dfl = use_dfl
        self.reg_max = reg_max
        self.iou_type = iou_type.lower()
        if self.iou_type not in _lib.IOU_TYPES:
            raise ValueError(f"iou_type must be one of {sorted(_lib.IOU_TYPES)}, got {iou_type!r}")
        self.loss_weight = loss_weight
        self._ws = {}

    # ------------------------------------------------------------------ host pieces, as in the reference
    def preprocess(self, targets, batch_size, scale_tensor):
        """loss.py:184-192: targets [N,6] = (image, class, cx, cy, w, h in 0..1) -> [B,G,5] fp32 (label, x1,y1,x2,y2 px), rows of
        an image in their original order, padding rows (-1, 0,0,0,0).  The reference packs through Python lists on the host
        (`targets.cpu().numpy().tolist()`); here the packing stays on the device (stable sort by image + scatter) and the only
        host round trip is the scalar G = most boxes in one image, which sizes the result."""
        dev = scale_tensor.device
        rows = targets.detach().to(dev, torch.float32).reshape(-1, 6)
        n = rows.shape[0]
        if n == 0:
            return torch.zeros((batch_size, 0, 5), dtype=torch.float32, device=dev)
        img = rows[:, 0].to(torch.int64)
        lo, hi = torch.aminmax(img)
        counts = torch.bincount(img.clamp(0, batch_size - 1), minlength=batch_size)
        lo, hi, max_len = (int(v) for v in torch.stack([lo, hi, counts.max()]).tolist())      # one host sync
        if lo < 0 or hi >= batch_size:
            raise IndexError(f"targets name image {hi if hi >= batch_size else lo} but the batch holds {batch_size} images")
        out = torch.zeros((batch_size, max_len, 5), dtype=torch.float32, device=dev)
        out[:, :, 0] = -1
        order = torch.argsort(img, stable=True)          # per image, rows keep their order (the reference appends in order)
        starts = torch.cumsum(counts, 0) - counts
        simg = img[order]
        pos = torch.arange(n, device=dev) - starts[simg]
        out[simg, pos] = rows[order, 1:]
        box = out[:, :, 1:5] * scale_tensor
        x1 = box[..., 0] - box[..., 2] * 0.5          # xywh2xyxy exactly as general.py:52-58 (x2 = x1 + w)
        y1 = box[..., 1] - box[..., 3] * 0.5
        out[:, :, 1:5] = torch.stack([x1, y1, x1 + box[..., 2], y1 + box[..., 3]], -1)
        return out

    def bbox_decode(self, anchor_points, pred_dist, use_dfl=None):
        lib = _lib.load()
        use_dfl = self.use_dfl if use_dfl is None else use_dfl
        pred_dist = pred_dist.detach().float().contiguous()
        _lib.require_gpu_tensor(pred_dist, "pred_dist")
        B, A = pred_dist.shape[:2]
        pts = anchor_points.float().contiguous()
        out = torch.empty((B, A, 4), dtype=torch.float32, device=pred_dist.device)
        _lib.check(lib.y6_bbox_decode(C.c_void_p(pred_dist.data_ptr()), C.c_void_p(pts.data_ptr()), B, A,
                                      int(use_dfl), int(self.reg_max if use_dfl else 0), C.c_void_p(out.data_ptr()),
                                      _lib.current_stream_ptr()), "bbox_decode")
        return out

    # ------------------------------------------------------------------ the call
    def __call__(self, outputs, targets, epoch_num, step_num, batch_height, batch_width):
        feats, pred_scores, pred_distri = outputs
        t = self._forward_terms(feats, pred_scores, pred_distri, targets, epoch_num, batch_height, batch_width)
        d, res = t["desc"], t["out"].float()
        ps_in, pd_in = outputs[1], outputs[2]
        if torch.is_grad_enabled() and (ps_in.requires_grad or pd_in.requires_grad):
            graph = getattr(ps_in, "_y6_graph", None)
            dscores = graph.grad_buffer_of(ps_in) if graph is not None else None     # the native backward plan reads these directly
            ddistri = graph.grad_buffer_of(pd_in) if graph is not None else None
            if dscores is None or ddistri is None:
                dscores, ddistri = torch.empty_like(t["pred_scores"]), torch.empty_like(t["pred_distri"])
            g = _lib.LossGradDesc()
            g.fwd = d
            g.dpred_scores, g.dpred_distri = C.c_void_p(dscores.data_ptr()), C.c_void_p(ddistri.data_ptr())
            holder = dict(desc=g, dscores=dscores, ddistri=ddistri, loss=res[0], keep=t["keep"])
            return _LossFn.apply(ps_in, pd_in, holder), res[1:4].detach()
        return res[0], res[1:4].detach()

    def _forward_terms(self, feats, pred_scores, pred_distri, targets, epoch_num, batch_height, batch_width, norm_mode=0,
                       warmup=True, use_dfl=None, weights=None, assigned=None):
        """Everything of __call__ up to and including y6_loss_forward: anchors, target packing, box decode, label assignment,
        the loss terms.  -> dict(desc, out [6] f64 device, the tensors the descriptor points at).  `norm_mode` / `warmup` /
        `use_dfl` / `weights` / `assigned` (the (labels, boxes, scores, fg) of an earlier call: no second assignment) let the
        self-distillation losses (loss_distill.py, loss_distill_ns.py) re-use it for their detection terms and for the extra IoU
        term of the plain-distance branch."""
        lib = _lib.load()
        use_dfl = self.use_dfl if use_dfl is None else use_dfl
        weights = self.loss_weight if weights is None else weights
        _lib.require_gpu_tensor(pred_scores, "pred_scores")
        dev = pred_scores.device
        # only the spatial sizes of `feats` matter (loss.py:63-69); lazily converted training-graph features are not touched
        sizes = feats.feat_sizes() if hasattr(feats, "feat_sizes") else [feat.shape[2:] for feat in feats]
        if len(sizes) == len(self.cached_feat_sizes) and all(tuple(a) == tuple(b) for a, b in zip(sizes, self.cached_feat_sizes)) \
                and self.cached_anchors is not None:
            anchors, anchor_points, n_anchors_list, stride_tensor = self.cached_anchors
        else:
            self.cached_feat_sizes = [torch.Size(sz) for sz in sizes]
            shape_feats = [torch.zeros(1, device=dev).expand(1, 1, int(sz[0]), int(sz[1])) for sz in sizes]
            anchors, anchor_points, n_anchors_list, stride_tensor = generate_anchors(
                shape_feats, self.fpn_strides, self.grid_cell_size, self.grid_cell_offset, device=dev, mode=self.anchor_mode)
            anchors, anchor_points, stride_tensor = (t.float().to(dev) for t in (anchors, anchor_points, stride_tensor))
            self.cached_anchors = anchors, anchor_points, n_anchors_list, stride_tensor
        assert pred_scores.type() == pred_distri.type()
        pred_scores = pred_scores.detach().float().contiguous()
        pred_distri = pred_distri.detach().float().contiguous()
        B, A, Cn = pred_scores.shape
        gt_bboxes_scale = torch.tensor([batch_width, batch_height, batch_width, batch_height], dtype=torch.float32, device=dev)
        targets = self.preprocess(targets, B, gt_bboxes_scale)
        gt_labels = targets[:, :, :1]
        gt_bboxes = targets[:, :, 1:].contiguous()
        mask_gt = (gt_bboxes.sum(-1, keepdim=True) > 0).float()

        anchor_points_s = (anchor_points / stride_tensor).contiguous()
        if self.box_mode == 1:      # loss_fuseab.py:75-76: pred_distri[..., :2] += anchor_points_s; xywh2xyxy (x2 = x1 + w)
            cxy = pred_distri[..., :2] + anchor_points_s
            x1y1 = cxy - pred_distri[..., 2:] * 0.5
            pred_bboxes = torch.cat([x1y1, x1y1 + pred_distri[..., 2:]], -1).contiguous()
        else:
            pred_bboxes = self.bbox_decode(anchor_points_s, pred_distri, use_dfl)
        if assigned is not None:
            target_labels, target_bboxes, target_scores, fg_mask = assigned
        elif warmup and epoch_num < self.warmup_epoch:
            target_labels, target_bboxes, target_scores, fg_mask = self.warmup_assigner(
                anchors, n_anchors_list, gt_labels, gt_bboxes, mask_gt, pred_bboxes * stride_tensor)
        else:
            target_labels, target_bboxes, target_scores, fg_mask = self.formal_assigner(
                pred_scores, pred_bboxes * stride_tensor, anchor_points, gt_labels, gt_bboxes, mask_gt)

        target_labels = target_labels.to(torch.int64).contiguous()
        target_bboxes = target_bboxes.float().contiguous()
        target_scores = target_scores.float().contiguous()
        fg_u8 = fg_mask.to(torch.uint8).contiguous()
        stride_flat = stride_tensor.reshape(-1).float().contiguous()
        out = torch.empty((6,), dtype=torch.float64, device=dev)
        ws_key = (dev, torch.cuda.current_stream(dev).cuda_stream)   # per stream: concurrent calls must not share it
        ws = self._ws.get(ws_key)
        if ws is None:
            ws = self._ws[ws_key] = torch.empty(int(lib.y6_loss_workspace_bytes()), dtype=torch.uint8, device=dev)
        d = _lib.LossDesc()
        for name, t in (("pred_scores", pred_scores), ("pred_distri", pred_distri), ("pred_bboxes", pred_bboxes),
                        ("anchor_points_s", anchor_points_s), ("stride", stride_flat), ("target_labels", target_labels),
                        ("target_bboxes", target_bboxes), ("target_scores", target_scores), ("fg_mask", fg_u8),
                        ("out", out), ("workspace", ws)):
            setattr(d, name, C.c_void_p(t.data_ptr()))
        d.B, d.A, d.C = B, A, Cn
        d.use_dfl, d.reg_max, d.iou_type = int(use_dfl), int(self.reg_max if use_dfl else 0), _lib.IOU_TYPES[self.iou_type]
        d.w_class, d.w_iou, d.w_dfl = (float(weights[k]) for k in ("class", "iou", "dfl"))
        d.workspace_bytes = ws.numel()
        d.box_mode = self.box_mode
        d.norm_mode = int(norm_mode)
        _lib.check(lib.y6_loss_forward(C.byref(d), _lib.current_stream_ptr()), "loss_forward")
        keep = (pred_scores, pred_distri, pred_bboxes, anchor_points_s, stride_flat, target_labels, target_bboxes,
                target_scores, fg_u8, out, ws)
        return dict(desc=d, out=out, keep=keep, pred_scores=pred_scores, pred_distri=pred_distri, pred_bboxes=pred_bboxes,
                    anchor_points_s=anchor_points_s, target_scores=target_scores, fg=fg_u8, B=B, A=A, C=Cn,
                    assigned=(target_labels, target_bboxes, target_scores, fg_u8))
