"""ComputeLoss of self-distillation training - mirror of reference yolov6/models/losses/loss_distill.py:15-208 (constructor
arguments, call signature, return value) on the HIP path, value AND gradient:

    detection terms      the kernels of loss.py's mirror (assigner, VarifocalLoss, IoU loss, DFL) with this file's normalisation
                         rules (:178-183 class term / target_scores_sum when that is > 0; BboxLoss :283-330 box terms unless it
                         is exactly 0)                                                     y6_loss_forward (norm_mode 1)
    distill_loss_cls     :210-221   KL(softmax(teacher / T) || softmax(student / T)) over the class scores of ALL anchors, x T^2
    distill_loss_dfl     :349-359   the same KL over the DFL bins of the POSITIVE anchors, summed over bins, MEAN over
                         (positives x 4 sides) - a scalar - x each positive's weight, summed, / target_scores_sum, x T^2
                                                                                           y6_distill_forward / _backward
    distill_loss_cw      :222-246   (distill_feat) channel-wise KL of the three neck maps   y6_distill_cw
    weights              :193-207   cosine decay ((1 - cos(epoch pi / max_epoch)) / 2)(0.01 - 1) + 1 on the three distillation
                         terms; loss = class (cls + d_cls w) + iou iou + dfl (dfl + d_dfl w) + cwd d_cw
The sums come from kernels; the handful of scalar operations that combine them run as torch ops on the device (no host sync).
`loss.backward()` writes d loss / d pred_scores and d loss / d pred_distri into the training graph's head-gradient buffers
like loss.py's mirror does.  The teacher's outputs are plain tensors (the reference computes them under torch.no_grad(),
core/engine.py:153-156).  Channel-wise feature distillation returns its gradient through autograd when the student feature
maps are autograd leaves; for the feature maps of the native training graph it is written into the graph's gradient inlets
(`model.distill_feat = True` before the first training forward: train_engine.TrainBuilder.grad_inlet).
"""
import ctypes as C
import math

import torch

from ... import _lib
from .loss import ComputeLoss as _ComputeLoss


class _DistillFn(torch.autograd.Function):
    """loss as a function of (pred_scores, pred_distri[, pred_lrtb], *student feature maps)."""

    @staticmethod
    def forward(ctx, holder, *inputs):
        ctx.holder = holder
        ctx.dtypes = [t.dtype for t in inputs]
        return holder["loss"].clone()

    @staticmethod
    def backward(ctx, gout):
        h = ctx.holder
        lib = _lib.load()
        stream = _lib.current_stream_ptr()
        gs = gout.detach().to(torch.float32).reshape(1).contiguous()
        g = h["grad_desc"]
        g.grad_scale = C.c_void_p(gs.data_ptr())
        _lib.check(lib.y6_loss_backward(C.byref(g), stream), "loss_backward")          # detection terms: overwrite dscores / ddistri
        grads = [h["dscores"], h["ddistri"]]
        if h.get("lrtb_desc") is not None:                                               # IoU term of the plain-distance branch
            g2 = h["lrtb_desc"]
            g2.grad_scale = C.c_void_p(gs.data_ptr())
            _lib.check(lib.y6_loss_backward(C.byref(g2), stream), "loss_backward")
            grads.append(h["dlrtb"])
        coef = (h["coef"] * gs).float().contiguous()                                     # [2]: d loss / d (sum KL cls), d (sum KL dfl)
        d = h["distill_desc"]
        d.coef = C.c_void_p(coef.data_ptr())
        d.dscores, d.ddistri = C.c_void_p(h["dscores"].data_ptr()), C.c_void_p(h["ddistri"].data_ptr())
        _lib.check(lib.y6_distill_backward(C.byref(d), stream), "distill_backward")      # adds the distillation terms
        for i, (sf, tf, rows, hw, k) in enumerate(h["cw"]):
            ck = (k * gs).float().contiguous()
            dsf = torch.empty_like(sf)
            _lib.check(lib.y6_distill_cw(C.c_void_p(sf.data_ptr()), C.c_void_p(tf.data_ptr()), rows, hw, 1.0, None,
                                         C.c_void_p(ck.data_ptr()), C.c_void_p(dsf.data_ptr()), stream), "distill_cw")
            if h.get("inlets") is not None:          # the native training graph: its backward plan starts from these buffers
                h["inlets"][i].to_nhwc_tensor().copy_(dsf.permute(0, 2, 3, 1))
            else:
                grads.append(dsf)
        h["keep_bwd"] = (gs, coef)
        out = [None] + [gr.to(dt) for gr, dt in zip(grads, ctx.dtypes)]
        return tuple(out + [None] * (1 + len(ctx.dtypes) - len(out)))


class ComputeLoss(_ComputeLoss):
    '''Loss computation func.'''
    has_lrtb = False          # loss_distill_ns.py: a fourth student output, plain (l, t, r, b) distances
    use_warmup = True         # loss_distill.py:96-104 warms up with ATSS; the N / S variant does not

    def __init__(self, fpn_strides=[8, 16, 32], grid_cell_size=5.0, grid_cell_offset=0.5, num_classes=80, ori_img_size=640,
                 warmup_epoch=0, use_dfl=True, reg_max=16, iou_type='giou',
                 loss_weight={'class': 1.0, 'iou': 2.5, 'dfl': 0.5, 'cwd': 10.0}, distill_feat=False,
                 distill_weight={'class': 1.0, 'dfl': 1.0}):
        super().__init__(fpn_strides, grid_cell_size, grid_cell_offset, num_classes, ori_img_size, warmup_epoch, use_dfl, reg_max,
                         iou_type, loss_weight)
        self.distill_feat = distill_feat
        self.distill_weight = distill_weight

    def __call__(self, outputs, t_outputs, s_featmaps, t_featmaps, targets, epoch_num, max_epoch, temperature, step_num,
                 batch_height, batch_width):
        lib = _lib.load()
        stream = _lib.current_stream_ptr()
        if self.has_lrtb:
            feats, pred_scores, pred_distri, pred_lrtb = outputs
        else:
            feats, pred_scores, pred_distri = outputs
            pred_lrtb = None
        t_pred_scores, t_pred_distri = t_outputs[-2], t_outputs[-1]
        if tuple(t_pred_scores.shape) != tuple(pred_scores.shape) or (self.use_dfl and tuple(t_pred_distri.shape) != tuple(pred_distri.shape)):
            raise RuntimeError(f"yolov6_amd: teacher outputs {tuple(t_pred_scores.shape)} / {tuple(t_pred_distri.shape)} do not match the "
                               f"student's {tuple(pred_scores.shape)} / {tuple(pred_distri.shape)} (t_outputs = (feats, cls_scores, reg_distri))")
        w = self.loss_weight
        t = self._forward_terms(feats, pred_scores, pred_distri, targets, epoch_num, batch_height, batch_width, norm_mode=1,
                                warmup=self.use_warmup)
        dev = t["out"].device
        out = t["out"]                                   # [total, w_iou iou, w_dfl dfl, w_class cls, target_scores_sum, positives]
        iou_w, dfl_w, cls_w, ts = out[1], out[2], out[3], out[4]
        keep = [t]
        t2 = None
        if pred_lrtb is not None:                        # loss_distill_ns.py:88-91, BboxLoss :265-325: + IoU loss of the plain distances
            t2 = self._forward_terms(feats, pred_scores, pred_lrtb, targets, epoch_num, batch_height, batch_width, norm_mode=1,
                                     use_dfl=False, weights={'class': 0.0, 'iou': w['iou'], 'dfl': 0.0}, assigned=t["assigned"])
            iou_w = iou_w + t2["out"][1]
            keep.append(t2)
        # ---- distillation sums
        ts_s = t_pred_scores.detach().float().contiguous()
        acc = torch.zeros(4, dtype=torch.float64, device=dev)
        d = _lib.DistillDesc()
        d.scores_s, d.scores_t = C.c_void_p(t["pred_scores"].data_ptr()), C.c_void_p(ts_s.data_ptr())
        td = None
        if self.use_dfl:
            td = t_pred_distri.detach().float().contiguous()
            d.distri_s, d.distri_t = C.c_void_p(t["pred_distri"].data_ptr()), C.c_void_p(td.data_ptr())
            d.fg_mask, d.target_scores = C.c_void_p(t["fg"].data_ptr()), C.c_void_p(t["target_scores"].data_ptr())
        d.BA, d.C, d.reg_max = t["B"] * t["A"], t["C"], int(self.reg_max if self.use_dfl else 0)
        d.temperature = float(temperature)
        d.acc = C.c_void_p(acc.data_ptr())
        _lib.check(lib.y6_distill_forward(C.byref(d), stream), "distill_forward")
        T2 = float(temperature) ** 2
        decay = ((1 - math.cos(epoch_num * math.pi / max_epoch)) / 2) * (0.01 - 1) + 1     # :193
        d_cls = acc[0] * T2
        npos = acc[3]
        # mean over (positives x 4 sides) x sum of the positives' weights / target_scores_sum  (BboxLoss :317-326)
        per_sum = torch.where(npos > 0, T2 * acc[2] / (4.0 * npos.clamp(min=1.0)), torch.zeros_like(npos))
        per_sum = torch.where(ts != 0, per_sum / torch.where(ts != 0, ts, torch.ones_like(ts)), per_sum)
        d_dfl = acc[1] * per_sum
        # ---- channel-wise feature distillation
        cw_items, d_cw = [], torch.zeros((), dtype=torch.float64, device=dev)
        if self.distill_feat:
            for sf, tf in zip(s_featmaps, t_featmaps):
                if not isinstance(sf, torch.Tensor):
                    raise NotImplementedError("yolov6_amd: distill_feat needs tensors as student feature maps")
                N, Cf, H, W = sf.shape
                sfc, tfc = sf.detach().float().contiguous(), tf.detach().float().contiguous()
                a1 = torch.zeros(1, dtype=torch.float64, device=dev)
                _lib.check(lib.y6_distill_cw(C.c_void_p(sfc.data_ptr()), C.c_void_p(tfc.data_ptr()), N * Cf, H * W, 1.0,
                                             C.c_void_p(a1.data_ptr()), None, None, stream), "distill_cw")
                d_cw = d_cw + a1[0] / (N * Cf)
                cw_items.append((sfc, tfc, N * Cf, H * W, float(w['cwd']) * decay / (N * Cf)))
        dwc, dwd = float(self.distill_weight['class']), float(self.distill_weight['dfl'])
        cls_all = cls_w + float(w['class']) * dwc * decay * d_cls
        dfl_all = dfl_w + float(w['dfl']) * dwd * decay * d_dfl
        cw_w = float(w['cwd']) * decay * d_cw
        loss = (cls_all + iou_w + dfl_all + cw_w).float()
        items = torch.stack([iou_w, dfl_all, cls_all, cw_w]).float().detach()
        leaves = [outputs[1], outputs[2]] + ([pred_lrtb] if pred_lrtb is not None else [])
        feat_leaves = [sf for sf in s_featmaps if isinstance(sf, torch.Tensor) and sf.requires_grad] if self.distill_feat else []
        inlets = None
        if self.distill_feat and not feat_leaves and torch.is_grad_enabled() and any(x.requires_grad for x in leaves):
            fg_ = getattr(s_featmaps, "_y6_graph", None)
            inlets = getattr(fg_, "feat_inlets", None)
            if inlets is None:
                raise NotImplementedError("yolov6_amd: channel-wise feature distillation needs gradient inlets on the neck feature maps: "
                                          "set `model.distill_feat = True` before the model's first training forward")
        if not (torch.is_grad_enabled() and any(x.requires_grad for x in leaves + feat_leaves)):
            return loss, items
        ps_in, pd_in = outputs[1], outputs[2]
        graph = getattr(ps_in, "_y6_graph", None)
        gb = (lambda x: graph.grad_buffer_of(x)) if graph is not None else (lambda x: None)
        dscores, ddistri = gb(ps_in), gb(pd_in)
        if dscores is None or ddistri is None:
            dscores, ddistri = torch.empty_like(t["pred_scores"]), torch.empty_like(t["pred_distri"])
        g = _lib.LossGradDesc()
        g.fwd = t["desc"]
        g.dpred_scores, g.dpred_distri = C.c_void_p(dscores.data_ptr()), C.c_void_p(ddistri.data_ptr())
        holder = dict(loss=loss, grad_desc=g, dscores=dscores, ddistri=ddistri, distill_desc=d, keep=(keep, ts_s, td, acc),
                      # d loss / d acc[0], d loss / d acc[1] (device scalars; the incoming gradient is multiplied in at backward time)
                      coef=torch.stack([torch.full((), float(w['class']) * dwc * decay * T2, dtype=torch.float64, device=dev),
                                        float(w['dfl']) * dwd * decay * per_sum]),
                      cw=cw_items if (feat_leaves or inlets is not None) else [], inlets=inlets)
        if t2 is not None:
            dlrtb = gb(pred_lrtb)
            if dlrtb is None:
                dlrtb = torch.empty_like(t2["pred_distri"])
            scratch = torch.empty_like(t["pred_scores"])        # (the class gradient of the second descriptor: weight 0)
            g2 = _lib.LossGradDesc()
            g2.fwd = t2["desc"]
            g2.dpred_scores, g2.dpred_distri = C.c_void_p(scratch.data_ptr()), C.c_void_p(dlrtb.data_ptr())
            holder.update(lrtb_desc=g2, dlrtb=dlrtb, scratch=scratch)
        return _DistillFn.apply(holder, *(leaves + feat_leaves)), items
