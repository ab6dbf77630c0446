"""ComputeLoss of the N / S self-distillation training.  Reference: yolov6/models/losses/loss_distill_ns.py:15-208 - the loss of
loss_distill.py for the head of heads/effidehead_distill_ns.py, whose training branch has a FOURTH output, plain (l, t, r, b)
distances from `reg_preds`:
    outputs            (feats, pred_scores, pred_distri, pred_lrtb)                                        (:75)
    assignment         TaskAlignedAssigner only, no ATSS warm-up                                           (:96-104)
    IoU loss           the SUM of the DFL-decoded boxes' and the plain-distance boxes' IoU losses          (BboxLoss :265-325)
Everything else (normalisation, distillation terms, weights) is loss_distill.py's - see that mirror."""
from .loss_distill import ComputeLoss as _DistillLoss


class ComputeLoss(_DistillLoss):
    has_lrtb = True
    use_warmup = False
