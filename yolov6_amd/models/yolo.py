"""Model assembly.  Reference: yolov6/models/yolo.py (Model :14-47, build_network :55-133,
build_model :136-138).  `Model.forward(x)` keeps the reference contract
`[detections[B,A,5+nc] fp32, featmaps]` but runs as one native plan of HIP kernels.
"""
import math
import sys
import weakref

import torch
import torch.nn as nn

from ..layers.common import HipModule, get_block
from ..utils import nms as _nms
from ..utils.torch_utils import initialize_weights
from . import efficientrep, reppan
from .effidehead import Detect, build_effidehead_layer


class _LazyFeatmaps(list):
    """The neck outputs as NCHW tensors, converted from the plan's NHWC buffers on first use.
    (Eval callers drop them: `outputs, _ = model(imgs)` evaler.py:128.)  The reference returns independent
    tensors, so a result that is still unread when the plan is about to run again is materialised first
    (`Model.forward` calls `_fill()` on the previous result through a weak reference)."""

    def __init__(self, refs, dtype):
        super().__init__()
        self._refs, self._dtype = refs, dtype

    def _fill(self):
        if self._refs is not None:
            refs, self._refs = self._refs, None
            for r in refs:
                super().append(r.to_nhwc_tensor().permute(0, 3, 1, 2).contiguous().to(self._dtype))

    def __len__(self):
        self._fill()
        return super().__len__()

    def __iter__(self):
        self._fill()
        return super().__iter__()

    def __getitem__(self, i):
        self._fill()
        return super().__getitem__(i)


class _ResultRing:
    """Result tensors of one plan, handed out WITHOUT a copy and never overwritten while somebody still holds them.

    The reference returns an independent tensor per call (yolo.py:37-47: a fresh `torch.cat`).  Here the decode op writes into
    one of `n` tensors of this ring, re-pointed per call (`Plan.rebind_output`).  A slot is written again only when nothing
    but the ring references its tensor: CPython's reference count of the tensor object covers the caller's own name, list
    entries and every view made from it (a view keeps its base alive through `_base`).  A slot that is still held is given
    away - the ring allocates a fresh tensor for that position (no copy, no clone; the caching allocator makes it one pointer
    bump) - so 3-pass TTA, results collected over several batches and consumers on other streams (which must hold a reference
    while their kernels run, as with any caching-allocator tensor) all see the reference's semantics.  The usual caller
    (evaler.py:128-132, inferer.py:61-63: straight into non_max_suppression, result dropped) never triggers an allocation."""

    def __init__(self, like, n):
        self.slots = [self._fresh(like) for _ in range(n)]
        self.pos = 0
        probe = [torch.empty(0)]
        self._base = self._refs(probe, 0)        # references to a tensor that only its list holds, counted the same way

    @staticmethod
    def _refs(lst, i):
        return sys.getrefcount(lst[i])

    @staticmethod
    def counts_views(t):
        """Does making a view of THIS tensor raise its CPython reference count?  True for ordinary and no_grad tensors on torch 2.x
        (a view keeps its base alive through the Python object), False for inference tensors (`torch.inference_mode()`: views of
        an inference tensor do not reference the base's PyObject - `boxes = det[..., :4]; del det` would leave the slot looking
        free).  Decided per slot on the slot's own tensor (a process-wide probe made in whatever grad mode the first
        forward happened to run in answered for every later slot)."""
        if t.is_inference():
            return False
        holder = [t]
        before = sys.getrefcount(holder[0])
        view = holder[0][:0]
        ok = sys.getrefcount(holder[0]) > before
        del view
        return ok

    @staticmethod
    def _fresh(like):
        # ring slots are ordinary tensors even when the forward runs under torch.inference_mode(): their views must count
        with torch.inference_mode(False):
            return torch.empty_like(like)

    def next(self, held_by_plan=None):
        """The tensor the next run writes.  `held_by_plan`: the plan's current output (its own reference is not a caller's)."""
        self.pos = (self.pos + 1) % len(self.slots)
        extra = 1 if held_by_plan is self.slots[self.pos] else 0
        if self._refs(self.slots, self.pos) > self._base + extra:
            self.slots[self.pos] = self._fresh(self.slots[self.pos])         # the caller keeps the old one
        return self.slots[self.pos]


class Model(HipModule):
    export = False
    # Eval results are handed out WITHOUT a copy and stay valid for as long as the caller holds them (`_ResultRing`): the
    # decode kernel alternates between this many output tensors per plan, re-using one only after the caller has dropped it.
    # 0: clone every result (+91 MB of traffic per b32 call).
    output_buffers = 2

    def __init__(self, config, channels=3, num_classes=None, fuse_ab=False, distill_ns=False):
        super().__init__()
        num_layers = config.model.head.num_layers
        self.backbone, self.neck, self.detect = build_network(config, channels, num_classes, num_layers,
                                                              fuse_ab=fuse_ab, distill_ns=distill_ns)
        self.stride = self.detect.stride
        self.detect.initialize_biases()
        initialize_weights(self)

    def lower(self, pb, x, out=None):
        feats = self.neck.lower(pb, self.backbone.lower(pb, x))
        self._featrefs = list(feats)
        pb.keep_fp16(self._featrefs)          # read lazily by the caller (int8 lowering: keep their fp16 form)
        return self.detect.lower(pb, list(feats))

    def lower_train(self, tb, x):
        feats = list(self.neck.lower(tb, self.backbone.lower(tb, x)))
        stems, heads = self.detect.lower_train(tb, feats)
        return stems, feats, heads

    def forward(self, x):
        if torch.onnx.is_in_onnx_export() or self.export:
            raise NotImplementedError("yolov6_amd: ONNX export mode is out of scope of the HIP path")
        if self.training:
            from ..train_engine import train_forward
            _require_gpu(x)
            return train_forward(self, x)
        plan = self.compile(x)
        prev = self.__dict__.get("_last_featmaps")
        prev = prev() if prev is not None else None
        if prev is not None:
            prev._fill()            # a caller still holds the previous result unread: copy it out before overwriting
        nbuf = int(self.output_buffers)
        if nbuf >= 2:
            ring = getattr(plan, "_det_ring", None)
            if ring is None or len(ring.slots) != nbuf:
                ring = plan._det_ring = _ResultRing(plan.outputs, nbuf)
            slot = ring.next(plan.outputs)
            if _ResultRing.counts_views(slot):
                plan.rebind_output(slot)
            else:
                nbuf = 0                  # views of this tensor are not counted (an inference tensor): hand out a clone
        det = plan.run()
        feats = _LazyFeatmaps(self._featrefs, x.dtype)
        self.__dict__["_last_featmaps"] = weakref.ref(feats)
        if nbuf < 2:
            det = det.clone()
        elif isinstance(det, torch.Tensor):
            _nms.note_model_output(det, plan)      # non_max_suppression(det, ...) may take the candidates this run selected
        return [det, feats]

    def _apply(self, fn):
        self = super()._apply(fn)
        self.detect.stride = fn(self.detect.stride)
        self.detect.grid = list(map(fn, self.detect.grid))
        return self


def _require_gpu(x):
    if not x.is_cuda:
        raise RuntimeError("yolov6_amd: the HIP hot path needs ROCm tensors; there is no CPU fallback "
                           f"(got a tensor on {x.device})")


def make_divisible(x, divisor):
    return math.ceil(x / divisor) * divisor


def build_network(config, channels, num_classes, num_layers, fuse_ab=False, distill_ns=False):
    m = config.model
    depth_mul, width_mul = m.depth_multiple, m.width_multiple
    reps = m.backbone.num_repeats + m.neck.num_repeats
    chans = m.backbone.out_channels + m.neck.out_channels
    num_repeat = [(max(round(i * depth_mul), 1) if i > 1 else i) for i in reps]
    channels_list = [make_divisible(i * width_mul, 8) for i in chans]
    block = get_block(config.training_mode)
    backbone_cls = getattr(efficientrep, m.backbone.type, None)
    neck_cls = getattr(reppan, m.neck.type, None)
    if backbone_cls is None or neck_cls is None:
        raise NotImplementedError(f"yolov6_amd: backbone/neck {m.backbone.type}/{m.neck.type} is outside the HIP hot path")
    bkw = dict(in_channels=channels, channels_list=channels_list, num_repeats=num_repeat, block=block,
               fuse_P2=m.backbone.get('fuse_P2'), cspsppf=m.backbone.get('cspsppf'))
    nkw = dict(channels_list=channels_list, num_repeats=num_repeat, block=block)
    if 'CSP' in m.backbone.type:
        stage_block_type = m.backbone.get("stage_block_type") or "BepC3"
        bkw.update(csp_e=m.backbone.csp_e, stage_block_type=stage_block_type)
        nkw.update(csp_e=m.neck.csp_e, stage_block_type=stage_block_type)
    backbone, neck = backbone_cls(**bkw), neck_cls(**nkw)
    if distill_ns:   # yolo.py:114-120: the N / S head with an extra DFL branch for self-distillation (inference: plain distances)
        from .heads.effidehead_distill_ns import Detect as DetectNS, build_effidehead_layer as build_ns
        if num_layers != 3:
            raise NotImplementedError("yolov6_amd: distill_ns needs a three-level head (the reference exits here, yolo.py:116-118)")
        head_layers = build_ns(channels_list, 1, num_classes, reg_max=m.head.reg_max)
        return backbone, neck, DetectNS(num_classes, num_layers, head_layers=head_layers, use_dfl=m.head.use_dfl)
    if fuse_ab:      # yolo.py:122-126: the head with the anchor-based auxiliary branch (training recipe `--fuse_ab`)
        from .heads.effidehead_fuseab import Detect as DetectAB, build_effidehead_layer as build_ab
        head_layers = build_ab(channels_list, 3, num_classes, reg_max=m.head.reg_max, num_layers=num_layers)
        return backbone, neck, DetectAB(num_classes, m.head.anchors_init, num_layers, head_layers=head_layers, use_dfl=m.head.use_dfl)
    head_layers = build_effidehead_layer(channels_list, 1, num_classes, reg_max=m.head.reg_max, num_layers=num_layers)
    # like the reference (yolo.py:128-130) Detect keeps its default reg_max=16 (proj has 17 bins even
    # when use_dfl is False); only build_effidehead_layer sees the config's reg_max
    head = Detect(num_classes, num_layers, head_layers=head_layers, use_dfl=m.head.use_dfl)
    return backbone, neck, head


def build_model(cfg, num_classes, device, fuse_ab=False, distill_ns=False):
    return Model(cfg, channels=3, num_classes=num_classes, fuse_ab=fuse_ab, distill_ns=distill_ns).to(device)
