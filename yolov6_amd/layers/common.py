"""Building blocks with the reference's names, constructor signatures and state_dict keys
(yolov6/layers/common.py), whose forward runs on the HIP hot path.

Each block knows how to *lower* itself into a native plan (yolov6_amd.engine.PlanBuilder):
  ConvModule / ConvBN{ReLU,SiLU,HS}   -> one fused conv+bias+act kernel        (common.py:26-94)
  RepVGGBlock / QARepVGGBlock[V2]     -> re-parameterised 3x3 conv (+post-BN)   (:197-477)
  RepBlock / BottleRep / BepC3        -> sequences, residual in the conv epilogue,(:569-650)
                                         concat-free channel slices
  SPPFModule / CSPSPPFModule          -> 1x1 convs + one pooling kernel         (:97-178)
  Transpose / BiFusion                -> convT as 4 scatter GEMMs, slice writes (:181-194, :695-718)

`forward(x)` keeps the reference convention (NCHW tensors in/out).  Modules hold ordinary
torch parameters (checkpoints / state_dicts stay interchangeable); packed MFMA weights are
derived caches inside the plan.  There is no aten fallback: CPU tensors, training-mode
BatchNorm and grouped convs raise.
"""
import os

import operator

import torch
import torch.nn as nn
import torch.nn.functional as F

from ..engine import NCHWInput, PlanBuilder, TRef

# the reference keeps one shared activation module per kind (common.py:14-17); state_dicts do
# not depend on it but `model.modules()` walkers (initialize_weights) do.
activation_table = {"relu": nn.ReLU(), "silu": nn.SiLU(), "hardswish": nn.Hardswish()}


# ------------------------------------------------------------------------------------------
# re-parameterisation math (fp32, on whatever device the parameters live)
# ------------------------------------------------------------------------------------------
def bn_scale_shift(bn: nn.BatchNorm2d):
    """Eval-mode BatchNorm as y = x*scale + shift (uses bn.eps, set to 1e-3 by initialize_weights)."""
    scale = bn.weight.detach().float() / torch.sqrt(bn.running_var.detach().float() + bn.eps)
    shift = bn.bias.detach().float() - bn.running_mean.detach().float() * scale
    return scale, shift


def fold_conv_bn(weight, bias, bn):
    """fuse_conv_and_bn (reference yolov6/utils/torch_utils.py:50-82) as tensors."""
    scale, shift = bn_scale_shift(bn)
    w = weight.detach().float() * scale.view(-1, 1, 1, 1)
    b = shift if bias is None else shift + bias.detach().float() * scale
    return w, b


def identity_kernel3x3(channels, device):
    k = torch.zeros(channels, channels, 3, 3, device=device)
    idx = torch.arange(channels, device=device)
    k[idx, idx, 1, 1] = 1.0
    return k


# ------------------------------------------------------------------------------------------
# forward machinery shared by every block
# ------------------------------------------------------------------------------------------
def _flatten(x):
    if isinstance(x, torch.Tensor):
        return [x]
    out = []
    for e in x:
        out += _flatten(e)
    return out


def _wrap(x, it):
    """Same nesting as x with every tensor replaced by NCHWInput(next contiguous tensor)."""
    if isinstance(x, torch.Tensor):
        return NCHWInput(next(it))
    return type(x)(_wrap(e, it) for e in x) if isinstance(x, tuple) else [_wrap(e, it) for e in x]


# Kernels that write parameters or buffers behind autograd's back (the fused SGD over the arena, the training forward's
# running-statistics update, the EMA lerp) do not bump `tensor._version`; they bump this process-wide generation instead,
# which is part of every plan-cache key: an eval forward after a native update re-derives its packed weights / folded BN.
_NATIVE_GENERATION = [0]
# Bumped by every `_apply` (.half() / .to() / .cuda()) and every `invalidate_plans()` of ANY HipModule: the fast path of
# `compile()` below trusts a cached plan only while no module anywhere in the process was moved or re-parameterised.
_STRUCTURE_GENERATION = [0]


def bump_native_generation():
    _NATIVE_GENERATION[0] += 1


def resolve_autotune(autotune):
    """How a plan gets its conv kernels.  `None` (what `model(x)` passes): from the layer shapes - the same kernels, fp32
    summation orders and output bits in every process (the drop-in contract: same outputs on the same inputs) - unless the
    environment says Y6_AUTOTUNE=1.  `True`: timed in this process (y6_plan_autotune), the table kept on disk per (device, library
    build) so that later processes on the machine replay it instead of re-timing (`autotune_cache_path`).  Y6_AUTOTUNE=0 forces
    shape-derived kernels for EVERY plan - compile(), new_plan(), InflightRunner - whatever the caller passed."""
    env = os.environ.get("Y6_AUTOTUNE")
    if env == "0":
        return False
    if autotune is None:
        return env == "1"
    return bool(autotune)


_cache_path_set = [False]


def autotune_cache_path():
    """Point the library's persistent kernel table (env Y6_AUTOTUNE_CACHE, read by y6_plan_autotune) at a per-(device, library
    build) file unless the caller chose one: <Y6_CACHE_DIR | $XDG_CACHE_HOME/yolov6_amd | ~/.cache/yolov6_amd>/autotune-<device>-
    <md5 of libyolov6_hip.so>.txt.  A line is `<layer signature> <variant name>`; the shapes are part of the signature.
    Y6_AUTOTUNE_CACHE="" disables the file."""
    if _cache_path_set[0] or "Y6_AUTOTUNE_CACHE" in os.environ:
        if os.environ.get("Y6_AUTOTUNE_CACHE") == "":
            os.environ.pop("Y6_AUTOTUNE_CACHE")
            _cache_path_set[0] = True
        return os.environ.get("Y6_AUTOTUNE_CACHE")
    _cache_path_set[0] = True
    try:
        import hashlib
        from .. import _lib
        with open(_lib.LIB_PATH, "rb") as f:
            h = hashlib.md5(f.read()).hexdigest()[:12]
        dev = torch.cuda.get_device_name(torch.cuda.current_device()).replace(" ", "_").replace("/", "_")
        root = os.environ.get("Y6_CACHE_DIR") or os.path.join(os.environ.get("XDG_CACHE_HOME") or os.path.expanduser("~/.cache"), "yolov6_amd")
        os.makedirs(root, exist_ok=True)
        os.environ["Y6_AUTOTUNE_CACHE"] = os.path.join(root, f"autotune-{dev}-{h}.txt")
    except Exception:       # noqa: BLE001 - an unwritable home directory must not stop a forward; the plan is tuned without a file
        return None
    return os.environ["Y6_AUTOTUNE_CACHE"]


def _params_version(module):
    """Plan-cache key part: identity, version and dtype of every parameter / buffer plus the scalar attributes that
    lowering bakes into the plan.  In-place edits through `.data` do not bump `_version`: after such an edit call
    `module.invalidate_plans()` (switch_to_deploy, fuse_model and `_apply` do)."""
    items = []
    for t in list(module.parameters()) + list(module.buffers()):
        items.append((t.data_ptr(), t._version, t.dtype))
    for m in module.modules():
        if isinstance(m, nn.BatchNorm2d):
            items.append(("eps", m.eps))
        for name in ("use_dfl", "reg_max", "grid_cell_offset", "shortcut", "deploy", "nc"):
            v = m.__dict__.get(name)
            if isinstance(v, (bool, int, float)):
                items.append((name, v))
        st = m.__dict__.get("stride")
        if isinstance(st, torch.Tensor):
            items.append(("stride", tuple(st.tolist())))
        a = m.__dict__.get("alpha")
        if isinstance(a, float):
            items.append(("alpha", a))
    return hash(tuple(items))


_VERSION_OF = operator.attrgetter("_version")


class HipModule(nn.Module):
    """nn.Module whose forward is a cached native plan of HIP kernels."""

    def lower(self, pb: PlanBuilder, x, out=None):
        raise NotImplementedError

    # plans hold ctypes handles: never pickle them with the module (checkpoints pickle modules)
    # (`copy.deepcopy(model)` and `torch.save(model)` both go through here: the reference's epoch-end path is
    # `deepcopy(de_parallel(model)).half()` + save_checkpoint, core/engine.py:192-203.)  Every `_y6_*` entry is native
    # state: plans, training graphs (ctypes handles), the parameter arena, the int8 calibration, the backward hook.
    def __getstate__(self):
        st = {k: v for k, v in self.__dict__.items() if not k.startswith("_y6_")}
        st.pop("_featrefs", None)
        st.pop("_last_featmaps", None)
        return st

    def invalidate_plans(self):
        """Drop every cached native plan below this module (packed weights are derived caches: call this after
        editing parameters through `.data`, which autograd's version counter does not see) and the training graphs with
        their parameter arena (their plans hold raw pointers to parameters, BatchNorm buffers and arena slots).  The
        Parameters keep their values: `p.data` stays a live view of the dropped arena until `_apply` / the next training
        graph re-points it."""
        _STRUCTURE_GENERATION[0] += 1
        for m in self.modules():
            d = m.__dict__
            d.pop("_y6_plans", None)
            d.pop("_y6_fast", None)
            d.pop("_y6_train_graphs", None)
            d.pop("_y6_arena", None)

    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)
        self.invalidate_plans()
        return out

    def _check_runnable(self):
        for m in self.modules():
            if isinstance(m, nn.BatchNorm2d) and m.training:
                raise NotImplementedError(
                    "yolov6_amd: a single block in .train() mode has no standalone HIP forward; batch-statistics "
                    "BatchNorm runs through the whole-model training graph (Model.forward in train mode, "
                    "yolov6_amd/train_engine.py) - call .eval() for block-level inference")

    def _finish_outputs(self, pb, outs, dtype):
        if isinstance(outs, TRef):
            return pb.to_nchw(outs, dtype)
        if isinstance(outs, torch.Tensor):
            return outs
        return type(outs)(self._finish_outputs(pb, o, dtype) for o in outs) if isinstance(outs, tuple) else \
            [self._finish_outputs(pb, o, dtype) for o in outs]

    def compile(self, *inputs, autotune=None):
        """Build (or fetch) the plan for these input shapes and bind it to these tensors.

        autotune=None (the default, what `model(x)` uses) and autotune=False take the kernel of every layer from its shape: the
        same bits in every process, within a few per cent of a tuned plan's speed.  autotune=True (or Y6_AUTOTUNE=1 in the
        environment for the default) times kernel variants (per layer, then the whole step: include/yolov6_hip.h
        y6_plan_autotune) and keeps the table on disk per (device, library build): the first process on a machine tunes, the
        others replay its choices (`resolve_autotune`, `autotune_cache_path`).  Y6_AUTOTUNE=0 forces shape-derived kernels."""
        autotune = resolve_autotune(autotune)
        x = inputs[0] if len(inputs) == 1 else list(inputs)
        flat = _flatten(x)
        for t in flat:
            if not t.is_cuda:
                raise RuntimeError("yolov6_amd: the HIP hot path needs ROCm tensors; there is no CPU fallback "
                                   f"(got a tensor on {t.device})")
        contig = [t.contiguous() for t in flat]
        quant = self.__dict__.get("_y6_quant")       # yolov6_amd.quant: calibration pass / int8 lowering
        # Fast path (the per-call cost of the reference-signature API): the full key below walks every module of the tree
        # (~1 ms of Python for YOLOv6-S); a repeat call is recognised by the input signature, the process-wide
        # generations and the SUM of the autograd version counters of the tensors the plan was built from (in-place
        # updates through torch - optimizers, load_state_dict, copy_ - bump them) plus the identity of every registered
        # parameter / buffer (a replaced Parameter is a different object).  Edits through `.data` or of scalar attributes
        # (eps, use_dfl, ...) need `invalidate_plans()`, as before.
        sig = (tuple((tuple(t.shape), t.dtype) for t in flat), self.training, autotune, None if quant is None else quant.key(),
               _NATIVE_GENERATION[0], _STRUCTURE_GENERATION[0])
        fast = self.__dict__.get("_y6_fast")
        # (both scans run in C: map() over operator callables - as generator expressions they were 35 of the ~60 us this check costs
        # per call, and the GPU idles from the NMS sync until the next forward's first launch, tools/dropin_profile.py)
        if (fast is not None and fast[0] == sig and sum(map(_VERSION_OF, fast[1])) == fast[2]
                and all(map(operator.is_, map(dict.get, fast[4][0], fast[4][1]), fast[4][2]))):   # every captured parameter / buffer is still the registered one
            plan = fast[3]
            plan.bind_inputs([contig[j] for j in plan.input_order])
            return plan
        key = (tuple((tuple(t.shape), t.dtype) for t in flat), _params_version(self), self.training, autotune,
               None if quant is None else quant.key(), _NATIVE_GENERATION[0])
        cache = self.__dict__.setdefault("_y6_plans", {})
        plan = cache.get(key)
        if plan is None:
            cache.clear()  # one live plan per module: buffers are large
            plan = self._lower_plan(x, flat, contig, quant, autotune)
            cache[key] = plan
        else:
            plan.bind_inputs([contig[j] for j in plan.input_order])
        tensors = list(self.parameters()) + list(self.buffers())
        # (registry dict, name, tensor) of every parameter and buffer below this module: `m.weight = nn.Parameter(...)` replaces
        # the entry, and the version counters of the OLD tensors this tuple keeps alive would not notice
        holders = [(d, n, t) for m in self.modules() for d in (m._parameters, m._buffers) for n, t in d.items() if t is not None]
        holders = tuple(list(col) for col in zip(*holders)) if holders else ([], [], [])      # (dicts, names, tensors): parallel lists
        self.__dict__["_y6_fast"] = (sig, tensors, sum(t._version for t in tensors), plan, holders)
        return plan

    def _lower_plan(self, x, flat, contig, quant, autotune, variants_from=None):
        self._check_runnable()
        autotune = resolve_autotune(autotune)        # (new_plan() arrives here too: Y6_AUTOTUNE=0 reaches every plan)
        if autotune and variants_from is None:
            autotune_cache_path()
        if quant is not None:
            quant.decisions = None
            if quant.mode == "int8" and quant.twins:
                # scan lowering: who reads / writes which buffer with which scale -> int8 twins (quant.plan_twins)
                from ..quant import plan_twins
                quant.begin_lowering()
                scan = PlanBuilder(flat[0].device, quant=quant)
                self.lower(scan, _wrap(x, iter(contig)))
                quant.decisions = plan_twins(scan)
                del scan
            quant.begin_lowering()
        pb = PlanBuilder(flat[0].device, quant=quant)
        outs = self.lower(pb, _wrap(x, iter(contig)))
        odt = flat[0].dtype if flat[0].dtype in (torch.float16, torch.float32) else torch.float16
        outs = self._finish_outputs(pb, outs, odt)
        plan = pb.finalize(outs, autotune=autotune, variants_from=variants_from)
        # builder order of the boundary tensors -> position in the caller's argument list
        plan.input_order = [next(j for j, c in enumerate(contig) if c is t) for t in plan.inputs]
        plan.params_version = _params_version(self)      # what the packed weights of this plan were derived from
        plan.quant_key = None if quant is None else quant.key()
        # native kernels (fused SGD, running-statistics update, EMA) and .half() / .to() / invalidate_plans() move the weights
        # without touching autograd's version counters: the two process-wide generations are part of what a plan was built from
        plan.generations = (_NATIVE_GENERATION[0], _STRUCTURE_GENERATION[0])
        plan.autotuned = bool(autotune)
        return plan

    def new_plan(self, *inputs, autotune=None, variants_from=None):
        """One MORE plan of this module for these inputs, outside the plan cache: its own activation buffers and its own packed
        copies of the weights, lowered from the SAME parameters, BatchNorm buffers and int8 calibration as the cached plan
        (pipeline.InflightRunner keeps N of them for N batches in flight).  The caller owns it; `plan.params_version` /
        `plan.quant_key` say what it was derived from (`plan_is_current()` re-checks)."""
        x = inputs[0] if len(inputs) == 1 else list(inputs)
        flat = _flatten(x)
        for t in flat:
            if not t.is_cuda:
                raise RuntimeError("yolov6_amd: the HIP hot path needs ROCm tensors; there is no CPU fallback "
                                   f"(got a tensor on {t.device})")
        contig = [t.contiguous() for t in flat]
        # variants_from: a plan of this module for the same shapes whose (tuned) kernel choices the new plan takes over
        return self._lower_plan(x, flat, contig, self.__dict__.get("_y6_quant"), autotune, variants_from=variants_from)

    def plan_is_current(self, plan) -> bool:
        """Does `plan` (compile() / new_plan()) still describe this module - same parameter tensors at the same autograd
        versions, no native update (fused SGD / running statistics / EMA) and no `_apply` / invalidate_plans() anywhere since,
        same lowering attributes, same int8 state?  (Edits through `.data` need invalidate_plans(), as for compile().)"""
        quant = self.__dict__.get("_y6_quant")
        return (getattr(plan, "generations", None) == (_NATIVE_GENERATION[0], _STRUCTURE_GENERATION[0])
                and getattr(plan, "params_version", None) == _params_version(self)
                and getattr(plan, "quant_key", None) == (None if quant is None else quant.key()))

    def forward(self, *inputs):
        plan = self.compile(*inputs)
        outs = plan.run()
        return _clone_tree(outs)


def _clone_tree(o):
    if isinstance(o, torch.Tensor):
        return o.clone()
    return type(o)(_clone_tree(e) for e in o) if isinstance(o, tuple) else [_clone_tree(e) for e in o]


# ------------------------------------------------------------------------------------------
# Conv + BN + activation
# ------------------------------------------------------------------------------------------
class ConvModule(HipModule):
    '''Conv2d(bias=False) -> BatchNorm2d -> activation.  Reference: common.py:26-54.'''

    def __init__(self, in_channels, out_channels, kernel_size, stride, activation_type, padding=None, groups=1,
                 bias=False):
        super().__init__()
        if padding is None:
            padding = kernel_size // 2
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size=kernel_size, stride=stride, padding=padding,
                              groups=groups, bias=bias)
        self.bn = nn.BatchNorm2d(out_channels)
        if activation_type is not None:
            self.act = activation_table.get(activation_type)
        self.activation_type = activation_type

    def fused_weight_bias(self):
        """(weight, bias) of the equivalent bias-conv: BN folded if it is still attached."""
        if hasattr(self, "bn"):
            return fold_conv_bn(self.conv.weight, self.conv.bias, self.bn)
        b = self.conv.bias
        return self.conv.weight.detach().float(), None if b is None else b.detach().float()

    def _activation_name(self):
        # modules un-pickled from a reference-written checkpoint carry `act` but not `activation_type`
        if "activation_type" in self.__dict__:
            return self.activation_type
        act = getattr(self, "act", None)
        for name, kind in (("relu", nn.ReLU), ("silu", nn.SiLU), ("hardswish", nn.Hardswish)):
            if isinstance(act, kind):
                return name
        if act is None:
            return None
        raise NotImplementedError(f"yolov6_amd: activation {type(act).__name__} is not on the HIP path")

    def lower(self, pb, x, out=None, res=None, res_alpha=None):
        c = self.conv
        if c.groups != 1 or c.dilation != (1, 1):
            raise NotImplementedError("yolov6_amd: grouped / dilated convs are not on the YOLOv6 N/S/M/L hot path")
        k = c.kernel_size[0]
        if c.padding != (k // 2, k // 2):
            raise NotImplementedError("yolov6_amd: only 'same' padding (k//2) is supported")
        if getattr(pb, "is_train", False):
            # training form (common.py:45-49): conv -> BatchNorm on batch statistics -> activation, with a backward tape
            if not hasattr(self, "bn") or c.bias is not None:
                raise NotImplementedError("yolov6_amd: training needs the un-fused ConvModule (conv without bias + BatchNorm)")
            y = pb.conv(x, c.weight, c.stride[0])
            return pb.bnact([(y, pb.bn(y, self.bn))], self._activation_name(), out=out, res=res, alpha=res_alpha)
        w, b = self.fused_weight_bias()
        return pb.conv(x, w, b, stride=c.stride[0], act=self._activation_name(), out=out, res=res, res_alpha=res_alpha)


class _ConvBNAct(HipModule):
    _act = None

    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, padding=None, groups=1, bias=False):
        super().__init__()
        self.block = ConvModule(in_channels, out_channels, kernel_size, stride, self._act, padding, groups, bias)

    def lower(self, pb, x, out=None, res=None, res_alpha=None):
        return self.block.lower(pb, x, out, res, res_alpha)


class ConvBNReLU(_ConvBNAct):
    '''Reference: common.py:57-64.'''
    _act = "relu"


class ConvBNSiLU(_ConvBNAct):
    '''Reference: common.py:67-74.'''
    _act = "silu"


class ConvBN(_ConvBNAct):
    '''Reference: common.py:77-84.'''
    _act = None


class ConvBNHS(_ConvBNAct):
    '''Reference: common.py:87-94.'''
    _act = "hardswish"


# ------------------------------------------------------------------------------------------
# SPPF family
# ------------------------------------------------------------------------------------------
class SPPFModule(HipModule):
    '''1x1 -> three chained 5x5 max-pools -> concat(4) -> 1x1.  Reference: common.py:97-112.'''

    def __init__(self, in_channels, out_channels, kernel_size=5, block=ConvBNReLU):
        super().__init__()
        hidden = in_channels // 2
        self.cv1 = block(in_channels, hidden, 1, 1)
        self.cv2 = block(hidden * 4, out_channels, 1, 1)
        self.m = nn.MaxPool2d(kernel_size=kernel_size, stride=1, padding=kernel_size // 2)

    def lower(self, pb, x, out=None):
        if self.m.kernel_size != 5:
            raise NotImplementedError("yolov6_amd: SPPF pooling kernel is specialised for kernel_size=5")
        x = pb.as_nhwc(x)
        hidden = self.cv1.block.conv.out_channels
        cat = pb.new_buffer(x.B, x.H, x.W, 4 * hidden)
        s = [cat.slice(i * hidden, hidden) for i in range(4)]
        self.cv1.lower(pb, x, out=s[0])
        pb.sppf_pool(s[0], s[1], s[2], s[3])
        return self.cv2.lower(pb, cat, out=out)


class SimSPPF(HipModule):
    '''Reference: common.py:115-122.'''

    def __init__(self, in_channels, out_channels, kernel_size=5, block=ConvBNReLU):
        super().__init__()
        self.sppf = SPPFModule(in_channels, out_channels, kernel_size, block)

    def lower(self, pb, x, out=None):
        return self.sppf.lower(pb, x, out)


class SPPF(HipModule):
    '''Reference: common.py:125-132.'''

    def __init__(self, in_channels, out_channels, kernel_size=5, block=ConvBNSiLU):
        super().__init__()
        self.sppf = SPPFModule(in_channels, out_channels, kernel_size, block)

    def lower(self, pb, x, out=None):
        return self.sppf.lower(pb, x, out)


class CSPSPPFModule(HipModule):
    '''CSP variant: bypass 1x1 || (1x1, 3x3, 1x1, pools, 1x1, 3x3) -> concat -> 1x1.
    Reference: common.py:135-158.'''

    def __init__(self, in_channels, out_channels, kernel_size=5, e=0.5, block=ConvBNReLU):
        super().__init__()
        hidden = int(out_channels * e)
        self.cv1 = block(in_channels, hidden, 1, 1)
        self.cv2 = block(in_channels, hidden, 1, 1)
        self.cv3 = block(hidden, hidden, 3, 1)
        self.cv4 = block(hidden, hidden, 1, 1)
        self.m = nn.MaxPool2d(kernel_size=kernel_size, stride=1, padding=kernel_size // 2)
        self.cv5 = block(4 * hidden, hidden, 1, 1)
        self.cv6 = block(hidden, hidden, 3, 1)
        self.cv7 = block(2 * hidden, out_channels, 1, 1)

    def lower(self, pb, x, out=None):
        if self.m.kernel_size != 5:
            raise NotImplementedError("yolov6_amd: SPPF pooling kernel is specialised for kernel_size=5")
        x = pb.as_nhwc(x)
        hidden = self.cv1.block.conv.out_channels
        pools = pb.new_buffer(x.B, x.H, x.W, 4 * hidden)
        ps = [pools.slice(i * hidden, hidden) for i in range(4)]
        tail = pb.new_buffer(x.B, x.H, x.W, 2 * hidden)     # cat((y0, y3))
        t = self.cv1.lower(pb, x)
        t = self.cv3.lower(pb, t)
        self.cv4.lower(pb, t, out=ps[0])
        self.cv2.lower(pb, x, out=tail.slice(0, hidden))
        pb.sppf_pool(ps[0], ps[1], ps[2], ps[3])
        t = self.cv5.lower(pb, pools)
        self.cv6.lower(pb, t, out=tail.slice(hidden, hidden))
        return self.cv7.lower(pb, tail, out=out)


class SimCSPSPPF(HipModule):
    '''Reference: common.py:161-168.'''

    def __init__(self, in_channels, out_channels, kernel_size=5, e=0.5, block=ConvBNReLU):
        super().__init__()
        self.cspsppf = CSPSPPFModule(in_channels, out_channels, kernel_size, e, block)

    def lower(self, pb, x, out=None):
        return self.cspsppf.lower(pb, x, out)


class CSPSPPF(HipModule):
    '''Reference: common.py:171-178.'''

    def __init__(self, in_channels, out_channels, kernel_size=5, e=0.5, block=ConvBNSiLU):
        super().__init__()
        self.cspsppf = CSPSPPFModule(in_channels, out_channels, kernel_size, e, block)

    def lower(self, pb, x, out=None):
        return self.cspsppf.lower(pb, x, out)


class Transpose(HipModule):
    '''ConvTranspose2d(k=2, s=2, bias) upsampling.  Reference: common.py:181-194.'''

    def __init__(self, in_channels, out_channels, kernel_size=2, stride=2):
        super().__init__()
        self.upsample_transpose = torch.nn.ConvTranspose2d(in_channels=in_channels, out_channels=out_channels,
                                                           kernel_size=kernel_size, stride=stride, bias=True)

    def lower(self, pb, x, out=None):
        ct = self.upsample_transpose
        if ct.kernel_size != (2, 2) or ct.stride != (2, 2):
            raise NotImplementedError("yolov6_amd: Transpose is specialised for kernel_size=2, stride=2")
        return pb.convt2x2(x, ct.weight, ct.bias, out=out)     # the training builder takes the Parameters themselves


# ------------------------------------------------------------------------------------------
# RepVGG family
# ------------------------------------------------------------------------------------------
class RepVGGBlock(HipModule):
    '''Training form: ReLU(conv3x3.BN + conv1x1.BN + BN_id(x)); deploy form: ReLU(conv3x3(x)+b).
    Reference: common.py:197-319.  On the HIP path both forms run as ONE fused 3x3 kernel: the
    eval-mode training form is re-parameterised while the plan is built.'''

    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, padding=1, dilation=1, groups=1,
                 padding_mode='zeros', deploy=False, use_se=False):
        super().__init__()
        assert kernel_size == 3
        assert padding == 1
        if use_se:
            raise NotImplementedError("se block not supported yet")
        self.deploy = deploy
        self.groups = groups
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.nonlinearity = nn.ReLU()
        self.se = nn.Identity()
        if deploy:
            self.rbr_reparam = nn.Conv2d(in_channels, out_channels, kernel_size, stride=stride, padding=padding,
                                         dilation=dilation, groups=groups, bias=True, padding_mode=padding_mode)
        else:
            has_identity = out_channels == in_channels and stride == 1
            self.rbr_identity = nn.BatchNorm2d(num_features=in_channels) if has_identity else None
            self.rbr_dense = ConvModule(in_channels, out_channels, kernel_size, stride, None, padding=padding,
                                        groups=groups)
            self.rbr_1x1 = ConvModule(in_channels, out_channels, 1, stride, None, padding=padding - kernel_size // 2,
                                      groups=groups)

    # ---- re-parameterisation (reference get_equivalent_kernel_bias :257-261, _fuse_bn_tensor :278-300)
    def _identity_branch(self):
        bn = getattr(self, "rbr_identity", None)
        if bn is None:
            return 0, 0
        scale, shift = bn_scale_shift(bn)
        k = identity_kernel3x3(self.in_channels, scale.device) * scale.view(-1, 1, 1, 1)
        return k, shift

    def get_equivalent_kernel_bias(self):
        k3, b3 = self.rbr_dense.fused_weight_bias()
        k1, b1 = self.rbr_1x1.fused_weight_bias()
        kid, bid = self._identity_branch()
        bias = sum(b for b in (b3, b1, bid) if b is not None)
        return k3 + F.pad(k1, [1, 1, 1, 1]) + kid, bias

    def _install_reparam(self, kernel, bias):
        d = self.rbr_dense.conv
        self.rbr_reparam = nn.Conv2d(d.in_channels, d.out_channels, d.kernel_size, stride=d.stride, padding=d.padding,
                                     dilation=d.dilation, groups=d.groups, bias=True).to(kernel.device)
        self.rbr_reparam.weight.data = kernel
        self.rbr_reparam.bias.data = bias
        for p in self.parameters():
            p.detach_()
        for name in ("rbr_dense", "rbr_1x1", "rbr_identity", "rbr_avg", "id_tensor"):
            if hasattr(self, name):
                self.__delattr__(name)
        self.deploy = True

    def switch_to_deploy(self):
        if hasattr(self, "rbr_reparam"):
            return
        if self.groups != 1:
            raise NotImplementedError("yolov6_amd: grouped RepVGG blocks are not supported")
        self._install_reparam(*self.get_equivalent_kernel_bias())

    def _deploy_weight_bias(self):
        if hasattr(self, "rbr_reparam"):
            return self.rbr_reparam.weight.detach().float(), self.rbr_reparam.bias.detach().float()
        return self.get_equivalent_kernel_bias()

    def _post_affine(self):
        return None

    def _stride(self):
        return (self.rbr_reparam if hasattr(self, "rbr_reparam") else self.rbr_dense.conv).stride[0]

    def _lower_train(self, pb, x, out, res, res_alpha):
        """Training form (common.py:250-255): ReLU(bn(conv3x3(x)) + bn(conv1x1(x)) + bn_id(x)), batch statistics."""
        if hasattr(self, "rbr_reparam"):
            raise NotImplementedError("yolov6_amd: a deployed RepVGG block cannot be trained (no branches left)")
        s = self.rbr_dense.conv.stride[0]
        y3 = pb.conv(x, self.rbr_dense.conv.weight, s)
        y1 = pb.conv(x, self.rbr_1x1.conv.weight, s)
        pairs = [(y3, self.rbr_dense.bn), (y1, self.rbr_1x1.bn)]
        if self.rbr_identity is not None:
            pairs.append((pb.as_nhwc(x), self.rbr_identity))
        stats = pb.bn_multi(pairs) if hasattr(pb, "bn_multi") else [pb.bn(t, bn) for t, bn in pairs]   # one statistics op for the block
        branches = [(t, st) for (t, _), st in zip(pairs, stats)]
        return pb.bnact(branches, "relu", out=out, res=res, alpha=res_alpha)

    def lower(self, pb, x, out=None, res=None, res_alpha=None):
        if self.groups != 1:
            raise NotImplementedError("yolov6_amd: grouped RepVGG blocks are not supported")
        if getattr(pb, "is_train", False):
            return self._lower_train(pb, x, out, res, res_alpha)
        w, b = self._deploy_weight_bias()
        return pb.conv(x, w, b, stride=self._stride(), act="relu", out=out, post=self._post_affine(), res=res,
                       res_alpha=res_alpha)


class QARepVGGBlock(RepVGGBlock):
    '''Quantisation-aware RepVGG: BN-free 1x1 branch, raw identity, post-BN kept after deploy.
    Reference: common.py:322-393.'''
    _with_avg = False

    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, padding=1, dilation=1, groups=1,
                 padding_mode='zeros', deploy=False, use_se=False):
        super().__init__(in_channels, out_channels, kernel_size, stride, padding, dilation, groups, padding_mode,
                         deploy, use_se)
        if not deploy:
            self.bn = nn.BatchNorm2d(out_channels)
            self.rbr_1x1 = nn.Conv2d(in_channels, out_channels, kernel_size=1, stride=stride, groups=groups,
                                     bias=False)
            same = out_channels == in_channels and stride == 1
            self.rbr_identity = nn.Identity() if same else None
            if self._with_avg:
                self.rbr_avg = nn.AvgPool2d(kernel_size=kernel_size, stride=stride, padding=padding) if same else None
        self._id_tensor = None

    def get_equivalent_kernel_bias(self):
        k3, b3 = self.rbr_dense.fused_weight_bias()
        kernel = k3 + F.pad(self.rbr_1x1.weight.detach().float(), [1, 1, 1, 1])
        if getattr(self, "rbr_avg", None) is not None:   # V2: 3x3 average pooling as a constant kernel (:430-432)
            kernel = kernel + identity_kernel3x3(self.in_channels, kernel.device).amax((2, 3), keepdim=True) / 9.0
        if self.rbr_identity is not None:
            kernel = kernel + identity_kernel3x3(self.in_channels, kernel.device)
        return kernel, b3

    def _lower_train(self, pb, x, out, res, res_alpha):
        """Training form (common.py:337-343 first version, :412-419 V2):
            ReLU(bn(bn3(conv3x3(x)) + conv1x1(x) [+ x [+ AvgPool3x3(x)]]))       batch statistics in both BatchNorms.
        The raw (un-normalised) 1x1 / identity / average-pool branches enter the branch sum with scale 1; identity and
        average pool are one tensor (`pb.avgpool3`).  Two branch-sum passes: the inner sum has no activation, the outer one is
        the block's BatchNorm + ReLU."""
        if hasattr(self, "rbr_reparam"):
            raise NotImplementedError("yolov6_amd: a deployed QARepVGG block cannot be trained (no branches left)")
        s = self.rbr_dense.conv.stride[0]
        y3 = pb.conv(x, self.rbr_dense.conv.weight, s)
        y1 = pb.conv(x, self.rbr_1x1.weight, s)
        branches = [(y3, pb.bn(y3, self.rbr_dense.bn)), (y1, None)]
        if self.rbr_identity is not None:
            xr = pb.as_nhwc(x)
            branches.append((pb.avgpool3(xr, True) if getattr(self, "rbr_avg", None) is not None else xr, None))
        z = pb.bnact(branches, None)
        return pb.bnact([(z, pb.bn(z, self.bn))], "relu", out=out, res=res, alpha=res_alpha)

    def _post_affine(self):
        # the deploy form keeps self.bn after the conv (:338-339, :390-392)
        return bn_scale_shift(self.bn) if hasattr(self, "bn") else None


class QARepVGGBlockV2(QARepVGGBlock):
    '''QARepVGG + 3x3 average-pool branch.  Reference: common.py:396-477.'''
    _with_avg = True


# ------------------------------------------------------------------------------------------
# stage blocks
# ------------------------------------------------------------------------------------------
class BottleRep(HipModule):
    '''Two basic blocks with an optional (learnably weighted) shortcut.  Reference: common.py:590-608.
    The shortcut `out + alpha*x` is the residual term of the second conv's epilogue.'''

    def __init__(self, in_channels, out_channels, basic_block=RepVGGBlock, weight=False):
        super().__init__()
        self.conv1 = basic_block(in_channels, out_channels)
        self.conv2 = basic_block(out_channels, out_channels)
        self.shortcut = in_channels == out_channels
        self.alpha = nn.Parameter(torch.ones(1)) if weight else 1.0

    def lower(self, pb, x, out=None):
        x = pb.as_nhwc(x)
        t = self.conv1.lower(pb, x)
        if not self.shortcut:
            return self.conv2.lower(pb, t, out=out)
        alpha = self.alpha if isinstance(self.alpha, torch.Tensor) else None
        return self.conv2.lower(pb, t, out=out, res=x, res_alpha=alpha)


class RepBlock(HipModule):
    '''Stage = conv1 + (n-1) more blocks (BottleRep stages halve n).  Reference: common.py:569-587.'''

    def __init__(self, in_channels, out_channels, n=1, block=RepVGGBlock, basic_block=RepVGGBlock):
        super().__init__()
        if block == BottleRep:
            make = lambda i, o: BottleRep(i, o, basic_block=basic_block, weight=True)
            n = n // 2
        else:
            make = lambda i, o: block(i, o)
        self.conv1 = make(in_channels, out_channels)
        self.block = nn.Sequential(*(make(out_channels, out_channels) for _ in range(n - 1))) if n > 1 else None

    def lower(self, pb, x, out=None):
        rest = list(self.block) if self.block is not None else []
        x = self.conv1.lower(pb, x, out=out if not rest else None)
        for i, m in enumerate(rest):
            x = m.lower(pb, x, out=out if i == len(rest) - 1 else None)
        return x


class BepC3(HipModule):
    '''CSPStackRep block: cv3(cat(m(cv1(x)), cv2(x))).  Reference: common.py:634-650.'''

    def __init__(self, in_channels, out_channels, n=1, e=0.5, block=RepVGGBlock):
        super().__init__()
        hidden = int(out_channels * e)
        conv = ConvBNSiLU if block == ConvBNSiLU else ConvBNReLU
        self.cv1 = conv(in_channels, hidden, 1, 1)
        self.cv2 = conv(in_channels, hidden, 1, 1)
        self.cv3 = conv(2 * hidden, out_channels, 1, 1)
        self.m = RepBlock(in_channels=hidden, out_channels=hidden, n=n, block=BottleRep, basic_block=block)

    def lower(self, pb, x, out=None):
        x = pb.as_nhwc(x)
        hidden = self.cv1.block.conv.out_channels
        cat = pb.new_buffer(x.B, x.H, x.W, 2 * hidden)
        t = self.cv1.lower(pb, x)
        self.m.lower(pb, t, out=cat.slice(0, hidden))
        self.cv2.lower(pb, x, out=cat.slice(hidden, hidden))
        return self.cv3.lower(pb, cat, out=out)


class BottleRep3(HipModule):
    '''Three basic blocks + weighted shortcut (the unit of MBLABlock).  Reference: common.py:611-632.'''

    def __init__(self, in_channels, out_channels, basic_block=RepVGGBlock, weight=False):
        super().__init__()
        self.conv1 = basic_block(in_channels, out_channels)
        self.conv2 = basic_block(out_channels, out_channels)
        self.conv3 = basic_block(out_channels, out_channels)
        self.shortcut = in_channels == out_channels
        self.alpha = nn.Parameter(torch.ones(1)) if weight else 1.0

    def lower(self, pb, x, out=None):
        x = pb.as_nhwc(x)
        t = self.conv2.lower(pb, self.conv1.lower(pb, x))
        if not self.shortcut:
            return self.conv3.lower(pb, t, out=out)
        alpha = self.alpha if isinstance(self.alpha, torch.Tensor) else None
        return self.conv3.lower(pb, t, out=out, res=x, res_alpha=alpha)


class MBLABlock(HipModule):
    '''Multi Branch Layer Aggregation block of the *_mbla models: cv1 (1x1) splits into 2-3 branches, branch b > 0 runs a
    chain of BottleRep3 whose EVERY intermediate result joins the concatenation, cv2 (1x1) merges.  Reference:
    common.py:653-692.  Lowering is concat-free: cv1 is issued as one 1x1 conv per branch (its output channels are
    independent), each writing its slot of the buffer cv2 reads; every BottleRep3 writes the next slot.'''

    def __init__(self, in_channels, out_channels, n=1, e=0.5, block=RepVGGBlock):
        super().__init__()
        n = n // 2
        if n <= 0:
            n = 1
        if n == 1:                      # at most one extra branch
            n_list = [0, 1]
        else:
            extra_branch_steps = 1
            while extra_branch_steps * 2 < n:
                extra_branch_steps *= 2
            n_list = [0, extra_branch_steps, n]
        branch_num = len(n_list)
        self.c = int(out_channels * e)
        act = "silu" if block == ConvBNSiLU else "relu"
        self.cv1 = ConvModule(in_channels, branch_num * self.c, 1, 1, act, bias=False)
        self.cv2 = ConvModule((sum(n_list) + branch_num) * self.c, out_channels, 1, 1, act, bias=False)
        self.m = nn.ModuleList()
        for n_i in n_list[1:]:
            self.m.append(nn.Sequential(*(BottleRep3(self.c, self.c, basic_block=block, weight=True) for _ in range(n_i))))
        self.split_num = tuple([self.c] * branch_num)

    def lower(self, pb, x, out=None):
        x = pb.as_nhwc(x)
        c = self.c
        if c % 8:
            raise NotImplementedError(f"yolov6_amd: MBLABlock branches of {c} channels (channel slices need a multiple of 8)")
        total = self.cv2.conv.in_channels
        cat = pb.new_buffer(x.B, x.H, x.W, total)
        if getattr(pb, "is_train", False):
            # training form (common.py:45-49 for cv1 / cv2): cv1 is ONE conv with ONE BatchNorm over its branch_num * c
            # channels; BatchNorm + activation are per channel, so each branch's slice of the normalised result is written
            # straight into its slot of the buffer cv2 reads (statistics and d-gamma / d-beta as channel slices)
            if not hasattr(self.cv1, "bn") or self.cv1.conv.bias is not None:
                raise NotImplementedError("yolov6_amd: training needs the un-fused ConvModule (conv without bias + BatchNorm)")
            y = pb.conv(x, self.cv1.conv.weight, 1)
            st = pb.bn(y, self.cv1.bn)
            act = self.cv1._activation_name()
            pb.bnact([(y.slice(0, c), st.slice(0, c))], act, out=cat.slice(0, c))
            off = c
            for bi, seq in enumerate(self.m):
                prev = pb.bnact([(y.slice((bi + 1) * c, c), st.slice((bi + 1) * c, c))], act, out=cat.slice(off, c))
                off += c
                for blk in seq:
                    prev = blk.lower(pb, prev, out=cat.slice(off, c))
                    off += c
            assert off == total
            return self.cv2.lower(pb, cat, out=out)
        w, b = self.cv1.fused_weight_bias()
        act = self.cv1._activation_name()

        def branch(bi, off):            # y[bi] of the reference's split: rows bi*c .. (bi+1)*c of cv1
            bb = None if b is None else b[bi * c:(bi + 1) * c]
            return pb.conv(x, w[bi * c:(bi + 1) * c], bb, stride=1, act=act, out=cat.slice(off, c))

        branch(0, 0)
        off = c
        for bi, seq in enumerate(self.m):
            prev = branch(bi + 1, off)
            off += c
            for blk in seq:
                prev = blk.lower(pb, prev, out=cat.slice(off, c))
                off += c
        assert off == total
        return self.cv2.lower(pb, cat, out=out)


class BiFusion(HipModule):
    '''cv3(cat(upsample(x0), cv1(x1), downsample(cv2(x2)))).  Reference: common.py:695-718.'''

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.cv1 = ConvBNReLU(in_channels[0], out_channels, 1, 1)
        self.cv2 = ConvBNReLU(in_channels[1], out_channels, 1, 1)
        self.cv3 = ConvBNReLU(out_channels * 3, out_channels, 1, 1)
        self.upsample = Transpose(in_channels=out_channels, out_channels=out_channels)
        self.downsample = ConvBNReLU(in_channels=out_channels, out_channels=out_channels, kernel_size=3, stride=2)

    def lower(self, pb, x, out=None):
        x0, x1, x2 = (pb.as_nhwc(t) for t in x)
        oc = self.cv1.block.conv.out_channels
        cat = pb.new_buffer(x1.B, x1.H, x1.W, 3 * oc)
        self.upsample.lower(pb, x0, out=cat.slice(0, oc))
        self.cv1.lower(pb, x1, out=cat.slice(oc, oc))
        hint = getattr(pb, "hint_single_use", None)
        if hint is not None:
            hint()            # cv2's output feeds the downsample conv only: the pair may run as one fused kernel (csrc/conv_fused.hip)
        t = self.cv2.lower(pb, x2)
        self.downsample.lower(pb, t, out=cat.slice(2 * oc, oc))
        return self.cv3.lower(pb, cat, out=out)


class DetectBackend(nn.Module):
    '''Checkpoint-backed inference wrapper used by core/inferer.py.  Reference: common.py:551-567.'''

    def __init__(self, weights='yolov6s.pt', device=None, dnn=True):
        super().__init__()
        import os
        from pathlib import Path
        assert isinstance(weights, str) and Path(weights).suffix == '.pt', f'{Path(weights).suffix} format is not supported.'
        if not os.path.exists(weights):
            raise FileNotFoundError(f"yolov6_amd: checkpoint {weights} not found (no network: download it beforehand)")
        from ..utils.checkpoint import load_checkpoint
        model = load_checkpoint(weights, map_location=device)
        stride = int(model.stride.max())
        self.__dict__.update(locals())  # assign all variables to self (the reference does the same)

    def forward(self, im, val=False):
        y, _ = self.model(im)
        return y


def get_block(mode):
    '''Reference: common.py:721-737 (hyper_search / repopt blocks are outside the hot path).'''
    table = {"repvgg": RepVGGBlock, "qarepvgg": QARepVGGBlock, "qarepvggv2": QARepVGGBlockV2,
             "conv_relu": ConvBNReLU, "conv_silu": ConvBNSiLU}
    if mode not in table:
        raise NotImplementedError("Undefied Repblock choice for mode {}".format(mode))
    return table[mode]
