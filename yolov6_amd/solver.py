"""Optimizer side of the training step on the parameter arena (yolov6_amd.train_engine.ParamArena).

    FusedSGD      torch.optim.SGD as yolov6/solver/build.py:10-30 configures it (Nesterov momentum; three parameter groups:
                  BatchNorm weights, conv weights with weight decay, biases), ONE kernel over the flat arena instead of
                  ~250 per-tensor updates, with the GradScaler's unscale + inf check + skip fused in
                  (core/engine.py:258-266 scaler.step / scaler.update)
    LossScaler    the dynamic loss scale of torch.cuda.amp.GradScaler kept ON DEVICE (no host sync per step)
    ArenaEMA      ModelEMA.update (yolov6/utils/ema.py:27-37) as one lerp over the arena + the BatchNorm buffers
The reference's own `torch.optim.SGD` / `GradScaler` also work on the arena-backed parameters (their `.grad`s are arena
views); these classes are the fast path the benchmark uses.
"""
import ctypes as C
import math

import torch
import torch.nn as nn

from . import _lib
from .layers.common import bump_native_generation


def param_groups(model):
    """(bn_weights, weights, biases) exactly as yolov6/solver/build.py:12-21 collects them."""
    g_bnw, g_w, g_b = [], [], []
    for v in model.modules():
        if hasattr(v, 'bias') and isinstance(v.bias, nn.Parameter):
            g_b.append(v.bias)
        if isinstance(v, nn.BatchNorm2d):
            g_bnw.append(v.weight)
        elif hasattr(v, 'weight') and isinstance(v.weight, nn.Parameter):
            g_w.append(v.weight)
    return g_bnw, g_w, g_b


class LossScaler:
    """Dynamic loss scaling with the GradScaler defaults (init 65536, growth 2 every 2000 clean steps, backoff 0.5)."""

    def __init__(self, device, init_scale=65536.0, growth_factor=2.0, backoff_factor=0.5, growth_interval=2000, enabled=True):
        self.scale = torch.tensor([init_scale if enabled else 1.0], dtype=torch.float32, device=device)
        self.found_inf = torch.zeros(1, dtype=torch.int32, device=device)
        self.tracker = torch.zeros(1, dtype=torch.int32, device=device)
        self.growth, self.backoff, self.interval, self.enabled = growth_factor, backoff_factor, growth_interval, enabled

    def scale_loss(self, loss):
        return loss * self.scale[0]

    def update(self):
        if self.enabled:
            _lib.check(_lib.load().y6_scaler_update(C.c_void_p(self.scale.data_ptr()), C.c_void_p(self.found_inf.data_ptr()),
                                                    C.c_void_p(self.tracker.data_ptr()), self.growth, self.backoff, self.interval,
                                                    _lib.current_stream_ptr()), "scaler_update")


class FusedSGD:
    def __init__(self, model, arena, lr=0.01, momentum=0.937, weight_decay=5e-4, nesterov=True):
        self.arena = arena
        g_bnw, g_w, g_b = param_groups(model)
        group = torch.full((arena.numel,), 3, dtype=torch.uint8)
        for k, ps in enumerate((g_bnw, g_w, g_b)):
            for p in ps:
                if id(p) in arena.index:
                    o = arena.offset_of(p)
                    group[o:o + p.numel()] = k
        self.group = group.to(arena.data.device)
        self.buf = torch.zeros_like(arena.data)
        self.lr = [lr, lr, lr]                      # the warm-up moves the bias group's rate separately (engine.py:433-441)
        self.wd = [0.0, weight_decay, 0.0]
        self.momentum, self.nesterov = momentum, nesterov
        self.steps = 0

    def zero_grad(self):
        self.arena.zero_grad()

    def step(self, scaler: LossScaler = None, grad_mul: float = 1.0):
        lib = _lib.load()
        a = self.arena
        s = _lib.current_stream_ptr()
        scale = found = None
        if scaler is not None and scaler.enabled:
            scale, found = C.c_void_p(scaler.scale.data_ptr()), C.c_void_p(scaler.found_inf.data_ptr())
            _lib.check(lib.y6_grad_finite_check(C.c_void_p(a.grad.data_ptr()), a.numel, found, s), "grad_finite_check")
        lr3, wd3 = (C.c_float * 3)(*self.lr), (C.c_float * 3)(*self.wd)
        _lib.check(lib.y6_sgd_step_grouped(C.c_void_p(a.data.data_ptr()), C.c_void_p(a.grad.data_ptr()), C.c_void_p(self.buf.data_ptr()),
                                           C.c_void_p(self.group.data_ptr()), a.numel, lr3, wd3, self.momentum, int(self.nesterov),
                                           int(self.steps == 0), float(grad_mul), scale, found, s), "sgd_step_grouped")
        self.steps += 1                              # (a skipped first step leaves a zero buffer: mom*0 + d == d)
        bump_native_generation()                     # parameters changed behind autograd's version counters: cached eval plans are stale


class ArenaEMA:
    """EMA of every parameter (one lerp over the arena) and every floating-point buffer (ModelEMA.update, utils/ema.py:27-37).
    `copy_to(model)` writes the averages into a module of the same architecture - the reference evaluates and checkpoints
    `ema.ema` (core/engine.py:192-203); `ema_module(model)` returns such a module (a deep copy of `model`)."""

    def __init__(self, model, arena, decay=0.9999, updates=0):
        self.arena = arena
        self.ema = arena.data.clone()
        self.buffers = [b for b in model.buffers() if b.dtype.is_floating_point]
        self.ema_buffers = [b.detach().clone() for b in self.buffers]
        self.updates = updates
        self.decay = lambda x: decay * (1 - math.exp(-x / 2000))

    @torch.no_grad()
    def update(self):
        self.updates += 1
        d = self.decay(self.updates)
        self.ema.lerp_(self.arena.data, 1.0 - d)
        if self.buffers:
            torch._foreach_lerp_(self.ema_buffers, self.buffers, 1.0 - d)

    @torch.no_grad()
    def copy_to(self, model):
        """Write the averaged parameters / buffers into `model` (same architecture as the trained one; parameters are matched
        by registration order, which is how the arena was laid out)."""
        a = self.arena
        params = [p for p in model.parameters() if p.requires_grad]
        seen, uniq = set(), []
        for p in reversed(params):
            if id(p) not in seen:
                seen.add(id(p))
                uniq.append(p)
        if [tuple(p.shape) for p in uniq] != [tuple(p.shape) for p in a.params]:
            raise RuntimeError("yolov6_amd: ArenaEMA.copy_to needs a module with the trained model's parameters")
        for p, o in zip(uniq, a.offsets):
            p.data.copy_(self.ema[o:o + p.numel()].view(p.shape))
        bufs = [b for b in model.buffers() if b.dtype.is_floating_point]
        if len(bufs) != len(self.ema_buffers):
            raise RuntimeError("yolov6_amd: ArenaEMA.copy_to needs a module with the trained model's buffers")
        for b, e in zip(bufs, self.ema_buffers):
            b.copy_(e)
        if hasattr(model, "invalidate_plans"):
            model.invalidate_plans()
        return model

    def ema_module(self, model):
        """A deep copy of `model` (plain tensors, no arena views, no native state) holding the averages: what the reference
        keeps as `ema.ema` for evaluation and for `save_checkpoint`."""
        import copy
        return self.copy_to(copy.deepcopy(model))
