"""Throughput inference with N batches in flight.

A forward pass is ~60 dependent kernel launches, and every launch has ramps - the first fetch of every block at once, the store
burst at the end, the tail while the last blocks finish - during which most of the chip idles (~10 us of a 26 us
small-map launch).  Consecutive layers cannot overlap them (data dependence); consecutive BATCHES can: `InflightRunner` keeps N
plans of the same module (own activation buffers, own packed copies of the weights and own NMS workspace, all lowered from the
same parameters: `HipModule.new_plan`) on N HIP streams and hands batch i to plan
i % N.  Measured on one MI355X, YOLOv6-S 640^2 b32 fp16 forward + NMS: 13 961 -> 15 730 img/s with N = 2 (bench.py, r04y).

The reference's eval loop (yolov6/core/evaler.py:98-120: `outputs = model(imgs)`; `non_max_suppression(outputs, ...)`; convert to
COCO json) is one batch at a time because the host consumes every result before the next forward; a caller that can defer the
host side by one batch gets the overlap:

    run = InflightRunner(model, example_batch, depth=2, conf_thres=0.03, iou_thres=0.65, multi_label=True)
    pending = None
    for imgs in loader:
        ticket = run.submit(imgs)            # enqueues forward + NMS of this batch, returns at once
        if pending is not None:
            dets, counts = pending.result()  # waits for the PREVIOUS batch only
            ...
        pending = ticket
"""
import time

import torch

from .utils.nms import nms_raw


class Ticket:
    def __init__(self, det, dets, index, count, event, runner=None, slot=0, generation=0):
        self.det, self.dets, self.index, self.count, self._event = det, dets, index, count, event
        self._runner, self._slot, self._generation = runner, slot, generation

    def result(self, copy=True):
        """Per-image detection lists as `non_max_suppression` returns them (one host synchronisation on this batch's event).

        The tensors of a slot (raw head output, NMS result) are reused by the submission `depth` later: a ticket whose slot has
        been handed to a later batch raises instead of returning that batch's rows.  `copy=True` (default) returns independent
        tensors, as the reference's `non_max_suppression` does; `copy=False` returns views into the slot's result buffer, valid
        until that slot is submitted to again."""
        r = self._runner
        if r is not None and r.generation[self._slot] != self._generation:
            raise RuntimeError("yolov6_amd.pipeline: this ticket's slot was reused by a later submit() before result() was called "
                               f"(depth {len(r.plans)}): consume a ticket within `depth` submissions")
        self._event.synchronize()
        counts = self.count.tolist()
        rows = [self.dets[i, :n] for i, n in enumerate(counts)]
        return ([t.clone() for t in rows] if copy else rows), counts


class InflightRunner:
    def __init__(self, model, example, depth=2, conf_thres=0.25, iou_thres=0.45, classes=None, agnostic=False, multi_label=False,
                 max_det=300, autotune=True):
        # (autotune=True: timed kernels, replayed from the on-disk table where one exists; Y6_AUTOTUNE=0 forces shape-derived kernels)
        if depth < 1:
            raise ValueError("yolov6_amd.pipeline: depth >= 1")
        self.kw = dict(conf_thres=conf_thres, iou_thres=iou_thres, classes=classes, agnostic=agnostic, multi_label=multi_label,
                       max_det=max_det)
        # ONE module, `depth` plans lowered from it (the extra ones outside the module's plan cache, never shared with
        # `model.forward`): the same parameters, BatchNorm buffers and int8 calibration behind every slot.  (Round 4 deep-copied
        # the module: N copies of the weights, and `HipModule.__getstate__` dropped the int8 state, so odd batches of a quantised
        # model ran the fp16 plan - ADVICE r4.)
        self.model = model
        self.autotune = autotune
        self.plans = [model.new_plan(example, autotune=autotune)]
        self.plans += [model.new_plan(example, autotune=autotune, variants_from=self.plans[0]) for _ in range(depth - 1)]   # tuned once
        # what the slots' packed weights were derived from (submit() re-checks, cheap things first): the autograd version counters
        # of the tensors, WHICH tensor objects are registered where (a replaced Parameter is another object; the old one, kept alive
        # here, would never change), the process-wide generations that native kernels (fused SGD, running-statistics update, EMA)
        # and .half() / .to() / invalidate_plans() bump instead of `_version`, and the int8 state
        from .layers import common as _c
        self._gens = _c
        self._tensors = list(model.parameters()) + list(model.buffers())
        self._holders = [(d, n, t) for m in model.modules() for d in (m._parameters, m._buffers) for n, t in d.items() if t is not None]
        self._vsum = sum(t._version for t in self._tensors)
        self._generations = (_c._NATIVE_GENERATION[0], _c._STRUCTURE_GENERATION[0])
        self._quant = model.__dict__.get("_y6_quant")
        self.inputs = [example] + [torch.empty_like(example) for _ in range(depth - 1)]
        for p, x in zip(self.plans[1:], self.inputs[1:]):
            p.bind_inputs([x])
        self.tokens = [p.attach_nms(conf_thres, classes, multi_label) for p in self.plans]
        self.streams = pick_streams(depth, self._trial)
        self.done = [None] * depth
        self.generation = [0] * depth       # per slot: how many batches it has been handed (Ticket.result checks it)
        self.i = 0

    def _trial(self, streams):
        torch.cuda.synchronize()
        t = time.perf_counter()
        for i in range(12):
            j = i % len(streams)
            with torch.cuda.stream(streams[j]):
                det = self.plans[j].run()
                nms_raw(det, candidates=self.tokens[j], **self.kw)
        torch.cuda.synchronize()
        return time.perf_counter() - t

    def submit(self, x: torch.Tensor) -> Ticket:
        """Enqueue forward + NMS of one batch (same shape / dtype as the example) and return at once.  The slot's tensors are
        overwritten: a ticket of this slot that has not been consumed yet expires (its `result()` raises)."""
        if x.shape != self.inputs[0].shape or x.dtype != self.inputs[0].dtype:
            raise RuntimeError(f"yolov6_amd.pipeline: batch {tuple(x.shape)} {x.dtype} does not match the example "
                               f"{tuple(self.inputs[0].shape)} {self.inputs[0].dtype} the plans were built for")
        j = self.i % len(self.plans)
        # (cheap first: the sum of the autograd version counters of the tensors the plans were lowered from and the identity of
        # the int8 state; the full key - a walk over every module - only when those moved)
        c = self._gens
        if ((sum(t._version for t in self._tensors) != self._vsum or self.model.__dict__.get("_y6_quant") is not self._quant
             or (c._NATIVE_GENERATION[0], c._STRUCTURE_GENERATION[0]) != self._generations
             or not all(d.get(n) is t for d, n, t in self._holders))
                and not self.model.plan_is_current(self.plans[j])):
            # load_state_dict / an optimizer step (torch's or the native fused SGD) / a replaced Parameter / .half() / quantize()
            # since the plans were built: every slot would serve stale weights
            raise RuntimeError("yolov6_amd.pipeline: the model's parameters (or its int8 state) changed after this runner was built; "
                               "build a new InflightRunner")
        self.i += 1
        self.generation[j] += 1
        st = self.streams[j]
        st.wait_stream(torch.cuda.current_stream())          # x was produced on the caller's stream
        with torch.cuda.stream(st):
            if x.data_ptr() != self.inputs[j].data_ptr():
                x.record_stream(st)                          # the caching allocator must not recycle x while the copy is pending
                self.inputs[j].copy_(x, non_blocking=True)
            det = self.plans[j].run()
            dets, index, count = nms_raw(det, candidates=self.tokens[j], **self.kw)
            ev = torch.cuda.Event()
            ev.record(st)
        self.done[j] = ev
        return Ticket(det, dets, index, count, ev, self, j, self.generation[j])


def pick_streams(n, trial):
    """n HIP streams that really run side by side.  HIP streams share a few hardware queues, and two streams on one queue do not
    overlap: candidates are timed (trial(streams) -> seconds for a few steps round-robin over them); the first is kept, each
    further one is the best partner found."""
    pool = [torch.cuda.Stream() for _ in range(max(n, 6 if n > 1 else 1))]
    if n == 1:
        return pool[:1]
    chosen = [pool[0]]
    for _ in range(1, n):
        rest = [q for q in pool if q not in chosen]
        for q in rest:
            trial(chosen + [q])                              # warm
        chosen.append(min(rest, key=lambda q: min(trial(chosen + [q]) for _ in range(2))))
    return chosen
