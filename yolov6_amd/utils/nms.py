"""non_max_suppression with the reference's signature and return convention
(yolov6/utils/nms.py:31-105), running as two HIP kernels (yolov6_amd/csrc/nms.hip).

Deviation (documented in DESIGN.md): the reference's 10 s wall-clock `time_limit` break
(nms.py:56, :101-103) has no analogue - all images are processed in one launch.
"""
import ctypes as C
import os
import weakref

import torch

from .. import _lib

_ws_cache = {}

# Results of `Model.forward` (eval) that non_max_suppression may still find their candidates for: id(result tensor) ->
# (weakref(result), weakref(plan), the plan's run number, the tensor's version counter, the stream the plan ran on).
# The plan's fused head tail can select the NMS candidates while the rows sit in LDS (engine.Plan.attach_nms) - but only for
# thresholds it knows when the forward runs, and the reference's API passes them afterwards (evaler.py:128-132, inferer.py:61-63:
# `non_max_suppression(model(x)[0], conf, iou, ...)`).  So the drop-in SPECULATES: the first call on a model's result arms the
# plan with that call's thresholds; later forwards select with them, and a later call takes the selected candidates iff
#   * `prediction` IS the tensor object that forward returned (not a view, clone or a new tensor at the same address),
#   * nothing wrote it in place since (version counter), the plan has not run again (its workspace holds ONE run's candidates),
#   * thresholds / classes / multi_label are the ones the plan was armed with, same stream.
# Anything else takes the full path (and re-arms); a plan whose callers keep changing thresholds is left alone after a few flips.
# Same detections either way (tests/test_gpu_dropin.py); Y6_DROPIN_SINK=0 switches the speculation off.
_PRODUCED = {}
_SINK_ON = os.environ.get("Y6_DROPIN_SINK", "1") != "0"
_MAX_FLIPS = 4


def note_model_output(det, plan):
    """Called by models.yolo.Model.forward with the tensor it is about to return."""
    if not _SINK_ON:
        return
    # (engine.Plan.run / run_range count the plan's launches in plan._run_seq: a launch made behind Model.forward's back - the
    # plan API on the same plan - also invalidates this entry)
    if len(_PRODUCED) > 32:
        for k in [k for k, v in _PRODUCED.items() if v[0]() is None or v[1]() is None]:
            del _PRODUCED[k]
        if len(_PRODUCED) > 32:
            _PRODUCED.clear()
    _PRODUCED[id(det)] = (weakref.ref(det), weakref.ref(plan), getattr(plan, "_run_seq", 0), det._version,
                          torch.cuda.current_stream(det.device).cuda_stream)


def _speculated_candidates(prediction, conf_thres, classes, multi_label):
    ent = _PRODUCED.pop(id(prediction), None) if _SINK_ON else None
    if ent is None or ent[0]() is not prediction:
        return None
    plan = ent[1]()
    if plan is None or prediction.dim() != 3 or prediction.dtype != torch.float32:
        return None
    ml = bool(multi_label) and prediction.shape[2] - 5 > 1
    want = (float(conf_thres), ml, None if classes is None else tuple(int(c) for c in classes), tuple(prediction.shape))
    tok = getattr(plan, "_nms_token", None)
    if tok is not None and (tok["conf_thres"], tok["multi_label"], tok["classes"], tuple(tok["shape"])) == want:
        fresh = (ent[2] == getattr(plan, "_run_seq", -1) and ent[3] == prediction._version
                 and ent[4] == torch.cuda.current_stream(prediction.device).cuda_stream and tok.get("armed_before_run", 0) < ent[2])
        return tok if fresh else None
    flips = getattr(plan, "_sink_flips", 0)
    if flips < _MAX_FLIPS:                   # arm (or re-arm) the plan: its NEXT forward selects with these thresholds
        plan._sink_flips = flips + 1
        t = plan.attach_nms(conf_thres, classes, multi_label)
        if t is not None:
            t["armed_before_run"] = getattr(plan, "_run_seq", 0)      # runs up to this number did not select with them
    elif tok is not None:
        plan.attach_nms(None)
    return None


def _workspace(device, nbytes):
    # one workspace per (device, stream): calls issued on different streams must not share keys/counts buffers
    key = (device.index, torch.cuda.current_stream(device).cuda_stream)
    ws = _ws_cache.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(nbytes, dtype=torch.uint8, device=device)
        _ws_cache[key] = ws
    return ws


def nms_raw(prediction, conf_thres=0.25, iou_thres=0.45, classes=None, agnostic=False, multi_label=False, max_det=300,
            candidates=None):
    """Device-side result without host synchronisation:
    (dets [B,max_det,6] f32, index [B,max_det] i32 (anchor*nc+cls), count [B] i32).
    candidates: the token of engine.Plan.attach_nms() when `prediction` is that plan's output of its LAST run on this stream -
    the candidates were selected by the decode launch, with the same thresholds (checked here)."""
    lib = _lib.load()
    _lib.require_gpu_tensor(prediction, "prediction")
    # the reference asserts the thresholds (nms.py:50-51)
    assert 0 <= conf_thres <= 1, f'conf_thresh must be in 0.0 to 1.0, however {conf_thres} is provided.'
    assert 0 <= iou_thres <= 1, f'iou_thres must be in 0.0 to 1.0, however {iou_thres} is provided.'
    pred = prediction
    if pred.dtype != torch.float32:
        pred = pred.float()          # the head emits fp32 even for .half() models
    pred = pred.contiguous()
    B, A, no = pred.shape
    nc = no - 5
    dev = pred.device
    ml = bool(multi_label) and nc > 1
    # no fills: y6_nms writes every element (rows past the kept count become 0 / -1)
    dets = torch.empty((B, max_det, 6), dtype=torch.float32, device=dev)
    index = torch.empty((B, max_det), dtype=torch.int32, device=dev)
    count = torch.empty((B,), dtype=torch.int32, device=dev)
    nbytes = lib.y6_nms_workspace_bytes(B, A, nc, int(ml))
    cls_t = None
    if candidates is not None:
        want = (float(conf_thres), ml, None if classes is None else tuple(int(c) for c in classes), (B, A, no))
        have = (candidates["conf_thres"], candidates["multi_label"], candidates["classes"], candidates["shape"])
        if want != have:
            raise RuntimeError(f"yolov6_amd: nms_raw(candidates=...) with other thresholds / shape than attach_nms(): {want} vs {have}")
        ws, cls_t = candidates["workspace"], candidates["classes_t"]
    else:
        ws = _workspace(dev, nbytes)
        if classes is not None:
            cls_t = torch.as_tensor(list(classes), dtype=torch.int32, device=dev)
    d = _lib.NmsDesc()
    d.pred = C.c_void_p(pred.data_ptr())
    d.B, d.A, d.nc = B, A, nc
    d.conf_thres, d.iou_thres = float(conf_thres), float(iou_thres)
    d.classes = C.c_void_p(cls_t.data_ptr()) if cls_t is not None else None
    d.n_classes = int(cls_t.numel()) if cls_t is not None else 0
    d.agnostic, d.multi_label = int(bool(agnostic)), int(ml)
    d.max_det, d.max_nms, d.max_wh = int(max_det), 30000, 4096.0
    d.out_dets, d.out_index, d.out_count = (C.c_void_p(t.data_ptr()) for t in (dets, index, count))
    d.workspace, d.workspace_bytes = C.c_void_p(ws.data_ptr()), ws.numel()
    d.candidates_ready = 1 if candidates is not None else 0
    _lib.check(lib.y6_nms(C.byref(d), _lib.current_stream_ptr()), "nms")
    return dets, index, count


def non_max_suppression(prediction, conf_thres=0.25, iou_thres=0.45, classes=None, agnostic=False, multi_label=False,
                        max_det=300):
    """Runs Non-Maximum Suppression on inference results.

    Args and return value as the reference: a list with one [n, 6] tensor (xyxy, conf, cls)
    per image, boxes in descending confidence order, at most `max_det` rows.
    """
    cand = _speculated_candidates(prediction, conf_thres, classes, multi_label)
    dets, _, count = nms_raw(prediction, conf_thres, iou_thres, classes, agnostic, multi_label, max_det, candidates=cand)
    counts = count.tolist()          # the one host sync; the reference syncs once per image
    # one dispatcher call for all per-image views (rows [i*max_det, i*max_det + n_i) of the flat result): the GPU idles while the
    # host builds this list, and B separate slicing calls cost twice as much (139 -> 74 us at B = 32)
    B, md = dets.shape[0], dets.shape[1]
    sizes = [0] * (2 * B)
    sizes[0::2] = counts
    sizes[1::2] = [md - n for n in counts]
    return list(dets.view(B * md, dets.shape[2]).split_with_sizes(sizes)[0::2])


def xywh2xyxy(x):
    '''[n,4] (cx, cy, w, h) -> (x1, y1, x2, y2).  Reference: nms.py:21-28 (host helper).'''
    y = x.clone() if isinstance(x, torch.Tensor) else x.copy()
    y[:, 0] = x[:, 0] - x[:, 2] / 2
    y[:, 1] = x[:, 1] - x[:, 3] / 2
    y[:, 2] = x[:, 0] + x[:, 2] / 2
    y[:, 3] = x[:, 1] + x[:, 3] / 2
    return y
