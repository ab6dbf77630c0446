"""Two-stream schedule of an inference plan (eager whole-plan runs, csrc/plan.hip y6_plan_set_schedule).

A YOLOv6 forward is one long dependent chain (backbone -> SPPF -> top-down neck -> bottom-up neck -> last head level) with a
few branches hanging off it: BiFusion's lateral `cv1` / `cv2 -> downsample` convs read backbone maps that have been ready
since the middle of the backbone (reference yolov6/layers/common.py:695-718, yolov6/models/reppan.py:215-237), the CSP-SPPF
bypass conv (common.py:135-158), and the stem / cls+reg convs of the head levels whose neck output is finished long before
the last one (yolov6/models/effidehead.py:93-139).  On one stream those branches run where the module tree happens to emit
them - many of them in the 20x20 / 40x40 stretches whose kernels leave most of the 256 CUs idle.  Here the ops OFF the
critical path go to a second HIP stream, each as early as its inputs allow, ordered by events against the chain.

Everything in this file is host logic on (reads, writes) spans: testable without a GPU (tests/test_host_cpu.py runs it
over the lowering graph of every model family and checks that the schedule enforces every data dependence).
"""
from typing import Dict, List, Optional, Sequence, Tuple

Span = Tuple[int, int, int, int]     # (first byte, one past the last byte of the allocation piece, first channel, one past the last)
WHOLE = 1 << 30


def _bytes_of(t) -> Tuple[int, int]:
    lo = int(t.data_ptr())
    return lo, lo + max(int(t.numel()) * int(t.element_size()), 1)


def _span(v) -> Optional[Span]:
    if v is None:
        return None
    if hasattr(v, "buf") and hasattr(v, "coff"):           # engine.TRef: channels [coff, coff + C) of an NHWC buffer
        lo, hi = _bytes_of(v.buf)
        return (lo, hi, int(v.coff), int(v.coff) + int(v.C))
    if hasattr(v, "data_ptr"):                             # a plain tensor (or a contiguous slice of one): its bytes
        lo, hi = _bytes_of(v)
        return (lo, hi, 0, WHOLE)
    raise TypeError(f"schedule: no span for {type(v)}")


def op_access(entry: dict) -> Optional[Tuple[List[Span], List[Span]]]:
    """(reads, writes) of one op_log entry of engine.PlanBuilder; None for a kind this file does not know (no schedule then)."""
    k = entry.get("kind")
    rd, wr = [], []
    if k in ("conv", "stem", "convt", "nchw2nhwc", "nhwc2nchw", "pw_s2", "stem_s2"):
        rd.append(entry["x"])
        if entry.get("res") is not None:
            rd.append(entry["res"])
        wr.append(entry["out"])                            # (the `mid` tensor of a fused pair is never written)
    elif k == "conv_i8":                                   # int8 conv: reads the fp16 view or the producer's int8 twin of it,
        rd.append(entry["x"])                              # writes the fp16 view and / or the int8 twin for ITS consumers
        if entry.get("q_in") is not None:
            rd.append(entry["q_in"])
        if entry.get("res") is not None:
            rd.append(entry["res"])
        wr.append(entry["out"])                            # (kept even when the fp16 store is dropped: conservative)
        if entry.get("q_out") is not None:
            wr.append(entry["q_out"])
    elif k == "sppf":
        rd.append(entry["x"])
        wr += list(entry["outs"])
        if entry.get("q_outs"):                            # int8 plans: the pools leave the int8 twins of their slices too
            wr += list(entry["q_outs"])
    elif k == "decode":
        rd += list(entry["cls"]) + list(entry["reg"])
        wr.append(entry["out"])
    elif k == "pred_decode":
        rd += list(entry["cls_feat"]) + list(entry["reg_feat"])
        wr.append(entry["out"])
    else:
        return None                                        # calibration slots, training ops: not scheduled
    return [_span(v) for v in rd], [_span(v) for v in wr]


def train_op_access(entry: dict) -> Optional[Tuple[List[Span], List[Span]]]:
    """(reads, writes) of one entry of train_engine.TrainBuilder.fwd_log (the training-form forward: conv -> batch-statistics
    BatchNorm -> branch sum + activation, reference yolov6/layers/common.py:44-49, :250-255).  Parameters (views of the flat
    arena) are only read by the forward and are left out; a BatchNorm's running statistics are written by its statistics op."""
    k = entry.get("kind")
    rd, wr = [], []
    if k in ("nchw2nhwc", "subsample2", "stem", "avgpool3", "convt"):
        rd.append(entry["x"])
        wr.append(entry["out"])
    elif k == "conv":
        if entry.get("acc"):
            return None                                    # an accumulating conv reads a tensor the log does not name
        rd.append(entry["x"])
        wr.append(entry["out"])
    elif k in ("bn_train_stats", "bn_train_stats_multi"):
        items = entry["items"] if k == "bn_train_stats_multi" else [(entry["x"], entry["bn"], entry["stats"])]
        for x, bn, st in items:
            rd.append(x)
            wr += [st.scale, st.shift, st.mean, st.invstd]
            if getattr(bn, "track_running_stats", False) and getattr(bn, "running_mean", None) is not None:
                wr += [bn.running_mean, bn.running_var, bn.num_batches_tracked]
    elif k == "bnact_forward":
        for t, st in entry["branches"]:
            rd.append(t)
            if st is not None:
                rd += [st.scale, st.shift]
        if entry.get("res") is not None:
            rd.append(entry["res"])
        wr.append(entry["out"])
    elif k == "sppf":
        rd.append(entry["x"])
        wr += list(entry["outs"])
        if entry.get("q_outs"):                            # int8 plans: the pools leave the int8 twins of their slices too
            wr += list(entry["q_outs"])
    elif k in ("head_pack", "head_ab_pack"):
        rd += list(entry["cls"]) + list(entry["reg"])
        wr += [t for t in (entry.get("scores"), entry.get("distri")) if t is not None]
    else:
        return None
    return [_span(v) for v in rd], [_span(v) for v in wr]


def _grad_span(arena, p) -> Span:
    lo = int(arena.grad.data_ptr()) + 4 * int(arena.offset_of(p))
    return (lo, lo + 4 * int(p.numel()), 0, WHOLE)


def train_bwd_access(entry: dict, arena) -> Optional[Tuple[List[Span], List[Span]]]:
    """(reads, writes) of one entry of train_engine.TrainBuilder.bwd_log: gradient buffers, operand planes, workspaces and the
    slices of the flat gradient arena the op writes.  Forward activations, saved statistics and (packed) weights are read-only
    during the backward and are left out of the reads of MAIN-stream ops; for side ops (weight-gradient work) every read is
    listed - that is what a later main-stream op must not overwrite."""
    k = entry.get("kind")
    rd, wr = [], []
    if k == "conv":                                   # data gradient (dgrad / convt_dgrad): x = dy view, out = dx
        rd.append(_span(entry["x"]))
        wr.append(_span(entry["out"]))
        if entry.get("acc"):
            rd.append(_span(entry["out"]))
    elif k == "dgrad_s2":                             # both branches of a stride-2 block: compact dy views in, dx out
        rd += [_span(t) for t in entry["dys"] if t is not None]
        wr.append(_span(entry["out"]))
        if entry.get("acc"):
            rd.append(_span(entry["out"]))
    elif k == "wgrad_transpose":
        rd.append(_span(entry["src"]))
        wr.append(_span(entry["dst"]))
    elif k == "wgrad":
        if entry.get("nhwc"):
            rd += [_span(entry["x"]), _span(entry["dy"])]
        else:
            rd += [_span(entry["a"])] + [_span(t) for t in entry["planes"]]
        wr.append(_grad_span(arena, entry["weight"]))
        wr.append(_span(entry["ws"]))
    elif k == "wgrad_stem":                           # both stem convs' weight gradients from the NCHW image and the compact dy views
        rd += [_span(entry["x"])] + [_span(t) for t in entry["dys"] if t is not None]
        wr += [_grad_span(arena, w) for w in entry["weights"] if w is not None]
        wr.append(_span(entry["ws"]))
    elif k == "channel_sum":
        rd.append(_span(entry["x"]))
        wr += [_grad_span(arena, entry["param"]), _span(entry["ws"])]
    elif k == "bnact_backward":
        rd.append(_span(entry["dout"]))
        for t, dil, acc in entry["dx"]:
            wr.append(_span(t))
            if acc:
                rd.append(_span(t))
        if entry.get("dres") is not None:
            t, acc = entry["dres"]
            wr.append(_span(t))
            if acc:
                rd.append(_span(t))
        for t, st in entry["branches"]:
            if st is not None:
                bn = st.module
                for p in (bn.weight, bn.bias):
                    if p is not None:
                        wr.append(_grad_span(arena, p))      # (a channel slice of it: the whole parameter is the safe superset)
        if entry.get("alpha") is not None:
            wr.append(_grad_span(arena, entry["alpha"]))
    elif k in ("avgpool3", "tensor_add", "space_to_depth2"):
        rd.append(_span(entry["x"]))
        wr.append(_span(entry["out"]))
        if entry.get("acc"):
            rd.append(_span(entry["out"]))
    elif k in ("head_unpack_backward", "head_ab_unpack_backward"):
        rd += [_span(entry[n]) for n in ("dscores", "ddistri") if entry.get(n) is not None]
        wr += [_span(t) for t in list(entry.get("dcls", [])) + list(entry.get("dreg", []))]
    elif k == "sppf_backward":
        rd += [_span(t) for t in entry["dys"]] + [_span(entry["dx"])]
        wr.append(_span(entry["dx"]))
    else:
        return None
    return [v for v in rd if v is not None], [v for v in wr if v is not None]


def side_conflicts(log: Sequence[dict], arena, only_last: bool = False):
    """The contract of the backward plan's side stream (csrc/plan.hip y6_plan_mark_side; train_engine marks weight-gradient
    work): ops are enqueued in plan order, a side op is ordered behind every EARLIER op (fork event) and the side stream joins
    the main stream only at the end of a run / range.  So for a side op i and a LATER main-stream op j nothing orders j behind
    i: j must not write what i reads, and must not read or write what i writes.  (Side ops among themselves are one stream.)
    Returns [(i, j, what)] - empty when the contract holds.  only_last: check only the log's last op as j (build-time use)."""
    acc = [train_bwd_access(e, arena) for e in log]
    out = []
    js = [len(log) - 1] if only_last else range(len(log))
    for j in js:
        if log[j].get("side") or acc[j] is None:
            continue
        rj, wj = acc[j]
        for i in range(j):
            if not log[i].get("side") or acc[i] is None:
                continue
            ri, wi = acc[i]
            if any(_overlap(w, r) for w in wj for r in ri):
                out.append((i, j, "writes an input"))
            elif any(_overlap(w, w2) for w in wj for w2 in wi) or any(_overlap(r, w2) for r in rj for w2 in wi):
                out.append((i, j, "touches an output"))
    return out


def train_costs(log: Sequence[dict]) -> List[float]:
    """Rough per-op times (microseconds) of a training-form forward from its shapes - enough to tell the chain from the
    branches (a device profile would re-run the statistics ops, which update running statistics).  Convs at 500 TFLOP/s,
    everything at 2.5 TB/s of the bytes its views cover, 6 us per launch."""
    out = []
    for e in log:
        acc = train_op_access(e)
        by = 0.0
        if acc is not None:
            for lo, hi, c0, c1 in acc[0] + acc[1]:
                by += (hi - lo)
        fl = 0.0
        if e.get("kind") == "conv":
            o, x, k = e["out"], e["x"], int(e.get("k", 1))
            fl = 2.0 * o.B * o.H * o.W * o.C * x.C * k * k
        out.append(6.0 + by / 2.5e6 + fl / 5.0e8)
    return out


def _overlap(a: Span, b: Span) -> bool:
    if not (a[0] < b[1] and b[0] < a[1]):
        return False
    if a[0] == b[0] and a[1] == b[1]:                       # two views of one buffer: by channel slice
        return a[2] < b[3] and b[2] < a[3]
    return True                                            # different pieces that share bytes (a slice of a vector, ...)


def dependences(acc: Sequence[Tuple[List[Span], List[Span]]]) -> List[List[int]]:
    """deps[j] = earlier ops i that op j must run after (RAW, WAR, WAW on overlapping channel slices of one buffer)."""
    n = len(acc)
    deps: List[List[int]] = [[] for _ in range(n)]
    for j in range(n):
        rj, wj = acc[j]
        for i in range(j):
            ri, wi = acc[i]
            d = any(_overlap(w, r) for w in wi for r in rj) or any(_overlap(w, w2) for w in wi for w2 in wj) or \
                any(_overlap(r, w2) for r in ri for w2 in wj)
            if d:
                deps[j].append(i)
    return deps


def is_chain(deps: List[List[int]]) -> bool:
    """No op has two producers or two consumers: nothing can overlap, whatever the costs."""
    users = [0] * len(deps)
    for d in deps:
        if len(d) > 1:
            return False
        for i in d:
            users[i] += 1
    return all(u <= 1 for u in users)


def build_schedule(deps: List[List[int]], cost: Optional[Sequence[float]] = None, min_side_cost: float = 0.0,
                   policy: str = "asap", margin: float = 2.0):
    """-> (order, stream, edges) or None when nothing is off the critical path.

    policy "asap": a side op is enqueued as soon as its inputs exist (it may then run beside the large backbone convs);
    policy "alap": as late as its consumer allows - the side ops that feed main op c start where the chain still has
    `margin` x their summed cost to go before c, i.e. beside the ops that immediately precede their consumer (for YOLOv6: the
    20x20 / 40x40 stretches), never earlier than their inputs exist.

    stream[i]: 0 = the caller's stream (the critical path: the longest chain by `cost`), 1 = the side stream (everything else).
    order: enqueue order - main ops in plan order; a side op right behind the latest op (in plan order) it depends on, side ops
           among themselves by that position (then plan order), so the side stream is a FIFO of ops sorted by readiness.
    edges: (src, dst) pairs on different streams: dst waits for the event recorded behind src.  Redundant waits (already implied
           by an earlier wait of the same stream) are dropped."""
    n = len(deps)
    if n == 0:
        return None
    c = [1.0] * n if cost is None else [max(float(x), 1e-6) for x in cost]
    # longest path ending at i / best predecessor
    dist, pred = [0.0] * n, [-1] * n
    for j in range(n):
        best, bp = 0.0, -1
        for i in deps[j]:
            if dist[i] > best:
                best, bp = dist[i], i
        dist[j], pred[j] = best + c[j], bp
    end = max(range(n), key=lambda i: (dist[i], i))
    on_path = [False] * n
    i = end
    while i >= 0:
        on_path[i] = True
        i = pred[i]
    stream = [0 if on_path[i] else 1 for i in range(n)]
    # an op with neither producer nor consumer inside the plan (or a side set too cheap to bother) stays on the main stream
    users = [0] * n
    for j in range(n):
        for i in deps[j]:
            users[i] += 1
    for i in range(n):
        if stream[i] == 1 and not deps[i] and users[i] == 0:
            stream[i] = 0
    if sum(c[i] for i in range(n) if stream[i] == 1) <= min_side_cost or not any(stream):
        return None
    # readiness of a side op: plan position of the latest MAIN op it (transitively through side ops) waits for
    ready = [0] * n
    for j in range(n):
        if stream[j] == 1:
            r = -1
            for i in deps[j]:
                r = max(r, i if stream[i] == 0 else ready[i])
            ready[j] = r
    if policy == "alap":
        # first main-stream consumer of a side op (through side successors), then one start position per consumer group
        cons = [n] * n
        for j in range(n - 1, -1, -1):
            for i in deps[j]:
                if stream[i] == 1:
                    cons[i] = min(cons[i], j if stream[j] == 0 else cons[j])
        groups: Dict[int, List[int]] = {}
        for j in range(n):
            if stream[j] == 1:
                groups.setdefault(cons[j], []).append(j)
        for c_op, members in groups.items():
            need = margin * sum(c[j] for j in members)
            p, acc = c_op - 1, 0.0
            while p >= 0 and not (stream[p] == 0 and acc >= need):     # walk back over the chain ops in front of the consumer
                if stream[p] == 0:
                    acc += c[p]
                p -= 1
            for j in members:
                ready[j] = max(ready[j], p)
        for j in range(n):                                           # a side op never starts before a side op it reads from
            if stream[j] == 1:
                for i in deps[j]:
                    if stream[i] == 1:
                        ready[j] = max(ready[j], ready[i])
    elif policy != "asap":
        raise ValueError(f"schedule: unknown policy {policy!r}")
    side = sorted((j for j in range(n) if stream[j] == 1), key=lambda j: (ready[j], j))
    order: List[int] = []
    k = 0
    while k < len(side) and ready[side[k]] < 0:              # side ops that wait for nothing inside the plan: first
        order.append(side[k])
        k += 1
    for i in range(n):
        if stream[i] == 0:
            order.append(i)
            while k < len(side) and ready[side[k]] <= i:
                order.append(side[k])
                k += 1
    assert k == len(side) and len(order) == n
    # cross-stream waits, minus the ones a stream already holds: known[s] = what stream s is certain to run behind
    pos = {op: p for p, op in enumerate(order)}
    edges: List[Tuple[int, int]] = []
    behind: Dict[int, set] = {}                              # op -> every op certain to be complete when it starts
    last = {0: None, 1: None}
    for op in order:
        s = stream[op]
        known = set()
        if last[s] is not None:
            known |= behind[last[s]] | {last[s]}
        for d in sorted(deps[op], key=lambda d: -pos[d]):    # latest first: its event usually covers the older ones
            assert pos[d] < pos[op], "schedule: a dependence points forward in the enqueue order"
            if d in known:
                continue
            assert stream[d] != s, "schedule: an unmet dependence on the op's own stream"
            edges.append((d, op))
            known |= behind[d] | {d}
        behind[op] = known
        last[s] = op
    return order, stream, edges


def check_schedule(deps: List[List[int]], order: Sequence[int], stream: Sequence[int], edges: Sequence[Tuple[int, int]]) -> None:
    """Independent re-statement of what the executor guarantees (stream FIFO + event waits + fork at the start): raises if an
    op could start before one of its dependences has finished.  Used by the CPU tests and by Plan.schedule()."""
    n = len(deps)
    assert sorted(order) == list(range(n)), "order is not a permutation"
    pos = {op: p for p, op in enumerate(order)}
    waits: Dict[int, List[int]] = {}
    for s, d in edges:
        assert pos[s] < pos[d] and stream[s] != stream[d], "edge must point forward, across streams"
        waits.setdefault(d, []).append(s)
    done_before: Dict[int, set] = {}
    last = {0: None, 1: None}
    for op in order:
        k = stream[op]
        known = set()
        if last[k] is not None:
            known |= done_before[last[k]] | {last[k]}
        for s in waits.get(op, []):
            known |= done_before[s] | {s}
        missing = [d for d in deps[op] if d not in known]
        assert not missing, f"op {op} may start before its dependences {missing}"
        done_before[op] = known
        last[k] = op
