"""Teacher-forced, per-layer replay of a native plan against the CPU oracle (test infrastructure).

The plan's op log (yolov6_amd.engine.PlanBuilder.op_log) says what every op reads and writes.  `OracleChain`
walks it in plan order and computes every op with the fp16-emulating oracle's own arithmetic
(`oracle.model_oracle.Oracle.conv_fused / transpose / pool5 / decode`) on the ORACLE's activation of the
previous layer, never on a HIP result.  Per op it

  1. uploads the oracle's input activation into the HIP op's input buffer (teacher forcing),
  2. launches that single op (`Plan.run_range(i, i + 1)`),
  3. compares the HIP output with the oracle's output:  max |hip - ref| / max(1, |ref|).

The chain's final tensor must equal `Oracle.forward(x)` (checked by the caller), which ties the per-layer
references to the oracle that is pinned to the reference's goldens.
"""
import numpy as np
import torch
import torch.nn.functional as F


class OracleChain:
    def __init__(self, plan, orc):
        self.plan, self.orc = plan, orc
        self.cpu = {}        # buffer data_ptr -> oracle activation, NCHW fp32 [B, cstride, H, W]
        self.rows = []       # per op: dict(op, kind, desc, err, err_abs)
        self.final = None
        self.i8_stats = []   # per int8 conv, plan order: accumulator abs-max (compared with Int8Oracle.stats)

    # ---------------------------------------------------------------- buffer bookkeeping
    def _get(self, ref):
        return self.cpu[ref.buf.data_ptr()][:, ref.coff:ref.coff + ref.C].contiguous()

    def _put(self, ref, val):
        key = ref.buf.data_ptr()
        if key not in self.cpu:
            self.cpu[key] = torch.zeros((ref.B, ref.cstride, ref.H, ref.W), dtype=torch.float32)
        self.cpu[key][:, ref.coff:ref.coff + ref.C] = val

    @staticmethod
    def _upload(ref, val):
        ref.to_nhwc_tensor().copy_(val.permute(0, 2, 3, 1).to(ref.buf.device, torch.float16))

    @staticmethod
    def _download(ref):
        return ref.to_nhwc_tensor().float().cpu().permute(0, 3, 1, 2)

    # ---------------------------------------------------------------- oracle arithmetic per op kind
    def _oracle_op(self, e):
        o = self.orc
        k = e["kind"]
        if k in ("conv", "stem"):
            x = o.q(e["x"].float().cpu()) if k == "stem" else self._get(e["x"])
            b = None if e["b"] is None else e["b"].detach().float().cpu()
            post = None if e["post"] is None else tuple(t.detach().float().cpu() for t in e["post"])
            # the op log says this op is an fp16 conv: use the base oracle's arithmetic even when `o` is an Int8Oracle
            # (whose own conv_fused decides by call order, which an op-by-op walk does not reproduce)
            from oracle.model_oracle import Oracle
            y = Oracle.conv_fused(o, x, e["w"].detach().float().cpu(), b, e["stride"], e["act"], post)
            if e["res"] is not None:       # BottleRep shortcut: q(y + q(alpha * x)), oracle.bottlerep / common.py:608
                a = 1.0 if e["alpha"] is None else o.q(e["alpha"].detach().float().cpu())
                y = o.q(y + o.q(a * self._get(e["res"])))
            return [(e["out"], y)]
        if k in ("pw_s2", "stem_s2"):      # producer -> 3x3 stride-2 conv as one op: the producer's output is an fp16 tensor too
            from oracle.model_oracle import Oracle
            pe, se = e["producer"], e["s2"]
            x = o.q(pe["x"].float().cpu()) if k == "stem_s2" else self._get(pe["x"])
            if k == "stem_s2" and pe["x"].dtype == torch.uint8:
                x = o.q(pe["x"].float().cpu() / 255.0)
            fb = lambda t: None if t is None else t.detach().float().cpu()
            m = Oracle.conv_fused(o, x, pe["w"].detach().float().cpu(), fb(pe["b"]), pe["stride"], pe["act"], None)
            y = Oracle.conv_fused(o, m, se["w"].detach().float().cpu(), fb(se["b"]), 2, se["act"], None)
            return [(e["out"], y)]
        if k == "conv_i8":       # int8 conv (oracle/int8_oracle.py defines the arithmetic; scales come from the plan's op log)
            from oracle.int8_oracle import int8_conv
            b = None if e["b"] is None else e["b"].detach().float().cpu()
            post = None if e["post"] is None else tuple(t.detach().float().cpu() for t in e["post"])
            y, acc = int8_conv(o, self._get(e["x"]), e["w"].detach().float().cpu(), b, e["stride"], e["act"], post, e["amax"])
            self.i8_stats.append(dict(acc_absmax=float(acc.abs().max()), desc=_describe(e), amax=e["amax"]))
            if e["res"] is not None:
                a = 1.0 if e["alpha"] is None else o.q(e["alpha"].detach().float().cpu())
                y = o.q(y + o.q(a * self._get(e["res"])))
            return [(e["out"], y)]
        if k == "convt":
            y = F.conv_transpose2d(self._get(e["x"]), o.q(e["w"].detach().float().cpu()),
                                   o.q(e["b"].detach().float().cpu()), stride=2)
            return [(e["out"], o.q(y))]
        if k == "sppf":
            y1 = o.pool5(self._get(e["x"]))
            y2 = o.pool5(y1)
            return list(zip(e["outs"], (y1, y2, o.pool5(y2))))
        if k == "nchw2nhwc":
            return [(e["out"], o.q(e["x"].float().cpu()))]
        raise NotImplementedError(k)

    # ---------------------------------------------------------------- the walk
    def run(self, teacher_force=True):
        plan = self.plan
        for i, e in enumerate(plan.op_log):
            k = e["kind"]
            if k in ("nhwc2nchw", "absmax"):
                continue
            if k in ("decode", "pred_decode"):
                if k == "pred_decode":      # the fused head tail: cls_pred / reg_pred (1x1 conv + bias, fp16 output) + decode
                    from oracle.model_oracle import Oracle
                    ins = e["cls_feat"] + e["reg_feat"]
                    vals = [self._get(r) for r in ins]
                    nl = len(e["cls_feat"])
                    cls = [Oracle.conv_fused(self.orc, v, w, b, 1, None) for v, (w, b) in zip(vals[:nl], e["cls_preds"])]
                    reg = [Oracle.conv_fused(self.orc, v, w, b, 1, None) for v, (w, b) in zip(vals[nl:], e["reg_preds"])]
                else:
                    ins = e["cls"] + e["reg"]
                    cls = [self._get(r) for r in e["cls"]]
                    reg = [self._get(r) for r in e["reg"]]
                    vals = cls + reg
                ref = self.orc.decode(cls, reg)
                if teacher_force:
                    for r, v in zip(ins, vals):
                        self._upload(r, v)
                    plan.run_range(i, i + 1)
                    torch.cuda.synchronize()
                    hip = e["out"].float().cpu()
                    d = (hip - ref).abs()
                    row = dict(op=i, kind=k, desc=f"{k} A={ref.shape[1]}",
                               err=float((d / ref.abs().clamp(min=1.0)).max()), err_abs=float(d.max()),
                               err_scores=float(d[..., 5:].max()), err_box_px=float(d[..., :4].max()))
                    if k == "pred_decode":
                        # the fused op contains the reg_pred conv: a one-ulp flip of its fp16 output (the bar of every conv op)
                        # moves a box edge by ulp(distance) x stride - the bound of the box columns of this op
                        dmax = [float(e["reg_max"]) if e["use_dfl"] else float(r_.abs().max()) for r_ in reg]
                        row["box_tol_px"] = 1.5 * max(dm * 2.0 ** -10 * float(st) for dm, st in zip(dmax, e["strides"]))
                    self.rows.append(row)
                self.final = ref
                continue
            if teacher_force:
                for key in ("x", "res"):
                    r = e.get(key)
                    if r is not None and not isinstance(r, torch.Tensor):
                        self._upload(r, self._get(r))
                if e.get("q_in") is not None:        # the op reads the producer's int8 twin: teacher-force that too
                    from oracle.int8_oracle import quantize_act
                    qv = quantize_act(self._get(e["x"]), e["amax"]).to(torch.int8)
                    e["q_in"].to_nhwc_tensor().copy_(qv.permute(0, 2, 3, 1).to(e["q_in"].buf.device))
            outs = self._oracle_op(e)
            for r, v in outs:
                self._put(r, v)
            if teacher_force:
                plan.run_range(i, i + 1)
                torch.cuda.synchronize()
                err = err_abs = 0.0
                twin_mismatch = None
                if e.get("q_out") is not None:       # the int8 copy for the consumers must be the quantised oracle output, exactly
                    from oracle.int8_oracle import quantize_act
                    want = quantize_act(outs[0][1], e["q_out_amax"]).to(torch.int8)
                    got = e["q_out"].to_nhwc_tensor().cpu().permute(0, 3, 1, 2)
                    twin_mismatch = float((got != want).float().mean())
                    a16 = float(torch.tensor(e["q_out_amax"]).half())
                    err = max(err, float((got.float() - want.float()).abs().max()) * a16 / 127.0 / max(1.0, a16))
                if e.get("has_out", True):
                    for r, v in outs:
                        d = (self._download(r) - v).abs()
                        err = max(err, float((d / v.abs().clamp(min=1.0)).max()))
                        err_abs = max(err_abs, float(d.max()))
                row = dict(op=i, kind=k, desc=_describe(e), err=err, err_abs=err_abs)
                if twin_mismatch is not None:
                    row["twin_mismatch"] = twin_mismatch
                self.rows.append(row)
        return self.rows


def _describe(e):
    k = e["kind"]
    if k in ("conv", "stem", "conv_i8"):
        w = e["w"]
        o = e["out"]
        extra = ("+post" if e["post"] is not None else "") + ("+res" if e["res"] is not None else "")
        return f"{k} {w.shape[1]}->{w.shape[0]} k{w.shape[-1]} s{e['stride']} {e['act']}{extra} out {o.B}x{o.H}x{o.W}"
    if k in ("pw_s2", "stem_s2"):
        pw, sw, o = e["producer"]["w"], e["s2"]["w"], e["out"]
        return (f"{k} {pw.shape[1]}->{pw.shape[0]} k{pw.shape[-1]} {e['producer']['act']} | {sw.shape[1]}->{sw.shape[0]} k3 s2 {e['s2']['act']} "
                f"out {o.B}x{o.H}x{o.W}")
    if k == "convt":
        return f"convT {e['w'].shape[0]}->{e['w'].shape[1]} out {e['out'].H}x{e['out'].W}"
    if k == "sppf":
        return f"sppf pools C={e['x'].C} {e['x'].H}x{e['x'].W}"
    return k


def box_report(det, ref):
    """Absolute pixel deviation of the decoded boxes (cx, cy, w, h in input pixels)."""
    d = np.abs(np.asarray(det[..., :4], np.float64) - np.asarray(ref[..., :4], np.float64))
    return dict(max_px=float(d.max()), mean_px=float(d.mean()), p999_px=float(np.quantile(d, 0.999)),
                scores_max=float(np.abs(np.asarray(det[..., 5:], np.float64) - np.asarray(ref[..., 5:], np.float64)).max()))
