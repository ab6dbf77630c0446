"""Teacher-forced, per-op replay of the TRAINING plans (forward + backward) against torch-CPU fp32 (test infrastructure).

`TrainBuilder.fwd_log` / `.bwd_log` (yolov6_amd/train_engine.py) say what every op of the two native plans reads and
writes.  `TrainChain` walks them in plan order.  Per op it

  1. computes the op's result on the CPU in fp32 torch (autograd of the forward statement for the backward ops) from the
     TEACHER's inputs - its own results of the previous ops, rounded to fp16 where the plan stores fp16 - never from a HIP
     result;
  2. launches that single op (`Plan.run_range(i, i + 1)`); the HIP buffers it reads hold the teacher's values, because
  3. after the comparison every output of the op is overwritten with the teacher's value.

The forward statement per op is the one `oracle.model_oracle.TrainOracle` composes (conv | batch-statistics BatchNorm with
eps / momentum of the module | branch sum | activation | shortcut; Detect training branch), so the teacher's head outputs
equal `TrainOracle(amp_fp16=True)` up to the rounding of the stored batch statistics - the caller checks that - which ties
the per-op references to the oracle that is pinned to the reference's training-mode goldens.

Error metric per output tensor: max |hip - ref| / max |ref| (relative to the tensor's largest magnitude).
"""
import torch
import torch.nn.functional as F

from yolov6_amd.engine import NCHWInput, TRef


def q16(t):
    return t.half().float()


ACT = {None: lambda t: t, "relu": F.relu, "silu": F.silu, "hardswish": F.hardswish}


class TrainChain:
    def __init__(self, graph, threads=None):
        self.g = graph
        self.cpu = {}          # buffer data_ptr -> teacher value, fp16 NCHW [B, cstride, H, W]
        self.stats = {}        # id(BnStats) -> dict(scale, shift, mean, invstd) fp32
        self.rows = []
        self.param_grads = {}  # id(param) -> teacher gradient (fp32)

    # ---------------------------------------------------------------- buffers
    def get(self, r: TRef):
        return self.cpu[r.buf.data_ptr()][:, r.coff:r.coff + r.C].float()

    def put(self, r: TRef, val, acc=False):
        key = r.buf.data_ptr()
        if key not in self.cpu:
            self.cpu[key] = torch.zeros((r.B, r.cstride, r.H, r.W), dtype=torch.float16)
        if acc:
            val = val + self.cpu[key][:, r.coff:r.coff + r.C].float()
        self.cpu[key][:, r.coff:r.coff + r.C] = val.half()
        return self.cpu[key][:, r.coff:r.coff + r.C].float()

    def has(self, r: TRef):
        return r.buf.data_ptr() in self.cpu

    @staticmethod
    def upload(r: TRef, val):
        r.to_nhwc_tensor().copy_(val.permute(0, 2, 3, 1).to(r.buf.device, torch.float16))

    @staticmethod
    def download(r: TRef):
        return r.to_nhwc_tensor().float().cpu().permute(0, 3, 1, 2)

    def _cmp(self, phase, i, e, desc, pairs, kind_tol, extra=None):
        """pairs: [(name, hip tensor, ref tensor)]"""
        err, worst = 0.0, ""
        n_over = 0
        for name, hip, ref in pairs:
            den = float(ref.abs().max())
            d = (hip.double() - ref.double()).abs()
            v = float(d.max()) / max(den, 1e-30) if den > 0 else float(hip.abs().max())
            n_over += int((d > kind_tol * max(den, 1e-30)).sum()) if den > 0 else int((hip != 0).sum())
            if v >= err:
                err, worst = v, name
        row = dict(phase=phase, op=i, kind=e["kind"], desc=desc, err=err, worst=worst, tol=kind_tol, n_over=n_over)
        if extra:
            row.update(extra)
        self.rows.append(row)

    # ---------------------------------------------------------------- forward
    def _w16(self, p):
        return q16(p.detach().float().cpu())

    def forward(self):
        g = self.g
        plan = g.fwd_plan
        for i, e in enumerate(g.fwd_log):
            k = e["kind"]
            outs = []          # (TRef, teacher value)   fp16 tensors
            extra = []         # (name, hip getter, ref, setter) for fp32 side outputs
            tol = 1e-3
            if k == "nchw2nhwc":
                outs.append((e["out"], q16(e["x"].float().cpu())))
                desc = "nchw2nhwc"
            elif k == "subsample2":
                outs.append((e["out"], self.get(e["x"])[:, :, ::2, ::2]))
                desc = "subsample2"
            elif k == "avgpool3":          # [x +] AvgPool2d(3, 1, 1)(x): QARepVGGBlockV2's raw identity + average-pool branches
                xv = self.get(e["x"])
                outs.append((e["out"], F.avg_pool2d(xv, 3, 1, 1) + (xv if e["with_identity"] else 0.0)))
                desc = f"avgpool3{'+id' if e['with_identity'] else ''} C={xv.shape[1]} {xv.shape[2]}x{xv.shape[3]}"
            elif k == "stem":
                x = q16(e["x"].float().cpu())
                w = self._w16(e["weight"])
                y = F.conv2d(x, w, None, stride=2, padding=e["k"] // 2)
                outs.append((e["out"], y))
                desc = f"stem k{e['k']} {w.shape[1]}->{w.shape[0]} out {y.shape[2]}x{y.shape[3]}"
            elif k == "conv":
                x = self.get(e["x"])
                w = self._w16(e["weight"])
                b = None if e["bias"] is None else e["bias"].detach().float().cpu()
                y = F.conv2d(x, w, b, stride=e["stride"], padding=e["k"] // 2)
                outs.append((e["out"], y))
                desc = f"conv {w.shape[1]}->{w.shape[0]} k{e['k']} s{e['stride']} out {y.shape[2]}x{y.shape[3]}"
            elif k in ("bn_train_stats", "bn_train_stats_multi"):
                # one op = the statistics of one tensor, or of the up-to-three branch tensors of a RepVGG block (one launch pair)
                items = e["items"] if k == "bn_train_stats_multi" else [(e["x"], e["bn"], e["stats"])]
                todo = []
                for xv_, bn, st in items:
                    x = self.get(xv_).double()
                    n = x.numel() / x.shape[1]
                    mean = x.mean((0, 2, 3))
                    var = x.var((0, 2, 3), unbiased=False)
                    invstd = 1.0 / torch.sqrt(var + bn.eps)
                    gamma = bn.weight.detach().double().cpu() if bn.weight is not None else torch.ones_like(mean)
                    beta = bn.bias.detach().double().cpu() if bn.bias is not None else torch.zeros_like(mean)
                    scale = gamma * invstd
                    shift = beta - mean * scale
                    ref = dict(scale=scale.float(), shift=shift.float(), mean=mean.float(), invstd=invstd.float())
                    self.stats[id(st)] = ref
                    self.stats[id(bn)] = ref           # channel slices of these statistics (BnStats.slice: MBLABlock's cv1) look them up by module
                    rm0, rv0 = bn.running_mean.detach().cpu().clone(), bn.running_var.detach().cpu().clone()
                    todo.append((x, bn, st, n, mean, var, ref, rm0, rv0))
                plan.run_range(i, i + 1)
                torch.cuda.synchronize()
                pairs = []
                for t_, (x, bn, st, n, mean, var, ref, rm0, rv0) in enumerate(todo):
                    C = x.shape[1]
                    tag = f"[{t_}]" if len(todo) > 1 else ""
                    pairs += [(nm + tag, getattr(st, nm)[:C].cpu(), ref[nm]) for nm in ("scale", "shift", "mean", "invstd")]
                    m = bn.momentum
                    pairs.append(("running_mean" + tag, bn.running_mean.detach().cpu(), ((1 - m) * rm0.double() + m * mean).float()))
                    pairs.append(("running_var" + tag, bn.running_var.detach().cpu(), ((1 - m) * rv0.double() + m * var * (n / max(n - 1, 1))).float()))
                self._cmp("fwd", i, e, f"bn_stats x{len(todo)} C={C} n={int(n)}", pairs, 1e-4)
                for (x, bn, st, n, mean, var, ref, rm0, rv0) in todo:
                    C = x.shape[1]
                    for nm in ("scale", "shift", "mean", "invstd"):
                        getattr(st, nm)[:C].copy_(ref[nm].to(st.scale.device))
                continue
            elif k == "bnact_forward":
                z = 0.0
                for t, st in e["branches"]:
                    xb = self.get(t)
                    if st is not None:
                        s_ = self.stats.get(id(st)) or self.stats[id(st.module)]
                        c0 = getattr(st, "c0", 0)
                        xb = xb * s_["scale"][c0:c0 + xb.shape[1]].view(1, -1, 1, 1) + s_["shift"][c0:c0 + xb.shape[1]].view(1, -1, 1, 1)
                    z = z + xb
                o = ACT[e["act"]](z)
                if e["res"] is not None:
                    a = 1.0 if e["alpha"] is None else float(e["alpha"].detach().float().cpu())
                    o = o + a * self.get(e["res"])
                outs.append((e["out"], o))
                tol = 3e-3
                desc = f"bnact x{len(e['branches'])} {e['act']}{'+res' if e['res'] is not None else ''} C={o.shape[1]} {o.shape[2]}x{o.shape[3]}"
            elif k == "sppf":
                y1 = F.max_pool2d(self.get(e["x"]), 5, 1, 2)
                y2 = F.max_pool2d(y1, 5, 1, 2)
                y3 = F.max_pool2d(y2, 5, 1, 2)
                outs += list(zip(e["outs"], (y1, y2, y3)))
                tol = 0.0
                desc = "sppf pools"
            elif k == "convt":
                y = F.conv_transpose2d(self.get(e["x"]), self._w16(e["weight"]), e["bias"].detach().float().cpu(), stride=2)
                outs.append((e["out"], y))
                desc = f"convT {e['weight'].shape[0]}->{e['weight'].shape[1]}"
            elif k == "head_pack":
                sc = torch.cat([torch.sigmoid(self.get(c)).flatten(2).permute(0, 2, 1) for c in e["cls"]], 1)
                di = torch.cat([self.get(r).flatten(2).permute(0, 2, 1) for r in e["reg"]], 1)
                plan.run_range(i, i + 1)
                torch.cuda.synchronize()
                self._cmp("fwd", i, e, "head_pack", [("scores", e["scores"].cpu(), sc), ("distri", e["distri"].cpu(), di)], 1e-5)
                e["scores"].copy_(sc.to(e["scores"].device))
                e["distri"].copy_(di.to(e["distri"].device))
                self.scores, self.distri = sc, di
                continue
            else:
                raise NotImplementedError(f"train_replay: forward op kind {k}")
            plan.run_range(i, i + 1)
            torch.cuda.synchronize()
            pairs = []
            for r, v in outs:
                v16 = self.put(r, v)
                pairs.append((f"out C={r.C}", self.download(r), v16))
                self.upload(r, v16)
            self._cmp("fwd", i, e, desc, pairs, tol)

    # ---------------------------------------------------------------- backward
    def _grad_view(self, p):
        a = self.g.arena
        o = a.offset_of(p)
        return a.grad[o:o + p.numel()].view(p.shape)

    def _check_param(self, phase, i, e, desc, items, tol):
        """items: [(name, param, ref grad)] - the op ACCUMULATES into the arena: the slots were zeroed before the launch."""
        pairs = []
        for name, p, ref in items:
            pairs.append((name, self._grad_view(p).detach().cpu().float(), ref.float()))
            prev = self.param_grads.get(id(p))
            # several ops may each own a channel slice of one parameter's gradient (BatchNorm of MBLABlock's cv1): their sum
            self.param_grads[id(p)] = ref.float() if (prev is None or not getattr(self, "_partial_bn", False)) else prev + ref.float()
        self._cmp(phase, i, e, desc, pairs, tol)

    def backward(self, dscores, ddistri):
        """dscores / ddistri: the teacher's loss gradient wrt the head outputs (fp32, CPU)."""
        g = self.g
        plan = g.bwd_plan
        dev = g.arena.grad.device
        g.arena.grad.zero_()
        g.dscores.copy_(dscores.to(dev))
        g.ddistri.copy_(ddistri.to(dev))
        for i, e in enumerate(g.bwd_log):
            k = e["kind"]
            if k == "wgrad_transpose":            # layout pass feeding the next wgrad: runs on the teacher's tensors, judged there
                plan.run_range(i, i + 1)
                continue
            if k == "head_unpack_backward":
                sc = self.scores
                dz = dscores * sc * (1 - sc)
                outs, a0 = [], 0
                for l, (dc, dr) in enumerate(zip(e["dcls"], e["dreg"])):
                    hw = dc.H * dc.W
                    nc, nreg = e["nc"][l], e["nreg"][l]
                    c = dz[:, a0:a0 + hw].permute(0, 2, 1).reshape(dc.B, nc, dc.H, dc.W)
                    r = ddistri[:, a0:a0 + hw].permute(0, 2, 1).reshape(dr.B, nreg, dr.H, dr.W)
                    outs.append((TRef(dc.buf, dc.B, dc.H, dc.W, nc, dc.cstride, dc.coff), c))
                    outs.append((TRef(dr.buf, dr.B, dr.H, dr.W, nreg, dr.cstride, dr.coff), r))
                    for full in (dc, dr):       # the 8-channel padding of the prediction-conv gradients stays zero
                        if not self.has(full):
                            self.put(full, torch.zeros(full.B, full.C, full.H, full.W))
                    a0 += hw
                plan.run_range(i, i + 1)
                torch.cuda.synchronize()
                pairs = []
                for r, v in outs:
                    v16 = self.put(r, v)
                    pairs.append((f"d C={r.C} {r.H}x{r.W}", self.download(r), v16))
                    self.upload(r, v16)
                self._cmp("bwd", i, e, "head_unpack", pairs, 1e-3)
                continue
            if k == "bnact_backward":
                self._bnact_backward(i, e)
                continue
            if k == "wgrad":
                self._wgrad(i, e)
                continue
            if k == "wgrad_stem":
                self._wgrad_stem(i, e)
                continue
            if k == "channel_sum":
                p = e["param"]
                ref = self.get(e["x"]).double().sum((0, 2, 3)).float()
                self._grad_view(p).zero_()
                plan.run_range(i, i + 1)
                torch.cuda.synchronize()
                self._check_param("bwd", i, e, f"bias grad C={ref.numel()}", [("db", p, ref)], 1e-3)
                continue
            if k == "conv":
                self._dgrad(i, e)
                continue
            if k == "dgrad_s2":
                self._dgrad_s2(i, e)
                continue
            if k == "avgpool3":            # backward of the branch: the same (self-adjoint) op on the gradient, maybe accumulating
                gv = self.get(e["x"])
                d = F.avg_pool2d(gv, 3, 1, 1) + (gv if e["with_identity"] else 0.0)
                if e["acc"]:
                    self.upload(e["out"], self.get(e["out"]))
                plan.run_range(i, i + 1)
                torch.cuda.synchronize()
                v16 = self.put(e["out"], d, acc=bool(e["acc"]))
                self._cmp("bwd", i, e, f"avgpool3 backward{' acc' if e['acc'] else ''}", [("dx", self.download(e["out"]), v16)], 3e-3)
                self.upload(e["out"], v16)
                continue
            if k == "space_to_depth2":
                x = self.get(e["x"])
                B, Cn, H2, W2 = x.shape
                o = torch.cat([x[:, :, dy::2, dx::2] for dy in (0, 1) for dx in (0, 1)], 1)
                plan.run_range(i, i + 1)
                torch.cuda.synchronize()
                v16 = self.put(e["out"], o)
                self._cmp("bwd", i, e, "space_to_depth2", [("out", self.download(e["out"]), v16)], 0.0)
                self.upload(e["out"], v16)
                continue
            if k == "sppf_backward":
                x = self.get(e["x"]).requires_grad_(True)
                y1 = F.max_pool2d(x, 5, 1, 2)
                y2 = F.max_pool2d(y1, 5, 1, 2)
                y3 = F.max_pool2d(y2, 5, 1, 2)
                g1, g2, g3 = (self.get(t) for t in e["dys"])
                (dx,) = torch.autograd.grad([y1, y2, y3], [x], [g1, g2, g3])
                plan.run_range(i, i + 1)
                torch.cuda.synchronize()
                v16 = self.put(e["dx"], dx, acc=True)
                self._cmp("bwd", i, e, "sppf pools backward", [("dx", self.download(e["dx"]), v16)], 3e-3)
                self.upload(e["dx"], v16)
                continue
            raise NotImplementedError(f"train_replay: backward op kind {k}")

    def _bnact_backward(self, i, e):
        plan = self.g.bwd_plan
        leaves, params = [], []
        z = 0.0
        for t, st in e["branches"]:
            xb = self.get(t).requires_grad_(True)
            leaves.append(xb)
            if st is not None:
                bn = st.module
                c0, nch = getattr(st, "c0", 0), xb.shape[1]          # a channel slice of a wider BatchNorm (MBLABlock's cv1)
                w = bn.weight.detach().float().cpu()[c0:c0 + nch].clone().requires_grad_(True) if bn.weight is not None else None
                b = bn.bias.detach().float().cpu()[c0:c0 + nch].clone().requires_grad_(True) if bn.bias is not None else None
                params.append((bn, w, b, c0, nch))
                yb = F.batch_norm(xb, None, None, w, b, True, 0.0, bn.eps)
            else:
                params.append(None)
                yb = xb
            z = z + yb
        o = ACT[e["act"]](z)
        res_leaf = alpha_leaf = None
        if e["res"] is not None:
            res_leaf = self.get(e["res"]).requires_grad_(True)
            if e["alpha"] is not None:
                alpha_leaf = e["alpha"].detach().float().cpu().clone().requires_grad_(True)
                o = o + alpha_leaf * res_leaf
            else:
                o = o + res_leaf
        dout = self.get(e["dout"])
        wanted = list(leaves)
        for pr in params:
            if pr is not None:
                wanted += [t for t in pr[1:3] if t is not None]
        if res_leaf is not None:
            wanted.append(res_leaf)
        if alpha_leaf is not None:
            wanted.append(alpha_leaf)
        grads = dict(zip(map(id, wanted), torch.autograd.grad(o, wanted, dout)))
        # zero the parameter-gradient slots this op accumulates into
        items = []
        def widen(g_, full, c0, nch):      # the op accumulates into channels [c0, c0 + nch) of the parameter's gradient only
            if nch == full.numel():
                return g_
            r = torch.zeros(full.numel())
            r[c0:c0 + nch] = g_
            return r
        for pr in params:
            if pr is None:
                continue
            bn, w, b, c0, nch = pr
            if w is not None:
                self._grad_view(bn.weight).zero_()
                items.append(("dgamma", bn.weight, widen(grads[id(w)], bn.weight, c0, nch)))
            if b is not None:
                self._grad_view(bn.bias).zero_()
                items.append(("dbeta", bn.bias, widen(grads[id(b)], bn.bias, c0, nch)))
        if alpha_leaf is not None:
            self._grad_view(e["alpha"]).zero_()
            items.append(("dalpha", e["alpha"], grads[id(alpha_leaf)]))
        plan.run_range(i, i + 1)
        torch.cuda.synchronize()
        # ReLU ties: where the pre-activation z is zero to within the rounding of its fp32 evaluation order, the mask - and with
        # it the whole element of dx - is decided by the order of three fused multiply-adds.  Such elements (a handful among
        # 26 M at 32ch x 320^2 x b8) are excluded from the element-wise comparison and counted; the per-channel sums
        # (dgamma / dbeta) keep them.
        keep, n_ties = None, 0
        if e["act"] == "relu":
            zd = z.detach()
            keep = zd.abs() > 1e-5 * float(zd.abs().max())
            n_ties = int((~keep).sum())
        pairs = []
        for xb, (dref, dil, acc) in zip(leaves, e["dx"]):
            dx = grads[id(xb)]
            kp = keep
            if dil == 2:
                full = torch.zeros(dref.B, dref.C, dref.H, dref.W)
                full[:, :, ::2, ::2] = dx
                dx = full
                if keep is not None:
                    kp = torch.ones(dref.B, dref.C, dref.H, dref.W, dtype=torch.bool)
                    kp[:, :, ::2, ::2] = keep
            v16 = self.put(dref, dx, acc=bool(acc))
            hip = self.download(dref)
            pairs.append((f"dx dil{dil} acc{acc}", hip * kp if kp is not None else hip, v16 * kp if kp is not None else v16))
            self.upload(dref, v16)
        if res_leaf is not None:
            dref, acc = e["dres"]
            v16 = self.put(dref, grads[id(res_leaf)], acc=bool(acc))
            pairs.append((f"dres acc{acc}", self.download(dref), v16))
            self.upload(dref, v16)
        o_ = e["out"]
        desc = f"bnact_bwd x{len(leaves)} {e['act']} C={o_.C} {o_.H}x{o_.W}"
        self._cmp("bwd", i, e, desc, pairs, 3e-3, extra=dict(relu_ties_excluded=n_ties))
        if items:
            self._partial_bn = any(pr is not None and pr[4] != pr[0].weight.numel() for pr in params)
            self._check_param("bwd", i, e, desc + " params", items, 3e-3)
            self._partial_bn = False

    def _dy_compact(self, e):
        dy = self.get(e["dy"]) if "dy" in e else self.get(e["x"])
        d = e.get("dil", 1)
        return dy[:, :, ::d, ::d] if d > 1 else dy

    def _wgrad(self, i, e):
        plan = self.g.bwd_plan
        p = e["weight"]
        dy = self._dy_compact(e)
        if e.get("convt"):
            x = self.get(e["x"])
            w = p.detach().float().cpu().clone().requires_grad_(True)
            (ref,) = torch.autograd.grad(F.conv_transpose2d(x, w, None, stride=2), [w], dy)
            desc = f"wgrad convT {p.shape[0]}->{p.shape[1]}"
        else:
            xsrc = e["x"]
            x = q16(xsrc.float().cpu()) if isinstance(xsrc, torch.Tensor) else self.get(xsrc)
            dy = dy[:, :e["cout"]]
            k, s = e["k"], e["stride"]
            w = p.detach().float().cpu().clone().requires_grad_(True)
            (ref,) = torch.autograd.grad(F.conv2d(x, w, None, stride=s, padding=k // 2), [w], dy)
            desc = f"wgrad {p.shape[1]}->{p.shape[0]} k{k} s{s} {dy.shape[2]}x{dy.shape[3]}"
        self._grad_view(p).zero_()
        plan.run_range(i, i + 1)
        torch.cuda.synchronize()
        self._check_param("bwd", i, e, desc, [("dW", p, ref)], 1e-3)

    def _wgrad_stem(self, i, e):
        """Both weight gradients of the stem block from the NCHW image (csrc/wgrad_stem.hip): 3x3 s2 and, if present, 1x1 s2."""
        plan = self.g.bwd_plan
        x = q16(e["x"].float().cpu())
        items = []
        for p, dyv, k in zip(e["weights"], e["dys"], (3, 1)):
            if p is None:
                continue
            dy = self.get(dyv)[:, :e["cout"]]
            w = p.detach().float().cpu().clone().requires_grad_(True)
            (ref,) = torch.autograd.grad(F.conv2d(x, w, None, stride=2, padding=k // 2), [w], dy)
            items.append((f"dW{k}", p, ref))
            self._grad_view(p).zero_()
        plan.run_range(i, i + 1)
        torch.cuda.synchronize()
        self._check_param("bwd", i, e, f"wgrad stem {x.shape[1]}->{e['cout']} k3+k1 s2 image {x.shape[2]}x{x.shape[3]}", items, 1e-3)

    def _dgrad(self, i, e):
        plan = self.g.bwd_plan
        role = e["role"]
        out = e["out"]
        if role == "dgrad":
            p = e["weight"]
            k, s = e["fwd_k"], e["fwd_stride"]
            dy = self.get(e["x"])
            d = e["dil"]
            if d > 1:
                dy = dy[:, :, ::d, ::d]
            dy = dy[:, :p.shape[0]]
            w = self._w16(p)
            opad = (out.H + 2 * (k // 2) - k) % s if s > 1 else 0
            dx = F.conv_transpose2d(dy, w, None, stride=s, padding=k // 2, output_padding=opad)
            desc = f"dgrad {p.shape[0]}->{p.shape[1]} k{k} s{s} out {out.H}x{out.W}{' acc' if e['acc'] else ''}"
        elif role == "convt_dgrad":
            p = e["weight"]                                   # [Cin, Cout, 2, 2]
            dout = self.get(e["dy"])
            dx = F.conv2d(dout, self._w16(p), None, stride=2)
            desc = f"convT dgrad {p.shape[1]}->{p.shape[0]}{' acc' if e['acc'] else ''}"
        else:
            raise NotImplementedError(role)
        assert dx.shape[2:] == (out.H, out.W), (dx.shape, out.H, out.W)
        plan.run_range(i, i + 1)
        torch.cuda.synchronize()
        v16 = self.put(out, dx, acc=bool(e["acc"]))
        self._cmp("bwd", i, e, desc, [("dx", self.download(out), v16)], 3e-3)
        self.upload(out, v16)

    def _dgrad_s2(self, i, e):
        """csrc/dgrad_s2.hip: dx of a stride-2 3x3 conv and, when present, of the 1x1 stride-2 conv of the same input, from the
        compact gradients, one launch."""
        plan = self.g.bwd_plan
        out = e["out"]
        dx = None
        names = []
        for dyv, p, k in zip(e["dys"], e["weights"], (3, 1)):
            if dyv is None:
                continue
            dy = self.get(dyv)[:, :p.shape[0]]
            t = F.conv_transpose2d(dy, self._w16(p), None, stride=2, padding=k // 2, output_padding=1)
            dx = t if dx is None else dx + t
            names.append(f"k{k}")
        p3 = e["weights"][0]
        assert dx.shape[2:] == (out.H, out.W), (dx.shape, out.H, out.W)
        plan.run_range(i, i + 1)
        torch.cuda.synchronize()
        v16 = self.put(out, dx, acc=bool(e["acc"]))
        desc = f"dgrad_s2 {p3.shape[0]}->{p3.shape[1]} {'+'.join(names)} out {out.H}x{out.W}{' acc' if e['acc'] else ''}"
        self._cmp("bwd", i, e, desc, [("dx", self.download(out), v16)], 3e-3)
        self.upload(out, v16)
