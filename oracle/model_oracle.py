"""ORACLE (test infrastructure - never imported by the product path).

CPU restatement, in plain functional PyTorch fp32, of the reference's model forward for the
hot path.  It is driven by a reference-format state_dict (the keys the reference's
`Model.state_dict()` produces) and the model config, and restates:

  * ConvModule eval forward / forward_fuse      yolov6/layers/common.py:45-54
  * RepVGGBlock train-form eval forward          common.py:250-255
  * QARepVGGBlock[V2] train-form eval forward    common.py:341-347, :416-426
  * fuse_model + switch_to_deploy (the ORDER the reference needs)
                                                 utils/torch_utils.py:50-94, common.py:257-319, :349-393, :428-477
  * RepBlock / BottleRep / BepC3                 common.py:569-650
  * SPPFModule / CSPSPPFModule                   common.py:97-158
  * Transpose / BiFusion                         common.py:181-194, :695-718
  * EfficientRep / CSPBepBackbone(_P6)           models/efficientrep.py:104-118, :501-516
  * RepBiFPANNeck / CSPRepBiFPANNeck(_P6)        models/reppan.py:215-237, :1086-1116
  * Detect eval forward + anchors + dist2bbox    models/effidehead.py:93-139,
                                                 assigners/anchor_generator.py:13-33, utils/general.py:32-43

Pinning: tests/test_oracle_cpu.py checks this file against tests/golden/*.npz, which
tests/golden/gen_golden.py produced by running the unmodified reference modules from
/root/reference on CPU with the same synthetic weights (oracle/synth.py).

`emulate_fp16=True` reproduces what `model.half()` does on the reference side: weights,
biases and every op output are rounded to fp16 (math in fp32, like the device kernels and
like cuDNN/MIOpen fp16 convs with fp32 accumulation).
"""
import math
import re

import torch
import torch.nn.functional as F

BN_EPS = 1e-3  # initialize_weights sets eps=1e-3 on every BatchNorm2d (utils/torch_utils.py:43-45)


def make_divisible(x, d):
    return math.ceil(x / d) * d


class Arch:
    """Structure numbers build_network derives from the config (models/yolo.py:55-69)."""

    def __init__(self, cfg, num_classes=80):
        m = cfg["model"]
        self.mode = cfg.get("training_mode", "repvgg")
        reps = list(m["backbone"]["num_repeats"]) + list(m["neck"]["num_repeats"])
        chans = list(m["backbone"]["out_channels"]) + list(m["neck"]["out_channels"])
        d, w = m["depth_multiple"], m["width_multiple"]
        self.n = [(max(round(i * d), 1) if i > 1 else i) for i in reps]
        self.c = [make_divisible(i * w, 8) for i in chans]
        self.backbone = m["backbone"]["type"]
        self.neck = m["neck"]["type"]
        self.csp = "CSP" in self.backbone
        self.csp_e = m["backbone"].get("csp_e", 0.5)
        self.mbla = m["backbone"].get("stage_block_type", "BepC3") == "MBLABlock"   # models/yolo.py:75-78
        self.fuse_P2 = bool(m["backbone"].get("fuse_P2"))
        self.cspsppf = bool(m["backbone"].get("cspsppf"))
        self.nl = m["head"]["num_layers"]
        self.use_dfl = bool(m["head"]["use_dfl"])
        self.reg_max = m["head"]["reg_max"]
        self.nc = num_classes
        self.strides = [8, 16, 32] if self.nl == 3 else [8, 16, 32, 64]
        self.p6 = self.backbone.endswith("6") or self.backbone.endswith("P6")
        # activation of plain conv blocks: conv_silu models replace RepVGG by ConvBNSiLU (common.py:734-735)
        self.silu_body = self.mode == "conv_silu"


class Oracle:
    def __init__(self, cfg, sd, num_classes=80, emulate_fp16=False):
        self.a = Arch(cfg, num_classes)
        self.fp16 = emulate_fp16
        self.sd = {k: v.detach().float() for k, v in sd.items()}
        self.trace = {}
        self._stride = {}
        self.train_form = False

    # ------------------------------------------------------------------ numerics helpers
    def q(self, t):
        return t.half().float() if self.fp16 else t

    # hooks of the training oracle (identity here): conv / block-output rounding of an AMP-style fp16 pipeline
    def c2d(self, x, w, b=None, **kw):
        return F.conv2d(x, w, b, **kw)

    def r16(self, t):
        return t

    @staticmethod
    def act(x, kind):
        if kind == "relu":
            return F.relu(x)
        if kind == "silu":
            return F.silu(x)
        if kind == "hardswish":
            return F.hardswish(x)
        return x

    def bn(self, x, p):
        sd = self.sd
        s = sd[p + ".weight"] / torch.sqrt(sd[p + ".running_var"] + BN_EPS)
        t = sd[p + ".bias"] - sd[p + ".running_mean"] * s
        return x * s.view(1, -1, 1, 1) + t.view(1, -1, 1, 1)

    def fold(self, w, b, p):
        """fuse_conv_and_bn (torch_utils.py:50-82)."""
        sd = self.sd
        s = sd[p + ".weight"] / torch.sqrt(sd[p + ".running_var"] + BN_EPS)
        t = sd[p + ".bias"] - sd[p + ".running_mean"] * s
        wf = w * s.view(-1, 1, 1, 1)
        bf = t if b is None else t + b * s
        return wf, bf

    def conv_fused(self, x, w, b, stride, act, post=None):
        k = w.shape[-1]
        y = F.conv2d(x, self.q(w), None if b is None else self.q(b), stride=stride, padding=k // 2)
        if post is not None:
            # QARepVGG deploy keeps a BatchNorm after the conv (a separate fp16 op in the reference)
            y = self.q(y) * self.q(post[0]).view(1, -1, 1, 1) + self.q(post[1]).view(1, -1, 1, 1)
        return self.q(self.act(self.q(y), act))

    # ------------------------------------------------------------------ blocks
    def conv_module_wb(self, p):
        """Weights of the equivalent bias-conv of a ConvModule at prefix p (fused or not)."""
        sd = self.sd
        w = sd[p + ".conv.weight"]
        b = sd.get(p + ".conv.bias")
        if p + ".bn.weight" in sd:
            w, b = self.fold(w, b, p + ".bn")
        return w, b

    def convbn(self, x, p, act, stride=1):
        """ConvBNReLU / ConvBNSiLU wrapper modules: parameters live under `<p>.block`."""
        w, b = self.conv_module_wb(p + ".block")
        return self.conv_fused(x, w, b, stride, act)

    def repvgg_wb(self, p):
        """Deploy kernel/bias of a (QA)RepVGG block: rbr_reparam if present, else re-parameterise
        (get_equivalent_kernel_bias common.py:257-261 / :349-364 / :428-448)."""
        sd = self.sd
        if p + ".rbr_reparam.weight" in sd:
            return sd[p + ".rbr_reparam.weight"], sd[p + ".rbr_reparam.bias"]
        k3, b3 = self.conv_module_wb(p + ".rbr_dense")
        cin = k3.shape[1]
        has_id = k3.shape[0] == cin and self._stride.get(p, 1) == 1
        ident = torch.zeros_like(k3)
        if has_id:
            idx = torch.arange(cin)
            ident[idx, idx, 1, 1] = 1.0
        if self.a.mode.startswith("qarepvgg"):
            k = k3 + F.pad(sd[p + ".rbr_1x1.weight"], [1, 1, 1, 1])
            if self.a.mode == "qarepvggv2" and has_id:
                avg = torch.zeros_like(k3)
                idx = torch.arange(cin)
                avg[idx, idx, :, :] = 1.0 / 9.0
                k = k + avg
            if has_id:
                k = k + ident
            return k, b3
        k1, b1 = self.conv_module_wb(p + ".rbr_1x1")
        k = k3 + F.pad(k1, [1, 1, 1, 1])
        b = b3 + b1
        if p + ".rbr_identity.weight" in sd:
            q = p + ".rbr_identity"
            s = sd[q + ".weight"] / torch.sqrt(sd[q + ".running_var"] + BN_EPS)
            k = k + ident * s.view(-1, 1, 1, 1)
            b = b + sd[q + ".bias"] - sd[q + ".running_mean"] * s
        return k, b

    def repvgg_train_form(self, x, p, stride):
        """Eval forward of the un-fused multi-branch block (common.py:250-255, :341-347, :416-426)."""
        sd = self.sd
        d = self.bn(self.c2d(x, sd[p + ".rbr_dense.conv.weight"], None, stride=stride, padding=1), p + ".rbr_dense.bn")
        cin, cout = sd[p + ".rbr_dense.conv.weight"].shape[1], sd[p + ".rbr_dense.conv.weight"].shape[0]
        has_id = cin == cout and stride == 1
        if self.a.mode.startswith("qarepvgg"):
            y = d + self.c2d(x, sd[p + ".rbr_1x1.weight"], None, stride=stride)
            if has_id:
                y = y + x
                if self.a.mode == "qarepvggv2":
                    y = y + F.avg_pool2d(x, 3, stride, 1)
            return self.r16(F.relu(self.bn(y, p + ".bn")))
        y = d + self.bn(self.c2d(x, sd[p + ".rbr_1x1.conv.weight"], None, stride=stride), p + ".rbr_1x1.bn")
        if has_id:
            y = y + self.bn(x, p + ".rbr_identity")
        return self.r16(F.relu(y))

    def block(self, x, p, stride=1):
        """`block = get_block(training_mode)` (common.py:721-737): RepVGG family or plain ConvBN{ReLU,SiLU}."""
        self._stride[p] = stride
        mode = self.a.mode
        if mode in ("conv_relu", "conv_silu"):
            return self.convbn(x, p, "relu" if mode == "conv_relu" else "silu", stride)
        if self.train_form and (p + ".rbr_dense.conv.weight") in self.sd:
            return self.repvgg_train_form(x, p, stride)
        w, b = self.repvgg_wb(p)
        post = None
        if mode.startswith("qarepvgg"):
            q = p + ".bn"
            s = self.sd[q + ".weight"] / torch.sqrt(self.sd[q + ".running_var"] + BN_EPS)
            post = (s, self.sd[q + ".bias"] - self.sd[q + ".running_mean"] * s)
        return self.conv_fused(x, w, b, stride, "relu", post)

    def bottlerep(self, x, p):
        y = self.block(self.block(x, p + ".conv1"), p + ".conv2")
        alpha = self.sd.get(p + ".alpha")
        a = self.q(alpha) if alpha is not None else 1.0
        # BottleRep always has equal in/out channels in the BepC3 stages -> shortcut (common.py:597-608)
        return self.q(y + self.q(a * x))

    def repblock(self, x, p, n, bottle=False):
        one = self.bottlerep if bottle else self.block
        if bottle:
            n = n // 2
        x = one(x, p + ".conv1")
        for i in range(n - 1):
            x = one(x, f"{p}.block.{i}")
        return x

    def _body_act(self):
        return "silu" if self.a.silu_body else "relu"

    def bepc3(self, x, p, n):
        act = self._body_act()
        a = self.repblock(self.convbn(x, p + ".cv1", act), p + ".m", n, bottle=True)
        b = self.convbn(x, p + ".cv2", act)
        return self.convbn(torch.cat((a, b), 1), p + ".cv3", act)

    def bottlerep3(self, x, p):
        """BottleRep3 (common.py:611-632): three basic blocks, shortcut weighted by alpha (in == out channels here)."""
        y = self.block(self.block(self.block(x, p + ".conv1"), p + ".conv2"), p + ".conv3")
        alpha = self.sd.get(p + ".alpha")
        a = self.q(alpha) if alpha is not None else 1.0
        return self.q(y + self.q(a * x))

    def mbla(self, x, p, n):
        """MBLABlock (common.py:653-692)."""
        act = self._body_act()
        n = n // 2
        if n <= 0:
            n = 1
        if n == 1:
            n_list = [0, 1]
        else:
            e = 1
            while e * 2 < n:
                e *= 2
            n_list = [0, e, n]
        y = self.conv_module(x, p + ".cv1", act)
        c = y.shape[1] // len(n_list)
        ys = list(y.split(c, 1))
        all_y = [ys[0]]
        for mi, steps in enumerate(n_list[1:]):
            all_y.append(ys[mi + 1])
            for j in range(steps):
                all_y.append(self.bottlerep3(all_y[-1], f"{p}.m.{mi}.{j}"))
        return self.conv_module(torch.cat(all_y, 1), p + ".cv2", act)

    def conv_module(self, x, p, act, stride=1):
        """A bare ConvModule at prefix p (conv -> BatchNorm -> activation, common.py:26-54), eval form: BN folded."""
        w, b = self.conv_module_wb(p)
        return self.conv_fused(x, w, b, stride, act)

    def stage(self, x, p, n):
        if self.a.csp and self.a.mbla:
            return self.mbla(x, p, n)
        return self.bepc3(x, p, n) if self.a.csp else self.repblock(x, p, n)

    def pool5(self, x):
        return F.max_pool2d(x, 5, 1, 2)

    def sppf(self, x, p, act):
        x = self.convbn(x, p + ".cv1", act)
        y1 = self.pool5(x)
        y2 = self.pool5(y1)
        return self.convbn(torch.cat([x, y1, y2, self.pool5(y2)], 1), p + ".cv2", act)

    def cspsppf(self, x, p, act):
        x1 = self.convbn(self.convbn(self.convbn(x, p + ".cv1", act), p + ".cv3", act), p + ".cv4", act)
        y0 = self.convbn(x, p + ".cv2", act)
        y1 = self.pool5(x1)
        y2 = self.pool5(y1)
        y3 = self.convbn(self.convbn(torch.cat([x1, y1, y2, self.pool5(y2)], 1), p + ".cv5", act), p + ".cv6", act)
        return self.convbn(torch.cat((y0, y3), 1), p + ".cv7", act)

    def merge(self, x, p):
        act = self._body_act()
        if self.a.cspsppf:
            return self.cspsppf(x, p + ".cspsppf", act)
        return self.sppf(x, p + ".sppf", act)

    def transpose(self, x, p):
        y = F.conv_transpose2d(x, self.q(self.sd[p + ".upsample_transpose.weight"]),
                               self.q(self.sd[p + ".upsample_transpose.bias"]), stride=2)
        return self.q(y)

    def bifusion(self, xs, p):
        x0 = self.transpose(xs[0], p + ".upsample")
        x1 = self.convbn(xs[1], p + ".cv1", "relu")
        x2 = self.convbn(self.convbn(xs[2], p + ".cv2", "relu"), p + ".downsample", "relu", stride=2)
        return self.convbn(torch.cat((x0, x1, x2), 1), p + ".cv3", "relu")

    # ------------------------------------------------------------------ backbone / neck / head
    def backbone(self, x):
        a = self.a
        last = 6 if a.p6 else 5
        outs = []
        x = self.block(x, "backbone.stem", 2)
        self.trace["stem"] = x
        first = 2 if (a.fuse_P2 or a.backbone == "CSPBepBackbone_P6") else 3
        for k in range(2, last + 1):
            p = f"backbone.ERBlock_{k}"
            x = self.block(x, p + ".0", 2)
            x = self.stage(x, p + ".1", a.n[k - 1])
            if k == last:
                x = self.merge(x, p + ".2")
            if k >= first:
                outs.append(x)
        return outs

    def neck(self, feats):
        a = self.a
        nb = 6 if a.p6 else 5  # number of backbone entries in channels_list / num_repeats
        if not a.p6:
            x3, x2, x1, x0 = feats if len(feats) == 4 else [None] + list(feats)     # the PAN necks take three maps (no fuse_P2)
            names = dict(td=[("reduce_layer0", "Bifusion0", "Rep_p4", nb + 0), ("reduce_layer1", "Bifusion1", "Rep_p3", nb + 1)],
                         bu=[("downsample2", "Rep_n3", nb + 2), ("downsample1", "Rep_n4", nb + 3)])
            pyr = [x3, x2, x1, x0]
        else:
            names = dict(td=[("reduce_layer0", "Bifusion0", "Rep_p5", nb + 0), ("reduce_layer1", "Bifusion1", "Rep_p4", nb + 1),
                             ("reduce_layer2", "Bifusion2", "Rep_p3", nb + 2)],
                         bu=[("downsample2", "Rep_n4", nb + 3), ("downsample1", "Rep_n5", nb + 4),
                             ("downsample0", "Rep_n6", nb + 5)])
            pyr = list(feats)
        cur = pyr[-1]
        laterals = []
        pan = "BiFPAN" not in a.neck      # RepPANNeck / RepPANNeck6 / CSPRepPANNeck / CSPRepPANNeck_P6 (reppan.py:81-102, :352-391)
        for i, (red, fus, stg, ri) in enumerate(names["td"]):
            fpn = self.convbn(cur, "neck." + red, "relu")
            laterals.append(fpn)
            if pan:     # cat([upsample_i(fpn_out_i), backbone map]) -> stage
                up = self.transpose(fpn, f"neck.upsample{i}")
                cur = self.stage(torch.cat([up, pyr[-2 - i]], 1), "neck." + stg, a.n[ri])
                continue
            cur = self.stage(self.bifusion([fpn, pyr[-2 - i], pyr[-3 - i]], "neck." + fus), "neck." + stg, a.n[ri])
        outs = [cur]
        for j, (down, stg, ri) in enumerate(names["bu"]):
            d = self.convbn(cur, "neck." + down, "relu", stride=2)
            cur = self.stage(torch.cat([d, laterals[len(laterals) - 1 - j]], 1), "neck." + stg, a.n[ri])
            outs.append(cur)
        return outs

    def head(self, feats):
        sd = self.sd
        cls_l, reg_l = [], []
        for i, x in enumerate(feats):
            f = self.convbn(x, f"detect.stems.{i}", "silu")
            c = self.convbn(f, f"detect.cls_convs.{i}", "silu")
            r = self.convbn(f, f"detect.reg_convs.{i}", "silu")
            co = self.q(F.conv2d(c, self.q(sd[f"detect.cls_preds.{i}.weight"]), self.q(sd[f"detect.cls_preds.{i}.bias"])))
            ro = self.q(F.conv2d(r, self.q(sd[f"detect.reg_preds.{i}.weight"]), self.q(sd[f"detect.reg_preds.{i}.bias"])))
            self.trace[f"cls_logits{i}"], self.trace[f"reg_raw{i}"] = co, ro
            cls_l.append(co)
            reg_l.append(ro)
        return self.decode(cls_l, reg_l)

    def decode(self, cls_logits, reg_raw):
        """Detect eval tail (effidehead.py:104-139) from the per-level prediction maps (NCHW): DFL softmax.proj,
        sigmoid, anchors (anchor_generator.py:13-33), dist2bbox 'xywh' (general.py:32-43), x stride, concat."""
        a, sd = self.a, self.sd
        cls_l, reg_l = [], []
        for co, ro in zip(cls_logits, reg_raw):
            b, _, h, w = co.shape
            if a.use_dfl:
                ro = ro.reshape(-1, 4, a.reg_max + 1, h * w).permute(0, 2, 1, 3)
                ro = self.q(F.conv2d(self.q(F.softmax(ro, dim=1)), sd["detect.proj_conv.weight"]))
            cls_l.append(self.q(torch.sigmoid(co)).reshape(b, a.nc, h * w))
            reg_l.append(ro.reshape(b, 4, h * w))
        cls = torch.cat(cls_l, -1).permute(0, 2, 1)
        reg = torch.cat(reg_l, -1).permute(0, 2, 1)
        pts, strd = [], []
        dev = cls_logits[0].device
        for x, s in zip(cls_logits, a.strides):
            h, w = x.shape[-2:]
            gy, gx = torch.meshgrid(torch.arange(h, device=dev) + 0.5, torch.arange(w, device=dev) + 0.5, indexing="ij")
            pts.append(torch.stack([gx, gy], -1).float().reshape(-1, 2))
            strd.append(torch.full((h * w, 1), float(s), device=dev))
        pts, strd = torch.cat(pts), torch.cat(strd)
        lt, rb = reg[..., :2], reg[..., 2:]
        x1y1, x2y2 = pts - lt, pts + rb
        box = torch.cat([(x1y1 + x2y2) / 2, x2y2 - x1y1], -1) * strd
        return torch.cat([box, torch.ones(box.shape[0], box.shape[1], 1, device=dev), cls], -1)

    def forward_device(self, x):
        """Calibration helper (tools/torch_baseline.py): the same graph on whatever device/dtype the
        state_dict and x live on (no fp32 upcast) - the reference's eager path on a GPU."""
        self.train_form = False
        feats = self.neck(self.backbone(x))
        return self.head(list(feats)), feats

    def forward(self, x, train_form=False):
        """Returns (det [B,A,5+nc] fp32, featmaps list).  train_form=True evaluates the un-fused
        multi-branch blocks literally (eval-mode BN); otherwise every block runs in deploy form."""
        self.train_form = train_form
        x = self.q(x.float())
        feats = self.neck(self.backbone(x))
        self.trace["feats"] = feats
        return self.head(list(feats)), feats


def deploy_state_dict(cfg, sd, num_classes=80):
    """fuse_model + switch_to_deploy as a state_dict -> state_dict transform (reference key names:
    `<block>.rbr_reparam.{weight,bias}`, `<convmodule>.conv.{weight,bias}`; QA blocks keep `<block>.bn.*`)."""
    o = Oracle(cfg, sd, num_classes)
    out = {}
    sdk = o.sd
    done = set()
    # discover strides by a dry structural walk: stride only matters for the identity test, and the
    # reference decides identity by in==out channels AND stride==1; stride-2 blocks are exactly
    # backbone.stem and backbone.ERBlock_k.0
    for k in sdk:
        if k.endswith(".rbr_dense.conv.weight"):
            p = k[: -len(".rbr_dense.conv.weight")]
            o._stride[p] = 2 if (p == "backbone.stem" or re.fullmatch(r"backbone\.ERBlock_\d+\.0", p)) else 1
    for k in list(sdk):
        if k.endswith(".rbr_dense.conv.weight"):
            p = k[: -len(".rbr_dense.conv.weight")]
            w, b = o.repvgg_wb(p)
            out[p + ".rbr_reparam.weight"], out[p + ".rbr_reparam.bias"] = w, b
            done.add(p)
    for k, v in sdk.items():
        owner = next((p for p in done if k.startswith(p + ".rbr_")), None)
        if owner is not None:
            continue
        if k.endswith(".conv.weight") and (k[: -len(".conv.weight")] + ".bn.weight") in sdk:
            p = k[: -len(".conv.weight")]
            w, b = o.conv_module_wb(p)
            out[p + ".conv.weight"], out[p + ".conv.bias"] = w, b
            done.add(p + ".bn")
            continue
        if any(k.startswith(d + ".") for d in done if d.endswith(".bn")):
            continue
        out[k] = v
    return out


class TrainOracle(Oracle):
    """ORACLE for the TRAINING-mode forward: the un-fused graph with batch-statistics
    BatchNorm (eps 1e-3, momentum 0.03: torch_utils.py:38-47) and the Detect training branch
    (effidehead.py:72-92).  fp32.  `new_stats[prefix]` holds the running_mean / running_var a BatchNorm would have after
    the step (unbiased variance in the running estimate, biased in the normalisation - torch semantics)."""
    BN_MOMENTUM = 0.03

    def __init__(self, cfg, sd, num_classes=80, amp_fp16=False):
        """amp_fp16=True: conv outputs and block outputs are rounded to fp16 (weights too, as autocast casts them), with a
        straight-through gradient - the noise floor of ANY fp16-activation training pipeline (the reference under
        torch.cuda.amp, this package's HIP path) relative to the fp32 graph.  Used to put measured deviations in scale."""
        super().__init__(cfg, sd, num_classes, emulate_fp16=False)
        self.train_form = True
        self.new_stats = {}
        self.amp = amp_fp16

    def r16(self, t):
        return t + (t.half().float() - t).detach() if self.amp else t

    def c2d(self, x, w, b=None, **kw):
        if not self.amp:
            return F.conv2d(x, w, b, **kw)
        return self.r16(F.conv2d(x, self.r16(w), None if b is None else self.r16(b), **kw))

    def transpose(self, x, p):
        y = F.conv_transpose2d(x, self.r16(self.sd[p + ".upsample_transpose.weight"]), self.r16(self.sd[p + ".upsample_transpose.bias"]),
                               stride=2)
        return self.r16(y)

    def bn(self, x, p):
        sd = self.sd
        mean = x.mean((0, 2, 3))
        var = x.var((0, 2, 3), unbiased=False)
        n = x.numel() / x.shape[1]
        m = self.BN_MOMENTUM
        self.new_stats[p] = ((1 - m) * sd[p + ".running_mean"] + m * mean,
                             (1 - m) * sd[p + ".running_var"] + m * var * (n / max(n - 1, 1)))
        y = (x - mean.view(1, -1, 1, 1)) / torch.sqrt(var.view(1, -1, 1, 1) + BN_EPS)
        return y * sd[p + ".weight"].view(1, -1, 1, 1) + sd[p + ".bias"].view(1, -1, 1, 1)

    def convbn(self, x, p, act, stride=1):
        sd = self.sd
        w = sd[p + ".block.conv.weight"]
        y = self.c2d(x, w, sd.get(p + ".block.conv.bias"), stride=stride, padding=w.shape[-1] // 2)
        if p + ".block.bn.weight" in sd:
            y = self.bn(y, p + ".block.bn")
        return self.r16(self.act(y, act))

    def block(self, x, p, stride=1):
        mode = self.a.mode
        if mode in ("conv_relu", "conv_silu"):
            return self.convbn(x, p, "relu" if mode == "conv_relu" else "silu", stride)
        return self.repvgg_train_form(x, p, stride)

    def conv_module(self, x, p, act, stride=1):
        """A bare ConvModule in training mode (common.py:45-49): conv -> BatchNorm on batch statistics -> activation."""
        sd = self.sd
        w = sd[p + ".conv.weight"]
        y = self.c2d(x, w, sd.get(p + ".conv.bias"), stride=stride, padding=w.shape[-1] // 2)
        if p + ".bn.weight" in sd:
            y = self.bn(y, p + ".bn")
        return self.r16(self.act(y, act))

    def head_train(self, feats):
        sd = self.sd
        xs, cls_l, reg_l = [], [], []
        for i, x in enumerate(feats):
            f = self.convbn(x, f"detect.stems.{i}", "silu")
            c = self.convbn(f, f"detect.cls_convs.{i}", "silu")
            r = self.convbn(f, f"detect.reg_convs.{i}", "silu")
            co = self.c2d(c, sd[f"detect.cls_preds.{i}.weight"], sd[f"detect.cls_preds.{i}.bias"])
            ro = self.c2d(r, sd[f"detect.reg_preds.{i}.weight"], sd[f"detect.reg_preds.{i}.bias"])
            xs.append(f)
            cls_l.append(torch.sigmoid(co).flatten(2).permute(0, 2, 1))
            reg_l.append(ro.flatten(2).permute(0, 2, 1))
        return xs, torch.cat(cls_l, 1), torch.cat(reg_l, 1)

    def head_train_fuseab(self, feats, anchors_init, na=3):
        """Detect training branch with the anchor-based auxiliary predictions (heads/effidehead_fuseab.py:94-139).
        -> stems, cls_ab [B,na*A,nc], reg_ab [B,na*A,4], cls_af [B,A,nc], reg_af [B,A,4*(reg_max+1)]"""
        sd = self.sd
        anc = torch.as_tensor(anchors_init, dtype=torch.float32).reshape(len(feats), na, 2) / \
            torch.tensor(self.a.strides, dtype=torch.float32).view(-1, 1, 1)
        xs, cab, rab, caf, raf = [], [], [], [], []
        for i, x in enumerate(feats):
            b, _, h, w = x.shape
            f = self.convbn(x, f"detect.stems.{i}", "silu")
            c = self.convbn(f, f"detect.cls_convs.{i}", "silu")
            r = self.convbn(f, f"detect.reg_convs.{i}", "silu")
            xs.append(f)
            co = torch.sigmoid(self.c2d(c, sd[f"detect.cls_preds_ab.{i}.weight"], sd[f"detect.cls_preds_ab.{i}.bias"]))
            cab.append(co.reshape(b, na, -1, h, w).permute(0, 1, 3, 4, 2).flatten(1, 3))
            ro = self.c2d(r, sd[f"detect.reg_preds_ab.{i}.weight"], sd[f"detect.reg_preds_ab.{i}.bias"])
            ro = ro.reshape(b, na, -1, h, w).permute(0, 1, 3, 4, 2)
            wh = ((ro[..., 2:4].sigmoid() * 2) ** 2) * anc[i].reshape(1, na, 1, 1, 2)
            rab.append(torch.cat([ro[..., :2], wh], -1).flatten(1, 3))
            caf.append(torch.sigmoid(self.c2d(c, sd[f"detect.cls_preds.{i}.weight"], sd[f"detect.cls_preds.{i}.bias"])).flatten(2).permute(0, 2, 1))
            raf.append(self.c2d(r, sd[f"detect.reg_preds.{i}.weight"], sd[f"detect.reg_preds.{i}.bias"]).flatten(2).permute(0, 2, 1))
        return xs, torch.cat(cab, 1), torch.cat(rab, 1), torch.cat(caf, 1), torch.cat(raf, 1)

    def head_train_distill_ns(self, feats):
        """Detect training branch of the N / S self-distillation head (heads/effidehead_distill_ns.py:80-103): a third output,
        plain (l, t, r, b) distances from `reg_preds`, next to the DFL logits of `reg_preds_dist`.
        -> stems, cls_scores [B,A,nc], reg_distri [B,A,4*(reg_max+1)], reg_lrtb [B,A,4]"""
        sd = self.sd
        xs, cls_l, dist_l, lrtb_l = [], [], [], []
        for i, x in enumerate(feats):
            f = self.convbn(x, f"detect.stems.{i}", "silu")
            c = self.convbn(f, f"detect.cls_convs.{i}", "silu")
            r = self.convbn(f, f"detect.reg_convs.{i}", "silu")
            co = self.c2d(c, sd[f"detect.cls_preds.{i}.weight"], sd[f"detect.cls_preds.{i}.bias"])
            do = self.c2d(r, sd[f"detect.reg_preds_dist.{i}.weight"], sd[f"detect.reg_preds_dist.{i}.bias"])
            lo = self.c2d(r, sd[f"detect.reg_preds.{i}.weight"], sd[f"detect.reg_preds.{i}.bias"])
            xs.append(f)
            cls_l.append(torch.sigmoid(co).flatten(2).permute(0, 2, 1))
            dist_l.append(do.flatten(2).permute(0, 2, 1))
            lrtb_l.append(lo.flatten(2).permute(0, 2, 1))
        return xs, torch.cat(cls_l, 1), torch.cat(dist_l, 1), torch.cat(lrtb_l, 1)

    def forward_train_distill_ns(self, x):
        self.new_stats = {}
        feats = self.neck(self.backbone(x.float()))
        return self.head_train_distill_ns(list(feats)), feats

    def forward_train_fuseab(self, x, anchors_init):
        self.new_stats = {}
        feats = self.neck(self.backbone(x.float()))
        return self.head_train_fuseab(list(feats), anchors_init), feats

    def forward_train(self, x):
        """-> (head stem outputs per level, cls_scores [B,A,nc], reg_distri [B,A,4*(reg_max+1)]), neck featmaps."""
        self.new_stats = {}
        feats = self.neck(self.backbone(x.float()))
        return self.head_train(list(feats)), feats
