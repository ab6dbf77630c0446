"""Test infrastructure: a PlanBuilder look-alike that EXECUTES every lowered op immediately on the CPU in fp32 with plain
torch ops, on the same buffer / channel-slice views (engine.TRef) the real builder hands out.

What it checks is the WIRING of the `lower()` methods - which producer writes which channel slice of which buffer, the
re-parameterised weights, epilogue order (bias -> post affine -> activation -> residual), strides, pooling chains, the decode
arguments - for every model family, without a GPU.  It is not a product path (the library has no CPU execution at all) and
says nothing about the HIP kernels; those are compared op by op and end to end in the `-m gpu` tests."""
import torch
import torch.nn.functional as F

from yolov6_amd.engine import NCHWInput, TRef


def _act(y, kind):
    if kind is None:
        return y
    return {"relu": F.relu, "silu": F.silu, "hardswish": F.hardswish}[kind](y)


class MockBuilder:
    is_train = False

    def __init__(self):
        self.device = torch.device("cpu")
        self.quant = None
        self.force_variant = -1
        self.fp16_reads = []
        self.op_log = []
        self._no_quant = 0

    # ---------------------------------------------------------------- buffers / views
    def new_buffer(self, B, H, W, C_):
        return TRef(torch.full((B, H, W, C_), float("nan")), B, H, W, C_, C_, 0)      # unwritten channels poison the result

    @staticmethod
    def _read(ref):      # -> NCHW fp32
        v = ref.to_nhwc_tensor()
        assert not torch.isnan(v).any(), "an op reads channels nobody wrote"
        return v.permute(0, 3, 1, 2).contiguous()

    @staticmethod
    def _write(ref, y):
        assert tuple(y.shape) == (ref.B, ref.C, ref.H, ref.W), (tuple(y.shape), (ref.B, ref.C, ref.H, ref.W))
        ref.to_nhwc_tensor().copy_(y.permute(0, 2, 3, 1))

    def as_nhwc(self, x):
        if isinstance(x, TRef):
            return x
        t = x.t.float()
        B, C_, H, W = t.shape
        out = self.new_buffer(B, H, W, C_)
        self._write(out, t)
        self.op_log.append(dict(kind="nchw2nhwc", x=x.t, out=out))
        return out

    def to_nchw(self, x, dtype=torch.float32):
        return self._read(x).to(dtype)

    def keep_fp16(self, refs):
        self.fp16_reads += list(refs)

    def no_quant(self):
        pb = self

        class _Ctx:
            def __enter__(self):
                pb._no_quant += 1

            def __exit__(self, *exc):
                pb._no_quant -= 1
        return _Ctx()

    # ---------------------------------------------------------------- ops
    def conv(self, x, weight, bias, stride=1, act=None, out=None, post=None, res=None, res_alpha=None):
        image = None
        if isinstance(x, NCHWInput):
            t = x.t
            image = t.float() / 255.0 if t.dtype == torch.uint8 else t
            x = self.as_nhwc(NCHWInput(image))
        w = weight.detach().float()
        Cout, Cin, K, _ = w.shape
        assert x.C == Cin
        y = F.conv2d(self._read(x), w, None if bias is None else bias.detach().float(), stride=stride, padding=K // 2)
        if post is not None:
            y = y * post[0].detach().float().view(1, -1, 1, 1) + post[1].detach().float().view(1, -1, 1, 1)
        y = _act(y, act)
        if res is not None:
            a = 1.0 if res_alpha is None else res_alpha.detach().float().reshape(())
            y = y + a * self._read(res)
        if out is None:
            out = self.new_buffer(x.B, y.shape[2], y.shape[3], Cout)
        self._write(out, y)
        # same schema as engine.PlanBuilder.op_log (tests/plan_replay.py walks either)
        self.op_log.append(dict(kind="stem" if image is not None else "conv", x=image if image is not None else x, out=out, w=w,
                                b=bias, stride=stride, act=act, post=post, res=res, alpha=res_alpha))
        return out

    def convt2x2(self, x, weight, bias, out=None):
        x = self.as_nhwc(x)
        y = F.conv_transpose2d(self._read(x), weight.detach().float(), bias.detach().float(), stride=2)
        if out is None:
            out = self.new_buffer(x.B, 2 * x.H, 2 * x.W, weight.shape[1])
        self._write(out, y)
        self.op_log.append(dict(kind="convt", x=x, out=out, w=weight, b=bias))
        return out

    def sppf_pool(self, x, y1, y2, y3):
        p = self._read(x)
        for dst in (y1, y2, y3):
            p = F.max_pool2d(p, 5, 1, 2)
            self._write(dst, p)
        self.op_log.append(dict(kind="sppf", x=x, outs=[y1, y2, y3]))

    def head_decode(self, cls, reg, strides, use_dfl, reg_max, proj, nc, grid_cell_offset=0.5):
        """Eval branch of Detect (effidehead.py:104-139), restated with torch ops on the lowered head outputs."""
        outs = []
        for c, r, s in zip(cls, reg, strides):
            assert c.C == nc
            B, H, W = c.B, c.H, c.W
            score = torch.sigmoid(self._read(c)).flatten(2).permute(0, 2, 1)                     # [B, HW, nc]
            d = self._read(r)
            if use_dfl:
                pj = proj.detach().float().reshape(-1)
                assert r.C == 4 * pj.numel()
                d = (F.softmax(d.reshape(B, 4, pj.numel(), H * W), 2) * pj.view(1, 1, -1, 1)).sum(2)   # [B, 4, HW]
            else:
                assert r.C == 4
                d = d.flatten(2)
            d = d.permute(0, 2, 1)                                                                    # (l, t, r, b)
            ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
            pts = torch.stack([xs + grid_cell_offset, ys + grid_cell_offset], -1).reshape(1, H * W, 2)
            x1y1, x2y2 = pts - d[..., :2], pts + d[..., 2:]
            box = torch.cat([(x1y1 + x2y2) / 2, x2y2 - x1y1], -1) * float(s)
            outs.append(torch.cat([box, torch.ones(B, H * W, 1), score], -1))
        det = torch.cat(outs, 1)
        self.op_log.append(dict(kind="decode", cls=list(cls), reg=list(reg), out=det, strides=list(strides), use_dfl=bool(use_dfl),
                                reg_max=int(reg_max), proj=proj, nc=nc))
        return det

    def head_pred_decode(self, cls_feat, reg_feat, cls_preds, reg_preds, strides, use_dfl, reg_max, proj, nc, grid_cell_offset=0.5):
        """The fused head tail (engine.PlanBuilder.head_pred_decode): cls_pred / reg_pred 1x1 convs of every level + decode,
        logged as ONE op."""
        n0 = len(self.op_log)
        cls = [self.conv(c, w, b, 1, None) for c, (w, b) in zip(cls_feat, cls_preds)]
        reg = [self.conv(r, w, b, 1, None) for r, (w, b) in zip(reg_feat, reg_preds)]
        det = self.head_decode(cls, reg, strides, use_dfl, reg_max, proj, nc, grid_cell_offset)
        del self.op_log[n0:]
        self.op_log.append(dict(kind="pred_decode", cls_feat=list(cls_feat), reg_feat=list(reg_feat),
                                cls_preds=[(w.detach().float(), None if b is None else b.detach().float()) for w, b in cls_preds],
                                reg_preds=[(w.detach().float(), None if b is None else b.detach().float()) for w, b in reg_preds],
                                out=det, strides=list(strides), use_dfl=bool(use_dfl), reg_max=int(reg_max), proj=proj, nc=nc))
        return det
