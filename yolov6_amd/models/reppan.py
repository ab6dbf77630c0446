"""Necks with the reference's class names and state_dict keys (yolov6/models/reppan.py).

The bi-directional FPN is lowered concat-free: every `torch.cat` of the reference
(reppan.py:228,232 and common.py:718) becomes a pre-allocated NHWC buffer whose channel
slices the producers write directly (conv epilogues take an output channel offset/stride).

The older uni-directional PAN necks (RepPANNeck, RepPANNeck6, CSPRepPANNeck, CSPRepPANNeck_P6: `configs/experiment/*`, the
v2.0 checkpoints) concatenate an up-sampled map with a BACKBONE output; that one is copied into its slot of the buffer by a
1x1 convolution with the identity matrix (exact: one product per output, fp32 accumulation, the fp16 value comes back
unchanged) - inference only.
"""
import torch

from ..layers.common import BepC3, BiFusion, BottleRep, ConvBNReLU, HipModule, MBLABlock, RepBlock, RepVGGBlock, Transpose


class _BiFPAN(HipModule):
    """Top-down (reduce -> BiFusion -> stage) then bottom-up (downsample -> cat -> stage).
    Subclasses create the attributes under the reference's names and list them in
    `_td` (top-down) and `_bu` (bottom-up) order."""
    _td = ()   # (reduce, bifusion, stage)  from the deepest level up
    _bu = ()   # (downsample, stage)        from the shallowest level down

    def lower(self, pb, x, out=None):
        feats = [pb.as_nhwc(t) for t in x]            # shallow ... deep (x_k ... x0)
        n_td = len(self._td)
        deep = feats[-1]
        lateral = []                                   # fpn_out_i, each living in its bottom-up cat buffer
        cur = deep
        for i, (reduce_n, fuse_n, stage_n) in enumerate(self._td):
            reduce, fuse, stage = getattr(self, reduce_n), getattr(self, fuse_n), getattr(self, stage_n)
            # bottom-up partner of this lateral: cat([down_feat, fpn_out_i]) (reference forward order)
            down = getattr(self, self._bu[n_td - 1 - i][0])
            c_down = down.block.conv.out_channels
            c_lat = reduce.block.conv.out_channels
            cat = pb.new_buffer(cur.B, cur.H, cur.W, c_down + c_lat)
            fpn_out = reduce.lower(pb, cur, out=cat.slice(c_down, c_lat))
            lateral.append((cat, c_down))
            mid, low = feats[-2 - i], feats[-3 - i]
            cur = stage.lower(pb, fuse.lower(pb, [fpn_out, mid, low]))
        outs = [cur]
        for j, (down_n, stage_n) in enumerate(self._bu):
            cat, c_down = lateral[n_td - 1 - j]
            getattr(self, down_n).lower(pb, cur, out=cat.slice(0, c_down))
            cur = getattr(self, stage_n).lower(pb, cat)
            outs.append(cur)
        return outs


class RepBiFPANNeck(_BiFPAN):
    '''RepBiFPAN neck (YOLOv6-N/S).  Reference: reppan.py:132-237.'''
    _td = (("reduce_layer0", "Bifusion0", "Rep_p4"), ("reduce_layer1", "Bifusion1", "Rep_p3"))
    _bu = (("downsample2", "Rep_n3"), ("downsample1", "Rep_n4"))

    def __init__(self, channels_list=None, num_repeats=None, block=RepVGGBlock):
        super().__init__()
        assert channels_list is not None
        assert num_repeats is not None
        c, n = channels_list, num_repeats
        stage = lambda i, o, r: RepBlock(in_channels=i, out_channels=o, n=r, block=block)
        self.reduce_layer0 = ConvBNReLU(in_channels=c[4], out_channels=c[5], kernel_size=1, stride=1)
        self.Bifusion0 = BiFusion(in_channels=[c[3], c[2]], out_channels=c[5])
        self.Rep_p4 = stage(c[5], c[5], n[5])
        self.reduce_layer1 = ConvBNReLU(in_channels=c[5], out_channels=c[6], kernel_size=1, stride=1)
        self.Bifusion1 = BiFusion(in_channels=[c[2], c[1]], out_channels=c[6])
        self.Rep_p3 = stage(c[6], c[6], n[6])
        self.downsample2 = ConvBNReLU(in_channels=c[6], out_channels=c[7], kernel_size=3, stride=2)
        self.Rep_n3 = stage(c[6] + c[7], c[8], n[7])
        self.downsample1 = ConvBNReLU(in_channels=c[8], out_channels=c[9], kernel_size=3, stride=2)
        self.Rep_n4 = stage(c[5] + c[9], c[10], n[8])


class CSPRepBiFPANNeck(_BiFPAN):
    '''CSP RepBiFPAN neck (YOLOv6-M/L).  Reference: reppan.py:666-785.'''
    _td = RepBiFPANNeck._td
    _bu = RepBiFPANNeck._bu

    def __init__(self, channels_list=None, num_repeats=None, block=BottleRep, csp_e=float(1) / 2,
                 stage_block_type="BepC3"):
        super().__init__()
        assert channels_list is not None
        assert num_repeats is not None
        if stage_block_type not in ("BepC3", "MBLABlock"):
            raise NotImplementedError
        stage_block = BepC3 if stage_block_type == "BepC3" else MBLABlock      # reppan.py:559-564, :684-689
        c, n = channels_list, num_repeats
        stage = lambda i, o, r: stage_block(in_channels=i, out_channels=o, n=r, e=csp_e, block=block)
        self.reduce_layer0 = ConvBNReLU(in_channels=c[4], out_channels=c[5], kernel_size=1, stride=1)
        self.Bifusion0 = BiFusion(in_channels=[c[3], c[2]], out_channels=c[5])
        self.Rep_p4 = stage(c[5], c[5], n[5])
        self.reduce_layer1 = ConvBNReLU(in_channels=c[5], out_channels=c[6], kernel_size=1, stride=1)
        self.Bifusion1 = BiFusion(in_channels=[c[2], c[1]], out_channels=c[6])
        self.Rep_p3 = stage(c[6], c[6], n[6])
        self.downsample2 = ConvBNReLU(in_channels=c[6], out_channels=c[7], kernel_size=3, stride=2)
        self.Rep_n3 = stage(c[6] + c[7], c[8], n[7])
        self.downsample1 = ConvBNReLU(in_channels=c[8], out_channels=c[9], kernel_size=3, stride=2)
        self.Rep_n4 = stage(c[5] + c[9], c[10], n[8])


class _BiFPAN6(_BiFPAN):
    """Four-level (P3..P6) wiring shared by the plain and the CSP neck (reppan.py:394-541, :955-1116)."""
    _td = (("reduce_layer0", "Bifusion0", "Rep_p5"), ("reduce_layer1", "Bifusion1", "Rep_p4"),
           ("reduce_layer2", "Bifusion2", "Rep_p3"))
    _bu = (("downsample2", "Rep_n4"), ("downsample1", "Rep_n5"), ("downsample0", "Rep_n6"))

    def _build6(self, c, n, stage):
        self.reduce_layer0 = ConvBNReLU(in_channels=c[5], out_channels=c[6], kernel_size=1, stride=1)
        self.Bifusion0 = BiFusion(in_channels=[c[4], c[6]], out_channels=c[6])
        self.Rep_p5 = stage(c[6], c[6], n[6])
        self.reduce_layer1 = ConvBNReLU(in_channels=c[6], out_channels=c[7], kernel_size=1, stride=1)
        self.Bifusion1 = BiFusion(in_channels=[c[3], c[7]], out_channels=c[7])
        self.Rep_p4 = stage(c[7], c[7], n[7])
        self.reduce_layer2 = ConvBNReLU(in_channels=c[7], out_channels=c[8], kernel_size=1, stride=1)
        self.Bifusion2 = BiFusion(in_channels=[c[2], c[8]], out_channels=c[8])
        self.Rep_p3 = stage(c[8], c[8], n[8])
        self.downsample2 = ConvBNReLU(in_channels=c[8], out_channels=c[8], kernel_size=3, stride=2)
        self.Rep_n4 = stage(c[8] + c[8], c[9], n[9])
        self.downsample1 = ConvBNReLU(in_channels=c[9], out_channels=c[9], kernel_size=3, stride=2)
        self.Rep_n5 = stage(c[7] + c[9], c[10], n[10])
        self.downsample0 = ConvBNReLU(in_channels=c[10], out_channels=c[10], kernel_size=3, stride=2)
        self.Rep_n6 = stage(c[6] + c[10], c[11], n[11])


class RepBiFPANNeck6(_BiFPAN6):
    '''RepBiFPAN neck with a P6 level (YOLOv6-N6/S6).  Reference: reppan.py:394-541.'''

    def __init__(self, channels_list=None, num_repeats=None, block=RepVGGBlock):
        super().__init__()
        assert channels_list is not None
        assert num_repeats is not None
        self._build6(channels_list, num_repeats, lambda i, o, r: RepBlock(in_channels=i, out_channels=o, n=r, block=block))


class CSPRepBiFPANNeck_P6(_BiFPAN6):
    '''CSP RepBiFPAN neck with a P6 level (YOLOv6-M6/L6).  Reference: reppan.py:955-1116.'''

    def __init__(self, channels_list=None, num_repeats=None, block=BottleRep, csp_e=float(1) / 2,
                 stage_block_type="BepC3"):
        super().__init__()
        assert channels_list is not None
        assert num_repeats is not None
        if stage_block_type not in ("BepC3", "MBLABlock"):
            raise NotImplementedError
        stage_block = BepC3 if stage_block_type == "BepC3" else MBLABlock      # reppan.py:559-564, :684-689
        self._build6(channels_list, num_repeats,
                     lambda i, o, r: stage_block(in_channels=i, out_channels=o, n=r, e=csp_e, block=block))


class _PAN(HipModule):
    """Top-down (reduce -> transposed-conv upsample -> cat with the backbone map -> stage), then the same bottom-up path as the
    bi-directional necks.  `_td`: (reduce, upsample, stage) from the deepest level up; `_bu`: (downsample, stage)."""
    _td = ()
    _bu = ()

    @staticmethod
    def _copy_into(pb, src, dst):
        eye = torch.eye(src.C, dtype=torch.float32).view(src.C, src.C, 1, 1)
        with pb.no_quant():
            pb.conv(src, eye, None, stride=1, act=None, out=dst)

    def lower(self, pb, x, out=None):
        if getattr(pb, "is_train", False):
            raise NotImplementedError("yolov6_amd: the uni-directional PAN necks are lowered for inference only")
        feats = [pb.as_nhwc(t) for t in x]            # shallow ... deep
        n_td = len(self._td)
        cur = feats[-1]
        lateral = []
        for i, (reduce_n, up_n, stage_n) in enumerate(self._td):
            reduce, up, stage = getattr(self, reduce_n), getattr(self, up_n), getattr(self, stage_n)
            down = getattr(self, self._bu[n_td - 1 - i][0])
            c_down = down.block.conv.out_channels
            c_lat = reduce.block.conv.out_channels
            cat_bu = pb.new_buffer(cur.B, cur.H, cur.W, c_down + c_lat)      # cat([down_feat, fpn_out_i]) of the way back up
            fpn_out = reduce.lower(pb, cur, out=cat_bu.slice(c_down, c_lat))
            lateral.append((cat_bu, c_down))
            skip = feats[-2 - i]
            cat_td = pb.new_buffer(skip.B, skip.H, skip.W, c_lat + skip.C)   # cat([upsample(fpn_out_i), backbone map])
            up.lower(pb, fpn_out, out=cat_td.slice(0, c_lat))
            self._copy_into(pb, skip, cat_td.slice(c_lat, skip.C))
            cur = stage.lower(pb, cat_td)
        outs = [cur]
        for j, (down_n, stage_n) in enumerate(self._bu):
            cat_bu, c_down = lateral[n_td - 1 - j]
            getattr(self, down_n).lower(pb, cur, out=cat_bu.slice(0, c_down))
            cur = getattr(self, stage_n).lower(pb, cat_bu)
            outs.append(cur)
        return outs


def _csp_stage(stage_block_type, block, csp_e):
    if stage_block_type not in ("BepC3", "MBLABlock"):
        raise NotImplementedError
    stage_block = BepC3 if stage_block_type == "BepC3" else MBLABlock
    return lambda i, o, r: stage_block(in_channels=i, out_channels=o, n=r, e=csp_e, block=block)


class _PAN5(_PAN):
    _td = (("reduce_layer0", "upsample0", "Rep_p4"), ("reduce_layer1", "upsample1", "Rep_p3"))
    _bu = (("downsample2", "Rep_n3"), ("downsample1", "Rep_n4"))

    def _build5(self, c, n, stage):          # attribute order = the reference's (state_dict order)
        self.Rep_p4 = stage(c[3] + c[5], c[5], n[5])
        self.Rep_p3 = stage(c[2] + c[6], c[6], n[6])
        self.Rep_n3 = stage(c[6] + c[7], c[8], n[7])
        self.Rep_n4 = stage(c[5] + c[9], c[10], n[8])
        self.reduce_layer0 = ConvBNReLU(in_channels=c[4], out_channels=c[5], kernel_size=1, stride=1)
        self.upsample0 = Transpose(in_channels=c[5], out_channels=c[5])
        self.reduce_layer1 = ConvBNReLU(in_channels=c[5], out_channels=c[6], kernel_size=1, stride=1)
        self.upsample1 = Transpose(in_channels=c[6], out_channels=c[6])
        self.downsample2 = ConvBNReLU(in_channels=c[6], out_channels=c[7], kernel_size=3, stride=2)
        self.downsample1 = ConvBNReLU(in_channels=c[8], out_channels=c[9], kernel_size=3, stride=2)


class _PAN6(_PAN):
    _td = (("reduce_layer0", "upsample0", "Rep_p5"), ("reduce_layer1", "upsample1", "Rep_p4"),
           ("reduce_layer2", "upsample2", "Rep_p3"))
    _bu = (("downsample2", "Rep_n4"), ("downsample1", "Rep_n5"), ("downsample0", "Rep_n6"))

    def _build6(self, c, n, stage):
        self.reduce_layer0 = ConvBNReLU(in_channels=c[5], out_channels=c[6], kernel_size=1, stride=1)
        self.upsample0 = Transpose(in_channels=c[6], out_channels=c[6])
        self.Rep_p5 = stage(c[4] + c[6], c[6], n[6])
        self.reduce_layer1 = ConvBNReLU(in_channels=c[6], out_channels=c[7], kernel_size=1, stride=1)
        self.upsample1 = Transpose(in_channels=c[7], out_channels=c[7])
        self.Rep_p4 = stage(c[3] + c[7], c[7], n[7])
        self.reduce_layer2 = ConvBNReLU(in_channels=c[7], out_channels=c[8], kernel_size=1, stride=1)
        self.upsample2 = Transpose(in_channels=c[8], out_channels=c[8])
        self.Rep_p3 = stage(c[2] + c[8], c[8], n[8])
        self.downsample2 = ConvBNReLU(in_channels=c[8], out_channels=c[8], kernel_size=3, stride=2)
        self.Rep_n4 = stage(c[8] + c[8], c[9], n[9])
        self.downsample1 = ConvBNReLU(in_channels=c[9], out_channels=c[9], kernel_size=3, stride=2)
        self.Rep_n5 = stage(c[7] + c[9], c[10], n[10])
        self.downsample0 = ConvBNReLU(in_channels=c[10], out_channels=c[10], kernel_size=3, stride=2)
        self.Rep_n6 = stage(c[6] + c[10], c[11], n[11])


class RepPANNeck(_PAN5):
    '''RepPAN neck (YOLOv6 v2.0 N / T / S, `configs/experiment/yolov6t.py`).  Reference: reppan.py:7-129.'''

    def __init__(self, channels_list=None, num_repeats=None, block=RepVGGBlock):
        super().__init__()
        assert channels_list is not None
        assert num_repeats is not None
        self._build5(channels_list, num_repeats, lambda i, o, r: RepBlock(in_channels=i, out_channels=o, n=r, block=block))


class RepPANNeck6(_PAN6):
    '''RepPAN neck with a P6 level.  Reference: reppan.py:240-391.'''

    def __init__(self, channels_list=None, num_repeats=None, block=RepVGGBlock):
        super().__init__()
        assert channels_list is not None
        assert num_repeats is not None
        self._build6(channels_list, num_repeats, lambda i, o, r: RepBlock(in_channels=i, out_channels=o, n=r, block=block))


class CSPRepPANNeck(_PAN5):
    '''CSP RepPAN neck (v2.0 M / L, `configs/experiment/yolov6s_csp_scaled.py`).  Reference: reppan.py:544-663.'''

    def __init__(self, channels_list=None, num_repeats=None, block=BottleRep, csp_e=float(1) / 2, stage_block_type="BepC3"):
        super().__init__()
        assert channels_list is not None
        assert num_repeats is not None
        self._build5(channels_list, num_repeats, _csp_stage(stage_block_type, block, csp_e))


class CSPRepPANNeck_P6(_PAN6):
    '''CSP RepPAN neck with a P6 level.  Reference: reppan.py:788-953.'''

    def __init__(self, channels_list=None, num_repeats=None, block=BottleRep, csp_e=float(1) / 2, stage_block_type="BepC3"):
        super().__init__()
        assert channels_list is not None
        assert num_repeats is not None
        self._build6(channels_list, num_repeats, _csp_stage(stage_block_type, block, csp_e))
