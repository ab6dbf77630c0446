"""Backbones with the reference's class names, constructor arguments and state_dict keys
(yolov6/models/efficientrep.py), lowered onto the HIP plan.

All four backbones share one shape: a stride-2 stem, then ERBlock_k = [stride-2 block,
stage block(, channel-merge layer on the last one)].  They differ in the stage block
(RepBlock vs BepC3) and in how many pyramid levels they emit.
"""
from torch import nn

from ..layers.common import (BepC3, ConvBNSiLU, CSPSPPF, HipModule, MBLABlock, RepBlock, RepVGGBlock, SimCSPSPPF, SimSPPF, SPPF)


def _merge_layer(block, cspsppf):
    if cspsppf:
        return CSPSPPF if block == ConvBNSiLU else SimCSPSPPF
    return SPPF if block == ConvBNSiLU else SimSPPF


class _PyramidBackbone(HipModule):
    """stem + ERBlock_2..ERBlock_{last}; `first_out` is the first ERBlock whose output is returned."""
    last = 5

    def _build(self, in_channels, channels_list, num_repeats, block, make_stage, cspsppf, fuse_P2):
        assert channels_list is not None
        assert num_repeats is not None
        self.fuse_P2 = fuse_P2
        self.stem = block(in_channels=in_channels, out_channels=channels_list[0], kernel_size=3, stride=2)
        for k in range(2, self.last + 1):
            cin, cout = channels_list[k - 2], channels_list[k - 1]
            layers = [block(in_channels=cin, out_channels=cout, kernel_size=3, stride=2),
                      make_stage(cout, num_repeats[k - 1])]
            if k == self.last:
                layers.append(_merge_layer(block, cspsppf)(in_channels=cout, out_channels=cout, kernel_size=5))
            setattr(self, f"ERBlock_{k}", nn.Sequential(*layers))

    def _first_out(self):
        return 2 if self.fuse_P2 else 3

    def lower(self, pb, x, out=None):
        outs = []
        trace = getattr(pb, "trace", None)      # optional {name: activation view}, filled for tools/train_trace.py
        hint = getattr(pb, "hint_single_use", None)
        if hint is not None and trace is None:
            hint()            # the stem's output feeds ERBlock_2[0] only: stem + first stride-2 block may run as one fused kernel
        x = self.stem.lower(pb, x)
        if trace is not None:
            trace["backbone.stem"] = x
        for k in range(2, self.last + 1):
            for i, layer in enumerate(getattr(self, f"ERBlock_{k}")):
                x = layer.lower(pb, x)
                if trace is not None:
                    trace[f"backbone.ERBlock_{k}.{i}"] = x
            if k >= self._first_out():
                outs.append(x)
        return tuple(outs)


class EfficientRep(_PyramidBackbone):
    '''EfficientRep backbone (YOLOv6-N/S).  Reference: efficientrep.py:7-118.'''

    def __init__(self, in_channels=3, channels_list=None, num_repeats=None, block=RepVGGBlock, fuse_P2=False,
                 cspsppf=False):
        super().__init__()
        self._build(in_channels, channels_list, num_repeats, block,
                    lambda c, n: RepBlock(in_channels=c, out_channels=c, n=n, block=block), cspsppf, fuse_P2)


class EfficientRep6(EfficientRep):
    '''EfficientRep with a P6 level (N6/S6).  Reference: efficientrep.py:121-247.'''
    last = 6


class CSPBepBackbone(_PyramidBackbone):
    '''CSPBep backbone (YOLOv6-M/L).  Reference: efficientrep.py:250-374.'''

    def __init__(self, in_channels=3, channels_list=None, num_repeats=None, block=RepVGGBlock, csp_e=float(1) / 2,
                 fuse_P2=False, cspsppf=False, stage_block_type="BepC3"):
        super().__init__()
        if stage_block_type not in ("BepC3", "MBLABlock"):
            raise NotImplementedError
        stage_block = BepC3 if stage_block_type == "BepC3" else MBLABlock      # efficientrep.py:271-276
        self._build(in_channels, channels_list, num_repeats, block,
                    lambda c, n: stage_block(in_channels=c, out_channels=c, n=n, e=csp_e, block=block), cspsppf, fuse_P2)


class CSPBepBackbone_P6(CSPBepBackbone):
    '''CSPBep backbone with a P6 level (YOLOv6-M6/L6).  Reference: efficientrep.py:377-516.
    The reference P6 variant always returns ERBlock_2 (its forward ignores fuse_P2, :501-516).'''
    last = 6

    def _first_out(self):
        return 2
