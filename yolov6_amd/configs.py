"""Model configurations and an `addict`-free attribute dict.

The reference describes models with mmcv-style `.py` files (configs/*.py) loaded by
yolov6/utils/config.py:33-63 (which needs the `addict` package).  `load_config(path)`
executes such a file unchanged; `get_config(name)` returns the built-in restatement of the
BASELINE configurations' `model` dicts (so nothing under /root/reference is needed at run
time): yolov6n / yolov6s (configs/yolov6n.py, yolov6s.py), yolov6m / yolov6l
(configs/yolov6m.py, yolov6l.py), yolov6l6 (configs/yolov6l6.py) yolov6s_qa
(configs/qarepvgg/yolov6s_qa.py) and yolov6{s,m,l,x}_mbla (configs/mbla/).
"""
import copy
import os


class ConfigDict(dict):
    """dict with attribute access, recursively (what the reference gets from addict.Dict)."""

    def __init__(self, *a, **kw):
        super().__init__()
        for k, v in dict(*a, **kw).items():
            self[k] = v

    def __setitem__(self, k, v):
        super().__setitem__(k, ConfigDict(v) if isinstance(v, dict) and not isinstance(v, ConfigDict) else v)

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(f"'ConfigDict' object has no attribute '{k}'") from None

    def __setattr__(self, k, v):
        self[k] = v

    def __deepcopy__(self, memo):
        return ConfigDict({k: copy.deepcopy(v, memo) for k, v in self.items()})


class Config(ConfigDict):
    @staticmethod
    def fromfile(path):
        return load_config(path)


def load_config(path):
    """Execute a reference-style config file and return its public names as a Config."""
    path = os.path.abspath(os.path.expanduser(path))
    if not path.endswith(".py"):
        raise IOError("Only .py type are supported now!")
    ns = {"__file__": path}
    with open(path) as f:
        exec(compile(f.read(), path, "exec"), ns)
    cfg = Config({k: v for k, v in ns.items() if not k.startswith("__") and not callable(v) and
                  not isinstance(v, type(os))})
    cfg.setdefault("training_mode", "repvgg")   # tools/train.py:99-100
    return cfg


_ANCHORS = [[10, 13, 19, 19, 33, 23], [30, 61, 59, 59, 59, 119], [116, 90, 185, 185, 373, 326]]


def _p5(depth, width, backbone, neck, csp, iou, use_dfl=False, reg_max=0, cspsppf=True, extra=None):
    b = dict(type=backbone, num_repeats=[1, 6, 12, 18, 6], out_channels=[64, 128, 256, 512, 1024], fuse_P2=True,
             cspsppf=cspsppf)
    n = dict(type=neck, num_repeats=[12, 12, 12, 12], out_channels=[256, 128, 128, 256, 256, 512])
    if csp is not None:
        b["csp_e"] = csp
        n["csp_e"] = csp
    m = dict(pretrained=None, depth_multiple=depth, width_multiple=width, backbone=b, neck=n,
             head=dict(type='EffiDeHead', in_channels=[128, 256, 512], num_layers=3, begin_indices=24, anchors=3,
                       anchors_init=_ANCHORS, out_indices=[17, 20, 23], strides=[8, 16, 32], atss_warmup_epoch=0,
                       iou_type=iou, use_dfl=use_dfl, reg_max=reg_max,
                       distill_weight={'class': 1.0, 'dfl': 1.0}))
    if extra:
        m.update(extra)
    return m


_MODELS = {
    "yolov6n": dict(model=dict(type='YOLOv6n', **_p5(0.33, 0.25, 'EfficientRep', 'RepBiFPANNeck', None, 'siou')),
                    training_mode="repvgg"),
    "yolov6s": dict(model=dict(type='YOLOv6s', **_p5(0.33, 0.50, 'EfficientRep', 'RepBiFPANNeck', None, 'giou')),
                    training_mode="repvgg"),
    "yolov6s_qa": dict(model=dict(type='YOLOv6s', **_p5(0.33, 0.50, 'EfficientRep', 'RepBiFPANNeck', None, 'giou')),
                       training_mode="qarepvggv2"),
    "yolov6m": dict(model=dict(type='YOLOv6m', **_p5(0.60, 0.75, 'CSPBepBackbone', 'CSPRepBiFPANNeck', float(2) / 3,
                                                      'giou', use_dfl=True, reg_max=16, cspsppf=False)),
                    training_mode="repvgg"),
    "yolov6l": dict(model=dict(type='YOLOv6l', **_p5(1.0, 1.0, 'CSPBepBackbone', 'CSPRepBiFPANNeck', float(1) / 2,
                                                      'giou', use_dfl=True, reg_max=16, cspsppf=False)),
                    training_mode="conv_silu"),
    # *_mbla (configs/mbla/yolov6{s,m,l,x}_mbla.py): CSP backbone / neck whose stage block is the MBLABlock
    **{f"yolov6{k}_mbla": dict(model=dict(
        type=f'YOLOv6{k}_mbla', pretrained=None, depth_multiple=d, width_multiple=w,
        backbone=dict(type='CSPBepBackbone', num_repeats=[1, 4, 8, 8, 4], out_channels=[64, 128, 256, 512, 1024],
                      csp_e=float(1) / 2, fuse_P2=True, stage_block_type="MBLABlock"),
        neck=dict(type='CSPRepBiFPANNeck', num_repeats=[8, 8, 8, 8], out_channels=[256, 128, 128, 256, 256, 512],
                  csp_e=float(1) / 2, stage_block_type="MBLABlock"),
        head=dict(type='EffiDeHead', in_channels=[128, 256, 512], num_layers=3, begin_indices=24, anchors=3,
                  anchors_init=_ANCHORS, out_indices=[17, 20, 23], strides=[8, 16, 32], atss_warmup_epoch=0,
                  iou_type='giou', use_dfl=True, reg_max=16, distill_weight={'class': 2.0, 'dfl': 1.0})),
        training_mode="conv_silu") for k, d, w in (("s", 0.5, 0.5), ("m", 0.5, 0.75), ("l", 0.5, 1.0), ("x", 1.0, 1.0))},
    "yolov6l6": dict(model=dict(
        type='YOLOv6l6', pretrained=None, depth_multiple=1.0, width_multiple=1.0,
        backbone=dict(type='CSPBepBackbone_P6', num_repeats=[1, 6, 12, 18, 6, 6],
                      out_channels=[64, 128, 256, 512, 768, 1024], csp_e=float(1) / 2, fuse_P2=True),
        neck=dict(type='CSPRepBiFPANNeck_P6', num_repeats=[12, 12, 12, 12, 12, 12],
                  out_channels=[512, 256, 128, 256, 512, 1024], csp_e=float(1) / 2),
        head=dict(type='EffiDeHead', in_channels=[128, 256, 512, 1024], num_layers=4, anchors=1,
                  strides=[8, 16, 32, 64], atss_warmup_epoch=4, iou_type='giou', use_dfl=True, reg_max=16,
                  distill_weight={'class': 1.0, 'dfl': 1.0})), training_mode="conv_silu"),
}


def _p6(name, depth, width, backbone, neck, csp, iou, use_dfl, reg_max, extra_backbone=None):
    """configs/yolov6{n,s,m}6.py: the 1280-pixel models with a stride-64 level."""
    b = dict(type=backbone, num_repeats=[1, 6, 12, 18, 6, 6], out_channels=[64, 128, 256, 512, 768, 1024], fuse_P2=True)
    n = dict(type=neck, num_repeats=[12, 12, 12, 12, 12, 12], out_channels=[512, 256, 128, 256, 512, 1024])
    if csp is not None:
        b["csp_e"] = csp
        n["csp_e"] = csp
    b.update(extra_backbone or {})
    return dict(type=name, pretrained=None, depth_multiple=depth, width_multiple=width, backbone=b, neck=n,
                head=dict(type='EffiDeHead', in_channels=[128, 256, 512, 1024], num_layers=4, anchors=1,
                          strides=[8, 16, 32, 64], atss_warmup_epoch=4, iou_type=iou, use_dfl=use_dfl, reg_max=reg_max,
                          distill_weight={'class': 1.0, 'dfl': 1.0}))


# configs/qarepvgg/yolov6{n,m}_qa.py (yolov6s_qa above): the N / M graphs with QARepVGGBlockV2
_MODELS.update({
    "yolov6n_qa": dict(model=dict(type='YOLOv6n', **_p5(0.33, 0.25, 'EfficientRep', 'RepBiFPANNeck', None, 'siou')),
                       training_mode="qarepvggv2"),
    "yolov6m_qa": dict(model=dict(type='YOLOv6m', **_p5(0.60, 0.75, 'CSPBepBackbone', 'CSPRepBiFPANNeck', float(2) / 3, 'giou',
                                                         use_dfl=True, reg_max=16, cspsppf=False)), training_mode="qarepvggv2"),
})
# configs/base/yolov6{n,s,m,l}_base.py: plain ConvBNReLU blocks ("conv_relu"), DFL head; N on the N / S graph, S / M / L on CSP
_MODELS.update({
    f"yolov6{k}_base": dict(model=dict(type=f'YOLOv6{k}_base', **_p5(d, w, bb, nk, csp, 'giou', use_dfl=True, reg_max=16,
                                                                     cspsppf=sp)), training_mode="conv_relu")
    for k, d, w, bb, nk, csp, sp in (("n", 0.33, 0.25, 'EfficientRep', 'RepBiFPANNeck', None, True),
                                     ("s", 0.70, 0.50, 'CSPBepBackbone', 'CSPRepBiFPANNeck', float(1) / 2, True),
                                     ("m", 0.80, 0.75, 'CSPBepBackbone', 'CSPRepBiFPANNeck', float(1) / 2, False),
                                     ("l", 1.0, 1.0, 'CSPBepBackbone', 'CSPRepBiFPANNeck', float(1) / 2, False))})


def _v2(name, depth, width, backbone, neck, csp, iou):
    """configs/experiment/yolov6t.py, yolov6s_csp_scaled.py: v2.0-style models (three backbone maps, uni-directional PAN neck)."""
    b = dict(type=backbone, num_repeats=[1, 6, 12, 18, 6], out_channels=[64, 128, 256, 512, 1024])
    n = dict(type=neck, num_repeats=[12, 12, 12, 12], out_channels=[256, 128, 128, 256, 256, 512])
    if csp is not None:
        b["csp_e"] = csp
        n["csp_e"] = csp
    return dict(type=name, pretrained=None, depth_multiple=depth, width_multiple=width, backbone=b, neck=n,
                head=dict(type='EffiDeHead', in_channels=[128, 256, 512], num_layers=3, begin_indices=24, anchors=1,
                          out_indices=[17, 20, 23], strides=[8, 16, 32], iou_type=iou, use_dfl=False, reg_max=0))


_MODELS.update({
    "yolov6t": dict(model=_v2('YOLOv6t', 0.33, 0.375, 'EfficientRep', 'RepPANNeck', None, 'siou'), training_mode="repvgg"),
    "yolov6s_csp": dict(model=_v2('YOLOv6s_csp', 0.70, 0.50, 'CSPBepBackbone', 'CSPRepPANNeck', float(1) / 2, 'giou'),
                        training_mode="repvgg"),
})
_MODELS.update({
    "yolov6n6": dict(model=_p6('YOLOv6n6', 0.33, 0.25, 'EfficientRep6', 'RepBiFPANNeck6', None, 'siou', False, 0,
                               dict(cspsppf=True)), training_mode="repvgg"),
    "yolov6s6": dict(model=_p6('YOLOv6s6', 0.33, 0.50, 'EfficientRep6', 'RepBiFPANNeck6', None, 'giou', False, 0,
                               dict(cspsppf=True)), training_mode="repvgg"),
    "yolov6m6": dict(model=_p6('YOLOv6m6', 0.60, 0.75, 'CSPBepBackbone_P6', 'CSPRepBiFPANNeck_P6', float(2) / 3, 'giou', True, 16),
                     training_mode="repvgg"),
})


def get_config(name):
    """Built-in model config by name ('yolov6s', ...)."""
    if name not in _MODELS:
        raise KeyError(f"unknown config {name!r}; known: {sorted(_MODELS)}")
    return Config(copy.deepcopy(_MODELS[name]))


def tiny_config(width=0.125, depth=0.17, training_mode="repvgg", p6=False):
    """A narrow/shallow variant of the N/S (or L6) graph for fast parity tests."""
    cfg = get_config("yolov6l6" if p6 else "yolov6s")
    cfg.model.width_multiple = width
    cfg.model.depth_multiple = depth
    cfg.training_mode = training_mode
    return cfg
