"""Host-side weight transforms.  Reference: yolov6/utils/torch_utils.py
(initialize_weights :38-47, fuse_conv_and_bn :50-82, fuse_model :85-94)."""
import torch
import torch.nn as nn


def initialize_weights(model):
    for m in model.modules():
        if type(m) is nn.BatchNorm2d:
            m.eps = 1e-3
            m.momentum = 0.03
        elif type(m) in (nn.Hardswish, nn.LeakyReLU, nn.ReLU, nn.ReLU6, nn.SiLU):
            m.inplace = True


def fuse_conv_and_bn(conv, bn):
    """Conv2d with the BatchNorm folded in (W' = diag(g/sqrt(var+eps)) W, b' = beta - g*mu/sqrt(var+eps) [+ scaled conv bias])."""
    from ..layers.common import fold_conv_bn
    fused = nn.Conv2d(conv.in_channels, conv.out_channels, kernel_size=conv.kernel_size, stride=conv.stride,
                      padding=conv.padding, groups=conv.groups, bias=True).requires_grad_(False).to(conv.weight.device)
    w, b = fold_conv_bn(conv.weight, conv.bias, bn)
    fused.weight.copy_(w.to(fused.weight.dtype))
    fused.bias.copy_(b.to(fused.bias.dtype))
    return fused


def fuse_model(model):
    """Fold BN into every ConvModule (exact type match, like the reference)."""
    from ..layers.common import ConvModule
    for m in model.modules():
        if type(m) is ConvModule and hasattr(m, "bn"):
            m.conv = fuse_conv_and_bn(m.conv, m.bn)
            delattr(m, "bn")
    _invalidate(model)
    return model


def switch_to_deploy(model):
    """The deploy transform the reference callers apply after fuse_model
    (evaler.py:70-73, inferer.py model_switch): re-parameterise every RepVGG-style block."""
    from ..layers.common import RepVGGBlock
    for m in model.modules():
        if isinstance(m, RepVGGBlock):
            m.switch_to_deploy()
    _invalidate(model)
    return model


def _invalidate(model):
    from ..layers.common import _STRUCTURE_GENERATION
    _STRUCTURE_GENERATION[0] += 1
    for m in model.modules():
        for k in ("_y6_plans", "_y6_fast", "_y6_train_graphs", "_y6_arena"):
            m.__dict__.pop(k, None)
