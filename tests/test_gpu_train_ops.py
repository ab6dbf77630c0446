"""Training-mode forward of the conv + BatchNorm blocks (batch statistics) through the HIP primitives
(y6_bn_stats, y6_bn_apply around the conv kernels) against torch's own training-mode modules on the CPU
(the same arithmetic the reference runs: common.py:44-49, :250-255).  Inputs and conv weights are fp16-representable,
activations are stored in fp16 on the HIP path: 3e-3 relative."""
import copy

import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

from tests.helpers import rel_err
from yolov6_amd.layers import common
from yolov6_amd.layers.train_ops import conv_module_train_forward, repvgg_train_forward

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _randomise(m, seed):
    g = torch.Generator().manual_seed(seed)
    for mod in m.modules():
        if isinstance(mod, nn.Conv2d):
            fan = mod.weight[0].numel()
            mod.weight.data = (torch.randn(mod.weight.shape, generator=g) / fan ** 0.5).half().float()
        if isinstance(mod, nn.BatchNorm2d):
            mod.weight.data = torch.rand(mod.weight.shape, generator=g) + 0.5
            mod.bias.data = torch.randn(mod.bias.shape, generator=g) * 0.1
            mod.running_mean.data = torch.randn(mod.running_mean.shape, generator=g) * 0.1
            mod.running_var.data = torch.rand(mod.running_var.shape, generator=g) + 0.5
            mod.momentum, mod.eps = 0.03, 1e-3          # initialize_weights (torch_utils.py:38-47)


def _bn_train(x, bn):
    return F.batch_norm(x, bn.running_mean, bn.running_var, bn.weight, bn.bias, True, bn.momentum, bn.eps)


def _check_stats(hip_bn, ref_bn):
    assert rel_err(hip_bn.running_mean.cpu().numpy(), ref_bn.running_mean.numpy()) < 2e-3
    assert rel_err(hip_bn.running_var.cpu().numpy(), ref_bn.running_var.numpy()) < 2e-3
    assert int(hip_bn.num_batches_tracked) == 1     # nn.BatchNorm2d.forward counts the step (the functional reference does not)


@pytest.mark.parametrize("cin,cout,stride,hw", [(64, 64, 1, (20, 24)), (32, 64, 2, (24, 24)), (128, 128, 1, (10, 14))])
def test_repvgg_train_forward(cin, cout, stride, hw):
    m = common.RepVGGBlock(cin, cout, 3, stride).train()
    _randomise(m, 3)
    ref = copy.deepcopy(m)
    x = (torch.randn(4, cin, *hw, generator=torch.Generator().manual_seed(4))).half().float()
    # the reference arithmetic (common.py:250-255) with torch's training-mode batch_norm, fp32 on the CPU
    with torch.no_grad():
        d = _bn_train(F.conv2d(x, ref.rbr_dense.conv.weight, None, stride, 1), ref.rbr_dense.bn)
        e = _bn_train(F.conv2d(x, ref.rbr_1x1.conv.weight, None, stride, 0), ref.rbr_1x1.bn)
        y_ref = d + e
        if ref.rbr_identity is not None:
            y_ref = y_ref + _bn_train(x, ref.rbr_identity)
        y_ref = F.relu(y_ref)
    m = m.to(DEV)
    with torch.no_grad():
        y = repvgg_train_forward(m, x.to(DEV).half())
    torch.cuda.synchronize()
    assert y.shape == y_ref.shape and y.dtype == torch.float16
    assert rel_err(y.float().cpu().numpy(), y_ref.numpy()) < 3e-3
    _check_stats(m.rbr_dense.bn, ref.rbr_dense.bn)
    _check_stats(m.rbr_1x1.bn, ref.rbr_1x1.bn)
    if ref.rbr_identity is not None:
        _check_stats(m.rbr_identity, ref.rbr_identity)


@pytest.mark.parametrize("k,stride,act", [(3, 1, "silu"), (1, 1, "relu"), (3, 2, "relu")])
def test_conv_module_train_forward(k, stride, act):
    m = common.ConvModule(48, 96, k, stride, act).train()
    _randomise(m, 5)
    ref = copy.deepcopy(m)
    x = torch.randn(3, 48, 18, 22, generator=torch.Generator().manual_seed(6)).half().float()
    with torch.no_grad():
        y_ref = _bn_train(F.conv2d(x, ref.conv.weight, None, stride, k // 2), ref.bn)
        y_ref = F.silu(y_ref) if act == "silu" else F.relu(y_ref)
    m = m.to(DEV)
    with torch.no_grad():
        y = conv_module_train_forward(m, x.to(DEV).half())
    torch.cuda.synchronize()
    assert rel_err(y.float().cpu().numpy(), y_ref.numpy()) < 3e-3
    _check_stats(m.bn, ref.bn)


def test_train_forward_refuses_autograd():
    m = common.ConvModule(16, 16, 3, 1, "relu").train().to(DEV)
    with pytest.raises(NotImplementedError):
        conv_module_train_forward(m, torch.zeros(1, 16, 8, 8, device=DEV).half())
