"""IR / IR-SE building blocks of the pSp W+ encoder (reference: psp_encoder_model/encoders/helpers.py)."""
from collections import namedtuple

import torch
import torch.nn.functional as F
from torch import nn

from op._native import amp_fwd as _amp_fwd, amp_bwd as _amp_bwd
from torch.nn import (AdaptiveAvgPool2d, BatchNorm2d, Conv2d, MaxPool2d, Module, ReLU, Sequential, Sigmoid)


class _PReLUFunction(torch.autograd.Function):
    """aten's prelu forward with the backward on the HIP kernel fmgan_prelu_backward_f32 (one pass over x and grad, the
    slope gradient as per-block partial sums): aten's prelu_backward writes two full-size tensors through a multi-output
    elementwise kernel that does not vectorise on NHWC data — 965 us per call at [16,64,256,256], 6.5 % of the
    forward+backward of the 3-encoder path.  When a graph of the backward is requested (create_graph=True) the same
    formulas run as differentiable torch ops."""

    @staticmethod
    @_amp_fwd
    def forward(ctx, x, weight):
        ctx.save_for_backward(x, weight)
        return F.prelu(x, weight)

    @staticmethod
    @_amp_bwd
    def backward(ctx, grad):
        x, weight = ctx.saved_tensors
        if not torch.is_grad_enabled():
            from op import _native
            n, c, h, w = x.shape
            x2 = x.permute(0, 2, 3, 1)
            g2 = grad.contiguous(memory_format=torch.channels_last).permute(0, 2, 3, 1)
            if x2.is_contiguous() and g2.is_contiguous():
                res = _native.prelu_backward(x2.reshape(-1, c), g2.reshape(-1, c), weight.contiguous())
                if res is not None:
                    gx, gw = res
                    return gx.view(n, h, w, c).permute(0, 3, 1, 2), gw
        neg = x <= 0
        a = weight.view(1, -1, 1, 1)
        return torch.where(neg, grad * a, grad), (grad * x * neg).sum((0, 2, 3))


class PReLU(nn.PReLU):
    """nn.PReLU(depth) (same parameter, same state_dict) whose training-time backward on NHWC float32 GPU activations
    runs on the HIP kernel above; everything else is nn.PReLU."""

    def forward(self, input):
        if (input.is_cuda and input.dtype == torch.float32 and input.dim() == 4 and torch.is_grad_enabled()
                and (input.requires_grad or self.weight.requires_grad) and self.weight.numel() == input.shape[1]
                and self.weight.numel() % 4 == 0
                and input.is_contiguous(memory_format=torch.channels_last) and not input.is_contiguous()):
            return _PReLUFunction.apply(input, self.weight)
        return super().forward(input)


class Flatten(Module):
    def forward(self, input):
        return input.view(input.size(0), -1)


def l2_norm(input, axis=1):
    return torch.div(input, torch.norm(input, 2, axis, True))


class Bottleneck(namedtuple('Block', ['in_channel', 'depth', 'stride'])):
    """(in_channel, depth, stride) of one residual unit."""


def get_block(in_channel, depth, num_units, stride=2):
    return [Bottleneck(in_channel, depth, stride)] + [Bottleneck(depth, depth, 1) for _ in range(num_units - 1)]


_UNITS = {18: (2, 2, 2, 2), 50: (3, 4, 14, 3), 100: (3, 13, 30, 3), 152: (3, 8, 36, 3)}


def get_blocks(num_layers):
    """Unit layout per stage (helpers.py:38-73): widths 64/128/256/512, first unit of each stage has stride 2."""
    if num_layers not in _UNITS:
        raise ValueError(f'Invalid number of layers: {num_layers}. Must be one of [18, 50, 100, 152]')
    widths = (64, 64, 128, 256, 512)
    return [get_block(widths[i], widths[i + 1], n) for i, n in enumerate(_UNITS[num_layers])]


class SEModule(Module):
    """Squeeze-and-excitation gate (helpers.py:76-92)."""

    def __init__(self, channels, reduction):
        super().__init__()
        self.avg_pool = AdaptiveAvgPool2d(1)
        self.fc1 = Conv2d(channels, channels // reduction, kernel_size=1, padding=0, bias=False)
        self.relu = ReLU(inplace=True)
        self.fc2 = Conv2d(channels // reduction, channels, kernel_size=1, padding=0, bias=False)
        self.sigmoid = Sigmoid()

    def forward(self, x):
        gate = self.sigmoid(self.fc2(self.relu(self.fc1(self.avg_pool(x)))))
        return x * gate


def _shortcut(in_channel, depth, stride):
    if in_channel == depth:
        return MaxPool2d(1, stride)
    return Sequential(Conv2d(in_channel, depth, (1, 1), stride, bias=False), BatchNorm2d(depth))


class bottleneck_IR(Module):
    """BN-conv3x3-PReLU-conv3x3(stride)-BN + shortcut (helpers.py:95-114)."""

    def __init__(self, in_channel, depth, stride):
        super().__init__()
        self.shortcut_layer = _shortcut(in_channel, depth, stride)
        self.res_layer = Sequential(BatchNorm2d(in_channel),
                                    Conv2d(in_channel, depth, (3, 3), (1, 1), 1, bias=False), PReLU(depth),
                                    Conv2d(depth, depth, (3, 3), stride, 1, bias=False), BatchNorm2d(depth))

    def forward(self, x):
        return self.res_layer(x) + self.shortcut_layer(x)


class bottleneck_IR_SE(Module):
    """bottleneck_IR with an SE gate at the end of the residual branch (helpers.py:117-139)."""

    def __init__(self, in_channel, depth, stride):
        super().__init__()
        self.shortcut_layer = _shortcut(in_channel, depth, stride)
        self.res_layer = Sequential(BatchNorm2d(in_channel),
                                    Conv2d(in_channel, depth, (3, 3), (1, 1), 1, bias=False), PReLU(depth),
                                    Conv2d(depth, depth, (3, 3), stride, 1, bias=False), BatchNorm2d(depth),
                                    SEModule(depth, 16))

    def forward(self, x):
        return self.res_layer(x) + self.shortcut_layer(x)
