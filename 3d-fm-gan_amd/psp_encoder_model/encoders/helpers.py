"""IR / IR-SE building blocks of the pSp W+ encoder (reference: psp_encoder_model/encoders/helpers.py)."""
from collections import namedtuple

import torch
from torch.nn import (AdaptiveAvgPool2d, BatchNorm2d, Conv2d, MaxPool2d, Module, PReLU, ReLU, Sequential, Sigmoid)


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
