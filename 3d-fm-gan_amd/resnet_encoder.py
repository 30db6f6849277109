"""ResNet encoders E_Tsr / E_W of 3D-FM GAN (host PyTorch-ROCm, MIOpen convolutions).

API and state_dict names follow the reference's resnet_encoder.py (a torchvision ResNet fork):
`resnet18(tensor_encoding=..., tensor_transform=...)` -> [N,512,4,4] (AvgPool2d(2,2), for 256^2 inputs) or
[N,512] (global pool + flatten), optionally also a 512-vector from `ten_fc` (resnet_encoder.py:152-283).
Only the BasicBlock depths are provided: the 3-encoder path uses resnet18 (train_3_encoder.py:318-319).
No custom kernel is required for these (north_star); they are dense convs that MIOpen runs on the MFMA units.
"""
import torch
from torch import nn

__all__ = ['ResNet', 'BasicBlock', 'resnet18', 'resnet34']


def conv3x3(in_planes, out_planes, stride=1):
    return nn.Conv2d(in_planes, out_planes, kernel_size=3, stride=stride, padding=1, bias=False)


def conv1x1(in_planes, out_planes, stride=1):
    return nn.Conv2d(in_planes, out_planes, kernel_size=1, stride=stride, bias=False)


class BasicBlock(nn.Module):
    """conv3x3-BN-ReLU-conv3x3-BN + identity/projection, ReLU (resnet_encoder.py:45-91)."""
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None, norm_layer=None):
        super().__init__()
        norm_layer = norm_layer or nn.BatchNorm2d
        self.conv1 = conv3x3(inplanes, planes, stride)
        self.bn1 = norm_layer(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = conv3x3(planes, planes)
        self.bn2 = norm_layer(planes)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        identity = x if self.downsample is None else self.downsample(x)
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        return self.relu(out + identity)


class ResNet(nn.Module):
    def __init__(self, block, layers, zero_init_residual=False, norm_layer=None, tensor_encoding=True,
                 tensor_transform=False):
        super().__init__()
        self._norm_layer = norm_layer or nn.BatchNorm2d
        self.tensor_encoding = tensor_encoding
        self.tensor_transform = tensor_transform
        self.inplanes = 64
        self.conv1 = nn.Conv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = self._norm_layer(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        self.layer1 = self._make_layer(block, 64, layers[0])
        self.layer2 = self._make_layer(block, 128, layers[1], stride=2)
        self.layer3 = self._make_layer(block, 256, layers[2], stride=2)
        self.layer4 = self._make_layer(block, 512, layers[3], stride=2)
        # tensor head: /2 average pool (256^2 input -> 4x4); vector head: global pool (resnet_encoder.py:206-209)
        self.avgpool = nn.AvgPool2d(kernel_size=2, stride=2) if tensor_encoding else nn.AdaptiveAvgPool2d((1, 1))
        if tensor_transform:
            self.ten_fc = nn.Linear(512 * 4 * 4, 512)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
            elif isinstance(m, (nn.BatchNorm2d, nn.GroupNorm)):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
        if zero_init_residual:
            for m in self.modules():
                if isinstance(m, BasicBlock):
                    nn.init.constant_(m.bn2.weight, 0)

    def _make_layer(self, block, planes, blocks, stride=1):
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = nn.Sequential(conv1x1(self.inplanes, planes * block.expansion, stride),
                                       self._norm_layer(planes * block.expansion))
        stages = [block(self.inplanes, planes, stride, downsample, self._norm_layer)]
        self.inplanes = planes * block.expansion
        stages += [block(self.inplanes, planes, norm_layer=self._norm_layer) for _ in range(1, blocks)]
        return nn.Sequential(*stages)

    def forward(self, x):
        x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        x = self.avgpool(x)
        if not self.tensor_encoding:
            x = torch.flatten(x, 1)
        if self.tensor_transform:
            return x, self.ten_fc(torch.flatten(x, 1))
        return x


def resnet18(pretrained=False, progress=True, **kwargs):
    if pretrained:
        raise RuntimeError('pretrained ImageNet weights are a network fetch; load a state_dict instead')
    return ResNet(BasicBlock, [2, 2, 2, 2], **kwargs)


def resnet34(pretrained=False, progress=True, **kwargs):
    if pretrained:
        raise RuntimeError('pretrained ImageNet weights are a network fetch; load a state_dict instead')
    return ResNet(BasicBlock, [3, 4, 6, 3], **kwargs)
