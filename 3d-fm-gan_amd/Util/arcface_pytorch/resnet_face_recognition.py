"""ArcFace identity network of the face-id loss (host PyTorch-ROCm, MIOpen convolutions).

The training step compares `face_rec_model(gray(output))` with `face_rec_model(gray(reference))`
(Util/training_util.py:178-205, train_3_encoder.py:535); the model is `resnet_face18(use_se=False)`
(Util/training_util.py:160, Util/arcface_pytorch/resnet_face_recognition.py:170-238,350-352): a 1-channel 128^2 input,
stem conv-BN-PReLU-maxpool, four stages of two pre-activation IR blocks (BN-conv-BN-PReLU-conv-BN [+SE] + shortcut, PReLU),
BN-dropout-flatten-fc(512*8*8 -> 512)-BN1d.  State_dict names follow the reference so `resnet18_arcfacenet.pth` loads
(the blob is absent offline, .MISSING_LARGE_BLOBS: the bench uses the reference's own initialisation).
Only the IR-block / depth-18 form the loss uses is provided; it is frozen, so it never needs weight gradients.
"""
import torch
from torch import nn

__all__ = ['ResNetFace', 'IRBlock', 'SEBlock', 'resnet_face18']


class SEBlock(nn.Module):
    def __init__(self, channel, reduction=16):
        super().__init__()
        self.avg_pool = nn.AdaptiveAvgPool2d(1)
        self.fc = nn.Sequential(nn.Linear(channel, channel // reduction), nn.PReLU(),
                                nn.Linear(channel // reduction, channel), nn.Sigmoid())

    def forward(self, x):
        gate = self.fc(self.avg_pool(x).flatten(1))
        return x * gate[:, :, None, None]


class IRBlock(nn.Module):
    """One PReLU module serves both activations of the block (shared slope), as in the reference (:80-116)."""
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None, use_se=True):
        super().__init__()
        self.bn0 = nn.BatchNorm2d(inplanes)
        self.conv1 = nn.Conv2d(inplanes, inplanes, 3, 1, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(inplanes)
        self.prelu = nn.PReLU()
        self.conv2 = nn.Conv2d(inplanes, planes, 3, stride, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample
        self.stride = stride
        self.use_se = use_se
        if use_se:
            self.se = SEBlock(planes)

    def forward(self, x):
        shortcut = x if self.downsample is None else self.downsample(x)
        y = self.prelu(self.bn1(self.conv1(self.bn0(x))))
        y = self.bn2(self.conv2(y))
        if self.use_se:
            y = self.se(y)
        return self.prelu(y + shortcut)


class ResNetFace(nn.Module):
    def __init__(self, block, layers, use_se=True):
        super().__init__()
        self.inplanes = 64
        self.use_se = use_se
        self.conv1 = nn.Conv2d(1, 64, kernel_size=3, padding=1, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.prelu = nn.PReLU()
        self.maxpool = nn.MaxPool2d(kernel_size=2, stride=2)
        self.layer1 = self._make_layer(block, 64, layers[0])
        self.layer2 = self._make_layer(block, 128, layers[1], stride=2)
        self.layer3 = self._make_layer(block, 256, layers[2], stride=2)
        self.layer4 = self._make_layer(block, 512, layers[3], stride=2)
        self.bn4 = nn.BatchNorm2d(512)
        self.dropout = nn.Dropout()
        self.fc5 = nn.Linear(512 * 8 * 8, 512)
        self.bn5 = nn.BatchNorm1d(512)
        for m in self.modules():        # the reference's initialisation (:189-198)
            if isinstance(m, (nn.Conv2d, nn.Linear)):
                nn.init.xavier_normal_(m.weight)
                if getattr(m, 'bias', None) is not None:
                    nn.init.constant_(m.bias, 0)
            elif isinstance(m, (nn.BatchNorm2d, nn.BatchNorm1d)):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def _make_layer(self, block, planes, blocks, stride=1):
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = nn.Sequential(nn.Conv2d(self.inplanes, planes * block.expansion, 1, stride, bias=False),
                                       nn.BatchNorm2d(planes * block.expansion))
        stage = [block(self.inplanes, planes, stride, downsample, use_se=self.use_se)]
        self.inplanes = planes
        stage += [block(planes, planes, use_se=self.use_se) for _ in range(1, blocks)]
        return nn.Sequential(*stage)

    def forward(self, x):
        x = self.maxpool(self.prelu(self.bn1(self.conv1(x))))
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        x = self.dropout(self.bn4(x))
        return self.bn5(self.fc5(torch.flatten(x, 1)))


def resnet_face18(use_se=True, **kwargs):
    return ResNetFace(IRBlock, [2, 2, 2, 2], use_se=use_se, **kwargs)
