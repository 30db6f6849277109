"""LPIPS perceptual loss, `net-lin` / VGG16 form — the term `lpips_loss_lambda * LPIPS_Loss(g_output, g_ref, lpips_model)`
of the reference's G step (train_3_encoder.py:373-374,529; Util/training_util.py:115-127).

Topology restated from lpips/networks_basic.py:36-121 (PNetLin: ScalingLayer -> VGG16 features cut after relu1_2,
relu2_2, relu3_3, relu4_3, relu5_3 (lpips/pretrained_networks.py:106-141) -> unit-normalise over channels -> squared
difference -> 1x1 conv to one channel -> spatial mean -> sum over the five taps) and lpips/__init__.py:22-53
(PerceptualLoss.forward(pred, target) evaluates net(target, pred)).

Weights: the reference downloads torchvision's ImageNet VGG16 and ships `weights/v0.1/vgg.pth` for the 1x1 layers; neither
is available offline, and the reference's module itself is not importable here (it needs torchvision,
skimage, IPython).  So this module is **load, not a parity row**: same layers, same tensor shapes, same arithmetic per
layer, random initialisation unless `load_state_dict` is given real weights (state_dict names follow the reference:
`net.slice1.0.weight` ... `lin4.model.1.weight`, `scaling_layer.shift/scale`).  It is frozen and in eval mode in the
training step, so only data gradients flow through it.  Host PyTorch-ROCm (MIOpen convs); no custom kernel.
"""
import torch
from torch import nn

# VGG16 `features` indices -> (slice, [(index, cin, cout) convs]); a MaxPool2d(2) opens slices 2..5
_VGG_SLICES = (
    ((0, 3, 64), (2, 64, 64)),
    ((5, 64, 128), (7, 128, 128)),
    ((10, 128, 256), (12, 256, 256), (14, 256, 256)),
    ((17, 256, 512), (19, 512, 512), (21, 512, 512)),
    ((24, 512, 512), (26, 512, 512), (28, 512, 512)),
)


class vgg16(nn.Module):
    """VGG16 conv trunk in five slices; module indices inside a slice are torchvision's `features` indices."""

    def __init__(self, requires_grad=False):
        super().__init__()
        self.N_slices = 5
        for si, convs in enumerate(_VGG_SLICES):
            seq = nn.Sequential()
            if si > 0:
                seq.add_module(str(convs[0][0] - 1), nn.MaxPool2d(kernel_size=2, stride=2))
            for idx, cin, cout in convs:
                seq.add_module(str(idx), nn.Conv2d(cin, cout, kernel_size=3, padding=1))
                seq.add_module(str(idx + 1), nn.ReLU(inplace=True))
            setattr(self, f'slice{si + 1}', seq)
        if not requires_grad:
            self.requires_grad_(False)

    def forward(self, x):
        taps = []
        for si in range(5):
            x = getattr(self, f'slice{si + 1}')(x)
            taps.append(x)
        return taps


class ScalingLayer(nn.Module):
    def __init__(self):
        super().__init__()
        self.register_buffer('shift', torch.tensor([-.030, -.088, -.188])[None, :, None, None])
        self.register_buffer('scale', torch.tensor([.458, .448, .450])[None, :, None, None])

    def forward(self, inp):
        return (inp - self.shift) / self.scale


class NetLinLayer(nn.Module):
    """Dropout (inactive in eval mode) + 1x1 conv without bias, non-negative weights (the reference clamps them)."""

    def __init__(self, chn_in, chn_out=1, use_dropout=False):
        super().__init__()
        layers = [nn.Dropout()] if use_dropout else []
        layers.append(nn.Conv2d(chn_in, chn_out, 1, stride=1, padding=0, bias=False))
        self.model = nn.Sequential(*layers)
        with torch.no_grad():
            self.model[-1].weight.abs_()


def normalize_tensor(in_feat, eps=1e-10):
    return in_feat / (torch.sqrt(torch.sum(in_feat ** 2, dim=1, keepdim=True)) + eps)


class PNetLin(nn.Module):
    chns = (64, 128, 256, 512, 512)

    def __init__(self, use_dropout=True):
        super().__init__()
        self.scaling_layer = ScalingLayer()
        self.net = vgg16(requires_grad=False)
        for i, c in enumerate(self.chns):
            setattr(self, f'lin{i}', NetLinLayer(c, use_dropout=use_dropout))

    def forward(self, in0, in1):
        f0, f1 = self.net(self.scaling_layer(in0)), self.net(self.scaling_layer(in1))
        val = None
        for i in range(len(self.chns)):
            diff = (normalize_tensor(f0[i]) - normalize_tensor(f1[i])) ** 2
            r = getattr(self, f'lin{i}').model(diff).mean([2, 3], keepdim=True)
            val = r if val is None else val + r
        return val


class PerceptualLoss(nn.Module):
    """`lpips.PerceptualLoss(model='net-lin', net='vgg', ...)` of the reference's Module_Fix_Setup; returns [N,1,1,1]."""

    def __init__(self, model='net-lin', net='vgg', colorspace='rgb', spatial=False, use_gpu=True, gpu_ids=(0,)):
        super().__init__()
        if model != 'net-lin' or net not in ('vgg', 'vgg16') or spatial:
            raise ValueError('only the configuration the training step uses is provided: net-lin / vgg, non-spatial')
        self.net = PNetLin(use_dropout=True)
        self.eval().requires_grad_(False)

    def forward(self, pred, target, normalize=False):
        if normalize:
            target, pred = 2 * target - 1, 2 * pred - 1
        return self.net(target, pred)
