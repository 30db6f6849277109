"""pSp W+ encoder E_W_Plus (reference: psp_encoder_model/encoders/psp_encoders.py:20-132).

IR-SE backbone with a 3-level feature pyramid; each of the n_styles heads reduces its pyramid level to 1x1 with
stride-2 convs and maps it through an EqualLinear.  Host PyTorch-ROCm (MIOpen); same constructor and state_dict
names as the reference.  On the GPU the encoder runs in channels_last (NHWC) memory format: MIOpen's fp32 implicit-GEMM
kernels are NHWC-native, so this removes ~260 layout-transpose launches per forward and makes the FPN's bilinear
resize 13x faster (measured on MI355X: 14.9 -> 10.7 ms at B=8, 51.3 -> 32.0 ms at B=32; identical values).  The two Backbone* encoders of the reference file are not on the 3-encoder path and are
not provided.
"""
import math
import os

import torch
import torch.nn.functional as F
from torch import nn
from torch.nn import BatchNorm2d, Conv2d, Module, Sequential

from .helpers import PReLU, bottleneck_IR, bottleneck_IR_SE, get_blocks
from stylegan2 import EqualLinear
from op import fused_leaky_relu
from op.live_weights import LiveWeights
from Util.streams import overlap_ok, run_deferred, side_streams

# Inference: the style heads are independent of each other and each ends in a tail of tiny launches (conv at 16^2 ... 1^2
# + bias + LeakyReLU, ~25 kernels of a few microseconds) that cannot fill 256 CUs; heads are dealt round-robin onto
# this many side streams so the tails overlap other heads' large first convs, the pyramid's lateral layers and — through
# forward_deferred — the synthesis network.  2 streams measured best (more streams than hardware queues serialise
# behind each other: 4 streams 325, 3 streams 331-340, 2 streams 337-340 pairs/s).
HEAD_STREAMS = int(os.environ.get('FMGAN_PSP_STREAMS', '2'))
# Optional (FMGAN_PSP_GROUPED=1): the heads of one pyramid level (3 coarse, 4 middle, up to 11 fine) have the same
# structure and, after their first conv, the same shapes — their tails can run as ONE chain of grouped convs (groups =
# heads of the level) after one wide first conv (Cout = heads*512) on the level's shared feature map; 98 convs + 98
# bias/LeakyReLU launches become 15 + 15, with the heads' parameters re-pointed once at slices of one buffer per
# (level, depth) (_flatten_heads) so that the wide weight IS the live parameters.  Measured (MI355X, B=8, MIOpen fp32,
# tools/exp_grouped_heads.py): standalone the 11 fine heads' tails drop from 2.28 to 1.54 ms (8^2->4^2 0.312 -> 0.139,
# 4^2->2^2 0.318 -> 0.062, 2^2->1^2 0.297 -> 0.054: launch-latency-bound when separate; the 64^2->32^2 first convs run at
# 133 TFLOP/s either way) — but IN A STEP the grouped form is 2-3 % SLOWER (pairs1024 348 -> 340 pairs/s, pairs256
# 583 -> 574): the per-head tails were already hidden under the synthesis network by the two head streams, whereas the
# wide convs compete with its MFMA kernels for every CU and hold back all latents of a level until its last head is
# done.  The per-head form therefore stays the default.
GROUP_HEADS = os.environ.get('FMGAN_PSP_GROUPED', '0') == '1'

_TAPS = {18: (3, 5, 7), 50: (6, 20, 23)}   # units whose outputs feed the pyramid (psp_encoders.py:105-108)


class GradualStyleBlock(Module):
    """log2(spatial) x (conv3x3/2 + LeakyReLU) down to 1x1, then EqualLinear (psp_encoders.py:20-41)."""

    def __init__(self, in_c, out_c, spatial):
        super().__init__()
        self.out_c = out_c
        self.spatial = spatial
        layers, c = [], in_c
        for _ in range(int(math.log2(spatial))):
            layers += [Conv2d(c, out_c, kernel_size=3, stride=2, padding=1), nn.LeakyReLU()]
            c = out_c
        self.convs = nn.Sequential(*layers)
        self.linear = EqualLinear(out_c, out_c, lr_mul=1)

    def forward(self, x):
        if x.is_cuda and x.dtype == torch.float32:
            # conv, then bias + LeakyReLU(0.01) in ONE pass of the HIP fused_bias_act kernel (the same arithmetic as
            # MIOpen's separate bias add followed by aten leaky_relu: (v + b) > 0 ? . : 0.01 * .  — bit-identical).
            for conv, act in zip(self.convs[0::2], self.convs[1::2]):
                y = F.conv2d(x, conv.weight, None, conv.stride, conv.padding)
                if y.is_contiguous(memory_format=torch.channels_last) and not y.is_contiguous():
                    n, c, h, w = y.shape                       # NHWC storage: channel is the fastest dimension
                    y = fused_leaky_relu(y.permute(0, 2, 3, 1).reshape(-1, c), conv.bias, act.negative_slope, 1.0)
                    x = y.view(n, h, w, c).permute(0, 3, 1, 2)
                else:
                    x = fused_leaky_relu(y, conv.bias, act.negative_slope, 1.0)
            return self.linear(x.reshape(-1, self.out_c))
        return self.linear(self.convs(x).reshape(-1, self.out_c))


class GradualStyleEncoder(Module):
    def __init__(self, num_layers, mode='ir', opts=None):
        super().__init__()
        assert num_layers in [18, 50, 100, 152], 'num_layers should be 18, 50, 100, or 152'
        assert mode in ['ir', 'ir_se'], 'mode should be ir or ir_se'
        self.num_layers = num_layers
        unit = bottleneck_IR if mode == 'ir' else bottleneck_IR_SE
        self.input_layer = Sequential(Conv2d(opts.input_nc, 64, (3, 3), 1, 1, bias=False), BatchNorm2d(64), PReLU(64))
        self.body = Sequential(*[unit(b.in_channel, b.depth, b.stride) for stage in get_blocks(num_layers) for b in stage])
        self.style_count = opts.n_styles
        self.coarse_ind = 3
        self.middle_ind = 7
        self.styles = nn.ModuleList()
        for i in range(self.style_count):
            spatial = 16 if i < self.coarse_ind else (32 if i < self.middle_ind else 64)
            self.styles.append(GradualStyleBlock(512, 512, spatial))
        self.latlayer1 = nn.Conv2d(256, 512, kernel_size=1, stride=1, padding=0)
        self.latlayer2 = nn.Conv2d(128, 512, kernel_size=1, stride=1, padding=0)
        self.channels_last = True     # GPU only; set False to keep NCHW activations
        self._cl_key = None
        self._live_weights = None
        self._flat = None             # per level: (head indices, [(wide weight, wide bias) per depth])

    def _to_channels_last(self):
        """Re-lay the conv weights as NHWC once per (device, parameter storage); values and shapes are untouched."""
        w = self.input_layer[0].weight
        key = (w.device, w.data_ptr())
        if self._cl_key != key:
            for m in self.modules():
                if isinstance(m, nn.Conv2d):
                    m.weight.data = m.weight.data.contiguous(memory_format=torch.channels_last)
            self._cl_key = (w.device, self.input_layer[0].weight.data_ptr())

    def _levels(self):
        n = self.style_count
        return [list(range(0, min(self.coarse_ind, n))), list(range(self.coarse_ind, min(self.middle_ind, n))),
                list(range(self.middle_ind, n))]

    def _flatten_heads(self):
        """One buffer per (pyramid level, conv depth) holding the heads' conv weights back to back ([G*512,512,3,3] in the
        current memory format) and one for the biases; every head's parameter is re-pointed at its slice, so the wide
        tensors are the live parameters themselves (optimizer steps, EMA `.data` updates and `load_state_dict` copies
        write through).  Re-pointing a parameter elsewhere (`p.data = ...`) is detected by address and flattens again."""
        def aliased(big, parts):
            step = big[0:parts[0].shape[0]].numel() * big.element_size()
            return all(q.data_ptr() == big.data_ptr() + g * step and q.stride() == big[0:q.shape[0]].stride()
                       for g, q in enumerate(parts))
        if self._flat is not None and all(aliased(wb, [self.styles[j].convs[2 * d].weight for j in idxs]) and
                                          aliased(bb, [self.styles[j].convs[2 * d].bias for j in idxs])
                                          for idxs, per_depth in self._flat for d, (wb, bb) in enumerate(per_depth)):
            return self._flat
        flat = []
        with torch.no_grad():
            for idxs in self._levels():
                if not idxs:
                    continue
                per_depth = []
                for d in range(len(self.styles[idxs[0]].convs) // 2):
                    convs = [self.styles[j].convs[2 * d] for j in idxs]
                    wb = torch.cat([c.weight.data for c in convs], 0)
                    if convs[0].weight.data.is_contiguous(memory_format=torch.channels_last) and wb.dim() == 4:
                        wb = wb.contiguous(memory_format=torch.channels_last)
                    bb = torch.cat([c.bias.data for c in convs], 0)
                    co = convs[0].weight.shape[0]
                    for g, c in enumerate(convs):
                        c.weight.data = wb[g * co:(g + 1) * co]
                        c.bias.data = bb[g * co:(g + 1) * co]
                    per_depth.append((wb, bb))
                flat.append((idxs, per_depth))
        self._flat = flat
        return flat

    def _grouped_heads(self, level, feat):
        """All heads of one pyramid level as one chain: wide first conv on the shared feature map, grouped convs after it,
        bias + LeakyReLU on the HIP fused_bias_act kernel, the EqualLinears as one batched matmul.  Returns the level's
        latents [B,512] in head order."""
        idxs, per_depth = level
        G = len(idxs)
        blocks = [self.styles[j] for j in idxs]
        x = feat
        for d, (wb, bb) in enumerate(per_depth):
            conv, act = blocks[0].convs[2 * d], blocks[0].convs[2 * d + 1]
            y = F.conv2d(x, wb, None, conv.stride, conv.padding, 1, 1 if d == 0 else G)
            n, c, h, w = y.shape
            if y.is_contiguous(memory_format=torch.channels_last) and not y.is_contiguous():
                y = fused_leaky_relu(y.permute(0, 2, 3, 1).reshape(-1, c), bb, act.negative_slope, 1.0)
                x = y.view(n, h, w, c).permute(0, 3, 1, 2)
            else:
                x = fused_leaky_relu(y, bb, act.negative_slope, 1.0)
        out_c = blocks[0].out_c
        xs = x.reshape(-1, G, out_c).transpose(0, 1)                          # [G, B, 512]
        ws, bs = zip(*[b.linear._scaled_params() for b in blocks])
        out = torch.baddbmm(torch.stack(bs, 0).unsqueeze(1), xs, torch.stack(ws, 0).transpose(1, 2))
        return tuple(out.unbind(0))

    def _upsample_add(self, x, y):
        """Bilinear (align_corners) resize of x to y's size, plus y (psp_encoders.py:82-98)."""
        return F.interpolate(x, size=y.shape[2:], mode='bilinear', align_corners=True) + y

    def forward(self, x):
        heads = self.forward_deferred(x)
        for wait, _ in heads:
            wait()
        return torch.stack([t for _, t in heads], dim=1)

    def forward_deferred(self, x):
        if (not torch.is_grad_enabled()) and x.is_cuda:
            # the heads' EqualLinear weight*scale / bias*lr_mul: one refresh launch from the live parameters
            if self._live_weights is None:
                self._live_weights = LiveWeights(self)
            with self._live_weights.fresh():
                return self._forward_deferred(x)
        return self._forward_deferred(x)

    def _forward_deferred(self, x):
        """The n_styles latents [B,512] as a list of (wait, tensor): call wait() before the current stream reads the
        tensor.  In inference on the GPU the heads run on side streams and wait() is a per-head event, so a consumer
        that needs the latents one layer at a time (the synthesis network) starts while later heads are still running;
        otherwise everything runs in order and wait() does nothing."""
        if self.channels_last and x.is_cuda:
            self._to_channels_last()
            x = x.contiguous(memory_format=torch.channels_last)
        x = self.input_layer(x)
        t1, t2, t3 = _TAPS[self.num_layers]
        feats = {}
        for i, unit in enumerate(self.body):
            x = unit(x)
            if i in (t1, t2, t3):
                feats[i] = x
        c1, c2, c3 = feats[t1], feats[t2], feats[t3]
        if GROUP_HEADS and x.is_cuda and x.dtype == torch.float32 and not torch.is_grad_enabled():
            flat = self._flatten_heads()
            if HEAD_STREAMS > 1 and overlap_ok(x):
                streams = side_streams(x.device, HEAD_STREAMS, 'psp-heads')

                def level(k, feat):
                    wait, outs = run_deferred(streams[k % HEAD_STREAMS], self._grouped_heads, flat[k], feat)
                    return [(wait, o) for o in outs]
            else:
                def level(k, feat):
                    return [((lambda: None), o) for o in self._grouped_heads(flat[k], feat)]
            latents = level(0, c3)
            if len(flat) > 1:
                p2 = self._upsample_add(c3, self.latlayer1(c2))
                latents += level(1, p2)
            if len(flat) > 2:
                p1 = self._upsample_add(p2, self.latlayer2(c1))
                latents += level(2, p1)
            return latents
        if HEAD_STREAMS > 1 and overlap_ok(x):
            streams = side_streams(x.device, HEAD_STREAMS, 'psp-heads')

            def head(j, feat):
                return run_deferred(streams[j % HEAD_STREAMS], self.styles[j], feat)
        else:
            def head(j, feat):
                return (lambda: None), self.styles[j](feat)
        latents = [head(j, c3) for j in range(min(self.coarse_ind, self.style_count))]
        p2 = self._upsample_add(c3, self.latlayer1(c2))
        latents += [head(j, p2) for j in range(self.coarse_ind, min(self.middle_ind, self.style_count))]
        p1 = self._upsample_add(p2, self.latlayer2(c1))
        latents += [head(j, p1) for j in range(self.middle_ind, self.style_count)]
        return latents
