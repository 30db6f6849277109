"""StyleGAN2 Generator / Discriminator of 3D-FM GAN on MI355X kernels.

Same public API, constructor arguments, attribute names and state_dict keys as the reference's stylegan2.py
(so train_3_encoder.py / Evaluation/visual_eval.py / reference checkpoints work unchanged — pinned by
tests/golden/*_manifest.json), but the hot ops are hand-written gfx950 kernels reached through `op`:
  upfirdn2d, fused_leaky_relu           -> op.upfirdn2d / op.fused_act          (reference: op/*.cu)
  ModulatedConv2d 3x3 / transposed 3x3  -> op.modconv (MFMA implicit GEMM)      (reference: stylegan2.py:250-298)
  StyledConv epilogue (noise+bias+act)  -> fused into the conv epilogue when no graph is needed
  ToRGB (1x1 + bias + skip add)         -> one HBM pass                         (reference: stylegan2.py:389-404)
Host code stays PyTorch-ROCm.  There is no CPU path: CPU tensors raise RuntimeError in `op`.
"""
import math
import os
import random

import torch
from torch import nn, autograd
from torch.nn import functional as F

from op import FusedLeakyReLU, fused_leaky_relu, upfirdn2d
from op import _native, conv_grad, modconv, placement
from op._native import amp_fwd as _amp_fwd, amp_bwd as _amp_bwd
from op.live_weights import LiveWeights, live
from Util.streams import side_streams, run_on, overlap_ok

_SQRT2 = math.sqrt(2.0)


def _channel_table(channel_multiplier):
    # stylegan2.py:441-451 / 779-789
    table = {4: 512, 8: 512, 16: 512, 32: 512}
    for res, base in ((64, 256), (128, 128), (256, 64), (512, 32), (1024, 16)):
        table[res] = base * channel_multiplier
    return table


def make_kernel(k):
    """Normalised 2-D FIR from 1-D (outer product) or 2-D taps (stylegan2.py:36-44)."""
    k = torch.as_tensor(k, dtype=torch.float32)
    if k.ndim == 1:
        k = torch.outer(k, k)
    return k / k.sum()


class PixelNorm(nn.Module):
    """x / sqrt(mean_c(x^2) + 1e-8) (stylegan2.py:23-33)."""

    def forward(self, input):
        return input * torch.rsqrt(input.square().mean(dim=1, keepdim=True) + 1e-8)


class Upsample(nn.Module):
    """x`factor` FIR upsampling, taps * factor^2 (stylegan2.py:47-65)."""

    def __init__(self, kernel, factor=2):
        super().__init__()
        self.factor = factor
        self.register_buffer('kernel', make_kernel(kernel) * (factor ** 2))
        p = self.kernel.shape[0] - factor
        self.pad = ((p + 1) // 2 + factor - 1, p // 2)

    def forward(self, input):
        return upfirdn2d(input, self.kernel, up=self.factor, down=1, pad=self.pad)


class Downsample(nn.Module):
    """FIR decimation (stylegan2.py:68-86); defined for API parity, unused on the generator path."""

    def __init__(self, kernel, factor=2):
        super().__init__()
        self.factor = factor
        self.register_buffer('kernel', make_kernel(kernel))
        p = self.kernel.shape[0] - factor
        self.pad = ((p + 1) // 2, p // 2)

    def forward(self, input):
        return upfirdn2d(input, self.kernel, up=1, down=self.factor, pad=self.pad)


class Blur(nn.Module):
    """FIR low-pass with explicit padding (stylegan2.py:89-105)."""

    def __init__(self, kernel, pad, upsample_factor=1):
        super().__init__()
        taps = make_kernel(kernel)
        if upsample_factor > 1:
            taps = taps * (upsample_factor ** 2)
        self.register_buffer('kernel', taps)
        self.pad = pad

    def forward(self, input):
        return upfirdn2d(input, self.kernel, pad=self.pad)


class EqualConv2d(nn.Module):
    """Conv2d with equalised learning rate: N(0,1) weights scaled by 1/sqrt(fan_in) at run time (stylegan2.py:108-143)."""

    def __init__(self, in_channel, out_channel, kernel_size, stride=1, padding=0, bias=True):
        super().__init__()
        self.weight = nn.Parameter(torch.randn(out_channel, in_channel, kernel_size, kernel_size))
        self.scale = 1 / math.sqrt(in_channel * kernel_size ** 2)
        self.stride = stride
        self.padding = padding
        self.bias = nn.Parameter(torch.zeros(out_channel)) if bias else None

    def forward(self, input):
        if input.is_cuda and torch.is_grad_enabled():
            # same convolution; its derivatives of every order are the library's backward-data / backward-weight
            # primitives (op/conv_grad.py: PyTorch's generic conv double-backward costs R1 1.6 s per step at 1024^2)
            return conv_grad.conv2d(input, self.weight * self.scale, self.bias, self.stride, self.padding)
        return F.conv2d(input, self.weight * self.scale, bias=self.bias, stride=self.stride, padding=self.padding)

    def __repr__(self):
        o, i, k, _ = self.weight.shape
        return f'{self.__class__.__name__}({i}, {o}, {k}, stride={self.stride}, padding={self.padding})'


class EqualLinear(nn.Module):
    """Linear with equalised learning rate and optional fused leaky ReLU (stylegan2.py:146-180)."""

    def __init__(self, in_dim, out_dim, bias=True, bias_init=0, lr_mul=1, activation=None):
        super().__init__()
        self.weight = nn.Parameter(torch.randn(out_dim, in_dim).div_(lr_mul))
        self.bias = nn.Parameter(torch.full((out_dim,), float(bias_init))) if bias else None
        self.activation = activation
        self.scale = (1 / math.sqrt(in_dim)) * lr_mul
        self.lr_mul = lr_mul
        self._live = None     # set by op.live_weights.LiveWeights of the enclosing network

    def _scaled_params(self):
        """weight*scale and bias*lr_mul (stylegan2.py:165-175).  Inside the inference forward of a network that owns a
        LiveWeights table they were re-derived from the live parameters by that forward's single refresh launch (the
        reference re-multiplies every 512x512 modulation matrix on every call: 88 elementwise launches per 1024^2
        forward); anywhere else they are computed here.  Nothing survives a forward, so in-place `.data` updates
        (EMA accumulate, train_3_encoder.py:195-200) can never be missed."""
        lv = live(self)
        if lv is not None:
            return lv
        w, b = self.weight, self.bias
        return w * self.scale, (None if b is None else b * self.lr_mul)

    def forward(self, input):
        weight, bias = self._scaled_params()
        if self.activation:
            return fused_leaky_relu(F.linear(input, weight), bias)
        # Inference at per-rank batch sizes (the modulation of every StyledConv / ToRGB: 26 launches per 1024^2 forward, on
        # the critical path between the contractions): a matrix-vector product per sample on this repo's kernel (the BLAS
        # library serves 8 x 512 x 512 with a 16 x 16 macro-tile GEMM in 19 us).  Same sums up to the order of addition;
        # measured 376.6 / 378.3 vs 375.6 / 377.2 pairs/s in alternating launches on one box.
        if (input.is_cuda and not torch.is_grad_enabled() and input.dim() == 2 and input.shape[0] <= 64
                and input.dtype == torch.float32 and weight.dtype == torch.float32 and not torch.is_autocast_enabled()):
            return _native.equal_linear(input, weight, bias)
        return F.linear(input, weight, bias=bias)

    def __repr__(self):
        return f'{self.__class__.__name__}({self.weight.shape[1]}, {self.weight.shape[0]})'


class ScaledLeakyReLU(nn.Module):
    def __init__(self, negative_slope=0.2):
        super().__init__()
        self.negative_slope = negative_slope

    def forward(self, input):
        return F.leaky_relu(input, negative_slope=self.negative_slope) * _SQRT2


def _fir_pads(n_taps, kernel_size, upsample):
    """Blur padding around the strided conv (stylegan2.py:216-230, 705-709)."""
    if upsample:
        p = (n_taps - 2) - (kernel_size - 1)
        return (p + 1) // 2 + 1, p // 2 + 1
    p = (n_taps - 2) + (kernel_size - 1)
    return (p + 1) // 2, p // 2


class ModulatedConv2d(nn.Module):
    """Per-sample modulated (and demodulated) convolution (stylegan2.py:195-298).

    The reference multiplies the weight by the style per sample and runs a grouped conv.  Here the style
    multiplies the INPUT while it is staged into LDS and the demodulation coefficient scales the OUTPUT in
    the kernel epilogue, so one weight matrix serves the whole batch on the MFMA units (op/modconv.py).
    """

    def __init__(self, in_channel, out_channel, kernel_size, style_dim, demodulate=True, upsample=False,
                 downsample=False, blur_kernel=[1, 3, 3, 1]):
        super().__init__()
        self.eps = 1e-8
        self.kernel_size = kernel_size
        self.in_channel = in_channel
        self.out_channel = out_channel
        self.upsample = upsample
        self.downsample = downsample
        if upsample:
            self.blur = Blur(blur_kernel, pad=_fir_pads(len(blur_kernel), kernel_size, True), upsample_factor=2)
        if downsample:
            self.blur = Blur(blur_kernel, pad=_fir_pads(len(blur_kernel), kernel_size, False))
        self.scale = 1 / math.sqrt(in_channel * kernel_size ** 2)
        self.padding = kernel_size // 2
        self.weight = nn.Parameter(torch.randn(1, out_channel, in_channel, kernel_size, kernel_size))
        self.modulation = EqualLinear(style_dim, in_channel, bias_init=1)
        self.demodulate = demodulate
        self._live = None     # set by op.live_weights.LiveWeights of the enclosing network

    def __repr__(self):
        return (f'{self.__class__.__name__}({self.in_channel}, {self.out_channel}, {self.kernel_size}, '
                f'upsample={self.upsample}, downsample={self.downsample})')

    def mfma_weight(self):
        """wt[i][tap][o] = scale * weight[o][i][tap]: this forward's refreshed table entry, else derived now."""
        lv = live(self)
        if lv is not None:
            return lv[0]
        with torch.no_grad():
            return _native.modconv_weight_prep(self.weight.detach(), self.scale)

    def mfma_wsq(self):
        """Per-(o,i) sum of squared taps for the demodulation kernel (same rule as mfma_weight)."""
        lv = live(self)
        if lv is not None:
            return lv[1]
        with torch.no_grad():
            return _native.modconv_wsq(self.weight.detach())

    def styles(self, style):
        return self.modulation(style)

    def forward(self, input, style, return_style_scalars=False):
        s = self.styles(style)
        hip = modconv.hip_conv_ok(input, self.weight)
        wt = self.mfma_weight() if hip else None
        if self.upsample:
            out = self.blur(modconv.modulated_conv2d(input, self.weight, s, wt, self.demodulate, 1, self.scale))
        elif self.downsample:
            out = modconv.modulated_conv2d(self.blur(input), self.weight, s, wt, self.demodulate, 2, self.scale)
        else:
            out = modconv.modulated_conv2d(input, self.weight, s, wt, self.demodulate, 0, self.scale)
        if return_style_scalars:
            return out, s.view(s.shape[0], 1, self.in_channel, 1, 1)
        return out


class _NoiseInjectionFunction(autograd.Function):
    """image + weight * noise with a cheaper and more accurate weight gradient.  Autograd's own rule materialises
    grad_out * noise at full [B,C,H,W] size (1 GB per layer at 1024^2, B=8) and sums ~10^8 signed fp32 terms into one
    scalar — heavy cancellation: measured 2e-3 relative error against float64, 10-20x the CPU reference's.  Here the
    channels are summed first (one read of grad_out, float64 accumulation) and the [B,1,H,W] result is contracted
    with the noise in float64.  The backward is made of differentiable ops, so it also serves create_graph=True."""

    @staticmethod
    @_amp_fwd
    def forward(ctx, image, weight, noise):
        ctx.save_for_backward(weight, noise)
        return image + weight * noise

    @staticmethod
    @_amp_bwd
    def backward(ctx, go):
        weight, noise = ctx.saved_tensors
        g_weight = g_noise = None
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            csum = go.sum(1, keepdim=True, dtype=torch.float64)                   # [B,1,H,W]
            if ctx.needs_input_grad[1]:
                g_weight = (csum * noise).sum().to(go.dtype).reshape(weight.shape)
            if ctx.needs_input_grad[2]:
                g_noise = (csum * weight).to(go.dtype)
                if noise.shape[0] != go.shape[0]:
                    g_noise = g_noise.sum(0, keepdim=True)
        return go, g_weight, g_noise


class NoiseInjection(nn.Module):
    """image + weight * noise, fresh N(0,1) noise when none is given (stylegan2.py:301-312)."""

    def __init__(self):
        super().__init__()
        self.weight = nn.Parameter(torch.zeros(1))

    def forward(self, image, noise=None):
        if noise is None:
            b, _, h, w = image.shape
            noise = image.new_empty(b, 1, h, w).normal_()
        if image.is_cuda and image.dtype == torch.float32 and image.ndim == 4 and noise.shape[1] == 1:
            return _NoiseInjectionFunction.apply(image, self.weight, noise)
        return image + self.weight * noise


class ConstantInput(nn.Module):
    """Learned 4x4 constant repeated over the batch (stylegan2.py:315-329)."""

    def __init__(self, channel, size=4):
        super().__init__()
        self.input = nn.Parameter(torch.randn(1, channel, size, size))

    def forward(self, input):
        return self.input.repeat(input.shape[0], 1, 1, 1)


WINOGRAD = os.environ.get('FMGAN_NO_WINOGRAD', '0') != '1'   # Winograd F(2x2,3x3) form of the wide plain convs (inference)


def winograd_pays(batch, cin, cout, h, w):
    """Plain StyledConv layers served by the Winograd form (op/_native.py: modconv2d_winograd) instead of the direct MFMA kernel.
    Measured at B = 8 (profiles/r03_winograd.md): 16^2..64^2 x 512 channels 1.6-1.9x, 128^2 x 256 channels 1.23x, 256^2 x 128
    channels 0.83x (its transforms move 4x the activation through HBM): wide layers of at most 128^2 only, with enough tiles
    for the 16 GEMMs to fill the chip."""
    if not WINOGRAD or _native.current_modconv_precision() != 'f32' or (h | w) & 1:
        return False
    return cin >= 256 and cout >= 256 and 16 <= h <= 128 and 16 <= w <= 128 and batch * (h // 2) * (w // 2) >= 512
FUSE_RGB = os.environ.get('FMGAN_NO_RGB_FUSE', '0') != '1'   # ToRGB in the preceding conv's epilogue (inference)


class StyledConv(nn.Module):
    """ModulatedConv2d -> NoiseInjection -> FusedLeakyReLU (stylegan2.py:332-376).

    When no autograd graph is being built the three steps after the contraction run inside the conv kernel's
    epilogue (plain conv) or in one elementwise pass (after the upsampling blur): the activation is written once
    instead of three times.
    """

    def __init__(self, in_channel, out_channel, kernel_size, style_dim, upsample=False, blur_kernel=[1, 3, 3, 1],
                 demodulate=True):
        super().__init__()
        self.conv = ModulatedConv2d(in_channel, out_channel, kernel_size, style_dim, upsample=upsample,
                                    blur_kernel=blur_kernel, demodulate=demodulate)
        self.noise = NoiseInjection()
        self.activate = FusedLeakyReLU(out_channel)

    def _fused(self, input, style, noise):
        conv, act = self.conv, self.activate
        s = conv.styles(style)
        lv = live(conv)     # (wt, wsq) refreshed by the enclosing network's forward, if any
        demod = (_native.modconv_demod(conv.weight, s, conv.scale, conv.eps, lv[1] if lv else None)
                 if conv.demodulate else None)
        b, _, h, w = input.shape
        oh, ow = (2 * h, 2 * w) if conv.upsample else (h, w)
        if noise is None:
            noise = input.new_empty(b, 1, oh, ow).normal_()
        if conv.upsample:
            # private intermediate [B,C,2H+1,2W+1] in the aligned-row layout: every dwordx4 load of the blur is
            # 16-byte aligned (the contiguous 2W+1-float rows never are)
            c = conv.out_channel
            pad0, pad1 = conv.blur.pad
            shape, off, ps, rs = _native.aligned_rows_shape(b, c, 2 * h + 1, 2 * w + 1, pad0)
            wt = conv.mfma_weight()

            def produce(buf_):
                _native.modconv2d(input, wt, s, demod, 1, strided_out=(buf_.data_ptr() + 4 * off, ps, rs))

            def consume(buf_, out_):
                # blur + noise + bias + act in the blur's store: row-march / LDS-DMA ring for planes >= 64 wide,
                # plane-tile (whole planes in LDS) for the 8^2..32^2 layers
                return _native.blur_noise_bias_act(buf_.data_ptr() + 4 * off, input.device, b, c, 2 * h + 1, 2 * w + 1, ps, rs,
                                                   conv.blur.kernel, (pad0, pad1), noise, self.noise.weight, act.bias,
                                                   act.negative_slope, act.scale, out=out_)
            nbytes = 4 * shape[0] * shape[1] * shape[2]
            ws = None
            if placement.active() and nbytes >= placement.MIN_BYTES:
                # the largest buffers of the forward: a persistent pair whose placement was selected by measurement
                # (op/placement.py: the same kernel runs at 4.85 or 5.17 TB/s depending on which two blocks it gets)
                ws = placement.workspace(self, (b, c, h, w), shape, (b, c, oh, ow), input.device, produce, consume)
            if ws is not None:
                buf, out = ws.buf, ws.out
                produce(buf)
                consume(buf, out)
            else:
                buf = torch.empty(shape, dtype=torch.float32, device=input.device)
                produce(buf)
                out = consume(buf, None)
                if out is None:   # shapes neither kernel serves (planes > ~110^2 narrower than 64): two passes
                    out = _native.upfirdn2d_strided(buf.data_ptr() + 4 * off, input.device, b * c, 2 * h + 1, 2 * w + 1, ps,
                                                    rs, conv.blur.kernel, pad0, pad1, pad0, pad1).view(b, c, oh, ow)
                    out = _native.noise_bias_act(out, noise, self.noise.weight, act.bias, act.negative_slope, act.scale)
            del buf
        elif winograd_pays(b, conv.in_channel, conv.out_channel, h, w):
            # 16 products per 2x2 output tile instead of 36 (own transform kernels around 16 batched library GEMMs): the
            # direct kernel is already at 0.84-0.88 of the fp32 matrix peak on these layers
            out = _native.modconv2d_winograd(input, conv.mfma_weight(), s, demod, noise=noise,
                                             noise_weight=self.noise.weight, bias=act.bias, fuse_act=True,
                                             alpha=act.negative_slope, act_scale=act.scale)
        else:
            out = _native.modconv2d(input, conv.mfma_weight(), s, demod, 0, noise=noise,
                                    noise_weight=self.noise.weight, bias=act.bias, fuse_act=True,
                                    alpha=act.negative_slope, act_scale=act.scale)
        return out, s

    def rgb_fusable(self, x_shape, x):
        """Can this (plain) StyledConv take the following ToRGB into its epilogue for an input of this shape?"""
        conv = self.conv
        if conv.upsample or conv.downsample or torch.is_grad_enabled() or not modconv.hip_conv_ok(x, conv.weight):
            return False
        if _native.current_modconv_precision() != 'f32':     # the RGB epilogue exists on the fp32 MFMA kernel only
            return False
        b, _, h, w = x_shape
        return _native.modconv2d_rgb_fusable(b, conv.in_channel, conv.out_channel, h, w)

    def fused_with_rgb(self, input, style, noise, to_rgb, rgb_latent, skip_up, keep_out):
        """StyledConv + ToRGB in one kernel (inference): returns (activation or None, rgb)."""
        if torch.is_autocast_enabled():
            with torch.autocast('cuda', enabled=False):
                return self.fused_with_rgb(input.float(), style.float(), None if noise is None else noise.float(), to_rgb,
                                           rgb_latent.float(), None if skip_up is None else skip_up.float(), keep_out)
        conv, act = self.conv, self.activate
        s = conv.styles(style)
        lv = live(conv)     # (wt, wsq) refreshed by the enclosing network's forward, if any
        demod = (_native.modconv_demod(conv.weight, s, conv.scale, conv.eps, lv[1] if lv else None)
                 if conv.demodulate else None)
        if noise is None:
            noise = input.new_empty(input.shape[0], 1, input.shape[2], input.shape[3]).normal_()
        return _native.modconv2d_rgb(input, conv.mfma_weight(), s, demod, noise, self.noise.weight, act.bias,
                                     act.negative_slope, act.scale, to_rgb.conv.weight, to_rgb.conv.styles(rgb_latent),
                                     to_rgb.bias, skip_up, to_rgb.conv.scale, keep_out)

    def forward(self, input, style, return_style_scalars=False, noise=None):
        if (not torch.is_grad_enabled()) and modconv.hip_conv_ok(input, self.conv.weight) and not self.conv.downsample:
            if torch.is_autocast_enabled():
                # the fused path hands raw fp32 pointers to the library: run it with autocast off on fp32 inputs (the style
                # MLP would otherwise produce a bf16 vector; the training path gets this from its autograd Functions)
                with torch.autocast('cuda', enabled=False):
                    out, s = self._fused(input.float(), style.float(), None if noise is None else noise.float())
            else:
                out, s = self._fused(input, style, noise)
            if return_style_scalars:
                return out, s.view(s.shape[0], 1, self.conv.in_channel, 1, 1)
            return out
        if return_style_scalars:
            out, styles = self.conv(input, style, True)
        else:
            out = self.conv(input, style)
        out = self.activate(self.noise(out, noise=noise))
        return (out, styles) if return_style_scalars else out


class ToRGB(nn.Module):
    """1x1 modulated conv to RGB + bias + upsampled skip (stylegan2.py:379-404)."""

    def __init__(self, in_channel, style_dim, upsample=True, blur_kernel=[1, 3, 3, 1]):
        super().__init__()
        if upsample:
            self.upsample = Upsample(blur_kernel)
        self.conv = ModulatedConv2d(in_channel, 3, 1, style_dim, demodulate=False)
        self.bias = nn.Parameter(torch.zeros(1, 3, 1, 1))

    def forward(self, input, style, skip=None, return_style_scalars=False):
        s = self.conv.styles(style)
        if skip is not None:
            skip = self.upsample(skip)
        out = modconv.to_rgb(input, self.conv.weight, s, self.bias, skip, self.conv.scale)
        if return_style_scalars:
            return out, s.view(s.shape[0], 1, self.conv.in_channel, 1, 1)
        return out


class _LatentColumns:
    """W+ given one column at a time: latent[:, i] calls provider(i) (which may first wait for the stream that is
    still producing that column).  Used by Forward_Inference_3_Encoder to start the synthesis network before the last
    style heads of the encoder have finished."""

    def __init__(self, provider, n_latent):
        self.provider, self.n = provider, n_latent

    def __getitem__(self, idx):
        if not (isinstance(idx, tuple) and len(idx) == 2 and isinstance(idx[1], int) and idx[0] == slice(None)):
            raise IndexError('only latent[:, i] is served column-wise')
        return self.provider(idx[1])


class Generator(nn.Module):
    """StyleGAN2 synthesis (+ mapping) network with the 3D-FM GAN extensions: external 4x4 input tensor, W+ input,
    in-forward path-length regulariser, RGB pyramid and style-scalar outputs (stylegan2.py:407-688)."""

    def __init__(self, size, style_dim, n_mlp, channel_multiplier=2, blur_kernel=[1, 3, 3, 1], lr_mlp=0.01,
                 generator_net_shape=None):
        super().__init__()
        self.size = size
        self.style_dim = style_dim
        mapping = [PixelNorm()]
        mapping += [EqualLinear(style_dim, style_dim, lr_mul=lr_mlp, activation='fused_lrelu') for _ in range(n_mlp)]
        self.style = nn.Sequential(*mapping)
        self.channels = _channel_table(channel_multiplier)
        self.log_size = int(math.log(size, 2))
        self.num_layers = (self.log_size - 2) * 2 + 1
        self.n_latent = self.log_size * 2 - 2

        # widths[j] = channels entering synthesis layer j; widths[-1] = channels leaving the last one.
        # generator_net_shape (pruned nets, Util/network_util.py:39-50) lists exactly that.
        if generator_net_shape is None:
            widths = [self.channels[4], self.channels[4]]
            for i in range(3, self.log_size + 1):
                widths += [self.channels[2 ** i]] * 2
        else:
            widths = list(generator_net_shape)

        self.input = ConstantInput(widths[0])
        self.conv1 = StyledConv(widths[0], widths[1], 3, style_dim, blur_kernel=blur_kernel)
        self.to_rgb1 = ToRGB(widths[1], style_dim, upsample=False)

        self.convs = nn.ModuleList()
        self.upsamples = nn.ModuleList()
        self.to_rgbs = nn.ModuleList()
        self.noises = nn.Module()
        for layer_idx in range(self.num_layers):
            res = 2 ** ((layer_idx + 5) // 2)
            self.noises.register_buffer(f'noise_{layer_idx}', torch.randn(1, 1, res, res))

        n_blocks = (self.log_size - 2) if generator_net_shape is None else (len(widths) // 2 - 1)
        for blk in range(1, n_blocks + 1):
            c_in, c_mid, c_out = widths[2 * blk - 1], widths[2 * blk], widths[2 * blk + 1]
            self.convs.append(StyledConv(c_in, c_mid, 3, style_dim, upsample=True, blur_kernel=blur_kernel))
            self.convs.append(StyledConv(c_mid, c_out, 3, style_dim, blur_kernel=blur_kernel))
            self.to_rgbs.append(ToRGB(c_out, style_dim))
        self._live_weights = None

    def make_noise(self):
        device = self.input.input.device
        noises = [torch.randn(1, 1, 4, 4, device=device)]
        for i in range(3, self.log_size + 1):
            noises += [torch.randn(1, 1, 2 ** i, 2 ** i, device=device) for _ in range(2)]
        return noises

    def mean_latent(self, n_latent):
        z = torch.randn(n_latent, self.style_dim, device=self.input.input.device)
        return self.style(z).mean(0, keepdim=True)

    def get_latent(self, input):
        return self.style(input)

    def _latents(self, styles, inject_index):
        """W+ tensor [B, n_latent, D] from one W/W+ batch or two W batches (style mixing), stylegan2.py:606-625."""
        if len(styles) < 2:
            w = styles[0]
            return w.unsqueeze(1).repeat(1, self.n_latent, 1) if w.ndim < 3 else w
        if inject_index is None:
            inject_index = random.randint(1, self.n_latent - 1)
        first = styles[0].unsqueeze(1).repeat(1, inject_index, 1)
        second = styles[1].unsqueeze(1).repeat(1, self.n_latent - inject_index, 1)
        return torch.cat([first, second], 1)

    def forward(self, *args, **kwargs):
        """Generator.forward of the reference (stylegan2.py:554-688; signature: _forward).  An inference call on the GPU
        first re-derives every scaled / MFMA-layout weight of the network from the live parameters in one launch
        (op/live_weights.py) and then runs the layers on those buffers."""
        if (not torch.is_grad_enabled()) and self.input.input.is_cuda:
            if self._live_weights is None:
                self._live_weights = LiveWeights(self)
            with self._live_weights.fresh(), placement.scope():
                return self._forward(*args, **kwargs)
        return self._forward(*args, **kwargs)

    def _forward(self, noise_z, return_latents=False, inject_index=None, truncation=1, truncation_latent=None,
                 latent_styles=None, input_is_latent=False, noise=None, randomize_noise=True,
                 use_external_input_tensor=False, external_input_tensor=None, PPL_regularize=False,
                 return_rgb_list=False, return_style_scalars=False, latent_columns=None):
        """latent_columns (not in the reference): callable i -> W+[:, i] replacing latent_styles; inference only, with
        an external input tensor (see _LatentColumns)."""
        if latent_columns is not None:
            if PPL_regularize or return_latents or return_style_scalars or not use_external_input_tensor:
                raise ValueError('latent_columns serves the plain inference forward only')
            styles = None
        else:
            styles = latent_styles if input_is_latent else [self.style(z) for z in noise_z]
        if noise is None:
            if randomize_noise:
                noise = [None] * self.num_layers
            else:
                noise = [getattr(self.noises, f'noise_{i}') for i in range(self.num_layers)]
        if latent_columns is not None:
            latent = _LatentColumns(latent_columns, self.n_latent)
        else:
            if truncation < 1:
                styles = [truncation_latent + truncation * (w - truncation_latent) for w in styles]
            latent = self._latents(styles, inject_index)

        if use_external_input_tensor:
            assert external_input_tensor is not None
            out = external_input_tensor
        else:
            out = self.input(latent)

        scalars = []

        def run(layer, x, w, **kw):
            if return_style_scalars:
                y, s = layer(x, w, return_style_scalars=True, **kw)
                scalars.append(s)
                return y
            return layer(x, w, **kw)

        # Inference: the RGB branch (ToRGB + skip upsample, HBM-bound) of resolution r has no consumer until the
        # image is returned, so it runs on a side stream beside the MFMA-bound convs of resolution 2r.
        overlap = overlap_ok(out) and not return_style_scalars
        joins = []

        def rgb(layer, x, w, skip):
            if not overlap:
                return layer(x, w, skip)
            join, y = run_on(side, layer, x, w, skip)
            joins.append(join)
            return y

        if overlap:
            side, = side_streams(out.device, 1)
        out = run(self.conv1, out, latent[:, 0], noise=noise[0])
        skip = rgb(self.to_rgb1, out, latent[:, 1], None)
        rgbs = [skip]
        for blk, to_rgb in enumerate(self.to_rgbs):
            i = 1 + 2 * blk
            conv_b = self.convs[2 * blk + 1]
            b_, _, h_, w_ = out.shape
            # Inference, last resolution: its ToRGB cannot hide beside later convs (there are none) and is the only
            # consumer of conv_b's activation, so it rides in conv_b's epilogue and the [B,C,size,size] activation is
            # never written or re-read (2.1 GB of HBM traffic at 1024^2, B=8).  Lower resolutions keep the separate
            # ToRGB kernel: it overlaps the next resolution's MFMA-bound convs on the side stream (measured: fusing
            # those too is time-neutral on the conv and puts the RGB reduction on the critical path).
            last = blk + 1 == len(self.to_rgbs)
            if (FUSE_RGB and last and not return_style_scalars and not return_rgb_list
                    and conv_b.rgb_fusable((b_, 0, 2 * h_, 2 * w_), out)):
                if overlap:
                    join_up, skip_up = run_on(side, to_rgb.upsample, skip)
                    joins.append(join_up)
                else:
                    skip_up = to_rgb.upsample(skip)
                out = self.convs[2 * blk](out, latent[:, i], noise=noise[i])
                if overlap:
                    join_up()
                out, skip = conv_b.fused_with_rgb(out, latent[:, i + 1], noise[i + 1], to_rgb, latent[:, i + 2], skip_up,
                                                  keep_out=False)
                rgbs.append(skip)
                continue
            out = run(self.convs[2 * blk], out, latent[:, i], noise=noise[i])
            out = run(self.convs[2 * blk + 1], out, latent[:, i + 1], noise=noise[i + 1])
            if return_style_scalars and i + 3 == latent.shape[1]:   # style scalars of the last ToRGB only (:660-662)
                skip = run(to_rgb, out, latent[:, i + 2], skip=skip)
            else:
                skip = rgb(to_rgb, out, latent[:, i + 2], skip)
            rgbs.append(skip)
        for join in joins[-1:]:
            join()                      # the side stream is in order: joining its last launch joins all of them
        image = skip
        for r in rgbs[:-1]:
            if overlap:
                r.record_stream(torch.cuda.current_stream(image.device))

        if PPL_regularize:
            # path-length regulariser evaluated inside forward so it shards with the batch (stylegan2.py:683-688)
            probe = torch.randn_like(image) / math.sqrt(image.shape[2] * image.shape[3])
            grad, = autograd.grad(outputs=(image * probe).sum(), inputs=latent, create_graph=True)
            return image, torch.sqrt(grad.pow(2).sum(2).mean(1))

        returns = rgbs if return_rgb_list else image
        return (returns, scalars) if return_style_scalars else returns


class ConvLayer(nn.Sequential):
    """[Blur] -> EqualConv2d -> [FusedLeakyReLU | ScaledLeakyReLU] (stylegan2.py:692-737)."""

    def __init__(self, in_channel, out_channel, kernel_size, downsample=False, blur_kernel=[1, 3, 3, 1], bias=True,
                 activate=True):
        layers = []
        if downsample:
            layers.append(Blur(blur_kernel, pad=_fir_pads(len(blur_kernel), kernel_size, False)))
            stride, self.padding = 2, 0
        else:
            stride, self.padding = 1, kernel_size // 2
        layers.append(EqualConv2d(in_channel, out_channel, kernel_size, padding=self.padding, stride=stride,
                                  bias=bias and not activate))
        if activate:
            layers.append(FusedLeakyReLU(out_channel) if bias else ScaledLeakyReLU(0.2))
        super().__init__(*layers)


class ResBlock(nn.Module):
    """(conv3x3 -> blur+conv3x3/2 + blur+conv1x1/2 skip) / sqrt(2) (stylegan2.py:740-759)."""

    def __init__(self, in_channel, out_channel, blur_kernel=[1, 3, 3, 1]):
        super().__init__()
        self.conv1 = ConvLayer(in_channel, in_channel, 3)
        self.conv2 = ConvLayer(in_channel, out_channel, 3, downsample=True)
        self.skip = ConvLayer(in_channel, out_channel, 1, downsample=True, activate=False, bias=False)

    def forward(self, input):
        return (self.conv2(self.conv1(input)) + self.skip(input)) / _SQRT2


class Discriminator(nn.Module):
    """StyleGAN2 residual discriminator with minibatch-stddev (stylegan2.py:762-820)."""

    def __init__(self, size, channel_multiplier=2, blur_kernel=[1, 3, 3, 1]):
        super().__init__()
        channels = _channel_table(channel_multiplier)
        log_size = int(math.log(size, 2))
        convs = [ConvLayer(3, channels[size], 1)]
        in_channel = channels[size]
        for i in range(log_size, 2, -1):
            out_channel = channels[2 ** (i - 1)]
            convs.append(ResBlock(in_channel, out_channel, blur_kernel))
            in_channel = out_channel
        self.convs = nn.Sequential(*convs)
        self.stddev_group = 4
        self.stddev_feat = 1
        self.final_conv = ConvLayer(in_channel + 1, channels[4], 3)
        self.final_linear = nn.Sequential(
            EqualLinear(channels[4] * 4 * 4, channels[4], activation='fused_lrelu'),
            EqualLinear(channels[4], 1),
        )

    def forward(self, input):
        out = self.convs(input)
        batch, channel, height, width = out.shape
        group = min(batch, self.stddev_group)
        sd = out.view(group, -1, self.stddev_feat, channel // self.stddev_feat, height, width)
        sd = torch.sqrt(sd.var(0, unbiased=False) + 1e-8)
        sd = sd.mean([2, 3, 4], keepdims=True).squeeze(2).repeat(group, 1, height, width)
        out = self.final_conv(torch.cat([out, sd], 1))
        return self.final_linear(out.view(batch, -1))
