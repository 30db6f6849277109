"""Host side of the modulated-convolution kernels (fmgan_modconv_* / fmgan_torgb_f32 in include/fmgan_hip.h).

Forward runs on the hand-written MFMA kernels.  The reference gets its gradients (to second order: path-length
regularisation differentiates through Generator.forward with create_graph=True, stylegan2.py:683-688) from
autograd over F.conv2d / F.conv_transpose2d.  Here:
  * plain loss.backward() (no graph requested): the data gradient runs on the MFMA kernel itself with swapped roles
    (ModulatedConv2dFunction.backward): 3 convolutions per layer per training step;
  * create_graph=True: the backward re-states the op in the input-modulated form  d (.) C(x (.) s, scale W)  whose
    dense contraction C is the DenseConv / DenseConvDgrad / DenseConvWgrad family — three autograd Functions on the
    MFMA kernel that differentiate into each other (the way the reference nests UpFirDn2d / UpFirDn2dBackward,
    op/upfirdn2d.py:28-94), so every derivative order runs this repo's kernels (the stride-2 weight-gradient
    primitive is MIOpen's unless FMGAN_HIP_WGRAD=2).
"""
import os

import torch
from torch.autograd import Function
from torch.nn import functional as F

from . import _native
from ._native import amp_fwd as _amp_fwd, amp_bwd as _amp_bwd

# Weight gradient of the plain conv: this repo's fmgan_modconv_wgrad_f32 (64 x 64 tile, sliding 3 x 3 window,
# register-prefetch pipeline; bit-reproducible) measures 111-123 TFLOP/s on MI355X against 96-116 for MIOpen's fp32
# wgrad on the same layers (profiles/r02_backward_layers.txt) and is the default wherever it applies (>= 48 channels on
# both sides, >= 16 pixels per row).  Its stride-2 forms (transposed / downsampling conv: 16-pixel steps, 7 LDS reads per 9
# MFMAs) measure 60-63 TFLOP/s against MIOpen's 79-95 and are opt-in.  FMGAN_HIP_WGRAD: 0 = MIOpen everywhere,
# 1 (default) = own kernel for the plain conv, 2 = own kernel for every mode.
HIP_WGRAD = int(os.environ.get('FMGAN_HIP_WGRAD', '1'))
COMPOSITE_MIOPEN = os.environ.get('FMGAN_COMPOSITE_MIOPEN', '0') == '1'


def modconv_composite(x, weight, s, demodulate, mode, scale, eps=1e-8):
    """y[b,o] = d[b,o] * conv(x[b,i] * s[b,i], scale * W[o,i]);  d = rsqrt(sum_i s^2 * sum_k (scale W)^2 + eps).
    Algebraically the reference's ModulatedConv2d (stylegan2.py:257-293); differentiable torch ops only.
    mode 0: pad k//2; mode 1: transposed stride 2 (-> 2H+1); mode 2: stride-2 valid conv (input pre-blurred)."""
    cout, cin, k, _ = weight.shape[-4:]
    w = weight.reshape(cout, cin, k, k) * scale
    xs = x * s[:, :, None, None]
    if mode == 1:
        y = F.conv_transpose2d(xs, w.transpose(0, 1), stride=2)
    elif mode == 2:
        y = F.conv2d(xs, w, stride=2)
    else:
        y = F.conv2d(xs, w, padding=k // 2)
    if demodulate:
        d = torch.rsqrt(s.pow(2) @ w.pow(2).sum([2, 3]).t() + eps)
        y = y * d[:, :, None, None]
    return y


# ----------------------------------------------------------------------------- dense 3x3 convs, closed under d/d.
# The reference gets every derivative order by nesting autograd Functions whose backward is the same native op again
# (op/upfirdn2d.py:28-94).  The modulated conv is  y = d (.) C(x (.) s, scale W)  with C a dense, batch-shared 3x3
# convolution; C is bilinear in (input, weight), and its two adjoints are again such convolutions:
#     C_m(u, W)       forward             mode 0 pad-1 | mode 1 transposed stride 2 | mode 2 stride-2 valid
#     D_m(g, W)       adjoint w.r.t. u    mode 0: C_0 with W^T flipped | mode 1: C_2 with W^T | mode 2: C_1 with W^T
#     G_m(u, g)       adjoint w.r.t. W    (weight gradient)
# and   dD_m/dg = C_m(., W),  dD_m/dW = G_m(., g),  dG_m/du = D_m(g, .),  dG_m/dg = C_m(u, .)
# — three Functions that differentiate into each other, so R1 / path-length regularisation (create_graph=True) run
# on the MFMA kernel at every order.  Modulation and demodulation stay elementwise torch ops around C.
_ONES = {}


def _ones(batch, ch, device):
    key = (batch, ch, device)
    if key not in _ONES:
        _ONES[key] = torch.ones(batch, ch, dtype=torch.float32, device=device)
    return _ONES[key]


def _kernel_conv(u, w4, mode, kind):
    """The MFMA kernel as a dense conv: channels contracted = w4's dim 1 (kind 0) or dim 0 (kinds 1, 2)."""
    wt = _native.modconv_weight_prep(w4, 1.0, kind=kind)
    return _native.modconv2d(u, wt, _ones(u.shape[0], u.shape[1], u.device), None, mode)


def _wgrad(u, g, mode):
    """G_m(u, g) -> [cout, cin, 3, 3].  mode 0 on this repo's MFMA wgrad kernel when enabled, else MIOpen's fp32 wgrad."""
    cout, cin = g.shape[1], u.shape[1]
    if (HIP_WGRAD >= 2 or (HIP_WGRAD == 1 and mode == 0)) and u.dtype == torch.float32:
        gw = _native.modconv_wgrad(g, None, u, _ones(u.shape[0], cin, u.device), 1.0, fast_only=True, mode=mode)
        if gw is not None:
            return gw
    if mode == 0:
        return torch.nn.grad.conv2d_weight(u, (cout, cin, 3, 3), g, padding=1)
    if mode == 1:   # y = conv_transpose2d(u, W^T, stride 2): adjoint of conv2d(., W^T, stride 2)
        return torch.nn.grad.conv2d_weight(g, (cin, cout, 3, 3), u, stride=2).transpose(0, 1)
    return torch.nn.grad.conv2d_weight(u, (cout, cin, 3, 3), g, stride=2)


_ADJ = {0: (0, 1), 1: (2, 2), 2: (1, 2)}      # mode -> (kernel mode, weight layout kind) of D_m


class DenseConv(Function):
    """C_m(u, W): u [B,Cin,H,W] f32, W [Cout,Cin,3,3]."""

    @staticmethod
    @_amp_fwd
    def forward(ctx, u, w4, mode):
        ctx.save_for_backward(u, w4)
        ctx.mode = mode
        return _kernel_conv(u.contiguous(), w4, mode, 0)

    @staticmethod
    @_amp_bwd
    def backward(ctx, g):
        u, w4 = ctx.saved_tensors
        gu = DenseConvDgrad.apply(g, w4, ctx.mode, tuple(u.shape[2:])) if ctx.needs_input_grad[0] else None
        gw = DenseConvWgrad.apply(u, g, ctx.mode) if ctx.needs_input_grad[1] else None
        return gu, gw, None


class DenseConvDgrad(Function):
    """D_m(g, W): the adjoint of C_m(., W), again on the MFMA kernel (roles of Cin / Cout swapped)."""

    @staticmethod
    @_amp_fwd
    def forward(ctx, g, w4, mode, in_hw):
        ctx.save_for_backward(g, w4)
        ctx.mode = mode
        km, kind = _ADJ[mode]
        out = _kernel_conv(g.contiguous(), w4, km, kind)
        if mode == 2 and tuple(out.shape[2:]) != tuple(in_hw):
            # a stride-2 valid conv ignores the last row/column of an even-sized input: their gradient is zero
            out = F.pad(out, (0, in_hw[1] - out.shape[3], 0, in_hw[0] - out.shape[2]))
        return out

    @staticmethod
    @_amp_bwd
    def backward(ctx, h):
        g, w4 = ctx.saved_tensors
        gg = DenseConv.apply(h, w4, ctx.mode) if ctx.needs_input_grad[0] else None
        gw = DenseConvWgrad.apply(h, g, ctx.mode) if ctx.needs_input_grad[1] else None
        return gg, gw, None, None


class DenseConvWgrad(Function):
    """G_m(u, g): weight gradient of C_m."""

    @staticmethod
    @_amp_fwd
    def forward(ctx, u, g, mode):
        ctx.save_for_backward(u, g)
        ctx.mode = mode
        return _wgrad(u.contiguous(), g.contiguous(), mode).contiguous()

    @staticmethod
    @_amp_bwd
    def backward(ctx, hw):
        u, g = ctx.saved_tensors
        hw = hw.contiguous()
        gu = DenseConvDgrad.apply(g, hw, ctx.mode, tuple(u.shape[2:])) if ctx.needs_input_grad[0] else None
        gg = DenseConv.apply(u, hw, ctx.mode) if ctx.needs_input_grad[1] else None
        return gu, gg, None


def modconv_composite_hip(x, weight, s, demodulate, mode, scale, eps=1e-8):
    """modconv_composite with the contraction on the MFMA kernel at every derivative order (float32, 3x3)."""
    cout, cin, k, _ = weight.shape[-4:]
    w = weight.reshape(cout, cin, k, k) * scale
    y = DenseConv.apply(x * s[:, :, None, None], w, mode)
    if demodulate:
        d = torch.rsqrt(s.pow(2) @ w.pow(2).sum([2, 3]).t() + eps)
        y = y * d[:, :, None, None]
    return y


def _regrad(fn, saved, need, grad_out):
    """Gradients of fn(*saved) w.r.t. the needed inputs; keeps the graph when called under create_graph."""
    keep = torch.is_grad_enabled()
    with torch.enable_grad():
        ins = list(saved) if keep else [t.detach().requires_grad_(n) for t, n in zip(saved, need)]
        y = fn(*ins)
        sel = [t for t, n in zip(ins, need) if n and t.requires_grad]
        grads = iter(torch.autograd.grad(y, sel, grad_out, create_graph=keep, allow_unused=True)) if sel else iter(())
    return [next(grads) if (n and t.requires_grad) else None for t, n in zip(ins, need)]


class ModulatedConv2dFunction(Function):
    """3x3 modulated conv on the MFMA kernel: mode 0 plain, 1 transposed stride 2, 2 stride-2 valid.

    backward, first order (no graph requested): the data gradient runs on the SAME kernel with swapped roles —
        d/dx of mode 0:  g_u = conv(go * d, W^T flipped)            -> mode 0 with weight layout kind 1
        d/dx of mode 1:  g_u = stride-2 conv(go * d, W^T)           -> mode 2 with weight layout kind 2
    where u = x * s is the modulated input and the factor d (demodulation) rides as the kernel's input modulation;
    g_x = g_u * s.  The weight gradient of the plain conv is fmgan_modconv_wgrad_mode_f32 (see HIP_WGRAD), of the
    transposed conv MIOpen's wgrad on (u, go*d).  The demodulation chain rule is small
    [B,Cout]x[Cout,Cin] algebra.  When a graph is requested (R1 / path-length regularisers), the whole
    backward is the differentiable composite instead.
    """

    @staticmethod
    @_amp_fwd
    def forward(ctx, x, weight, s, wt, demodulate, mode, scale):
        demod = _native.modconv_demod(weight, s, scale) if demodulate else None
        out = _native.modconv2d(x, wt, s, demod, mode)
        saved = [x, weight, s] + ([demod, out] if demodulate else [])
        ctx.save_for_backward(*saved)
        ctx.cfg = (demodulate, mode, scale)
        return out

    @staticmethod
    @_amp_bwd
    def backward(ctx, grad_out):
        demodulate, mode, scale = ctx.cfg
        x, weight, s = ctx.saved_tensors[:3]
        need = ctx.needs_input_grad[:3]
        if torch.is_grad_enabled() or mode == 2 or grad_out.dtype != torch.float32:
            # a graph is wanted (R1 / path-length regularisation): differentiate the composite whose contraction is
            # the DenseConv family above — HIP kernels at every order (FMGAN_COMPOSITE_MIOPEN=1: the all-MIOpen form)
            comp = modconv_composite if (COMPOSITE_MIOPEN or grad_out.dtype != torch.float32) else modconv_composite_hip
            gx, gw, gs = _regrad(lambda a, b, c: comp(a, b, c, demodulate, mode, scale), (x, weight, s), need, grad_out)
            return gx, gw, gs, None, None, None, None
        go = grad_out.contiguous()
        cout, cin, k, _ = weight.shape[-4:]
        w4 = weight.reshape(cout, cin, k, k)
        batch = x.shape[0]
        d = ctx.saved_tensors[3] if demodulate else None
        dstyle = d if demodulate else torch.ones(batch, cout, dtype=torch.float32, device=x.device)
        gx = gw = gs = None
        gu = None
        if need[0] or need[2]:
            # data gradient on the MFMA kernel (roles of Cin / Cout swapped)
            wt_b = _native.modconv_weight_prep(w4, scale, kind=1 if mode == 0 else 2)
            gu = _native.modconv2d(go, wt_b, dstyle, None, 0 if mode == 0 else 2)
            if need[0]:
                gx = gu * s[:, :, None, None]
            if need[2]:
                gs = (gu * x).sum((2, 3))
        gq = None
        if demodulate and (need[1] or need[2]):
            out = ctx.saved_tensors[4]
            # d = q^-1/2 with q = scale^2 sum_i s^2 Wsq + eps;  dL/dq = dL/dd * (-1/2) d^3,  dL/dd = sum_p go*out / d
            gq = -0.5 * (go * out).sum((2, 3)) * d * d
            wsq = w4.square().sum((2, 3))                                  # [cout, cin]
            if need[2]:
                gs = gs + (2.0 * scale * scale) * s * (gq @ wsq)
        if need[1]:
            hip_w = HIP_WGRAD >= 2 or (HIP_WGRAD == 1 and mode == 0)
            gw = _native.modconv_wgrad(go, d, x, s, scale, fast_only=True, mode=mode) if hip_w else None
            if gw is None:
                gz = go * d[:, :, None, None] if demodulate else go
                u = x * s[:, :, None, None]
                if mode == 0:
                    gw = torch.nn.grad.conv2d_weight(u, (cout, cin, k, k), gz, padding=k // 2)
                else:   # y = conv_transpose2d(u, W^T, stride 2) is the adjoint of conv2d(., W^T, stride 2)
                    gw = torch.nn.grad.conv2d_weight(gz, (cin, cout, k, k), u, stride=2).transpose(0, 1)
                gw = gw * scale
            if gq is not None:
                gw = gw + (2.0 * scale * scale) * w4 * (gq.t() @ s.square())[:, :, None, None]
            gw = gw.reshape(weight.shape)
        return gx, gw, gs, None, None, None, None


def _torgb_conv(x, weight, s, scale):
    """ToRGB's 1x1 modulated conv without demodulation as a batched matmul: y[b] = (scale * W * s[b]) @ x[b].
    Differentiable torch ops whose derivatives of every order are matmuls again.  (As a grouped F.conv2d its
    double-backward — the path-length regulariser differentiates through ToRGB's backward — goes through PyTorch's generic
    conv formula with the whole 1024^2 feature map as the FILTER: one 292 ms fall-back kernel per step, 60 % of the
    path-length phase at 1024^2; profiles/r03_ppl_1024_kernels_before.md.)"""
    b, cin, h, w = x.shape
    wmod = (weight.reshape(1, -1, cin) * scale) * s[:, None, :]
    return torch.bmm(wmod, x.reshape(b, cin, h * w)).view(b, -1, h, w)


def _torgb_composite(x, weight, s, bias, skip, scale):
    y = _torgb_conv(x, weight, s, scale)
    if bias is not None:
        y = y + bias.view(1, -1, 1, 1)
    if skip is not None:
        y = y + skip
    return y


class ToRGBFunction(Function):
    """1x1 modulated conv without demodulation + bias + skip add in one HBM pass."""

    @staticmethod
    @_amp_fwd
    def forward(ctx, x, weight, s, bias, skip, scale):
        out = _native.torgb(x, weight, s, bias, skip, scale)
        ctx.has_skip = skip is not None
        ctx.save_for_backward(x, weight, s, bias, *([skip] if ctx.has_skip else []))
        ctx.scale = scale
        return out

    @staticmethod
    @_amp_bwd
    def backward(ctx, grad_out):
        x, weight, s, bias = ctx.saved_tensors[:4]
        scale = ctx.scale
        both = None
        if not torch.is_grad_enabled():      # plain backward: one pass over x on the HIP kernel
            both = _native.torgb_backward(x, grad_out, weight, s, scale)
        if both is not None:
            gx, m = both                      # m[b,c,i] = sum_p grad_out[b,c,p] * x[b,i,p]
            w2 = weight.reshape(-1, x.shape[1])
            gw = (scale * (m * s[:, None, :]).sum(0)).view_as(weight) if ctx.needs_input_grad[1] else None
            gs = scale * (m * w2[None]).sum(1) if ctx.needs_input_grad[2] else None
            if not ctx.needs_input_grad[0]:
                gx = None
        else:                                 # graph requested (path-length regulariser) / unserved shape: composite
            gx, gw, gs = _regrad(lambda a, b, c: _torgb_conv(a, b, c, scale), (x, weight, s),
                                 ctx.needs_input_grad[:3], grad_out)
        gb = grad_out.sum([0, 2, 3]).view(bias.shape) if ctx.needs_input_grad[3] else None
        gk = grad_out if (ctx.has_skip and ctx.needs_input_grad[4]) else None
        return gx, gw, gs, gb, gk, None


def hip_conv_ok(x, weight):
    return x.is_cuda and x.dtype == torch.float32 and weight.dtype == torch.float32 and weight.shape[-1] == 3


def modulated_conv2d(x, weight, s, wt, demodulate, mode, scale):
    """Dispatch: f32 3x3 -> MFMA kernel; anything else the reference allows (other kernel sizes, f64 for
    gradcheck, the unused downsample branch) -> the PyTorch-ROCm composite.  Both run on the GPU."""
    _native.require_gpu(x, 'input')
    if mode in (0, 1, 2) and hip_conv_ok(x, weight) and wt is not None:
        return ModulatedConv2dFunction.apply(x, weight, s, wt, demodulate, mode, scale)
    return modconv_composite(x, weight, s, demodulate, mode, scale)


def to_rgb(x, weight, s, bias, skip, scale):
    _native.require_gpu(x, 'input')
    if x.dtype == torch.float32 and weight.shape[-1] == 1 and weight.shape[-4] <= 4:
        return ToRGBFunction.apply(x, weight, s, bias, skip, scale)
    return _torgb_composite(x, weight, s, bias, skip, scale)
