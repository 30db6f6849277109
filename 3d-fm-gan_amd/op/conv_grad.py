"""conv2d whose every derivative order is a native convolution primitive (host PyTorch-ROCm / MIOpen; no custom kernel).

Why.  The Discriminator's convolutions are F.conv2d (stylegan2.py:108-143) and the R1 penalty differentiates THROUGH their
first derivative (Util/training_util.py:46-52).  PyTorch's generic double-backward of convolution expresses the weight-
gradient-like terms as a forward convolution in which the batch and channel dimensions are swapped and the other
operand's whole feature map acts as the FILTER (a 1024 x 1024 "kernel" at Discriminator(1024)'s first block).  MIOpen
serves that shape with fall-back implicit-GEMM kernels: measured on MI355X, D_Reg_BackProp at 1024^2, B=8 spent 4.2 s of
6.5 s of kernel time in 12 such launches of 219-620 ms each (profiles/r03_r1_1024_kernels.md), 1.6 s per R1 step against
0.14 s for the whole D step.

Here conv2d is a family of three autograd Functions that differentiate into each other — the way the reference nests
UpFirDn2d / UpFirDn2dBackward (op/upfirdn2d.py:28-94) and op/modconv.py nests DenseConv*:

    C(x, w) = conv2d(x, w)            dC/dx . g = D(g, w)        dC/dw . g = G(x, g)
    D(g, w) = data gradient           dD/dg . h = C(h, w)        dD/dw . h = G(h, g)
    G(x, g) = weight gradient         dG/dx . v = D(g, v)        dG/dg . v = C(x, v)

with D and G issued as the library's own backward-data / backward-weight convolutions (aten.convolution_backward through
torch.nn.grad), so every order runs on the kernels MIOpen tunes for training.  Same values as F.conv2d's autograd up to
the rounding of a different kernel choice.
"""
import torch
from torch.autograd import Function
from torch.nn import functional as F
from torch.nn import grad as nn_grad


# (No autocast decorators: these wrap library convolutions, which keep following the ambient autocast state — the bf16
# leg runs them in bf16; in backward the saved fp32 operand is cast to the incoming gradient's dtype.)

class _Conv(Function):
    @staticmethod
    def forward(ctx, x, w, stride, padding):
        ctx.save_for_backward(x, w)
        ctx.cfg = (stride, padding)
        return F.conv2d(x, w, None, stride, padding)

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        stride, padding = ctx.cfg
        gx = _ConvDgrad.apply(g, w, tuple(x.shape), stride, padding) if ctx.needs_input_grad[0] else None
        gw = _ConvWgrad.apply(x, g, tuple(w.shape), stride, padding) if ctx.needs_input_grad[1] else None
        return gx, gw, None, None


class _ConvDgrad(Function):
    @staticmethod
    def forward(ctx, g, w, x_shape, stride, padding):
        ctx.save_for_backward(g, w)
        ctx.cfg = (stride, padding)
        return nn_grad.conv2d_input(x_shape, w.to(g.dtype), g, stride, padding)

    @staticmethod
    def backward(ctx, h):
        g, w = ctx.saved_tensors
        stride, padding = ctx.cfg
        gg = _Conv.apply(h, w, stride, padding) if ctx.needs_input_grad[0] else None
        gw = _ConvWgrad.apply(h, g, tuple(w.shape), stride, padding) if ctx.needs_input_grad[1] else None
        return gg, gw, None, None, None


class _ConvWgrad(Function):
    @staticmethod
    def forward(ctx, x, g, w_shape, stride, padding):
        ctx.save_for_backward(x, g)
        ctx.cfg = (stride, padding)
        return nn_grad.conv2d_weight(x.to(g.dtype), w_shape, g, stride, padding)

    @staticmethod
    def backward(ctx, v):
        x, g = ctx.saved_tensors
        stride, padding = ctx.cfg
        gx = _ConvDgrad.apply(g, v, tuple(x.shape), stride, padding) if ctx.needs_input_grad[0] else None
        gg = _Conv.apply(x, v, stride, padding) if ctx.needs_input_grad[1] else None
        return gx, gg, None, None, None


def conv2d(input, weight, bias=None, stride=1, padding=0):
    """F.conv2d(input, weight, bias, stride, padding) (square stride / padding, no dilation, one group) with the
    derivative structure described above."""
    y = _Conv.apply(input, weight, int(stride), int(padding))
    if bias is not None:
        y = y + bias.view(1, -1, 1, 1)
    return y
