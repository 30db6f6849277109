"""`op.upfirdn2d` — interface of the reference's op/upfirdn2d.py on the MI355X kernels.

Public surface kept (op/upfirdn2d.py:154-165): upfirdn2d(input, kernel, up=1, down=1, pad=(0, 0)) on an
NCHW tensor, differentiable to second order.  `upfirdn2d_op.upfirdn2d(...)` keeps the positional signature
of the reference's pybind module (op/upfirdn2d.cpp:12-23) for callers that used the extension directly.

Differences, on purpose:
  * no import-time JIT (op/upfirdn2d.py:19-25): the library is built ahead of time for gfx950;
  * no CPU branch (op/upfirdn2d.py:155-158): a CPU tensor raises RuntimeError, as the reference's
    extension does for non-CUDA tensors (op/upfirdn2d.cpp:8).
Gradient algebra (op/upfirdn2d.py:108-125, 28-94): the adjoint of upfirdn2d(k, up, down, pad) is
upfirdn2d(flip(k), up=down, down=up, g_pad) and the adjoint of that is the original op again.
"""
import torch
from torch.autograd import Function

from . import _native
from ._native import amp_fwd as _amp_fwd, amp_bwd as _amp_bwd


class _Ext:
    """Stand-in for the reference's compiled module object `upfirdn2d_op` (op/upfirdn2d.py:19)."""
    upfirdn2d = staticmethod(_native.upfirdn2d)


upfirdn2d_op = _Ext()


def _adjoint_pads(in_h, in_w, out_h, out_w, kh, kw, up, down, pad):
    # op/upfirdn2d.py:120-123
    (ux, uy), (dx, dy), (px0, px1, py0, py1) = up, down, pad
    return (kw - px0 - 1, in_w * ux - out_w * dx + px0 - ux + 1,
            kh - py0 - 1, in_h * uy - out_h * dy + py0 - uy + 1)


class UpFirDn2dBackward(Function):
    """grad_input = upfirdn2d(grad_output; flip(k), up<->down, g_pad); its own backward is the forward op."""

    @staticmethod
    @_amp_fwd
    def forward(ctx, grad_output, kernel, grad_kernel, up, down, pad, g_pad, in_size, out_size):
        go = grad_output.reshape(-1, out_size[0], out_size[1], 1)
        gi = upfirdn2d_op.upfirdn2d(go, grad_kernel, down[0], down[1], up[0], up[1], *g_pad)
        ctx.save_for_backward(kernel)
        ctx.cfg = (up, down, pad, in_size, out_size)
        return gi.view(in_size)

    @staticmethod
    @_amp_bwd
    def backward(ctx, gradgrad_input):
        kernel, = ctx.saved_tensors
        up, down, pad, in_size, out_size = ctx.cfg
        ggi = gradgrad_input.reshape(-1, in_size[2], in_size[3], 1)
        ggo = upfirdn2d_op.upfirdn2d(ggi, kernel, up[0], up[1], down[0], down[1], *pad)
        return (ggo.view(in_size[0], in_size[1], out_size[0], out_size[1]),) + (None,) * 8


class UpFirDn2d(Function):
    @staticmethod
    @_amp_fwd
    def forward(ctx, input, kernel, up, down, pad):
        batch, channel, in_h, in_w = input.shape
        kh, kw = kernel.shape
        out = upfirdn2d_op.upfirdn2d(input.reshape(-1, in_h, in_w, 1), kernel, up[0], up[1], down[0], down[1], *pad)
        out_h, out_w = out.shape[1], out.shape[2]
        if ctx.needs_input_grad[0]:      # (inference never reads the flipped taps: one launch less per call)
            ctx.save_for_backward(kernel, torch.flip(kernel, [0, 1]))
        ctx.cfg = (up, down, pad, _adjoint_pads(in_h, in_w, out_h, out_w, kh, kw, up, down, pad),
                   tuple(input.shape), (out_h, out_w))
        return out.view(-1, channel, out_h, out_w)

    @staticmethod
    @_amp_bwd
    def backward(ctx, grad_output):
        kernel, grad_kernel = ctx.saved_tensors
        up, down, pad, g_pad, in_size, out_size = ctx.cfg
        gi = UpFirDn2dBackward.apply(grad_output, kernel, grad_kernel, up, down, pad, g_pad, in_size, out_size)
        return gi, None, None, None, None


def upfirdn2d(input, kernel, up=1, down=1, pad=(0, 0)):
    _native.require_gpu(input, 'input')
    return UpFirDn2d.apply(input, kernel, (up, up), (down, down), (pad[0], pad[1], pad[0], pad[1]))
