"""`op.fused_leaky_relu` / `op.FusedLeakyReLU` — interface of the reference's op/fused_act.py on the MI355X kernel.

Kept: fused_leaky_relu(input, bias=None, negative_slope=0.2, scale=2**0.5) and the module
FusedLeakyReLU(channel, bias=True, negative_slope=0.2, scale=2**0.5) with its parameter named `bias`
(op/fused_act.py:96-128), differentiable to second order; `fused.fused_bias_act(...)` keeps the pybind
signature (op/fused_bias_act.cpp:11-21).
The backward saves the OUTPUT, not the input (op/fused_act.py:76): sign(out) == sign(x + b) because
scale > 0 and the slope is positive, so grad_input = act'(grad_output; ref=out).
Differences, on purpose: no import-time JIT, no CPU branch (a CPU tensor raises RuntimeError), and
`negative_slope` is honoured as the CUDA kernel does (the reference's CPU branch ignores it).
"""
import torch
from torch import nn
from torch.autograd import Function

from . import _native
from ._native import amp_fwd as _amp_fwd, amp_bwd as _amp_bwd


class _Ext:
    """Stand-in for the reference's compiled module object `fused` (op/fused_act.py:20)."""
    fused_bias_act = staticmethod(_native.fused_bias_act)


fused = _Ext()


def _reduce_dims(t):
    return [0] + list(range(2, t.ndim))


class FusedLeakyReLUFunctionBackward(Function):
    @staticmethod
    @_amp_fwd
    def forward(ctx, grad_output, out, bias, negative_slope, scale):
        ctx.save_for_backward(out)
        ctx.cfg = (negative_slope, scale)
        empty = grad_output.new_empty(0)
        if bias:
            # one pass: the kernel that writes grad_input also leaves per-block partial sums for the bias gradient
            # (the reference: a full-size reduction kernel after the elementwise one, op/fused_act.py:42-50)
            both = _native.fused_bias_act_backward(grad_output, out, negative_slope, scale)
            if both is not None:
                return both
        grad_input = fused.fused_bias_act(grad_output, empty, out, 3, 1, negative_slope, scale)
        grad_bias = grad_input.sum(_reduce_dims(grad_input)).detach() if bias else empty
        return grad_input, grad_bias

    @staticmethod
    @_amp_bwd
    def backward(ctx, gradgrad_input, gradgrad_bias):
        out, = ctx.saved_tensors
        negative_slope, scale = ctx.cfg
        ggo = fused.fused_bias_act(gradgrad_input, gradgrad_bias, out, 3, 1, negative_slope, scale)
        return ggo, None, None, None, None


class FusedLeakyReLUFunction(Function):
    @staticmethod
    @_amp_fwd
    def forward(ctx, input, bias, negative_slope, scale):
        ctx.has_bias = bias is not None
        out = fused.fused_bias_act(input, bias if ctx.has_bias else input.new_empty(0), input.new_empty(0), 3, 0,
                                   negative_slope, scale)
        ctx.save_for_backward(out)
        ctx.cfg = (negative_slope, scale)
        return out

    @staticmethod
    @_amp_bwd
    def backward(ctx, grad_output):
        out, = ctx.saved_tensors
        negative_slope, scale = ctx.cfg
        grad_input, grad_bias = FusedLeakyReLUFunctionBackward.apply(grad_output, out, ctx.has_bias, negative_slope,
                                                                     scale)
        return grad_input, (grad_bias if ctx.has_bias else None), None, None


class FusedLeakyReLU(nn.Module):
    def __init__(self, channel, bias=True, negative_slope=0.2, scale=2 ** 0.5):
        super().__init__()
        self.bias = nn.Parameter(torch.zeros(channel)) if bias else None
        self.negative_slope = negative_slope
        self.scale = scale

    def forward(self, input):
        return fused_leaky_relu(input, self.bias, self.negative_slope, self.scale)


def fused_leaky_relu(input, bias=None, negative_slope=0.2, scale=2 ** 0.5):
    _native.require_gpu(input, 'input')
    return FusedLeakyReLUFunction.apply(input, bias, negative_slope, scale)
