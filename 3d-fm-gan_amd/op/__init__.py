"""Drop-in for the reference's `op` package (op/__init__.py:10-11): same three public names,
backed by hand-written gfx950 kernels in libfmgan_hip.so instead of JIT-compiled CUDA."""
from .fused_act import FusedLeakyReLU, fused_leaky_relu
from .upfirdn2d import upfirdn2d

__all__ = ['FusedLeakyReLU', 'fused_leaky_relu', 'upfirdn2d']
