"""Deterministic, repo-owned synthetic tensors and weights.

Golden fixtures store only the reference's OUTPUTS; inputs and weights are regenerated on both
sides (tools/make_golden.py here, the tests on the GPU box) from a name + shape with numpy's
Philox bit generator, so 100+ MB of weights never have to be committed and nothing depends on
torch's RNG or on the order in which a module's constructor draws its parameters.
"""
import zlib

import numpy as np
import torch


def tensor(name, shape, seed=0, dist='normal', scale=1.0, shift=0.0, dtype=torch.float32):
    """Deterministic tensor for (name, shape, seed)."""
    key = (zlib.crc32(name.encode()) + 0x9E3779B1 * (seed + 1)) & 0xFFFFFFFFFFFFFFFF
    rng = np.random.Generator(np.random.Philox(key=key))
    shape = tuple(int(s) for s in shape)
    if dist == 'normal':
        a = rng.standard_normal(shape)
    elif dist == 'uniform':          # U(-1, 1)
        a = rng.uniform(-1.0, 1.0, shape)
    else:
        raise ValueError(dist)
    a = a * scale + shift
    return torch.from_numpy(np.ascontiguousarray(a)).to(dtype)


def _rule(kind, key, ref):
    """(dist-scale, shift) rule per parameter, chosen so activations stay O(1) through each net."""
    shape = tuple(ref.shape)
    leaf = key.split('.')[-1]
    if leaf == 'num_batches_tracked':
        return None
    if leaf == 'kernel':                      # Blur / Upsample FIR buffers: deterministic, keep
        return None
    if leaf == 'running_var':
        return ('uniform', 0.5, 1.0)           # U(0.5, 1.5)
    if leaf == 'running_mean':
        return ('normal', 0.1, 0.0)
    if kind in ('generator', 'discriminator'):
        if key.startswith('noises.'):
            return ('normal', 1.0, 0.0)
        if key.endswith('noise.weight'):
            return ('normal', 0.1, 0.0)        # non-zero so the noise path is exercised
        if key.endswith('modulation.bias'):
            return ('normal', 0.1, 1.0)
        if key.startswith('style.') and leaf == 'weight':
            return ('normal', 100.0, 0.0)      # EqualLinear(lr_mul=0.01): randn / lr_mul
        if leaf == 'bias':
            return ('normal', 0.1, 0.0)
        return ('normal', 1.0, 0.0)            # equalised-lr weights are N(0,1); the scale is applied in forward
    if kind == 'arcface':                      # ResNetFace: convs, one fc, BatchNorm, scalar PReLU slopes
        if ref.ndim in (2, 4):
            return ('normal', (1.0 / ref[0].numel()) ** 0.5, 0.0)
        if 'prelu' in key:
            return ('normal', 0.05, 0.25)
        return ('normal', 0.1, 0.0) if leaf == 'bias' else ('normal', 0.1, 1.0)
    # encoders: plain nn.Conv2d / BatchNorm2d / PReLU / EqualLinear
    if ref.ndim == 4:
        fan_in = shape[1] * shape[2] * shape[3]
        return ('normal', (1.0 / fan_in) ** 0.5, 0.0)
    if ref.ndim == 2:
        return ('normal', 1.0, 0.0)            # GradualStyleBlock.linear is an EqualLinear
    if leaf == 'bias':
        return ('normal', 0.1, 0.0)
    if leaf == 'weight' and ref.ndim == 1:
        if 'res_layer.2' in key or key == 'input_layer.2.weight':
            return ('normal', 0.05, 0.25)      # PReLU slopes
        return ('normal', 0.1, 1.0)            # BatchNorm gamma
    return ('normal', 1.0, 0.0)


def state_dict(kind, ref_sd, seed=0):
    """Fill a state_dict with deterministic values; names/shapes/dtypes come from `ref_sd`
    (a module's own state_dict or a {name: shape} manifest with kernel buffers supplied)."""
    out = {}
    for key, ref in ref_sd.items():
        rule = _rule(kind, key, ref)
        if rule is None:
            out[key] = ref.clone()
            continue
        dist, scale, shift = rule
        out[key] = tensor(f'{kind}/{key}', ref.shape, seed=seed, dist=dist, scale=scale, shift=shift, dtype=ref.dtype)
    return out


def manifest(sd):
    """{name: [shape]} — what the golden file stores to pin state_dict compatibility."""
    return {k: list(v.shape) for k, v in sd.items()}
