#!/usr/bin/env python3
"""Does the in-step time of the headline blur depend on WHERE its two 1 GB buffers sit?  (GPU box.)
The producer conv writes the aligned-row intermediate, the fused blur reads it and writes the activation; both buffers are
carved out of one big pool at varying offsets; the blur is timed with events right after its producer."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, '3d-fm-gan_amd')):
    sys.path.insert(0, p)
import torch
from op import _native
d = torch.device('cuda', 0)
k = torch.tensor([1., 3., 3., 1.], device=d); k = k[None] * k[:, None]; k = k / k.sum() * 4
B, C, H = 8, 32, 512
xin = torch.randn(B, 64, H, H, device=d); w = torch.randn(C, 64, 3, 3, device=d); s_ = torch.randn(B, 64, device=d)
wt = _native.modconv_weight_prep(w, 1.0 / 24.0); dm = _native.modconv_demod(w, s_, 1.0 / 24.0)
nz = torch.randn(1, 1, 2 * H, 2 * H, device=d); nw = torch.tensor([0.3], device=d); bias = torch.randn(C, device=d)
oh = ow = 2 * H + 1
rs = (ow + 1 + 31) // 32 * 32
n_in = B * C * oh * rs
n_out = B * C * 2 * H * 2 * H
MB = 1 << 18   # floats per MiB
pool = torch.empty(n_in + n_out + 2048 * MB, dtype=torch.float32, device=d)
L = _native.lib()


def run(off_in, off_out, reps=6):
    buf = pool[off_in:off_in + n_in]
    out = pool[off_out:off_out + n_out].view(B, C, 2 * H, 2 * H)
    p0 = buf.data_ptr() + 4
    ts = []
    for _ in range(reps):
        _native.modconv2d(xin, wt, s_, dm, 1, strided_out=(p0, oh * rs, rs))
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        with _native.on_device(out) as stream:
            st = L.fmgan_blur_noise_bias_act_f32(p0, _native.ptr(k), _native.ptr(out), B, C, oh, ow, oh * rs, rs, 4, 4, 1, 1, 1, 1,
                                                 _native.ptr(nz), _native.ptr(nw), _native.ptr(bias), 1, 0.2, 2 ** 0.5, stream)
        b.record(); b.synchronize()
        assert st == 0
        ts.append(a.elapsed_time(b) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


base_out = n_in + 1024 * MB
print('in_off MiB | out_off MiB (after in + 1 GiB) | blur us (median of 6, right after its producer)')
for oi in (0, 1, 2, 16, 100, 256, 512, 1000):
    row = []
    for oo in (0, 1, 2, 16, 100, 256, 512, 1000):
        row.append(run(oi * MB, base_out + oo * MB - (oi * MB if False else 0)))
    print(f'{oi:5d} | ' + ' '.join(f'{v:6.1f}' for v in row))
