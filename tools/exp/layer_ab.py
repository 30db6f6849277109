#!/usr/bin/env python3
"""Generator(1024)'s modulated convs as the forward runs them, under experiment switches (GPU box; experiments library).

Layers: the transposed convs 4^2..512^2, the plain convs 4^2..128^2, and the plain convs with the fused ToRGB epilogue at
256^2, 512^2 and 1024^2 (the last without an activation output).  Settings: LAYER_SETTINGS="label=ENV:val,ENV:val;..."
(switches of the experiments build, re-read per call).  Microseconds, median of 7, B = 8; same-bits check per layer.
"""
import hashlib
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
_EXP = os.path.join(ROOT, 'tools', 'exp', 'lib', 'libfmgan_hip_exp.so')
if 'FMGAN_LIB' not in os.environ:
    if not os.path.exists(_EXP):
        subprocess.check_call(['make', '-C', os.path.join(ROOT, '3d-fm-gan_amd', 'csrc'), 'experiments'])
    os.environ['FMGAN_LIB'] = _EXP
for p in (ROOT, os.path.join(ROOT, '3d-fm-gan_amd')):
    sys.path.insert(0, p)
import torch  # noqa: E402
from op import _native  # noqa: E402

d = torch.device('cuda', 0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
SETTINGS = [(lab, dict(kv.split(':') for kv in envs.split(',') if kv)) for lab, _, envs in
            (x.partition('=') for x in os.environ.get('LAYER_SETTINGS', 'default=').split(';'))]
SWITCHES = sorted({k for _, e in SETTINGS for k in e})
CH = {4: 512, 8: 512, 16: 512, 32: 512, 64: 512, 128: 256, 256: 128, 512: 64, 1024: 32}
LAYERS = [(4, 512, 512, 0, False)]
for r in (4, 8, 16, 32, 64, 128, 256, 512):
    LAYERS.append((r, CH[r], CH[2 * r], 1, False))
    LAYERS.append((2 * r, CH[2 * r], CH[2 * r], 0, 2 * r >= 256))


def med(fn, n=7):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        b.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    ts.sort()
    return ts[n // 2]


print(f'| layer (B={B}) | mode | ' + ' | '.join(lab for lab, _ in SETTINGS) + ' | bits |')
print('|---|---|' + '---|' * (len(SETTINGS) + 1))
tot = [0.0] * len(SETTINGS)
for r, cin, cout, mode, rgb in LAYERS:
    x = torch.randn(B, cin, r, r, device=d)
    w = torch.randn(cout, cin, 3, 3, device=d)
    s = torch.rand(B, cin, device=d) + 0.5
    scale = 1.0 / (cin * 9) ** 0.5
    wt = _native.modconv_weight_prep(w, scale)
    dm = _native.modconv_demod(w, s, scale)
    nz = torch.randn(1, 1, r, r, device=d)
    nw = torch.tensor([0.3], device=d)
    bias = torch.randn(cout, device=d)
    rw = torch.randn(3, cout, device=d)
    rs = torch.rand(B, cout, device=d) + 0.5
    rb = torch.randn(3, device=d)
    sk = torch.randn(B, 3, r, r, device=d)
    if rgb:
        assert _native.modconv2d_rgb_fusable(B, cin, cout, r, r)
        keep = r < 1024

        def run():
            return _native.modconv2d_rgb(x, wt, s, dm, nz, nw, bias, 0.2, 2 ** 0.5, rw, rs, rb, sk, 1.0 / cout ** 0.5, keep_out=keep)
    elif mode == 0:
        def run():
            return (_native.modconv2d(x, wt, s, dm, 0, noise=nz, noise_weight=nw, bias=bias, fuse_act=True, precision='f32'),)
    else:
        def run():
            return (_native.modconv2d(x, wt, s, dm, 1, precision='f32'),)
    row, digests = [], set()
    for i, (lab, envs) in enumerate(SETTINGS):
        for k in SWITCHES:
            os.environ.pop(k, None)
        os.environ.update(envs)
        t = med(run)
        h = hashlib.sha256()
        for o in run():
            if o is not None:
                h.update(o.cpu().numpy().tobytes())
        digests.add(h.hexdigest())
        row.append(t)
        tot[i] += t
    print(f"| {r}^2 {cin}->{cout}{' +RGB' if rgb else ''} | {mode} | " + ' | '.join(f'{t:.0f}' for t in row) +
          f" | {'same' if len(digests) == 1 else 'DIFFER'} |", flush=True)
    del x
print('| total | | ' + ' | '.join(f'{t:.0f}' for t in tot) + ' | |')
