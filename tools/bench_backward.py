#!/usr/bin/env python3
"""Backward timing of the modulated conv layers (GPU box): HIP first-order path vs the PyTorch-ROCm composite."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, '3d-fm-gan_amd'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import torch
from op import _native, modconv
d = torch.device('cuda', 0)
def t(fn, it=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record(); b.synchronize()
    return a.elapsed_time(b) / it
B = 16
for (cin, cout, h, mode) in ((512, 512, 32, 0), (512, 512, 64, 0), (256, 256, 128, 0), (128, 128, 256, 0), (512, 512, 32, 1), (256, 128, 128, 1)):
    x = torch.randn(B, cin, h, h, device=d, requires_grad=True)
    w = torch.randn(1, cout, cin, 3, 3, device=d, requires_grad=True)
    s = torch.randn(B, cin, device=d, requires_grad=True)
    scale = 1 / (cin * 9) ** 0.5
    wt = _native.modconv_weight_prep(w.detach(), scale)
    def hip():
        y = modconv.ModulatedConv2dFunction.apply(x, w, s, wt, True, mode, scale)
        y.backward(torch.ones_like(y))
    def comp():
        y = modconv.modconv_composite(x, w, s, True, mode, scale)
        y.backward(torch.ones_like(y))
    def fwd():
        with torch.no_grad():
            modconv.ModulatedConv2dFunction.apply(x, w, s, wt, True, mode, scale)
    fl = 2 * 9 * cin * cout * B * h * h / 1e9
    th, tc, tf = t(hip), t(comp), t(fwd)
    print(f'mode {mode} {cin}->{cout} @{h}^2 B={B}: fwd {tf:.2f} ms | fwd+bwd HIP {th:.2f} ms ({3*fl/th:.0f} GF/ms-equiv) | composite {tc:.2f} ms')
    # wgrad alone
    oh = h if mode == 0 else 2 * h + 1
    go = torch.randn(B, cout, oh, oh, device=d); dm = torch.rand(B, cout, device=d) + 0.5
    tw = t(lambda: _native.modconv_wgrad(go, dm, x.detach(), s.detach(), scale, mode=mode))
    if mode == 0:
        tm = t(lambda: torch.nn.grad.conv2d_weight(x.detach() * s.detach()[:, :, None, None], (cout, cin, 3, 3), go * dm[:, :, None, None], padding=1))
    else:
        tm = t(lambda: torch.nn.grad.conv2d_weight(go * dm[:, :, None, None], (cin, cout, 3, 3), x.detach() * s.detach()[:, :, None, None], stride=2))
    print(f'     wgrad: HIP {tw:.2f} ms ({fl/tw:.1f} TF) | MIOpen conv2d_weight {tm:.2f} ms ({fl/tm:.1f} TF)')
