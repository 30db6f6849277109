#!/usr/bin/env python3
"""Feasibility probe for a three-kernel Winograd F(2x2,3x3) form of the plain modulated conv (GPU box):
the 16 batched GEMMs  M_xi[cout, B*tiles] = U_xi[cout, cin] @ V_xi[cin, B*tiles]  through torch.bmm (hipBLASLt, fp32), plus the
HBM bytes the input / output transforms would move (timed as torch copies of that size), against the direct kernel."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, '3d-fm-gan_amd')):
    sys.path.insert(0, p)
import torch  # noqa: E402
from op import _native  # noqa: E402

d = torch.device('cuda', 0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8


def t(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    b.synchronize()
    return a.elapsed_time(b) / n * 1e3


print(f'| layer (B={B}) | direct kernel us | 16 x bmm us | bmm TFLOP/s | V + M bytes MB | copy of those bytes us | Winograd estimate us | Winograd form measured us | direct / measured | err direct | err Winograd |')
print('|---|---|---|---|---|---|---|---|---|---|---|')
for r, c in ((16, 512), (32, 512), (64, 512), (128, 256), (256, 128)):
    tiles = B * (r // 2) ** 2
    U = torch.randn(16, c, c, device=d)
    V = torch.randn(16, c, tiles, device=d)
    M = torch.empty(16, c, tiles, device=d)
    t_bmm = t(lambda: torch.bmm(U, V, out=M))
    fl = 16 * 2.0 * c * c * tiles
    x = torch.randn(B, c, r, r, device=d)
    w = torch.randn(c, c, 3, 3, device=d)
    s = torch.rand(B, c, device=d) + 0.5
    wt = _native.modconv_weight_prep(w, 1.0 / (c * 9) ** 0.5)
    dm = _native.modconv_demod(w, s, 1.0 / (c * 9) ** 0.5)
    t_dir = t(lambda: _native.modconv2d(x, wt, s, dm, 0, precision='f32'))
    # transforms: read x, write V (4x the input); read M (4x the output), write out
    nbytes = 4.0 * (x.numel() + V.numel() + M.numel() + x.numel())
    src = torch.empty(int(nbytes // 8), dtype=torch.float32, device=d)
    dst = torch.empty_like(src)
    t_cp = t(lambda: dst.copy_(src))
    est = t_bmm + t_cp
    del U, V, M, src, dst
    nz = torch.randn(1, 1, r, r, device=d)
    nwt = torch.tensor([0.3], device=d)
    bias = torch.randn(c, device=d)
    uu = _native.wino_weight(wt)
    t_w = t(lambda: _native.modconv2d_winograd(x, wt, s, dm, noise=nz, noise_weight=nwt, bias=bias, fuse_act=True, u=uu))
    t_dir = t(lambda: _native.modconv2d(x, wt, s, dm, 0, noise=nz, noise_weight=nwt, bias=bias, fuse_act=True, precision='f32'))
    ref = torch.nn.functional.conv2d((x * s[:, :, None, None]).double(), (w.double() / (c * 9) ** 0.5), padding=1) * dm.double()[:, :, None, None]
    mx = float(ref.abs().max())
    e_d = float((_native.modconv2d(x, wt, s, dm, 0, precision='f32').double() - ref).abs().max()) / mx
    e_w = float((_native.modconv2d_winograd(x, wt, s, dm, u=uu).double() - ref).abs().max()) / mx
    print(f'| {r}^2 {c}->{c} | {t_dir:.0f} | {t_bmm:.0f} | {fl / t_bmm / 1e6:.1f} | {nbytes / 1e6:.0f} | {t_cp:.0f} | {est:.0f} | {t_w:.0f} | {t_dir / t_w:.2f} | {e_d:.1e} | {e_w:.1e} |', flush=True)
    del ref
