#!/usr/bin/env python3
"""Per-layer time of the modulated conv, fp32 kernel vs bf16-operand kernel (GPU box):  python tools/bench_bf16_layers.py [B]
Layers of Generator(1024) the bf16 kernel serves (position grid >= 32 wide), forward modes 0 / 1 and the stride-2 data
gradient (mode 2).  TFLOP/s = 2*9*cin*cout*positions / time."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, '3d-fm-gan_amd')):
    sys.path.insert(0, p)
import torch  # noqa: E402
from op import _native  # noqa: E402

d = torch.device('cuda', 0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
LAYERS = [(32, 512, 512, 0), (32, 512, 512, 1), (64, 512, 512, 0), (64, 512, 256, 1), (128, 256, 256, 0), (128, 256, 128, 1),
          (256, 128, 128, 0), (256, 128, 64, 1), (512, 64, 64, 0), (512, 64, 32, 1), (1024, 32, 32, 0),
          (1025, 32, 64, 2), (513, 64, 128, 2)]


def t(fn, n=5):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    b.synchronize()
    return a.elapsed_time(b) / n * 1e3


print(f'| layer (B={B}) | mode | fp32 us | TFLOP/s | bf16 us | TFLOP/s | speed-up | max |bf16-fp32| / max | bf16x3 us | eff. TFLOP/s | vs fp32 | max |bf16x3-fp32| / max |')
print('|---|---|---|---|---|---|---|---|---|---|---|---|')
for r, cin, cout, mode in LAYERS:
    x = torch.randn(B, cin, r, r, device=d)
    w = torch.randn(cout, cin, 3, 3, device=d)
    s = torch.rand(B, cin, device=d) + 0.5
    scale = 1.0 / (cin * 9) ** 0.5
    wt = _native.modconv_weight_prep(w, scale)
    dm = _native.modconv_demod(w, s, scale)
    pos = B * (r * r if mode != 2 else ((r - 3) // 2 + 1) ** 2)
    fl = 2.0 * 9 * cin * cout * pos
    assert _native.lib().fmgan_modconv2d_bf16_supported(B, cin, cout, r, r, mode)
    t32 = t(lambda: _native.modconv2d(x, wt, s, dm, mode, precision='f32'))
    t16 = t(lambda: _native.modconv2d(x, wt, s, dm, mode, precision='bf16'))
    y32 = _native.modconv2d(x, wt, s, dm, mode, precision='f32')
    y16 = _native.modconv2d(x, wt, s, dm, mode, precision='bf16')
    err = float((y16 - y32).abs().max() / y32.abs().max())
    x3 = ' - | - | - | - |'
    if _native.lib().fmgan_modconv2d_bf16x3_supported(B, cin, cout, r, r, mode):
        t3 = t(lambda: _native.modconv2d(x, wt, s, dm, mode, precision='bf16x3'))
        y3 = _native.modconv2d(x, wt, s, dm, mode, precision='bf16x3')
        e3 = float((y3 - y32).abs().max() / y32.abs().max())
        x3 = f' {t3:.0f} | {fl / t3 / 1e6:.1f} | {t32 / t3:.2f} | {e3:.1e} |'
    print(f'| {r}^2 {cin}->{cout} | {mode} | {t32:.0f} | {fl / t32 / 1e6:.1f} | {t16:.0f} | {fl / t16 / 1e6:.1f} | {t32 / t16:.2f} | {err:.1e} |' + x3)
    del x, y32, y16
