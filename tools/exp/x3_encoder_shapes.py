#!/usr/bin/env python3
"""Planning measurement: the split-operand contraction (fmgan_modconv2d_bf16x3, fp32-accurate) as a PLAIN 3x3 stride-1 conv
on the pSp body's layer shapes, against MIOpen's fp32 kernels (channels_last, measured find) and this repo's fp32 MFMA
kernel.  style = 1, no demodulation.  GPU box:  python tools/exp/x3_encoder_shapes.py [B]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, '3d-fm-gan_amd')):
    sys.path.insert(0, p)
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402
from op import _native  # noqa: E402

d = torch.device('cuda', 0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
torch.backends.cudnn.benchmark = True
LAYERS = [(64, 64, 256, 1), (64, 64, 128, 4), (64, 128, 128, 1), (128, 128, 64, 6), (128, 256, 64, 1), (256, 256, 32, 26), (256, 512, 32, 1)]


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


print(f'| layer (B={B}) | count | MIOpen NHWC us | own fp32 MFMA us | split-operand us | vs MIOpen | err fp32 MFMA | err split-operand | err MIOpen |')
print('|---|---|---|---|---|---|---|---|---|')
tot = [0.0, 0.0, 0.0]
with torch.no_grad():
    for cin, cout, r, cnt in LAYERS:
        x = torch.randn(B, cin, r, r, device=d)
        w = torch.randn(cout, cin, 3, 3, device=d) / (cin * 9) ** 0.5
        s = torch.ones(B, cin, device=d)
        wt = _native.modconv_weight_prep(w, 1.0)
        xc, wc = x.contiguous(memory_format=torch.channels_last), w.contiguous(memory_format=torch.channels_last)
        t_mi = t(lambda: F.conv2d(xc, wc, None, 1, 1))
        t_32 = t(lambda: _native.modconv2d(x, wt, s, None, 0, precision='f32'))
        ok3 = bool(_native.lib().fmgan_modconv2d_bf16x3_supported(B, cin, cout, r, r, 0))
        t_x3 = t(lambda: _native.modconv2d(x, wt, s, None, 0, precision='bf16x3')) if ok3 else float('nan')
        ref = F.conv2d(x.double(), w.double(), None, 1, 1)
        mx = float(ref.abs().max())
        e32 = float((_native.modconv2d(x, wt, s, None, 0, precision='f32').double() - ref).abs().max()) / mx
        e3 = float((_native.modconv2d(x, wt, s, None, 0, precision='bf16x3').double() - ref).abs().max()) / mx if ok3 else float('nan')
        emi = float((F.conv2d(xc, wc, None, 1, 1).double() - ref).abs().max()) / mx
        for i, v in enumerate((t_mi, t_32, t_x3)):
            tot[i] += cnt * v
        print(f'| {cin}->{cout} @{r} | {cnt} | {t_mi:.1f} | {t_32:.1f} | {t_x3:.1f} | {t_mi / t_x3:.2f} | {e32:.1e} | {e3:.1e} | {emi:.1e} |', flush=True)
print(f'| **sum x count** | | {tot[0] / 1e3:.2f} ms | {tot[1] / 1e3:.2f} ms | {tot[2] / 1e3:.2f} ms | {tot[0] / tot[2]:.2f} | | | |')
