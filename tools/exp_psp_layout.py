#!/usr/bin/env python3
"""pSp encoder body convolutions (IR-SE-50, 256^2 input), forward, fp32: NCHW vs channels_last under MIOpen's measured
find (cudnn.benchmark).  GPU box:  python tools/exp_psp_layout.py [batch]
The body runs channels_last since round 1 (14.9 -> 10.7 ms at B=8, measured on whole-encoder time); this asks per layer
whether NCHW's Winograd kernels would beat the NHWC implicit-GEMM ones on the layers that dominate (26 x 256->256 @32^2)."""
import sys

import torch
import torch.nn.functional as F

d = torch.device('cuda', 0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
torch.backends.cudnn.benchmark = True
# (name, cin, cout, stride, input size, count per forward)
LAYERS = [('stem 3->64', 3, 64, 1, 256, 1), ('64->64 s1 @256', 64, 64, 1, 256, 1), ('64->64 s2 @256', 64, 64, 2, 256, 1),
          ('64->64 s1 @128', 64, 64, 1, 128, 4), ('64->128 s1 @128', 64, 128, 1, 128, 1), ('128->128 s2 @128', 128, 128, 2, 128, 1),
          ('128->128 s1 @64', 128, 128, 1, 64, 6), ('128->256 s1 @64', 128, 256, 1, 64, 1), ('256->256 s2 @64', 256, 256, 2, 64, 1),
          ('256->256 s1 @32', 256, 256, 1, 32, 26), ('256->512 s1 @32', 256, 512, 1, 32, 1), ('512->512 s2 @32', 512, 512, 2, 32, 1),
          ('512->512 s1 @16', 512, 512, 1, 16, 4)]


def t(fn, n=20):
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


print(f'| layer (B={B}) | count | NCHW us | TFLOP/s | channels_last us | TFLOP/s | NCHW / NHWC |')
print('|---|---|---|---|---|---|---|')
tot = [0.0, 0.0]
with torch.no_grad():
    for name, cin, cout, s, r, cnt in LAYERS:
        res = []
        for fmt in (torch.contiguous_format, torch.channels_last):
            x = torch.randn(B, cin, r, r, device=d).contiguous(memory_format=fmt)
            w = torch.randn(cout, cin, 3, 3, device=d).contiguous(memory_format=fmt)
            res.append(t(lambda: F.conv2d(x, w, None, s, 1)))
        fl = 2.0 * 9 * cin * cout * B * (r // s) ** 2
        tot[0] += cnt * res[0]
        tot[1] += cnt * res[1]
        print(f'| {name} | {cnt} | {res[0]:.1f} | {fl / res[0] / 1e6:.1f} | {res[1]:.1f} | {fl / res[1] / 1e6:.1f} | {res[0] / res[1]:.2f} |', flush=True)
print(f'| **sum x count** | | {tot[0] / 1e3:.2f} ms | | {tot[1] / 1e3:.2f} ms | | {tot[0] / tot[1]:.2f} |')
