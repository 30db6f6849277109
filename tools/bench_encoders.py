#!/usr/bin/env python3
"""Encoder timing experiments on the GPU box: NCHW vs channels_last, bilinear upsample cost."""
import os, sys, time, types
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, '3d-fm-gan_amd'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import torch
import torch.nn.functional as F
import resnet_encoder
from psp_encoder_model.encoders import psp_encoders

def timeit(fn, iters=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3

d = torch.device('cuda', 0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
torch.manual_seed(0)
x = torch.rand(B, 3, 256, 256, device=d) * 2 - 1
with torch.no_grad():
    for name, mk in (('resnet18', lambda: resnet_encoder.resnet18(tensor_encoding=True)),
                     ('psp18', lambda: psp_encoders.GradualStyleEncoder(18, 'ir_se', types.SimpleNamespace(input_nc=3, n_styles=18)))):
        m = mk().to(d).eval()
        ref = m(x)
        t = timeit(lambda: m(x))
        m2 = m.to(memory_format=torch.channels_last)
        xc = x.contiguous(memory_format=torch.channels_last)
        out = m2(xc)
        t2 = timeit(lambda: m2(xc))
        print(f'{name} B={B}: NCHW {t:.2f} ms | channels_last {t2:.2f} ms | max diff {(out - ref).abs().max().item():.2e} (ref max {ref.abs().max().item():.2e})')
    a = torch.randn(B, 512, 16, 16, device=d); b = torch.randn(B, 512, 32, 32, device=d)
    t = timeit(lambda: F.interpolate(a, size=(32, 32), mode='bilinear', align_corners=True) + b)
    a2 = torch.randn(B, 512, 32, 32, device=d); b2 = torch.randn(B, 512, 64, 64, device=d)
    t2 = timeit(lambda: F.interpolate(a2, size=(64, 64), mode='bilinear', align_corners=True) + b2)
    print(f'bilinear+add 16->32: {t:.3f} ms, 32->64: {t2:.3f} ms')
    ac = a2.contiguous(memory_format=torch.channels_last); bc = b2.contiguous(memory_format=torch.channels_last)
    t3 = timeit(lambda: F.interpolate(ac, size=(64, 64), mode='bilinear', align_corners=True) + bc)
    print(f'bilinear+add 32->64 channels_last: {t3:.3f} ms')
