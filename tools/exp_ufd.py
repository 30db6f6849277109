import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, '3d-fm-gan_amd')):
    sys.path.insert(0, p)
import torch
from op import _native
d = torch.device('cuda', 0)
k = torch.tensor([1., 3., 3., 1.], device=d); k = k[None] * k[:, None]; k = k / k.sum() * 4
def t(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); e.synchronize()
    return s.elapsed_time(e) / it
x = torch.randn(256, 1025, 1025, 1, device=d)
ms = t(lambda: _native.upfirdn2d(x, k, 1, 1, 1, 1, 1, 1, 1, 1))
gb = 4.0 * 256 * (1025 * 1025 + 1024 * 1024) / 1e9
print(f'TH={os.environ.get("FMGAN_UFD_TH")} NT={os.environ.get("FMGAN_UFD_NT")} headline: {ms*1e3:.1f} us {gb/ms*1e3:.0f} GB/s')
xa = torch.randn(256, 1025, 1024, 1, device=d)
ms = t(lambda: _native.upfirdn2d(xa, k, 1, 1, 1, 1, 0, 3, 1, 1))
print(f'  aligned rows (in_w=1024, pad_x0=0): {ms*1e3:.1f} us {4.0*256*(1025*1024+1024*1024)/1e9/ms*1e3:.0f} GB/s')
xb = torch.randn(256, 1025, 1024, 1, device=d)
ms = t(lambda: _native.upfirdn2d(xb, k, 1, 1, 1, 1, 1, 2, 1, 1))
print(f'  aligned rows, pad_x0=1: {ms*1e3:.1f} us')
y = torch.empty(256, 1024, 1024, device=d); z = torch.randn(256, 1024, 1024, device=d)
ms = t(lambda: y.copy_(z))
print(f'  torch copy 1.07GB: {ms*1e3:.1f} us {2*4.0*256*1024*1024/1e9/ms*1e3:.0f} GB/s')
# strided aligned layout (product path of the fused upsample StyledConv)
buf, p0, ps, rs = _native.aligned_rows_buffer(256, 1, 1025, 1025, 1, d)
buf.normal_()
ms = t(lambda: _native.upfirdn2d_strided(p0, d, 256, 1025, 1025, ps, rs, k, 1, 1, 1, 1))
print(f'  strided aligned layout (rs={rs}): {ms*1e3:.1f} us {gb/ms*1e3:.0f} GB/s')
# in-situ like: producer conv (64->32 @512^2, B=8) then blur, blur timed alone with events
xin = torch.randn(8, 64, 512, 512, device=d); w = torch.randn(32, 64, 3, 3, device=d); s_ = torch.randn(8, 64, device=d)
wt = _native.modconv_weight_prep(w, 1.0 / 24.0); dm = _native.modconv_demod(w, s_, 1.0 / 24.0)
ts = []
for i in range(8):
    _native.modconv2d(xin, wt, s_, dm, 1, strided_out=(p0, ps, rs))
    a, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); _native.upfirdn2d_strided(p0, d, 256, 1025, 1025, ps, rs, k, 1, 1, 1, 1); b_.record(); b_.synchronize()
    ts.append(a.elapsed_time(b_))
print('  blur right after its producer conv (us):', [round(v * 1e3) for v in ts])
ts = []
for i in range(8):
    _native.modconv2d(xin, wt, s_, dm, 1, strided_out=(p0, ps, rs))
    torch.cuda.synchronize()
    a, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); _native.upfirdn2d_strided(p0, d, 256, 1025, 1025, ps, rs, k, 1, 1, 1, 1); b_.record(); b_.synchronize()
    ts.append(a.elapsed_time(b_))
print('  blur after conv + sync (us):', [round(v * 1e3) for v in ts])
