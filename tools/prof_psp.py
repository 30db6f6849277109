import os, sys, types, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, '3d-fm-gan_amd'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import torch
from psp_encoder_model.encoders import psp_encoders
torch.backends.cudnn.benchmark = True
d = torch.device('cuda', 0)
x = torch.rand(8, 3, 256, 256, device=d) * 2 - 1
m = psp_encoders.GradualStyleEncoder(18, 'ir_se', types.SimpleNamespace(input_nc=3, n_styles=18)).to(d).eval()
with torch.no_grad():
    for _ in range(4): m(x)
    torch.cuda.synchronize()
    # per-module timing with events
    times = {}
    def hook(name):
        def pre(mod, inp):
            e = torch.cuda.Event(enable_timing=True); e.record(); mod._t0 = e
        def post(mod, inp, out):
            e = torch.cuda.Event(enable_timing=True); e.record(); times.setdefault(name, []).append((mod._t0, e))
        return pre, post
    for name, mod in m.named_modules():
        if name and name.count('.') <= 1 and (name.startswith('body.') or name.startswith('styles.') or name in ('input_layer', 'latlayer1', 'latlayer2')):
            pre, post = hook(name); mod.register_forward_pre_hook(pre); mod.register_forward_hook(post)
    for _ in range(3): m(x)
    torch.cuda.synchronize()
    tot = 0
    for k, v in times.items():
        ms = sum(a.elapsed_time(b) for a, b in v) / len(v)
        tot += ms
        print(f'{k:14s} {ms:7.3f} ms')
    print('sum', round(tot, 2))
