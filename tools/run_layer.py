#!/usr/bin/env python3
"""Launch one modulated-conv layer a few times (for rocprofv3 --pmc passes).  usage: run_layer.py RES CIN COUT MODE [BATCH] [ITERS]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, '3d-fm-gan_amd')):
    sys.path.insert(0, p)
import torch  # noqa: E402
from op import _native  # noqa: E402

r, cin, cout, mode = (int(v) for v in sys.argv[1:5])
batch = int(sys.argv[5]) if len(sys.argv) > 5 else 8
iters = int(sys.argv[6]) if len(sys.argv) > 6 else 5
d = torch.device('cuda', 0)
x = torch.randn(batch, cin, r, r, device=d)
w = torch.randn(cout, cin, 3, 3, device=d)
s = torch.randn(batch, cin, device=d) * 0.5 + 1
wt = _native.modconv_weight_prep(w, 1.0 / (cin * 9) ** 0.5)
dm = _native.modconv_demod(w, s, 1.0 / (cin * 9) ** 0.5)
for _ in range(iters):
    y = _native.modconv2d(x, wt, s, dm, mode)
torch.cuda.synchronize()
print('done', tuple(y.shape))
