#!/usr/bin/env python3
"""The two short-K layers alone, product library, 3 launches each (for rocprofv3 --pmc passes)."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, '3d-fm-gan_amd')):
    sys.path.insert(0, p)
import torch  # noqa: E402
from op import _native  # noqa: E402
d = torch.device('cuda', 0)
for r, cin, cout, mode in [(512, 64, 32, 1), (1024, 32, 32, 0), (64, 512, 512, 0)]:
    x = torch.randn(8, cin, r, r, device=d)
    w = torch.randn(cout, cin, 3, 3, device=d)
    s = torch.rand(8, cin, device=d) + 0.5
    wt = _native.modconv_weight_prep(w, 1.0 / (cin * 9) ** 0.5)
    dm = _native.modconv_demod(w, s, 1.0 / (cin * 9) ** 0.5)
    for _ in range(3):
        _native.modconv2d(x, wt, s, dm, mode, precision='f32')
    torch.cuda.synchronize()
    del x
