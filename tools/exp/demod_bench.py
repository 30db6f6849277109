#!/usr/bin/env python3
"""Time of the demodulation launch per layer width (GPU box): python tools/exp/demod_bench.py [B]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, '3d-fm-gan_amd')):
    sys.path.insert(0, p)
import torch  # noqa: E402
from op import _native  # noqa: E402
d = torch.device('cuda', 0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
for cin, cout in ((512, 512), (512, 256), (256, 256), (128, 128), (64, 64), (32, 32)):
    w = torch.randn(cout, cin, 3, 3, device=d)
    s = torch.rand(B, cin, device=d) + 0.5
    wsq = _native.modconv_wsq(w)
    sc = 1.0 / (cin * 9) ** 0.5
    ref = _native.modconv_demod(w, s, sc)
    out = _native.modconv_demod(w, s, sc, 1e-8, wsq)
    assert torch.equal(ref, out), (ref - out).abs().max()
    for name, fn in (('cached wsq', lambda: _native.modconv_demod(w, s, sc, 1e-8, wsq)), ('raw weight', lambda: _native.modconv_demod(w, s, sc))):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(50):
            fn()
        b.record()
        b.synchronize()
        print(f'{cin}->{cout} B={B} {name}: {a.elapsed_time(b) / 50 * 1e3:.1f} us per launch (back to back)')
