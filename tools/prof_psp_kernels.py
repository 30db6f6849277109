#!/usr/bin/env python3
"""Run only E_W_Plus (pSp) N times — for `rocprofv3 --kernel-trace --stats -- python3 tools/prof_psp_kernels.py`."""
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, '3d-fm-gan_amd')):
    sys.path.insert(0, p)
import torch  # noqa: E402
from psp_encoder_model.encoders import psp_encoders  # noqa: E402

torch.backends.cudnn.benchmark = True
d = torch.device('cuda', 0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
x = torch.rand(B, 3, 256, 256, device=d) * 2 - 1
m = psp_encoders.GradualStyleEncoder(18, 'ir_se', types.SimpleNamespace(input_nc=3, n_styles=18)).to(d).eval()
with torch.no_grad():
    for _ in range(3):
        m(x)
    torch.cuda.synchronize()
    torch.cuda.profiler.start() if False else None
    for _ in range(10):
        m(x)
    torch.cuda.synchronize()
