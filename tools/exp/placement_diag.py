#!/usr/bin/env python3
"""Diagnostic (GPU box): what op/placement.py selects for the pairs1024 step and what the headline blur then runs at.
    FMGAN_PLACEMENT_LOG=1 python tools/exp/placement_diag.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, '3d-fm-gan_amd'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
os.environ.setdefault('FMGAN_PLACEMENT_LOG', '1')
import torch  # noqa: E402

import bench  # noqa: E402
from op import _native, placement  # noqa: E402

sys.path.insert(0, os.path.join(ROOT, 'tools', 'exp'))
from bimodal_probe import in_step  # noqa: E402

d = torch.device('cuda', 0)
bench.warm_miopen_cache()
os.environ.setdefault('MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_FWD', '0')
torch.backends.cudnn.benchmark = True
nets = bench.build_models(1024, d)
step, _ = bench.make_step(nets, 8, d, 0)
for rnd in range(4):
    placement.forget()
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    rate = in_step(step)
    ws = [(k, w) for m in nets['g'].modules() if m in placement._STORE for k, w in placement._STORE[m].items()]
    big = max(ws, key=lambda kw: kw[1].buf.numel())[1]
    bytes_ = 4.0 * 256 * (1025 * 1025 + 1024 * 1024)
    print(f'round {rnd}: in-step headline blur {rate:.0f} GB/s; selected pair probe time {big.rate:.4f} ms = '
          f'{bytes_ / (big.rate * 1e-3) / 1e9:.0f} GB/s stand-alone ({big.tried} candidates)', flush=True)
# the same step without workspaces
placement.ENABLED = False
for _ in range(2):
    step()
print('placement off:', in_step(step), 'GB/s', flush=True)
