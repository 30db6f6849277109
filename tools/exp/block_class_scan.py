#!/usr/bin/env python3
"""Class of each of N consecutively hipMalloc'ed ~1 GiB blocks relative to block 0 (GPU box): fused 1024^2 blur with the
intermediate in block 0 and the output in block k, and the reverse.  Shows the period of the fast / slow pattern along the
allocation frontier.    python tools/exp/block_class_scan.py [N] > gpurun_out/r03_block_class_scan.md"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = [sys.argv[0], sys.argv[1] if len(sys.argv) > 1 else '40']
sys.path.insert(0, os.path.join(ROOT, 'tools', 'exp'))
import io
import contextlib

buf = io.StringIO()
N = int(sys.argv[1])
# reuse the probe's setup (it prints its own tables for N blocks: silence them by running with N = 2 first)
sys.argv = [sys.argv[0], '2']
with contextlib.redirect_stdout(buf):
    import block_speed_probe as P
import torch  # noqa: E402

blocks = P.blocks
while len(blocks) < N:
    blocks.append(torch.empty(P.n, dtype=torch.float32, device=P.d))
print(f'# Block class scan: {N} blocks of {P.n * 4 / 2**30:.2f} GiB allocated back to back; fused blur GB/s with block 0\n')
print('| k | address | offset from block 0 (GiB) | in = 0, out = k | in = k, out = 0 |')
print('|---|---|---|---|---|')
a0 = blocks[0].data_ptr()
for k in range(1, N):
    print(f'| {k} | {blocks[k].data_ptr():#x} | {(blocks[k].data_ptr() - a0) / 2**30:+.2f} | {P.blur(0, k):.0f} | {P.blur(k, 0):.0f} |', flush=True)
