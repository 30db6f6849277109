#!/usr/bin/env python3
"""Throughput with two batches in flight: consecutive steps alternate between two 'main' streams, so step k+1's
encoders overlap step k's synthesis network (run on the GPU box).  usage: python tools/exp_two_lanes.py [lanes] [steps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402

torch.backends.cudnn.benchmark = True
d = torch.device('cuda', 0)
lanes_n = int(sys.argv[1]) if len(sys.argv) > 1 else 2
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
nets = bench.build_models(1024, d)
step, _ = bench.make_step(nets, 8, d, 0)
for _ in range(5):
    step()
torch.cuda.synchronize()
for n in (1, lanes_n):
    lanes = [torch.cuda.Stream() for _ in range(n)] if n > 1 else [torch.cuda.current_stream()]
    outs = []
    for k in range(4):
        with torch.cuda.stream(lanes[k % n]):
            outs.append(step())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        with torch.cuda.stream(lanes[k % n]):
            out = step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f'lanes {n}: {8 * steps / dt:.1f} pairs/s, {1e3 * dt / steps:.2f} ms/step')
