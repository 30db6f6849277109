#!/usr/bin/env python3
"""Does MIOpen's exhaustive tuning (MIOPEN_FIND_ENFORCE=3) find faster kernels for the pSp body's fp32 NHWC convolutions than
its measured find does?  GPU box:  python tools/exp_miopen_tune.py {tune|use|plain} [batch]
  tune   run every layer once with MIOPEN_FIND_ENFORCE=3 into a scratch user db (gpurun_out/miopen_tuned), then time it
  use    time with that db, no enforce (what a later process would get)
  plain  time with an empty scratch db (the untuned baseline on the same box)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
mode = sys.argv[1] if len(sys.argv) > 1 else 'plain'
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
d = os.path.join(ROOT, 'gpurun_out', 'miopen_tuned' if mode != 'plain' else 'miopen_plain')
os.makedirs(os.path.join(d, 'db'), exist_ok=True)
os.makedirs(os.path.join(d, 'cache'), exist_ok=True)
os.environ['MIOPEN_USER_DB_PATH'] = os.path.join(d, 'db')
os.environ['MIOPEN_CUSTOM_CACHE_DIR'] = os.path.join(d, 'cache')
if mode == 'tune':
    os.environ['MIOPEN_FIND_ENFORCE'] = '3'
import time  # noqa: E402

import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

dev = torch.device('cuda', 0)
torch.backends.cudnn.benchmark = True
LAYERS = [('64->64 s1 @256', 64, 64, 1, 256, 1), ('64->64 s2 @256', 64, 64, 2, 256, 1), ('64->64 s1 @128', 64, 64, 1, 128, 4),
          ('64->128 s1 @128', 64, 128, 1, 128, 1), ('128->128 s2 @128', 128, 128, 2, 128, 1), ('128->128 s1 @64', 128, 128, 1, 64, 6),
          ('128->256 s1 @64', 128, 256, 1, 64, 1), ('256->256 s2 @64', 256, 256, 2, 64, 1), ('256->256 s1 @32', 256, 256, 1, 32, 26),
          ('256->512 s1 @32', 256, 512, 1, 32, 1), ('512->512 s2 @32', 512, 512, 2, 32, 1), ('512->512 s1 @16', 512, 512, 1, 16, 4)]
print(f'| layer (B={B}, channels_last, {mode}) | count | us | TFLOP/s | first call s |')
print('|---|---|---|---|---|')
tot = 0.0
with torch.no_grad():
    for name, cin, cout, s, r, cnt in LAYERS:
        x = torch.randn(B, cin, r, r, device=dev).contiguous(memory_format=torch.channels_last)
        w = torch.randn(cout, cin, 3, 3, device=dev).contiguous(memory_format=torch.channels_last)
        t0 = time.time()
        F.conv2d(x, w, None, s, 1)
        torch.cuda.synchronize()
        first = time.time() - t0
        for _ in range(3):
            F.conv2d(x, w, None, s, 1)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20):
            F.conv2d(x, w, None, s, 1)
        b.record()
        b.synchronize()
        us = a.elapsed_time(b) / 20 * 1e3
        fl = 2.0 * 9 * cin * cout * B * (r // s) ** 2
        tot += cnt * us
        print(f'| {name} | {cnt} | {us:.1f} | {fl / us / 1e6:.1f} | {first:.1f} |', flush=True)
print(f'| **sum x count** | | {tot / 1e3:.2f} ms | | |')
