#!/usr/bin/env python3
"""Is the Discriminator's conv stack faster in channels_last?  (GPU box)  python tools/exp_d_layout.py {256|1024} [batch]

MIOpen serves most fp32 convs of D with NHWC implicit-GEMM kernels and brackets them with batched_transpose_* when the
tensors are NCHW (profiles/r02_trainstep256_kernels.md: 2 023 transposes, 62 ms of 1 580).  Before teaching the blur and
the bias+LeakyReLU kernels an NHWC layout, measure what the convolutions alone would gain: every conv of
Discriminator(size) (stylegan2.py:692-820: 1x1 stem, per ResBlock a 3x3, a 3x3 stride 2 on a blurred input and a 1x1
stride 2 skip, the final 3x3), forward + backward (data and weight gradient), NCHW vs channels_last tensors and weights.
"""
import math
import sys

import torch
import torch.nn.functional as F

d = torch.device('cuda', 0)


def layers(size):
    ch = {4: 512, 8: 512, 16: 512, 32: 512, 64: 512, 128: 256, 256: 128, 512: 64, 1024: 32}
    out = [('stem 1x1', 3, ch[size], 1, 1, 0, size)]
    cin = ch[size]
    r = size
    while r > 4:
        cout = ch[r // 2]
        out.append((f'res{r} conv1 3x3', cin, cin, 3, 1, 1, r))
        out.append((f'res{r} conv2 3x3/2 (blurred in)', cin, cout, 3, 2, 0, r + 3))
        out.append((f'res{r} skip 1x1/2 (blurred in)', cin, cout, 1, 2, 0, r + 1))
        cin, r = cout, r // 2
    out.append(('final 3x3', cin + 1, 512, 3, 1, 1, 4))
    return out


def bench(fn, n=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    b.synchronize()
    return a.elapsed_time(b) / n


def main():
    size = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    batch = int(sys.argv[2]) if len(sys.argv) > 2 else (16 if size == 256 else 8)
    print(f'# Discriminator({size}) convolutions, B={batch}, fp32, forward + backward (ms)\n')
    print('| layer | cin->cout k/s @in | NCHW | channels_last | ratio |')
    print('|---|---|---|---|---|')
    tot = [0.0, 0.0]
    for name, cin, cout, k, s, pad, r in layers(size):
        res = []
        for fmt in (torch.contiguous_format, torch.channels_last):
            x = torch.randn(batch, cin, r, r, device=d).contiguous(memory_format=fmt).requires_grad_(True)
            w = torch.randn(cout, cin, k, k, device=d).contiguous(memory_format=fmt).requires_grad_(True)
            go = torch.randn_like(F.conv2d(x, w, stride=s, padding=pad))

            def run():
                y = F.conv2d(x, w * (1.0 / math.sqrt(cin * k * k)), stride=s, padding=pad)
                torch.autograd.grad(y, [x, w], go)
            res.append(bench(run))
        tot[0] += res[0]
        tot[1] += res[1]
        print(f'| {name} | {cin}->{cout} {k}/{s} @{r} | {res[0]:.3f} | {res[1]:.3f} | {res[0] / res[1]:.2f} |')
    print(f'| **sum** | | {tot[0]:.2f} | {tot[1]:.2f} | {tot[0] / tot[1]:.2f} |')


if __name__ == '__main__':
    main()
