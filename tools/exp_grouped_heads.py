#!/usr/bin/env python3
"""Would batching the pSp style heads pay?  11 heads on the 64^2 level: separate convs vs one wide conv + grouped convs
(run on the GPU box)."""
import torch
import torch.nn.functional as F

torch.backends.cudnn.benchmark = True
d = torch.device('cuda', 0)
B, G = 8, 11


def timeit(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record(); e.synchronize()
    return s.elapsed_time(e) / iters


cl = torch.channels_last
with torch.no_grad():
    x = torch.randn(B, 512, 64, 64, device=d).contiguous(memory_format=cl)
    ws = [torch.randn(512, 512, 3, 3, device=d).contiguous(memory_format=cl) for _ in range(G)]
    wcat = torch.cat(ws, 0).contiguous(memory_format=cl)
    t_sep = timeit(lambda: [F.conv2d(x, w, None, 2, 1) for w in ws])
    t_cat = timeit(lambda: F.conv2d(x, wcat, None, 2, 1))
    fl = 2 * 512 * 512 * 9 * 32 * 32 * B * G / 1e9
    print(f'first conv 64^2->32^2, {G} heads: separate {t_sep:.3f} ms ({fl / t_sep:.0f} TF) | one conv Cout={512 * G}: {t_cat:.3f} ms ({fl / t_cat:.0f} TF)')
    for res in (32, 16, 8, 4, 2):
        xs = [torch.randn(B, 512, res, res, device=d).contiguous(memory_format=cl) for _ in range(G)]
        xg = torch.cat(xs, 1).contiguous(memory_format=cl)
        t_sep = timeit(lambda: [F.conv2d(a, w, None, 2, 1) for a, w in zip(xs, ws)])
        t_grp = timeit(lambda: F.conv2d(xg, wcat, None, 2, 1, 1, G))
        fl = 2 * 512 * 512 * 9 * (res // 2) ** 2 * B * G / 1e9
        print(f'conv {res}^2->{res // 2}^2: separate {t_sep:.3f} ms ({fl / t_sep:.1f} TF) | grouped (groups={G}): {t_grp:.3f} ms ({fl / t_grp:.1f} TF)')
