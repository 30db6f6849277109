#!/usr/bin/env python3
"""Is the fast / slow mode of the headline blur a property of single 1 GB blocks, of the read side, or of the write side?
(GPU box.)  N blocks of 1.11 GB are hipMalloc'ed one after another; the fused blur is timed for every ordered pair
(intermediate in block i, output in block j), next to each block's own streaming read (sum), write (fill) and self-copy rate.
If time(i, j) ~ f(i) + g(j), slow blocks can be told apart one by one and kept out of the allocator's hands.
    python tools/exp/block_speed_probe.py [N] > gpurun_out/r03_block_speed.md
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, '3d-fm-gan_amd')):
    sys.path.insert(0, p)
import torch  # noqa: E402
from op import _native  # noqa: E402

d = torch.device('cuda', 0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
B, C, H = 8, 32, 512
k = torch.tensor([1., 3., 3., 1.], device=d)
k = k[None] * k[:, None]
k = k / k.sum() * 4
xin = torch.randn(B, 64, H, H, device=d)
w = torch.randn(C, 64, 3, 3, device=d)
s_ = torch.randn(B, 64, device=d)
wt = _native.modconv_weight_prep(w, 1.0 / 24.0)
dm = _native.modconv_demod(w, s_, 1.0 / 24.0)
nz = torch.randn(1, 1, 2 * H, 2 * H, device=d)
nw = torch.tensor([0.3], device=d)
bias = torch.randn(C, device=d)
oh = ow = 2 * H + 1
rs = (ow + 1 + 31) // 32 * 32
n_in, n_out = B * C * oh * rs, B * C * 2 * H * 2 * H
n = max(n_in, n_out)
BYTES = 4.0 * B * C * (oh * ow + 4 * H * H)
L = _native.lib()
blocks = [torch.empty(n, dtype=torch.float32, device=d) for _ in range(N)]


def ev(fn, reps=3):
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        b.synchronize()
        ts.append(a.elapsed_time(b))
    ts.sort()
    return ts[len(ts) // 2]


def blur(i, j):
    p0 = blocks[i].data_ptr() + 4
    out = blocks[j]
    _native.modconv2d(xin, wt, s_, dm, 1, strided_out=(p0, oh * rs, rs))

    def run():
        with _native.on_device(out) as stream:
            st = L.fmgan_blur_noise_bias_act_f32(p0, _native.ptr(k), _native.ptr(out), B, C, oh, ow, oh * rs, rs, 4, 4, 1, 1, 1, 1,
                                                 _native.ptr(nz), _native.ptr(nw), _native.ptr(bias), 1, 0.2, 2 ** 0.5, stream)
        assert st == 0
    return BYTES / (ev(run) * 1e-3) / 1e9


print(f'# Per-block placement probe: {N} blocks of {n * 4 / 2**30:.2f} GiB, allocated in this order ({torch.cuda.get_device_name(0)})\n')
print('| block | address | read GB/s (sum) | write GB/s (fill) | self copy GB/s (2x bytes) |')
print('|---|---|---|---|---|')
for i, blk in enumerate(blocks):
    rd = 4.0 * n / (ev(lambda: blk.sum()) * 1e-3) / 1e9
    wr = 4.0 * n / (ev(lambda: blk.fill_(1.0)) * 1e-3) / 1e9
    half = n // 2
    cp = 8.0 * half / (ev(lambda: blk[:half].copy_(blk[half:2 * half])) * 1e-3) / 1e9
    print(f'| {i} | {blk.data_ptr():#x} | {rd:.0f} | {wr:.0f} | {cp:.0f} |')
print('\n## fused blur GB/s: rows = block of the intermediate (read), columns = block of the output (written)\n')
print('| in \\\\ out | ' + ' | '.join(str(j) for j in range(N)) + ' | row mean |')
print('|---|' + '---|' * (N + 1))
M = [[None] * N for _ in range(N)]
for i in range(N):
    for j in range(N):
        if i != j:
            M[i][j] = blur(i, j)
    vals = [v for v in M[i] if v is not None]
    print(f'| {i} | ' + ' | '.join('-' if v is None else f'{v:.0f}' for v in M[i]) + f' | {sum(vals) / len(vals):.0f} |')
cm = []
for j in range(N):
    vals = [M[i][j] for i in range(N) if i != j]
    cm.append(sum(vals) / len(vals))
print('| column mean | ' + ' | '.join(f'{v:.0f}' for v in cm) + ' | |')
