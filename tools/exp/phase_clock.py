#!/usr/bin/env python3
"""Where does a modulated-conv block spend its life?  (GPU box; experiments library: make -C 3d-fm-gan_amd/csrc experiments)

Wave 0 of every 64th block stamps s_memtime at: start, staging plan done, K loop done, last store issued, and adds up the
cycles it waited at each chunk's `s_waitcnt + s_barrier`.  Per layer: kernel time, the clock the chip held, average block
life and its split, the solo MFMA time of one wave (64 cycles x its MFMAs: what the K loop would take with the matrix
pipe to itself) and the average number of blocks alive per CU (sum of lives / kernel cycles / 256 CUs).
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
_EXP = os.path.join(ROOT, 'tools', 'exp', 'lib', 'libfmgan_hip_exp.so')
if not os.path.exists(_EXP):
    subprocess.check_call(['make', '-C', os.path.join(ROOT, '3d-fm-gan_amd', 'csrc'), 'experiments'])
os.environ['FMGAN_LIB'] = _EXP
for p in (ROOT, os.path.join(ROOT, '3d-fm-gan_amd')):
    sys.path.insert(0, p)
import torch  # noqa: E402
from op import _native  # noqa: E402

d = torch.device('cuda', 0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
# settings: label=ENV1:val,ENV2:val;...   (experiments-library switches, re-read by the library on every call)
SETTINGS = [(lab, dict(kv.split(':') for kv in envs.split(',') if kv)) for lab, _, envs in
            (x.partition('=') for x in os.environ.get('PHASE_SETTINGS', 'narrow=FMGAN_MC_WIDE:0;wide=FMGAN_MC_WIDE:1').split(';'))]
SWITCHES = sorted({k for _, e in SETTINGS for k in e})
clk = torch.zeros(16, dtype=torch.int64, device=d)
os.environ['FMGAN_MC_CLOCKPTR'] = str(clk.data_ptr())      # read once, at the library's first launch
# (res, cin, cout, mode, RM*RNP of the tile that serves it, tile positions, tile channels)
LAYERS = [(32, 512, 512, 0, 8, 256, 128), (32, 512, 512, 1, 2, 128, 64), (64, 512, 512, 0, 8, 256, 128), (64, 512, 256, 1, 2, 128, 64),
          (128, 256, 256, 0, 8, 256, 128), (128, 256, 128, 1, 2, 128, 64), (256, 128, 128, 0, 8, 256, 128), (256, 128, 64, 1, 2, 128, 64),
          (512, 64, 64, 0, 4, 256, 64), (512, 64, 32, 1, 1, 128, 32), (1024, 32, 32, 0, 2, 256, 32)]

print(f'| layer (B={B}) | setting | mode | us | GHz | blocks | life cyc | set-up | K loop | of which chunk waits | epilogue | of which before the row loop | then until its loads landed | store drain after the stamp | solo MFMA cyc | K loop / solo | blocks alive per CU |')
print('|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|')
for r, cin, cout, mode, tiles, bn, bm in LAYERS:
    x = torch.randn(B, cin, r, r, device=d)
    w = torch.randn(cout, cin, 3, 3, device=d)
    s = torch.rand(B, cin, device=d) + 0.5
    scale = 1.0 / (cin * 9) ** 0.5
    wt = _native.modconv_weight_prep(w, scale)
    dm = _native.modconv_demod(w, s, scale)
    for flags, envs in SETTINGS:
        for k in SWITCHES:
            os.environ.pop(k, None)
        os.environ.update(envs)
        for _ in range(3):
            _native.modconv2d(x, wt, s, dm, mode, precision='f32')
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            clk.zero_()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            _native.modconv2d(x, wt, s, dm, mode, precision='f32')
            b.record()
            b.synchronize()
            ts.append(a.elapsed_time(b) * 1e3)
        us = sorted(ts)[2]
        c = clk.cpu().tolist()
        n = max(1, c[5])
        ghz = c[0] / max(1, c[1]) * 0.1
        life, setup, kloop, epi, waits, epi0 = c[0] / n, c[2] / n, c[3] / n, c[4] / n, c[6] / n, c[7] / n
        solo = 64 * tiles * 9 * cin // 2
        pos = B * (r + (1 if mode == 1 else 0)) ** 2
        blocks = -(-pos // bn) * -(-cout // bm)                 # approximate (ignores the ragged edge tiles)
        alive = life * blocks / (ts[-1] * ghz * 1e3) / 256
        print(f'| {r}^2 {cin}->{cout} | {flags} | {mode} | {us:.0f} | {ghz:.2f} | ~{blocks} ({n} sampled) | {life:.0f} | {setup:.0f} | {kloop:.0f} | {waits:.0f} | '
              f'{epi:.0f} | {epi0:.0f} | {c[8] / n:.0f} | {c[9] / n:.0f} | {solo} | {kloop / solo:.2f} | {alive:.2f} |', flush=True)
    del x
