#!/usr/bin/env python3
"""Per-kernel micro-benchmark through the C-ABI (run on the GPU box).  Prints GB/s or TFLOP/s per shape.
usage: python tools/bench_kernels.py [ufd] [fba] [conv] [rgb] [--batch B] [--size S]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, '3d-fm-gan_amd'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import torch  # noqa: E402
from op import _native  # noqa: E402


def timeit(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(iters):
        s.record(); fn(); e.record(); e.synchronize()
        ts.append(s.elapsed_time(e))
    ts.sort()
    return ts[len(ts) // 2], ts[0]


def channels(res, mult=2):
    return {4: 512, 8: 512, 16: 512, 32: 512, 64: 256 * mult, 128: 128 * mult, 256: 64 * mult, 512: 32 * mult,
            1024: 16 * mult}[res]


def main():
    args = sys.argv[1:]
    B = int(args[args.index('--batch') + 1]) if '--batch' in args else 8
    S = int(args[args.index('--size') + 1]) if '--size' in args else 1024
    what = [a for a in args if not a.startswith('--') and not a.isdigit()] or ['ufd', 'fba', 'conv', 'rgb']
    d = torch.device('cuda', 0)
    k = torch.tensor([1., 3., 3., 1.], device=d)
    k = (k[None] * k[:, None]); k = k / k.sum() * 4
    res_list = [r for r in (8, 16, 32, 64, 128, 256, 512, 1024) if r <= S]
    if 'ufd' in what:
        print(f'== upfirdn2d blur [B*C,2H+1,2W+1]->[B*C,2H,2W], B={B}')
        for r in res_list:
            c = channels(r)
            x = torch.randn(B * c, r + 1, r + 1, 1, device=d)
            for path in (-1, 0):
                if path == 0 and r > 256:
                    continue
                ms, mn = timeit(lambda: _native.upfirdn2d(x, k, 1, 1, 1, 1, 1, 1, 1, 1, path))
                gb = 4.0 * B * c * ((r + 1) ** 2 + r * r) / 1e9
                print(f'  res {r:5d} C {c:4d} path {path:2d}: {ms * 1e3:9.1f} us  {gb / ms * 1e3:8.1f} GB/s  (min {mn * 1e3:.1f} us)')
        print(f'== upfirdn2d up=2 skip [B*3,H,W]->[B*3,2H,2W]')
        for r in res_list:
            x = torch.randn(B * 3, r // 2, r // 2, 1, device=d)
            ms, mn = timeit(lambda: _native.upfirdn2d(x, k, 2, 2, 1, 1, 2, 1, 2, 1))
            gb = 4.0 * B * 3 * ((r // 2) ** 2 + r * r) / 1e9
            print(f'  res {r:5d}: {ms * 1e3:9.1f} us  {gb / ms * 1e3:8.1f} GB/s')
    if 'fba' in what:
        print(f'== fused_bias_act / noise_bias_act [B,C,H,W], B={B}')
        for r in res_list:
            c = channels(r)
            x = torch.randn(B, c, r, r, device=d)
            b = torch.randn(c, device=d)
            e = x.new_empty(0)
            nz = torch.randn(B, 1, r, r, device=d)
            nw = torch.zeros(1, device=d)
            ms, _ = timeit(lambda: _native.fused_bias_act(x, b, e, 3, 0, 0.2, 1.41))
            ms2, _ = timeit(lambda: _native.noise_bias_act(x, nz, nw, b, 0.2, 1.41))
            gb = 8.0 * x.numel() / 1e9
            print(f'  res {r:5d} C {c:4d}: fba {ms * 1e3:9.1f} us {gb / ms * 1e3:8.1f} GB/s | noise+bias+act {ms2 * 1e3:9.1f} us {gb / ms2 * 1e3:8.1f} GB/s')
    if 'conv' in what:
        print(f'== modconv2d (mode 0 plain / mode 1 transposed), B={B}')
        cin = channels(4)
        for r in [4] + res_list:
            c = channels(r)
            for mode in (1, 0):
                if mode == 1 and r == 4:
                    continue
                ci = cin if mode == 1 else c
                h = r // 2 if mode == 1 else r
                x = torch.randn(B, ci, h, h, device=d)
                w = torch.randn(c, ci, 3, 3, device=d)
                s = torch.randn(B, ci, device=d)
                scale = 1.0 / (ci * 9) ** 0.5
                wt = _native.modconv_weight_prep(w, scale)
                dm = _native.modconv_demod(w, s, scale)
                ms, mn = timeit(lambda: _native.modconv2d(x, wt, s, dm, mode), iters=6, warm=2)
                fl = 2.0 * 9 * ci * c * B * h * h
                print(f'  res {r:5d} mode {mode} {ci:4d}->{c:4d} in {h:4d}^2: {ms * 1e3:9.1f} us  {fl / ms / 1e9:7.1f} TFLOP/s')
            cin = c
    if 'rgb' in what:
        print(f'== torgb, B={B}')
        for r in res_list:
            c = channels(r)
            x = torch.randn(B, c, r, r, device=d)
            w = torch.randn(3, c, device=d)
            s = torch.randn(B, c, device=d)
            bias = torch.zeros(3, device=d)
            skip = torch.randn(B, 3, r, r, device=d)
            ms, _ = timeit(lambda: _native.torgb(x, w, s, bias, skip, 1.0))
            gb = 4.0 * (x.numel() + 2 * skip.numel()) / 1e9
            print(f'  res {r:5d} C {c:4d}: {ms * 1e3:9.1f} us {gb / ms * 1e3:8.1f} GB/s')
    if 'rgbfuse' in what:
        print(f'== StyledConv (plain) + ToRGB: two kernels vs ToRGB in the conv epilogue, B={B}')
        for r in res_list:
            c = channels(r)
            if not _native.modconv2d_rgb_fusable(B, c, c, r, r):
                continue
            x = torch.randn(B, c, r, r, device=d)
            wgt = torch.randn(c, c, 3, 3, device=d)
            s = torch.randn(B, c, device=d)
            wt = _native.modconv_weight_prep(wgt, 1.0 / (c * 9) ** 0.5)
            dm = _native.modconv_demod(wgt, s, 1.0 / (c * 9) ** 0.5)
            noise = torch.randn(B, 1, r, r, device=d)
            nw = torch.zeros(1, device=d)
            bias = torch.zeros(c, device=d)
            rw = torch.randn(3, c, device=d)
            rb = torch.zeros(3, device=d)
            skip = torch.randn(B, 3, r, r, device=d)

            def two():
                y = _native.modconv2d(x, wt, s, dm, 0, noise=noise, noise_weight=nw, bias=bias, fuse_act=True)
                return _native.torgb(y, rw, s, rb, skip, 1.0)
            t_conv, _ = timeit(lambda: _native.modconv2d(x, wt, s, dm, 0, noise=noise, noise_weight=nw, bias=bias, fuse_act=True))
            t_two, _ = timeit(two)
            t_keep, _ = timeit(lambda: _native.modconv2d_rgb(x, wt, s, dm, noise, nw, bias, 0.2, 2 ** 0.5, rw, s, rb, skip, 1.0, True))
            t_drop, _ = timeit(lambda: _native.modconv2d_rgb(x, wt, s, dm, noise, nw, bias, 0.2, 2 ** 0.5, rw, s, rb, skip, 1.0, False))
            print(f'  res {r:5d} C {c:4d}: conv {t_conv * 1e3:8.1f} us | conv+torgb {t_two * 1e3:8.1f} us | fused {t_keep * 1e3:8.1f} us | '
                  f'fused, activation not stored {t_drop * 1e3:8.1f} us')
    if 'blurfuse' in what:
        print(f'== upsampling StyledConv tail: blur + noise + bias + lrelu in one pass (aligned-row strided input), B={B}')
        for r in res_list:
            if r < 128:
                continue
            c = channels(r)
            buf, p0, ps, rs = _native.aligned_rows_buffer(B, c, r + 1, r + 1, 1, d)
            buf.normal_()
            noise = torch.randn(B, 1, r, r, device=d)
            nw = torch.full((1,), 0.1, device=d)
            bias = torch.randn(c, device=d)
            fn = lambda: _native.blur_noise_bias_act(p0, d, B, c, r + 1, r + 1, ps, rs, k, (1, 1), noise, nw, bias, 0.2, 2 ** 0.5)
            if fn() is None:
                continue
            ms, mn = timeit(fn)
            gb = 4.0 * B * c * ((r + 1) ** 2 + r * r) / 1e9
            print(f'  res {r:5d} C {c:4d}: {ms * 1e3:9.1f} us  {gb / ms * 1e3:8.1f} GB/s  (min {mn * 1e3:.1f} us)')
    if 'io' in what:
        print('== input/output pipeline (uint8 images <-> tensors)')
        for b, hin, hout in ((32, 1024, 256), (32, 512, 256), (8, 1024, 1024), (32, 256, 256)):
            img = torch.randint(0, 256, (b, hin, hin, 3), dtype=torch.uint8, device=d)
            if hin != hout:
                ms, _ = timeit(lambda: _native.resize_images(img, hout, hout, to_tensor=True))
                gb = (img.numel() + 4.0 * b * 3 * hout * hout) / 1e9
                print(f'  resize+ToTensor+Normalize B={b} {hin}^2 -> {hout}^2: {ms * 1e3:8.1f} us {gb / ms * 1e3:8.1f} GB/s')
            else:
                ms, _ = timeit(lambda: _native.images_to_tensor(img))
                gb = (img.numel() * 5.0) / 1e9
                print(f'  ToTensor+Normalize        B={b} {hin}^2          : {ms * 1e3:8.1f} us {gb / ms * 1e3:8.1f} GB/s')
                t = _native.images_to_tensor(img)
                ms, _ = timeit(lambda: _native.tensor_to_images(t))
                print(f'  tensor2im                 B={b} {hin}^2          : {ms * 1e3:8.1f} us {gb / ms * 1e3:8.1f} GB/s')


if __name__ == '__main__':
    main()
