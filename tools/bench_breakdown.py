#!/usr/bin/env python3
"""Where a pairs1024 step goes: each encoder alone, the generator alone, the whole forward (run on the GPU box).
usage: python tools/bench_breakdown.py [size] [batch]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402  (puts the package on sys.path)


def timeit(fn, iters=10, warm=4):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record(); e.synchronize()
    return s.elapsed_time(e) / iters


def main():
    size = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    batch = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    torch.backends.cudnn.benchmark = True
    d = torch.device('cuda', 0)
    nets = bench.build_models(size, d)
    step, (photo, render) = bench.make_step(nets, batch, d, 0)
    with torch.no_grad():
        tsr = nets['e_tsr'](photo)
        w = nets['e_w'](render)
        wp = nets['e_wp'](photo)
        lat = w.unsqueeze(1) * wp
        g = nets['g']
        rows = [
            ('e_tsr (resnet18 -> [N,512,4,4])', lambda: nets['e_tsr'](photo)),
            ('e_w   (resnet18 -> [N,512])', lambda: nets['e_w'](render)),
            ('e_wp  (pSp GradualStyleEncoder)', lambda: nets['e_wp'](photo)),
            ('generator', lambda: g(noise_z=None, latent_styles=[lat], input_is_latent=True,
                                    use_external_input_tensor=True, external_input_tensor=tsr)),
            ('whole forward', step),
        ]
        for name, fn in rows:
            print(f'{name:36s} {timeit(fn):8.3f} ms', flush=True)


if __name__ == '__main__':
    main()
