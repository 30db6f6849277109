#!/usr/bin/env python3
"""Upper bound of encoder/generator pipelining: pSp encoder and Generator(1024) on two streams with independent inputs
vs back to back (run on the GPU box)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402


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


torch.backends.cudnn.benchmark = True
d = torch.device('cuda', 0)
nets = bench.build_models(1024, d)
step, (photo, render) = bench.make_step(nets, 8, d, 0)
with torch.no_grad():
    tsr = nets['e_tsr'](photo)
    lat = nets['e_w'](render).unsqueeze(1) * nets['e_wp'](photo)
    g = nets['g']
    gen = lambda: g(noise_z=None, latent_styles=[lat], input_is_latent=True, use_external_input_tensor=True,
                    external_input_tensor=tsr)
    enc = lambda: nets['e_wp'](photo)
    s2 = torch.cuda.Stream()

    def both():
        s2.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s2):
            enc()
        gen()
        torch.cuda.current_stream().wait_stream(s2)

    def seq():
        enc(); gen()
    print(f'pSp {timeit(enc):.2f} ms | generator {timeit(gen):.2f} ms | back to back {timeit(seq):.2f} ms | '
          f'two streams {timeit(both):.2f} ms | whole forward {timeit(step):.2f} ms')
