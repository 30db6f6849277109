#!/usr/bin/env python3
"""Why is the in-step rate of the headline blur bimodal between process launches?  (GPU box; one JSON line per run.)

Round 2 established: consecutive launches of the same bench command alternate between ~4.7 and ~5.05 TB/s for the fused
1024^2 blur, every step inside a process repeats its value, and virtual-address offsets inside one pool do not matter
(tools/exp/placement_probe.py).  This probe separates the remaining suspects inside ONE process:

  step      the real pairs1024 step (bench.py's make_step), headline blur bracketed by HIP events: the in-step rate
  arenas    N separately hipMalloc'ed arenas, each holding the aligned-row intermediate + the output; producer conv then
            fused blur on each, median of 5: do DIFFERENT PHYSICAL allocations inside one process differ?
  step2     the in-step rate again after the arenas were allocated and freed (does the process keep its mode?)
  recycle   torch.cuda.empty_cache() (all blocks back to the driver), then the step again: does a re-allocation inside
            the same process change the mode?

Run several times back to back and with pauses (tools/exp/bimodal_probe.sh) to test the hypothesis that the previous
process's memory is returned lazily, so consecutive launches alternate between two physical regions.
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, '3d-fm-gan_amd'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import torch  # noqa: E402

import bench  # noqa: E402
from op import _native  # noqa: E402

d = torch.device('cuda', 0)
B, C, H = 8, 32, 512
HEAD = (B * C, 2 * H + 1, 2 * H + 1, 2 * H, 2 * H, 1, 1, 4)
BYTES = 4.0 * HEAD[0] * (HEAD[1] * HEAD[2] + HEAD[3] * HEAD[4])


class Probe:
    def __init__(self):
        self.ev = []
        self.ptrs = []

    def begin(self, name, info):
        if name != 'upfirdn2d' or info != HEAD:
            return None
        s = torch.cuda.Event(enable_timing=True)
        s.record()
        return s

    def end(self, tok):
        if tok is None:
            return
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        self.ev.append((tok, e))

    def rate(self):
        torch.cuda.synchronize()
        ms = sorted(s.elapsed_time(e) for s, e in self.ev)
        return BYTES / (ms[len(ms) // 2] * 1e-3) / 1e9


def in_step(step, n=6):
    pr = Probe()
    _native.set_observer(pr)
    for _ in range(n):
        step()
    _native.set_observer(None)
    return round(pr.rate(), 1)


def arenas(n):
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
    L = _native.lib()
    held, res = [], []
    for _ in range(n):
        a_in = torch.empty(n_in, dtype=torch.float32, device=d)
        a_out = torch.empty(n_out, dtype=torch.float32, device=d)
        held.append((a_in, a_out))
        p0 = a_in.data_ptr() + 4
        ts = []
        for _ in range(5):
            _native.modconv2d(xin, wt, s_, dm, 1, strided_out=(p0, oh * rs, rs))
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            with _native.on_device(a_out) as stream:
                st = L.fmgan_blur_noise_bias_act_f32(p0, _native.ptr(k), _native.ptr(a_out), B, C, oh, ow, oh * rs, rs, 4, 4,
                                                     1, 1, 1, 1, _native.ptr(nz), _native.ptr(nw), _native.ptr(bias), 1, 0.2,
                                                     2 ** 0.5, stream)
            b.record()
            b.synchronize()
            assert st == 0
            ts.append(a.elapsed_time(b))
        ts.sort()
        res.append({'in': hex(a_in.data_ptr()), 'out': hex(a_out.data_ptr()), 'GBps': round(BYTES / (ts[2] * 1e-3) / 1e9, 1)})
    return res


def main():
    t0 = time.time()
    bench.warm_miopen_cache()
    os.environ.setdefault('MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_FWD', '0')
    torch.backends.cudnn.benchmark = True          # as bench.py does for the forward workloads
    nets = bench.build_models(1024, d)
    step, _ = bench.make_step(nets, 8, d, 0)
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    out = {'tag': sys.argv[1] if len(sys.argv) > 1 else '', 'start_unix': round(t0, 1)}
    out['step'] = in_step(step)
    out['reserved_gb_step'] = round(torch.cuda.memory_reserved() / 1e9, 2)
    out['arenas'] = arenas(int(os.environ.get('PROBE_ARENAS', '5')))
    out['step2'] = in_step(step)
    torch.cuda.empty_cache()
    out['recycle'] = in_step(step)
    torch.cuda.empty_cache()
    ballast = torch.empty(int(24e9) // 4, dtype=torch.float32, device=d)     # shifts where the next blocks land
    out['recycle_with_24GB_ballast'] = in_step(step)
    del ballast
    print(json.dumps(out), flush=True)


if __name__ == '__main__':
    main()
