#!/usr/bin/env python3
"""Per-layer time of the modulated-conv kernel under each pipeline variant (A = register prefetch, B/C = LDS-DMA), on the
GPU box.  One subprocess per variant (the library reads FMGAN_MC_V<mode><cfg> once).  Also checks that every variant
returns the same bits as variant A.
    python tools/bench_conv_variants.py [--batch B]            -> table + suggested default per (mode, cfg)"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# The variant / ablation / clock switches exist only in the experiments build of the library (round 3: the product library
# reads no environment variable): `make -C 3d-fm-gan_amd/csrc experiments` -> tools/exp/lib/libfmgan_hip_exp.so
_EXP = os.path.join(ROOT, 'tools', 'exp', 'lib', 'libfmgan_hip_exp.so')
if 'FMGAN_LIB' not in os.environ:
    if not os.path.exists(_EXP):
        subprocess.check_call(['make', '-C', os.path.join(ROOT, '3d-fm-gan_amd', 'csrc'), 'experiments'])
    os.environ['FMGAN_LIB'] = _EXP
VARIANTS = os.environ.get('FMGAN_BENCH_VARIANTS', 'ABC')
LAYERS = [(4, 512, 512, 0), (4, 512, 512, 1), (8, 512, 512, 0), (8, 512, 512, 1), (16, 512, 512, 0), (16, 512, 512, 1),
          (32, 512, 512, 0), (32, 512, 512, 1), (64, 512, 512, 0), (64, 512, 256, 1), (128, 256, 256, 0), (128, 256, 128, 1),
          (256, 128, 128, 0), (256, 128, 64, 1), (512, 64, 64, 0), (512, 64, 32, 1), (1024, 32, 32, 0)]


def worker(batch):
    for p in (ROOT, os.path.join(ROOT, '3d-fm-gan_amd'), os.path.join(ROOT, 'tests')):
        sys.path.insert(0, p)
    import hashlib
    import torch
    from op import _native
    d = torch.device('cuda', 0)
    out = []
    clk = torch.zeros(16, dtype=torch.int64, device=d)
    use_clk = os.environ.get('FMGAN_MC_CLOCKPTR') == str(clk.data_ptr())
    if os.environ.get('FMGAN_MC_CLOCK') == '1' and not use_clk:
        # the library reads the address once at its first launch: hand it over through the environment before that
        os.environ['FMGAN_MC_CLOCKPTR'] = str(clk.data_ptr())
        use_clk = True
    for (r, cin, cout, mode) in LAYERS:
        g = torch.Generator(device=d).manual_seed(r * 7 + mode)
        x = torch.randn(batch, cin, r, r, device=d, generator=g)
        w = torch.randn(cout, cin, 3, 3, device=d, generator=g)
        s = torch.randn(batch, cin, device=d, generator=g) * 0.5 + 1
        wt = _native.modconv_weight_prep(w, 1.0 / (cin * 9) ** 0.5)
        dm = _native.modconv_demod(w, s, 1.0 / (cin * 9) ** 0.5)
        y = _native.modconv2d(x, wt, s, dm, mode)
        digest = hashlib.sha256(y.cpu().numpy().tobytes()).hexdigest()[:16]
        for _ in range(3):
            _native.modconv2d(x, wt, s, dm, mode)
        torch.cuda.synchronize()
        ts = []
        for _ in range(7):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); _native.modconv2d(x, wt, s, dm, mode); b.record(); b.synchronize()
            ts.append(a.elapsed_time(b))
        ts.sort()
        ghz = None
        if use_clk:
            clk.zero_()
            _native.modconv2d(x, wt, s, dm, mode)
            torch.cuda.synchronize()
            c = clk.cpu().tolist()
            ghz = c[0] / max(1, c[1]) * 0.1          # s_memrealtime ticks at 100 MHz
        out.append(dict(res=r, cin=cin, cout=cout, mode=mode, us=ts[len(ts) // 2] * 1e3, digest=digest, ghz=ghz))
        del x, y
    print('RESULT ' + json.dumps(out))


def main():
    batch = int(sys.argv[sys.argv.index('--batch') + 1]) if '--batch' in sys.argv else 8
    if '--worker' in sys.argv:
        return worker(batch)
    if '--debug' in sys.argv:       # ablations (FMGAN_MC_DEBUG bits, csrc/modconv.hip MCParams::debug); results are not valid outputs
        for dbg in sys.argv[sys.argv.index('--debug') + 1].split(','):
            print(f'\n## FMGAN_MC_DEBUG={dbg}\n')
            os.environ['FMGAN_MC_DEBUG'] = dbg
            table(batch)
        return
    table(batch)


def table(batch):
    res = {}
    for v in VARIANTS:
        env = dict(os.environ)
        for m in range(3):
            for c in range(3):
                env[f'FMGAN_MC_V{m}{c}'] = v
        pr = subprocess.run([sys.executable, os.path.abspath(__file__), '--worker', '--batch', str(batch)], env=env,
                            capture_output=True, text=True)
        line = [ln for ln in pr.stdout.splitlines() if ln.startswith('RESULT ')]
        if not line:
            print(f'variant {v} failed:\n{pr.stderr[-2000:]}')
            continue
        res[v] = json.loads(line[0][7:])
    print(f'| layer (B={batch}) | mode | ' + ' | '.join(f'{v} us' for v in VARIANTS) + ' | best | TFLOP/s best | bits equal |')
    print('|---|---|' + '---|' * len(VARIANTS) + '---|---|---|')
    tot = {v: 0.0 for v in res}
    best_tot = 0.0
    flops_tot = 0.0
    for i, (r, cin, cout, mode) in enumerate(LAYERS):
        us = {v: res[v][i]['us'] for v in res}
        same = len({res[v][i]['digest'] for v in res}) == 1
        bv = min(us, key=us.get)
        fl = 2.0 * 9 * cin * cout * batch * r * r
        for v in res:
            tot[v] += us[v]
        best_tot += us[bv]
        flops_tot += fl
        ghz = res[bv][i].get('ghz')
        print(f"| {r}^2 {cin}->{cout} | {mode} | " + ' | '.join(f"{us.get(v, float('nan')):.1f}" for v in VARIANTS) +
              f" | {bv} | {fl / us[bv] / 1e6:.1f} | {'yes' if same else 'NO'} |" + (f' clock {ghz:.2f} GHz' if ghz else ''))
    print(f"| total | | " + ' | '.join(f"{tot.get(v, float('nan')):.0f}" for v in VARIANTS) + f" | {best_tot:.0f} | {flops_tot / best_tot / 1e6:.1f} | |")


if __name__ == '__main__':
    main()
