#!/usr/bin/env python3
"""Per-phase time of one training iteration (GPU box):  python tools/trainstep_phases.py {256|1024} [batch] [reps] [bf16]

Times D_Loss_BackProp, D_Reg_BackProp (R1), G_Loss_BackProp, G_Reg_BackProp (path length) and the EMA separately
(synchronised around each), the way bench.py's trainstep workloads build them, and prints the lazy-regularisation
average  D + R1/16 + G + PPL/4 + EMA  next to the time of 16 consecutive Trainer.step() calls.
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, '3d-fm-gan_amd'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    size = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    name = f'trainstep{size}'
    wl = bench.WORKLOADS[name]
    batch = int(sys.argv[2]) if len(sys.argv) > 2 else wl['batch']
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    bf16 = len(sys.argv) > 4 and sys.argv[4] == 'bf16'
    bench.warm_miopen_cache()
    import train_3_encoder as T
    from Util.training_util import accumulate
    d = torch.device('cuda', 0)
    nets = bench.build_models(size, d)
    for m in nets.values():
        m.requires_grad_(True)
    step, tr = bench.make_trainstep(nets, batch, d, 0, 1, size, wl.get('loss_nets', False), 'bf16' if bf16 else 'f32')
    gen = torch.Generator(device='cpu').manual_seed(1234)
    photo = (torch.rand(batch, 3, 256, 256, generator=gen) * 2 - 1).to(d)
    render = (torch.rand(batch, 3, 256, 256, generator=gen) * 2 - 1).to(d)
    ref = (torch.rand(batch, 3, size, size, generator=gen) * 2 - 1).to(d)
    a, n, ld = tr.args, tr.nets, tr.loss_dict
    G, E_Tsr, E_W, E_W_Plus, Dn = n['G'], n['E_Tsr'], n['E_W'], n['E_W_Plus'], n['D']
    phases = {
        'D': lambda: T.D_Loss_BackProp(G, E_Tsr, E_W, E_W_Plus, Dn, photo, render, ref, a, ld, tr.d_optim),
        'R1': lambda: T.D_Reg_BackProp(ref, Dn, a, tr.d_optim),
        'G': lambda: T.G_Loss_BackProp(G, E_Tsr, E_W, E_W_Plus, Dn, photo, render, ref, a, ld, tr.g_enc_optim,
                                       tr.lpips_model, tr.face_rec_model),
        'PPL': lambda: T.G_Reg_BackProp(G, E_Tsr, E_W, E_W_Plus, photo, render, a, 0, tr.g_enc_optim),
        'EMA': lambda: accumulate(tr.g_ema, tr.bare['G'], tr.accum),
    }
    if bf16:
        from op import _native

        def reduced(fn):
            def run():
                with torch.autocast('cuda', dtype=torch.bfloat16), _native.modconv_precision('bf16'):
                    return fn()
            return run
        phases = {k: reduced(v) for k, v in phases.items()}
        name += '_bf16'
    only = os.environ.get('PHASES')          # e.g. PHASES=R1 under rocprofv3: that phase alone, no 16-iteration window
    if only:
        phases = {k: v for k, v in phases.items() if k in only.split(',')}
    res = {}
    for k, fn in phases.items():
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        first = time.perf_counter() - t0
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        ts.sort()
        res[k] = {'first_call_s': round(first, 2), 'ms': round(1e3 * ts[len(ts) // 2], 2)}
        print(f'[{name} B={batch}] {k}: first call {first:.1f} s, then {res[k]["ms"]:.1f} ms', file=sys.stderr, flush=True)
    if only:
        print(json.dumps({'workload': name, 'batch': batch, 'phases': res}))
        return
    amort = res['D']['ms'] + res['R1']['ms'] / a.d_reg_every + res['G']['ms'] + res['PPL']['ms'] / a.g_reg_every + res['EMA']['ms']
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(16):
        step()
    torch.cuda.synchronize()
    win = (time.perf_counter() - t0) / 16
    print(json.dumps({'workload': name, 'batch': batch, 'phases': res, 'amortised_ms_from_phases': round(amort, 2),
                      'ms_per_iteration_16_window': round(1e3 * win, 2), 'pairs_per_s': round(batch / win, 2),
                      'peak_hbm_gb': round(torch.cuda.max_memory_allocated() / 1e9, 1)}))


if __name__ == '__main__':
    main()
