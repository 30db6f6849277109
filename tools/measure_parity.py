#!/usr/bin/env python3
"""Measured parity errors on the GPU box -> markdown (kept as profiles/rNN_parity_errors.md).

For every model-level fixture: max-abs error of the HIP(+MIOpen) result against the reference's fp32 CPU output
(tests/golden/*.npz) and against the reference run in float64 (tests/golden/fp64.npz), next to the reference's own
fp32-vs-fp64 error — all relative to max|fp64|.  Gradients (cfg3, the four training phases): per-tensor errors against
the fp64 gradients, HIP vs the reference's fp32.  The test tolerances are set from this table.
    python tools/measure_parity.py > gpurun_out/parity_errors.md
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, '3d-fm-gan_amd'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import cases  # noqa: E402
import synth  # noqa: E402
import test_hip_train as H  # noqa: E402

d = torch.device('cuda', 0)
G = lambda n: np.load(os.path.join(ROOT, 'tests', 'golden', n + '.npz'))  # noqa: E731


def rel(a, b, ref):
    return float(np.abs(np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64)).max()) / max(float(np.abs(ref).max()), 1e-30)


def row(name, hip, r32, r64):
    print(f'| {name} | {rel(hip, r32, r64):.2e} | {rel(hip, r64, r64):.2e} | {rel(r32, r64, r64):.2e} | {float(np.abs(r64).max()):.3g} |')


def main():
    import stylegan2
    from Util.network_util import Forward_Inference_3_Encoder
    g32, e32, d32, f64 = G('generator'), G('e2e'), G('discriminator'), G('fp64')
    print(f'# Measured parity errors ({torch.cuda.get_device_name(0)}, torch {torch.__version__})\n')
    print('All errors are max-abs over the fixture, divided by max|fp64 reference|.\n')
    print('| output | HIP vs reference fp32 | HIP vs reference fp64 | reference fp32 vs fp64 | max abs value |')
    print('|---|---|---|---|---|')
    with torch.no_grad():
        for c in cases.GENERATOR_CASES:
            if c['mode'] != 'latent':
                continue
            gen = H._load(stylegan2.Generator(c['size'], 512, c['n_mlp'], generator_net_shape=c['shape']), 'generator', 4)
            cin0 = c['shape'][0] if c['shape'] else 512
            lat = synth.tensor(c['name'] + '/latent', (c['b'], gen.n_latent, 512)).to(d)
            tsr = synth.tensor(c['name'] + '/tsr', (c['b'], cin0, 4, 4)).to(d)
            img = gen(None, latent_styles=[lat], input_is_latent=True, use_external_input_tensor=True,
                      external_input_tensor=tsr, randomize_noise=False)
            s = c['stride']
            row('Generator ' + c['name'], img.cpu().numpy()[..., ::s, ::s], g32[c['name'] + '/sub'], f64[c['name'] + '/sub'])
            del gen
        for c in cases.E2E_CASES:
            nets = H.build_nets(c['size'])
            p = synth.tensor(c['name'] + '/photo', (c['b'], 3, 256, 256), dist='uniform').to(d)
            r = synth.tensor(c['name'] + '/render', (c['b'], 3, 256, 256), dist='uniform').to(d)
            for k, x, key in (('e_tsr', p, 'e_tsr'), ('e_w', r, 'e_w'), ('e_wp', p, 'e_wplus')):
                row(f'{c["name"]} {key}', nets[k](x).cpu().numpy(), e32[f'{c["name"]}/{key}'], f64[f'{c["name"]}/{key}'])
            img = Forward_Inference_3_Encoder(p, r, nets['e_tsr'], nets['e_w'], nets['e_wp'], H.PinNoise(nets['g']),
                                              tsr_encode=c['tsr_encode'], sliced_layer=c['sliced_layer'],
                                              use_tanh=c['use_tanh'])
            s = c['stride']
            row(f'{c["name"]} image', img.cpu().numpy()[..., ::s, ::s], e32[c['name'] + '/sub'], f64[c['name'] + '/sub'])
            del nets
        for c in cases.DISCRIMINATOR_CASES:
            D = H._load(stylegan2.Discriminator(c['size']), 'discriminator', 8)
            x = synth.tensor(c['name'] + '/x', (c['b'], 3, c['size'], c['size']), dist='uniform').to(d)
            row('Discriminator ' + c['name'], D(x).cpu().numpy(), d32[c['name'] + '/out'], f64[c['name'] + '/out'])

    def grad_table(title, reports):
        print(f'\n## {title}\n')
        print('Per parameter tensor: e = max|sample - fp64| / max|fp64| over the strided sample; n = |norm - norm64| / norm64.\n')
        print('| network | tensors | max e HIP | max e ref-fp32 | median e HIP | median e ref-fp32 | max e_HIP / (e_ref + 2e-4) | max n HIP | max n ref-fp32 |')
        print('|---|---|---|---|---|---|---|---|---|')
        for name, rep in reports:
            eh = np.array([r[1] for r in rep]); er = np.array([r[2] for r in rep])
            nh = np.array([r[3] for r in rep]); nr = np.array([r[4] for r in rep])
            print(f'| {name} | {len(rep)} | {eh.max():.2e} | {er.max():.2e} | {np.median(eh):.2e} | {np.median(er):.2e} | '
                  f'{(eh / (er + 2e-4)).max():.2f} | {nh.max():.2e} | {nr.max():.2e} |')
            worst = sorted(rep, key=lambda r: -r[1] / (r[2] + 2e-4))[:3]
            for w in worst:
                print(f'| &nbsp;&nbsp;worst: {w[0]} | | {w[1]:.2e} | {w[2]:.2e} | | | {w[1] / (w[2] + 2e-4):.2f} | {w[3]:.2e} | {w[4]:.2e} |')

    g = G('e2e_grad')
    nets, img, loss = H.run_e2e_grad()
    c = cases.E2E_GRAD_CASE
    print(f'\ncfg3 image (fwd with autograd on): HIP vs fp64 {rel(img.detach().cpu().numpy()[..., ::c["stride"], ::c["stride"]], g["img/sub64"], g["img/sub64"]):.2e}, '
          f'reference fp32 vs fp64 {rel(g["img/sub"], g["img/sub64"], g["img/sub64"]):.2e}; '
          f'L1 loss HIP {loss.item():.7f}, reference fp32 {float(g["loss"]):.7f}, fp64 {float(g["loss64"]):.7f}')
    reports = []
    for k, m in nets.items():
        rep = []
        H.check_grads(g, k, m.named_parameters(), report=rep, margin=1e9, floor=1e9)
        reports.append((k, rep))
    grad_table('cfg3: gradients of the L1 loss through G and the three encoders (e2e_256_grad, B=2)', reports)
    del nets

    g = G('train_step')
    c = cases.TRAIN_STEP_CASE
    nets = H.build_nets(c['size'], with_d=True, n_mlp=2)
    photo, render, ref, probe = H.train_inputs()
    reports = []
    for phase in ('d', 'r1', 'g', 'ppl'):
        for m in nets.values():
            m.zero_grad(set_to_none=True)
        ld = H.run_phase(phase, nets, H.train_args(), photo, render, ref, probe, c['ppl_idx'])
        ks = ('d',) if phase in ('d', 'r1') else ('g', 'e_tsr', 'e_w', 'e_wp')
        for k in ks:
            rep = []
            H.check_grads(g, f'{phase}/{k}', nets[k].named_parameters(), report=rep, margin=1e9, floor=1e9)
            reports.append((f'{phase}: {k}', rep))
        vals = {k: (v.item() if v.numel() == 1 else v.detach().cpu().numpy().tolist()) for k, v in ld.items()}
        print(f'\nphase {phase}: HIP {vals}; fixture fp64 ' +
              str({k[len(phase) + 1:-2]: (float(g[k]) if g[k].ndim == 0 else g[k].tolist()) for k in g.files
                   if k.startswith(phase + '/') and k.endswith('64') and k.count('/') == 1}))
    grad_table('Training iteration (train_64, B=4): D loss, R1, G loss, path length', reports)


if __name__ == '__main__':
    main()
