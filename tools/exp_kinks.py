#!/usr/bin/env python3
"""Experiment: are the gradient outliers of the cfg3 backward kink flips?  (GPU box; writes markdown to stdout.)

The gradient tests tolerate a few tensors beyond their floor on the hypothesis that MIOpen's forward convolutions are
not run-to-run reproducible (1e-7), so a piecewise-linear unit (PReLU / LeakyReLU / ReLU / the L1 loss's sign) whose
pre-activation lies within that noise of zero takes the other slope in one run out of two, and the weight/bias gradients
of the small layers around it jump by a discrete amount.  This script tests that instead of assuming it:

  1. run the cfg3 forward + backward (tests/test_hip_train.py::run_e2e_grad) N times in ONE process, recording the
     pre-activation of every piecewise-linear unit of the three encoders and the Generator and every parameter gradient;
  2. for every pair of runs: which units changed side, and how close to zero they were;
  3. for every pair of runs: every gradient tensor's run-to-run difference, split into pairs of runs WITH and WITHOUT a
     flipped unit in the same network, and for each tensor above the floor the flipped units of its network.

If the hypothesis holds: (a) flipped units have |pre-activation| at the forward's noise level; (b) without a flip in a
network its gradients differ by rounding only; (c) every tensor beyond the floor sits in a network with a flip.
    python tools/exp_kinks.py [runs] > gpurun_out/r03_kink_experiment.md
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, '3d-fm-gan_amd'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from torch import nn  # noqa: E402

import test_hip_train as H  # noqa: E402


class Recorder:
    """Sign bitmap + magnitude of every pre-activation, keyed by site name."""

    def __init__(self):
        self.sites = {}
        self.count = {}

    def add(self, name, pre):
        k = self.count.get(name, 0)
        self.count[name] = k + 1
        self.sites[f'{name}#{k}'] = pre.detach().float().reshape(-1).clone()


def instrument(nets, rec):
    from op.fused_act import FusedLeakyReLU
    from psp_encoder_model.encoders import psp_encoders
    handles = []
    for net_name, net in nets.items():
        for mod_name, m in net.named_modules():
            site = f'{net_name}/{mod_name}'
            if isinstance(m, FusedLeakyReLU):
                handles.append(m.register_forward_pre_hook(
                    lambda mod, inp, site=site: rec.add(site, inp[0] + mod.bias.view(1, -1, *([1] * (inp[0].ndim - 2))))))
            elif isinstance(m, (nn.ReLU, nn.PReLU, nn.LeakyReLU)):
                handles.append(m.register_forward_pre_hook(lambda mod, inp, site=site: rec.add(site, inp[0])))
    orig = psp_encoders.fused_leaky_relu

    def recording_flr(y, bias, slope, scale):           # the style heads' conv bias + LeakyReLU (functional call)
        rec.add('e_wp/heads.fused_leaky_relu', y + bias.view(*([1] * (y.ndim - 1)), -1) if y.ndim == 2 else
                y + bias.view(1, -1, *([1] * (y.ndim - 2))))
        return orig(y, bias, slope, scale)
    psp_encoders.fused_leaky_relu = recording_flr
    return handles, lambda: setattr(psp_encoders, 'fused_leaky_relu', orig)


def one_run():
    rec = Recorder()
    import test_hip_train
    build = test_hip_train.build_nets

    def build_and_instrument(size, **kw):
        nets = build(size, **kw)
        one_run.cleanup = instrument(nets, rec)
        return nets
    test_hip_train.build_nets = build_and_instrument
    try:
        nets, img, loss = H.run_e2e_grad()
    finally:
        test_hip_train.build_nets = build
        handles, restore = one_run.cleanup
        for h in handles:
            h.remove()
        restore()
    import cases
    import synth
    c = cases.E2E_GRAD_CASE
    target = synth.tensor(c['name'] + '/target', (c['b'], 3, c['size'], c['size']), dist='uniform').to(img.device)
    rec.add('loss/l1_sign', img - target)
    grads = {f'{k}/{n}': p.grad.detach().reshape(-1).clone() for k, m in nets.items() for n, p in m.named_parameters()
             if p.grad is not None}
    torch.cuda.synchronize()
    return rec.sites, grads, img.detach().clone()


def net_of(name):
    return name.split('/')[0]


def main():
    runs = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    res = [one_run() for _ in range(runs)]
    print(f'# Kink experiment: cfg3 forward + backward, {runs} runs in one process ({torch.cuda.get_device_name(0)})\n')
    sites0 = res[0][0]
    n_units = sum(v.numel() for v in sites0.values())
    print(f'{len(sites0)} activation sites, {n_units:,} piecewise-linear units recorded per run '
          f'({len(res[0][1])} gradient tensors).\n')
    # forward noise level
    fwd = max(float((res[i][2] - res[0][2]).abs().max()) for i in range(1, runs)) / float(res[0][2].abs().max())
    print(f'Forward image, worst run-to-run difference: {fwd:.2e} of max|image|.\n')
    floors = {'g': H.FLOOR, 'e_tsr': H.FLOOR_MIOPEN, 'e_w': H.FLOOR_MIOPEN, 'e_wp': H.FLOOR_MIOPEN}
    print('## Pairs of runs\n')
    print('| pair | flipped units (site: count, max |pre-activation| of the flipped units, site max) | tensors with run-to-run '
          'diff > 1e-5 / > 1e-4 / > 1e-3 of max | worst tensor |')
    print('|---|---|---|---|')
    clean, dirty = {}, {}      # net -> list of worst diffs in pairs without / with a flip in that net
    over_floor = []
    for i in range(runs):
        for j in range(i + 1, runs):
            sa, sb = res[i][0], res[j][0]
            flips = {}
            for name in sa:
                a, b = sa[name], sb[name]
                f = (a > 0) != (b > 0)
                nflip = int(f.sum())
                if nflip:
                    mag = float(torch.maximum(a.abs(), b.abs())[f].max())
                    flips[name] = (nflip, mag, float(a.abs().max()))
            flipped_nets = {net_of(n) for n in flips}
            if 'loss' in flipped_nets:      # the L1 sign feeds every network's gradient
                flipped_nets |= {'g', 'e_tsr', 'e_w', 'e_wp'}
            ga, gb = res[i][1], res[j][1]
            diffs = {}
            for name in ga:
                sc = float(ga[name].abs().max())
                diffs[name] = float((ga[name] - gb[name]).abs().max()) / max(sc, 1e-30)
            for net in ('g', 'e_tsr', 'e_w', 'e_wp'):
                worst = max(v for k, v in diffs.items() if net_of(k) == net)
                (dirty if net in flipped_nets else clean).setdefault(net, []).append(worst)
            for name, d in diffs.items():
                if d > floors[net_of(name)]:
                    over_floor.append((i, j, name, d, {k: v for k, v in flips.items() if net_of(k) in (net_of(name), 'loss')}))
            wname = max(diffs, key=diffs.get)
            fl = '; '.join(f'{k}: {v[0]}, {v[1]:.1e}, {v[2]:.1e}' for k, v in sorted(flips.items())) or 'none'
            c5, c4, c3 = (sum(1 for v in diffs.values() if v > t) for t in (1e-5, 1e-4, 1e-3))
            print(f'| {i}-{j} | {fl} | {c5} / {c4} / {c3} | {wname} {diffs[wname]:.2e} |')
    print('\n## Run-to-run gradient difference per network: pairs without vs with a flipped unit in that network\n')
    print('| network | pairs without a flip: worst tensor diff (max over pairs) | pairs with a flip: worst tensor diff (max over pairs) |')
    print('|---|---|---|')
    for net in ('g', 'e_tsr', 'e_w', 'e_wp'):
        c = clean.get(net, [])
        d = dirty.get(net, [])
        print(f"| {net} | {len(c)} pairs, {max(c) if c else float('nan'):.2e} | {len(d)} pairs, {max(d) if d else float('nan'):.2e} |")
    print('\n## Tensors beyond their test floor (HIP run vs HIP run) and the flipped units of their network\n')
    if not over_floor:
        print('none in these runs.')
    else:
        print('| pair | tensor | diff / max | flipped units in its network (site: count, max |pre| flipped) |')
        print('|---|---|---|---|')
        for i, j, name, d, fl in over_floor:
            s = '; '.join(f'{k}: {v[0]}, {v[1]:.1e}' for k, v in sorted(fl.items())) or '**NONE — not explained by a kink**'
            print(f'| {i}-{j} | {name} | {d:.2e} | {s} |')
    # units that COULD flip: within 10x the forward noise of zero
    print('\n## Units within reach of the forward noise\n')
    print('| network | units | |pre| < 1e-6 x site max | < 1e-7 x site max |')
    print('|---|---|---|---|')
    tot = {}
    for name, a in sites0.items():
        net = net_of(name)
        m = float(a.abs().max())
        t = tot.setdefault(net, [0, 0, 0])
        t[0] += a.numel()
        t[1] += int((a.abs() < 1e-6 * m).sum())
        t[2] += int((a.abs() < 1e-7 * m).sum())
    for net, t in tot.items():
        print(f'| {net} | {t[0]:,} | {t[1]} | {t[2]} |')


if __name__ == '__main__':
    main()
