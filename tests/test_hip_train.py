"""GPU parity of the TRAINING side of the hot path against fixtures generated from the imported reference
(tools/make_golden.py: gen_e2e_grad, gen_train_step):

  * BASELINE config 3 — full 3-encoder forward + backward @256^2 through the data_parallel wrappers: every parameter
    gradient of E_Tsr / E_W / E_W_Plus / G vs the reference's, judged against the reference's own fp64 run;
  * the four gradient computations of one training iteration (D loss, R1, G loss, path length) of train_3_encoder.py
    (this build: 3d-fm-gan_amd/train_3_encoder.py) vs the reference modules + the reference's loss functions;
  * a world-size-2 run (two processes on this GPU, gloo) of those phases == the single-process run on the
    concatenated batch, for DDP and for each explicit gather_grad algorithm.

Tolerance rule for gradients.  The reference's fp32 CPU gradients themselves differ from its fp64 gradients by up to
3e-2 of a tensor's max (L1's sign(), long reductions with cancellation), so a fixed relative tolerance vs the fp32
fixture would be either vacuous or flaky.  Each tensor is therefore held to
      |hip - fp64|  <=  MARGIN * |ref_fp32 - fp64|  +  FLOOR * max|fp64|
i.e. "no farther from the exact value than the reference itself, up to a small factor".  Floors, from measurement
(profiles/r03_kink_experiment.md, profiles/r02_parity_errors.md, tools/measure_parity.py):
  * Generator alone on fixture inputs (only this repo's kernels run): 2e-4, no exceptions (OWN_FLOOR; test below);
  * Generator tensors inside the end-to-end path (its latents come from MIOpen encoders and carry their 1e-7 run-to-run
    noise): FLOOR = 5e-4 — measured 2.3e-4 vs fp64 and 2.5e-4 run to run;
  * encoder tensors (MIOpen fp32 wgrad kernels, not run-to-run reproducible): FLOOR_MIOPEN = 4e-3 — measured 2.0e-3 vs fp64
    on resnet conv1.weight and 1.3e-3 on a pSp head conv, 5.7e-5 run to run when no unit flips; norms agree to 4e-5.

Kinks — tested, not assumed (profiles/r03_kink_experiment.md: four runs in one process, 113 M piecewise-linear units
recorded per run).  MIOpen's forward convs are not run-to-run reproducible (1.8e-6 of the image), and in EVERY pair of runs a
handful of PReLU / LeakyReLU units of the pSp encoder (1-2 per affected site, |pre-activation| <= 1.2e-6 against site maxima
of 3-6) land on the other side of zero.  Networks in which no unit flipped (both ResNets, all six pairs) agree to 5.7e-5
run to run; the pSp encoder, with flips in all six pairs, shows 2-3 tensors per pair beyond 1e-3 and up to 1.5e-2 (head
convs `styles.N.convs.0.weight`: 2x2 / 4x4 feature maps, where one unit is a visible fraction of a weight gradient).  The
reference's own fp32-vs-fp64 differences have the same cause.  So:
  * the allowance applies to ENCODER tensors only (prefix e_*); Generator tensors get none — flips inside G (3-5 units per
    128^2-256^2 layer) move its gradients by 2.5e-4 at most, inside FLOOR;
  * at most KINK_TENSORS tensors per backward pass may exceed the floor, each by at most KINK_MAX = 0.1 of its max (largest
    observed against the fp64 fixture: 7.0e-2 on g/e_wp/styles.5.convs.0.weight at B=2, where one unit of an 8x8 map is a
    large share of a weight gradient; 1.5e-2 run to run) with its norm within KINK_NORM;
  * and the deviation must be LOCAL: at most KINK_ELEMS of the tensor's 48 strided sample elements may lie beyond the
    floor, all others must meet it.  One flipped unit changes the gradient of ONE output channel of its layer (one row of
    a conv / linear weight, one element of a bias), and consecutive sample elements are 10+ rows apart, so a flip shows in
    one or two sample elements — a wrong kernel shows in most of them.  (A re-run based rule — "the tensor must differ
    between two HIP runs" — was tried first and is wrong: a kink need not flip run to run.  At Generator(1024), B=2,
    `g/e_wp/styles.5.convs.0.weight` sits 7.0e-2 from the fp64 fixture in one sample element in BOTH of two runs: the
    fp32 pre-activation lands on the other side of zero than the fp64 one, reproducibly.)
"""
import os
import sys
import types

import numpy as np
import pytest
import torch

import cases
import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

MARGIN, FLOOR, FLOOR_MIOPEN, FLOOR_NORM, FLOOR_SCALAR = 4.0, 5e-4, 4e-3, 5e-4, 8e-3
KINK_TENSORS, KINK_MAX, KINK_NORM, KINK_ELEMS = 6, 0.1, 2e-2, 3     # one flip shows in the weight AND the bias gradient of its layer


def dev():
    return torch.device('cuda', 0)


def _load(module, kind, seed):
    module.load_state_dict(synth.state_dict(kind, module.state_dict(), seed=seed))
    return module.to(dev()).eval()


def build_nets(size, with_d=False, n_mlp=8):
    import stylegan2
    import resnet_encoder
    from psp_encoder_model.encoders import psp_encoders
    n_latent = int(np.log2(size)) * 2 - 2
    nets = dict(
        e_tsr=_load(resnet_encoder.resnet18(tensor_encoding=True, tensor_transform=False), 'resnet', 5),
        e_w=_load(resnet_encoder.resnet18(tensor_encoding=False, tensor_transform=False), 'resnet', 6),
        e_wp=_load(psp_encoders.GradualStyleEncoder(18, 'ir_se', types.SimpleNamespace(input_nc=3, n_styles=n_latent)),
                   'psp', 7),
        g=_load(stylegan2.Generator(size, 512, n_mlp), 'generator', 4))
    if with_d:
        nets['d'] = _load(stylegan2.Discriminator(size), 'discriminator', 8)
    return nets


class PinNoise(torch.nn.Module):
    """Forward_Inference_3_Encoder never passes noise; the fixtures pinned randomize_noise=False.
    `.module` is what the reference reads n_latent from (Util/network_util.py:317-318)."""

    def __init__(self, g, call=None):
        super().__init__()
        self.module = g
        self._call = [call if call is not None else g]      # in a list: not a registered submodule twice

    def forward(self, **kw):
        return self._call[0](randomize_noise=False, **kw)


def check_grads(g, prefix, named_params, report=None, margin=MARGIN, floor=None, kinks=None, floor_norm=None):
    """Every parameter gradient vs the fixture (strided sample + norm) under the rule in the module docstring.
    `kinks`: a list collecting the tensors that exceed the floor by a kink-sized amount (the caller bounds their number);
    None = no such allowance."""
    n = 0
    worst = 0.0
    auto_floor = floor is None
    if auto_floor:
        net = prefix.split('/')[-1]
        # the Discriminator's weight gradients are MIOpen's too; measured inside 1e-3 (round 2 gate, kept)
        floor = FLOOR_MIOPEN if net.startswith('e_') else (1e-3 if net == 'd' else FLOOR)
    for name, p in named_params:
        key = f'{prefix}/{name}'
        if key + '/s' not in g.files:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, f'{key}: reference has no gradient here'
            continue
        assert p.grad is not None, f'{key}: no gradient'
        s, nrm = cases.grad_sample(p.grad)
        s64, n64 = g[key + '/s64'], float(g[key + '/n64'])
        s32, n32 = g[key + '/s'], float(g[key + '/n'])
        scale = max(float(np.abs(s64).max()), 1e-30)
        e_hip = float(np.abs(s - s64).max()) / scale
        e_ref = float(np.abs(s32 - s64).max()) / scale
        en_hip, en_ref = abs(nrm - n64) / max(n64, 1e-30), abs(n32 - n64) / max(n64, 1e-30)
        if report is not None:
            report.append((key, e_hip, e_ref, en_hip, en_ref))
        fl, fn = floor, (FLOOR_NORM if floor_norm is None else floor_norm)
        if auto_floor and p.numel() == 1:
            # NoiseInjection.weight: ONE scalar = sum over B*C*H*W signed products grad*noise that cancel almost
            # completely (|sum| / sum|terms| ~ 1e-4, tools/measure_parity.py prints it); its relative error is the
            # relative error of the upstream gradient amplified by that cancellation.  Measured 2.0e-3 (reference 1e-4).
            fl = fn = FLOOR_SCALAR
        over = e_hip > margin * e_ref + fl or en_hip > margin * en_ref + fn
        encoder = prefix.split('/')[-1].startswith('e_')
        outliers = int((np.abs(s - s64) / scale > margin * e_ref + fl).sum())
        if (kinks is not None and encoder and over and e_hip <= KINK_MAX and en_hip <= margin * en_ref + KINK_NORM
                and outliers <= KINK_ELEMS):
            # (a flipped unit of a tiny layer — an SE gate's ReLU has B x C/16 outputs — moves a whole row of the weight
            # gradient: the strided sample may miss it while the norm shows it, e.g. ppl/e_wp/body.0.res_layer.5.fc1.weight
            # norm 1.3e-3 off in one run)
            kinks.append((key, e_hip, en_hip, outliers))
        else:
            assert e_hip <= margin * e_ref + fl, f'{key}: sample err {e_hip:.3e} vs reference-fp32 err {e_ref:.3e}'
            assert en_hip <= margin * en_ref + fn, f'{key}: norm err {en_hip:.3e} vs reference-fp32 err {en_ref:.3e}'
        worst = max(worst, e_hip)
        n += 1
    return n, worst


def confirm_kinks(kinks):
    """At most KINK_TENSORS encoder tensors per backward pass may carry a (local, bounded) kink deviation."""
    assert len(kinks) <= KINK_TENSORS, kinks


def run_e2e_grad(report=None):
    from Miscellaneous import distributed as D
    from Util.network_util import Forward_Inference_3_Encoder
    from Util.training_util import L1_Loss
    c = cases.E2E_GRAD_CASE
    nets = build_nets(c['size'])
    for m in nets.values():
        m.requires_grad_(True)
    wrapped = {k: D.data_parallel(m, dev()) for k, m in nets.items()}
    assert all(hasattr(w, 'module') for w in wrapped.values())
    p = synth.tensor(c['name'] + '/photo', (c['b'], 3, 256, 256), dist='uniform').to(dev())
    r = synth.tensor(c['name'] + '/render', (c['b'], 3, 256, 256), dist='uniform').to(dev())
    target = synth.tensor(c['name'] + '/target', (c['b'], 3, c['size'], c['size']), dist='uniform').to(dev())
    img = Forward_Inference_3_Encoder(p, r, wrapped['e_tsr'], wrapped['e_w'], wrapped['e_wp'],
                                      PinNoise(nets['g'], wrapped['g']), tsr_encode=c['tsr_encode'],
                                      sliced_layer=c['sliced_layer'], use_tanh=c['use_tanh'])
    loss = L1_Loss(img, target)
    loss.backward()
    return nets, img, loss


def test_cfg3_forward_backward_golden(golden):
    """BASELINE config 3: (photo, render) -> image -> L1 -> backward through G and the three encoders, called through
    Miscellaneous.distributed.data_parallel wrappers (train_3_encoder.py:495-558 with the L1 term)."""
    g = golden('e2e_grad')
    nets, img, loss = run_e2e_grad()
    c = cases.E2E_GRAD_CASE
    a = img.detach().cpu().numpy()[..., ::c['stride'], ::c['stride']]
    ref64 = g['img/sub64']
    scale = float(np.abs(ref64).max())
    e_hip, e_ref = np.abs(a - ref64).max() / scale, np.abs(g['img/sub'] - ref64).max() / scale
    assert e_hip <= MARGIN * e_ref + 2e-5, (e_hip, e_ref)
    np.testing.assert_allclose(loss.item(), float(g['loss64']), rtol=2e-5)
    total, kinks = 0, []
    for k, m in nets.items():
        n, _ = check_grads(g, k, m.named_parameters(), kinks=kinks)
        total += n
    confirm_kinks(kinks)
    assert total == len([k for k in g.files if k.endswith('/n64')])     # every fixture tensor was compared
    assert nets['g'].style[1].weight.grad is None                       # mapping network unused (input_is_latent)


def run_generator_only():
    """Forward + backward of the Generator ALONE on the encoder outputs the reference computed (fixture
    e2e_grad_latents): the co-modulated latent and the input tensor are constants, so every kernel between them and the
    `g/*` gradients is this repo's own — bit-reproducible, no MIOpen."""
    import stylegan2
    from Util.training_util import L1_Loss
    c = cases.E2E_GRAD_CASE
    lat = np.load(os.path.join(ROOT, 'tests', 'golden', 'e2e_grad_latents.npz'))
    G = _load(stylegan2.Generator(c['size'], 512, 8), 'generator', 4).requires_grad_(True)
    tsr = torch.from_numpy(lat['e_tsr']).to(dev())
    latent = torch.from_numpy(lat['e_w']).to(dev()).unsqueeze(1) * torch.from_numpy(lat['e_wplus']).to(dev())
    target = synth.tensor(c['name'] + '/target', (c['b'], 3, c['size'], c['size']), dist='uniform').to(dev())
    conds = {}

    def hook(name, mod):
        def fn(m, gin, gout):      # condition number of the scalar noise-weight gradient: sum|terms| / |sum terms|
            nz = getattr(G.noises, f'noise_{name}')
            t = (gout[0].double().sum(1, keepdim=True) * nz.double())
            conds[name] = float(t.abs().sum() / t.sum().abs().clamp_min(1e-300))
        return mod.register_full_backward_hook(fn)

    layers = [G.conv1] + list(G.convs)
    handles = [hook(i, l.noise) for i, l in enumerate(layers)]
    img = G(None, latent_styles=[latent], input_is_latent=True, use_external_input_tensor=True,
            external_input_tensor=tsr, randomize_noise=False)
    loss = L1_Loss(img, target)
    loss.backward()
    for h in handles:
        h.remove()
    names = {('conv1.noise.weight' if i == 0 else f'convs.{i - 1}.noise.weight'): v for i, v in conds.items()}
    return G, img, loss, names


OWN_FLOOR, OWN_FLOOR_NORM = 2e-4, 1e-4


def test_generator_only_backward_own_kernels_tight(golden):
    """All 93 Generator gradients of cfg3 with the Generator's inputs taken from the fixture: held to
    4 x (the reference's own fp32 error) + 2e-4 of the tensor's max, NO kink allowance, no exception list.  The one
    principled widening: a NoiseInjection.weight gradient is ONE scalar, the sum of B*C*H*W signed products that cancel
    to a fraction 1/cond of their magnitude — its achievable relative accuracy is cond x (relative accuracy of the
    upstream gradient), and cond is measured in this test, not assumed."""
    g = golden('e2e_grad')
    G, img, loss, conds = run_generator_only()
    c = cases.E2E_GRAD_CASE
    a = img.detach().cpu().numpy()[..., ::c['stride'], ::c['stride']]
    scale = float(np.abs(g['img/sub64']).max())
    assert np.abs(a - g['img/sub64']).max() / scale <= MARGIN * np.abs(g['img/sub'] - g['img/sub64']).max() / scale + 2e-5
    np.testing.assert_allclose(loss.item(), float(g['loss64']), rtol=2e-5)
    n = 0
    for name, p in G.named_parameters():
        key = f'g/{name}'
        if key + '/s' not in g.files:
            assert p.grad is None
            continue
        fl = OWN_FLOOR
        if name in conds:
            fl = max(OWN_FLOOR, 2e-6 * conds[name])
        k, _ = check_grads(g, 'g', [(name, p)], floor=fl, floor_norm=max(OWN_FLOOR_NORM, fl if name in conds else 0.0))
        n += k
    assert n == len([k for k in g.files if k.startswith('g/') and k.endswith('/n64')]) == 93


@pytest.mark.parametrize('own_wgrad', [False, True])
def test_generator_only_backward_is_bit_reproducible(own_wgrad, monkeypatch):
    """Two runs in one process.  Every gradient this repo's kernels produce is bit-identical (ToRGB included since its
    backward moved from the autograd composite to fmgan_torgb_backward_f32).  One family comes from library kernels that
    are not run-to-run reproducible and agrees to rounding only: in the default configuration the weight gradients of the
    six upsampling convs (MIOpen's stride-2 fp32 wgrad, op/modconv.py HIP_WGRAD = 1: faster there than the own kernel).
    With HIP_WGRAD = 2 they join the bit-identical set and all 93 tensors still meet the tight gate."""
    from op import modconv
    if own_wgrad:
        monkeypatch.setattr(modconv, 'HIP_WGRAD', 2)
    G1, img1, _, conds = run_generator_only()
    g1 = {n: p.grad.clone() for n, p in G1.named_parameters() if p.grad is not None}
    if own_wgrad:
        g = np.load(os.path.join(ROOT, 'tests', 'golden', 'e2e_grad.npz'))
        for name, p in G1.named_parameters():
            if f'g/{name}/s' in g.files:
                fl = max(OWN_FLOOR, 2e-6 * conds[name]) if name in conds else OWN_FLOOR
                check_grads(g, 'g', [(name, p)], floor=fl, floor_norm=max(OWN_FLOOR_NORM, fl if name in conds else 0.0))
    del G1
    G2, img2, _, _ = run_generator_only()
    assert torch.equal(img1, img2)
    miopen = {f'convs.{i}.conv.weight' for i in range(0, 12, 2)}       # transposed convs of Generator(256)
    exact, inexact = 0, []
    for n, p in G2.named_parameters():
        if p.grad is None:
            continue
        if torch.equal(p.grad, g1[n]):
            exact += 1
            continue
        rel = float((p.grad - g1[n]).abs().max() / g1[n].abs().max())
        inexact.append((n, rel))
        assert rel <= 2e-5 if (n in miopen and not own_wgrad) else rel <= 1e-6, (n, rel)
    # Own kernels are bit-identical: the plain convs' weight gradients are this repo's kernel in the product default.  (Two
    # full-suite runs of round 3 failed here on convs.5/7/9/11.conv.weight at 1e-7: an older test had left
    # modconv.HIP_WGRAD at 0 for the rest of the process, so MIOpen's wgrad served them — fixed in that test; the default is
    # asserted here.)
    assert modconv.HIP_WGRAD == (2 if own_wgrad else 1)
    library = sum(1 for n, _ in inexact if n in miopen and not own_wgrad)
    if inexact:
        print('not bit-identical between two runs:', inexact)
    assert len(inexact) == library, inexact
    assert exact + len(inexact) == 93


class FixedProbe:
    """Generator.forward draws the path-length probe with torch.randn_like (stylegan2.py:684); the fixture pinned it."""

    def __init__(self, probe):
        self.probe = probe

    def __enter__(self):
        self.orig = torch.randn_like
        torch.randn_like = lambda t, **kw: self.probe.to(dtype=t.dtype, device=t.device)

    def __exit__(self, *exc):
        torch.randn_like = self.orig


def train_args(**over):
    import train_3_encoder as T
    hp = cases.TRAIN_HP
    return T.default_args(tsr_encode='Photo Image', lr=hp['lr'], r1=hp['r1'], d_reg_every=hp['d_reg_every'],
                          g_reg_every=hp['g_reg_every'], generator_path_reg_weight=hp['path_reg_weight'],
                          path_reg_batch_shrink=hp['path_reg_batch_shrink'], l1_loss_lambda=hp['l1_loss_lambda'],
                          **over)


def train_inputs(c=None):
    c = c or cases.TRAIN_STEP_CASE
    photo = synth.tensor(c['name'] + '/photo', (c['b'], 3, 256, 256), dist='uniform')
    render = synth.tensor(c['name'] + '/render', (c['b'], 3, 256, 256), dist='uniform')
    ref = synth.tensor(c['name'] + '/ref', (c['b'], 3, c['size'], c['size']), dist='uniform')
    probe = synth.tensor(c['name'] + '/probe', (len(c['ppl_idx']), 3, c['size'], c['size']))
    return tuple(t.to(dev()) for t in (photo, render, ref, probe))


def run_phase(phase, nets, args, photo, render, ref, probe, ppl_idx):
    """One of the four gradient computations, no optimiser step; returns the loss dict."""
    import train_3_encoder as T
    G = PinNoise(nets['g'].module if hasattr(nets['g'], 'module') else nets['g'], nets['g'])
    ld = {}
    if phase == 'd':
        T.D_Loss_BackProp(G, nets['e_tsr'], nets['e_w'], nets['e_wp'], nets['d'], photo, render, ref, args, ld, None)
    elif phase == 'r1':
        ld['r1'] = T.D_Reg_BackProp(ref, nets['d'], args, None)
    elif phase == 'g':
        T.G_Loss_BackProp(G, nets['e_tsr'], nets['e_w'], nets['e_wp'], nets['d'], photo, render, ref, args, ld, None)
    else:
        with FixedProbe(probe):
            ld['ppl'], ld['lengths'], _ = T.G_Reg_BackProp(G, nets['e_tsr'], nets['e_w'], nets['e_wp'], photo, render,
                                                           args, 0, None, choice=ppl_idx)
    return ld


@pytest.mark.parametrize('phase', ['d', 'r1', 'g', 'ppl'])
@pytest.mark.parametrize('case', ['train_step', 'train_step_1024'])
def test_train_step_phase_golden(case, phase, golden):
    """D_Loss_BackProp / D_Reg_BackProp / G_Loss_BackProp / G_Reg_BackProp vs the reference modules and the reference's
    loss functions at the same weights (train_3_encoder.py:448-596): Generator(64) + Discriminator(64) at B=4, and
    BASELINE config 5's networks — Generator(1024) + Discriminator(1024), 18 styles — at B=2."""
    g = golden(case)
    c = cases.TRAIN_STEP_CASE if case == 'train_step' else cases.TRAIN_STEP_1024_CASE
    nets = build_nets(c['size'], with_d=True, n_mlp=2)
    photo, render, ref, probe = train_inputs(c)
    ld = run_phase(phase, nets, train_args(), photo, render, ref, probe, c['ppl_idx'])
    if phase == 'd':
        np.testing.assert_allclose(ld['d'].item(), float(g['d/loss64']), rtol=1e-4)
        np.testing.assert_allclose(ld['ref_score'].item(), float(g['d/ref_score64']), rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(ld['out_score'].item(), float(g['d/out_score64']), rtol=1e-4, atol=1e-5)
        n, _ = check_grads(g, 'd/d', nets['d'].named_parameters())
        assert n == len(list(nets['d'].parameters()))
        assert all(p.grad is None for p in nets['g'].parameters())       # producers frozen
    elif phase == 'r1':
        np.testing.assert_allclose(ld['r1'].item(), float(g['r1/loss64']), rtol=1e-3)
        n, _ = check_grads(g, 'r1/d', nets['d'].named_parameters())
        assert n == len(list(nets['d'].parameters()))
    elif phase == 'g':
        np.testing.assert_allclose(ld['g'].item(), float(g['g/loss64']), rtol=1e-4)
        np.testing.assert_allclose(ld['l1'].item(), float(g['g/l164']), rtol=1e-4)
        kinks = []
        for k in ('g', 'e_tsr', 'e_w', 'e_wp'):
            n, _ = check_grads(g, 'g/' + k, nets[k].named_parameters(), kinks=kinks)
            assert n > 20
        confirm_kinks(kinks)
        assert all(p.grad is None for p in nets['d'].parameters())       # D frozen
    else:
        np.testing.assert_allclose(ld['lengths'].detach().cpu().numpy(), g['ppl/lengths64'], rtol=1e-3)
        np.testing.assert_allclose(ld['ppl'].item(), float(g['ppl/loss64']), rtol=2e-3)
        kinks = []
        for k in ('g', 'e_tsr', 'e_w', 'e_wp'):
            # second-order gradients (double backward through every op): measured worst case 1.42e-3 on
            # convs.7.conv.modulation.weight (reference fp32: 4e-5) -> floor 3e-3 for G, the MIOpen floor for encoders
            # (norms of the second-order G gradients: 5.5e-4 measured at 1024^2 in one of two runs -> 1e-3)
            n, _ = check_grads(g, 'ppl/' + k, nets[k].named_parameters(), floor=3e-3 if k == 'g' else None, kinks=kinks,
                               floor_norm=1e-3 if k == 'g' else None)
            assert n > 20
        confirm_kinks(kinks)


def test_trainer_iteration_runs_and_updates_everything():
    """One Trainer.step (D, R1, G, path length, EMA — train_3_encoder.py:801-822): finite losses, every trained
    network and g_ema moved, and the next no_grad g_ema forward sees the EMA update."""
    import train_3_encoder as T
    c = cases.TRAIN_STEP_CASE
    nets = build_nets(c['size'], with_d=True, n_mlp=2)
    photo, render, ref, _ = train_inputs()
    lat = synth.tensor('tr/lat', (1, nets['g'].n_latent, 512)).to(dev())
    tsr = synth.tensor('tr/tsr', (1, 512, 4, 4)).to(dev())
    tr = T.Trainer(dict(G=nets['g'], E_Tsr=nets['e_tsr'], E_W=nets['e_w'], E_W_Plus=nets['e_wp'], D=nets['d']),
                   train_args(), dev())

    def ema_image():
        with torch.no_grad():
            return tr.g_ema(None, latent_styles=[lat], input_is_latent=True, use_external_input_tensor=True,
                            external_input_tensor=tsr, randomize_noise=False).clone()

    before = {k: [p.detach().clone() for p in m.parameters()] for k, m in nets.items()}
    img0 = ema_image()
    ld = tr.step(photo, render, ref)
    for k in ('d', 'r1', 'g', 'l1', 'g_reg'):
        assert torch.isfinite(ld[k]).all(), k
    for k, m in nets.items():
        moved = sum(int(not torch.equal(a, b)) for a, b in zip(before[k], m.parameters()))
        assert moved > 0.5 * len(before[k]), f'{k}: {moved} of {len(before[k])} tensors updated'
    assert not torch.equal(ema_image(), img0)
    assert tr.iter_idx == 1 and float(tr.mean_path_length) > 0
    ld = tr.step(photo, render, ref)           # iteration 1: no R1, no path-length step
    assert torch.isfinite(ld['d']).all() and tr.iter_idx == 2


# ----------------------------------------------------------------------------------------------- world size 2
_WORKER = r'''
import os, sys, json
root = sys.argv[1]; mode = sys.argv[2]; out = sys.argv[3]
for p in (root, os.path.join(root, '3d-fm-gan_amd'), os.path.join(root, 'tests')):
    sys.path.insert(0, p)
import numpy as np, torch
import cases, synth
from Miscellaneous import distributed as D
import test_hip_train as H
rank, world, device = D.init_distributed(backend='gloo')
assert world == 2
c = cases.TRAIN_STEP_CASE
nets = H.build_nets(c['size'], with_d=True, n_mlp=2)
for m in nets.values():
    m.requires_grad_(True)
ddp = mode == 'ddp'
wrapped = {k: D.data_parallel(m, device, overlap=ddp) for k, m in nets.items()}
args = H.train_args(grad_sync='ddp' if ddp else 'flat', grad_algorithm=mode if not ddp else 'all_reduce')
photo, render, ref, probe = H.train_inputs()
lo, hi = D.shard_range(c['b'])
res = {}
for phase in ('d', 'r1', 'g', 'ppl'):
    for m in nets.values():
        m.zero_grad(set_to_none=True)
    # path length: local sample 0 of each rank (global samples 0 and 2, the fixture's choice)
    H.run_phase(phase, wrapped, args, photo[lo:hi], render[lo:hi], ref[lo:hi], probe[rank:rank + 1], [0])
    for k, m in nets.items():
        for name, p in m.named_parameters():
            if p.grad is not None:
                s, n = cases.grad_sample(p.grad)
                res[f'{phase}/{k}/{name}/s'] = s; res[f'{phase}/{k}/{name}/n'] = np.float64(n)
np.savez(out + f'.{rank}.npz', **res)
D.synchronize()
torch.distributed.destroy_process_group()
'''


@pytest.mark.timeout(1500)
@pytest.mark.parametrize('mode', ['ddp', 'direct', 'reduce_scatter'])
def test_world2_phases_equal_single_process(mode, tmp_path):
    """Two ranks (two processes sharing this GPU, gloo), each on half of the batch: the rank-averaged gradients of
    every phase equal the single-process gradients on the whole batch (the reference's DataParallel semantics).
    D's minibatch-stddev layer groups samples i, i+B/4, ... (stylegan2.py:805-813): the single-process batch is laid
    out so that its groups are the ranks' groups (DataParallel computes the statistic per replica too)."""
    import subprocess
    c = cases.TRAIN_STEP_CASE
    script = tmp_path / 'worker.py'
    script.write_text(_WORKER)
    out = str(tmp_path / 'res')
    port = 29500 + (os.getpid() % 400) + {'ddp': 0, 'direct': 1, 'reduce_scatter': 2}[mode]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr',
           '127.0.0.1', '--master-port', str(port), str(script), ROOT, mode, out]
    proc = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1400)
    assert proc.returncode == 0, proc.stdout[-3000:] + proc.stderr[-3000:]
    # round-1 failure: NHWC-laid-out pSp conv weights whose gradients did not match DDP's bucket views.  Tensors with
    # a size-1 dimension have ambiguous strides (same bytes, different stride tuples): for those the warning is noise.
    import re
    for m in re.finditer(r'grad\.sizes\(\) = \[([\d, ]+)\], strides\(\) = \[([\d, ]+)\]\s*bucket_view\.sizes\(\) = \[[\d, ]+\], '
                         r'strides\(\) = \[([\d, ]+)\]', proc.stderr):
        sizes, gs, bs = ([int(v) for v in m.group(i).split(',')] for i in (1, 2, 3))
        assert all(a == b for n_, a, b in zip(sizes, gs, bs) if n_ > 1), f'DDP bucket view mismatch: {m.group(0)}'
    r0, r1 = np.load(out + '.0.npz'), np.load(out + '.1.npz')
    # single process on the whole batch; minibatch-stddev groups: rank r's samples at positions r, r+2 (b = 4 -> group
    # size 4 needs b = 8; with b = 4 per process and 2 per rank the group is min(batch, 4) = the whole local batch)
    nets = build_nets(c['size'], with_d=True, n_mlp=2)
    photo, render, ref, probe = train_inputs()
    args = train_args()
    checked = 0
    kinks = []     # elements that sit on a kink of the loss (see below)
    for phase in ('d', 'r1', 'g', 'ppl'):
        for m in nets.values():
            m.zero_grad(set_to_none=True)
        single_nets = dict(nets)
        if phase in ('d', 'r1', 'g'):
            single_nets['d'] = _PerHalfD(nets['d'])
        run_phase(phase, single_nets, args, photo, render, ref, probe, c['ppl_idx'])
        for k, m in nets.items():
            grads = [(name, p.grad) for name, p in m.named_parameters()]
            # fp32 noise floor of a network's gradients: 1e-5 of its largest gradient entry (several tensors of the
            # second-order phases are pure rounding noise around zero, e.g. the R1 gradient of a bias)
            net_scale = max([float(gr.abs().max()) for _, gr in grads if gr is not None] + [0.0])
            for name, gr in grads:
                key = f'{phase}/{k}/{name}'
                if gr is None:
                    assert key + '/s' not in r0.files
                    continue
                s, n = cases.grad_sample(gr)
                np.testing.assert_array_equal(r0[key + '/s'], r1[key + '/s'])       # ranks agree bit for bit
                scale = float(np.abs(s).max())
                diff = np.abs(r0[key + '/s'] - s)
                # encoder tensors: MIOpen's fp32 wgrad / double-backward kernels are not reproducible run to run and put
                # isolated elements up to 6e-3 of the tensor's max apart (same measurement as FLOOR_MIOPEN)
                bad = diff > (1e-2 if k.startswith('e_') else 2e-3) * scale + 1e-5 * net_scale
                # The losses have kinks (|x| of the L1 term, the leaky ReLUs): a pixel that sits on one flips the side it
                # takes with the last bit of the forward, and the gradient of a few weights jumps by a discrete amount —
                # the reference's own fp32 and fp64 runs disagree on exactly such elements (g/e_wp/styles.3.convs.0.weight,
                # element 2 of the fixture sample: -8.03e-3 vs -8.60e-3).  They are tolerated as isolated elements
                # (bounded in number and size below); everything else must agree.
                if bad.any():       # one flipped unit can move a whole row of a small weight gradient: count tensors
                    assert diff.max() <= 0.1 * scale, (key, float(diff.max()), scale)
                    kinks.append((key, int(bad.sum()), float(diff.max() / scale)))
                if not bad.any():
                    assert abs(float(r0[key + '/n']) - n) <= 5e-3 * n + 1e-5 * net_scale * np.sqrt(gr.numel()), key
                checked += 1
    assert checked > 600
    assert len(kinks) <= 6, kinks          # tensors, of ~650 compared


class _PerHalfD(torch.nn.Module):
    """D applied to each rank's half separately: the minibatch-stddev statistic (stylegan2.py:805-813) is per replica in
    the reference's DataParallel and per rank here; everything else in D is per sample."""

    def __init__(self, d):
        super().__init__()
        self.module = d

    def forward(self, x):
        h = x.shape[0] // 2
        return torch.cat([self.module(x[:h]), self.module(x[h:])], 0)


_RCCL_WORKER = r'''
import os, sys
root = sys.argv[1]
for p in (root, os.path.join(root, '3d-fm-gan_amd'), os.path.join(root, 'tests')):
    sys.path.insert(0, p)
import numpy as np, torch
import torch.distributed as dist
import cases, synth
from Miscellaneous import distributed as D
import test_hip_train as H
import train_3_encoder as T
rank, world, device = D.init_distributed(backend='nccl', force=True)
assert (rank, world) == (0, 1) and dist.get_backend() == 'nccl' and device.type == 'cuda'
# every collective the training path and bench.py use, on the RCCL communicator
x = torch.arange(1024, dtype=torch.float32, device=device)
for alg in D.GRAD_ALGORITHMS:
    f = x.clone(); D._reduce_flat(f, 1, alg); assert torch.equal(f, x), alg
m = torch.tensor([3, 0, 1], dtype=torch.int32, device=device); dist.all_reduce(m, op=dist.ReduceOp.MAX)
assert m.tolist() == [3, 0, 1]
t = torch.tensor([1.5], dtype=torch.float64, device=device); dist.all_reduce(t, op=dist.ReduceOp.MAX); assert t.item() == 1.5
v = torch.tensor([2.0], device=device, requires_grad=True)
g, = torch.autograd.grad(T._global_mean(v * 3.0), v); assert g.item() == 3.0
from torch.distributed.nn import functional as dist_fn      # what _global_mean uses when world > 1
y = dist_fn.all_reduce(v * 3.0); g, = torch.autograd.grad(y.sum(), v); assert y.item() == 6.0 and g.item() == 3.0
assert D.reduce_sum(x).equal(x) and D.all_gather({'a': 1}) == [{'a': 1}]
dist.barrier()
# the training iteration with its networks inside DistributedDataParallel over RCCL (bucket views, NHWC pSp weights)
c = cases.TRAIN_STEP_CASE
nets = H.build_nets(c['size'], with_d=True, n_mlp=2)
for n_ in nets.values():
    n_.requires_grad_(True)
wrapped = {k: D.data_parallel(n_, device, single_rank_ddp=True) for k, n_ in nets.items()}
assert all(isinstance(w, torch.nn.parallel.DistributedDataParallel) for w in wrapped.values())
photo, render, ref, probe = H.train_inputs()
res = {}
for phase in ('d', 'r1', 'g', 'ppl'):
    for n_ in nets.values():
        n_.zero_grad(set_to_none=True)
    H.run_phase(phase, wrapped, H.train_args(grad_sync='ddp'), photo, render, ref, probe, c['ppl_idx'])
    for k, n_ in nets.items():
        for name, p in n_.named_parameters():
            if p.grad is not None:
                assert torch.isfinite(p.grad).all(), (phase, k, name)
                res[f'{phase}/{k}/{name}'] = cases.grad_sample(p.grad)[0]
np.savez(sys.argv[2], **res)
D.synchronize()
dist.destroy_process_group()
print('rccl-ok')
'''


@pytest.mark.timeout(900)
def test_rccl_group_of_one(tmp_path):
    """`backend="nccl"` (= RCCL) has to come up on this box: a process group of ONE rank is enough to load RCCL, build a
    communicator on the GPU, run every collective the path uses (all_reduce SUM/MAX in f32/f64/i32, reduce_scatter +
    all_gather, all_to_all, the differentiable all-reduce, barrier, object gather) and drive the four training phases
    through DistributedDataParallel with gradient-as-bucket-view on the NHWC pSp weights.  The gradients must equal the
    no-process-group run of the same phases (a group of one averages nothing).  Multi-rank arithmetic is covered by the
    gloo tests; what only RCCL can show on one GPU is that this branch works at all."""
    import subprocess
    script = tmp_path / 'rccl_worker.py'
    script.write_text(_RCCL_WORKER)
    out = str(tmp_path / 'rccl.npz')
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0', MASTER_ADDR='127.0.0.1',
               MASTER_PORT=str(29900 + os.getpid() % 90), RANK='0', WORLD_SIZE='1', LOCAL_RANK='0')
    proc = subprocess.run([sys.executable, str(script), ROOT, out], env=env, capture_output=True, text=True, timeout=800)
    assert proc.returncode == 0 and 'rccl-ok' in proc.stdout, proc.stdout[-2000:] + proc.stderr[-4000:]
    r = np.load(out)
    c = cases.TRAIN_STEP_CASE
    nets = build_nets(c['size'], with_d=True, n_mlp=2)
    photo, render, ref, probe = train_inputs()
    checked = 0
    kink_tensors = []
    for phase in ('d', 'r1', 'g', 'ppl'):
        for m in nets.values():
            m.zero_grad(set_to_none=True)
        run_phase(phase, nets, train_args(), photo, render, ref, probe, c['ppl_idx'])
        for k, m in nets.items():
            grads = [(n, p.grad) for n, p in m.named_parameters() if p.grad is not None]
            net_scale = max([float(g.abs().max()) for _, g in grads] + [0.0])
            for name, gr in grads:
                s, _ = cases.grad_sample(gr)
                d = np.abs(r[f'{phase}/{k}/{name}'] - s)
                tol = (1e-2 if k.startswith('e_') else 2e-3) * float(np.abs(s).max()) + 1e-5 * net_scale
                if (d > tol).any():
                    assert d.max() <= 0.1 * float(np.abs(s).max()) + 1e-5 * net_scale, (phase, k, name)
                    kink_tensors.append((phase, k, name, float(d.max())))
                checked += 1
    assert checked > 600
    assert len(kink_tensors) <= 6, kink_tensors
