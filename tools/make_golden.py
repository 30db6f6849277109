#!/usr/bin/env python3
"""Generate tests/golden/*.npz by importing the reference (adobe/3D-FM-GAN) on CPU.

Run in the build container only (the reference does not exist on the GPU box):
    PYTHONDONTWRITEBYTECODE=1 python tools/make_golden.py
Recipe: stub torch.utils.cpp_extension.load before `import stylegan2` (the reference
JIT-compiles CUDA at import, op/upfirdn2d.py:19-25), stub torchvision (only Convert_Tensor_To_Image
uses it).  CPU tensors route to the reference's own pure-PyTorch paths (op/upfirdn2d.py:155-163,
op/fused_act.py:114-128), which the reference treats as interchangeable with its CUDA kernels.

Fixtures hold OUTPUTS (and name/shape manifests) only; inputs and weights are regenerated from
tests/synth.py on both sides.  No reference source text is stored.
"""
import json
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get('FMGAN_REFERENCE', '/root/reference')
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import synth  # noqa: E402
import cases  # noqa: E402

import torch.utils.cpp_extension as _ce  # noqa: E402
_ce.load = lambda *a, **k: None
sys.path.insert(0, REF)
_tv = types.ModuleType('torchvision'); _tvu = types.ModuleType('torchvision.utils'); _tv.utils = _tvu
sys.modules['torchvision'] = _tv; sys.modules['torchvision.utils'] = _tvu

import stylegan2  # noqa: E402
import resnet_encoder  # noqa: E402
from psp_encoder_model.encoders import psp_encoders  # noqa: E402
import Util.network_util as network_util  # noqa: E402
from op import upfirdn2d as ref_upfirdn2d, fused_leaky_relu as ref_fused_leaky_relu  # noqa: E402

OUT = os.path.join(ROOT, 'tests', 'golden')
torch.set_grad_enabled(True)


def npy(t):
    return t.detach().cpu().numpy()


def gen_upfirdn2d():
    out = {}
    for c in cases.UPFIRDN2D_CASES:
        x = synth.tensor(c['name'] + '/x', c['shape']).requires_grad_(True)
        k = cases.make_fir(c['kernel'])
        y = ref_upfirdn2d(x, k, up=c['up'], down=c['down'], pad=tuple(c['pad']))
        out[c['name'] + '/out'] = npy(y)
        if c.get('grad'):
            go = synth.tensor(c['name'] + '/go', y.shape).requires_grad_(True)
            gi, = torch.autograd.grad(y, x, go, create_graph=True)
            out[c['name'] + '/grad_input'] = npy(gi)
            ggi = synth.tensor(c['name'] + '/ggi', x.shape)
            gg, = torch.autograd.grad(gi, go, ggi)
            out[c['name'] + '/gradgrad_out'] = npy(gg)
    np.savez_compressed(os.path.join(OUT, 'upfirdn2d.npz'), **out)
    print('upfirdn2d', len(out))


def gen_fused_act():
    out = {}
    for c in cases.FUSED_ACT_CASES:
        x, b = cases.fused_act_inputs(c)
        x.requires_grad_(True)
        if b is not None:
            b.requires_grad_(True)
        y = ref_fused_leaky_relu(x, b)
        out[c['name'] + '/out'] = npy(y)
        go = synth.tensor(c['name'] + '/go', y.shape).requires_grad_(True)
        ins = [x] + ([b] if b is not None else [])
        grads = torch.autograd.grad(y, ins, go, create_graph=True)
        out[c['name'] + '/grad_input'] = npy(grads[0])
        if b is not None:
            out[c['name'] + '/grad_bias'] = npy(grads[1])
        # second order: d<grad_input, ggi>/d(grad_output)
        ggi = synth.tensor(c['name'] + '/ggi', x.shape)
        gg, = torch.autograd.grad(grads[0], go, ggi)
        out[c['name'] + '/gradgrad_out'] = npy(gg)
    np.savez_compressed(os.path.join(OUT, 'fused_act.npz'), **out)
    print('fused_act', len(out))


def gen_modules():
    out = {}
    man = {}
    for c in cases.MODCONV_CASES:
        m = stylegan2.ModulatedConv2d(c['cin'], c['cout'], c['k'], 512, demodulate=c['demod'], upsample=c['up'])
        sd = synth.state_dict('generator', m.state_dict(), seed=1)
        m.load_state_dict(sd)
        man[c['name']] = synth.manifest(sd)
        x = synth.tensor(c['name'] + '/x', (c['b'], c['cin'], c['h'], c['h']))
        w = synth.tensor(c['name'] + '/w', (c['b'], 512))
        out[c['name'] + '/out'] = npy(m(x, w))
    for c in cases.STYLEDCONV_CASES:
        m = stylegan2.StyledConv(c['cin'], c['cout'], 3, 512, upsample=c['up'])
        sd = synth.state_dict('generator', m.state_dict(), seed=2)
        m.load_state_dict(sd)
        man[c['name']] = synth.manifest(sd)
        oh = c['h'] * 2 if c['up'] else c['h']
        x = synth.tensor(c['name'] + '/x', (c['b'], c['cin'], c['h'], c['h']))
        w = synth.tensor(c['name'] + '/w', (c['b'], 512))
        nz = synth.tensor(c['name'] + '/noise', (c['nb'], 1, oh, oh))
        out[c['name'] + '/out'] = npy(m(x, w, noise=nz))
    for c in cases.TORGB_CASES:
        m = stylegan2.ToRGB(c['cin'], 512, upsample=c['skip'])
        sd = synth.state_dict('generator', m.state_dict(), seed=3)
        m.load_state_dict(sd)
        man[c['name']] = synth.manifest(sd)
        x = synth.tensor(c['name'] + '/x', (c['b'], c['cin'], c['h'], c['h']))
        w = synth.tensor(c['name'] + '/w', (c['b'], 512))
        skip = synth.tensor(c['name'] + '/skip', (c['b'], 3, c['h'] // 2, c['h'] // 2)) if c['skip'] else None
        out[c['name'] + '/out'] = npy(m(x, w, skip))
    np.savez_compressed(os.path.join(OUT, 'modules.npz'), **out)
    json.dump(man, open(os.path.join(OUT, 'modules_manifest.json'), 'w'), indent=0, sort_keys=True)
    print('modules', len(out))


def subsample(img, stride):
    return npy(img)[..., ::stride, ::stride].copy()


def stats(img):
    a = npy(img).astype(np.float64)
    return np.array([a.mean(), np.abs(a).mean(), a.min(), a.max(), (a * a).sum()])


def gen_generator():
    out = {}
    man = {}
    with torch.no_grad():
        for c in cases.GENERATOR_CASES:
            g = stylegan2.Generator(c['size'], 512, c['n_mlp'], generator_net_shape=c['shape'])
            sd = synth.state_dict('generator', g.state_dict(), seed=4)
            g.load_state_dict(sd)
            g.eval()
            man[c['name']] = synth.manifest(sd)
            cin0 = c['shape'][0] if c['shape'] else 512
            if c['mode'] == 'latent':
                lat = synth.tensor(c['name'] + '/latent', (c['b'], g.n_latent, 512))
                tsr = synth.tensor(c['name'] + '/tsr', (c['b'], cin0, 4, 4))
                img = g(None, latent_styles=[lat], input_is_latent=True, use_external_input_tensor=True,
                        external_input_tensor=tsr, randomize_noise=False)
            else:   # mapping network + ConstantInput
                z = synth.tensor(c['name'] + '/z', (c['b'], 512))
                img = g([z], randomize_noise=False)
            assert tuple(img.shape) == (c['b'], 3, c['size'], c['size'])
            out[c['name'] + '/sub'] = subsample(img, c['stride'])
            out[c['name'] + '/stats'] = stats(img)
            print(' ', c['name'], tuple(img.shape), float(img.abs().max()))
    np.savez_compressed(os.path.join(OUT, 'generator.npz'), **out)
    json.dump(man, open(os.path.join(OUT, 'generator_manifest.json'), 'w'), indent=0, sort_keys=True)
    print('generator', len(out))


class _GWrap:
    """Forward_Inference_3_Encoder touches g_ema.module (Util/network_util.py:317-318) and never
    passes noise: expose .module and pin randomize_noise=False."""

    def __init__(self, g):
        self.module = g

    def __call__(self, **kw):
        return self.module(randomize_noise=False, **kw)


def build_encoders(n_styles):
    e_tsr = resnet_encoder.resnet18(tensor_encoding=True, tensor_transform=False)
    e_w = resnet_encoder.resnet18(tensor_encoding=False, tensor_transform=False)
    opts = types.SimpleNamespace(input_nc=3, n_styles=n_styles)
    e_wp = psp_encoders.GradualStyleEncoder(18, 'ir_se', opts)
    for kind, m, seed in (('resnet', e_tsr, 5), ('resnet', e_w, 6), ('psp', e_wp, 7)):
        m.load_state_dict(synth.state_dict(kind, m.state_dict(), seed=seed))
        m.eval()
    return e_tsr, e_w, e_wp


def gen_encoders_e2e():
    out = {}
    man = {}
    with torch.no_grad():
        for c in cases.E2E_CASES:
            size = c['size']
            n_latent = int(np.log2(size)) * 2 - 2
            e_tsr, e_w, e_wp = build_encoders(n_latent)
            man[f'resnet'] = synth.manifest(e_tsr.state_dict())
            man[f'psp{n_latent}'] = synth.manifest(e_wp.state_dict())
            p = synth.tensor(c['name'] + '/photo', (c['b'], 3, 256, 256), dist='uniform')
            r = synth.tensor(c['name'] + '/render', (c['b'], 3, 256, 256), dist='uniform')
            out[c['name'] + '/e_tsr'] = npy(e_tsr(p))
            out[c['name'] + '/e_w'] = npy(e_w(r))
            out[c['name'] + '/e_wplus'] = npy(e_wp(p))
            g = stylegan2.Generator(size, 512, 8)
            g.load_state_dict(synth.state_dict('generator', g.state_dict(), seed=4))
            g.eval()
            wrap = _GWrap(g)
            img = network_util.Forward_Inference_3_Encoder(p, r, e_tsr, e_w, e_wp, wrap,
                                                           tsr_encode=c['tsr_encode'],
                                                           sliced_layer=c['sliced_layer'],
                                                           use_tanh=c['use_tanh'])
            out[c['name'] + '/sub'] = subsample(img, c['stride'])
            out[c['name'] + '/stats'] = stats(img)
            print(' ', c['name'], tuple(img.shape), float(img.abs().max()))
    np.savez_compressed(os.path.join(OUT, 'e2e.npz'), **out)
    json.dump(man, open(os.path.join(OUT, 'encoders_manifest.json'), 'w'), indent=0, sort_keys=True)
    print('e2e', len(out))


def gen_discriminator():
    out = {}
    man = {}
    with torch.no_grad():
        for c in cases.DISCRIMINATOR_CASES:
            d = stylegan2.Discriminator(c['size'])
            sd = synth.state_dict('discriminator', d.state_dict(), seed=8)
            d.load_state_dict(sd)
            man[c['name']] = synth.manifest(sd)
            x = synth.tensor(c['name'] + '/x', (c['b'], 3, c['size'], c['size']), dist='uniform')
            out[c['name'] + '/out'] = npy(d(x))
    np.savez_compressed(os.path.join(OUT, 'discriminator.npz'), **out)
    json.dump(man, open(os.path.join(OUT, 'discriminator_manifest.json'), 'w'), indent=0, sort_keys=True)
    print('discriminator', len(out))


def gen_image_io():
    """tensor2im of the reference (Evaluation/visual_eval.py:24-38; numpy only — `imageio` is stubbed, it is used by
    the GIF writers, not by tensor2im).  transforms.ToTensor()/Normalize live in torchvision, which is not installed:
    the input-side converter is pinned by its documented formula only."""
    sys.modules.setdefault('imageio', types.ModuleType('imageio'))
    from Evaluation import visual_eval
    out = {}
    for c in cases.TENSOR2IM_CASES:
        x = cases.tensor2im_input(c)
        out[c['name'] + '/im'] = visual_eval.tensor2im(x)
    np.savez_compressed(os.path.join(OUT, 'image_io.npz'), **out)
    print('image_io', len(out))


if __name__ == '__main__':
    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(0)
    which = sys.argv[1:] or ['upfirdn2d', 'fused_act', 'modules', 'generator', 'e2e', 'discriminator', 'image_io']
    for w in which:
        {'upfirdn2d': gen_upfirdn2d, 'fused_act': gen_fused_act, 'modules': gen_modules,
         'generator': gen_generator, 'e2e': gen_encoders_e2e, 'discriminator': gen_discriminator,
         'image_io': gen_image_io}[w]()
