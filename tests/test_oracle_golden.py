"""Pins the CPU oracle (oracle/fmgan_oracle.c and oracle/torch_oracle.py) to golden vectors produced by the
reference itself (tools/make_golden.py).  CPU only; the oracle is then the checker for the HIP path."""
import numpy as np
import pytest
import torch

import cases
import synth
from oracle import c_oracle, torch_oracle as T

OP_TOL = dict(atol=1e-5, rtol=1e-5)


def _c_upfirdn2d(x, k, up, down, pad):
    n, c, h, w = x.shape
    y = c_oracle.upfirdn2d(x.reshape(n * c, h, w, 1).numpy(), k.numpy(), (up, up), (down, down), (pad[0], pad[1], pad[0], pad[1]))
    return y.reshape(n, c, y.shape[1], y.shape[2])


@pytest.mark.parametrize('c', cases.UPFIRDN2D_CASES, ids=lambda c: c['name'])
def test_upfirdn2d_oracles(c, golden):
    g = golden('upfirdn2d')
    x = synth.tensor(c['name'] + '/x', c['shape'])
    k = cases.make_fir(c['kernel'])
    ref = g[c['name'] + '/out']
    np.testing.assert_allclose(_c_upfirdn2d(x, k, c['up'], c['down'], c['pad']), ref, **OP_TOL)
    np.testing.assert_allclose(T.upfirdn2d(x, k, c['up'], c['down'], tuple(c['pad'])).numpy(), ref, **OP_TOL)
    # float64 C oracle agrees with the float32 golden to float32 rounding
    y64 = _c_upfirdn2d(x.double(), k.double(), c['up'], c['down'], c['pad'])
    np.testing.assert_allclose(y64, ref, atol=2e-5, rtol=2e-5)


@pytest.mark.parametrize('c', cases.FUSED_ACT_CASES, ids=lambda c: c['name'])
def test_fused_act_oracles(c, golden):
    g = golden('fused_act')
    x, b = cases.fused_act_inputs(c)
    ref = g[c['name'] + '/out']
    y = c_oracle.fused_bias_act(x.numpy(), None if b is None else b.numpy(), None, 3, 0, 0.2, 2 ** 0.5)
    np.testing.assert_allclose(y, ref, atol=1e-6, rtol=1e-6)
    np.testing.assert_allclose(T.fused_leaky_relu(x, b).numpy(), ref, atol=1e-6, rtol=1e-6)
    # backward = the same kernel with grad=1 and ref=out (op/fused_act.py:38-40)
    go = synth.tensor(c['name'] + '/go', x.shape)
    gi = c_oracle.fused_bias_act(go.numpy(), None, ref, 3, 1, 0.2, 2 ** 0.5)
    np.testing.assert_allclose(gi, g[c['name'] + '/grad_input'], atol=1e-6, rtol=1e-6)
    if b is not None:
        dims = tuple([0] + list(range(2, x.ndim)))
        np.testing.assert_allclose(gi.sum(axis=dims), g[c['name'] + '/grad_bias'], atol=2e-5, rtol=2e-5)
    ggi = synth.tensor(c['name'] + '/ggi', x.shape)
    ggo = c_oracle.fused_bias_act(ggi.numpy(), None, ref, 3, 1, 0.2, 2 ** 0.5)
    np.testing.assert_allclose(ggo, g[c['name'] + '/gradgrad_out'], atol=1e-6, rtol=1e-6)


def _module_sd(golden, name, seed):
    man = golden.manifest('modules')[name]
    shapes = {k: torch.empty(v) for k, v in man.items()}
    for k in shapes:
        if k.endswith('kernel'):
            taps = T.make_kernel([1, 3, 3, 1])
            shapes[k] = taps * 4
    return synth.state_dict('generator', shapes, seed=seed)


@pytest.mark.parametrize('c', cases.MODCONV_CASES, ids=lambda c: c['name'])
def test_modconv_oracles(c, golden):
    g = golden('modules')
    sd = _module_sd(golden, c['name'], 1)
    x = synth.tensor(c['name'] + '/x', (c['b'], c['cin'], c['h'], c['h']))
    w = synth.tensor(c['name'] + '/w', (c['b'], 512))
    ref = g[c['name'] + '/out']
    tol = dict(atol=2e-5 * max(1.0, np.abs(ref).max()), rtol=1e-5)
    y = T.modulated_conv2d(x, w, sd['weight'], sd['modulation.weight'], sd['modulation.bias'], c['demod'], c['up'],
                           [1, 3, 3, 1])
    np.testing.assert_allclose(y.numpy(), ref, **tol)
    # C oracle: style = modulation(w); transposed conv then the blur as a separate upfirdn2d
    style = T.equal_linear(w, sd['modulation.weight'], sd['modulation.bias'])
    yc = c_oracle.modulated_conv2d(x.numpy(), sd['weight'][0].numpy(), style.numpy(), mode=1 if c['up'] else 0,
                                   demodulate=c['demod'])
    if c['up']:
        yc = _c_upfirdn2d(torch.from_numpy(yc), T.make_kernel([1, 3, 3, 1]) * 4, 1, 1, (1, 1))
    np.testing.assert_allclose(yc, ref, **tol)


@pytest.mark.parametrize('c', cases.STYLEDCONV_CASES, ids=lambda c: c['name'])
def test_styledconv_oracle(c, golden):
    g = golden('modules')
    sd = {'m.' + k: v for k, v in _module_sd(golden, c['name'], 2).items()}
    oh = c['h'] * 2 if c['up'] else c['h']
    x = synth.tensor(c['name'] + '/x', (c['b'], c['cin'], c['h'], c['h']))
    w = synth.tensor(c['name'] + '/w', (c['b'], 512))
    nz = synth.tensor(c['name'] + '/noise', (c['nb'], 1, oh, oh))
    ref = g[c['name'] + '/out']
    y = T.styled_conv(sd, 'm', x, w, nz, c['up'])
    np.testing.assert_allclose(y.numpy(), ref, atol=2e-5 * max(1.0, np.abs(ref).max()), rtol=1e-5)


@pytest.mark.parametrize('c', cases.TORGB_CASES, ids=lambda c: c['name'])
def test_torgb_oracles(c, golden):
    g = golden('modules')
    sd = {'m.' + k: v for k, v in _module_sd(golden, c['name'], 3).items()}
    x = synth.tensor(c['name'] + '/x', (c['b'], c['cin'], c['h'], c['h']))
    w = synth.tensor(c['name'] + '/w', (c['b'], 512))
    skip = synth.tensor(c['name'] + '/skip', (c['b'], 3, c['h'] // 2, c['h'] // 2)) if c['skip'] else None
    ref = g[c['name'] + '/out']
    tol = dict(atol=2e-5 * max(1.0, np.abs(ref).max()), rtol=1e-5)
    np.testing.assert_allclose(T.to_rgb(sd, 'm', x, w, skip).numpy(), ref, **tol)
    style = T.equal_linear(w, sd['m.conv.modulation.weight'], sd['m.conv.modulation.bias'])
    up = None
    if skip is not None:
        up = _c_upfirdn2d(skip, T.make_kernel([1, 3, 3, 1]) * 4, 2, 1, (2, 1))
    yc = c_oracle.to_rgb(x.numpy(), sd['m.conv.weight'].reshape(3, c['cin']).numpy(), style.numpy(),
                         sd['m.bias'].reshape(3).numpy(), up)
    np.testing.assert_allclose(yc, ref, **tol)


def _sd_from_manifest(kind, man, seed):
    shapes = {}
    for k, v in man.items():
        if k.endswith('num_batches_tracked'):
            shapes[k] = torch.zeros(v, dtype=torch.long)
        elif k.endswith('.kernel'):
            taps = T.make_kernel([1, 3, 3, 1])
            # Upsample / Blur(upsample_factor=2) hold taps*4; the discriminator's Blur holds plain taps
            shapes[k] = taps * 4 if kind == 'generator' else taps
        else:
            shapes[k] = torch.empty(v)
    return synth.state_dict(kind, shapes, seed=seed)


def _img_close(img, g, name, stride, rel=1e-4):
    sub = img.detach().numpy()[..., ::stride, ::stride]
    ref = g[name + '/sub']
    scale = np.abs(ref).max()
    np.testing.assert_allclose(sub, ref, atol=rel * scale, rtol=rel)
    st = g[name + '/stats']
    a = img.detach().numpy().astype(np.float64)
    np.testing.assert_allclose([a.mean(), np.abs(a).mean()], st[:2], atol=rel * scale, rtol=rel)


@pytest.mark.parametrize('c', cases.GENERATOR_CASES, ids=lambda c: c['name'])
def test_generator_oracle(c, golden):
    g = golden('generator')
    sd = _sd_from_manifest('generator', golden.manifest('generator')[c['name']], 4)
    with torch.no_grad():
        if c['mode'] == 'latent':
            cin0 = c['shape'][0] if c['shape'] else 512
            n_latent = int(np.log2(c['size'])) * 2 - 2
            lat = synth.tensor(c['name'] + '/latent', (c['b'], n_latent, 512))
            tsr = synth.tensor(c['name'] + '/tsr', (c['b'], cin0, 4, 4))
            img = T.generator_forward(sd, c['size'], lat, external_input_tensor=tsr, noise='buffers')
        else:
            z = synth.tensor(c['name'] + '/z', (c['b'], 512))
            img = T.generator_forward(sd, c['size'], T.mapping_network(sd, z, c['n_mlp']), noise='buffers')
    _img_close(img, g, c['name'], c['stride'])


@pytest.mark.parametrize('c', cases.E2E_CASES, ids=lambda c: c['name'])
def test_e2e_oracle(c, golden):
    g = golden('e2e')
    man = golden.manifest('encoders')
    n_latent = int(np.log2(c['size'])) * 2 - 2
    sd_tsr = _sd_from_manifest('resnet', man['resnet'], 5)
    sd_w = _sd_from_manifest('resnet', man['resnet'], 6)
    sd_wp = _sd_from_manifest('psp', man[f'psp{n_latent}'], 7)
    gname = 'g256_full' if c['size'] == 256 else 'g1024_full'
    sd_g = _sd_from_manifest('generator', golden.manifest('generator')[gname], 4)
    p = synth.tensor(c['name'] + '/photo', (c['b'], 3, 256, 256), dist='uniform')
    r = synth.tensor(c['name'] + '/render', (c['b'], 3, 256, 256), dist='uniform')
    with torch.no_grad():
        enc_tol = dict(atol=2e-4, rtol=2e-4)
        np.testing.assert_allclose(T.resnet18_forward(sd_tsr, p, True).numpy(), g[c['name'] + '/e_tsr'], **enc_tol)
        np.testing.assert_allclose(T.resnet18_forward(sd_w, r, False).numpy(), g[c['name'] + '/e_w'], **enc_tol)
        wp = T.psp_forward(sd_wp, p, n_latent).numpy()
        ref_wp = g[c['name'] + '/e_wplus']
        np.testing.assert_allclose(wp, ref_wp, atol=2e-4 * np.abs(ref_wp).max(), rtol=2e-4)
        img = T.forward_inference_3_encoder(p, r, sd_tsr, sd_w, sd_wp, sd_g, c['size'], c['tsr_encode'],
                                            c['sliced_layer'], c['use_tanh'])
    _img_close(img, g, c['name'], c['stride'], rel=5e-4)


@pytest.mark.parametrize('c', cases.DISCRIMINATOR_CASES, ids=lambda c: c['name'])
def test_discriminator_oracle(c, golden):
    g = golden('discriminator')
    sd = _sd_from_manifest('discriminator', golden.manifest('discriminator')[c['name']], 8)
    x = synth.tensor(c['name'] + '/x', (c['b'], 3, c['size'], c['size']), dist='uniform')
    with torch.no_grad():
        y = T.discriminator_forward(sd, x, c['size'])
    ref = g[c['name'] + '/out']
    np.testing.assert_allclose(y.numpy(), ref, atol=1e-4 * max(1.0, np.abs(ref).max()), rtol=1e-4)


@pytest.mark.parametrize('c', cases.TENSOR2IM_CASES, ids=lambda c: c['name'])
def test_tensor2im_oracle(c, golden):
    ref = golden('image_io')[c['name'] + '/im']
    np.testing.assert_array_equal(T.tensor2im_batch(cases.tensor2im_input(c))[0], ref)
