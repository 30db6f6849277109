"""Derived weights (EqualLinear weight*scale, ModulatedConv2d MFMA layout + demodulation sums) are re-derived from the
LIVE parameters by every inference forward (op/live_weights.py): an in-place update through `.data` — how the reference's
EMA `accumulate` writes g_ema (train_3_encoder.py:195-200) — must change the next no_grad output, on every path that
used to cache (round-1 ADVICE, high)."""
import numpy as np
import pytest
import torch

import synth


def test_refresh_entry_layout_matches_library():
    from op import _native, live_weights
    assert _native.lib().fmgan_refresh_entry_bytes() == live_weights._ENTRY.itemsize
    L = _native.lib()
    assert L.fmgan_weight_refresh_blocks(0, 0, 0, 0, 5000) == 3
    # kind 1: one block per 64 output channels x 16 input channels (csrc/live_weights.hip LW_TO x LW_TI)
    assert L.fmgan_weight_refresh_blocks(1, 70, 9, 9, 0) == 2 * 1
    assert L.fmgan_weight_refresh_blocks(1, 130, 40, 9, 0) == 3 * 3
    assert L.fmgan_weight_refresh_blocks(1, 8, 8, 25, 0) == -1
    assert L.fmgan_weight_refresh_f32(None, 0, 0, None) == 0
    assert L.fmgan_weight_refresh_f32(None, 3, 10, None) == -1


def dev():
    return torch.device('cuda', 0)


def _gen(size=32, shape=None, n_mlp=2, seed=31):
    import stylegan2
    G = stylegan2.Generator(size, 512, n_mlp, generator_net_shape=shape)
    G.load_state_dict(synth.state_dict('generator', G.state_dict(), seed=seed))
    return G.to(dev()).eval()


def _run(G, lat, tsr, **kw):
    with torch.no_grad():
        return G(None, latent_styles=[lat], input_is_latent=True, use_external_input_tensor=True,
                 external_input_tensor=tsr, randomize_noise=False, **kw).clone()


@pytest.mark.gpu
def test_refresh_equals_per_module_derivation_bitwise():
    """One refresh launch == weight*scale / bias*lr_mul / modconv_weight_prep / modconv_wsq of each module."""
    from op import _native
    from op.live_weights import LiveWeights
    import stylegan2
    G = _gen(64, [24, 24, 40, 40, 136, 136, 70, 70, 9, 9])      # ragged channel counts: partial 32x8 tiles
    lw = LiveWeights(G)
    assert lw.refresh()
    n_lin = n_conv = 0
    for m in G.modules():
        if isinstance(m, stylegan2.EqualLinear):
            _, ws, bs = m._live
            assert torch.equal(ws, (m.weight * m.scale).detach())
            if m.bias is not None:
                assert torch.equal(bs, (m.bias * m.lr_mul).detach())
            n_lin += 1
        elif isinstance(m, stylegan2.ModulatedConv2d) and m.kernel_size == 3:
            _, wt, wsq = m._live
            assert torch.equal(wt, _native.modconv_weight_prep(m.weight.detach(), m.scale))
            assert torch.equal(wsq, _native.modconv_wsq(m.weight.detach()))
            n_conv += 1
    assert n_conv == 9 and n_lin >= 9 + 5 + 2


@pytest.mark.gpu
@pytest.mark.parametrize('which', ['modulation', 'conv', 'rgb_conv', 'last_conv', 'mapping'])
def test_generator_sees_data_mutation(which):
    """no_grad forward, p.data.mul_(0.5), no_grad forward: the output must change and must equal a fresh network
    loaded with the mutated weights."""
    G = _gen(32, None, 2)
    lat = synth.tensor('lw/lat', (2, G.n_latent, 512)).to(dev())
    tsr = synth.tensor('lw/tsr', (2, 512, 4, 4)).to(dev())
    z = synth.tensor('lw/z', (2, 512)).to(dev())

    def run(g):
        if which == 'mapping':
            with torch.no_grad():
                return g([z], randomize_noise=False).clone()
        return _run(g, lat, tsr)

    a = run(G)
    assert torch.equal(run(G), a)
    p = {'modulation': G.convs[1].conv.modulation.weight, 'conv': G.convs[0].conv.weight,
         'rgb_conv': G.to_rgbs[1].conv.modulation.weight, 'last_conv': G.convs[-1].conv.weight,
         'mapping': G.style[1].weight}[which]
    v0 = p._version
    p.data.mul_(0.5)
    assert p._version == v0, 'the hazard under test: .data writes do not bump the version'
    b = run(G)
    assert not torch.equal(a, b), f'{which}: stale derived weights'
    G2 = _gen(32, None, 2)
    G2.load_state_dict(G.state_dict())
    assert torch.equal(run(G2), b)


@pytest.mark.gpu
def test_standalone_modules_see_data_mutation():
    import stylegan2
    lin = stylegan2.EqualLinear(64, 32, lr_mul=0.5, activation='fused_lrelu').to(dev())
    x = synth.tensor('lw/x', (3, 64)).to(dev())
    with torch.no_grad():
        a = lin(x).clone()
        lin.weight.data.mul_(0.0)
        b = lin(x)
    assert not torch.equal(a, b)
    for up in (False, True):
        sc = stylegan2.StyledConv(16, 16, 3, 512, upsample=up).to(dev())
        sc.load_state_dict({k: v.to(dev()) for k, v in synth.state_dict('generator', sc.state_dict(), seed=2).items()})
        xi = synth.tensor('lw/xi', (2, 16, 8, 8)).to(dev())
        w = synth.tensor('lw/w', (2, 512)).to(dev())
        nz = synth.tensor('lw/nz', (1, 1, 16 if up else 8, 16 if up else 8)).to(dev())
        with torch.no_grad():
            a = sc(xi, w, noise=nz).clone()
            sc.conv.weight.data.mul_(0.5)
            b = sc(xi, w, noise=nz).clone()
            sc.conv.modulation.weight.data.mul_(0.5)
            c = sc(xi, w, noise=nz)
        assert not torch.equal(a, b) and not torch.equal(b, c)


@pytest.mark.gpu
def test_ema_accumulate_reaches_the_next_inference_forward():
    """accumulate(g_ema, g) (train_3_encoder.py:195-200) followed by a no_grad g_ema forward — both with this build's
    accumulate and with the reference's literal `.data.mul_().add_()` form."""
    import copy
    from Util.training_util import accumulate
    G = _gen(32, None, 2, seed=31)
    g_ema = copy.deepcopy(G)
    lat = synth.tensor('lw/lat', (2, G.n_latent, 512)).to(dev())
    tsr = synth.tensor('lw/tsr', (2, 512, 4, 4)).to(dev())
    a = _run(g_ema, lat, tsr)
    with torch.no_grad():
        for p in G.parameters():
            p.mul_(1.25)
    accumulate(g_ema, G, 0.5)
    b = _run(g_ema, lat, tsr)
    assert not torch.equal(a, b)
    ref = copy.deepcopy(G)          # reference arithmetic, written the reference's way
    par1, par2 = dict(ref.named_parameters()), dict(G.named_parameters())
    for k in par1:
        par1[k].data.mul_(0.5).add_(par2[k].data, alpha=0.5)
    g2 = copy.deepcopy(g_ema)
    for k, p in g2.named_parameters():
        p.data.mul_(0.5).add_(par2[k].data, alpha=0.5)
    c = _run(g2, lat, tsr)
    assert not torch.equal(b, c)
    fresh = _gen(32, None, 2)
    fresh.load_state_dict(g2.state_dict())
    assert torch.equal(_run(fresh, lat, tsr), c)


@pytest.mark.gpu
def test_psp_heads_see_data_mutation():
    import types
    from psp_encoder_model.encoders import psp_encoders
    e = psp_encoders.GradualStyleEncoder(18, 'ir_se', types.SimpleNamespace(input_nc=3, n_styles=4))
    e.load_state_dict(synth.state_dict('psp', e.state_dict(), seed=7))
    e = e.to(dev()).eval()
    p = synth.tensor('lw/photo', (1, 3, 256, 256), dist='uniform').to(dev())
    with torch.no_grad():
        a = e(p).clone()
        e.styles[2].linear.weight.data.mul_(0.5)
        b = e(p)
    np.testing.assert_allclose(b[:, :2].cpu().numpy(), a[:, :2].cpu().numpy(), atol=1e-4 * float(a.abs().max()))
    assert float((b[:, 2] - a[:, 2]).abs().max()) > 1e-3 * float(a[:, 2].abs().max())


@pytest.mark.gpu
def test_psp_grouped_heads_equal_per_head_form_and_track_the_live_parameters(monkeypatch):
    """GradualStyleEncoder's inference path runs the heads of a pyramid level as one chain of wide / grouped convs over
    parameters re-pointed at slices of one buffer (psp_encoders._flatten_heads).  Same values as the per-head form
    (MIOpen picks other kernels: rounding-level differences only), and the wide weights ARE the parameters: in-place
    updates, load_state_dict and re-pointed `.data` are all seen by the next forward."""
    import types
    from psp_encoder_model.encoders import psp_encoders
    e = psp_encoders.GradualStyleEncoder(18, 'ir_se', types.SimpleNamespace(input_nc=3, n_styles=10))
    sd = synth.state_dict('psp', e.state_dict(), seed=7)
    e.load_state_dict(sd)
    e = e.to(dev()).eval()
    p = synth.tensor('gh/photo', (2, 3, 256, 256), dist='uniform').to(dev())
    with torch.no_grad():
        monkeypatch.setattr(psp_encoders, 'GROUP_HEADS', False)
        ref = e(p).clone()
        monkeypatch.setattr(psp_encoders, 'GROUP_HEADS', True)
        a = e(p).clone()
        scale = float(ref.abs().max())
        assert float((a - ref).abs().max()) <= 5e-6 * scale
        assert e._flat is not None and [len(idx) for idx, _ in e._flat] == [3, 4, 3]
        # the parameters are views of the wide buffers
        wide = e._flat[2][1][0][0]
        assert e.styles[8].convs[0].weight.data_ptr() == wide.data_ptr() + wide[0:512].numel() * 4
        # in-place update of ONE head's conv weight (optimizer step / EMA accumulate): only that head's latent moves
        e.styles[8].convs[2].weight.data.mul_(0.5)
        b = e(p).clone()
        assert float((b[:, 8] - a[:, 8]).abs().max()) > 1e-3 * float(a[:, 8].abs().max())
        keep = [j for j in range(10) if j != 8]
        assert float((b[:, keep] - a[:, keep]).abs().max()) <= 5e-6 * scale
        # load_state_dict copies in place: back to the first result
        e.load_state_dict({k: v.to(dev()) for k, v in sd.items()})
        assert float((e(p) - a).abs().max()) <= 5e-6 * scale
        # a parameter re-pointed elsewhere is noticed (flattened again) instead of silently ignored
        e.styles[1].convs[0].bias.data = e.styles[1].convs[0].bias.data.clone() + 1.0
        c = e(p).clone()
        assert float((c[:, 1] - a[:, 1]).abs().max()) > 1e-3 * float(a[:, 1].abs().max())
        assert e.styles[1].convs[0].bias.data_ptr() == e._flat[0][1][0][1].data_ptr() + 512 * 4
        # the per-head form on the flattened parameters gives the same thing
        monkeypatch.setattr(psp_encoders, 'GROUP_HEADS', False)
        assert float((e(p) - c).abs().max()) <= 5e-6 * scale
