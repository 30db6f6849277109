"""GPU parity tests for upfirdn2d and fused_bias_act: HIP path (through the C-ABI) vs the golden vectors captured
from the reference and vs the CPU oracle, plus size-independent properties at BASELINE's full sizes."""
import numpy as np
import pytest
import torch

import cases
import synth

pytestmark = pytest.mark.gpu
OP_TOL = dict(atol=1e-5, rtol=1e-5)   # SURVEY.md §8c / BASELINE.md §4: per-op fp32 tolerance


def dev():
    return torch.device('cuda', 0)


def _run_ufd(x, k, c, force_path=-1):
    from op import _native
    n, ch, h, w = x.shape
    p0, p1 = c['pad']
    y = _native.upfirdn2d(x.reshape(-1, h, w, 1), k, c['up'], c['up'], c['down'], c['down'], p0, p1, p0, p1, force_path)
    return y.view(n, ch, y.shape[1], y.shape[2])


@pytest.mark.parametrize('c', cases.UPFIRDN2D_CASES, ids=lambda c: c['name'])
def test_upfirdn2d_golden(c, golden):
    from op import upfirdn2d, _native
    g = golden('upfirdn2d')
    x = synth.tensor(c['name'] + '/x', c['shape']).to(dev())
    k = cases.make_fir(c['kernel']).to(dev())
    ref = g[c['name'] + '/out']
    y = upfirdn2d(x, k, up=c['up'], down=c['down'], pad=tuple(c['pad']))
    np.testing.assert_allclose(y.cpu().numpy(), ref, **OP_TOL)
    # every kernel path that accepts these arguments gives the same answer
    for path in (0, 1, 2, 3):
        try:
            yp = _run_ufd(x, k, c, path)
        except RuntimeError as e:
            assert 'unsupported' in str(e)
            continue
        np.testing.assert_allclose(yp.cpu().numpy(), ref, err_msg=f'path {path}', **OP_TOL)
    # float64 through the generic kernel
    y64 = upfirdn2d(x.double(), k.double(), up=c['up'], down=c['down'], pad=tuple(c['pad']))
    np.testing.assert_allclose(y64.cpu().numpy(), ref, atol=2e-5, rtol=2e-5)
    # float16: storage rounding only (fp32 accumulation)
    y16 = upfirdn2d(x.half(), k.half(), up=c['up'], down=c['down'], pad=tuple(c['pad']))
    np.testing.assert_allclose(y16.float().cpu().numpy(), ref, atol=2e-2 * max(1.0, np.abs(ref).max()), rtol=2e-2)


@pytest.mark.parametrize('c', [c for c in cases.UPFIRDN2D_CASES if c['grad']], ids=lambda c: c['name'])
def test_upfirdn2d_gradients_golden(c, golden):
    from op import upfirdn2d
    g = golden('upfirdn2d')
    x = synth.tensor(c['name'] + '/x', c['shape']).to(dev()).requires_grad_(True)
    k = cases.make_fir(c['kernel']).to(dev())
    y = upfirdn2d(x, k, up=c['up'], down=c['down'], pad=tuple(c['pad']))
    go = synth.tensor(c['name'] + '/go', y.shape).to(dev()).requires_grad_(True)
    gi, = torch.autograd.grad(y, x, go, create_graph=True)
    np.testing.assert_allclose(gi.detach().cpu().numpy(), g[c['name'] + '/grad_input'], **OP_TOL)
    ggi = synth.tensor(c['name'] + '/ggi', x.shape).to(dev())
    gg, = torch.autograd.grad(gi, go, ggi)
    np.testing.assert_allclose(gg.cpu().numpy(), g[c['name'] + '/gradgrad_out'], **OP_TOL)


def test_upfirdn2d_gradcheck_f64():
    from op import upfirdn2d
    k = cases.make_fir(('rand', 4, 4, 21)).double().to(dev())
    for up, down, pad in ((1, 1, (1, 1)), (2, 1, (2, 1)), (1, 2, (1, 1))):
        x = synth.tensor(f'gc/{up}{down}', (1, 2, 6, 6), dtype=torch.float64).to(dev()).requires_grad_(True)
        assert torch.autograd.gradcheck(lambda t: upfirdn2d(t, k, up=up, down=down, pad=pad), (x,))
        assert torch.autograd.gradgradcheck(lambda t: upfirdn2d(t, k, up=up, down=down, pad=pad), (x,))


@pytest.mark.parametrize('shape,kernel,pad', [
    ((3, 5, 131, 257), ('rand', 4, 4, 31), (1, 1)),      # ragged: partial strips, odd widths, VEC=4
    ((2, 3, 100, 100), ('rand', 4, 4, 32), (2, 2)),      # VEC=2 strips, bigger output than input
    ((1, 4, 70, 67), ('rand', 3, 4, 33), (1, 2)),        # VEC=1, non-square taps
    ((1, 2, 513, 513), 'blur4', (1, 1)),
    ((2, 2, 300, 1025), ('rand', 4, 4, 34), (1, 1)),     # 1025-wide rows: never 16-byte aligned
    ((1, 1, 1024, 1024), ('rand', 4, 4, 35), (2, 2)),    # grad of the headline blur: 1025x1025 out
])
def test_upfirdn2d_rowmarch_vs_c_oracle(shape, kernel, pad):
    from oracle import c_oracle
    c = dict(up=1, down=1, pad=pad)
    x = synth.tensor(f'rm/{shape}', shape)
    k = cases.make_fir(kernel)
    n, ch, h, w = shape
    ref = c_oracle.upfirdn2d(x.reshape(n * ch, h, w, 1).numpy(), k.numpy(), (1, 1), (1, 1), (pad[0], pad[1], pad[0], pad[1]))
    y1 = _run_ufd(x.to(dev()), k.to(dev()), c, 1)
    y0 = _run_ufd(x.to(dev()), k.to(dev()), c, 0)
    np.testing.assert_allclose(y1.cpu().numpy().reshape(ref.shape), ref, **OP_TOL)
    np.testing.assert_allclose(y0.cpu().numpy().reshape(ref.shape), ref, **OP_TOL)


@pytest.mark.parametrize('shape,pad', [((3, 129, 129), (1, 1)), ((2, 65, 65), (1, 1)), ((2, 257, 300), (2, 2)),
                                       ((5, 9, 9), (1, 1)), ((1, 1025, 1025), (1, 1))])
def test_upfirdn2d_strided_input_matches_contiguous(shape, pad):
    """The aligned-row layout of the private conv_transpose -> blur intermediate (row stride padded to 4 floats,
    pad0 floats of left offset) must give bit-identical results to the contiguous tensor."""
    from op import _native
    n, h, w = shape
    k = cases.make_fir(('rand', 4, 4, 51)).to(dev())
    x = synth.tensor(f'strided/{shape}', shape).to(dev())
    ref = _native.upfirdn2d(x.reshape(n, h, w, 1), k, 1, 1, 1, 1, pad[0], pad[1], pad[0], pad[1])
    buf, p0, ps, rs = _native.aligned_rows_buffer(n, 1, h, w, pad[0], dev())
    buf.fill_(float('nan'))                      # the padding must never leak into the result
    off = pad[0] % 4
    buf[:, :, off:off + w] = x
    assert p0 == buf.data_ptr() + 4 * off and rs % 32 == 0
    y = _native.upfirdn2d_strided(p0, dev(), n, h, w, ps, rs, k, pad[0], pad[1], pad[0], pad[1])
    assert torch.equal(y, ref.view_as(y))


@pytest.mark.parametrize('cfg', [(2, 5, 64, 2), (1, 3, 129, 1), (3, 4, 100, 3), (1, 2, 300, 1)])
def test_blur_with_fused_epilogue_matches_two_passes(cfg):
    """fmgan_blur_noise_bias_act_f32 (noise + bias + lrelu folded into the blur's store) must equal the blur followed by
    fmgan_noise_bias_act_f32 bit for bit — same roundings, same order."""
    from op import _native
    b, c, hw, nb = cfg
    k = (cases.make_fir('blur4')).to(dev())
    x = synth.tensor(f'fep/{cfg}/x', (b * c, 2 * hw + 1, 2 * hw + 1)).to(dev())
    nz = synth.tensor(f'fep/{cfg}/n', (nb, 1, 2 * hw, 2 * hw)).to(dev())
    nw = torch.tensor([0.37], device=dev())
    bias = synth.tensor(f'fep/{cfg}/b', (c,)).to(dev())
    buf, p0, ps, rs = _native.aligned_rows_buffer(b, c, 2 * hw + 1, 2 * hw + 1, 1, dev())
    buf.fill_(float('nan'))
    buf[:, :, 1:2 * hw + 2] = x
    y2 = _native.upfirdn2d_strided(p0, dev(), b * c, 2 * hw + 1, 2 * hw + 1, ps, rs, k, 1, 1, 1, 1).view(b, c, 2 * hw, 2 * hw)
    ref = _native.noise_bias_act(y2, nz, nw, bias, 0.2, 2 ** 0.5)
    y = _native.blur_noise_bias_act(p0, dev(), b, c, 2 * hw + 1, 2 * hw + 1, ps, rs, k, (1, 1), nz, nw, bias, 0.2, 2 ** 0.5)
    assert y is not None and torch.equal(y, ref)
    # without noise / bias
    y0 = _native.blur_noise_bias_act(p0, dev(), b, c, 2 * hw + 1, 2 * hw + 1, ps, rs, k, (1, 1), None, None, None, 0.2, 2 ** 0.5)
    assert torch.equal(y0, _native.noise_bias_act(y2, None, None, None, 0.2, 2 ** 0.5))


@pytest.mark.parametrize('cfg', [(8, 512, 4, 8), (3, 40, 8, 1), (2, 24, 16, 2), (1, 7, 31, 1), (2, 3, 50, 2)])
def test_small_plane_blur_with_fused_epilogue(cfg):
    """The 4^2 -> 8^2 ... 16^2 -> 32^2 upsampling layers (and any plane narrower than 64): the plane-tile kernel reads the
    aligned-row intermediate and applies noise + bias + lrelu in its store.  Bitwise equal to blur + fmgan_noise_bias_act,
    the blur itself against the C oracle, NaN padding must not leak."""
    from op import _native
    from oracle import c_oracle
    b, c, hw, nb = cfg
    n = 2 * hw + 1
    k = cases.make_fir('blur4').to(dev())
    x = synth.tensor(f'spf/{cfg}/x', (b * c, n, n)).to(dev())
    nz = synth.tensor(f'spf/{cfg}/n', (nb, 1, 2 * hw, 2 * hw)).to(dev())
    nw = torch.tensor([-0.83], device=dev())
    bias = synth.tensor(f'spf/{cfg}/b', (c,)).to(dev())
    buf, p0, ps, rs = _native.aligned_rows_buffer(b, c, n, n, 1, dev())
    buf.fill_(float('nan'))
    buf[:, :, 1:n + 1] = x
    assert _native.lib().fmgan_upfirdn2d_select(0, b * c, n, n, 1, 4, 4, 1, 1, 1, 1, 1, 1, 1, 1) == (2 if 2 * hw < 64 else 1)
    y2 = _native.upfirdn2d_strided(p0, dev(), b * c, n, n, ps, rs, k, 1, 1, 1, 1).view(b, c, 2 * hw, 2 * hw)
    yc = _native.upfirdn2d(x.reshape(b * c, n, n, 1), k, 1, 1, 1, 1, 1, 1, 1, 1).view(b, c, 2 * hw, 2 * hw)
    assert torch.equal(y2, yc)                                  # strided read == contiguous read
    ref = c_oracle.upfirdn2d(x[:3].reshape(3, n, n, 1).cpu().numpy(), k.cpu().numpy(), (1, 1), (1, 1), (1, 1, 1, 1))
    np.testing.assert_allclose(y2.reshape(b * c, 2 * hw, 2 * hw)[:3].cpu().numpy().reshape(ref.shape), ref, **OP_TOL)
    y = _native.blur_noise_bias_act(p0, dev(), b, c, n, n, ps, rs, k, (1, 1), nz, nw, bias, 0.2, 2 ** 0.5)
    assert y is not None and not torch.isnan(y).any()
    assert torch.equal(y, _native.noise_bias_act(y2, nz, nw, bias, 0.2, 2 ** 0.5))
    y0 = _native.blur_noise_bias_act(p0, dev(), b, c, n, n, ps, rs, k, (1, 1), None, None, None, 0.2, 2 ** 0.5)
    assert torch.equal(y0, _native.noise_bias_act(y2, None, None, None, 0.2, 2 ** 0.5))
    # insisting on a row-march variant for such planes is refused, not silently served by another kernel
    assert _native.blur_noise_bias_act(p0, dev(), b, c, n, n, ps, rs, k, (1, 1), nz, nw, bias, 0.2, 1.0, force_path=5) is None


@pytest.mark.parametrize('cfg', [
    # (batch, channels, in_h, in_w, pad, taps, noise_batch)
    (1, 2, 257, 257, (1, 1), 'blur4', 1),          # out 256 x 256: one strip
    (2, 3, 101, 261, (1, 1), 'blur4', 2),          # out 260 wide (second strip nearly empty), 100 rows: partial last tile
    (1, 2, 65, 512, (2, 1), ('rand', 4, 4, 7), 1), # pad0 = 2: two masked columns on the left; out 512 x 65
    (1, 1, 40, 1027, (3, 2), ('rand', 3, 4, 9), 1),# 3-row kernel, pad0 = 3, out 1029 - not a multiple of 4 -> path 1 on both sides
    (1, 2, 36, 1029, (3, 2), ('rand', 4, 3, 11), 1),  # out 1032 x 38, 3-column kernel
    (2, 2, 1025, 1025, (1, 1), 'blur4', 1),        # the headline geometry, 4 planes
    # production tile heights (the cases above get 8-row tiles: 11 steps, the steady-state loop's back-edge is never
    # taken): 512 planes of [257, 1025] -> 64-row tiles (67 steps), per-sample noise; 256 planes -> 32-row tiles
    (16, 32, 257, 1025, (1, 1), 'blur4', 16),
    (8, 32, 257, 1025, (1, 1), 'blur4', 1),
])
def test_dma_ring_blur_equals_register_row_march(cfg):
    """upfirdn2d.hip path 1b (LDS-DMA ring, hand-counted s_waitcnt) against path 1 (register staging) on the same
    aligned-row buffer: bit-identical, plain and with the fused noise/bias/lrelu store; NaN padding must not leak."""
    from op import _native
    from oracle import c_oracle
    b, c, h, w, pad, taps, nb = cfg
    k = cases.make_fir(taps).to(dev())
    kh, kw = k.shape
    oh, ow = h + pad[0] + pad[1] - kh + 1, w + pad[0] + pad[1] - kw + 1
    x = synth.tensor(f'dma/{cfg}/x', (b * c, h, w)).to(dev())
    nz = synth.tensor(f'dma/{cfg}/n', (nb, 1, oh, ow)).to(dev())
    nw = torch.tensor([-0.61], device=dev())
    bias = synth.tensor(f'dma/{cfg}/b', (c,)).to(dev())
    buf, p0, ps, rs = _native.aligned_rows_buffer(b, c, h, w, pad[0], dev())
    buf.fill_(float('nan'))
    off = pad[0] % 4
    buf[:, :, off:off + w] = x
    res = {}
    ring_ok = ow % 4 == 0      # path 1b needs 16-byte output rows; otherwise both sides are path 1 (kept as a case)
    for mode, path in (('0', 4), ('1', 5 if ring_ok else -1)):
        plain = _native.upfirdn2d_strided(p0, dev(), b * c, h, w, ps, rs, k, pad[0], pad[1], pad[0], pad[1], force_path=path)
        fused = _native.blur_noise_bias_act(p0, dev(), b, c, h, w, ps, rs, k, pad, nz, nw, bias, 0.2, 2 ** 0.5,
                                            force_path=path)
        res[mode] = (plain, fused)
    auto = _native.blur_noise_bias_act(p0, dev(), b, c, h, w, ps, rs, k, pad, nz, nw, bias, 0.2, 2 ** 0.5)
    assert torch.equal(auto, res['1'][1])
    assert res['0'][0].shape[-2:] == (oh, ow)
    assert torch.equal(res['0'][0], res['1'][0])
    assert res['0'][1] is not None and torch.equal(res['0'][1], res['1'][1])
    assert not torch.isnan(res['1'][1]).any()
    ref = c_oracle.upfirdn2d(x[:1].reshape(1, h, w, 1).cpu().numpy(), k.cpu().numpy(), (1, 1), (1, 1),
                             (pad[0], pad[1], pad[0], pad[1]))
    np.testing.assert_allclose(res['1'][0][:1].cpu().numpy().reshape(ref.shape), ref, **OP_TOL)


def test_modconv_strided_output_matches_contiguous():
    from op import _native
    for (b, cin, cout, h, w) in ((2, 8, 40, 16, 16), (1, 16, 130, 9, 7), (9, 12, 20, 4, 4)):
        x = synth.tensor(f'so/{cout}/x', (b, cin, h, w)).to(dev())
        wgt = synth.tensor(f'so/{cout}/w', (cout, cin, 3, 3)).to(dev())
        s = synth.tensor(f'so/{cout}/s', (b, cin), shift=1.0, scale=0.5).to(dev())
        scale = 1.0 / (cin * 9) ** 0.5
        wt = _native.modconv_weight_prep(wgt, scale)
        dm = _native.modconv_demod(wgt, s, scale)
        ref = _native.modconv2d(x, wt, s, dm, 1)
        oh, ow = 2 * h + 1, 2 * w + 1
        buf, p0, ps, rs = _native.aligned_rows_buffer(b, cout, oh, ow, 1, dev())
        buf.fill_(float('nan'))
        _native.modconv2d(x, wt, s, dm, 1, strided_out=(p0, ps, rs))
        assert torch.equal(buf[:, :, 1:1 + ow].reshape(b, cout, oh, ow), ref)
        assert torch.isnan(buf[:, :, 0]).all() and torch.isnan(buf[:, :, 1 + ow:]).all()


def test_upfirdn2d_many_planes_and_minor():
    """major > 16384 (the reference's loop_major path, op/upfirdn2d_kernel.cu:296) and minor > 1 (generic kernel)."""
    from op import _native
    from oracle import c_oracle
    k = cases.make_fir(('rand', 4, 4, 41))
    x = synth.tensor('many/x', (20000, 9, 9, 1))
    ref = c_oracle.upfirdn2d(x.numpy(), k.numpy(), (1, 1), (1, 1), (1, 1, 1, 1))
    y = _native.upfirdn2d(x.to(dev()), k.to(dev()), 1, 1, 1, 1, 1, 1, 1, 1)
    np.testing.assert_allclose(y.cpu().numpy(), ref, **OP_TOL)
    xm = synth.tensor('minor/x', (3, 10, 11, 5))
    refm = c_oracle.upfirdn2d(xm.numpy(), k.numpy(), (2, 1), (1, 2), (2, 1, 1, 1))
    ym = _native.upfirdn2d(xm.to(dev()), k.to(dev()), 2, 1, 1, 2, 2, 1, 1, 1)
    np.testing.assert_allclose(ym.cpu().numpy(), refm, **OP_TOL)


def test_upfirdn2d_headline_properties():
    """BASELINE headline call [B*32,1025,1025] -> [B*32,1024,1024] at B=8 (2.15 GB): too big for the CPU oracle
    in seconds, so check (a) sampled planes against the C oracle, (b) linearity, (c) DC gain = sum(taps),
    (d) the adjoint identity <blur(x), g> == <x, blur^T(g)> that the backward relies on."""
    from op import upfirdn2d, _native
    from oracle import c_oracle
    d = dev()
    k = cases.make_fir('blur4').to(d)
    gen = torch.Generator(device=d).manual_seed(1234)
    x = torch.randn(8, 32, 1025, 1025, device=d, generator=gen)
    y = upfirdn2d(x, k, pad=(1, 1))
    assert tuple(y.shape) == (8, 32, 1024, 1024)
    assert _native.lib().fmgan_upfirdn2d_select(0, 256, 1025, 1025, 1, 4, 4, 1, 1, 1, 1, 1, 1, 1, 1) == 1
    for (b, c) in ((0, 0), (3, 17), (7, 31)):
        ref = c_oracle.upfirdn2d(x[b, c].cpu().numpy().reshape(1, 1025, 1025, 1), k.cpu().numpy(), (1, 1), (1, 1), (1, 1, 1, 1))
        np.testing.assert_allclose(y[b, c].cpu().numpy(), ref.reshape(1024, 1024), **OP_TOL)
    x2 = torch.randn(1, 32, 1025, 1025, device=d, generator=gen)
    lin = upfirdn2d(2.0 * x[:1] - 0.5 * x2, k, pad=(1, 1))
    torch.testing.assert_close(lin, 2.0 * y[:1] - 0.5 * upfirdn2d(x2, k, pad=(1, 1)), atol=2e-5, rtol=1e-5)
    ones = torch.ones(1, 1, 1025, 1025, device=d)
    dc = upfirdn2d(ones, k, pad=(1, 1))
    torch.testing.assert_close(dc[0, 0, 2:-2, 2:-2], torch.full((1020, 1020), 4.0, device=d), atol=1e-5, rtol=1e-6)
    g = torch.randn(1, 32, 1024, 1024, device=d, generator=gen)
    xa = x[:1].clone().double().requires_grad_(True)
    ya = upfirdn2d(xa, k.double(), pad=(1, 1))
    gi, = torch.autograd.grad(ya, xa, g.double())
    lhs = (ya.detach() * g.double()).sum()
    rhs = (xa.detach() * gi).sum()
    assert abs((lhs - rhs) / lhs) < 1e-10


# ------------------------------------------------------------------------------------------------ fused_bias_act
@pytest.mark.parametrize('c', cases.FUSED_ACT_CASES, ids=lambda c: c['name'])
def test_fused_act_golden(c, golden):
    from op import fused_leaky_relu
    g = golden('fused_act')
    x, b = cases.fused_act_inputs(c)
    x = x.to(dev()).requires_grad_(True)
    b = b.to(dev()).requires_grad_(True) if b is not None else None
    y = fused_leaky_relu(x, b)
    np.testing.assert_allclose(y.detach().cpu().numpy(), g[c['name'] + '/out'], atol=1e-6, rtol=1e-6)
    go = synth.tensor(c['name'] + '/go', y.shape).to(dev()).requires_grad_(True)
    ins = [x] + ([b] if b is not None else [])
    grads = torch.autograd.grad(y, ins, go, create_graph=True)
    np.testing.assert_allclose(grads[0].detach().cpu().numpy(), g[c['name'] + '/grad_input'], atol=1e-6, rtol=1e-6)
    if b is not None:
        np.testing.assert_allclose(grads[1].detach().cpu().numpy(), g[c['name'] + '/grad_bias'], atol=2e-5, rtol=2e-5)
    ggi = synth.tensor(c['name'] + '/ggi', x.shape).to(dev())
    gg, = torch.autograd.grad(grads[0], go, ggi)
    np.testing.assert_allclose(gg.cpu().numpy(), g[c['name'] + '/gradgrad_out'], atol=1e-6, rtol=1e-6)


@pytest.mark.parametrize('act,grad', [(3, 0), (3, 1), (3, 2), (1, 0), (1, 1), (1, 2)])
@pytest.mark.parametrize('shape', [(2, 6, 8, 8), (3, 5, 7, 3), (2, 32, 64, 64), (4, 512), (1, 3, 1, 1), (1031, 128)])
@pytest.mark.parametrize('dtype', [torch.float32, torch.float64, torch.float16])
def test_fused_bias_act_all_modes_vs_c_oracle(act, grad, shape, dtype):
    """Every act*10+grad code of the reference kernel (op/fused_bias_act_kernel.cu:36-45), a non-default alpha
    (the CUDA kernel honours it), with and without bias / refer — bit-exact in fp32."""
    from op import _native
    from oracle import c_oracle
    x = synth.tensor(f'fba/{shape}/x', shape)
    b = synth.tensor(f'fba/{shape}/b', (shape[1],))
    r = synth.tensor(f'fba/{shape}/r', shape)
    x.view(-1)[::7] = 0.0
    r.view(-1)[::5] = 0.0
    for use_b, use_r in ((True, True), (False, True), (True, False), (False, False)):
        xn = x.numpy().astype(np.float64 if dtype == torch.float64 else np.float32)
        if dtype == torch.float16:
            xn = x.half().float().numpy()
        bn = None if not use_b else (b.half().float() if dtype == torch.float16 else b).numpy().astype(xn.dtype)
        rn = None if not use_r else (r.half().float() if dtype == torch.float16 else r).numpy().astype(xn.dtype)
        # the reference passes alpha/scale as C++ floats converted to scalar_t (fused_bias_act_kernel.cu:52-53,79-89)
        a32, s32 = float(np.float32(0.3)), float(np.float32(1.7))
        ref = c_oracle.fused_bias_act(xn, bn, rn, act, grad, a32, s32)
        e = torch.empty(0, device=dev(), dtype=dtype)
        y = _native.fused_bias_act(x.to(dev(), dtype), b.to(dev(), dtype) if use_b else e,
                                   r.to(dev(), dtype) if use_r else e, act, grad, 0.3, 1.7)
        if dtype == torch.float16:
            np.testing.assert_allclose(y.float().cpu().numpy(), ref, atol=2e-3 * max(1, np.abs(ref).max()), rtol=2e-3)
        else:
            np.testing.assert_array_equal(y.cpu().numpy(), ref)


def test_fused_act_full_size_bit_exact():
    """[8,32,1024,1024] (1.07 GB), BASELINE cfg4's largest activation: bit-exact against the C oracle on sampled
    planes, idempotence of the sign pattern, and grad_bias == plane sums."""
    from op import fused_leaky_relu, _native
    from oracle import c_oracle
    d = dev()
    gen = torch.Generator(device=d).manual_seed(7)
    x = torch.randn(8, 32, 1024, 1024, device=d, generator=gen)
    b = torch.randn(32, device=d, generator=gen)
    y = fused_leaky_relu(x, b)
    for (n, c) in ((0, 0), (5, 13), (7, 31)):
        ref = c_oracle.fused_bias_act(x[n:n + 1, c:c + 1].cpu().numpy(), b[c:c + 1].cpu().numpy(), None, 3, 0, 0.2, 2 ** 0.5)
        np.testing.assert_array_equal(y[n:n + 1, c:c + 1].cpu().numpy(), ref)
    assert torch.equal(y > 0, (x + b.view(1, -1, 1, 1)) > 0)
    e = x.new_empty(0)
    gi = _native.fused_bias_act(torch.ones_like(x), e, y, 3, 1, 0.2, 2 ** 0.5)
    expect = torch.where(y > 0, torch.tensor(2 ** 0.5, device=d), torch.tensor(0.2 * 2 ** 0.5, device=d))
    torch.testing.assert_close(gi, expect, atol=0, rtol=0)


@pytest.mark.parametrize('shape', [(2, 6, 8, 8), (3, 5, 16, 20), (2, 32, 64, 64), (4, 16, 256, 256), (1, 3, 1024, 1024),
                                   (2, 3, 8, 8, 4)])
def test_fused_act_backward_with_bias_partials(shape):
    """One-pass backward (grad_input + per-block partial sums of it): grad_input has the bits of the reference form
    fused_bias_act(g, empty, out, 3, 1) — i.e. of the C oracle — and the bias gradient equals the float64 plane sums of
    that grad_input to fp32 summation accuracy (the reference: a float32 torch .sum of the same values)."""
    from op import _native
    from oracle import c_oracle
    g = synth.tensor(f'fbb/{shape}/g', shape)
    out = synth.tensor(f'fbb/{shape}/o', shape)
    out.view(-1)[::5] = 0.0          # `ref > 0` is strict (fused_bias_act_kernel.cu:42)
    both = _native.fused_bias_act_backward(g.to(dev()), out.to(dev()), 0.2, 2 ** 0.5)
    assert both is not None, 'shape should be served by the one-pass kernel'
    gi, gb = both
    ref = c_oracle.fused_bias_act(g.numpy(), None, out.numpy(), 3, 1, 0.2, 2 ** 0.5)
    np.testing.assert_array_equal(gi.cpu().numpy(), ref)
    dims = tuple([0] + list(range(2, len(shape))))
    exact = ref.astype(np.float64).sum(axis=dims)
    mag = np.abs(ref).astype(np.float64).sum(axis=dims)
    np.testing.assert_allclose(gb.cpu().numpy(), exact, atol=float(2e-6 * mag.max()), rtol=0)
    # and it is what autograd uses: same bits as the function's grad_bias, run twice (bit-reproducible)
    from op import fused_leaky_relu
    x = synth.tensor(f'fbb/{shape}/x', shape).to(dev()).requires_grad_(True)
    b = synth.tensor(f'fbb/{shape}/b', (shape[1],)).to(dev()).requires_grad_(True)
    go = g.to(dev())
    g1 = torch.autograd.grad(fused_leaky_relu(x, b), [x, b], go)
    g2 = torch.autograd.grad(fused_leaky_relu(x, b), [x, b], go)
    assert torch.equal(g1[0], g2[0]) and torch.equal(g1[1], g2[1])
    y = fused_leaky_relu(x, b).detach()
    e = go.new_empty(0)
    two_pass = _native.fused_bias_act(go, e, y, 3, 1, 0.2, 2 ** 0.5)
    assert torch.equal(g1[0], two_pass)
    torch.testing.assert_close(g1[1], two_pass.double().sum(dims).float(), atol=float(2e-6 * two_pass.abs().double().sum(dims).max()), rtol=0)


def test_fused_act_backward_falls_back_for_unserved_shapes():
    """2-D activations (the mapping network) and tiny / odd planes keep the two-step form."""
    from op import _native, fused_leaky_relu
    for shape in ((4, 512), (2, 6, 4, 4), (3, 5, 7, 3)):
        g = synth.tensor(f'fbf/{shape}/g', shape).to(dev())
        assert _native.fused_bias_act_backward(g, g, 0.2, 2 ** 0.5) is None
        x = g.clone().requires_grad_(True)
        b = synth.tensor(f'fbf/{shape}/b', (shape[1],)).to(dev()).requires_grad_(True)
        gx, gb = torch.autograd.grad(fused_leaky_relu(x, b), [x, b], g)
        torch.testing.assert_close(gb, gx.sum([0] + list(range(2, len(shape)))))


def test_noise_bias_act_matches_unfused_bitwise():
    from op import _native, fused_leaky_relu
    d = dev()
    for (b, c, h, nb) in ((2, 16, 8, 2), (3, 24, 16, 1), (1, 5, 7, 1), (2, 32, 64, 2)):
        x = synth.tensor(f'nba/{b}{c}{h}/x', (b, c, h, h)).to(d)
        nz = synth.tensor(f'nba/{b}{c}{h}/n', (nb, 1, h, h)).to(d)
        nw = torch.tensor([0.37], device=d)
        bias = synth.tensor(f'nba/{b}{c}{h}/b', (c,)).to(d)
        ref = fused_leaky_relu(x + nw * nz, bias)
        y = _native.noise_bias_act(x, nz, nw, bias, 0.2, 2 ** 0.5)
        assert torch.equal(y, ref)


# ------------------------------------------------------------------------------------------------ §8 f-4: image I/O
def test_images_to_tensor_bit_exact():
    from Util.image_io import images_to_tensor
    from oracle import torch_oracle as T
    g = torch.Generator().manual_seed(3)
    for shape in ((2, 256, 256, 3), (3, 17, 31, 3), (1, 1024, 1024, 3)):
        u8 = torch.randint(0, 256, shape, dtype=torch.uint8, generator=g)
        u8.view(-1)[:256] = torch.arange(256, dtype=torch.uint8)          # every byte value
        y = images_to_tensor(u8.to(dev()))
        ref = T.images_to_tensor(u8)
        assert y.shape == ref.shape and torch.equal(y.cpu(), ref)
    with pytest.raises(RuntimeError):
        images_to_tensor(torch.zeros(1, 4, 4, 3, dtype=torch.uint8))       # CPU tensor: no CPU path


@pytest.mark.parametrize('c', cases.TENSOR2IM_CASES, ids=lambda c: c['name'])
def test_tensor2im_golden(c, golden):
    from Evaluation.visual_eval import tensor2im
    ref = golden('image_io')[c['name'] + '/im']
    np.testing.assert_array_equal(tensor2im(cases.tensor2im_input(c).to(dev())), ref)


def test_tensor2im_bit_exact():
    from Evaluation.visual_eval import tensor2im, tensor2im_batch
    from oracle import torch_oracle as T
    for shape in ((2, 3, 64, 64), (3, 3, 19, 23), (1, 3, 1024, 1024)):
        x = synth.tensor(f't2i/{shape}', shape, scale=0.8)
        x.view(-1)[:8] = torch.tensor([-1.0, 1.0, -1.5, 1.5, 0.0, 0.999999, -0.999999, 0.00392])
        ref = T.tensor2im_batch(x)
        y = tensor2im_batch(x.to(dev()))
        np.testing.assert_array_equal(y.cpu().numpy(), ref)
        np.testing.assert_array_equal(tensor2im(x.to(dev())), ref[0])
    # round trip through the input converter: bytes -> [-1,1] -> bytes is the identity up to the truncation bias
    u8 = torch.randint(0, 256, (1, 32, 32, 3), dtype=torch.uint8)
    from Util.image_io import images_to_tensor
    back = tensor2im_batch(images_to_tensor(u8.to(dev()))).cpu()
    assert int((back.int() - u8.int()).abs().max()) <= 1


def test_empty_batches_are_no_ops():
    """Zero-sized leading dimension: every entry point returns an empty result without launching (the reference's ops
    accept empty tensors the same way: at::empty + a zero-block launch guard)."""
    from op import _native, upfirdn2d, fused_leaky_relu
    d = dev()
    k = torch.ones(4, 4, device=d) / 16
    y = upfirdn2d(torch.zeros(0, 3, 8, 8, device=d), k, pad=(2, 1))
    assert tuple(y.shape) == (0, 3, 8, 8)
    y = fused_leaky_relu(torch.zeros(0, 5, 4, 4, device=d), torch.zeros(5, device=d))
    assert tuple(y.shape) == (0, 5, 4, 4)
    wt = _native.modconv_weight_prep(torch.randn(6, 4, 3, 3, device=d), 0.1)
    for mode, shape in ((0, (0, 6, 8, 8)), (1, (0, 6, 17, 17)), (2, (0, 6, 3, 3))):
        out = _native.modconv2d(torch.zeros(0, 4, 8, 8, device=d), wt, torch.zeros(0, 4, device=d), None, mode)
        assert tuple(out.shape) == shape
    assert tuple(_native.modconv_demod(torch.randn(6, 4, 3, 3, device=d), torch.zeros(0, 4, device=d), 0.1).shape) == (0, 6)
    rgb = _native.torgb(torch.zeros(0, 4, 8, 8, device=d), torch.randn(3, 4, device=d), torch.zeros(0, 4, device=d),
                        torch.zeros(3, device=d), None, 0.5)
    assert tuple(rgb.shape) == (0, 3, 8, 8)
    assert tuple(_native.images_to_tensor(torch.zeros(0, 8, 8, 3, dtype=torch.uint8, device=d)).shape) == (0, 3, 8, 8)
    assert tuple(_native.tensor_to_images(torch.zeros(0, 3, 8, 8, device=d)).shape) == (0, 8, 8, 3)
    assert tuple(_native.resize_images(torch.zeros(0, 8, 8, 3, dtype=torch.uint8, device=d), 4, 4).shape) == (0, 4, 4, 3)


def test_upfirdn2d_random_arguments_vs_c_oracle():
    """Seeded sweep over the whole argument space of the native entry point (asymmetric up/down/pads incl. negative
    (cropping) pads, 1..5-tap kernels, minor > 1, f32 and f64): every dispatch path vs the C oracle."""
    from op import _native
    from oracle import c_oracle
    rng = np.random.default_rng(2024)
    done = 0
    while done < 60:
        major, h, w = int(rng.integers(1, 7)), int(rng.integers(1, 80)), int(rng.integers(1, 300))
        minor = int(rng.choice([1, 1, 1, 2, 3]))
        kh, kw = int(rng.integers(1, 6)), int(rng.integers(1, 6))
        ux, uy, dx, dy = (int(v) for v in rng.integers(1, 4, 4))
        pads = tuple(int(v) for v in rng.integers(-2, 5, 4))
        oh = (h * uy + pads[2] + pads[3] - kh) // dy + 1
        ow = (w * ux + pads[0] + pads[1] - kw) // dx + 1
        if oh <= 0 or ow <= 0 or h * uy + pads[2] + pads[3] < kh or w * ux + pads[0] + pads[1] < kw:
            continue
        if (min(pads[0], 0) + min(pads[1], 0) + w * ux) <= 0 or (min(pads[2], 0) + min(pads[3], 0) + h * uy) <= 0:
            continue
        dt = np.float64 if done % 7 == 0 else np.float32
        x = rng.standard_normal((major, h, w, minor)).astype(dt)
        k = rng.standard_normal((kh, kw)).astype(dt)
        ref = c_oracle.upfirdn2d(x, k, (ux, uy), (dx, dy), pads)
        y = _native.upfirdn2d(torch.from_numpy(x).to(dev()), torch.from_numpy(k).to(dev()), ux, uy, dx, dy, *pads)
        tol = 1e-5 if dt == np.float32 else 1e-12
        np.testing.assert_allclose(y.cpu().numpy(), ref, atol=tol * max(1.0, float(np.abs(ref).max())), rtol=tol,
                                   err_msg=f'{(major, h, w, minor)} k{kh}x{kw} up{(ux, uy)} down{(dx, dy)} pad{pads}')
        done += 1


@pytest.mark.parametrize('shape', [(2, 64, 33, 17), (3, 128, 8, 8), (1, 512, 5, 7), (4, 96, 16, 16), (2, 6, 9, 9)])
def test_prelu_backward_kernel_vs_float64_autograd(shape):
    """helpers.PReLU: the HIP backward (fmgan_prelu_backward_f32: grad_x in one pass, slope gradient from per-block partial
    sums) against float64 autograd of nn.PReLU; C = 6 is not served by the kernel and must fall back to aten."""
    from psp_encoder_model.encoders.helpers import PReLU
    n, c, h, w = shape
    x = synth.tensor(f'prelu/{shape}/x', shape).to(dev()).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    g = synth.tensor(f'prelu/{shape}/g', shape).to(dev()).contiguous(memory_format=torch.channels_last)
    a = synth.tensor(f'prelu/{shape}/a', (c,), scale=0.3).to(dev())
    m = PReLU(c).to(dev())
    m.weight.data.copy_(a)
    y = m(x)
    y.backward(g)
    ref = torch.nn.PReLU(c).double()
    ref.weight.data.copy_(a.double().cpu())
    x64 = x.detach().double().cpu().contiguous().requires_grad_(True)
    y64 = ref(x64)
    y64.backward(g.double().cpu().contiguous())
    np.testing.assert_allclose(y.detach().cpu().numpy(), y64.detach().numpy(), rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(x.grad.cpu().numpy(), x64.grad.numpy(), rtol=1e-6, atol=1e-7)
    gw64 = ref.weight.grad.numpy()
    np.testing.assert_allclose(m.weight.grad.cpu().numpy(), gw64, rtol=2e-5, atol=2e-6 * float(np.abs(gw64).max()) + 1e-6)
    # create_graph=True takes the differentiable torch formulas: same first-order values, and a second derivative exists
    x2 = x.detach().clone().requires_grad_(True)
    gx, = torch.autograd.grad(m(x2), x2, g, create_graph=True)
    np.testing.assert_allclose(gx.detach().cpu().numpy(), x64.grad.numpy(), rtol=1e-6, atol=1e-7)
    gg, = torch.autograd.grad(gx.sum(), m.weight)
    assert torch.isfinite(gg).all()


@pytest.mark.parametrize('shape', [(8, 512, 512), (1, 512, 32), (3, 40, 7), (64, 512, 512), (5, 100, 130), (2, 2048, 64)])
def test_equal_linear_matches_f_linear(shape):
    """fmgan_equal_linear_f32 (EqualLinear at inference batch sizes, stylegan2.py:146-180 / :226) against float64 F.linear:
    within fp32 summation error, bit-reproducible, and independent of which other samples share the batch."""
    from op import _native
    b, k, n = shape
    x = synth.tensor(f'lin/{shape}/x', (b, k))
    w = synth.tensor(f'lin/{shape}/w', (n, k))
    bias = synth.tensor(f'lin/{shape}/b', (n,))
    ref = torch.nn.functional.linear(x.double(), w.double(), bias.double()).numpy()
    xd, wd, bd = x.to(dev()), w.to(dev()), bias.to(dev())
    y = _native.equal_linear(xd, wd, bd)
    tol = 4e-6 * float(np.abs(x.numpy()).max() * np.abs(w.numpy()).max()) * k ** 0.5
    np.testing.assert_allclose(y.cpu().numpy(), ref, atol=tol, rtol=0)
    assert torch.equal(_native.equal_linear(xd, wd, bd), y)
    assert torch.equal(_native.equal_linear(xd[b - 1:], wd, bd), y[b - 1:])
    y0 = _native.equal_linear(xd, wd, None)
    np.testing.assert_allclose(y0.cpu().numpy(), ref - bias.numpy()[None].astype(np.float64), atol=tol, rtol=0)
    with pytest.raises(RuntimeError):
        _native.equal_linear(xd, wd[:, :-1].contiguous(), None)
