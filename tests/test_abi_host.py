"""CPU tests of the boundary and the host logic: the C-ABI library loads and exports every symbol the header
declares, pure-host entry points behave, the product has NO CPU path, and the product modules keep the
reference's parameter/buffer names and shapes (manifests captured from the reference)."""
import ctypes
import os
import re

import pytest
import torch
import torch.nn.functional as F

import cases
import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _lib():
    from op import _native
    return _native.lib()


def test_library_exports_header_symbols():
    hdr = open(os.path.join(ROOT, 'include', 'fmgan_hip.h')).read()
    hdr = re.sub(r'/\*.*?\*/', '', hdr, flags=re.S)
    names = sorted(set(re.findall(r'\b(fmgan_\w+)\s*\(', hdr)))
    assert len(names) >= 11
    L = _lib()
    for n in names:
        assert hasattr(L, n), f'{n} declared in include/fmgan_hip.h but not exported'
    assert L.fmgan_abi_version() == 1
    assert L.fmgan_status_string(0) == b'ok'
    assert b'invalid' in L.fmgan_status_string(-1)


def test_out_size_matches_reference_formula():
    from op import _native
    for c in cases.UPFIRDN2D_CASES:
        k = cases.make_fir(c['kernel'])
        h, w = c['shape'][2:]
        p0, p1 = c['pad']
        oh, ow = _native.upfirdn2d_out_size(h, w, k.shape[0], k.shape[1], c['up'], c['up'], c['down'], c['down'],
                                           p0, p1, p0, p1)
        # op/upfirdn2d.py:112-113
        assert oh == (h * c['up'] + p0 + p1 - k.shape[0]) // c['down'] + 1
        assert ow == (w * c['up'] + p0 + p1 - k.shape[1]) // c['down'] + 1


def test_path_selection_is_host_logic():
    L = _lib()
    # headline blur [256,1025,1025] -> row-march; 9x9 plane -> not row-march; f64 -> generic; bad args -> EINVAL
    assert L.fmgan_upfirdn2d_select(0, 256, 1025, 1025, 1, 4, 4, 1, 1, 1, 1, 1, 1, 1, 1) == 1
    assert L.fmgan_upfirdn2d_select(0, 1024, 9, 9, 1, 4, 4, 1, 1, 1, 1, 1, 1, 1, 1) != 1
    assert L.fmgan_upfirdn2d_select(1, 256, 1025, 1025, 1, 4, 4, 1, 1, 1, 1, 1, 1, 1, 1) == 0
    assert L.fmgan_upfirdn2d_select(0, 1, 0, 8, 1, 4, 4, 1, 1, 1, 1, 1, 1, 1, 1) == -1
    assert L.fmgan_upfirdn2d_select(7, 1, 8, 8, 1, 4, 4, 1, 1, 1, 1, 1, 1, 1, 1) == -2


def test_invalid_arguments_return_status_without_gpu():
    L = _lib()
    # null pointers / bad dims are rejected before any HIP call
    assert L.fmgan_upfirdn2d(0, None, None, None, 1, 8, 8, 1, 4, 4, 1, 1, 1, 1, 1, 1, 1, 1, -1, None) == -1
    assert L.fmgan_upfirdn2d(0, None, None, None, 1, 8, 8, 1, 4, 4, 0, 1, 1, 1, 1, 1, 1, 1, -1, None) == -1
    assert L.fmgan_fused_bias_act(0, None, None, None, None, 16, 0, 1, 3, 0, 0.2, 1.4, None) == -1
    assert L.fmgan_fused_bias_act(9, None, None, None, None, 16, 0, 1, 3, 0, 0.2, 1.4, None) == -2
    assert L.fmgan_modconv2d_f32(None, None, None, None, None, 1, 8, 8, 4, 4, 3, None, None, None, 1, 0, 0.2, 1.4, 0, 0, None, 0, None) == -2
    assert L.fmgan_modconv2d_f32(None, None, None, None, None, 1, 8, 8, 4, 4, 0, None, None, None, 1, 0, 0.2, 1.4, 0, 0, None, 0, None) == -1
    assert L.fmgan_torgb_f32(None, None, None, None, None, None, 1, 8, 5, 16, 1.0, None) == -1
    # split-K workspace query is pure host logic: tiny layer -> non-zero, big layer -> 0
    assert L.fmgan_modconv2d_workspace_bytes(8, 512, 512, 4, 4, 0) > 0
    assert L.fmgan_modconv2d_workspace_bytes(8, 32, 32, 1024, 1024, 0) == 0
    # empty batch is a no-op
    assert L.fmgan_upfirdn2d(0, None, None, None, 0, 8, 8, 1, 4, 4, 1, 1, 1, 1, 1, 1, 1, 1, -1, None) == 0


def test_prelu_backward_host_logic():
    """fmgan_prelu_backward_blocks / _f32: partial-sum rows per launch and argument checks, all before any HIP call."""
    L = _lib()
    assert L.fmgan_prelu_backward_blocks(0, 64) == 0 and L.fmgan_prelu_backward_blocks(10, 0) == 0
    for rows, c in ((1, 64), (1 << 20, 64), (1 << 20, 512), (300, 96), (7, 2048)):
        b = L.fmgan_prelu_backward_blocks(rows, c)
        q = 1
        while q < c // 4 and q < 256:
            q *= 2
        assert 1 <= b <= max(1, -(-rows // (256 // q))) and b <= 256 * 8        # never more blocks than row groups / 8 per CU
    assert L.fmgan_prelu_backward_f32(None, None, None, None, None, 0, 64, None) == 0          # empty: nothing to do
    assert L.fmgan_prelu_backward_f32(None, None, None, None, None, 16, 64, None) == -1        # null pointers
    assert L.fmgan_prelu_backward_f32(None, None, None, None, None, 16, 6, None) == -2         # channels % 4: unsupported
    assert L.fmgan_prelu_backward_f32(None, None, None, None, None, -1, 64, None) == -1


def test_fused_bias_act_backward_host_logic():
    """fmgan_fused_bias_act_bwd_blocks / _f32: which planes the one-pass backward serves, partial columns per plane,
    argument checks — all before any HIP call."""
    L = _lib()
    L.fmgan_fused_bias_act_bwd_blocks.argtypes = [ctypes.c_longlong, ctypes.c_int]
    for planes, hw, want in ((0, 64, 0), (8, 16, 0), (8, 63, 0), (8, 66, 0), (8, 64, 1), (8, 4096, 1), (8, 4100, 2),
                             (256, 1024 * 1024, 64), (1 << 31, 64, 0)):
        assert L.fmgan_fused_bias_act_bwd_blocks(planes, hw) == want, (planes, hw)
    L.fmgan_fused_bias_act_bwd_f32.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_longlong, ctypes.c_int, ctypes.c_float,
                                               ctypes.c_float, ctypes.c_void_p]
    assert L.fmgan_fused_bias_act_bwd_f32(None, None, None, None, 0, 64, 0.2, 1.4, None) == 0       # empty
    assert L.fmgan_fused_bias_act_bwd_f32(None, None, None, None, 4, 64, 0.2, 1.4, None) == -1      # null pointers
    assert L.fmgan_fused_bias_act_bwd_f32(None, None, None, None, -1, 64, 0.2, 1.4, None) == -1
    assert L.fmgan_fused_bias_act_bwd_f32(16, 16, 16, 16, 4, 30, 0.2, 1.4, None) == -2              # unserved plane size
    assert L.fmgan_fused_bias_act_bwd_f32(20, 16, 16, 16, 4, 64, 0.2, 1.4, None) == -2              # misaligned


def test_torgb_backward_host_logic():
    L = _lib()
    assert L.fmgan_torgb_backward_splits(8, 32, 1024 * 1024) >= 64          # enough blocks to fill the chip
    assert L.fmgan_torgb_backward_splits(2, 512, 16) == 1
    assert L.fmgan_torgb_backward_splits(2, 512, 18) == 0                   # H*W % 4 != 0: composite instead
    assert L.fmgan_torgb_backward_splits(0, 512, 16) == 0
    L.fmgan_torgb_backward_f32.argtypes = [ctypes.c_void_p] * 6 + [ctypes.c_int] * 4 + [ctypes.c_float, ctypes.c_void_p]
    assert L.fmgan_torgb_backward_f32(None, None, None, None, None, None, 0, 32, 3, 64, 1.0, None) == 0
    assert L.fmgan_torgb_backward_f32(None, None, None, None, None, None, 2, 32, 3, 64, 1.0, None) == -1
    assert L.fmgan_torgb_backward_f32(16, 16, 16, 16, 16, 16, 2, 32, 5, 64, 1.0, None) == -1     # cout > 4
    assert L.fmgan_torgb_backward_f32(16, 16, 16, 16, 16, 16, 2, 32, 3, 66, 1.0, None) == -2


def test_product_has_no_cpu_path():
    from op import upfirdn2d, fused_leaky_relu, FusedLeakyReLU
    x = torch.randn(1, 2, 8, 8)
    k = cases.make_fir('blur')
    with pytest.raises(RuntimeError, match='CUDA tensor'):
        upfirdn2d(x, k, pad=(1, 1))
    with pytest.raises(RuntimeError, match='CUDA tensor'):
        fused_leaky_relu(x, torch.zeros(2))
    with pytest.raises(RuntimeError, match='CUDA tensor'):
        FusedLeakyReLU(2)(x)
    import stylegan2
    g = stylegan2.Generator(16, 512, 1, generator_net_shape=[8, 8, 8, 8, 8, 8])
    with pytest.raises(RuntimeError, match='CUDA tensor'):
        g(None, latent_styles=[torch.randn(1, g.n_latent, 512)], input_is_latent=True,
          use_external_input_tensor=True, external_input_tensor=torch.randn(1, 8, 4, 4))


def test_raw_pointer_entry_points_refuse_other_dtypes():
    """The f32 entry points take raw pointers; a bf16 / f64 / CPU tensor must be refused on the host (RuntimeError), never
    handed to a kernel that would read past its end (round 3: a bf16 style vector under autocast faulted the GPU)."""
    from op import _native
    assert _native.fp(None) is None
    for t in (torch.zeros(4, dtype=torch.bfloat16), torch.zeros(4, dtype=torch.float64), torch.zeros(4)):
        with pytest.raises(RuntimeError):
            _native.fp(t)
    src = open(os.path.join(ROOT, '3d-fm-gan_amd', 'op', '_native.py')).read()
    for fn in ('modconv2d', 'modconv_demod', 'torgb', 'torgb_backward', 'noise_bias_act', 'blur_noise_bias_act',
               'modconv2d_rgb', 'modconv_wgrad', 'prelu_backward', 'fused_bias_act_backward'):
        body = re.search(r'^def ' + fn + r'\(.*?(?=^def |\Z)', src, re.S | re.M).group(0)
        assert ' ptr(' not in body.replace('ptr(wtb)', '').replace('ptr(wts)', '') and 'fp(' in body, fn


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, '3d-fm-gan_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.hip', '.h', '.cpp')):
                src = open(os.path.join(dirpath, f)).read()
                assert 'oracle' not in src.replace('# oracle', ''), f'{f} mentions the oracle'


def _check_manifest(module, man):
    sd = module.state_dict()
    assert list(sd.keys()) == sorted(sd.keys(), key=list(sd.keys()).index)  # trivial, keeps order explicit
    assert set(sd.keys()) == set(man.keys()), (set(sd.keys()) ^ set(man.keys()))
    for k, shape in man.items():
        assert list(sd[k].shape) == shape, k


def test_generator_state_dict_matches_reference(golden):
    import stylegan2
    man = golden.manifest('generator')
    for c in cases.GENERATOR_CASES:
        if c['size'] > 256:
            continue
        g = stylegan2.Generator(c['size'], 512, c['n_mlp'], generator_net_shape=c['shape'])
        _check_manifest(g, man[c['name']])
        assert g.n_latent == int(torch.log2(torch.tensor(float(c['size'])))) * 2 - 2
    g = stylegan2.Generator(256, 512, 8)
    assert len(g.state_dict()) == 135   # SURVEY.md §5 checkpoint row
    assert tuple(g.state_dict()['conv1.conv.weight'].shape) == (1, 512, 512, 3, 3)


def test_module_state_dicts_match_reference(golden):
    import stylegan2
    man = golden.manifest('modules')
    for c in cases.MODCONV_CASES:
        _check_manifest(stylegan2.ModulatedConv2d(c['cin'], c['cout'], c['k'], 512, demodulate=c['demod'],
                                                  upsample=c['up']), man[c['name']])
    for c in cases.STYLEDCONV_CASES:
        _check_manifest(stylegan2.StyledConv(c['cin'], c['cout'], 3, 512, upsample=c['up']), man[c['name']])
    for c in cases.TORGB_CASES:
        _check_manifest(stylegan2.ToRGB(c['cin'], 512, upsample=c['skip']), man[c['name']])


def test_discriminator_and_encoder_state_dicts_match_reference(golden):
    import types
    import stylegan2
    import resnet_encoder
    from psp_encoder_model.encoders import psp_encoders
    _check_manifest(stylegan2.Discriminator(64), golden.manifest('discriminator')['d64'])
    man = golden.manifest('encoders')
    _check_manifest(resnet_encoder.resnet18(tensor_encoding=True, tensor_transform=False), man['resnet'])
    _check_manifest(resnet_encoder.resnet18(tensor_encoding=False), man['resnet'])
    for n in (14, 18):
        opts = types.SimpleNamespace(input_nc=3, n_styles=n)
        _check_manifest(psp_encoders.GradualStyleEncoder(18, 'ir_se', opts), man[f'psp{n}'])


def test_encoders_match_golden_on_cpu(golden):
    """The encoders are plain PyTorch modules (no custom kernel), so they also run on CPU: check the product's
    modules against the reference's outputs here; the GPU run repeats this on MIOpen."""
    import types
    import numpy as np
    import resnet_encoder
    from psp_encoder_model.encoders import psp_encoders
    g = golden('e2e')
    name = 'e2e_256'
    p = synth.tensor(name + '/photo', (1, 3, 256, 256), dist='uniform')
    r = synth.tensor(name + '/render', (1, 3, 256, 256), dist='uniform')
    e_tsr = resnet_encoder.resnet18(tensor_encoding=True)
    e_w = resnet_encoder.resnet18(tensor_encoding=False)
    e_wp = psp_encoders.GradualStyleEncoder(18, 'ir_se', types.SimpleNamespace(input_nc=3, n_styles=14))
    for kind, m, seed in (('resnet', e_tsr, 5), ('resnet', e_w, 6), ('psp', e_wp, 7)):
        m.load_state_dict(synth.state_dict(kind, m.state_dict(), seed=seed))
        m.eval()
    with torch.no_grad():
        np.testing.assert_allclose(e_tsr(p).numpy(), g[name + '/e_tsr'], atol=2e-4, rtol=2e-4)
        np.testing.assert_allclose(e_w(r).numpy(), g[name + '/e_w'], atol=2e-4, rtol=2e-4)
        ref = g[name + '/e_wplus']
        np.testing.assert_allclose(e_wp(p).numpy(), ref, atol=2e-4 * np.abs(ref).max(), rtol=2e-4)


def test_network_shape_helpers():
    import stylegan2
    from Util import network_util
    shape = [16, 16, 16, 16, 16, 16, 8, 8, 8, 8]
    g = stylegan2.Generator(64, 512, 2, generator_net_shape=shape)
    assert network_util.Get_Network_Shape(g.state_dict()) == shape
    g2 = network_util.Build_Generator_From_Dict(g.state_dict(), size=64, n_mlp=2)
    assert [tuple(v.shape) for v in g2.state_dict().values()] == [tuple(v.shape) for v in g.state_dict().values()]


def test_training_losses_on_cpu_tensors():
    """Util/training_util.py mirrors the reference's GAN losses (training_util.py:24-58): closed forms on CPU tensors."""
    import math
    from Util import training_util as TU
    real, fake = torch.tensor([[0.5], [-1.0]]), torch.tensor([[2.0], [0.0]])
    sp = lambda v: math.log1p(math.exp(v))
    assert abs(TU.d_logistic_loss(real, fake).item() - ((sp(-0.5) + sp(1.0)) / 2 + (sp(2.0) + sp(0.0)) / 2)) < 1e-6
    assert abs(TU.g_nonsaturating_loss(fake).item() - (sp(-2.0) + sp(0.0)) / 2) < 1e-6
    x = torch.tensor([[1.0, 2.0], [3.0, 4.0]], requires_grad=True)
    pred = (x ** 2).sum(1, keepdim=True)                      # d/dx = 2x  ->  R1 = mean_b sum 4 x^2
    r1 = TU.d_r1_loss(pred, x)
    assert abs(r1.item() - (4 * (1 + 4) + 4 * (9 + 16)) / 2) < 1e-5
    r1.backward()                                             # second order: d R1 / dx = 8x / batch
    assert torch.allclose(x.grad, 4 * x.detach())
    lat = torch.ones(2, 3, 4, requires_grad=True)
    img = (lat.sum((1, 2))[:, None, None, None] * torch.ones(2, 1, 2, 2))
    pen, mean, pl = TU.g_path_regularize(img, lat, torch.tensor(0.0), probe=torch.ones_like(img))
    # probe / sqrt(4) = 0.5 per pixel, 4 pixels -> d/dlat = 2 everywhere; length = sqrt(mean_over_latents(sum_512 4))
    assert torch.allclose(pl, torch.full((2,), math.sqrt(4 * 4))) and abs(mean.item() - 0.04) < 1e-6
    assert abs(pen.item() - (4 - 0.04) ** 2) < 1e-4
    assert abs(TU.L1_Loss(torch.zeros(2, 3), torch.full((2, 3), -2.0)).item() - 2.0) < 1e-7


@pytest.mark.skipif(torch.cuda.is_available(), reason='passes fake pointers: host-side guard check for the CPU suite only')
def test_modconv_index_range_guards_refuse_without_launching():
    """Shapes just OUTSIDE each index-range guard of fmgan_modconv2d_f32 / fmgan_modconv_wgrad_f32 return
    FMGAN_EOVERFLOW before anything is launched (the pointers here are fake).  The same guards from the inside —
    shapes just below them must compute correctly — are GPU tests (tests/test_hip_modconv.py::test_guard_boundary_*)."""
    L = _lib()
    fake = ctypes.c_void_p(0x1000)
    EOVER, EINVAL = -4, -1

    def conv(batch, cin, cout, h, w, mode):
        return L.fmgan_modconv2d_f32(fake, fake, fake, None, fake, batch, cin, cout, h, w, mode, None, None, None, 1, 0,
                                     0.2, 1.4, 0, 0, None, 0, None)

    # per-tile input offsets are 32-bit: (samples per tile) * cin * h * w must stay below 2^31 elements
    assert conv(1, 8, 8, 16384, 16384, 0) == EOVER            # exactly 2^31
    assert conv(1, 8, 8, 16384, 16384, 1) == EOVER
    assert conv(1, 16, 8, 16384, 8192, 2) == EOVER
    assert conv(1, 2049, 8, 1024, 1024, 0) == EOVER           # 2^31 + 2^20
    # pixel indices y*ow + x are 32-bit
    assert conv(1, 1, 1, 46341, 46341, 0) == EOVER
    assert conv(1, 1, 1, 23171, 23171, 1) == EOVER            # output 46343^2
    # channel counts beyond the 32-bit weight offsets
    assert conv(1, (1 << 20) + 1, 8, 4, 4, 0) == EOVER
    assert conv(1, 8, (1 << 20) + 1, 4, 4, 0) == EOVER
    # whole-output element count
    assert conv(70000, 512, 512, 256, 256, 0) == EOVER
    # and the neighbouring failure classes stay distinct
    assert conv(1, 8, 8, 0, 16, 0) == EINVAL
    assert conv(1, 8, 8, 2, 2, 2) == EINVAL
    assert L.fmgan_modconv_wgrad_f32(fake, None, fake, fake, fake, 1, 8, 8, 46341, 46341, 1.0, fake, 1 << 30, None) == EOVER
    assert L.fmgan_modconv_wgrad_f32(fake, None, fake, fake, fake, 1, (1 << 20) + 1, 8, 32, 32, 1.0, fake, 1 << 30, None) == EOVER


@pytest.mark.parametrize('geom', [(3, 1, 1), (3, 2, 0), (1, 2, 0), (1, 1, 0)])
def test_conv_grad_family_matches_conv2d_autograd_at_every_order(geom):
    """op/conv_grad.py: conv2d as three Functions that differentiate into each other (host logic, library primitives —
    runs on CPU too).  First, second and third order agree with F.conv2d's own autograd on the Discriminator's four
    geometries (3x3 pad 1; 3x3 stride 2; 1x1 stride 2; 1x1), in float64."""
    from op import conv_grad
    k, stride, pad = geom
    torch.manual_seed(3)
    x = torch.randn(2, 3, 9, 8, dtype=torch.float64, requires_grad=True)
    w = torch.randn(4, 3, k, k, dtype=torch.float64, requires_grad=True)
    b = torch.randn(4, dtype=torch.float64, requires_grad=True)

    def ours(x_, w_, b_):
        return conv_grad.conv2d(x_, w_, b_, stride, pad)

    def ref(x_, w_, b_):
        return F.conv2d(x_, w_, b_, stride, pad)
    assert torch.autograd.gradcheck(ours, (x, w, b))
    assert torch.autograd.gradgradcheck(ours, (x, w, b))
    # R1-shaped use: penalty on the input gradient, differentiated w.r.t. the weight — and once more
    outs = []
    for f in (ours, ref):
        y = f(x, w, b)
        gx, = torch.autograd.grad(y.pow(2).sum(), x, create_graph=True)
        pen = gx.pow(2).sum()
        gw, = torch.autograd.grad(pen, w, create_graph=True)
        g3, = torch.autograd.grad(gw.pow(2).sum(), x)
        outs.append((y.detach(), gx.detach(), gw.detach(), g3))
    for a, c in zip(*outs):
        torch.testing.assert_close(a, c, rtol=1e-10, atol=1e-10)


def test_live_weights_deepcopy_starts_without_table():
    """copy.deepcopy(G) (how Trainer builds g_ema): the copy's LiveWeights must not inherit the original's device table or
    pointer key (raw pointers of the ORIGINAL's parameters and buffers); it is bound to the copied network."""
    import copy
    import stylegan2
    from op.live_weights import LiveWeights
    G = stylegan2.Generator(8, 16, 1, generator_net_shape=[8, 8, 8])
    lw = LiveWeights(G)
    lw._key, lw._table, lw._n, lw._blocks, lw._buffers = ('stale',), torch.zeros(4), 3, 7, [torch.zeros(1)]
    G._live_weights = lw
    G2 = copy.deepcopy(G)
    lw2 = G2._live_weights
    assert lw2 is not lw and lw2.root is G2
    assert lw2._key is None and lw2._table is None and lw2._buffers == [] and not lw2.active
    assert lw._key == ('stale',)


def test_winograd_selection_rule_is_host_logic():
    """stylegan2.winograd_pays: which plain StyledConv layers take the Winograd F(2x2,3x3) form — wide layers of
    16^2..128^2 with enough tiles, fp32 precision only, even sizes only."""
    import stylegan2
    from op import _native
    assert stylegan2.winograd_pays(8, 512, 512, 64, 64) and stylegan2.winograd_pays(8, 256, 256, 128, 128)
    assert stylegan2.winograd_pays(8, 512, 512, 16, 16) and stylegan2.winograd_pays(32, 512, 512, 16, 16)
    assert not stylegan2.winograd_pays(8, 128, 128, 256, 256)       # transforms cost more than the saved MACs
    assert not stylegan2.winograd_pays(8, 64, 64, 512, 512) and not stylegan2.winograd_pays(8, 512, 512, 8, 8)
    assert not stylegan2.winograd_pays(1, 512, 512, 16, 16)         # 64 tiles: too few columns for the 16 GEMMs
    assert not stylegan2.winograd_pays(8, 512, 512, 63, 64) and not stylegan2.winograd_pays(8, 512, 256, 64, 65)
    with _native.modconv_precision('bf16x3'):
        assert not stylegan2.winograd_pays(8, 512, 512, 64, 64)
    old = stylegan2.WINOGRAD
    try:
        stylegan2.WINOGRAD = False
        assert not stylegan2.winograd_pays(8, 512, 512, 64, 64)
    finally:
        stylegan2.WINOGRAD = old
