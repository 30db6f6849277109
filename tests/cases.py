"""Case tables shared by tools/make_golden.py (reference side) and the tests (oracle / HIP side)."""
import torch

import synth


def make_fir(spec):
    """FIR taps for a case: 'blur' = make_kernel([1,3,3,1]) (stylegan2.py:36-44), 'blur4' = that * 4
    (Upsample / Blur(upsample_factor=2), stylegan2.py:52,98-99), or ('rand', kh, kw, seed): an ASYMMETRIC
    kernel — [1,3,3,1] is symmetric and would hide a missing flip."""
    if spec in ('blur', 'blur4'):
        k = torch.tensor([1., 3., 3., 1.])
        k = k[None, :] * k[:, None]
        k = k / k.sum()
        return k * 4 if spec == 'blur4' else k
    _, kh, kw, seed = spec
    return synth.tensor(f'fir/{kh}x{kw}/{seed}', (kh, kw), seed=seed)


def _u(name, shape, kernel, up, down, pad, grad=False):
    return dict(name=name, shape=shape, kernel=kernel, up=up, down=down, pad=pad, grad=grad)


UPFIRDN2D_CASES = [
    # G-a: blur after the transposed conv, [.,2H+1,2W+1] -> [.,2H,2W]  (stylegan2.py:216-222, 279)
    _u('ga_9', (2, 3, 9, 9), 'blur4', 1, 1, (1, 1), grad=True),
    _u('ga_17', (1, 32, 17, 17), 'blur4', 1, 1, (1, 1)),
    _u('ga_33', (2, 3, 33, 33), 'blur4', 1, 1, (1, 1)),
    _u('ga_65', (1, 2, 65, 65), 'blur4', 1, 1, (1, 1), grad=True),
    _u('ga_129', (1, 2, 129, 129), 'blur4', 1, 1, (1, 1)),
    _u('ga_257_rand', (1, 1, 257, 257), ('rand', 4, 4, 1), 1, 1, (1, 1)),
    _u('ga_65_rand', (2, 3, 65, 65), ('rand', 4, 4, 2), 1, 1, (1, 1), grad=True),
    # G-b: ToRGB skip upsample, up=2 pad=(2,1)  (stylegan2.py:47-65, 401)
    _u('gb_4', (2, 3, 4, 4), 'blur4', 2, 1, (2, 1), grad=True),
    _u('gb_7', (1, 3, 7, 7), 'blur4', 2, 1, (2, 1)),
    _u('gb_16', (2, 3, 16, 16), 'blur4', 2, 1, (2, 1)),
    _u('gb_33_rand', (1, 3, 33, 33), ('rand', 4, 4, 3), 2, 1, (2, 1), grad=True),
    _u('gb_64', (1, 3, 64, 64), 'blur4', 2, 1, (2, 1)),
    # D-a / D-b: blur before the stride-2 convs  (stylegan2.py:705-711, 746-750)
    _u('da_16', (2, 4, 16, 16), 'blur', 1, 1, (2, 2), grad=True),
    _u('da_64', (1, 3, 64, 64), 'blur', 1, 1, (2, 2)),
    _u('db_16', (2, 4, 16, 16), 'blur', 1, 1, (1, 1)),
    _u('db_128_rand', (1, 2, 128, 128), ('rand', 4, 4, 4), 1, 1, (1, 1)),
    # backward configurations (op/upfirdn2d.py:120-123): grad of G-a is pad (2,2); grad of G-b is down=2 pad (1,1)
    _u('bwd_ga_16', (1, 3, 16, 16), 'blur4', 1, 1, (2, 2)),
    _u('bwd_ga_64', (1, 2, 64, 64), ('rand', 4, 4, 5), 1, 1, (2, 2)),
    _u('bwd_gb_8', (2, 3, 8, 8), 'blur4', 1, 2, (1, 1), grad=True),
    _u('bwd_gb_14', (1, 3, 14, 14), ('rand', 4, 4, 6), 1, 2, (1, 1)),
    _u('bwd_gb_66', (1, 2, 66, 66), 'blur4', 1, 2, (1, 1)),
    # Downsample module (defined, unused on the path; stylegan2.py:68-86): down=2 pad (1,1) == bwd_gb
    # generic corners: crop (negative pad), other kernel sizes, up/down 3, non-square taps, 1x1
    _u('crop', (1, 2, 12, 12), ('rand', 4, 4, 7), 1, 1, (-1, 2)),
    _u('crop_up2', (1, 2, 9, 9), ('rand', 4, 4, 8), 2, 1, (-2, 1)),
    _u('k3', (1, 2, 10, 10), ('rand', 3, 3, 9), 1, 1, (1, 1), grad=True),
    _u('k2_up2', (1, 2, 6, 6), ('rand', 2, 2, 10), 2, 1, (1, 0)),
    _u('k2_down2', (1, 2, 8, 8), ('rand', 2, 2, 11), 1, 2, (0, 0)),
    _u('k2x4', (1, 2, 9, 9), ('rand', 2, 4, 12), 1, 1, (2, 1)),
    _u('k1', (1, 2, 5, 5), ('rand', 1, 1, 13), 1, 1, (0, 0)),
    _u('up3', (1, 1, 5, 5), ('rand', 4, 4, 14), 3, 1, (2, 2)),
    _u('down3', (1, 1, 13, 13), ('rand', 4, 4, 15), 1, 3, (1, 1)),
    _u('up2_down2', (1, 2, 9, 9), ('rand', 4, 4, 16), 2, 2, (1, 2), grad=True),
    _u('k3_128', (1, 1, 130, 130), ('rand', 3, 3, 17), 1, 1, (1, 1)),
]

FUSED_ACT_CASES = [
    dict(name='act4d_bias', shape=(2, 5, 6, 8), bias=True),
    dict(name='act4d_nobias', shape=(2, 5, 6, 8), bias=False),
    dict(name='act4d_odd', shape=(3, 7, 5, 3), bias=True),
    dict(name='act2d_bias', shape=(3, 8), bias=True),
    dict(name='act2d_512', shape=(4, 512), bias=True),
    dict(name='act3d_bias', shape=(2, 4, 9), bias=True),
    dict(name='act4d_big', shape=(1, 3, 64, 64), bias=True),
]


def fused_act_inputs(c):
    """x straddles 0 and contains exact zeros AFTER the bias add (`x > 0` is strict,
    op/fused_bias_act_kernel.cu:42): element 0 of every channel is set to -bias."""
    x = synth.tensor(c['name'] + '/x', c['shape'])
    b = synth.tensor(c['name'] + '/b', (c['shape'][1],)) if c['bias'] else None
    xf = x.reshape(c['shape'][0], c['shape'][1], -1).clone()
    xf[:, :, 0] = 0.0 if b is None else -b.view(1, -1)
    return xf.reshape(c['shape']).contiguous(), b


def _m(name, cin, cout, k, up, demod, b, h):
    return dict(name=name, cin=cin, cout=cout, k=k, up=up, demod=demod, b=b, h=h)


MODCONV_CASES = [
    _m('mc_plain', 16, 32, 3, False, True, 2, 8),
    _m('mc_up', 16, 32, 3, True, True, 2, 8),
    _m('mc_rgb', 32, 3, 1, False, False, 2, 8),
    _m('mc_plain_odd', 6, 10, 3, False, True, 3, 5),
    _m('mc_up_odd', 6, 10, 3, True, True, 3, 5),
    _m('mc_plain_4', 24, 40, 3, False, True, 3, 4),
    _m('mc_up_4', 24, 40, 3, True, True, 3, 4),
    _m('mc_plain_wide', 20, 136, 3, False, True, 1, 16),
    _m('mc_up_wide', 20, 136, 3, True, True, 1, 16),
    _m('mc_plain_64', 9, 70, 3, False, True, 1, 40),
    _m('mc_up_64', 9, 70, 3, True, True, 1, 36),
    _m('mc_nodemod', 8, 8, 3, False, False, 2, 8),
]

STYLEDCONV_CASES = [
    dict(name='sc_plain', cin=16, cout=16, up=False, b=2, h=8, nb=2),
    dict(name='sc_plain_bcast', cin=16, cout=24, up=False, b=3, h=16, nb=1),
    dict(name='sc_up', cin=16, cout=8, up=True, b=2, h=8, nb=1),
]

TORGB_CASES = [
    dict(name='rgb_noskip', cin=16, b=2, h=4, skip=False),
    dict(name='rgb_skip', cin=16, b=2, h=8, skip=True),
    dict(name='rgb_skip_32', cin=8, b=1, h=32, skip=True),
]

GENERATOR_CASES = [
    dict(name='g64_narrow', size=64, n_mlp=2, shape=[16, 16, 16, 16, 16, 16, 8, 8, 8, 8], b=2, mode='latent', stride=1),
    dict(name='g64_narrow_z', size=64, n_mlp=2, shape=[16, 16, 16, 16, 16, 16, 8, 8, 8, 8], b=2, mode='z', stride=1),
    dict(name='g256_narrow', size=256, n_mlp=2, shape=[32] * 6 + [16] * 4 + [8] * 4, b=1, mode='latent', stride=4),
    dict(name='g256_full', size=256, n_mlp=8, shape=None, b=1, mode='latent', stride=8),
    dict(name='g1024_full', size=1024, n_mlp=8, shape=None, b=1, mode='latent', stride=32),
]

E2E_CASES = [
    dict(name='e2e_256', size=256, b=1, tsr_encode='Photo Image', sliced_layer=None, use_tanh=False, stride=8),
    dict(name='e2e_256_render_tanh', size=256, b=2, tsr_encode='Render Image', sliced_layer=list(range(4, 14)),
         use_tanh=True, stride=8),
    dict(name='e2e_1024', size=1024, b=1, tsr_encode='Photo Image', sliced_layer=None, use_tanh=False, stride=32),
]

# BASELINE config 3 (full 3-encoder forward + backward @256^2): L1 loss to a synthetic target, gradients of EVERY
# trainable parameter of E_Tsr / E_W / E_W_Plus / G, stored as a strided sample + the L2 norm of each tensor.
E2E_GRAD_CASE = dict(name='e2e_256_grad', size=256, b=2, tsr_encode='Photo Image', sliced_layer=None, use_tanh=False,
                     stride=8)
GRAD_SAMPLES = 48


def grad_sample(t):
    """(strided sample of <= GRAD_SAMPLES elements, L2 norm in float64) of a gradient tensor."""
    flat = t.detach().reshape(-1)
    step = max(1, flat.numel() // GRAD_SAMPLES)
    return flat[::step][:GRAD_SAMPLES].cpu().numpy(), float(flat.double().pow(2).sum().sqrt())


# One training iteration's four gradient computations (train_3_encoder.py:448-596) at a size the CPU reference
# finishes in a minute: Generator(64) + Discriminator(64), encoders on 256^2 (they are shape-locked to it).
TRAIN_STEP_CASE = dict(name='train_64', size=64, b=4, ppl_idx=[0, 2])
# BASELINE config 5's networks (Generator(1024) + Discriminator(1024), 18 styles) at the batch the CPU reference still
# finishes in minutes, fp32 and fp64; path length on one sample (batch / path_reg_batch_shrink).
TRAIN_STEP_1024_CASE = dict(name='train_1024', size=1024, b=2, ppl_idx=[1])
# identity term of the G step: ArcFace features of grey, pooled images and the two loss forms (training_util.py:148-205)
FACE_ID_CASE = dict(name='face_id', b=2, size=256)
# train_3_encoder_hyperparams.py:53-63
TRAIN_HP = dict(lr=0.001, r1=10, d_reg_every=16, g_reg_every=4, path_reg_weight=2, path_reg_batch_shrink=2,
                l1_loss_lambda=3)

DISCRIMINATOR_CASES = [
    dict(name='d64', size=64, b=4),
    dict(name='d256', size=256, b=2),
]

TENSOR2IM_CASES = [
    dict(name='t2i_64', shape=(2, 3, 64, 64)),
    dict(name='t2i_ragged', shape=(1, 3, 19, 23)),
]


def tensor2im_input(c):
    """Values inside and outside [-1,1] plus the edge values of the clip and of the uint8 truncation."""
    x = synth.tensor(c['name'] + '/x', c['shape'], scale=0.8)
    x.view(-1)[:8] = torch.tensor([-1.0, 1.0, -1.5, 1.5, 0.0, 0.999999, -0.999999, 0.00392])
    return x

# (name, in_h, in_w, out_h, out_w): PIL.Image.resize(BILINEAR) cases — shrink (antialiased), enlarge, identity axes,
# odd sizes, extreme aspect, 1-pixel outputs
RESIZE_CASES = [
    ('rz_quarter', 64, 64, 16, 16), ('rz_odd_down', 37, 53, 11, 20), ('rz_up', 17, 5, 40, 40),
    ('rz_mixed', 33, 77, 33, 20), ('rz_wide', 50, 50, 7, 91), ('rz_third', 99, 66, 33, 22),
    ('rz_one', 5, 5, 1, 1), ('rz_tiny_up', 2, 3, 9, 8), ('rz_512_smooth', 128, 128, 32, 32),
    ('rz_frac_smooth', 100, 60, 64, 38),
]
