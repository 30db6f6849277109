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
# Util/training_util.py imports Util.landmark_util, which imports `face_alignment` from a hard-coded home path
# (train_3_encoder.py:39-41; absent third-party dependency): stubbed like torchvision; none of the loss functions used
# here touch it.
_lm = types.ModuleType('Util.landmark_util'); _lm.Get_HeatMap_PyTorch = None
sys.modules['Util.landmark_util'] = _lm
import Util.training_util as ref_training_util  # noqa: E402

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


def _sample_grads(out, prefix, named_params, suffix=''):
    """Strided sample + L2 norm of every parameter's .grad (tests/cases.py::grad_sample); unused parameters
    (grad None: the mapping network and constant input under input_is_latent / external tensor) are recorded as absent."""
    for name, p in named_params:
        if p.grad is None:
            continue
        s, n = cases.grad_sample(p.grad)
        out[f'{prefix}/{name}/s{suffix}'] = s
        out[f'{prefix}/{name}/n{suffix}'] = np.float64(n)


def gen_e2e_grad():
    """BASELINE config 3: (photo, render) -> image -> L1 loss (Util/training_util.py:103-113) -> backward through the
    Generator and the three encoders (train_3_encoder.py:495-558 with only the L1 term), in fp32 (the reference's
    precision) and in fp64 (the yardstick both fp32 implementations are measured against)."""
    c = cases.E2E_GRAD_CASE
    out = {}
    n_latent = int(np.log2(c['size'])) * 2 - 2
    for dt, sfx in ((torch.float32, ''), (torch.float64, '64')):
        e_tsr, e_w, e_wp = build_encoders(n_latent)
        g = stylegan2.Generator(c['size'], 512, 8)
        g.load_state_dict(synth.state_dict('generator', g.state_dict(), seed=4))
        nets = dict(e_tsr=e_tsr, e_w=e_w, e_wp=e_wp, g=g)
        for m in nets.values():
            m.to(dt)
        g.eval()
        p = synth.tensor(c['name'] + '/photo', (c['b'], 3, 256, 256), dist='uniform').to(dt)
        r = synth.tensor(c['name'] + '/render', (c['b'], 3, 256, 256), dist='uniform').to(dt)
        target = synth.tensor(c['name'] + '/target', (c['b'], 3, c['size'], c['size']), dist='uniform').to(dt)
        img = network_util.Forward_Inference_3_Encoder(p, r, e_tsr, e_w, e_wp, _GWrap(g), tsr_encode=c['tsr_encode'],
                                                       sliced_layer=c['sliced_layer'], use_tanh=c['use_tanh'])
        loss = ref_training_util.L1_Loss(img, target)
        loss.backward()
        out['img/sub' + sfx] = subsample(img, c['stride'])
        out['loss' + sfx] = np.float64(loss.item())
        for k, m in nets.items():
            _sample_grads(out, k, m.named_parameters(), sfx)
        print('  e2e_grad', dt, loss.item())
    np.savez_compressed(os.path.join(OUT, 'e2e_grad.npz'), **out)
    print('e2e_grad', len(out))


def gen_e2e_grad_latents():
    """The encoder outputs of the e2e_grad case (reference modules, fp32 and fp64): with them a test can run the
    Generator's forward + backward ALONE on exactly the inputs the reference's Generator saw, so that only this repo's
    own bit-reproducible kernels stand between it and the `g/*` gradients of e2e_grad.npz (no MIOpen kernel involved)."""
    c = cases.E2E_GRAD_CASE
    out = {}
    n_latent = int(np.log2(c['size'])) * 2 - 2
    with torch.no_grad():
        for dt, sfx in ((torch.float32, ''), (torch.float64, '64')):
            e_tsr, e_w, e_wp = build_encoders(n_latent)
            for m in (e_tsr, e_w, e_wp):
                m.to(dt)
            p = synth.tensor(c['name'] + '/photo', (c['b'], 3, 256, 256), dist='uniform').to(dt)
            r = synth.tensor(c['name'] + '/render', (c['b'], 3, 256, 256), dist='uniform').to(dt)
            out['e_tsr' + sfx] = npy(e_tsr(p if c['tsr_encode'] == 'Photo Image' else r))
            out['e_w' + sfx] = npy(e_w(r))
            out['e_wplus' + sfx] = npy(e_wp(p))
    np.savez_compressed(os.path.join(OUT, 'e2e_grad_latents.npz'), **out)
    print('e2e_grad_latents', {k: v.shape for k, v in out.items()})


def gen_fp64():
    """The reference modules in float64 on the same inputs/weights: the yardstick for 'how far is an fp32 result from
    the exact value'.  tests compare |HIP - fp64| with |reference fp32 (the golden files above) - fp64|."""
    out = {}
    with torch.no_grad():
        for c in cases.GENERATOR_CASES:
            if c['mode'] != 'latent':
                continue
            g = stylegan2.Generator(c['size'], 512, c['n_mlp'], generator_net_shape=c['shape'])
            g.load_state_dict(synth.state_dict('generator', g.state_dict(), seed=4))
            g.double().eval()
            cin0 = c['shape'][0] if c['shape'] else 512
            lat = synth.tensor(c['name'] + '/latent', (c['b'], g.n_latent, 512)).double()
            tsr = synth.tensor(c['name'] + '/tsr', (c['b'], cin0, 4, 4)).double()
            img = g(None, latent_styles=[lat], input_is_latent=True, use_external_input_tensor=True,
                    external_input_tensor=tsr, randomize_noise=False)
            out[c['name'] + '/sub'] = subsample(img, c['stride'])
            print(' ', c['name'])
        for c in cases.E2E_CASES:
            n_latent = int(np.log2(c['size'])) * 2 - 2
            e_tsr, e_w, e_wp = build_encoders(n_latent)
            for m in (e_tsr, e_w, e_wp):
                m.double()
            p = synth.tensor(c['name'] + '/photo', (c['b'], 3, 256, 256), dist='uniform').double()
            r = synth.tensor(c['name'] + '/render', (c['b'], 3, 256, 256), dist='uniform').double()
            out[c['name'] + '/e_tsr'] = npy(e_tsr(p))
            out[c['name'] + '/e_w'] = npy(e_w(r))
            out[c['name'] + '/e_wplus'] = npy(e_wp(p))
            g = stylegan2.Generator(c['size'], 512, 8)
            g.load_state_dict(synth.state_dict('generator', g.state_dict(), seed=4))
            g.double().eval()
            img = network_util.Forward_Inference_3_Encoder(p, r, e_tsr, e_w, e_wp, _GWrap(g), tsr_encode=c['tsr_encode'],
                                                           sliced_layer=c['sliced_layer'], use_tanh=c['use_tanh'])
            out[c['name'] + '/sub'] = subsample(img, c['stride'])
            print(' ', c['name'])
        for c in cases.DISCRIMINATOR_CASES:
            d = stylegan2.Discriminator(c['size'])
            d.load_state_dict(synth.state_dict('discriminator', d.state_dict(), seed=8))
            d.double()
            x = synth.tensor(c['name'] + '/x', (c['b'], 3, c['size'], c['size']), dist='uniform').double()
            out[c['name'] + '/out'] = npy(d(x))
            print(' ', c['name'])
    np.savez_compressed(os.path.join(OUT, 'fp64.npz'), **out)
    print('fp64', len(out))


class _FixedProbe:
    """Generator.forward draws its path-length probe with torch.randn_like (stylegan2.py:684); pin it to a synth tensor
    for the duration of one call so both sides use the same probe."""

    def __init__(self, probe):
        self.probe = probe

    def __enter__(self):
        self.orig = torch.randn_like
        torch.randn_like = lambda t, **kw: self.probe.to(dtype=t.dtype, device=t.device)

    def __exit__(self, *exc):
        torch.randn_like = self.orig


def gen_face_id():
    """The identity term of the G step with the reference's ArcFace topology (Util/arcface_pytorch/
    resnet_face_recognition.py:170-238, resnet_face18(use_se=False)) and its wrappers (Util/training_util.py:131-205):
    features of the converted image, both loss forms, and the gradient of the MSE form w.r.t. the output image."""
    from Util.arcface_pytorch.resnet_face_recognition import resnet_face18
    c = cases.FACE_ID_CASE
    out = {}
    m = resnet_face18(use_se=False)
    sd = synth.state_dict('arcface', m.state_dict(), seed=9)
    m.load_state_dict(sd)
    m.eval().requires_grad_(False)
    json.dump(synth.manifest(sd), open(os.path.join(OUT, 'arcface_manifest.json'), 'w'), indent=0, sort_keys=True)
    a = synth.tensor(c['name'] + '/a', (c['b'], 3, c['size'], c['size']), dist='uniform').requires_grad_(True)
    b = synth.tensor(c['name'] + '/b', (c['b'], 3, c['size'], c['size']), dist='uniform')
    conv = ref_training_util.Convert_Tensor_For_Face_Recognition_Loss(a)
    out['converted'] = npy(conv)
    out['features'] = npy(m(conv))
    mse = ref_training_util.Face_Identity_Loss(a, b, m, 'MSE')
    cos = ref_training_util.Face_Identity_Loss(a, b, m, 'CosineSimilarity')
    ga, = torch.autograd.grad(mse, a)
    out['mse'], out['cos'] = np.float64(mse.item()), np.float64(cos.item())
    out['grad_a/sub'] = subsample(ga, 8)
    # LPIPS_Loss is a batch mean over whatever module it is given (the reference's lpips package itself cannot be
    # imported offline: torchvision / skimage / IPython): pinned with a stand-in distance
    out['lpips_wrapper'] = np.float64(ref_training_util.LPIPS_Loss(a, b, lambda x, y: (x - y).abs().mean([1, 2, 3])).item())
    np.savez_compressed(os.path.join(OUT, 'face_id.npz'), **out)
    print('face_id', len(out), mse.item(), cos.item())


def gen_train_step(c=None, fname='train_step.npz'):
    """The four gradient computations of one training iteration (train_3_encoder.py:448-596) with the reference's
    modules and the reference's own loss functions (Util/training_util.py), all evaluated at the same weights:
      d:   D_Loss_BackProp   d_logistic_loss(D(ref), D(fake)),  fake from frozen G/encoders           (:448-477)
      r1:  D_Reg_BackProp    r1/2 * d_r1_loss * d_reg_every + 0 * real_pred[0]                        (:479-493)
      g:   G_Loss_BackProp   g_nonsaturating_loss(D(fake)) + l1_lambda * L1_Loss  (LPIPS / ArcFace / landmark terms
           need pretrained third-party weights that are not available offline)             (:495-558)
      ppl: G_Reg_BackProp    path-length penalty on batch/2 through PPL_regularize=True               (:561-596)
    Stored: the loss values and a strided sample + norm of every parameter gradient, fp32 and fp64."""
    c = c or cases.TRAIN_STEP_CASE
    hp = cases.TRAIN_HP
    out = {}
    size, b = c['size'], c['b']
    n_latent = int(np.log2(size)) * 2 - 2
    for dt, sfx in ((torch.float32, ''), (torch.float64, '64')):
        e_tsr, e_w, e_wp = build_encoders(n_latent)
        g = stylegan2.Generator(size, 512, 2)
        g.load_state_dict(synth.state_dict('generator', g.state_dict(), seed=4))
        d = stylegan2.Discriminator(size)
        d.load_state_dict(synth.state_dict('discriminator', d.state_dict(), seed=8))
        ge = dict(g=g, e_tsr=e_tsr, e_w=e_w, e_wp=e_wp)
        for m in list(ge.values()) + [d]:
            m.to(dt)
        wrap = _GWrap(g)
        photo = synth.tensor(c['name'] + '/photo', (b, 3, 256, 256), dist='uniform').to(dt)
        render = synth.tensor(c['name'] + '/render', (b, 3, 256, 256), dist='uniform').to(dt)
        ref = synth.tensor(c['name'] + '/ref', (b, 3, size, size), dist='uniform').to(dt)

        def fwd(p_, r_, **kw):
            return network_util.Forward_Inference_3_Encoder(p_, r_, e_tsr, e_w, e_wp, wrap, 'Photo Image', None, False, **kw)

        def zero():
            for m in list(ge.values()) + [d]:
                m.zero_grad(set_to_none=True)

        def req(ge_flag, d_flag):
            for m in ge.values():
                m.requires_grad_(ge_flag)
            d.requires_grad_(d_flag)

        # ---- d
        req(False, True); zero()
        fake = fwd(photo, render)
        out_pred, ref_pred = d(fake), d(ref)
        d_loss = ref_training_util.d_logistic_loss(ref_pred, out_pred)
        d_loss.backward()
        out['d/loss' + sfx] = np.float64(d_loss.item())
        out['d/ref_score' + sfx] = np.float64(ref_pred.mean().item())
        out['d/out_score' + sfx] = np.float64(out_pred.mean().item())
        _sample_grads(out, 'd/d', d.named_parameters(), sfx)
        # ---- r1
        zero()
        real = ref.clone().requires_grad_(True)
        real_pred = d(real)
        r1 = ref_training_util.d_r1_loss(real_pred, real)
        (hp['r1'] / 2 * r1 * hp['d_reg_every'] + 0 * real_pred[0]).backward()
        out['r1/loss' + sfx] = np.float64(r1.item())
        _sample_grads(out, 'r1/d', d.named_parameters(), sfx)
        # ---- g
        req(True, False); zero()
        fake = fwd(photo, render)
        g_loss = ref_training_util.g_nonsaturating_loss(d(fake))
        l1 = hp['l1_loss_lambda'] * ref_training_util.L1_Loss(fake, ref)
        (g_loss + l1).backward()
        out['g/loss' + sfx] = np.float64(g_loss.item())
        out['g/l1' + sfx] = np.float64(l1.item())
        for k, m in ge.items():
            _sample_grads(out, 'g/' + k, m.named_parameters(), sfx)
        # ---- ppl (batch / path_reg_batch_shrink; the reference picks the samples at random: fixed here)
        zero()
        idx = c['ppl_idx']
        probe = synth.tensor(c['name'] + '/probe', (len(idx), 3, size, size)).to(dt)
        with _FixedProbe(probe):
            img, path_lengths = fwd(photo[idx], render[idx], PPL_regularize=True)
        mean_path_length = 0
        path_mean = mean_path_length + 0.01 * (path_lengths.mean() - mean_path_length)
        path_loss = (path_lengths - path_mean).pow(2).mean()
        weighted = hp['path_reg_weight'] * hp['g_reg_every'] * path_loss
        weighted = weighted + 0 * img[0, 0, 0, 0]
        weighted.backward()
        out['ppl/loss' + sfx] = np.float64(path_loss.item())
        out['ppl/lengths' + sfx] = npy(path_lengths)
        for k, m in ge.items():
            _sample_grads(out, 'ppl/' + k, m.named_parameters(), sfx)
        print('  train_step', c['name'], dt, d_loss.item(), r1.item(), g_loss.item(), l1.item(), path_loss.item(), flush=True)
        del fake, out_pred, ref_pred, d_loss, real, real_pred, r1, g_loss, l1, img, path_lengths, weighted
    np.savez_compressed(os.path.join(OUT, fname), **out)
    print(fname, len(out))


def gen_train_step_1024():
    """The same four computations at BASELINE config 5's sizes: Generator(1024), Discriminator(1024), 18 styles, B=2."""
    gen_train_step(cases.TRAIN_STEP_1024_CASE, 'train_step_1024.npz')


if __name__ == '__main__':
    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(0)
    which = sys.argv[1:] or ['upfirdn2d', 'fused_act', 'modules', 'generator', 'e2e', 'discriminator', 'image_io',
                             'e2e_grad', 'e2e_grad_latents', 'fp64', 'train_step', 'face_id', 'train_step_1024']
    for w in which:
        {'upfirdn2d': gen_upfirdn2d, 'fused_act': gen_fused_act, 'modules': gen_modules,
         'generator': gen_generator, 'e2e': gen_encoders_e2e, 'discriminator': gen_discriminator,
         'image_io': gen_image_io, 'e2e_grad': gen_e2e_grad, 'fp64': gen_fp64, 'train_step': gen_train_step,
         'train_step_1024': gen_train_step_1024, 'face_id': gen_face_id,
         'e2e_grad_latents': gen_e2e_grad_latents}[w]()
