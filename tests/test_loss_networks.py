"""The frozen loss networks of the G step (BASELINE config 5): ArcFace identity term pinned to the reference's module and
wrappers (fixture: tools/make_golden.py::gen_face_id), LPIPS topology checked structurally (the reference's lpips package
cannot be imported offline — torchvision / skimage / IPython — so it is load, not a parity row; tests say so)."""
import numpy as np
import pytest
import torch

import cases
import synth


def _face_model(device='cpu'):
    from Util.arcface_pytorch.resnet_face_recognition import resnet_face18
    m = resnet_face18(use_se=False)
    m.load_state_dict(synth.state_dict('arcface', m.state_dict(), seed=9))
    return m.eval().requires_grad_(False).to(device)


def _check_face_id(g, device, tol):
    from Util import training_util as TU
    c = cases.FACE_ID_CASE
    m = _face_model(device)
    a = synth.tensor(c['name'] + '/a', (c['b'], 3, c['size'], c['size']), dist='uniform').to(device).requires_grad_(True)
    b = synth.tensor(c['name'] + '/b', (c['b'], 3, c['size'], c['size']), dist='uniform').to(device)
    conv = TU.Convert_Tensor_For_Face_Recognition_Loss(a)
    np.testing.assert_allclose(conv.detach().cpu().numpy(), g['converted'], atol=1e-6, rtol=1e-6)
    feat = m(conv)
    np.testing.assert_allclose(feat.detach().cpu().numpy(), g['features'], atol=tol * float(np.abs(g['features']).max()), rtol=0)
    mse = TU.Face_Identity_Loss(a, b, m, 'MSE')
    cos = TU.Face_Identity_Loss(a, b, m, 'CosineSimilarity')
    np.testing.assert_allclose(mse.item(), float(g['mse']), rtol=20 * tol)
    np.testing.assert_allclose(cos.item(), float(g['cos']), rtol=20 * tol, atol=1e-6)
    ga, = torch.autograd.grad(mse, a)
    ref = g['grad_a/sub']
    np.testing.assert_allclose(ga.detach().cpu().numpy()[..., ::8, ::8], ref, atol=20 * tol * float(np.abs(ref).max()), rtol=0)
    lp = TU.LPIPS_Loss(a, b, lambda x, y: (x - y).abs().mean([1, 2, 3]))
    np.testing.assert_allclose(lp.item(), float(g['lpips_wrapper']), rtol=1e-5)


def test_face_identity_loss_matches_reference_on_cpu(golden):
    _check_face_id(golden('face_id'), 'cpu', 2e-6)


def test_arcface_state_dict_matches_reference(golden):
    from Util.arcface_pytorch.resnet_face_recognition import resnet_face18
    sd = resnet_face18(use_se=False).state_dict()
    assert {k: list(v.shape) for k, v in sd.items()} == golden.manifest('arcface')


def test_face_conversion_pools_to_128_at_any_training_size():
    from Util import training_util as TU
    for size in (256, 1024):
        x = synth.tensor(f'fc/{size}', (1, 3, size, size), dist='uniform')
        y = TU.Convert_Tensor_For_Face_Recognition_Loss(x)
        assert tuple(y.shape) == (1, 1, 128, 128)
        k = size // 128
        ref = TU.RGB_to_GrayScale(x).reshape(1, 1, 128, k, 128, k).mean((3, 5))
        torch.testing.assert_close(y, ref, atol=1e-6, rtol=1e-5)


def test_lpips_topology_and_properties():
    """Layer list of VGG16's conv trunk as torchvision numbers it, five taps with LPIPS's channel counts, the wrapper's
    argument order and output shape; d(x, x) = 0, d >= 0 with non-negative 1x1 weights, only data gradients."""
    import lpips
    torch.manual_seed(0)
    m = lpips.PerceptualLoss(model='net-lin', net='vgg')
    names = [k for k in m.state_dict() if k.startswith('net.net.') and k.endswith('.weight')]
    assert names == [f'net.net.slice{s}.{i}.weight' for s, idx in enumerate(((0, 2), (5, 7), (10, 12, 14), (17, 19, 21),
                                                                              (24, 26, 28)), 1) for i in idx]
    assert [m.state_dict()[f'net.lin{i}.model.1.weight'].shape[1] for i in range(5)] == [64, 128, 256, 512, 512]
    assert not m.training and not any(p.requires_grad for p in m.parameters())
    a = synth.tensor('lp/a', (2, 3, 64, 64), dist='uniform').requires_grad_(True)
    b = synth.tensor('lp/b', (2, 3, 64, 64), dist='uniform')
    taps = m.net.net(m.net.scaling_layer(a))
    assert [tuple(t.shape[1:]) for t in taps] == [(64, 64, 64), (128, 32, 32), (256, 16, 16), (512, 8, 8), (512, 4, 4)]
    d = m(a, b)
    assert tuple(d.shape) == (2, 1, 1, 1) and float(d.detach().min()) > 0
    assert float(m(b, b).abs().max()) == 0.0
    torch.testing.assert_close(m(a, b), m(b, a))            # symmetric: (target, pred) order does not change the value
    g, = torch.autograd.grad(d.mean(), a)
    assert torch.isfinite(g).all() and float(g.abs().max()) > 0
    with pytest.raises(ValueError):
        lpips.PerceptualLoss(model='net', net='alex')


@pytest.mark.gpu
def test_face_identity_loss_matches_reference_on_gpu(golden):
    _check_face_id(golden('face_id'), 'cuda', 5e-6)
