"""Adversarial losses and regularisers of the training step (reference: Util/training_util.py:24-58, 103-113).

Only the pieces the hot path's callers need: the non-saturating logistic GAN pair, R1 on real images, the path-length
regulariser and the L1 reconstruction loss.  R1 and path length differentiate *through* a first derivative, i.e. they
exercise the double-backward of the HIP ops (upfirdn2d, fused_bias_act) and of the modulated conv.  The perceptual /
identity / landmark losses of the reference need pretrained third-party networks that are not on this path.
"""
import math

import torch
from torch import autograd
from torch.nn import functional as F


def d_logistic_loss(real_pred, fake_pred):
    """softplus(-D(real)) + softplus(D(fake)), batch means (training_util.py:39-43)."""
    return F.softplus(-real_pred).mean() + F.softplus(fake_pred).mean()


def g_nonsaturating_loss(fake_pred):
    """softplus(-D(G(z))) (training_util.py:55-58)."""
    return F.softplus(-fake_pred).mean()


def d_r1_loss(real_pred, real_img):
    """R1: E ||dD(x)/dx||^2 over real x; keeps the graph so the penalty itself can be back-propagated
    (training_util.py:46-52)."""
    grad_real, = autograd.grad(outputs=real_pred.sum(), inputs=real_img, create_graph=True)
    return grad_real.pow(2).reshape(grad_real.shape[0], -1).sum(1).mean()


def g_path_regularize(fake_img, latents, mean_path_length, decay=0.01, probe=None):
    """Path-length regulariser (training_util.py:24-37): |J^T y| for a random image-space probe y ~ N(0, 1/HW),
    penalised towards its running mean.  Returns (penalty, new running mean (detached), per-sample lengths).
    `probe` (unit-variance noise shaped like fake_img) may be supplied for reproducible tests."""
    if probe is None:
        probe = torch.randn_like(fake_img)
    probe = probe / math.sqrt(fake_img.shape[2] * fake_img.shape[3])
    grad, = autograd.grad(outputs=(fake_img * probe).sum(), inputs=latents, create_graph=True)
    path_lengths = torch.sqrt(grad.pow(2).sum(2).mean(1))
    path_mean = mean_path_length + decay * (path_lengths.mean() - mean_path_length)
    path_penalty = (path_lengths - path_mean).pow(2).mean()
    return path_penalty, path_mean.detach(), path_lengths


def L1_Loss(output_tensor, target_tensor):
    """Mean absolute error between two image batches in [-1, 1] (training_util.py:103-113)."""
    return torch.mean(torch.abs(output_tensor - target_tensor))


def requires_grad(model, flag=True):
    """Freeze / unfreeze every parameter of a network (train_3_encoder.py:188-190)."""
    for p in model.parameters():
        p.requires_grad = flag


def accumulate(model1, model2, decay=0.999):
    """EMA of the generator weights: model1 <- decay * model1 + (1 - decay) * model2, parameter by parameter
    (train_3_encoder.py:195-200; the reference writes through `.data`, which is also safe here: no module of this
    build keeps a derived copy of a weight across forwards, see op/live_weights.py).  Buffers (the fixed noise maps)
    are not averaged, exactly as in the reference."""
    par1 = dict(model1.named_parameters())
    par2 = dict(model2.named_parameters())
    with torch.no_grad():
        for k in par1.keys():
            par1[k].mul_(decay).add_(par2[k].detach(), alpha=1 - decay)
