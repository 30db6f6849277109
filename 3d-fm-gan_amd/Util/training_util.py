"""Adversarial losses and regularisers of the training step (reference: Util/training_util.py:24-58, 103-113).

The pieces the training iteration needs: the non-saturating logistic GAN pair, R1 on real images, the path-length
regulariser, the L1 reconstruction loss, and the wrappers of the perceptual (LPIPS) and identity (ArcFace) terms
(Util/training_util.py:115-127, 131-205).  R1 and path length differentiate *through* a first derivative, i.e. they
exercise the double-backward of the HIP ops (upfirdn2d, fused_bias_act) and of the modulated conv.  The landmark /
face-region terms need `face_alignment`, a third-party package that is absent offline; they stay out.
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


def LPIPS_Loss(output_tensor, target_tensor, lpips_module):
    """Batch mean of the LPIPS distance (training_util.py:115-127)."""
    return torch.mean(lpips_module(output_tensor, target_tensor))


RGB_TO_GRAYSCALE_COEF = (0.2989, 0.587, 0.114)
FACE_ID_LOSS_TYPE = ('MSE', 'CosineSimilarity')


def RGB_to_GrayScale(rgb_img):
    """[N,3,H,W] in [-1,1] -> [N,1,H,W], coefficients and summation order of training_util.py:131-146."""
    gray = 0
    for i, c in enumerate(RGB_TO_GRAYSCALE_COEF):
        gray = gray + c * rgb_img[:, i:i + 1, ...]
    return gray


def Convert_Tensor_For_Face_Recognition_Loss(img_tensor, face_size=128):
    """Grey image average-pooled to the 128^2 input the ArcFace network is built for (its fc5 takes 512*8*8).  The
    reference pools by 2 because it trains at 256^2 (training_util.py:148-162); at 1024^2 (BASELINE config 5) its fc
    layer would not fit, so the pooling factor is size/128 here: identical at 256^2."""
    gray = RGB_to_GrayScale(img_tensor)
    k = max(1, img_tensor.shape[-1] // face_size)
    return F.avg_pool2d(gray, kernel_size=k, stride=k)


def Face_Identity_Loss(output_tensor, target_tensor, face_rec_model, loss_type='MSE'):
    """Distance between the identity features of two image batches (training_util.py:178-205)."""
    assert loss_type in FACE_ID_LOSS_TYPE
    target_feature = face_rec_model(Convert_Tensor_For_Face_Recognition_Loss(target_tensor))
    output_feature = face_rec_model(Convert_Tensor_For_Face_Recognition_Loss(output_tensor))
    if loss_type == 'MSE':
        return F.mse_loss(output_feature, target_feature)
    return torch.mean(1 - F.cosine_similarity(output_feature, target_feature))


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
