"""One training iteration of the 3-encoder scheme on the MI355X path, data-parallel as one process per GPU.

The reference's train_3_encoder.py cannot be imported (a duplicated tail after main() is a SyntaxError) and
drives its GPUs through single-process nn.DataParallel.  This module re-states the step functions `train()` calls
(train_3_encoder.py:756-828), with the reference's names, argument order and arithmetic:

    D_Loss_BackProp   :448-477   logistic D loss on (reference image, generated image), G/encoders frozen
    D_Reg_BackProp    :479-493   R1 on the reference images every d_reg_every iterations (second order)
    G_Loss_BackProp   :495-558   non-saturating G loss + weighted reconstruction losses, G + encoders trained
    G_Reg_BackProp    :561-596   path-length regulariser every g_reg_every iterations on batch / shrink (second order)
    accumulate        :195-200   EMA of G into g_ema            Optimizer_Initilization :399-444

What is different, on purpose:
  * gradients are averaged over ranks (RCCL all-reduce: DDP buckets, or Miscellaneous.distributed.gather_grad) instead
    of DataParallel's reduce-to-device-0; the path-length mean is taken over the GLOBAL batch through a differentiable
    all-reduce, as DataParallel's gather made it;
  * the generated image of the D step comes from the no_grad inference schedule (fused epilogues, side streams): its
    producers are frozen there (requires_grad(G, False), :453-457), so values are identical and no graph is kept;
  * the LPIPS and ArcFace terms (:529,:535) are computed when `lpips_model` / `face_rec_model` are passed, with the
    reference's weights and arithmetic; their pretrained weights are not available offline, so
    Module_Fix_Setup builds the two networks (lpips/, Util/arcface_pytorch/) with their own initialisation — the cost of
    the iteration is that of the reference's, the loss VALUES are not.  The landmark heat-map and face-region terms
    (:538-545) need the third-party `face_alignment` package (absent; both weights default to 0 in the reference's
    reconstruction setting, train_3_encoder_hyperparams.py:66-68) and are not provided; `extra_losses` takes any
    further (name, weight, fn(output, reference) -> scalar) terms.
"""
import types

import numpy as np
import torch
from torch import nn, optim

from Miscellaneous import distributed as D_
from Util.network_util import Forward_Inference_3_Encoder, MODULATION_ENCODING
from Util.training_util import (Face_Identity_Loss, L1_Loss, LPIPS_Loss, accumulate, d_logistic_loss, d_r1_loss,
                                g_nonsaturating_loss, requires_grad)


def default_args(**over):
    """The fields of `args` the step functions read, with train_3_encoder_hyperparams.py:23-68 defaults."""
    a = types.SimpleNamespace(
        tsr_encode=MODULATION_ENCODING[0], w_plus_sliced_layer=None, use_tanh=False,
        tsr_train=True, w_train=True, w_plus_train=True,
        lr=0.001, rec_batch=16, r1=10, d_reg_every=16, use_g_reg=True, g_reg_every=4, generator_path_reg_weight=2,
        path_reg_batch_shrink=2, l1_loss_lambda=3, lpips_loss_lambda=3, ep_lpips_l1_weight_shrink=10,
        face_id_loss_lambda=30, face_id_loss_type='MSE', grad_sync='ddp')
    for k, v in over.items():
        setattr(a, k, v)
    return a


def _local(net):
    """The network without DDP's gradient hooks, for calls in which all of its parameters are frozen (a DDP forward
    with nothing to reduce would still arm the reducer); any other wrapper is called as given."""
    return net.module if isinstance(net, nn.parallel.DistributedDataParallel) else net


def _sync_grads(args, nets):
    """Explicit gradient averaging for Replica-wrapped networks (grad_sync='flat'); DDP does it inside backward."""
    if getattr(args, 'grad_sync', 'ddp') != 'flat' or D_.get_world_size() == 1:
        return
    params = [p for n in nets for p in n.parameters() if p.requires_grad]
    D_.gather_grad(params, algorithm=getattr(args, 'grad_algorithm', 'reduce_scatter'))


def Module_Fix_Setup(args, device):
    """The frozen loss networks of train_3_encoder.py:366-396 that can be built offline: LPIPS (net-lin / VGG16) and the
    ArcFace identity network (resnet_face18, use_se=False), eval mode, no parameter gradients.  Returns
    (lpips_model, face_rec_model).  The VGG trunk runs channels_last: MIOpen's fp32 implicit-GEMM kernels are NHWC-native
    (same finding as for the pSp encoder)."""
    import lpips
    from Util.arcface_pytorch.resnet_face_recognition import resnet_face18
    lpips_model = lpips.PerceptualLoss(model='net-lin', net='vgg').to(device)
    face_rec_model = resnet_face18(use_se=False).to(device)
    requires_grad(face_rec_model, False)
    face_rec_model.eval()
    if torch.device(device).type == 'cuda' and getattr(args, 'loss_nets_channels_last', True):
        lpips_model = lpips_model.to(memory_format=torch.channels_last)
        face_rec_model = face_rec_model.to(memory_format=torch.channels_last)
    return lpips_model, face_rec_model


def Optimizer_Initilization(args, G, E_Tsr, E_W, E_W_Plus, D, D_edit=None, ckpt=None):
    """Adam with the lazy-regularisation learning-rate / beta correction (train_3_encoder.py:399-444)."""
    g_reg_ratio = args.g_reg_every / (args.g_reg_every + 1)
    d_reg_ratio = args.d_reg_every / (args.d_reg_every + 1)
    g_enc_params = list(G.parameters())
    if args.tsr_train:
        g_enc_params += list(E_Tsr.parameters())
    if args.w_train:
        g_enc_params += list(E_W.parameters())
    if args.w_plus_train:
        g_enc_params += list(E_W_Plus.parameters())
    g_enc_optim = optim.Adam(g_enc_params, lr=args.lr * g_reg_ratio, betas=(0 ** g_reg_ratio, 0.99 ** g_reg_ratio))
    d_optim = optim.Adam(D.parameters(), lr=args.lr * d_reg_ratio, betas=(0 ** d_reg_ratio, 0.99 ** d_reg_ratio))
    d_edit_optim = None
    if D_edit is not None:
        d_edit_optim = optim.Adam(D_edit.parameters(), lr=args.lr * d_reg_ratio,
                                  betas=(0 ** d_reg_ratio, 0.99 ** d_reg_ratio))
    if ckpt is not None and getattr(args, 'load_train_state', False):
        g_enc_optim.load_state_dict(ckpt['g_enc_optim'])
        d_optim.load_state_dict(ckpt['d_optim'])
        if 'd_edit_optim' in ckpt and d_edit_optim is not None:
            d_edit_optim.load_state_dict(ckpt['d_edit_optim'])
    return g_enc_optim, d_optim, d_edit_optim


def D_Loss_BackProp(G, E_Tsr, E_W, E_W_Plus, D, g_input, r_input, g_ref, args, loss_dict, d_optim, d_type='D'):
    """Update D on the logistic GAN loss (train_3_encoder.py:448-477)."""
    requires_grad(G, False)
    requires_grad(E_Tsr, False)
    requires_grad(E_W, False)
    requires_grad(E_W_Plus, False)
    requires_grad(D, True)
    with torch.no_grad():       # every producer is frozen: same values, inference schedule, no graph
        g_output = Forward_Inference_3_Encoder(g_input, r_input, _local(E_Tsr), _local(E_W), _local(E_W_Plus), G,
                                               args.tsr_encode, args.w_plus_sliced_layer, args.use_tanh)
    out_pred = D(g_output)
    ref_pred = D(g_ref)
    d_loss = d_logistic_loss(ref_pred, out_pred)
    if d_type == 'D':
        loss_dict['d'] = d_loss
        loss_dict['ref_score'] = ref_pred.mean()
        loss_dict['out_score'] = out_pred.mean()
    else:
        loss_dict['d_edit'] = d_loss
        loss_dict['ref_score_ffhq'] = ref_pred.mean()
        loss_dict['out_score_ffhq'] = out_pred.mean()
    D.zero_grad()
    d_loss.backward()
    _sync_grads(args, [D])
    if d_optim is not None:
        d_optim.step()


def D_Reg_BackProp(real_img, D, args, d_optim):
    """Update D on the R1 penalty (train_3_encoder.py:479-493); returns the unweighted penalty."""
    real_img = real_img.detach().requires_grad_(True)
    real_pred = D(real_img)
    r1_loss = d_r1_loss(real_pred, real_img)
    D.zero_grad()
    (args.r1 / 2 * r1_loss * args.d_reg_every + 0 * real_pred[0]).backward()
    _sync_grads(args, [D])
    if d_optim is not None:
        d_optim.step()
    return r1_loss


def _trained(args, G, E_Tsr, E_W, E_W_Plus):
    nets = [G]
    if args.tsr_train:
        nets.append(E_Tsr)
    if args.w_train:
        nets.append(E_W)
    if args.w_plus_train:
        nets.append(E_W_Plus)
    return nets


def G_Loss_BackProp(G, E_Tsr, E_W, E_W_Plus, D, g_input, r_input, g_ref, args, loss_dict, g_enc_optim,
                    lpips_model=None, face_rec_model=None, fa_model=None, iter_idx=0, extreme_ds_flag=False,
                    ds_flag=False, extra_losses=()):
    """Update G and the encoders on the adversarial + reconstruction losses (train_3_encoder.py:495-558; the
    reference's argument order).  lpips_model / face_rec_model = None leaves that term out."""
    requires_grad(G, True)
    requires_grad(E_Tsr, args.tsr_train)
    requires_grad(E_W, args.w_train)
    requires_grad(E_W_Plus, args.w_plus_train)
    requires_grad(D, False)
    g_output = Forward_Inference_3_Encoder(g_input, r_input, E_Tsr, E_W, E_W_Plus, G, args.tsr_encode,
                                           args.w_plus_sliced_layer, args.use_tanh)
    out_pred = _local(D)(g_output)            # D is frozen here: nothing of it to synchronise
    g_loss = g_nonsaturating_loss(out_pred)
    loss_dict['g'] = g_loss
    shrink = args.ep_lpips_l1_weight_shrink if extreme_ds_flag else 1        # :517-519
    face_id_reference = g_input if extreme_ds_flag else g_ref
    l1_loss = args.l1_loss_lambda / shrink * L1_Loss(g_output, g_ref)
    loss_dict['l1'] = l1_loss
    total_loss = g_loss + l1_loss
    if lpips_model is not None:
        loss_dict['lpips'] = args.lpips_loss_lambda / shrink * LPIPS_Loss(g_output, g_ref, lpips_model)
        total_loss = total_loss + loss_dict['lpips']
    if face_rec_model is not None:
        loss_dict['face_id'] = args.face_id_loss_lambda * Face_Identity_Loss(g_output, face_id_reference, face_rec_model,
                                                                              args.face_id_loss_type)
        total_loss = total_loss + loss_dict['face_id']
    for name, weight, fn in extra_losses:
        loss_dict[name] = weight * fn(g_output, g_ref)
        total_loss = total_loss + loss_dict[name]
    nets = _trained(args, G, E_Tsr, E_W, E_W_Plus)
    for n in nets:
        n.zero_grad()
    total_loss.backward()
    _sync_grads(args, nets)
    if g_enc_optim is not None:
        g_enc_optim.step()


def _global_mean(x):
    """Mean over the samples of ALL ranks, differentiable (backward = the same all-reduce on the gradient)."""
    m = x.mean()
    if D_.get_world_size() == 1:
        return m
    from torch.distributed.nn import functional as dist_fn
    return dist_fn.all_reduce(m) / D_.get_world_size()


def G_Reg_BackProp(G, E_Tsr, E_W, E_W_Plus, g_input, r_input, args, mean_path_length, g_enc_optim, choice=None):
    """Update G and the encoders on the path-length regulariser (train_3_encoder.py:561-596).
    `choice`: indices of the batch/shrink samples (the reference draws them with np.random.choice)."""
    batch = g_input.shape[0]
    path_batch_size = max(1, batch // args.path_reg_batch_shrink)
    if choice is None:
        choice = np.random.choice(range(batch), size=path_batch_size, replace=False)
    idx = torch.as_tensor(np.asarray(choice), device=g_input.device, dtype=torch.long)
    g_input_reg, r_input_reg = g_input.index_select(0, idx), r_input.index_select(0, idx)
    g_output, path_lengths = Forward_Inference_3_Encoder(g_input_reg, r_input_reg, E_Tsr, E_W, E_W_Plus, G,
                                                         args.tsr_encode, args.w_plus_sliced_layer, args.use_tanh,
                                                         PPL_regularize=True)
    decay = 0.01
    path_mean = mean_path_length + decay * (_global_mean(path_lengths) - mean_path_length)
    path_loss = (path_lengths - path_mean).pow(2).mean()
    mean_path_length = path_mean.detach()
    nets = _trained(args, G, E_Tsr, E_W, E_W_Plus)
    for n in nets:
        n.zero_grad()
    weighted_path_loss = args.generator_path_reg_weight * args.g_reg_every * path_loss
    if args.path_reg_batch_shrink:
        weighted_path_loss = weighted_path_loss + 0 * g_output[0, 0, 0, 0]
    weighted_path_loss.backward()
    _sync_grads(args, nets)
    if g_enc_optim is not None:
        g_enc_optim.step()
    return path_loss, path_lengths, mean_path_length


class Trainer:
    """State of `train()` (train_3_encoder.py:756-828) for one rank: wrapped networks, g_ema, optimisers, counters.

    nets: dict with G, E_Tsr, E_W, E_W_Plus, D (bare modules on this rank's device; BatchNorm of the encoders in eval
    mode).  grad_sync: 'ddp' wraps them in DistributedDataParallel, 'flat' in Replica + gather_grad."""

    def __init__(self, nets, args, device=None, g_ema=None, lpips_model=None, face_rec_model=None):
        import copy
        self.args = args
        self.lpips_model, self.face_rec_model = lpips_model, face_rec_model
        ddp = getattr(args, 'grad_sync', 'ddp') == 'ddp'
        self.bare = dict(nets)
        self.g_ema = g_ema if g_ema is not None else copy.deepcopy(nets['G']).eval().requires_grad_(False)
        for m in nets.values():
            m.requires_grad_(True)
        # unused parameters exist in G (mapping network, constant input) and — when only some W+ columns are
        # co-modulated — in the pSp encoder's style heads of the other columns
        unused = {'G': True, 'E_W_Plus': args.w_plus_sliced_layer is not None}
        self.nets = {k: D_.data_parallel(m, device, overlap=ddp, find_unused_parameters=unused.get(k, False))
                     for k, m in nets.items()}
        self.g_enc_optim, self.d_optim, _ = Optimizer_Initilization(
            args, self.bare['G'], self.bare['E_Tsr'], self.bare['E_W'], self.bare['E_W_Plus'], self.bare['D'])
        self.accum = 0.5 ** (32 / (10 * 1000))
        self.mean_path_length = 0
        self.iter_idx = 0
        self.loss_dict = {'r1': torch.zeros((), device=device), 'g_reg': torch.zeros((), device=device)}

    def step(self, g_input, r_input, g_ref, ppl_choice=None):
        """One iteration: D, (R1), G, (path length), EMA — the order of train_3_encoder.py:801-822."""
        a, n, ld = self.args, self.nets, self.loss_dict
        G, E_Tsr, E_W, E_W_Plus, Dn = n['G'], n['E_Tsr'], n['E_W'], n['E_W_Plus'], n['D']
        D_Loss_BackProp(G, E_Tsr, E_W, E_W_Plus, Dn, g_input, r_input, g_ref, a, ld, self.d_optim)
        if self.iter_idx % a.d_reg_every == 0:
            ld['r1'] = D_Reg_BackProp(g_ref, Dn, a, self.d_optim)
        G_Loss_BackProp(G, E_Tsr, E_W, E_W_Plus, Dn, g_input, r_input, g_ref, a, ld, self.g_enc_optim,
                        self.lpips_model, self.face_rec_model, None, self.iter_idx)
        if self.iter_idx % a.g_reg_every == 0 and a.use_g_reg:
            ld['g_reg'], _, self.mean_path_length = G_Reg_BackProp(G, E_Tsr, E_W, E_W_Plus, g_input, r_input, a,
                                                                   self.mean_path_length, self.g_enc_optim, ppl_choice)
        accumulate(self.g_ema, self.bare['G'], self.accum)
        self.iter_idx += 1
        return ld
