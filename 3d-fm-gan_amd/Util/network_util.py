"""Composition of the hot path: (photo, render) -> image  (reference: Util/network_util.py).

Kept for the callers (train_3_encoder.py:460,507,574; Evaluation/visual_eval.py:116,180,296; quant_eval.py:93,162):
Forward_Inference_3_Encoder, Build_Generator_From_Dict, Get_Network_Shape, Get_Conv_Kernel_Key.
Image/PIL helpers of the reference file (torchvision-based) are not on the path and not provided.
"""
import os

import torch

from stylegan2 import Generator
from Util.streams import side_streams, run_on, overlap_ok

MODULATION_ENCODING = ['Render Image', 'Photo Image']
CO_MODULATION_MODE = ['Multiplication', 'Concatenation', 'Tensor Transform']


def Get_Conv_Kernel_Key(model_dict):
    """Keys of the main-path modulated conv weights, first to last (Util/network_util.py:22-36)."""
    return ['conv1.conv.weight'] + [k for k in model_dict.keys() if 'convs' in k and 'conv.weight' in k]


def Get_Network_Shape(model_dict):
    """Channel widths per synthesis layer, inferred from the conv weight shapes [1,Cout,Cin,k,k] (:39-50)."""
    keys = Get_Conv_Kernel_Key(model_dict)
    return [model_dict[k].shape[2] for k in keys] + [model_dict[keys[-1]].shape[1]]


def Build_Generator_From_Dict(model_dict, size=256, latent=512, n_mlp=8):
    """Generator whose (possibly pruned) widths are read off a state_dict, then loaded (:101-115)."""
    generator = Generator(size, latent, n_mlp, generator_net_shape=Get_Network_Shape(model_dict))
    generator.load_state_dict(model_dict, strict=False)
    return generator


def _unwrapped(net):
    # The reference reads g_ema.module.n_latent and therefore only works on a DataParallel-wrapped generator
    # (Util/network_util.py:317-318).  DDP and Miscellaneous.distributed.Replica also expose .module;
    # a bare Generator is accepted too.
    return getattr(net, 'module', net)


PIPELINE = os.environ.get('FMGAN_NO_PIPELINE', '0') != '1'
# Side streams share the device's 4 hardware queues with the main stream: measured on MI355X (pairs1024), ResNets on
# one stream + 2 head streams 339.8 pairs/s; 2 + 2: 335; 2 + 4: 325; 8 hardware queues (GPU_MAX_HW_QUEUES): 246.
RESNET_STREAMS = int(os.environ.get('FMGAN_RESNET_STREAMS', '1'))


def _pipelined(p_input, r_input, tsr_input, E_Tsr, E_W, e_wp, g_ema, sliced_layer, use_tanh):
    """Inference schedule of Forward_Inference_3_Encoder: the synthesis network consumes W+ one column per layer, so it
    starts as soon as the first style heads of the pSp encoder are done and runs beside the remaining heads (each a
    large conv followed by a tail of tiny launches) instead of after all of them.  Same arithmetic per element as the
    serial form: latent[:, i] = W * W+[:, i] (or W), image = g(latent, tsr)."""
    rs = side_streams(p_input.device, RESNET_STREAMS, 'resnets')
    s1, s2 = rs[0], rs[-1]
    join1, encoded_tensor = run_on(s1, E_Tsr, tsr_input)
    join2, encoded_W = run_on(s2, E_W, r_input)
    heads = e_wp.forward_deferred(p_input)
    join1(); join2()
    n_styles = len(heads)
    g = _unwrapped(g_ema)
    if sliced_layer is None:
        sliced_layer = range(g.n_latent)
    sliced = {i for i in sliced_layer if 0 <= i < n_styles}
    cache = {}

    def column(i):
        if i not in cache:
            wait, wp = heads[i]
            wait()
            cache[i] = encoded_W * wp if i in sliced else encoded_W
        return cache[i]

    g_output = g_ema(noise_z=None, use_external_input_tensor=True, external_input_tensor=encoded_tensor,
                     latent_columns=column)
    for wait, _ in heads:          # columns the generator did not read: still join their streams
        wait()
    return torch.tanh(g_output) if use_tanh else g_output


def Forward_Inference_3_Encoder(p_input, r_input, E_Tsr, E_W, E_W_Plus, g_ema, tsr_encode='Photo Image',
                                sliced_layer=None, use_tanh=False, PPL_regularize=False):
    """One forward of the 3-encoder scheme with multiplicative co-modulation (Util/network_util.py:293-338).

    tsr = E_Tsr(photo | render) [N,512,4,4];  W = E_W(render) [N,512];  W+ = E_W_Plus(photo) [N,n_latent,512];
    latent[:, i] = W * W+[:, i] for i in sliced_layer else W;  image = g_ema(latent, external_input_tensor=tsr).
    """
    if tsr_encode not in MODULATION_ENCODING:
        raise ValueError(f'tsr_encode must be one of {MODULATION_ENCODING}')
    tsr_input = p_input if tsr_encode == 'Photo Image' else r_input
    if (PIPELINE and overlap_ok(p_input) and not PPL_regularize and hasattr(_unwrapped(E_W_Plus), 'forward_deferred')
            and isinstance(_unwrapped(g_ema), Generator)):
        return _pipelined(p_input, r_input, tsr_input, E_Tsr, E_W, _unwrapped(E_W_Plus), g_ema, sliced_layer, use_tanh)
    if overlap_ok(p_input):
        # the three encoders are independent: the two small ResNets run on side streams beside the pSp encoder
        rs = side_streams(p_input.device, RESNET_STREAMS, 'resnets')
        join1, encoded_tensor = run_on(rs[0], E_Tsr, tsr_input)
        join2, encoded_W = run_on(rs[-1], E_W, r_input)
        encoded_W_plus = E_W_Plus(p_input)
        join1(); join2()
    else:
        encoded_tensor = E_Tsr(tsr_input)
        encoded_W = E_W(r_input)
        encoded_W_plus = E_W_Plus(p_input)

    n_styles = encoded_W_plus.shape[1]
    if sliced_layer is None:
        sliced_layer = range(_unwrapped(g_ema).n_latent)
    sliced = {i for i in sliced_layer if 0 <= i < n_styles}
    # W (.) W+ where sliced, W elsewhere.  Built from device tensors only (no host-side mask upload), so the whole
    # forward stays capturable in a HIP graph.
    if len(sliced) == n_styles:
        encoded_latent = encoded_W.unsqueeze(1) * encoded_W_plus
    else:
        encoded_latent = torch.stack([encoded_W * encoded_W_plus[:, i] if i in sliced else encoded_W
                                      for i in range(n_styles)], 1)

    g_output = g_ema(noise_z=None, latent_styles=[encoded_latent], input_is_latent=True,
                     use_external_input_tensor=True, external_input_tensor=encoded_tensor,
                     PPL_regularize=PPL_regularize)
    if use_tanh:
        if PPL_regularize:
            g_output = (torch.tanh(g_output[0]),) + tuple(g_output[1:])
        else:
            g_output = torch.tanh(g_output)
    return g_output
