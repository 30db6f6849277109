"""CPU restatement (plain PyTorch, functional, keyed by the reference's state_dict names) of the
reference's (photo, render) -> image path.

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg, never by the product package (3d-fm-gan_amd/), which has no CPU fallback at all.

Parity is PINNED against golden vectors produced by importing the reference itself on CPU
(tools/make_golden.py -> tests/golden/*.npz; checked in tests/test_oracle_golden.py).

Every function cites the reference lines it restates (paths relative to the reference root).
It is deliberately NOT structured like the reference (no nn.Module classes): it walks a
state_dict, so it also pins the parameter/buffer names and shapes the product must keep.
"""
import math

import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------- ops (op/*.py CPU branches)
def upfirdn2d(input, kernel, up=1, down=1, pad=(0, 0)):
    """upfirdn2d_native, op/upfirdn2d.py:168-209 (up_x=up_y, pad_x*=pad_y* as the dispatcher :154-165 passes)."""
    n, c, in_h, in_w = input.shape
    kh, kw = kernel.shape
    pad0, pad1 = pad
    x = input.reshape(-1, in_h, 1, in_w, 1)                      # :172-177 (minor == 1)
    x = F.pad(x, [0, up - 1, 0, 0, 0, up - 1])                   # :178 zero-stuffing
    x = x.reshape(-1, in_h * up, in_w * up)
    x = F.pad(x, [max(pad0, 0), max(pad1, 0), max(pad0, 0), max(pad1, 0)])   # :181-183
    x = x[:, max(-pad0, 0): x.shape[1] - max(-pad1, 0), max(-pad0, 0): x.shape[2] - max(-pad1, 0)]  # :184-189
    x = x.reshape(-1, 1, in_h * up + pad0 + pad1, in_w * up + pad0 + pad1)   # :192-194
    w = torch.flip(kernel, [0, 1]).view(1, 1, kh, kw)            # :195
    x = F.conv2d(x, w)                                           # :196
    x = x[:, :, ::down, ::down]                                  # :204
    out_h = (in_h * up + pad0 + pad1 - kh) // down + 1           # :206-207
    out_w = (in_w * up + pad0 + pad1 - kw) // down + 1
    return x.reshape(n, c, out_h, out_w)


def fused_leaky_relu(input, bias=None, negative_slope=0.2, scale=2 ** 0.5):
    """CUDA semantics of fused_bias_act act=3 grad=0 (op/fused_bias_act_kernel.cu:26-47), which is what
    the GPU path computes; equals the CPU branch op/fused_act.py:114-125 at the default slope 0.2."""
    if bias is not None:
        rest = [1] * (input.ndim - bias.ndim - 1)
        input = input + bias.view(1, bias.shape[0], *rest)
    return F.leaky_relu(input, negative_slope=negative_slope) * scale


def make_kernel(k):
    """stylegan2.py:36-44"""
    k = torch.tensor(k, dtype=torch.float32)
    if k.ndim == 1:
        k = k[None, :] * k[:, None]
    return k / k.sum()


def equal_linear(x, weight, bias, lr_mul=1.0, activation=False):
    """EqualLinear.forward, stylegan2.py:165-175; scale = lr_mul / sqrt(in_dim) (:162)."""
    scale = (1 / math.sqrt(weight.shape[1])) * lr_mul
    if activation:
        return fused_leaky_relu(F.linear(x, weight * scale), bias * lr_mul)
    return F.linear(x, weight * scale, bias=None if bias is None else bias * lr_mul)


def modulated_conv2d(x, w_latent, weight, mod_weight, mod_bias, demodulate=True, upsample=False,
                     blur_kernel=None):
    """ModulatedConv2d.forward, stylegan2.py:250-298 (plain and upsample branches; weight-modulated,
    groups=batch, exactly as the reference)."""
    batch, in_channel, height, width = x.shape
    _, out_channel, _, k, _ = weight.shape
    scale = 1 / math.sqrt(in_channel * k * k)                                  # :231-232
    style = equal_linear(w_latent, mod_weight, mod_bias).view(batch, 1, in_channel, 1, 1)   # :257
    w = scale * weight * style                                                 # :258
    if demodulate:
        demod = torch.rsqrt(w.pow(2).sum([2, 3, 4]) + 1e-8)                    # :261
        w = w * demod.view(batch, out_channel, 1, 1, 1)                        # :262
    if upsample:
        w = w.transpose(1, 2).reshape(batch * in_channel, out_channel, k, k)   # :270-275
        out = F.conv_transpose2d(x.reshape(1, batch * in_channel, height, width), w, padding=0, stride=2,
                                 groups=batch)                                 # :276
        out = out.view(batch, out_channel, out.shape[2], out.shape[3])
        # Blur(pad=(pad0,pad1), upsample_factor=2): kernel*4, :216-222, :98-99
        p = (len(blur_kernel) - 2) - (k - 1)
        kern = make_kernel(blur_kernel).to(x) * 4
        return upfirdn2d(out, kern, pad=((p + 1) // 2 + 1, p // 2 + 1))        # :279
    w = w.view(batch * out_channel, in_channel, k, k)
    out = F.conv2d(x.reshape(1, batch * in_channel, height, width), w, padding=k // 2, groups=batch)  # :291
    return out.view(batch, out_channel, out.shape[2], out.shape[3])


def modconv_bf16_reference(x, weight, style, demod, mode, scale):
    """What fmgan_modconv2d_bf16 computes, in float64: the input-modulated restatement of ModulatedConv2d
    (stylegan2.py:250-298; mode 0 plain / 1 transposed stride 2 / 2 stride-2 valid) with BOTH contraction operands
    rounded to bfloat16 (round-to-nearest-even) — the scaled weight fl32(scale*W) and the modulated activation
    fl32(x*style) — exact products and a wide accumulator.  x [B,Cin,H,W], weight [Cout,Cin,3,3], style [B,Cin],
    demod [B,Cout] or None.  The reduced-precision leg has no counterpart in the reference (parity unpinned by
    construction): this is the definition the kernel is tested against."""
    u = (x.float() * style.float()[:, :, None, None]).to(torch.bfloat16).double()
    wq = (weight.float() * torch.tensor(scale, dtype=torch.float32)).to(torch.bfloat16).double()
    if mode == 0:
        y = F.conv2d(u, wq, padding=1)
    elif mode == 1:
        y = F.conv_transpose2d(u, wq.transpose(0, 1), stride=2)
    else:
        y = F.conv2d(u, wq, stride=2)
    if demod is not None:
        y = y * demod.double()[:, :, None, None]
    return y


def styled_conv(sd, prefix, x, w_latent, noise, upsample, blur_kernel=(1, 3, 3, 1)):
    """StyledConv.forward, stylegan2.py:360-376 (conv -> NoiseInjection :307-312 -> FusedLeakyReLU)."""
    out = modulated_conv2d(x, w_latent, sd[prefix + '.conv.weight'], sd[prefix + '.conv.modulation.weight'],
                           sd[prefix + '.conv.modulation.bias'], True, upsample, list(blur_kernel))
    if noise is None:
        noise = torch.randn(out.shape[0], 1, out.shape[2], out.shape[3], dtype=out.dtype)
    out = out + sd[prefix + '.noise.weight'] * noise
    return fused_leaky_relu(out, sd[prefix + '.activate.bias'])


def to_rgb(sd, prefix, x, w_latent, skip=None, blur_kernel=(1, 3, 3, 1)):
    """ToRGB.forward, stylegan2.py:389-404; Upsample :47-65 (kernel*factor^2, pad (2,1))."""
    out = modulated_conv2d(x, w_latent, sd[prefix + '.conv.weight'], sd[prefix + '.conv.modulation.weight'],
                           sd[prefix + '.conv.modulation.bias'], demodulate=False)
    out = out + sd[prefix + '.bias']
    if skip is not None:
        kern = make_kernel(list(blur_kernel)).to(x) * 4
        p = kern.shape[0] - 2
        out = out + upfirdn2d(skip, kern, up=2, down=1, pad=((p + 1) // 2 + 1, p // 2))
    return out


def mapping_network(sd, z, n_mlp, lr_mlp=0.01):
    """Generator.style: PixelNorm (:32) + n_mlp EqualLinear(lr_mul=lr_mlp, fused_lrelu), stylegan2.py:427-436."""
    x = z * torch.rsqrt(torch.mean(z ** 2, dim=1, keepdim=True) + 1e-8)
    for i in range(n_mlp):
        x = equal_linear(x, sd[f'style.{i + 1}.weight'], sd[f'style.{i + 1}.bias'], lr_mul=lr_mlp, activation=True)
    return x


def generator_forward(sd, size, latent, external_input_tensor=None, noise=None, return_rgb_list=False):
    """Generator.forward with input_is_latent=True, stylegan2.py:554-688.
    latent [B, n_latent, 512] (or [B,512], repeated :612-613); external tensor replaces ConstantInput (:628-632);
    noise: list of num_layers tensors, or 'buffers' for randomize_noise=False (:589-594), or None (fresh)."""
    log_size = int(math.log(size, 2))
    num_layers = (log_size - 2) * 2 + 1
    n_latent = log_size * 2 - 2
    if latent.ndim < 3:
        latent = latent.unsqueeze(1).repeat(1, n_latent, 1)
    if noise == 'buffers':
        noise = [sd[f'noises.noise_{i}'] for i in range(num_layers)]
    elif noise is None:
        noise = [None] * num_layers
    if external_input_tensor is not None:
        out = external_input_tensor
    else:
        out = sd['input.input'].repeat(latent.shape[0], 1, 1, 1)               # ConstantInput :325-329
    out = styled_conv(sd, 'conv1', out, latent[:, 0], noise[0], upsample=False)  # :640
    skip = to_rgb(sd, 'to_rgb1', out, latent[:, 1])                            # :643
    rgbs = [skip]
    i = 1
    for j in range(log_size - 2):                                              # :647-666
        out = styled_conv(sd, f'convs.{2 * j}', out, latent[:, i], noise[1 + 2 * j], upsample=True)
        out = styled_conv(sd, f'convs.{2 * j + 1}', out, latent[:, i + 1], noise[2 + 2 * j], upsample=False)
        skip = to_rgb(sd, f'to_rgbs.{j}', out, latent[:, i + 2], skip)
        rgbs.append(skip)
        i += 2
    return rgbs if return_rgb_list else skip


# --------------------------------------------------------------------------- encoders
def _bn(sd, prefix, x, training=False):
    return F.batch_norm(x, sd[prefix + '.running_mean'], sd[prefix + '.running_var'], sd[prefix + '.weight'],
                        sd[prefix + '.bias'], training=training, momentum=0.0, eps=1e-5)


def resnet18_forward(sd, x, tensor_encoding=True):
    """resnet_encoder.ResNet._forward_impl with BasicBlock [2,2,2,2], resnet_encoder.py:258-283, 45-91.
    tensor_encoding: AvgPool2d(2,2) -> [N,512,4,4]; else AdaptiveAvgPool(1,1)+flatten -> [N,512] (:206-209,272-273).
    BatchNorm in eval mode."""
    x = F.conv2d(x, sd['conv1.weight'], stride=2, padding=3)
    x = F.relu(_bn(sd, 'bn1', x))
    x = F.max_pool2d(x, kernel_size=3, stride=2, padding=1)
    for li in range(1, 5):
        for bi in range(2):
            p = f'layer{li}.{bi}'
            stride = 2 if (li > 1 and bi == 0) else 1
            identity = x
            out = F.conv2d(x, sd[p + '.conv1.weight'], stride=stride, padding=1)
            out = F.relu(_bn(sd, p + '.bn1', out))
            out = F.conv2d(out, sd[p + '.conv2.weight'], stride=1, padding=1)
            out = _bn(sd, p + '.bn2', out)
            if (p + '.downsample.0.weight') in sd:
                identity = _bn(sd, p + '.downsample.1', F.conv2d(x, sd[p + '.downsample.0.weight'], stride=stride))
            x = F.relu(out + identity)
    if tensor_encoding:
        return F.avg_pool2d(x, kernel_size=2, stride=2)
    return torch.flatten(F.adaptive_avg_pool2d(x, (1, 1)), 1)


def psp_forward(sd, x, n_styles):
    """GradualStyleEncoder(18, 'ir_se').forward, psp_encoders.py:100-132; bottleneck_IR_SE helpers.py:117-139,
    SEModule helpers.py:76-92, get_blocks(18) helpers.py:38-45, GradualStyleBlock psp_encoders.py:20-41."""
    x = F.conv2d(x, sd['input_layer.0.weight'], stride=1, padding=1)
    x = F.prelu(_bn(sd, 'input_layer.1', x), sd['input_layer.2.weight'])
    units = [(64, 64, 2), (64, 64, 1), (64, 128, 2), (128, 128, 1), (128, 256, 2), (256, 256, 1), (256, 512, 2),
             (512, 512, 1)]
    feats = {}
    for n, (cin, depth, stride) in enumerate(units):
        p = f'body.{n}'
        if cin == depth:
            shortcut = x[:, :, ::stride, ::stride]                              # MaxPool2d(1, stride)
        else:
            shortcut = _bn(sd, p + '.shortcut_layer.1', F.conv2d(x, sd[p + '.shortcut_layer.0.weight'], stride=stride))
        r = _bn(sd, p + '.res_layer.0', x)
        r = F.prelu(F.conv2d(r, sd[p + '.res_layer.1.weight'], stride=1, padding=1), sd[p + '.res_layer.2.weight'])
        r = _bn(sd, p + '.res_layer.4', F.conv2d(r, sd[p + '.res_layer.3.weight'], stride=stride, padding=1))
        se = F.adaptive_avg_pool2d(r, 1)
        se = torch.sigmoid(F.conv2d(F.relu(F.conv2d(se, sd[p + '.res_layer.5.fc1.weight'])),
                                    sd[p + '.res_layer.5.fc2.weight']))
        x = r * se + shortcut
        feats[n] = x
    c1, c2, c3 = feats[3], feats[5], feats[7]                                   # :108-118

    def style_block(j, f, spatial):
        p = f'styles.{j}'
        for m in range(int(math.log2(spatial))):
            f = F.leaky_relu(F.conv2d(f, sd[f'{p}.convs.{2 * m}.weight'], sd[f'{p}.convs.{2 * m}.bias'], stride=2,
                                      padding=1), 0.01)
        return equal_linear(f.view(-1, 512), sd[p + '.linear.weight'], sd[p + '.linear.bias'])

    def up_add(a, b):                                                           # :82-98
        return F.interpolate(a, size=b.shape[2:], mode='bilinear', align_corners=True) + b

    latents = [style_block(j, c3, 16) for j in range(min(3, n_styles))]
    p2 = up_add(c3, F.conv2d(c2, sd['latlayer1.weight'], sd['latlayer1.bias']))
    latents += [style_block(j, p2, 32) for j in range(3, min(7, n_styles))]
    p1 = up_add(p2, F.conv2d(c1, sd['latlayer2.weight'], sd['latlayer2.bias']))
    latents += [style_block(j, p1, 64) for j in range(7, n_styles)]
    return torch.stack(latents, dim=1)


def forward_inference_3_encoder(p_input, r_input, sd_tsr, sd_w, sd_wplus, sd_g, size, tsr_encode='Photo Image',
                                sliced_layer=None, use_tanh=False, noise='buffers'):
    """Forward_Inference_3_Encoder, Util/network_util.py:293-338 (multiplicative co-modulation :320-327)."""
    n_latent = int(math.log(size, 2)) * 2 - 2
    tsr = resnet18_forward(sd_tsr, p_input if tsr_encode == 'Photo Image' else r_input, tensor_encoding=True)
    w = resnet18_forward(sd_w, r_input, tensor_encoding=False)
    w_plus = psp_forward(sd_wplus, p_input, n_latent)
    if sliced_layer is None:
        sliced_layer = range(n_latent)
    lat = [w * w_plus[:, i, :] if i in sliced_layer else w for i in range(w_plus.shape[1])]
    lat = torch.transpose(torch.stack(lat), 0, 1)
    out = generator_forward(sd_g, size, lat, external_input_tensor=tsr, noise=noise)
    return torch.tanh(out) if use_tanh else out


# --------------------------------------------------------------------------- discriminator
def discriminator_forward(sd, x, size, blur_kernel=(1, 3, 3, 1)):
    """Discriminator.forward, stylegan2.py:797-820; ConvLayer :692-737, ResBlock :740-759."""
    kern = make_kernel(list(blur_kernel)).to(x)
    log_size = int(math.log(size, 2))

    def conv_layer(p, x, k, downsample=False, activate=True, bias=True):
        idx = 0
        if downsample:
            pp = (len(blur_kernel) - 2) + (k - 1)
            x = upfirdn2d(x, kern, pad=((pp + 1) // 2, pp // 2))
            idx = 1
        w = sd[f'{p}.{idx}.weight']
        scale = 1 / math.sqrt(w.shape[1] * k * k)
        b = sd.get(f'{p}.{idx}.bias') if (bias and not activate) else None
        x = F.conv2d(x, w * scale, bias=b, stride=2 if downsample else 1, padding=0 if downsample else k // 2)
        if activate:
            x = fused_leaky_relu(x, sd[f'{p}.{idx + 1}.bias']) if bias else F.leaky_relu(x, 0.2) * math.sqrt(2)
        return x

    out = conv_layer('convs.0', x, 1)
    for n in range(1, log_size - 1):
        p = f'convs.{n}'
        y = conv_layer(p + '.conv1', out, 3)
        y = conv_layer(p + '.conv2', y, 3, downsample=True)
        s = conv_layer(p + '.skip', out, 1, downsample=True, activate=False, bias=False)
        out = (y + s) / math.sqrt(2)
    batch, channel, height, width = out.shape
    group = min(batch, 4)
    stddev = out.view(group, -1, 1, channel, height, width)
    stddev = torch.sqrt(stddev.var(0, unbiased=False) + 1e-8)
    stddev = stddev.mean([2, 3, 4], keepdim=True).squeeze(2).repeat(group, 1, height, width)
    out = torch.cat([out, stddev], 1)
    out = conv_layer('final_conv', out, 3)
    out = out.view(batch, -1)
    out = equal_linear(out, sd['final_linear.0.weight'], sd['final_linear.0.bias'], activation=True)
    return equal_linear(out, sd['final_linear.1.weight'], sd['final_linear.1.bias'])


# --------------------------------------------------------------------------- either side of the path (§8 f-4)
def images_to_tensor(images_u8_hwc, mean=0.5, std=0.5):
    """transforms.ToTensor() (HWC uint8 -> CHW float / 255) then Normalize(mean, std) = (t - mean) / std
    (train_3_encoder.py:233-239), float32 arithmetic in that order."""
    t = images_u8_hwc.permute(0, 3, 1, 2).to(torch.float32).div(255)
    return t.sub(mean).div(std)


def tensor2im_batch(image_tensor, cent=1.0, factor=255.0 / 2.0):
    """tensor2im, Evaluation/visual_eval.py:24-38, applied to every sample: clip to [-1,1], CHW -> HWC,
    (x + cent) * factor in float32, astype(uint8) (truncation)."""
    import numpy as np
    a = image_tensor.cpu().float().numpy()
    a = np.clip(a, a_min=-1, a_max=1)
    a = (np.transpose(a, (0, 2, 3, 1)) + np.float32(cent)) * np.float32(factor)
    return a.astype(np.uint8)
