"""ctypes binding of libfmgan_hip.so (include/fmgan_hip.h) — the only way the host code reaches a kernel.

There is NO fallback: if the library is missing, or a tensor is not on the GPU, this raises.
PyTorch is used for device memory and streams only; every launch goes on torch's current HIP stream
of the tensor's device (as the reference launches on at::cuda::getCurrentCUDAStream,
op/upfirdn2d_kernel.cu:213-215), asynchronously, so calls can be captured in a HIP graph.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('FMGAN_LIB') or os.path.join(os.path.dirname(_HERE), 'csrc', 'libfmgan_hip.so')

F32, F64, F16 = 0, 1, 2
_DTYPES = {torch.float32: F32, torch.float64: F64, torch.float16: F16}

_lib = None

# Under torch.autocast (BASELINE config 5's bf16 leg) the MIOpen convolutions hand bf16 tensors to their consumers; the
# HIP kernels compute in fp32 (their reference counterparts dispatch float / double / half only): every custom autograd
# Function of this package is entered with autocast off and its floating-point CUDA inputs cast to fp32.  A no-op when
# autocast is not active.
amp_fwd = torch.amp.custom_fwd(device_type='cuda', cast_inputs=torch.float32)
amp_bwd = torch.amp.custom_bwd(device_type='cuda')


def _sig(fn, argtypes, restype=ctypes.c_int):
    fn.argtypes = argtypes
    fn.restype = restype
    return fn


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f'{LIB_PATH} not found: the HIP extension is not built. Run `make -C {os.path.dirname(LIB_PATH)}` '
            f'(or __graft_entry__.build()). There is no CPU or PyTorch fallback for these ops.')
    L = ctypes.CDLL(LIB_PATH)
    vp, i, f, ll = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_longlong
    _sig(L.fmgan_abi_version, [])
    _sig(L.fmgan_status_string, [i], ctypes.c_char_p)
    _sig(L.fmgan_upfirdn2d_select, [i] * 15)
    _sig(L.fmgan_upfirdn2d_out_size, [i] * 12 + [ctypes.POINTER(i)] * 2)
    _sig(L.fmgan_upfirdn2d, [i, vp, vp, vp] + [i] * 15 + [vp])
    _sig(L.fmgan_upfirdn2d_strided, [i, vp, vp, vp] + [i] * 4 + [ll, i] + [i] * 11 + [vp])
    _sig(L.fmgan_blur_noise_bias_act_f32, [vp] * 3 + [i] * 4 + [ll, i] + [i] * 6 + [vp] * 3 + [i, f, f, vp])
    _sig(L.fmgan_blur_noise_bias_act_select, [vp] * 3 + [i] * 4 + [ll, i] + [i] * 6)
    _sig(L.fmgan_blur_noise_bias_act_path_f32, [vp] * 3 + [i] * 4 + [ll, i] + [i] * 6 + [vp] * 3 + [i, f, f, i, vp])
    _sig(L.fmgan_fused_bias_act, [i, vp, vp, vp, vp, ll, i, i, i, i, f, f, vp])
    _sig(L.fmgan_noise_bias_act_f32, [vp] * 5 + [i] * 4 + [f, f, vp])
    _sig(L.fmgan_fused_bias_act_bwd_blocks, [ll, i])
    _sig(L.fmgan_fused_bias_act_bwd_f32, [vp] * 4 + [ll, i, f, f, vp])
    _sig(L.fmgan_prelu_backward_blocks, [ll, i])
    _sig(L.fmgan_prelu_backward_f32, [vp] * 5 + [ll, i, vp])
    _sig(L.fmgan_modconv_demod_f32, [vp] * 3 + [i] * 4 + [f, f, vp])
    _sig(L.fmgan_modconv_wsq_f32, [vp] * 2 + [i] * 3 + [vp])
    _sig(L.fmgan_modconv_demod_wsq_f32, [vp] * 3 + [i] * 3 + [f, f, vp])
    _sig(L.fmgan_equal_linear_f32, [vp] * 4 + [i] * 3 + [vp])
    _sig(L.fmgan_wino_weight_f32, [vp, vp, i, i, vp])
    _sig(L.fmgan_wino_input_f32, [vp] * 3 + [i] * 4 + [vp])
    _sig(L.fmgan_wino_output_f32, [vp] * 6 + [i] * 6 + [f, f, vp])
    _sig(L.fmgan_modconv_weight_prep_f32, [vp, vp, i, i, i, f, i, vp])
    _sig(L.fmgan_modconv2d_workspace_bytes, [i] * 6, ll)
    _sig(L.fmgan_modconv2d_f32, [vp] * 5 + [i] * 6 + [vp] * 3 + [i, i, f, f, ll, i, vp, ll, vp])
    _sig(L.fmgan_modconv_weight_bf16_bytes, [i] * 3, ll)
    _sig(L.fmgan_modconv_weight_to_bf16, [vp, vp, i, i, i, vp])
    _sig(L.fmgan_modconv2d_bf16_supported, [i] * 6)
    _sig(L.fmgan_modconv2d_bf16, [vp] * 5 + [i] * 6 + [vp] * 3 + [i, i, f, f, ll, i, vp])
    _sig(L.fmgan_modconv_weight_bf16x3_bytes, [i] * 3, ll)
    _sig(L.fmgan_modconv_weight_to_bf16x3, [vp, vp, i, i, i, vp])
    _sig(L.fmgan_modconv2d_bf16x3_supported, [i] * 6)
    _sig(L.fmgan_modconv2d_bf16x3, [vp] * 5 + [i] * 6 + [vp] * 3 + [i, i, f, f, ll, i, vp])
    _sig(L.fmgan_modconv2d_rgb_fusable, [i] * 5)
    _sig(L.fmgan_torgb_weight_mod_f32, [vp] * 3 + [i] * 3 + [f, vp])
    _sig(L.fmgan_modconv2d_rgb_f32, [vp] * 5 + [i] * 5 + [vp] * 3 + [i, i, f, f] + [vp] * 4 + [i, vp])
    _sig(L.fmgan_modconv_wgrad_workspace_bytes, [i] * 5, ll)
    _sig(L.fmgan_modconv_wgrad_f32, [vp] * 5 + [i] * 5 + [f, vp, ll, vp])
    _sig(L.fmgan_modconv_wgrad_mode_workspace_bytes, [i] * 6, ll)
    _sig(L.fmgan_modconv_wgrad_mode_f32, [vp] * 5 + [i] * 6 + [f, vp, ll, vp])
    _sig(L.fmgan_images_to_tensor, [vp, vp, i, i, i, f, f, vp])
    _sig(L.fmgan_resize_output_size, [i, i, i, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)])
    _sig(L.fmgan_resize_plan_ints, [i] * 4, ll)
    _sig(L.fmgan_resize_plan, [i] * 4 + [vp, ll])
    _sig(L.fmgan_resize_bilinear_u8, [vp] * 4 + [i] * 5 + [f, f, vp])
    _sig(L.fmgan_tensor_to_images, [vp, vp, i, i, i, f, f, vp])
    _sig(L.fmgan_torgb_f32, [vp] * 6 + [i] * 4 + [f, vp])
    _sig(L.fmgan_torgb_backward_splits, [i] * 3)
    _sig(L.fmgan_torgb_backward_f32, [vp] * 6 + [i] * 4 + [f, vp])
    _sig(L.fmgan_refresh_entry_bytes, [])
    _sig(L.fmgan_weight_refresh_blocks, [i, i, i, i, ll], ll)
    _sig(L.fmgan_weight_refresh_f32, [vp, i, ll, vp])
    if L.fmgan_abi_version() != 1:
        raise RuntimeError('libfmgan_hip.so ABI version mismatch')
    _lib = L
    return L


def check(status, what):
    if status != 0:
        raise RuntimeError(f'{what}: {lib().fmgan_status_string(status).decode()} (status {status})')


def require_gpu(t, name):
    # op/upfirdn2d.cpp:8 CHECK_CUDA — same exception type (RuntimeError) and wording
    if not t.is_cuda:
        raise RuntimeError(f'{name} must be a CUDA tensor (this build has no CPU path)')


def dtype_code(t):
    try:
        return _DTYPES[t.dtype]
    except KeyError:
        raise RuntimeError(f'unsupported dtype {t.dtype}: float32/float64/float16 only') from None


def ptr(t):
    return None if t is None else t.data_ptr()


def fp(t):
    """data_ptr of an optional float32 GPU tensor.  The f32 entry points take raw pointers: a tensor of another dtype
    (a bf16 style vector under torch.autocast, a float64 gradcheck input) would be read past its end on the device — it is
    refused here, on the host, as a RuntimeError."""
    if t is None:
        return None
    if t.dtype != torch.float32 or not t.is_cuda:
        raise RuntimeError(f'libfmgan_hip f32 entry point got a {t.dtype} tensor on {t.device}: float32 GPU tensors only '
                           f'(cast before the call; under autocast the package\'s autograd Functions do)')
    return t.data_ptr()


class on_device:
    """Make the tensor's device current for the launch and hand out its current stream."""

    def __init__(self, t):
        self.dev = t.device
        self.guard = None

    def __enter__(self):
        if torch.cuda.current_device() != self.dev.index:
            self.guard = torch.cuda.device(self.dev)
            self.guard.__enter__()
        return torch.cuda.current_stream(self.dev).cuda_stream

    def __exit__(self, *exc):
        if self.guard is not None:
            self.guard.__exit__(*exc)
        return False


# ----------------------------------------------------------------------------- launch observer (bench.py)
BLUR_PATHS = {}      # (planes, in_h, in_w) -> kernel id of the last fused blur of that shape, while an observer asks


class _NullObserver:
    """bench.py installs an observer that brackets selected launches with HIP events on the launch stream."""
    wants_paths = False

    def begin(self, name, info):
        return None

    def end(self, token):
        pass


_observer = _NullObserver()


def set_observer(obs=None):
    global _observer
    _observer = obs if obs is not None else _NullObserver()


# ----------------------------------------------------------------------------- raw ops (no autograd)
def upfirdn2d_out_size(in_h, in_w, kh, kw, up_x, up_y, down_x, down_y, px0, px1, py0, py1):
    oh, ow = ctypes.c_int(), ctypes.c_int()
    check(lib().fmgan_upfirdn2d_out_size(in_h, in_w, kh, kw, up_x, up_y, down_x, down_y, px0, px1, py0, py1,
                                         ctypes.byref(oh), ctypes.byref(ow)), 'upfirdn2d_out_size')
    return oh.value, ow.value


def upfirdn2d(input, kernel, up_x, up_y, down_x, down_y, pad_x0, pad_x1, pad_y0, pad_y1, force_path=-1):
    """Same positional signature and layout as the reference's pybind `upfirdn2d` (op/upfirdn2d.cpp:12-23):
    input [major,in_h,in_w,minor], kernel [kh,kw] -> new tensor [major,out_h,out_w,minor]."""
    require_gpu(input, 'input')
    require_gpu(kernel, 'kernel')
    x = input.contiguous()
    k = kernel.to(dtype=x.dtype).contiguous()
    major, in_h, in_w, minor = x.shape
    kh, kw = k.shape
    out_h, out_w = upfirdn2d_out_size(in_h, in_w, kh, kw, up_x, up_y, down_x, down_y, pad_x0, pad_x1, pad_y0, pad_y1)
    if out_h <= 0 or out_w <= 0:
        raise RuntimeError(f'upfirdn2d: empty output {out_h}x{out_w}')
    out = torch.empty((major, out_h, out_w, minor), dtype=x.dtype, device=x.device)
    with on_device(x) as stream:
        tok = _observer.begin('upfirdn2d', (major, in_h, in_w, out_h, out_w, up_x, down_x, x.element_size()))
        check(lib().fmgan_upfirdn2d(dtype_code(x), ptr(x), ptr(k), ptr(out), major, in_h, in_w, minor, kh, kw,
                                    up_x, up_y, down_x, down_y, pad_x0, pad_x1, pad_y0, pad_y1, force_path, stream),
              'upfirdn2d')
        _observer.end(tok)
    return out


def upfirdn2d_strided(in_ptr, device, major, in_h, in_w, plane_stride, row_stride, kernel, pad_x0, pad_x1, pad_y0, pad_y1,
                      force_path=-1):
    """up=down=1 FIR of a strided f32 input [major, in_h, in_w] (see aligned_rows_buffer) -> contiguous [major,out_h,out_w].
    force_path: -1 automatic; 4 = register row-march (path 1), 5 = LDS-DMA ring (path 1b) — A/B tests only."""
    k = kernel.contiguous()
    kh, kw = k.shape
    out_h, out_w = upfirdn2d_out_size(in_h, in_w, kh, kw, 1, 1, 1, 1, pad_x0, pad_x1, pad_y0, pad_y1)
    out = torch.empty((major, out_h, out_w), dtype=torch.float32, device=device)
    with on_device(out) as stream:
        tok = _observer.begin('upfirdn2d', (major, in_h, in_w, out_h, out_w, 1, 1, 4))
        check(lib().fmgan_upfirdn2d_strided(F32, in_ptr, fp(k), fp(out), major, in_h, in_w, 1, plane_stride,
                                            row_stride, kh, kw, 1, 1, 1, 1, pad_x0, pad_x1, pad_y0, pad_y1, force_path,
                                            stream), 'upfirdn2d_strided')
        _observer.end(tok)
    return out


def blur_noise_bias_act(in_ptr, device, batch, channels, in_h, in_w, plane_stride, row_stride, kernel, pad, noise,
                        noise_weight, bias, alpha, scale, force_path=-1, out=None):
    """blur -> (+noise) -> +bias -> lrelu*scale in one pass over a strided f32 input; returns [B,C,out_h,out_w] or None
    when neither the row-march kernels (out_w >= 64) nor the plane-tile kernel (planes up to ~110^2) serve the shape
    (the caller then uses the two-pass form)."""
    k = kernel.contiguous()
    kh, kw = k.shape
    pad0, pad1 = pad
    out_h, out_w = upfirdn2d_out_size(in_h, in_w, kh, kw, 1, 1, 1, 1, pad0, pad1, pad0, pad1)
    if out_w <= 0 or out_h <= 0:
        return None
    if out is None:
        out = torch.empty((batch, channels, out_h, out_w), dtype=torch.float32, device=device)
    elif tuple(out.shape) != (batch, channels, out_h, out_w) or not out.is_contiguous() or out.dtype != torch.float32:
        raise RuntimeError('blur_noise_bias_act: `out` must be a contiguous f32 [B,C,out_h,out_w] tensor')
    nz = noise.contiguous() if noise is not None else None
    if getattr(_observer, 'wants_paths', False):
        BLUR_PATHS[(batch * channels, in_h, in_w)] = lib().fmgan_blur_noise_bias_act_select(
            in_ptr, fp(out), fp(nz), batch, channels, in_h, in_w, plane_stride, row_stride, kh, kw, pad0, pad1, pad0, pad1)
    with on_device(out) as stream:
        tok = _observer.begin('upfirdn2d', (batch * channels, in_h, in_w, out_h, out_w, 1, 1, 4))
        st = lib().fmgan_blur_noise_bias_act_path_f32(in_ptr, fp(k), fp(out), batch, channels, in_h, in_w,
                                                      plane_stride, row_stride, kh, kw, pad0, pad1, pad0, pad1, fp(nz),
                                                      fp(noise_weight), fp(bias), 1 if nz is None else nz.shape[0],
                                                      float(alpha), float(scale), force_path, stream)
        _observer.end(tok)
    if st == -2:
        return None
    check(st, 'blur_noise_bias_act')
    return out


def fused_bias_act(input, bias, refer, act, grad, alpha, scale):
    """Same positional signature as the reference's pybind `fused_bias_act` (op/fused_bias_act.cpp:11-21);
    empty `bias` / `refer` tensors mean "absent" (op/fused_bias_act_kernel.cu:62-63)."""
    require_gpu(input, 'input')
    if bias is not None and bias.numel():
        require_gpu(bias, 'bias')
    x = input.contiguous()
    b = bias.to(dtype=x.dtype).contiguous() if bias is not None and bias.numel() else None
    r = refer.to(dtype=x.dtype).contiguous() if refer is not None and refer.numel() else None
    if r is not None and r.numel() != x.numel():
        raise RuntimeError('fused_bias_act: refer must have as many elements as input')
    step_b = 1
    for d in x.shape[2:]:
        step_b *= d
    out = torch.empty_like(x)
    with on_device(x) as stream:
        tok = _observer.begin('fused_bias_act', (x.numel(), x.element_size()))
        check(lib().fmgan_fused_bias_act(dtype_code(x), ptr(x), ptr(b), ptr(r), ptr(out), x.numel(),
                                         0 if b is None else b.numel(), step_b, int(act), int(grad), float(alpha),
                                         float(scale), stream), 'fused_bias_act')
        _observer.end(tok)
    return out


def fused_bias_act_backward(grad_output, out, alpha, scale):
    """grad_input of lrelu(x + b) * scale AND the bias gradient from ONE pass over the data: returns
    (grad_input, grad_bias [C]) or None when the kernel does not serve the shape (not f32, < 3 dims, planes of fewer
    than 64 or not a multiple of 4 elements) — the caller then runs fused_bias_act(..., 3, 1) and a torch sum."""
    require_gpu(grad_output, 'input')
    if grad_output.dtype != torch.float32 or grad_output.ndim < 3 or out.dtype != torch.float32:
        return None
    g = grad_output.contiguous()
    r = out.contiguous()
    b, c = g.shape[:2]
    hw = g[0, 0].numel()
    gx = lib().fmgan_fused_bias_act_bwd_blocks(b * c, hw)
    if gx == 0 or r.numel() != g.numel():
        return None
    gi = torch.empty_like(g)
    partial = torch.empty((b, c, gx), dtype=torch.float32, device=g.device)
    with on_device(g) as stream:
        tok = _observer.begin('fused_bias_act', (g.numel(), 4))
        st = lib().fmgan_fused_bias_act_bwd_f32(fp(g), fp(r), fp(gi), fp(partial), b * c, hw, float(alpha),
                                                float(scale), stream)
        _observer.end(tok)
    if st == -2:
        return None
    check(st, 'fused_bias_act_backward')
    return gi, partial.sum((0, 2))


def noise_bias_act(x, noise, noise_weight, bias, alpha, scale):
    """lrelu(x + noise_weight*noise + bias[c]) * scale in one pass; x [B,C,H,W] f32, noise [1|B,1,H,W]."""
    require_gpu(x, 'input')
    x = x.contiguous()
    b, c, h, w = x.shape
    nz = noise.contiguous() if noise is not None else None
    nb = 1 if nz is None else nz.shape[0]
    out = torch.empty_like(x)
    with on_device(x) as stream:
        tok = _observer.begin('noise_bias_act', (x.numel(), 4))
        check(lib().fmgan_noise_bias_act_f32(fp(x), fp(nz), fp(noise_weight), fp(bias), fp(out), b, c, h * w, nb,
                                             float(alpha), float(scale), stream), 'noise_bias_act')
        _observer.end(tok)
    return out


def prelu_backward(x, grad, slope):
    """Backward of PReLU on channels-innermost data: x, grad [rows, C] f32 contiguous, slope [C] ->
    (grad_x [rows, C], grad_slope [C]).  Returns None when the kernel does not serve the shape (C % 4 != 0)."""
    require_gpu(x, 'input')
    rows, c = x.shape
    if c % 4 != 0 or rows == 0:
        return None
    blocks = lib().fmgan_prelu_backward_blocks(rows, c)
    gx = torch.empty_like(x)
    partial = torch.empty((blocks, c), dtype=torch.float32, device=x.device)
    with on_device(x) as stream:
        st = lib().fmgan_prelu_backward_f32(fp(x), fp(grad), fp(slope), fp(gx), fp(partial), rows, c, stream)
    if st == -2:
        return None
    check(st, 'prelu_backward')
    return gx, partial.sum(0)


def modconv_demod(weight, style, scale, eps=1e-8, wsq=None):
    """weight [cout,cin,k,k] (or [1,cout,cin,k,k]) f32, style [B,cin] -> demod [B,cout].
    wsq: optional cached modconv_wsq(weight) — same bits, 1/k^2 of the reads."""
    cout, cin, kh, kw = weight.shape[-4:]
    style = style.contiguous()
    demod = torch.empty((style.shape[0], cout), dtype=torch.float32, device=style.device)
    with on_device(style) as stream:
        if wsq is not None:
            check(lib().fmgan_modconv_demod_wsq_f32(fp(wsq), fp(style), fp(demod), style.shape[0], cout, cin,
                                                    float(scale), float(eps), stream), 'modconv_demod_wsq')
        else:
            check(lib().fmgan_modconv_demod_f32(fp(weight), fp(style), fp(demod), style.shape[0], cout, cin,
                                                kh * kw, float(scale), float(eps), stream), 'modconv_demod')
    return demod


def equal_linear(x, weight, bias=None):
    """out = x @ weight.T (+ bias): x [B,K] f32, weight [N,K] f32 (already scaled), bias [N] or None -> [B,N]."""
    require_gpu(x, 'input')
    x, weight = x.contiguous(), weight.contiguous()
    b, k = x.shape
    n = weight.shape[0]
    if weight.shape[1] != k:
        raise RuntimeError(f'equal_linear: weight {tuple(weight.shape)} does not match input {tuple(x.shape)}')
    out = torch.empty((b, n), dtype=torch.float32, device=x.device)
    with on_device(x) as stream:
        check(lib().fmgan_equal_linear_f32(fp(x), fp(weight), fp(bias), fp(out), b, n, k, stream), 'equal_linear')
    return out


def modconv_wsq(weight):
    """[.., cout,cin,k,k] -> per-(o,i) sum of squared taps [cout,cin] (input of modconv_demod(wsq=...))."""
    cout, cin, kh, kw = weight.shape[-4:]
    wsq = torch.empty((cout, cin), dtype=torch.float32, device=weight.device)
    with on_device(weight) as stream:
        check(lib().fmgan_modconv_wsq_f32(fp(weight), fp(wsq), cout, cin, kh * kw, stream), 'modconv_wsq')
    return wsq


def modconv_weight_prep(weight, scale, kind=0):
    """[.., cout,cin,k,k] -> MFMA A-operand layout of scale*weight: kind 0 [cin, k*k, cout] (forward);
    kind 1 [cout, k*k, cin] with flipped taps (data-gradient of the plain conv); kind 2 [cout, k*k, cin]
    (data-gradient of the transposed conv)."""
    cout, cin, kh, kw = weight.shape[-4:]
    shape = (cin, kh * kw, cout) if kind == 0 else (cout, kh * kw, cin)
    wt = torch.empty(shape, dtype=torch.float32, device=weight.device)
    with on_device(weight) as stream:
        check(lib().fmgan_modconv_weight_prep_f32(fp(weight), fp(wt), cout, cin, kh * kw, float(scale), int(kind),
                                                  stream), 'modconv_weight_prep')
    return wt


def aligned_rows_shape(b, c, oh, ow, pad0):
    """(storage shape, element offset of logical (0,0,0,0), plane stride, row stride) of aligned_rows_buffer."""
    off = pad0 % 4
    rs = (ow + off + 31) // 32 * 32
    return (b * c, oh, rs), off, oh * rs, rs


def aligned_rows_buffer(b, c, oh, ow, pad0, device):
    """Private conv_transpose -> blur intermediate: rows padded to a multiple of 32 floats (one 128-byte line) and
    shifted right by pad0 (mod 4) floats, so that the blur's first tap column (x = -pad0) sits on a 16-byte boundary
    and every wave-wide 1 KiB row load covers exactly 8 cache lines.  Measured on the 1024^2 blur (MI355X):
    contiguous 2W+1 rows 512-527 us, 16-byte-aligned rows 498 us, line-aligned rows 438-450 us (= copy speed).
    Returns (storage, data_ptr of logical element (0,0,0,0), plane stride, row stride) — strides in elements."""
    off = pad0 % 4
    rs = (ow + off + 31) // 32 * 32
    buf = torch.empty((b * c, oh, rs), dtype=torch.float32, device=device)
    return buf, buf.data_ptr() + 4 * off, oh * rs, rs


# Contraction of modconv2d: 'f32' (default, the parity path: v_mfma_f32_32x32x2_f32), 'bf16' (bf16 MFMA operands, fp32
# accumulation; BASELINE config 5's reduced-precision leg) or 'bf16x3' (fp32 operands split into three bf16 pieces, six
# bf16 MFMAs per product: fp32 accuracy at 2.7x the fp32 matrix rate; forward only, labelled).  Context manager only.
_mc_precision = 'f32'


class modconv_precision:
    """`with modconv_precision('bf16'):` — every modconv2d call inside (forward and data-gradient contractions of the
    modulated conv) whose shape the bf16 kernel serves runs on v_mfma_f32_32x32x16_bf16; the rest stay fp32."""

    def __init__(self, precision):
        if precision not in ('f32', 'bf16', 'bf16x3'):
            raise ValueError("precision must be 'f32', 'bf16' or 'bf16x3'")
        self.precision = precision

    def __enter__(self):
        global _mc_precision
        self.prev, _mc_precision = _mc_precision, self.precision
        return self

    def __exit__(self, *exc):
        global _mc_precision
        _mc_precision = self.prev
        return False


def modconv_weight_to_bf16(wt):
    """fp32 MFMA layout wt [cin, taps, cout] (modconv_weight_prep) -> bf16 operand image (opaque int16 tensor)."""
    cin, taps, cout = wt.shape
    nbytes = lib().fmgan_modconv_weight_bf16_bytes(cin, cout, taps)
    wtb = torch.empty(nbytes // 2, dtype=torch.int16, device=wt.device)
    with on_device(wt) as stream:
        check(lib().fmgan_modconv_weight_to_bf16(fp(wt), ptr(wtb), cin, cout, taps, stream), 'modconv_weight_to_bf16')
    return wtb


def wino_weight(wt):
    """fp32 MFMA layout wt [cin, 9, cout] (scaled) -> Winograd F(2x2,3x3) weight U [16, cout, cin]."""
    cin, taps, cout = wt.shape
    if taps != 9:
        raise RuntimeError('wino_weight: 3x3 weights only')
    u = torch.empty((16, cout, cin), dtype=torch.float32, device=wt.device)
    with on_device(wt) as stream:
        check(lib().fmgan_wino_weight_f32(fp(wt), fp(u), cin, cout, stream), 'wino_weight')
    return u


def modconv2d_winograd(x, wt, style, demod, noise=None, noise_weight=None, bias=None, fuse_act=False, alpha=0.2,
                       act_scale=2 ** 0.5, u=None):
    """Plain 3x3 modulated conv in Winograd F(2x2,3x3) form: input transform (own kernel), 16 batched fp32 GEMMs (torch.bmm ->
    the BLAS library), output transform + StyledConv epilogue (own kernel).  x [B,cin,H,W] with H, W even; wt as for
    modconv2d (or u = wino_weight(wt) prepared by the caller)."""
    require_gpu(x, 'input')
    x, style = x.contiguous(), style.contiguous()
    b, cin, h, w = x.shape
    if (h | w) & 1:
        raise RuntimeError('modconv2d_winograd: H and W must be even')
    if u is None:
        u = wino_weight(wt)
    cout = u.shape[1]
    n = b * (h // 2) * (w // 2)
    v = torch.empty((16, cin, n), dtype=torch.float32, device=x.device)
    m = torch.empty((16, cout, n), dtype=torch.float32, device=x.device)
    out = torch.empty((b, cout, h, w), dtype=torch.float32, device=x.device)
    nz = noise.contiguous() if noise is not None else None
    with on_device(x) as stream:
        tok = _observer.begin('modconv2d_winograd', (b, cin, cout, h, w, 0))
        check(lib().fmgan_wino_input_f32(fp(x), fp(style), fp(v), b, cin, h, w, stream), 'wino_input')
        torch.bmm(u, v, out=m)
        check(lib().fmgan_wino_output_f32(fp(m), fp(demod), fp(nz), fp(noise_weight), fp(bias), fp(out), b, cout, h, w,
                                          1 if nz is None else nz.shape[0], int(bool(fuse_act)), float(alpha),
                                          float(act_scale), stream), 'wino_output')
        _observer.end(tok)
    return out


def modconv_weight_to_bf16x3(wt):
    """fp32 MFMA layout wt [cin, taps, cout] -> its three bf16 operand images (hi, mid, lo) in one opaque tensor."""
    cin, taps, cout = wt.shape
    nbytes = lib().fmgan_modconv_weight_bf16x3_bytes(cin, cout, taps)
    wts = torch.empty(nbytes // 2, dtype=torch.int16, device=wt.device)
    with on_device(wt) as stream:
        check(lib().fmgan_modconv_weight_to_bf16x3(fp(wt), ptr(wts), cin, cout, taps, stream), 'modconv_weight_to_bf16x3')
    return wts


def current_modconv_precision():
    return _mc_precision


def modconv2d(x, wt, style, demod, mode, noise=None, noise_weight=None, bias=None, fuse_act=False, alpha=0.2,
              act_scale=2 ** 0.5, strided_out=None, precision=None):
    """x [B,cin,H,W] f32, wt from modconv_weight_prep (3x3), style [B,cin], demod [B,cout] or None.
    strided_out = (ptr, plane_stride, row_stride) writes into a caller-owned strided buffer and returns None.
    precision: None = the ambient modconv_precision (default 'f32')."""
    require_gpu(x, 'input')
    x = x.contiguous()
    style = style.contiguous()
    b, cin, h, w = x.shape
    cout = wt.shape[2]
    oh, ow = (2 * h + 1, 2 * w + 1) if mode == 1 else (((h - 3) // 2 + 1, (w - 3) // 2 + 1) if mode == 2 else (h, w))
    if strided_out is None:
        out = torch.empty((b, cout, oh, ow), dtype=torch.float32, device=x.device)
        out_ptr, ops, ors = out.data_ptr(), 0, 0
    else:
        out = None
        out_ptr, ops, ors = strided_out
    nz = noise.contiguous() if noise is not None else None
    prec = precision or _mc_precision
    if prec == 'bf16x3' and lib().fmgan_modconv2d_bf16x3_supported(b, cin, cout, h, w, mode):
        wts = modconv_weight_to_bf16x3(wt)
        with on_device(x) as stream:
            tok = _observer.begin('modconv2d_bf16x3', (b, cin, cout, h, w, mode))
            check(lib().fmgan_modconv2d_bf16x3(fp(x), ptr(wts), fp(style), fp(demod), out_ptr, b, cin, cout, h, w, mode,
                                               fp(nz), fp(noise_weight), fp(bias), 1 if nz is None else nz.shape[0],
                                               int(bool(fuse_act)), float(alpha), float(act_scale), ops, ors, stream),
                  'modconv2d_bf16x3')
            _observer.end(tok)
        return out
    if prec == 'bf16' and lib().fmgan_modconv2d_bf16_supported(b, cin, cout, h, w, mode):
        wtb = modconv_weight_to_bf16(wt)
        with on_device(x) as stream:
            tok = _observer.begin('modconv2d_bf16', (b, cin, cout, h, w, mode))
            check(lib().fmgan_modconv2d_bf16(fp(x), ptr(wtb), fp(style), fp(demod), out_ptr, b, cin, cout, h, w, mode,
                                             fp(nz), fp(noise_weight), fp(bias), 1 if nz is None else nz.shape[0],
                                             int(bool(fuse_act)), float(alpha), float(act_scale), ops, ors, stream),
                  'modconv2d_bf16')
            _observer.end(tok)
        return out
    ws_bytes = lib().fmgan_modconv2d_workspace_bytes(b, cin, cout, h, w, mode)
    ws = torch.empty(ws_bytes // 4, dtype=torch.float32, device=x.device) if ws_bytes else None
    with on_device(x) as stream:
        tok = _observer.begin('modconv2d', (b, cin, cout, h, w, mode))
        check(lib().fmgan_modconv2d_f32(fp(x), fp(wt), fp(style), fp(demod), out_ptr, b, cin, cout, h, w, mode,
                                        fp(nz), fp(noise_weight), fp(bias), 1 if nz is None else nz.shape[0],
                                        int(bool(fuse_act)), float(alpha), float(act_scale), ops, ors, fp(ws), ws_bytes,
                                        stream), 'modconv2d')
        _observer.end(tok)
    return out


def modconv2d_rgb_fusable(batch, cin, cout, h, w):
    return bool(lib().fmgan_modconv2d_rgb_fusable(int(batch), int(cin), int(cout), int(h), int(w)))


def modconv2d_rgb(x, wt, style, demod, noise, noise_weight, bias, alpha, act_scale, rgb_weight, rgb_style, rgb_bias,
                  rgb_skip, rgb_scale, keep_out=True):
    """Plain 3x3 modulated conv + noise/bias/LeakyReLU with the following ToRGB (1x1 modulated conv + bias + skip) in
    the same kernel.  Returns (activation or None, rgb [B,3,H,W]).  Shapes must satisfy modconv2d_rgb_fusable()."""
    require_gpu(x, 'input')
    x = x.contiguous()
    style, rgb_style = style.contiguous(), rgb_style.contiguous()
    b, cin, h, w = x.shape
    cout = wt.shape[2]
    rgb_c = rgb_weight.numel() // cout
    out = torch.empty((b, cout, h, w), dtype=torch.float32, device=x.device) if keep_out else None
    rgb = torch.empty((b, rgb_c, h, w), dtype=torch.float32, device=x.device)
    nz = noise.contiguous() if noise is not None else None
    sk = rgb_skip.contiguous() if rgb_skip is not None else None
    wmod = torch.empty((b, 3, cout), dtype=torch.float32, device=x.device)
    with on_device(x) as stream:
        check(lib().fmgan_torgb_weight_mod_f32(fp(rgb_weight), fp(rgb_style), fp(wmod), b, cout, rgb_c,
                                               float(rgb_scale), stream), 'torgb_weight_mod')
        tok = _observer.begin('modconv2d', (b, cin, cout, h, w, 0))
        check(lib().fmgan_modconv2d_rgb_f32(fp(x), fp(wt), fp(style), fp(demod), fp(out), b, cin, cout, h, w,
                                            fp(nz), fp(noise_weight), fp(bias), 1 if nz is None else nz.shape[0],
                                            1, float(alpha), float(act_scale), fp(wmod), fp(rgb_bias), fp(sk),
                                            fp(rgb), rgb_c, stream), 'modconv2d_rgb')
        _observer.end(tok)
    return out, rgb


def modconv_wgrad(go, demod, x, style, scale, fast_only=False, mode=0):
    """Conv part of the weight gradient of the modulated conv (mode as in modconv2d; x is the conv's input, go the
    gradient of its output) -> [cout,cin,3,3]; None if the shape is not served by the kernel, or — with fast_only — not
    by its 64 x 64-tile form (>= 48 channels on both sides)."""
    go, x, style = go.contiguous(), x.contiguous(), style.contiguous()
    b, cout = go.shape[:2]
    cin, h, w = x.shape[1:]
    if (fast_only or mode != 0) and (cin < 48 or cout < 48):
        return None
    ws_bytes = lib().fmgan_modconv_wgrad_mode_workspace_bytes(b, cin, cout, h, w, int(mode))
    if ws_bytes == 0:
        return None
    ws = torch.empty(ws_bytes // 4, dtype=torch.float32, device=x.device)
    gw = torch.empty((cout, cin, 3, 3), dtype=torch.float32, device=x.device)
    with on_device(x) as stream:
        tok = _observer.begin('modconv_wgrad', (b, cin, cout, h, w, int(mode)))
        check(lib().fmgan_modconv_wgrad_mode_f32(fp(go), fp(demod), fp(x), fp(style), fp(gw), b, cin, cout, h, w,
                                                 int(mode), float(scale), fp(ws), ws_bytes, stream), 'modconv_wgrad')
        _observer.end(tok)
    return gw


def torgb(x, weight, style, bias, skip, scale):
    """x [B,cin,H,W] f32, weight [cout,cin] (any leading/trailing 1 dims), style [B,cin], bias [cout], skip [B,cout,H,W]."""
    require_gpu(x, 'input')
    x = x.contiguous()
    style = style.contiguous()
    b, cin, h, w = x.shape
    cout = weight.numel() // cin
    sk = skip.contiguous() if skip is not None else None
    out = torch.empty((b, cout, h, w), dtype=torch.float32, device=x.device)
    with on_device(x) as stream:
        tok = _observer.begin('torgb', (b, cin, cout, h * w))
        check(lib().fmgan_torgb_f32(fp(x), fp(weight), fp(style), fp(bias), fp(sk), fp(out), b, cin, cout, h * w,
                                    float(scale), stream), 'torgb')
        _observer.end(tok)
    return out


def torgb_backward(x, grad_out, weight, style, scale):
    """Data gradient of ToRGB's 1x1 modulated conv and the pixel contraction M[b,c,i] = sum_p grad_out[b,c,p]*x[b,i,p]
    from one pass over x.  Returns (grad_x [B,cin,H,W], M [B,cout,cin]) or None when the kernel does not serve the
    shape (H*W % 4 != 0, not f32)."""
    require_gpu(x, 'input')
    if x.dtype != torch.float32 or grad_out.dtype != torch.float32:
        return None
    x, go, style = x.contiguous(), grad_out.contiguous(), style.contiguous()
    b, cin, h, w = x.shape
    cout = weight.numel() // cin
    splits = lib().fmgan_torgb_backward_splits(b, cin, h * w)
    if splits == 0 or cout > 4:
        return None
    gx = torch.empty_like(x)
    mpart = torch.empty((splits, b, cout, cin), dtype=torch.float32, device=x.device)
    wgt = weight.contiguous()
    with on_device(x) as stream:
        tok = _observer.begin('torgb_backward', (b, cin, cout, h * w))
        st = lib().fmgan_torgb_backward_f32(fp(x), fp(go), fp(wgt), fp(style), fp(gx), fp(mpart), b, cin, cout, h * w,
                                            float(scale), stream)
        _observer.end(tok)
    if st == -2:
        return None
    check(st, 'torgb_backward')
    return gx, (mpart.sum(0) if splits > 1 else mpart[0])


def images_to_tensor(images, mean=0.5, std=0.5):
    """uint8 [B,H,W,3] (GPU) -> float32 [B,3,H,W] = ((images/255) - mean)/std: ToTensor + Normalize in one pass."""
    require_gpu(images, 'images')
    if images.dtype != torch.uint8 or images.ndim != 4 or images.shape[-1] != 3:
        raise RuntimeError('images_to_tensor: expected a uint8 [B,H,W,3] tensor')
    x = images.contiguous()
    b, h, w, _ = x.shape
    out = torch.empty((b, 3, h, w), dtype=torch.float32, device=x.device)
    with on_device(x) as stream:
        check(lib().fmgan_images_to_tensor(ptr(x), ptr(out), b, h, w, float(mean), float(std), stream), 'images_to_tensor')
    return out


def resize_output_size(h, w, size):
    """torchvision Resize(int) rule: (out_h, out_w) for an [h, w] image (host logic, no GPU needed)."""
    oh, ow = ctypes.c_int(), ctypes.c_int()
    check(lib().fmgan_resize_output_size(int(h), int(w), int(size), ctypes.byref(oh), ctypes.byref(ow)), 'resize_output_size')
    return oh.value, ow.value


def resize_plan(in_h, in_w, out_h, out_w):
    """Pillow's bilinear coefficient tables for this size pair as a CPU int32 tensor (host logic, no GPU needed)."""
    n = lib().fmgan_resize_plan_ints(int(in_h), int(in_w), int(out_h), int(out_w))
    if n <= 0:
        raise RuntimeError('resize_plan: invalid sizes')
    plan = torch.empty(n, dtype=torch.int32)
    check(lib().fmgan_resize_plan(int(in_h), int(in_w), int(out_h), int(out_w), plan.data_ptr(), n), 'resize_plan')
    return plan


_PLANS = {}


def resize_images(images, out_h, out_w, to_tensor=False, mean=0.5, std=0.5):
    """uint8 [B,H,W,3] (GPU) -> PIL-exact bilinear resize.  to_tensor=False: uint8 [B,out_h,out_w,3];
    to_tensor=True: float32 [B,3,out_h,out_w] = ((v/255) - mean)/std (Resize + ToTensor + Normalize in one pass)."""
    require_gpu(images, 'images')
    if images.dtype != torch.uint8 or images.ndim != 4 or images.shape[-1] != 3:
        raise RuntimeError('resize_images: expected a uint8 [B,H,W,3] tensor')
    x = images.contiguous()
    b, h, w, _ = x.shape
    key = (x.device, h, w, out_h, out_w)
    plan = _PLANS.get(key)
    if plan is None:
        plan = _PLANS[key] = resize_plan(h, w, out_h, out_w).to(x.device)
    if to_tensor:
        out = torch.empty((b, 3, out_h, out_w), dtype=torch.float32, device=x.device)
        o8, o32 = None, out
    else:
        out = torch.empty((b, out_h, out_w, 3), dtype=torch.uint8, device=x.device)
        o8, o32 = out, None
    with on_device(x) as stream:
        tok = _observer.begin('resize', (b, h, w, out_h, out_w))
        check(lib().fmgan_resize_bilinear_u8(ptr(x), ptr(plan), ptr(o8), ptr(o32), b, h, w, out_h, out_w, float(mean),
                                             float(std), stream), 'resize_bilinear_u8')
        _observer.end(tok)
    return out


def tensor_to_images(tensor, cent=1.0, factor=255.0 / 2.0):
    """float32 [B,3,H,W] (GPU) -> uint8 [B,H,W,3] = uint8((clip(t,-1,1)+cent)*factor): tensor2im for the whole batch."""
    require_gpu(tensor, 'tensor')
    if tensor.dtype != torch.float32 or tensor.ndim != 4 or tensor.shape[1] != 3:
        raise RuntimeError('tensor_to_images: expected a float32 [B,3,H,W] tensor')
    x = tensor.contiguous()
    b, _, h, w = x.shape
    out = torch.empty((b, h, w, 3), dtype=torch.uint8, device=x.device)
    with on_device(x) as stream:
        check(lib().fmgan_tensor_to_images(ptr(x), ptr(out), b, h, w, float(cent), float(factor), stream), 'tensor_to_images')
    return out
