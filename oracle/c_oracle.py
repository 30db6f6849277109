"""ctypes/numpy front end of oracle/libfmgan_oracle.so (the plain-C restatement, fmgan_oracle.c).

TEST INFRASTRUCTURE ONLY (checker and CPU baseline); never imported by the product package."""
import ctypes
import os
import subprocess

import numpy as np

_DIR = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_DIR, 'libfmgan_oracle.so')
_lib = None


def build():
    subprocess.check_call(['make', '-C', _DIR, '-s'])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = ctypes.CDLL(_LIB_PATH)
    return _lib


def num_threads():
    return int(lib().oracle_num_threads())


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def upfirdn2d(x, kernel, up=(1, 1), down=(1, 1), pad=(0, 0, 0, 0)):
    """x [major,in_h,in_w,minor] float32/float64; kernel [kh,kw]; up=(up_x,up_y); down=(down_x,down_y);
    pad=(pad_x0,pad_x1,pad_y0,pad_y1).  Returns [major,out_h,out_w,minor]."""
    dt = x.dtype
    assert dt in (np.float32, np.float64)
    x = np.ascontiguousarray(x)
    kernel = np.ascontiguousarray(kernel, dtype=dt)
    major, in_h, in_w, minor = x.shape
    kh, kw = kernel.shape
    up_x, up_y = up
    down_x, down_y = down
    px0, px1, py0, py1 = pad
    out_h = (in_h * up_y + py0 + py1 - kh) // down_y + 1
    out_w = (in_w * up_x + px0 + px1 - kw) // down_x + 1
    out = np.empty((major, out_h, out_w, minor), dtype=dt)
    fn = lib().oracle_upfirdn2d_f32 if dt == np.float32 else lib().oracle_upfirdn2d_f64
    st = fn(_p(x), _p(kernel), _p(out), major, in_h, in_w, minor, kh, kw, up_x, up_y, down_x, down_y, px0, px1, py0, py1)
    if st != 0:
        raise RuntimeError(f'oracle_upfirdn2d status {st}')
    return out


def fused_bias_act(x, bias=None, ref=None, act=3, grad=0, alpha=0.2, scale=2 ** 0.5):
    dt = x.dtype
    assert dt in (np.float32, np.float64)
    x = np.ascontiguousarray(x)
    out = np.empty_like(x)
    step_b = 1
    for d in x.shape[2:]:
        step_b *= d
    size_b = 0
    if bias is not None:
        bias = np.ascontiguousarray(bias, dtype=dt)
        size_b = bias.size
    if ref is not None:
        ref = np.ascontiguousarray(ref, dtype=dt)
    if dt == np.float32:
        fn, ct = lib().oracle_fused_bias_act_f32, ctypes.c_float
    else:
        fn, ct = lib().oracle_fused_bias_act_f64, ctypes.c_double
    fn.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_longlong, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ct, ct]
    st = fn(_p(x), _p(bias), _p(ref), _p(out), x.size, size_b, step_b, act, grad, alpha, scale)
    if st != 0:
        raise RuntimeError(f'oracle_fused_bias_act status {st}')
    return out


def modulated_conv2d(x, weight, style, mode=0, demodulate=True):
    """x [B,Cin,H,W] f32; weight [Cout,Cin,k,k]; style [B,Cin] (output of the modulation linear).
    mode 0 plain / 1 transposed stride 2 / 2 stride-2 valid conv.  scale = 1/sqrt(Cin*k*k)."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    weight = np.ascontiguousarray(weight, dtype=np.float32)
    style = np.ascontiguousarray(style, dtype=np.float32)
    b, cin, h, w = x.shape
    cout, _, k, _ = weight.shape
    if mode == 0:
        oh, ow = h, w
    elif mode == 1:
        oh, ow = (h - 1) * 2 + k, (w - 1) * 2 + k
    else:
        oh, ow = (h - k) // 2 + 1, (w - k) // 2 + 1
    out = np.empty((b, cout, oh, ow), dtype=np.float32)
    fn = lib().oracle_modulated_conv2d_f32
    fn.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int] * 8 + [ctypes.c_float]
    st = fn(_p(x), _p(weight), _p(style), _p(out), b, cin, cout, h, w, k, mode, int(demodulate),
            1.0 / np.sqrt(cin * k * k))
    if st != 0:
        raise RuntimeError(f'oracle_modulated_conv2d status {st}')
    return out


def to_rgb(x, weight, style, bias=None, skip=None):
    """x [B,Cin,H,W]; weight [Cout,Cin]; style [B,Cin]; bias [Cout]; skip [B,Cout,H,W] (already upsampled)."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    weight = np.ascontiguousarray(weight, dtype=np.float32)
    style = np.ascontiguousarray(style, dtype=np.float32)
    b, cin, h, w = x.shape
    cout = weight.shape[0]
    if bias is not None:
        bias = np.ascontiguousarray(bias, dtype=np.float32)
    if skip is not None:
        skip = np.ascontiguousarray(skip, dtype=np.float32)
    out = np.empty((b, cout, h, w), dtype=np.float32)
    fn = lib().oracle_to_rgb_f32
    fn.argtypes = [ctypes.c_void_p] * 6 + [ctypes.c_int] * 4 + [ctypes.c_float]
    st = fn(_p(x), _p(weight), _p(style), _p(bias), _p(skip), _p(out), b, cin, cout, h * w, 1.0 / np.sqrt(cin))
    if st != 0:
        raise RuntimeError(f'oracle_to_rgb status {st}')
    return out


def resize_bilinear_u8(img, out_h, out_w):
    """img [H,W,C] uint8 -> [out_h,out_w,C] uint8, Pillow's Image.resize(BILINEAR) arithmetic."""
    img = np.ascontiguousarray(img, dtype=np.uint8)
    h, w, c = img.shape
    out = np.empty((out_h, out_w, c), dtype=np.uint8)
    fn = lib().oracle_resize_bilinear_u8
    fn.argtypes = [ctypes.c_void_p] * 2 + [ctypes.c_int] * 5
    st = fn(_p(img), _p(out), h, w, c, out_h, out_w)
    if st != 0:
        raise RuntimeError(f'oracle_resize_bilinear_u8 status {st}')
    return out


def resized_output_size(h, w, size):
    """torchvision.transforms.Resize(int) on a PIL image: the shorter edge becomes `size`, the longer
    int(size * long / short); unchanged when the shorter edge already equals size."""
    short, long_ = (w, h) if w <= h else (h, w)
    if short == size:
        return h, w
    new_short, new_long = size, int(size * long_ / short)
    return (new_long, new_short) if w <= h else (new_short, new_long)
