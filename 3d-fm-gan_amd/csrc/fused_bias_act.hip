// fused_bias_act for gfx950: y = act'(x + bias[c]; ref) * scale, elementwise, HBM-bound.
//
// Replaces fused_bias_act_op / fused_bias_act_kernel (op/fused_bias_act_kernel.cu:18-49, 52-99).
// The reference spends an integer div + mod per 4-byte element and moves 4 B per lane per
// instruction.  Here: 16 B per lane (global_load/store_dwordx4), and the bias index is either
// per (batch*channel) plane (one scalar per block, no per-element division) or one division
// per float4.  fmgan_noise_bias_act_f32 additionally folds NoiseInjection's broadcast add into
// the same pass (stylegan2.py:307-312 + 360-376), removing one read+write of the activation.
#include "common.h"

namespace {

template <typename A> __device__ __forceinline__ A act_apply(A x, A ref, int code, A alpha) {
  // code = act*10 + grad  (op/fused_bias_act_kernel.cu:36-45)
  switch (code) {
    case 12: return A(0);
    case 30: return (x > A(0)) ? x : x * alpha;
    case 31: return (ref > A(0)) ? x : x * alpha;
    case 32: return A(0);
    default: return x;  // 10, 11 and anything else: linear
  }
}

// generic: any dtype, any step_b
template <typename T>
__global__ __launch_bounds__(256) void fba_generic(T* __restrict__ out, const T* __restrict__ x,
                                                   const T* __restrict__ b, const T* __restrict__ ref,
                                                   long long size_x, int step_b, int size_b, int code, float alpha,
                                                   float scale) {
  using A = typename AccT<T>::type;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < size_x; i += stride) {
    A v = to_acc<T>(x[i]);
    if (b) v += to_acc<T>(b[(i / step_b) % size_b]);
    const A r = ref ? to_acc<T>(ref[i]) : A(0);
    out[i] = from_acc<T>(act_apply<A>(v, r, code, (A)alpha) * (A)scale);
  }
}

// f32, step_b % 4 == 0: each float4 lies inside one (batch, channel) plane.
// grid.y walks planes (bias is a per-block scalar), grid.x strides the plane.
template <int CODE, bool HAS_REF>
__global__ __launch_bounds__(256) void fba_f32_planes(float* __restrict__ out, const float* __restrict__ x,
                                                      const float* __restrict__ b, const float* __restrict__ ref,
                                                      int planes, int step4, int size_b, float alpha, float scale) {
  for (int pl = blockIdx.y; pl < planes; pl += gridDim.y) {
    const float bias = b ? b[pl % size_b] : 0.f;
    const long long base = (long long)pl * step4;
    const f32x4* x4 = reinterpret_cast<const f32x4*>(x) + base;
    const f32x4* r4 = reinterpret_cast<const f32x4*>(ref) + base;
    f32x4* o4 = reinterpret_cast<f32x4*>(out) + base;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < step4; i += gridDim.x * blockDim.x) {
      f32x4 v = x4[i];
      f32x4 r = {0.f, 0.f, 0.f, 0.f};
      if constexpr (HAS_REF) r = r4[i];
      f32x4 y;
      y.x = act_apply<float>(v.x + bias, r.x, CODE, alpha) * scale;
      y.y = act_apply<float>(v.y + bias, r.y, CODE, alpha) * scale;
      y.z = act_apply<float>(v.z + bias, r.z, CODE, alpha) * scale;
      y.w = act_apply<float>(v.w + bias, r.w, CODE, alpha) * scale;
      o4[i] = y;
    }
  }
}

// f32 flat float4 variant: no bias, or small planes with step_b % 4 == 0 (one 32-bit division per float4).
template <int CODE, bool HAS_REF>
__global__ __launch_bounds__(256) void fba_f32_flat4(float* __restrict__ out, const float* __restrict__ x,
                                                     const float* __restrict__ b, const float* __restrict__ ref,
                                                     unsigned n4, unsigned step4, unsigned size_b, float alpha,
                                                     float scale) {
  const f32x4* x4 = reinterpret_cast<const f32x4*>(x);
  const f32x4* r4 = reinterpret_cast<const f32x4*>(ref);
  f32x4* o4 = reinterpret_cast<f32x4*>(out);
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += gridDim.x * blockDim.x) {
    const float bias = b ? b[(i / step4) % size_b] : 0.f;
    f32x4 v = x4[i];
    f32x4 r = {0.f, 0.f, 0.f, 0.f};
    if constexpr (HAS_REF) r = r4[i];
    f32x4 y;
    y.x = act_apply<float>(v.x + bias, r.x, CODE, alpha) * scale;
    y.y = act_apply<float>(v.y + bias, r.y, CODE, alpha) * scale;
    y.z = act_apply<float>(v.z + bias, r.z, CODE, alpha) * scale;
    y.w = act_apply<float>(v.w + bias, r.w, CODE, alpha) * scale;
    o4[i] = y;
  }
}

// f32, step_b == 1 and size_b % 4 == 0: the bias runs along the innermost dimension ([rows, channels] — the NHWC
// activations of the pSp encoder's style heads).  One float4 of bias per float4 of data, one 32-bit modulo per float4
// (fba_generic: a 64-bit div + mod and 4 bytes per lane per element — 59 us per call on [32768, 512], 8 % of the 256^2
// step at B=32).
template <int CODE, bool HAS_REF>
__global__ __launch_bounds__(256) void fba_f32_inner4(float* __restrict__ out, const float* __restrict__ x,
                                                      const float* __restrict__ b, const float* __restrict__ ref,
                                                      unsigned n4, unsigned size_b4, float alpha, float scale) {
  const f32x4* x4 = reinterpret_cast<const f32x4*>(x);
  const f32x4* r4 = reinterpret_cast<const f32x4*>(ref);
  const f32x4* b4 = reinterpret_cast<const f32x4*>(b);
  f32x4* o4 = reinterpret_cast<f32x4*>(out);
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += gridDim.x * blockDim.x) {
    const f32x4 bias = b4[i % size_b4];
    f32x4 v = x4[i];
    f32x4 r = {0.f, 0.f, 0.f, 0.f};
    if constexpr (HAS_REF) r = r4[i];
    f32x4 y;
    y.x = act_apply<float>(v.x + bias.x, r.x, CODE, alpha) * scale;
    y.y = act_apply<float>(v.y + bias.y, r.y, CODE, alpha) * scale;
    y.z = act_apply<float>(v.z + bias.z, r.z, CODE, alpha) * scale;
    y.w = act_apply<float>(v.w + bias.w, r.w, CODE, alpha) * scale;
    o4[i] = y;
  }
}

// (x + w*noise) + bias with the reference's roundings: torch computes `image + weight * noise`
// as a separate multiply and add (stylegan2.py:312), then the bias add (fused_bias_act_kernel.cu:27)
// — no fused multiply-add anywhere, so none here.
__device__ __forceinline__ float nadd(float x, float nw, float n, float bv) {
  return __fadd_rn(__fadd_rn(x, __fmul_rn(nw, n)), bv);
}

// NoiseInjection + bias + leaky ReLU * scale in one pass.  grid.y = batch*channel planes.
__global__ __launch_bounds__(256) void noise_bias_act_f32(float* __restrict__ out, const float* __restrict__ x,
                                                          const float* __restrict__ noise,
                                                          const float* __restrict__ noise_weight,
                                                          const float* __restrict__ bias, int planes, int channel,
                                                          int hw, int noise_batch, float alpha, float scale) {
  const float nw = (noise && noise_weight) ? noise_weight[0] : 0.f;
  const bool vec = (hw & 3) == 0;
  for (int pl = blockIdx.y; pl < planes; pl += gridDim.y) {
    const int c = pl % channel, bidx = pl / channel;
    const float bv = bias ? bias[c] : 0.f;
    const float* xp = x + (long long)pl * hw;
    float* op = out + (long long)pl * hw;
    const float* np = noise ? noise + (long long)(noise_batch == 1 ? 0 : bidx) * hw : nullptr;
    if (vec) {
      const int hw4 = hw >> 2;
      for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < hw4; i += gridDim.x * blockDim.x) {
        f32x4 v = reinterpret_cast<const f32x4*>(xp)[i];
        f32x4 n = {0.f, 0.f, 0.f, 0.f};
        if (np) n = reinterpret_cast<const f32x4*>(np)[i];
        f32x4 y;
        // same association as the reference: (x + w*noise) + bias
        y.x = act_apply<float>(nadd(v.x, nw, n.x, bv), 0.f, 30, alpha) * scale;
        y.y = act_apply<float>(nadd(v.y, nw, n.y, bv), 0.f, 30, alpha) * scale;
        y.z = act_apply<float>(nadd(v.z, nw, n.z, bv), 0.f, 30, alpha) * scale;
        y.w = act_apply<float>(nadd(v.w, nw, n.w, bv), 0.f, 30, alpha) * scale;
        reinterpret_cast<f32x4*>(op)[i] = y;
      }
    } else {
      for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < hw; i += gridDim.x * blockDim.x) {
        const float n = np ? np[i] : 0.f;
        op[i] = act_apply<float>(nadd(xp[i], nw, n, bv), 0.f, 30, alpha) * scale;
      }
    }
  }
}

template <typename T>
int launch_generic(const void* x, const void* b, const void* ref, void* out, long long size_x, int size_b, int step_b,
                   int code, float alpha, float scale, hipStream_t s) {
  long long blocks = (size_x + 255) / 256;
  const long long cap = (long long)FMGAN_NUM_CU * 32;
  if (blocks > cap) blocks = cap;
  hipLaunchKernelGGL(fba_generic<T>, dim3((unsigned)blocks), dim3(256), 0, s, (T*)out, (const T*)x, (const T*)b,
                     (const T*)ref, size_x, step_b, size_b, code, alpha, scale);
  return fmgan_check_launch();
}

template <int CODE, bool HAS_REF>
int launch_f32_inner(const float* x, const float* b, const float* ref, float* out, long long size_x, int size_b,
                     float alpha, float scale, hipStream_t s) {
  const unsigned n4 = (unsigned)(size_x / 4);
  long long blocks = ((long long)n4 + 255) / 256;
  const long long cap = (long long)FMGAN_NUM_CU * 32;
  if (blocks > cap) blocks = cap;
  hipLaunchKernelGGL((fba_f32_inner4<CODE, HAS_REF>), dim3((unsigned)blocks), dim3(256), 0, s, out, x, b, ref, n4,
                     (unsigned)(size_b / 4), alpha, scale);
  return fmgan_check_launch();
}

template <int CODE, bool HAS_REF>
int launch_f32_fast(const float* x, const float* b, const float* ref, float* out, long long size_x, int size_b,
                    int step_b, float alpha, float scale, hipStream_t s) {
  const int step4 = step_b / 4;
  const long long planes = size_x / step_b;
  if (b && step4 >= 1024 && planes <= 0x7fffffffLL) {
    // big planes: one bias scalar per block, no per-element index maths
    int gx = (step4 + 255) / 256;
    if (gx > 64) gx = 64;
    long long gy = planes;
    const long long cap = (long long)FMGAN_NUM_CU * 32 / gx;
    if (gy > cap) gy = cap > 0 ? cap : 1;
    if (gy > 65535) gy = 65535;
    hipLaunchKernelGGL((fba_f32_planes<CODE, HAS_REF>), dim3(gx, (unsigned)gy), dim3(256), 0, s, out, x, b, ref,
                       (int)planes, step4, size_b, alpha, scale);
  } else {
    const unsigned n4 = (unsigned)(size_x / 4);
    long long blocks = ((long long)n4 + 255) / 256;
    const long long cap = (long long)FMGAN_NUM_CU * 32;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL((fba_f32_flat4<CODE, HAS_REF>), dim3((unsigned)blocks), dim3(256), 0, s, out, x, b, ref, n4,
                       (unsigned)(b ? step4 : 1), (unsigned)(b ? size_b : 1), alpha, scale);
  }
  return fmgan_check_launch();
}

}  // namespace

extern "C" int fmgan_fused_bias_act(int dtype, const void* x, const void* bias, const void* refer, void* out,
                                    long long size_x, int size_b, int step_b, int act, int grad, float alpha,
                                    float scale, void* stream) {
  if (dtype != FMGAN_F32 && dtype != FMGAN_F64 && dtype != FMGAN_F16) return FMGAN_EUNSUPPORTED;
  if (size_x < 0 || size_b < 0 || step_b <= 0) return FMGAN_EINVAL;
  if (size_x == 0) return FMGAN_OK;
  if (!x || !out) return FMGAN_EINVAL;
  if (size_b == 0) bias = nullptr;
  if (!bias) size_b = 1;
  const int code = act * 10 + grad;
  hipStream_t s = (hipStream_t)stream;
  const bool fast_code = (code == 30 || code == 31 || code == 10 || code == 11);
  const bool aligned = ((((uintptr_t)x) | ((uintptr_t)out) | ((uintptr_t)refer)) & 15) == 0;
  if (dtype == FMGAN_F32 && fast_code && aligned && (size_x % 4) == 0 && size_x / 4 < 0xffffffffLL &&
      (!bias || step_b % 4 == 0)) {
    const float* xf = (const float*)x; const float* bf = (const float*)bias; const float* rf = (const float*)refer;
    float* of = (float*)out;
    if (code == 30) return launch_f32_fast<30, false>(xf, bf, nullptr, of, size_x, size_b, step_b, alpha, scale, s);
    if (code == 31) {
      if (rf) return launch_f32_fast<31, true>(xf, bf, rf, of, size_x, size_b, step_b, alpha, scale, s);
      return launch_f32_fast<31, false>(xf, bf, nullptr, of, size_x, size_b, step_b, alpha, scale, s);
    }
    return launch_f32_fast<10, false>(xf, bf, nullptr, of, size_x, size_b, step_b, alpha, scale, s);
  }
  if (dtype == FMGAN_F32 && fast_code && aligned && bias && step_b == 1 && (size_b % 4) == 0 && (size_x % size_b) == 0 &&
      size_x / 4 < 0xffffffffLL && (((uintptr_t)bias) & 15) == 0) {
    const float* xf = (const float*)x; const float* bf = (const float*)bias; const float* rf = (const float*)refer;
    float* of = (float*)out;
    if (code == 30) return launch_f32_inner<30, false>(xf, bf, nullptr, of, size_x, size_b, alpha, scale, s);
    if (code == 31) {
      if (rf) return launch_f32_inner<31, true>(xf, bf, rf, of, size_x, size_b, alpha, scale, s);
      return launch_f32_inner<31, false>(xf, bf, nullptr, of, size_x, size_b, alpha, scale, s);
    }
    return launch_f32_inner<10, false>(xf, bf, nullptr, of, size_x, size_b, alpha, scale, s);
  }
  if (dtype == FMGAN_F32) return launch_generic<float>(x, bias, refer, out, size_x, size_b, step_b, code, alpha, scale, s);
  if (dtype == FMGAN_F64) return launch_generic<double>(x, bias, refer, out, size_x, size_b, step_b, code, alpha, scale, s);
  return launch_generic<__half>(x, bias, refer, out, size_x, size_b, step_b, code, alpha, scale, s);
}

// ---------------------------------------------------------------- activation backward + bias-gradient partials
// FusedLeakyReLUFunctionBackward (op/fused_act.py:29-50) is `grad_input = fused_bias_act(grad_output, empty, out, 3, 1,
// ...)` followed by `grad_input.sum(dims)` for the bias: a second kernel that reads the whole gradient again (667 torch
// reduce launches, 30 ms per 4 training iterations at 256^2).  Here the pass that writes grad_input also leaves one
// partial sum per (plane, block): the bias gradient is then a [B, C, blocks] -> [C] sum over a few KB.  Fixed
// association (per-lane serial, wave butterfly, 4 waves in order): bit-reproducible.
__global__ __launch_bounds__(256) void fba_bwd_bias_f32(float* __restrict__ gi, float* __restrict__ partial,
                                                        const float* __restrict__ g, const float* __restrict__ ref,
                                                        int planes, int step4, float alpha, float scale) {
  __shared__ float red[4];
  for (int pl = blockIdx.y; pl < planes; pl += gridDim.y) {
    const long long base = (long long)pl * step4;
    const f32x4* g4 = reinterpret_cast<const f32x4*>(g) + base;
    const f32x4* r4 = reinterpret_cast<const f32x4*>(ref) + base;
    f32x4* o4 = reinterpret_cast<f32x4*>(gi) + base;
    float acc = 0.f;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < step4; i += gridDim.x * blockDim.x) {
      const f32x4 v = g4[i];
      const f32x4 r = r4[i];
      f32x4 y;
      y.x = act_apply<float>(v.x, r.x, 31, alpha) * scale;
      y.y = act_apply<float>(v.y, r.y, 31, alpha) * scale;
      y.z = act_apply<float>(v.z, r.z, 31, alpha) * scale;
      y.w = act_apply<float>(v.w, r.w, 31, alpha) * scale;
      o4[i] = y;
      acc += (y.x + y.y) + (y.z + y.w);
    }
    for (int m = 32; m > 0; m >>= 1) acc += __shfl_xor(acc, m, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[(long long)pl * gridDim.x + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
    __syncthreads();
  }
}

// blocks per plane the backward pass uses for planes of `hw` elements (= columns of the partial array); 0: shape not
// served (hw % 4 != 0 or tiny planes) -> the caller keeps the two-step form
extern "C" int fmgan_fused_bias_act_bwd_blocks(long long planes, int hw) {
  if (planes <= 0 || hw < 64 || (hw & 3) || planes > 0x7fffffffLL) return 0;
  int gx = ((hw >> 2) + 1023) / 1024;          // >= 4 float4 per lane
  if (gx > 64) gx = 64;
  return gx;
}

extern "C" int fmgan_fused_bias_act_bwd_f32(const float* grad_out, const float* ref_out, float* grad_in, float* partial,
                                            long long planes, int hw, float alpha, float scale, void* stream) {
  if (planes < 0 || hw <= 0) return FMGAN_EINVAL;
  if (planes == 0) return FMGAN_OK;
  if (!grad_out || !ref_out || !grad_in || !partial) return FMGAN_EINVAL;
  const int gx = fmgan_fused_bias_act_bwd_blocks(planes, hw);
  if (gx == 0 || ((((uintptr_t)grad_out) | ((uintptr_t)ref_out) | ((uintptr_t)grad_in)) & 15) != 0) return FMGAN_EUNSUPPORTED;
  long long gy = planes;
  const long long cap = (long long)FMGAN_NUM_CU * 32 / gx;
  if (gy > cap) gy = cap > 0 ? cap : 1;
  if (gy > 65535) gy = 65535;
  hipLaunchKernelGGL(fba_bwd_bias_f32, dim3(gx, (unsigned)gy), dim3(256), 0, (hipStream_t)stream, grad_in, partial,
                     grad_out, ref_out, (int)planes, hw >> 2, alpha, scale);
  return fmgan_check_launch();
}

extern "C" int fmgan_noise_bias_act_f32(const float* x, const float* noise, const float* noise_weight,
                                        const float* bias, float* out, int batch, int channel, int hw,
                                        int noise_batch, float alpha, float scale, void* stream) {
  if (batch < 0 || channel <= 0 || hw <= 0) return FMGAN_EINVAL;
  if (batch == 0) return FMGAN_OK;
  if (!x || !out) return FMGAN_EINVAL;
  if (noise && noise_batch != 1 && noise_batch != batch) return FMGAN_EINVAL;
  const long long planes = (long long)batch * channel;
  if (planes > 0x7fffffffLL) return FMGAN_EOVERFLOW;
  const bool vec = (hw & 3) == 0 && ((((uintptr_t)x) | ((uintptr_t)out) | ((uintptr_t)noise)) & 15) == 0;
  (void)vec;
  int gx = ((hw + 3) / 4 + 255) / 256;
  if (gx > 64) gx = 64;
  long long gy = planes;
  const long long cap = (long long)FMGAN_NUM_CU * 32 / gx;
  if (gy > cap) gy = cap > 0 ? cap : 1;
  if (gy > 65535) gy = 65535;
  hipLaunchKernelGGL(noise_bias_act_f32, dim3(gx, (unsigned)gy), dim3(256), 0, (hipStream_t)stream, out, x, noise,
                     noise_weight, bias, (int)planes, channel, hw, noise_batch, alpha, scale);
  return fmgan_check_launch();
}


// ---------------------------------------------------------------- PReLU backward (channels innermost)
// The pSp encoder's PReLU(depth) units (psp_encoder_model/encoders/helpers.py:107,130) on NHWC activations seen as
// [rows = N*H*W, channels]:  gx = g * (x > 0 ? 1 : a[c]),  ga[c] = sum_rows g * x * (x <= 0).
// aten's prelu_backward writes TWO full-size tensors (grad_input and the per-element weight-gradient terms) with a
// multi-output elementwise kernel that does not vectorise on this layout (965 us per call at [16,64,256,256], 1.1 TB/s,
// 6.5 % of a forward+backward step), then reduces the second one.  Here: read x and g once, write gx once, keep the
// channel sums in registers over the block's rows, reduce across the block's row lanes in LDS and write ONE partial row
// per block; the caller sums the [blocks, channels] partials (deterministic, unlike atomics).
__global__ __launch_bounds__(256) void prelu_bwd_nhwc_f32(const float* __restrict__ x, const float* __restrict__ g,
                                                          const float* __restrict__ slope, float* __restrict__ gx,
                                                          float* __restrict__ partial, long long rows, int channels,
                                                          int q_log2) {
  __shared__ f32x4 red[256];
  const int Q = 1 << q_log2;                 // float4 groups per row handled by one block column (channels/4 <= 256)
  const int tid = threadIdx.x;
  const int q = tid & (Q - 1), r = tid >> q_log2, R = 256 >> q_log2;
  const int c4 = blockIdx.y * Q + q;         // float4 index inside a row
  const int nq = channels >> 2;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  if (c4 < nq) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(slope + 4 * c4);
    for (long long row = (long long)blockIdx.x * R + r; row < rows; row += (long long)gridDim.x * R) {
      const long long o = row * nq + c4;
      const f32x4 xv = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(x) + o);
      const f32x4 gv = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(g) + o);
      f32x4 out;
      out.x = xv.x > 0.f ? gv.x : gv.x * a.x; out.y = xv.y > 0.f ? gv.y : gv.y * a.y;
      out.z = xv.z > 0.f ? gv.z : gv.z * a.z; out.w = xv.w > 0.f ? gv.w : gv.w * a.w;
      acc.x += xv.x > 0.f ? 0.f : gv.x * xv.x; acc.y += xv.y > 0.f ? 0.f : gv.y * xv.y;
      acc.z += xv.z > 0.f ? 0.f : gv.z * xv.z; acc.w += xv.w > 0.f ? 0.f : gv.w * xv.w;
      reinterpret_cast<f32x4*>(gx)[o] = out;
    }
  }
  red[tid] = acc;
  __syncthreads();
  for (int s = R >> 1; s > 0; s >>= 1) {       // fixed tree over the block's row lanes: deterministic
    if (r < s) {
      const f32x4 o = red[tid + (s << q_log2)];
      f32x4 m = red[tid];
      m.x += o.x; m.y += o.y; m.z += o.z; m.w += o.w;
      red[tid] = m;
    }
    __syncthreads();
  }
  if (r == 0 && c4 < nq) reinterpret_cast<f32x4*>(partial + (long long)blockIdx.x * channels)[c4] = red[tid];
}

extern "C" int fmgan_prelu_backward_blocks(long long rows, int channels) {
  if (rows <= 0 || channels <= 0) return 0;
  int q = 1; while (q < (channels >> 2) && q < 256) q <<= 1;
  const int R = 256 / q;
  long long b = (rows + R - 1) / R;
  const long long cap = (long long)FMGAN_NUM_CU * 8;
  if (b > cap) b = cap;
  return (int)(b < 1 ? 1 : b);
}

extern "C" int fmgan_prelu_backward_f32(const float* x, const float* grad, const float* slope, float* grad_x,
                                        float* partial, long long rows, int channels, void* stream) {
  if (rows < 0 || channels <= 0 || (channels & 3)) return channels > 0 && (channels & 3) ? FMGAN_EUNSUPPORTED : FMGAN_EINVAL;
  if (rows == 0) return FMGAN_OK;
  if (!x || !grad || !slope || !grad_x || !partial) return FMGAN_EINVAL;
  if (((((uintptr_t)x) | ((uintptr_t)grad) | ((uintptr_t)grad_x) | ((uintptr_t)slope) | ((uintptr_t)partial)) & 15) != 0)
    return FMGAN_EUNSUPPORTED;
  int q_log2 = 0; while ((1 << q_log2) < (channels >> 2) && q_log2 < 8) ++q_log2;
  const int Q = 1 << q_log2;
  const unsigned gy = (unsigned)(((channels >> 2) + Q - 1) / Q);
  const int gx = fmgan_prelu_backward_blocks(rows, channels);
  hipLaunchKernelGGL(prelu_bwd_nhwc_f32, dim3((unsigned)gx, gy), dim3(256), 0, (hipStream_t)stream, x, grad, slope, grad_x,
                     partial, rows, channels, q_log2);
  return fmgan_check_launch();
}
