// The steps either side of the hot path, on the GPU:
//   in : uint8 HWC image batch -> float CHW in [-1,1]   == transforms.ToTensor() + Normalize(0.5, 0.5)
//        (train_3_encoder.py:233-239; Resize(size) is the identity for the 256^2 datasets the reference uses)
//   out: float CHW in [-1,1] -> uint8 HWC               == tensor2im (Evaluation/visual_eval.py:24-38)
// Both are HBM-bound layout changes: one pass, 3 channels gathered/scattered per pixel, rounding identical to
// the reference's numpy/torch arithmetic so the tests can demand bit equality.
#include "common.h"

namespace {

// out[b,c,y,x] = ((in[b,y,x,c] / 255) - mean) / std   with the reference's operation order (div, sub, div)
__global__ __launch_bounds__(256) void u8hwc_to_f32chw(const unsigned char* __restrict__ in, float* __restrict__ out,
                                                       long long pixels, int hw, float mean, float stdv) {
  for (long long p = (long long)blockIdx.x * 256 + threadIdx.x; p < pixels; p += (long long)gridDim.x * 256) {
    const long long b = p / hw;
    const int q = (int)(p - b * hw);
    const unsigned char* src = in + p * 3;
    float* dst = out + b * 3 * hw + q;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float t = __fdiv_rn((float)src[c], 255.0f);
      dst[(long long)c * hw] = __fdiv_rn(__fsub_rn(t, mean), stdv);
    }
  }
}

// out[b,y,x,c] = (uint8)((clip(in[b,c,y,x], -1, 1) + cent) * factor)   (astype(np.uint8): truncation)
__global__ __launch_bounds__(256) void f32chw_to_u8hwc(const float* __restrict__ in, unsigned char* __restrict__ out,
                                                       long long pixels, int hw, float cent, float factor) {
  for (long long p = (long long)blockIdx.x * 256 + threadIdx.x; p < pixels; p += (long long)gridDim.x * 256) {
    const long long b = p / hw;
    const int q = (int)(p - b * hw);
    const float* src = in + b * 3 * hw + q;
    unsigned char* dst = out + p * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      float v = src[(long long)c * hw];
      v = fminf(fmaxf(v, -1.0f), 1.0f);
      v = __fmul_rn(__fadd_rn(v, cent), factor);
      dst[c] = (unsigned char)(int)v;
    }
  }
}

// ---- PIL-exact bilinear resize (+ optional ToTensor/Normalize), tables from resize_plan.cpp
constexpr int RZ_TY = 8, RZ_TX = 32;   // output tile of one block

__device__ __forceinline__ int rz_clip8(int v) {
  v >>= 22;
  return v < 0 ? 0 : (v > 255 ? 255 : v);
}

// Pass 1: the input rows the tile needs, resampled along x into LDS as uint8 (Pillow rounds between the passes);
// pass 2: along y out of LDS.  Integer arithmetic end to end: bit-exact with Pillow's ImagingResample.
__global__ __launch_bounds__(256) void resize_bilinear_u8_k(const unsigned char* __restrict__ in,
                                                            const int* __restrict__ plan,
                                                            unsigned char* __restrict__ out_u8,
                                                            float* __restrict__ out_f32, int H, int W, int oh, int ow,
                                                            int ksx, int ksy, float mean, float stdv) {
  extern __shared__ unsigned char hbuf[];   // [rows][RZ_TX][3]
  const int* bx = plan + 8;
  const int* kx = bx + 2 * ow;
  const int* by = kx + (long long)ow * ksx;
  const int* ky = by + 2 * oh;
  const int tid = threadIdx.x;
  const int x0 = blockIdx.x * RZ_TX, y0 = blockIdx.y * RZ_TY;
  const long long b = blockIdx.z;
  const int ylast = min(y0 + RZ_TY, oh) - 1;
  const int r0 = by[2 * y0];
  const int nrows = by[2 * ylast] + by[2 * ylast + 1] - r0;
  const unsigned char* img = in + b * H * W * 3;
  for (int e = tid; e < nrows * RZ_TX * 3; e += 256) {
    const int row = e / (RZ_TX * 3), rem = e - row * (RZ_TX * 3);
    const int xl = rem / 3, c = rem - xl * 3;
    const int x = x0 + xl;
    int v = 0;
    if (x < ow) {
      const int first = bx[2 * x], n = bx[2 * x + 1];
      const unsigned char* src = img + ((long long)(r0 + row) * W + first) * 3 + c;
      const int* k = kx + (long long)x * ksx;
      int ss = 1 << 21;
      for (int t = 0; t < n; ++t) ss += (int)src[3 * t] * k[t];
      v = rz_clip8(ss);
    }
    hbuf[e] = (unsigned char)v;
  }
  __syncthreads();
  for (int e = tid; e < RZ_TY * RZ_TX * 3; e += 256) {
    // x fastest within a channel plane: coalesced float stores (the uint8 HWC stores of 3 planes interleave in L2)
    const int c = e / (RZ_TY * RZ_TX), rem = e - c * (RZ_TY * RZ_TX);
    const int yl = rem / RZ_TX, xl = rem - yl * RZ_TX;
    const int y = y0 + yl, x = x0 + xl;
    if (y >= oh || x >= ow) continue;
    const int first = by[2 * y] - r0, n = by[2 * y + 1];
    const int* k = ky + (long long)y * ksy;
    int ss = 1 << 21;
    for (int t = 0; t < n; ++t) ss += (int)hbuf[((first + t) * RZ_TX + xl) * 3 + c] * k[t];
    const int v = rz_clip8(ss);
    if (out_u8) out_u8[((b * oh + y) * ow + x) * 3 + c] = (unsigned char)v;
    if (out_f32) out_f32[((b * 3 + c) * oh + y) * ow + x] = __fdiv_rn(__fsub_rn(__fdiv_rn((float)v, 255.0f), mean), stdv);
  }
}

// Fast variant for <= KMAX horizontal taps (shrink factors up to (KMAX-1)/2: 9 taps = the 1024^2 -> 256^2 case).
// The byte-at-a-time kernel above is instruction-bound (27 byte loads per intermediate pixel); here a thread owns one
// output column: its tap window (<= KMAX pixels = 3*KMAX bytes) is fetched as whole dwords and re-aligned to the
// window's first byte with v_alignbyte, so every tap byte sits at a compile-time position; the x coefficients stay in
// registers for all rows of the tile.  The intermediate is kept as one packed RGB dword per pixel in LDS, so the
// vertical pass reads one dword per tap.  Same integer arithmetic, same bits.
template <int KMAX>
__global__ __launch_bounds__(256) void resize_bilinear_u8_fast(const unsigned char* __restrict__ in,
                                                               const int* __restrict__ plan,
                                                               unsigned char* __restrict__ out_u8,
                                                               float* __restrict__ out_f32, int H, int W, int oh, int ow,
                                                               int ksx, int ksy, long long total_bytes, float mean,
                                                               float stdv) {
  extern __shared__ unsigned int hpix[];    // [rows][RZ_TX] packed R | G<<8 | B<<16
  constexpr int ND = (3 * KMAX + 3 + 3) / 4;   // dwords covering a window that starts at any byte phase
  const int* bx = plan + 8;
  const int* kx = bx + 2 * ow;
  const int* by = kx + (long long)ow * ksx;
  const int* ky = by + 2 * oh;
  const int tid = threadIdx.x;
  const int xl = tid & (RZ_TX - 1), rl = tid / RZ_TX;            // 32 columns x 8 row lanes
  const int x0 = blockIdx.x * RZ_TX, y0 = blockIdx.y * RZ_TY;
  const long long b = blockIdx.z;
  const int ylast = min(y0 + RZ_TY, oh) - 1;
  const int r0 = by[2 * y0];
  const int nrows = by[2 * ylast] + by[2 * ylast + 1] - r0;
  const int x = x0 + xl;
  const bool xin = x < ow;
  const int first = xin ? bx[2 * x] : 0, n = xin ? bx[2 * x + 1] : 0;
  int k[KMAX];
#pragma unroll
  for (int t = 0; t < KMAX; ++t) k[t] = (t < n) ? kx[(long long)x * ksx + t] : 0;
  for (int row = rl; row < nrows; row += 256 / RZ_TX) {
    const long long byte0 = ((b * H + r0 + row) * W + first) * 3;
    const int mis = (int)(byte0 & 3);
    const long long a0 = byte0 - mis;
    unsigned int w[ND + 1];
#pragma unroll
    for (int i = 0; i <= ND; ++i) {
      const long long off = a0 + 4 * i;
      unsigned int v = 0;
      if (off + 4 <= total_bytes) {
        v = *reinterpret_cast<const unsigned int*>(in + off);
      } else {
        for (int j = 0; j < 4; ++j)
          if (off + j < total_bytes) v |= (unsigned int)in[off + j] << (8 * j);
      }
      w[i] = v;
    }
    unsigned int a[ND];
#pragma unroll
    for (int i = 0; i < ND; ++i) a[i] = __builtin_amdgcn_alignbyte(w[i + 1], w[i], mis);
    int s0 = 1 << 21, s1 = 1 << 21, s2 = 1 << 21;
#pragma unroll
    for (int t = 0; t < KMAX; ++t) {
      const int j0 = 3 * t, j1 = 3 * t + 1, j2 = 3 * t + 2;
      s0 += __mul24((int)((a[j0 >> 2] >> (8 * (j0 & 3))) & 255u), k[t]);
      s1 += __mul24((int)((a[j1 >> 2] >> (8 * (j1 & 3))) & 255u), k[t]);
      s2 += __mul24((int)((a[j2 >> 2] >> (8 * (j2 & 3))) & 255u), k[t]);
    }
    hpix[row * RZ_TX + xl] = xin ? (unsigned)rz_clip8(s0) | ((unsigned)rz_clip8(s1) << 8) | ((unsigned)rz_clip8(s2) << 16) : 0u;
  }
  __syncthreads();
  const int y = y0 + rl;
  if (y >= oh || !xin) return;
  const int vfirst = by[2 * y] - r0, vn = by[2 * y + 1];
  const int* kv = ky + (long long)y * ksy;
  int s0 = 1 << 21, s1 = 1 << 21, s2 = 1 << 21;
  for (int t = 0; t < vn; ++t) {
    const unsigned int p = hpix[(vfirst + t) * RZ_TX + xl];
    const int c = kv[t];
    s0 += __mul24((int)(p & 255u), c);
    s1 += __mul24((int)((p >> 8) & 255u), c);
    s2 += __mul24((int)((p >> 16) & 255u), c);
  }
  const int v[3] = {rz_clip8(s0), rz_clip8(s1), rz_clip8(s2)};
  if (out_u8) {
    unsigned char* d = out_u8 + ((b * oh + y) * ow + x) * 3;
    d[0] = (unsigned char)v[0]; d[1] = (unsigned char)v[1]; d[2] = (unsigned char)v[2];
  }
  if (out_f32) {
#pragma unroll
    for (int c = 0; c < 3; ++c)
      out_f32[((b * 3 + c) * oh + y) * ow + x] = __fdiv_rn(__fsub_rn(__fdiv_rn((float)v[c], 255.0f), mean), stdv);
  }
}

inline unsigned grid_for(long long n) {
  long long b = (n + 255) / 256;
  const long long cap = (long long)FMGAN_NUM_CU * 32;
  return (unsigned)(b > cap ? cap : (b < 1 ? 1 : b));
}

}  // namespace

extern "C" int fmgan_images_to_tensor(const unsigned char* in, float* out, int batch, int h, int w, float mean,
                                      float stdv, void* stream) {
  if (batch < 0 || h <= 0 || w <= 0 || stdv == 0.f) return FMGAN_EINVAL;
  if (batch == 0) return FMGAN_OK;
  if (!in || !out) return FMGAN_EINVAL;
  const long long pixels = (long long)batch * h * w;
  hipLaunchKernelGGL(u8hwc_to_f32chw, dim3(grid_for(pixels)), dim3(256), 0, (hipStream_t)stream, in, out, pixels, h * w,
                     mean, stdv);
  return fmgan_check_launch();
}

extern "C" int fmgan_resize_bilinear_u8(const unsigned char* in, const int* plan, unsigned char* out_u8, float* out_f32,
                                        int batch, int in_h, int in_w, int out_h, int out_w, float mean, float stdv,
                                        void* stream) {
  if (batch < 0 || in_h <= 0 || in_w <= 0 || out_h <= 0 || out_w <= 0) return FMGAN_EINVAL;
  if (out_f32 && stdv == 0.f) return FMGAN_EINVAL;
  if (batch == 0) return FMGAN_OK;
  if (!in || !plan || (!out_u8 && !out_f32)) return FMGAN_EINVAL;
  if (batch > 65535 || (out_h + RZ_TY - 1) / RZ_TY > 65535) return FMGAN_EOVERFLOW;
  // header of the plan, recomputed on the host (the plan itself lives in device memory)
  const long long ints = fmgan_resize_plan_ints(in_h, in_w, out_h, out_w);
  if (ints <= 0) return FMGAN_EINVAL;
  const double sx = (double)in_w / out_w, sy = (double)in_h / out_h;
  const int ksx = (int)ceil(sx < 1.0 ? 1.0 : sx) * 2 + 1, ksy = (int)ceil(sy < 1.0 ? 1.0 : sy) * 2 + 1;
  // rows of one tile: RZ_TY output rows span at most RZ_TY*scale + ksy input rows
  const long long span = (long long)ceil(RZ_TY * sy) + ksy + 1;
  const dim3 grid((out_w + RZ_TX - 1) / RZ_TX, (out_h + RZ_TY - 1) / RZ_TY, batch);
  const long long total_bytes = (long long)batch * in_h * in_w * 3;
  const size_t lds4 = (size_t)span * RZ_TX * 4;
  hipStream_t hs = (hipStream_t)stream;
  if (ksx <= 9 && lds4 <= 64 * 1024 && (((uintptr_t)in) & 3) == 0) {
    if (ksx <= 3)
      hipLaunchKernelGGL(resize_bilinear_u8_fast<3>, grid, dim3(256), lds4, hs, in, plan, out_u8, out_f32, in_h, in_w,
                         out_h, out_w, ksx, ksy, total_bytes, mean, stdv);
    else if (ksx <= 5)
      hipLaunchKernelGGL(resize_bilinear_u8_fast<5>, grid, dim3(256), lds4, hs, in, plan, out_u8, out_f32, in_h, in_w,
                         out_h, out_w, ksx, ksy, total_bytes, mean, stdv);
    else
      hipLaunchKernelGGL(resize_bilinear_u8_fast<9>, grid, dim3(256), lds4, hs, in, plan, out_u8, out_f32, in_h, in_w,
                         out_h, out_w, ksx, ksy, total_bytes, mean, stdv);
    return fmgan_check_launch();
  }
  const long long lds = span * RZ_TX * 3;
  if (lds > 64 * 1024) return FMGAN_EUNSUPPORTED;    // shrinking by more than ~80x
  hipLaunchKernelGGL(resize_bilinear_u8_k, grid, dim3(256), (size_t)lds, (hipStream_t)stream, in, plan, out_u8, out_f32,
                     in_h, in_w, out_h, out_w, ksx, ksy, mean, stdv);
  return fmgan_check_launch();
}

extern "C" int fmgan_tensor_to_images(const float* in, unsigned char* out, int batch, int h, int w, float cent,
                                      float factor, void* stream) {
  if (batch < 0 || h <= 0 || w <= 0) return FMGAN_EINVAL;
  if (batch == 0) return FMGAN_OK;
  if (!in || !out) return FMGAN_EINVAL;
  const long long pixels = (long long)batch * h * w;
  hipLaunchKernelGGL(f32chw_to_u8hwc, dim3(grid_for(pixels)), dim3(256), 0, (hipStream_t)stream, in, out, pixels, h * w,
                     cent, factor);
  return fmgan_check_launch();
}
