// Winograd F(2x2, 3x3) transforms for the plain 3x3 modulated conv (stride 1, pad 1) — fp32.
//
// Same operator as fmgan_modconv2d_f32 mode 0 (/root/reference/stylegan2.py:250-298, restated input-modulated):
//     out[b,o,y,x] = demod[b,o] * sum_{i,ky,kx} (scale*W[o,i,ky,kx]) * (style[b,i] * in[b,i,y+ky-1,x+kx-1])   (+ fused epilogue)
// as  Y = At [ (G g Gt) . (Bt d B) ] A  per 2x2 output tile (Lavin & Gray): 16 products per tile instead of 36, i.e. 2.25x fewer
// MACs on layers whose direct form already runs at 0.84-0.88 of the fp32 matrix peak (32^2..128^2, 256-512 channels).
// Three steps, the middle one a plain batched GEMM the caller hands to the BLAS library (torch.bmm -> hipBLASLt, fp32):
//     fmgan_wino_weight_f32:  U[xi][o][i]        = (G g Gt)[xi]  of the scaled weight                    [16, cout, cin]
//     fmgan_wino_input_f32:   V[xi][i][b*T + t]  = (Bt (style[b,i] * d) B)[xi] of tile t's 4x4 input window  [16, cin, B*T]
//     (caller)                M[xi]              = U[xi] @ V[xi]                                        [16, cout, B*T]
//     fmgan_wino_output_f32:  out tile           = epilogue( demod * At M A )                             [B, cout, H, W]
// HBM-bound: the input transform writes 4x the input, the output transform reads 4x the output; measured against copies of
// those bytes plus the 16 GEMMs (profiles/r03_winograd.md) the form pays at 32^2..128^2 and not at 256^2 and above.
// The sums are re-associated: results differ from the direct kernel by fp32 rounding (the GPU tests hold it to the
// direct kernel's tolerance against the CPU restatement of the reference).  T = (H/2) * (W/2) tiles per sample, H and W even.
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void wino_weight_f32(const float* __restrict__ wt, float* __restrict__ u, int cin, int cout) {
  const long long total = (long long)cin * cout;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    const int i = (int)(idx % cin), o = (int)(idx / cin);           // consecutive threads: consecutive i (the write order)
    float g[3][3];
#pragma unroll
    for (int t = 0; t < 9; ++t) g[t / 3][t % 3] = wt[((long long)i * 9 + t) * cout + o];
    float tmp[4][3];      // G g,  G = [1 0 0; 1/2 1/2 1/2; 1/2 -1/2 1/2; 0 0 1]
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      tmp[0][k] = g[0][k];
      tmp[1][k] = 0.5f * (g[0][k] + g[1][k] + g[2][k]);
      tmp[2][k] = 0.5f * (g[0][k] - g[1][k] + g[2][k]);
      tmp[3][k] = g[2][k];
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const float uu[4] = {tmp[a][0], 0.5f * (tmp[a][0] + tmp[a][1] + tmp[a][2]), 0.5f * (tmp[a][0] - tmp[a][1] + tmp[a][2]),
                           tmp[a][2]};
#pragma unroll
      for (int bx = 0; bx < 4; ++bx) u[((long long)(a * 4 + bx) * cout + o) * cin + i] = uu[bx];
    }
  }
}

// one thread per (sample, channel, tile): 16 loads of the zero-padded 4x4 window, 32 adds, 16 coalesced stores
__global__ __launch_bounds__(256) void wino_input_f32(const float* __restrict__ x, const float* __restrict__ style,
                                                      float* __restrict__ v, int batch, int c, int h, int w) {
  const int tw = w >> 1, th = h >> 1, T = tw * th;
  const long long n_cols = (long long)batch * T, total = n_cols * c;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    const int t = (int)(idx % T);
    const int ch = (int)((idx / T) % c);
    const int b = (int)(idx / ((long long)T * c));
    const int ty = t / tw, tx = t - ty * tw;
    const float* xp = x + ((long long)b * c + ch) * h * w;
    const float s = style[(long long)b * c + ch];
    float d[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int y = 2 * ty - 1 + r;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int xx = 2 * tx - 1 + q;
        d[r][q] = (y >= 0 && y < h && xx >= 0 && xx < w) ? xp[(long long)y * w + xx] * s : 0.f;
      }
    }
    // Bt d B,  Bt = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]
    float e[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      e[r][0] = d[r][0] - d[r][2]; e[r][1] = d[r][1] + d[r][2]; e[r][2] = d[r][2] - d[r][1]; e[r][3] = d[r][1] - d[r][3];
    }
    const long long col = (long long)b * T + t;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float v0 = e[0][q] - e[2][q], v1 = e[1][q] + e[2][q], v2 = e[2][q] - e[1][q], v3 = e[1][q] - e[3][q];
      v[((long long)(0 * 4 + q) * c + ch) * n_cols + col] = v0;
      v[((long long)(1 * 4 + q) * c + ch) * n_cols + col] = v1;
      v[((long long)(2 * 4 + q) * c + ch) * n_cols + col] = v2;
      v[((long long)(3 * 4 + q) * c + ch) * n_cols + col] = v3;
    }
  }
}

// one thread per (sample, output channel, tile): 16 coalesced loads, At M A, the StyledConv epilogue, two 8-byte stores
__global__ __launch_bounds__(256) void wino_output_f32(const float* __restrict__ m, const float* __restrict__ demod,
                                                       const float* __restrict__ noise, const float* __restrict__ noise_weight,
                                                       const float* __restrict__ bias, float* __restrict__ out, int batch,
                                                       int cout, int h, int w, int noise_batch, int fuse_act, float alpha,
                                                       float act_scale) {
  const int tw = w >> 1, th = h >> 1, T = tw * th;
  const long long n_cols = (long long)batch * T, total = n_cols * cout;
  const float nw = (fuse_act && noise && noise_weight) ? noise_weight[0] : 0.f;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    const int t = (int)(idx % T);
    const int o = (int)((idx / T) % cout);
    const int b = (int)(idx / ((long long)T * cout));
    const int ty = t / tw, tx = t - ty * tw;
    const long long col = (long long)b * T + t;
    float mm[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int q = 0; q < 4; ++q) mm[a][q] = m[((long long)(a * 4 + q) * cout + o) * n_cols + col];
    // At M A,  At = [1 1 1 0; 0 1 -1 -1]
    float z[2][4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      z[0][q] = mm[0][q] + mm[1][q] + mm[2][q];
      z[1][q] = mm[1][q] - mm[2][q] - mm[3][q];
    }
    const float dm = demod ? demod[(long long)b * cout + o] : 1.f;
    const float bv = (fuse_act && bias) ? bias[o] : 0.f;
    float* op = out + (((long long)b * cout + o) * h + 2 * ty) * w + 2 * tx;
    const float* np = (fuse_act && noise) ? noise + ((long long)(noise_batch == 1 ? 0 : b) * h + 2 * ty) * w + 2 * tx : nullptr;
#pragma unroll
    for (int jy = 0; jy < 2; ++jy) {
      float y0 = (z[jy][0] + z[jy][1] + z[jy][2]) * dm, y1 = (z[jy][1] - z[jy][2] - z[jy][3]) * dm;
      if (fuse_act) {
        const float n0 = np ? __fmul_rn(nw, np[jy * w]) : 0.f, n1 = np ? __fmul_rn(nw, np[jy * w + 1]) : 0.f;
        y0 = __fadd_rn(__fadd_rn(y0, n0), bv);
        y1 = __fadd_rn(__fadd_rn(y1, n1), bv);
        y0 = (y0 > 0.f ? y0 : y0 * alpha) * act_scale;
        y1 = (y1 > 0.f ? y1 : y1 * alpha) * act_scale;
      }
      f32x2_u st; st.x = y0; st.y = y1;
      *reinterpret_cast<f32x2_u*>(op + (long long)jy * w) = st;
    }
  }
}

inline unsigned wino_grid(long long total) {
  long long blocks = (total + 255) / 256;
  const long long cap = (long long)FMGAN_NUM_CU * 32;
  if (blocks > cap) blocks = cap;
  return (unsigned)(blocks < 1 ? 1 : blocks);
}

}  // namespace

extern "C" int fmgan_wino_weight_f32(const float* wt, float* u, int cin, int cout, void* stream) {
  if (cin <= 0 || cout <= 0) return FMGAN_EINVAL;
  if (!wt || !u) return FMGAN_EINVAL;
  hipLaunchKernelGGL(wino_weight_f32, dim3(wino_grid((long long)cin * cout)), dim3(256), 0, (hipStream_t)stream, wt, u, cin, cout);
  return fmgan_check_launch();
}

extern "C" int fmgan_wino_input_f32(const float* x, const float* style, float* v, int batch, int c, int h, int w, void* stream) {
  if (batch < 0 || c <= 0 || h <= 0 || w <= 0 || (h & 1) || (w & 1)) return FMGAN_EINVAL;
  if (batch == 0) return FMGAN_OK;
  if (!x || !style || !v) return FMGAN_EINVAL;
  hipLaunchKernelGGL(wino_input_f32, dim3(wino_grid((long long)batch * c * (h / 2) * (w / 2))), dim3(256), 0, (hipStream_t)stream,
                     x, style, v, batch, c, h, w);
  return fmgan_check_launch();
}

extern "C" int fmgan_wino_output_f32(const float* m, const float* demod, const float* noise, const float* noise_weight,
                                     const float* bias, float* out, int batch, int cout, int h, int w, int noise_batch,
                                     int fuse_act, float alpha, float act_scale, void* stream) {
  if (batch < 0 || cout <= 0 || h <= 0 || w <= 0 || (h & 1) || (w & 1)) return FMGAN_EINVAL;
  if (batch == 0) return FMGAN_OK;
  if (!m || !out) return FMGAN_EINVAL;
  if (fuse_act && noise && noise_batch != 1 && noise_batch != batch) return FMGAN_EINVAL;
  hipLaunchKernelGGL(wino_output_f32, dim3(wino_grid((long long)batch * cout * (h / 2) * (w / 2))), dim3(256), 0,
                     (hipStream_t)stream, m, demod, noise, noise_weight, bias, out, batch, cout, h, w, noise_batch, fuse_act, alpha,
                     act_scale);
  return fmgan_check_launch();
}
