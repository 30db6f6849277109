// ModulatedConv2d / ToRGB for gfx950 (MI355X).
//
// Replaces the reference's per-sample weight materialisation + grouped conv
// (stylegan2.py:250-298: `weight = scale*W*style; weight *= demod; F.conv2d(..., groups=batch)` /
// `F.conv_transpose2d(..., stride=2, groups=batch)`) and ToRGB (stylegan2.py:389-404).
//
// Design.  The reference builds a [B*Cout, Cin, 3, 3] weight per call (302 MB at B=32, 512 ch) and
// runs B independent small GEMMs.  Here the modulation moves to the input and the demodulation to
// the epilogue, so all samples share ONE weight matrix and the whole batch is a single implicit GEMM
//     out[b,o,p] = demod[b,o] * sum_{i,tap} wt[i,tap,o] * (style[b,i] * in[b,i,p+tap])
// with M = Cout, N = B*H*W, K = Cin*9, contracted on v_mfma_f32_32x32x2_f32 (exact fp32; gfx950 has no
// reduced-precision f32 path and bf16 cannot hold the 1e-5 parity bar — SURVEY.md §7 "Hard parts").
//   * A operand (weights): wt[i][tap][o] staged to LDS as [kc][tap][o] — a 32-lane read is one bank row.
//   * B operand (pixels): a (TH+2)x(TW+2) halo patch per channel is staged once per 8-channel chunk,
//     multiplied by style[b,i] on the way in; the 9 taps are 9 constant LDS offsets from one base.
//   * 4 waves per block, 2-4 independent 32x32 accumulators per wave (64-cycle MFMA issue needs no
//     more), 2-3 blocks per CU so one block's staging overlaps another's MFMAs.
//   * Transposed (upsampling) conv = 4 output phases of the same contraction; a block owns one row
//     parity and both column parities so each lane stores the two adjacent columns as one 8-byte store.
//   * Tiny layers (4x4 .. 8x8) pack several samples into one pixel tile.
// Demodulation: one wave per output channel, sum over Cin by wave-shuffle butterfly.
#include "common.h"

namespace {

// ------------------------------------------------------------------ demod
__global__ __launch_bounds__(256) void modconv_demod_f32(const float* __restrict__ W,
                                                         const float* __restrict__ style,
                                                         float* __restrict__ demod, int batch, int cout, int cin,
                                                         int ktaps, float scale, float eps) {
  const int lane = threadIdx.x & 63;
  const int o = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (o >= cout) return;  // wave-uniform
  const float* wo = W + (long long)o * cin * ktaps;
  constexpr int MAXJ = 8;
  const bool cached = cin <= 64 * MAXJ;
  float wsq[MAXJ];
  if (cached) {
#pragma unroll
    for (int j = 0; j < MAXJ; ++j) {
      const int i = lane + 64 * j;
      float q = 0.f;
      if (i < cin)
        for (int t = 0; t < ktaps; ++t) { const float w = wo[i * ktaps + t]; q = fmaf(w, w, q); }
      wsq[j] = q;
    }
  }
  for (int b = 0; b < batch; ++b) {
    const float* sb = style + (long long)b * cin;
    float acc = 0.f;
    if (cached) {
#pragma unroll
      for (int j = 0; j < MAXJ; ++j) {
        const int i = lane + 64 * j;
        if (i < cin) { const float m = sb[i]; acc = fmaf(wsq[j], m * m, acc); }
      }
    } else {
      for (int i = lane; i < cin; i += 64) {
        float q = 0.f;
        for (int t = 0; t < ktaps; ++t) { const float w = wo[i * ktaps + t]; q = fmaf(w, w, q); }
        const float m = sb[i];
        acc = fmaf(q, m * m, acc);
      }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if (lane == 0) demod[(long long)b * cout + o] = 1.0f / sqrtf(scale * scale * acc + eps);
  }
}

// ------------------------------------------------------------------ weight prep: wt[i][t][o] = scale*W[o][i][t]
__global__ __launch_bounds__(256) void modconv_weight_prep_f32(const float* __restrict__ W, float* __restrict__ wt,
                                                               int cout, int cin, int ktaps, float scale) {
  const long long total = (long long)cout * cin * ktaps;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int o = (int)(idx % cout);
    const long long r = idx / cout;  // i*ktaps + t
    wt[idx] = scale * W[(long long)o * cin * ktaps + r];
  }
}

// ------------------------------------------------------------------ MFMA conv
struct MCParams {
  const float* in; const float* wt; const float* style; const float* demod; float* out;
  int batch, cin, cout, h, w, oh, ow;
  int th, nb;                        // tile rows per sample, samples per tile
  int tiles_x, tiles_y, tiles_b, o_tiles;
  const float* noise; const float* noise_weight; const float* bias;
  int noise_batch, fuse_act; float alpha, act_scale;
};

constexpr int MC_KC = 8;  // input channels per LDS chunk

template <int MODE, int RM, int RN, int WM, int WN, int TW>
__global__ __launch_bounds__(256) void modconv_mfma_f32(const MCParams p) {
  constexpr int KC = MC_KC;
  constexpr int BM = 32 * RM * WM;
  constexpr int NPX = MODE == 1 ? 2 : 1;
  constexpr int RNP = RN / NPX;  // 32-position groups per wave
  constexpr int PWP = TW + 2;
  static_assert(WM * WN == 4, "4 waves per block");
  static_assert(RN % NPX == 0, "RN must cover both column phases");
  extern __shared__ float smem[];
  float* Ws = smem;                 // [KC][9][BM]
  float* Xs = smem + KC * 9 * BM;   // [nb][KC][PH][PWP]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int l31 = lane & 31, khalf = lane >> 5;

  const unsigned lb = xcd_remap(blockIdx.x, gridDim.x);
  const int o_tile = lb % p.o_tiles;
  unsigned pt = lb / p.o_tiles;
  const int tx_i = pt % p.tiles_x; pt /= p.tiles_x;
  const int ty_i = pt % p.tiles_y;
  const int tb_i = pt / p.tiles_y;
  const int py = MODE == 1 ? (int)blockIdx.y : 0;
  const int o0 = o_tile * BM, x0 = tx_i * TW, y0 = ty_i * p.th, b0 = tb_i * p.nb;
  const int PH = p.th + 2;
  const int plane = PH * PWP;
  const int samp = KC * plane;

  // this lane's position in each of the wave's 32-position groups
  int pbase[RNP], pos_b[RNP], pos_y[RNP], pos_x[RNP];
#pragma unroll
  for (int g = 0; g < RNP; ++g) {
    const int pos = (wn * RNP + g) * 32 + l31;
    const int tx = pos % TW, r = pos / TW;
    const int ty = r % p.th, nbi = r / p.th;
    pbase[g] = nbi * samp + ty * PWP + tx;
    pos_b[g] = b0 + nbi; pos_y[g] = y0 + ty; pos_x[g] = x0 + tx;
  }

  f32x16 acc[RM][RN];
#pragma unroll
  for (int a = 0; a < RM; ++a)
#pragma unroll
    for (int b = 0; b < RN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  const int xs_total = p.nb * samp;
  const bool wvec = (p.cout & 3) == 0;

  for (int i0 = 0; i0 < p.cin; i0 += KC) {
    __syncthreads();
    // ---- stage weights  Ws[(kc*9+t)*BM + o]
    for (int idx = tid; idx < KC * 9 * (BM / 4); idx += 256) {
      const int row = idx / (BM / 4), c4 = idx % (BM / 4);
      const int i = i0 + row / 9;
      const int o = o0 + c4 * 4;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (i < p.cin) {
        const float* src = p.wt + ((long long)i * 9 + row % 9) * p.cout + o;
        if (wvec && o + 3 < p.cout) {
          v = *reinterpret_cast<const f32x4*>(src);
        } else {
          if (o + 0 < p.cout) v.x = src[0];
          if (o + 1 < p.cout) v.y = src[1];
          if (o + 2 < p.cout) v.z = src[2];
          if (o + 3 < p.cout) v.w = src[3];
        }
      }
      *reinterpret_cast<f32x4*>(Ws + row * BM + c4 * 4) = v;
    }
    // ---- stage modulated input patch  Xs[nb][kc][r][c], origin (y0-1, x0-1)
    for (int idx = tid; idx < xs_total; idx += 256) {
      const int c = idx % PWP;
      int t = idx / PWP;
      const int r = t % PH; t /= PH;
      const int kc = t % KC, nbi = t / KC;
      const int b = b0 + nbi, i = i0 + kc, y = y0 + r - 1, x = x0 + c - 1;
      float v = 0.f;
      if (b < p.batch && i < p.cin && y >= 0 && y < p.h && x >= 0 && x < p.w) {
        const long long ch = (long long)b * p.cin + i;
        v = p.in[(ch * p.h + y) * p.w + x] * p.style[ch];
      }
      Xs[idx] = v;
    }
    __syncthreads();
    // ---- contract
#pragma unroll 2
    for (int kk = 0; kk < KC / 2; ++kk) {
      const int kc = 2 * kk + khalf;
      const float* wrow = Ws + kc * 9 * BM + wm * 32 * RM + l31;
      const float* xrow = Xs + kc * plane;
      if constexpr (MODE == 0) {
#pragma unroll
        for (int t = 0; t < 9; ++t) {
          float a[RM], bv[RN];
#pragma unroll
          for (int m = 0; m < RM; ++m) a[m] = wrow[t * BM + m * 32];
#pragma unroll
          for (int g = 0; g < RN; ++g) bv[g] = xrow[pbase[g] + (t / 3) * PWP + (t % 3)];
#pragma unroll
          for (int m = 0; m < RM; ++m)
#pragma unroll
            for (int g = 0; g < RN; ++g)
              acc[m][g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m], bv[g], acc[m][g], 0, 0, 0);
        }
      } else {
        // out[Y=2m+py, X=2n+px] += w[ky][kx] * in[m - ky/2, n - kx/2],  ky = py (mod 2), kx = px (mod 2)
        // patch offsets: ro = 1 - ky/2, co = 1 - kx/2
        auto step = [&](int tap, int ro, int co, int px) {
          float a[RM];
#pragma unroll
          for (int m = 0; m < RM; ++m) a[m] = wrow[tap * BM + m * 32];
#pragma unroll
          for (int g = 0; g < RNP; ++g) {
            const float bvv = xrow[pbase[g] + ro * PWP + co];
#pragma unroll
            for (int m = 0; m < RM; ++m)
              acc[m][g * 2 + px] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m], bvv, acc[m][g * 2 + px], 0, 0, 0);
          }
        };
        if (py == 0) {
          step(0, 1, 1, 0); step(2, 1, 0, 0); step(6, 0, 1, 0); step(8, 0, 0, 0);  // (ky,kx) = 00 02 20 22
          step(1, 1, 1, 1); step(7, 0, 1, 1);                                      // 01 21
        } else {
          step(3, 1, 1, 0); step(5, 1, 0, 0);                                      // 10 12
          step(4, 1, 1, 1);                                                        // 11
        }
      }
    }
  }

  // ---- epilogue
  const float nw = (p.fuse_act && p.noise && p.noise_weight) ? p.noise_weight[0] : 0.f;
#pragma unroll
  for (int m = 0; m < RM; ++m) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int o = o0 + wm * 32 * RM + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
      if (o >= p.cout) continue;
#pragma unroll
      for (int g = 0; g < RNP; ++g) {
        const int b = pos_b[g];
        if (b >= p.batch) continue;
        const float d = p.demod ? p.demod[(long long)b * p.cout + o] : 1.f;
        if constexpr (MODE == 0) {
          const int y = pos_y[g], x = pos_x[g];
          if (y >= p.h || x >= p.w) continue;
          float v = acc[m][g][r] * d;
          if (p.fuse_act) {
            const float n = p.noise ? p.noise[(long long)(p.noise_batch == 1 ? 0 : b) * p.h * p.w + y * p.w + x] : 0.f;
            const float bv = p.bias ? p.bias[o] : 0.f;
            v = __fadd_rn(__fadd_rn(v, __fmul_rn(nw, n)), bv);
            v = (v > 0.f ? v : v * p.alpha) * p.act_scale;
          }
          p.out[(((long long)b * p.cout + o) * p.h + y) * p.w + x] = v;
        } else {
          const int Y = 2 * pos_y[g] + py, X = 2 * pos_x[g];
          if (Y >= p.oh || X >= p.ow) continue;
          float* dst = p.out + (((long long)b * p.cout + o) * p.oh + Y) * p.ow + X;
          const float v0 = acc[m][g * 2][r] * d, v1 = acc[m][g * 2 + 1][r] * d;
          if (X + 1 < p.ow) {
            f32x2_u t; t.x = v0; t.y = v1;
            *reinterpret_cast<f32x2_u*>(dst) = t;
          } else {
            dst[0] = v0;
          }
        }
      }
    }
  }
}

template <int MODE, int RM, int RN, int WM, int WN, int TW>
int launch_cfg(MCParams& p, int gh, int gw, hipStream_t s) {
  constexpr int BM = 32 * RM * WM;
  constexpr int NPX = MODE == 1 ? 2 : 1;
  constexpr int BN = 32 * (RN / NPX) * WN;
  const int rows_total = BN / TW;
  int th = rows_total, nb = 1;
  if (gh < rows_total) {
    th = 1;
    while (th < gh) th <<= 1;
    nb = rows_total / th;
  }
  p.th = th; p.nb = nb;
  p.tiles_x = (gw + TW - 1) / TW;
  p.tiles_y = (gh + th - 1) / th;
  p.tiles_b = (p.batch + nb - 1) / nb;
  p.o_tiles = (p.cout + BM - 1) / BM;
  const long long blocks = (long long)p.o_tiles * p.tiles_x * p.tiles_y * p.tiles_b;
  if (blocks > 0x7fffffffLL) return FMGAN_EOVERFLOW;
  const size_t lds = sizeof(float) * ((size_t)MC_KC * 9 * BM + (size_t)nb * MC_KC * (th + 2) * (TW + 2));
  hipLaunchKernelGGL((modconv_mfma_f32<MODE, RM, RN, WM, WN, TW>), dim3((unsigned)blocks, MODE == 1 ? 2 : 1), dim3(256),
                     lds, s, p);
  return fmgan_check_launch();
}

template <int MODE, int RM, int RN, int WM, int WN>
int launch_tw(MCParams& p, int gh, int gw, hipStream_t s) {
  if (gw <= 4) return launch_cfg<MODE, RM, RN, WM, WN, 4>(p, gh, gw, s);
  if (gw <= 8) return launch_cfg<MODE, RM, RN, WM, WN, 8>(p, gh, gw, s);
  if (gw <= 16) return launch_cfg<MODE, RM, RN, WM, WN, 16>(p, gh, gw, s);
  return launch_cfg<MODE, RM, RN, WM, WN, 32>(p, gh, gw, s);
}

template <int MODE>
int launch_mode(MCParams& p, int gh, int gw, hipStream_t s) {
  if (p.cout >= 96) return launch_tw<MODE, 2, 2, 2, 2>(p, gh, gw, s);  // 128 x 128 (mode 1: 64 pos x 2 px)
  if (p.cout >= 48) return launch_tw<MODE, 2, 2, 1, 4>(p, gh, gw, s);  //  64 x 256
  return launch_tw<MODE, 1, 4, 1, 4>(p, gh, gw, s);                    //  32 x 512
}

// ------------------------------------------------------------------ ToRGB (1x1, <= 4 output channels, HBM-bound)
template <int VEC>
__global__ __launch_bounds__(256) void torgb_f32(const float* __restrict__ in, const float* __restrict__ weight,
                                                 const float* __restrict__ style, const float* __restrict__ bias,
                                                 const float* __restrict__ skip, float* __restrict__ out, int cin,
                                                 int cout, int hw, float scale) {
  extern __shared__ float ws[];  // [cout][cin]  scale*W*style for this sample
  const int b = blockIdx.y;
  for (int idx = threadIdx.x; idx < cout * cin; idx += 256)
    ws[idx] = scale * weight[idx] * style[(long long)b * cin + idx % cin];
  __syncthreads();
  const int n = hw / VEC;
  const float* inb = in + (long long)b * cin * hw;
  for (int pidx = blockIdx.x * 256 + threadIdx.x; pidx < n; pidx += gridDim.x * 256) {
    float acc[4][VEC];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int e = 0; e < VEC; ++e) acc[c][e] = 0.f;
#pragma unroll 8
    for (int i = 0; i < cin; ++i) {
      float v[VEC];
      if constexpr (VEC == 4) {
        const f32x4 t = reinterpret_cast<const f32x4*>(inb + (long long)i * hw)[pidx];
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
      } else {
        v[0] = inb[(long long)i * hw + pidx];
      }
#pragma unroll
      for (int c = 0; c < 4; ++c)
        if (c < cout) {
          const float wv = ws[c * cin + i];
#pragma unroll
          for (int e = 0; e < VEC; ++e) acc[c][e] = fmaf(wv, v[e], acc[c][e]);
        }
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      if (c >= cout) continue;
      const long long off = ((long long)b * cout + c) * hw + (long long)pidx * VEC;
      const float bv = bias ? bias[c] : 0.f;
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        float r = __fadd_rn(acc[c][e], bv);
        if (skip) r = __fadd_rn(r, skip[off + e]);
        acc[c][e] = r;
      }
      if constexpr (VEC == 4) {
        f32x4 t; t.x = acc[c][0]; t.y = acc[c][1]; t.z = acc[c][2]; t.w = acc[c][3];
        *reinterpret_cast<f32x4*>(out + off) = t;
      } else {
        out[off] = acc[c][0];
      }
    }
  }
}

}  // namespace

extern "C" int fmgan_modconv_demod_f32(const float* weight, const float* style, float* demod, int batch, int cout,
                                       int cin, int ktaps, float scale, float eps, void* stream) {
  if (batch < 0 || cout <= 0 || cin <= 0 || ktaps <= 0) return FMGAN_EINVAL;
  if (batch == 0) return FMGAN_OK;
  if (!weight || !style || !demod) return FMGAN_EINVAL;
  hipLaunchKernelGGL(modconv_demod_f32, dim3((cout + 3) / 4), dim3(256), 0, (hipStream_t)stream, weight, style, demod,
                     batch, cout, cin, ktaps, scale, eps);
  return fmgan_check_launch();
}

extern "C" int fmgan_modconv_weight_prep_f32(const float* weight, float* wt, int cout, int cin, int ktaps, float scale,
                                             void* stream) {
  if (cout <= 0 || cin <= 0 || ktaps <= 0) return FMGAN_EINVAL;
  if (!weight || !wt) return FMGAN_EINVAL;
  const long long total = (long long)cout * cin * ktaps;
  long long blocks = (total + 255) / 256;
  if (blocks > FMGAN_NUM_CU * 16) blocks = FMGAN_NUM_CU * 16;
  hipLaunchKernelGGL(modconv_weight_prep_f32, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, weight, wt,
                     cout, cin, ktaps, scale);
  return fmgan_check_launch();
}

extern "C" int fmgan_modconv2d_f32(const float* in, const float* wt, const float* style, const float* demod,
                                   float* out, int batch, int cin, int cout, int h, int w, int mode,
                                   const float* noise, const float* noise_weight, const float* bias, int noise_batch,
                                   int fuse_act, float alpha, float act_scale, void* stream) {
  if (batch < 0 || cin <= 0 || cout <= 0 || h <= 0 || w <= 0) return FMGAN_EINVAL;
  if (mode != 0 && mode != 1) return FMGAN_EUNSUPPORTED;
  if (mode == 1 && fuse_act) return FMGAN_EUNSUPPORTED;  // the blur sits between conv and activation
  if (batch == 0) return FMGAN_OK;
  if (!in || !wt || !style || !out) return FMGAN_EINVAL;
  if (fuse_act && noise && noise_batch != 1 && noise_batch != batch) return FMGAN_EINVAL;
  MCParams p{};
  p.in = in; p.wt = wt; p.style = style; p.demod = demod; p.out = out;
  p.batch = batch; p.cin = cin; p.cout = cout; p.h = h; p.w = w;
  p.oh = mode == 1 ? 2 * h + 1 : h;
  p.ow = mode == 1 ? 2 * w + 1 : w;
  p.noise = noise; p.noise_weight = noise_weight; p.bias = bias;
  p.noise_batch = noise_batch; p.fuse_act = fuse_act; p.alpha = alpha; p.act_scale = act_scale;
  if ((long long)batch * cout * p.oh * p.ow > (1LL << 40)) return FMGAN_EOVERFLOW;
  hipStream_t s = (hipStream_t)stream;
  if (mode == 0) return launch_mode<0>(p, h, w, s);
  return launch_mode<1>(p, h + 1, w + 1, s);  // position grid (m, n): Y = 2m+py, X = 2n+px
}

extern "C" int fmgan_torgb_f32(const float* in, const float* weight, const float* style, const float* bias,
                               const float* skip, float* out, int batch, int cin, int cout, int hw, float scale,
                               void* stream) {
  if (batch < 0 || cin <= 0 || cout <= 0 || cout > 4 || hw <= 0) return FMGAN_EINVAL;
  if (batch == 0) return FMGAN_OK;
  if (!in || !weight || !style || !out) return FMGAN_EINVAL;
  if (batch > 65535) return FMGAN_EOVERFLOW;
  const size_t lds = sizeof(float) * (size_t)cout * cin;
  if (lds > 64 * 1024) return FMGAN_EUNSUPPORTED;
  const bool vec = (hw & 3) == 0 && ((((uintptr_t)in) | ((uintptr_t)out) | ((uintptr_t)skip)) & 15) == 0;
  const int n = vec ? hw / 4 : hw;
  int gx = (n + 255) / 256;
  const int cap = (FMGAN_NUM_CU * 16 + batch - 1) / batch;
  if (gx > cap) gx = cap;
  if (gx < 1) gx = 1;
  hipStream_t s = (hipStream_t)stream;
  if (vec) hipLaunchKernelGGL(torgb_f32<4>, dim3(gx, batch), dim3(256), lds, s, in, weight, style, bias, skip, out, cin, cout, hw, scale);
  else hipLaunchKernelGGL(torgb_f32<1>, dim3(gx, batch), dim3(256), lds, s, in, weight, style, bias, skip, out, cin, cout, hw, scale);
  return fmgan_check_launch();
}
