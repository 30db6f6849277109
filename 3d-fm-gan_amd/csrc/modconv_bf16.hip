// ModulatedConv2d with bf16 operands and fp32 accumulation on v_mfma_f32_32x32x16_bf16 (gfx950) — the reduced-precision
// contraction of BASELINE config 5's bf16 leg.  NOT the parity path: the fp32 kernel (modconv.hip) stays the default and
// the only one the parity tests of the forward hot path see.
//
// Same operator as fmgan_modconv2d_f32 (stylegan2.py:250-298 restated input-modulated):
//     out[b,o,p] = demod[b,o] * sum_{i,tap} bf16(scale*W[o,i,tap]) * bf16(style[b,i] * in[b,i,p+tap])
// fp32 tensors in HBM on both sides; the modulated activation is rounded to bf16 (RNE, v_cvt_pk_bf16_f32) on its way
// into LDS, the scaled weight once per forward by fmgan_modconv_weight_prep_bf16.  Products of two bf16 values are exact
// in fp32, so the result equals a wide-accumulator convolution of the ROUNDED operands up to fp32 summation error — that
// is what the tests check (tolerance 2e-5 of max|out| against a float64 convolution of the rounded operands), next to the
// bf16-sized distance (~3e-3) from the fp32 kernel.
//
// Mapping.  v_mfma_f32_32x32x16_bf16: lane l (r = l & 31, h = l >> 5) supplies A[row r][k = 8h + j] and
// B[k = 8h + j][col r], j = 0..7, as one 16-byte register quad each (cdna_hip_programming.md §3).  K = input channels
// (16 per chunk), rows = output channels, columns = 32 consecutive x positions of one image row.  Both LDS images are
// therefore laid out [k-half h][row or position][8 bf16]: a wave's operand fetch is ONE ds_read_b128 per 32x32x16 MFMA
// operand with consecutive lanes on consecutive 16-byte slots (conflict-free), against 8 ds_read_b32 for the same K in
// the fp32 kernel.
//   * block = 4 waves, all on the same BM = 32*RM output channels, each on RNP rows of a (4*RNP) x 32 position tile;
//   * patch staging: a thread owns up to NU patch positions, loads their 16 channels (coalesced along x per channel),
//     multiplies by style[b,i], packs to bf16 and writes two 16-byte LDS slots — one chunk ahead in registers;
//   * weight staging: the per-chunk image [tap][h][o][8] is contiguous per (tap, h) in the prepared array and goes
//     HBM -> LDS by `buffer_load_dwordx4 ... lds` (LDS-DMA, no staging registers) into the weight image the current
//     chunk is not reading;
//   * MODE 1 (transposed conv): position (m, n) owns the output quad (2m+py, 2n+px) over the (h+1) x (w+1) quad grid; the
//     positions m = h / n = w own only the last output row / column (their other outputs are masked in the store).  A
//     separate direct kernel for that row and column was the first form: one thread per element with a serial loop over
//     Cin made the 32^2..64^2 layers SLOWER than the fp32 kernel (1083 vs 504 us at 32^2); the extra tile column costs
//     the MFMA launch far less.
// Shapes it serves: cin % 16 == 0, cout % 32 == 0, position grid at least 32 wide; everything else returns
// FMGAN_EUNSUPPORTED and the caller keeps the fp32 kernel (the 4^2..16^2 layers: < 3 % of the FLOPs at B=8).
#include <type_traits>
#include "common.h"

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

namespace {

typedef short bf16x8 __attribute__((ext_vector_type(8)));       // 8 bf16 = one MFMA operand (4 VGPRs)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned pack_bf16(float lo, float hi) {   // RNE; lo -> bits [15:0], hi -> [31:16]
  unsigned r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
  return r;
}
__device__ __forceinline__ float bf16_to_f32(unsigned short v) { return __uint_as_float((unsigned)v << 16); }

// ------------------------------------------------------------------ weight prep
// From the fp32 MFMA layout wt[k][tap][m] = scale * W (fmgan_modconv_weight_prep_f32, any kind: k = the conv's input
// channel, m = its output channel) to  wtb[chunk = k/16][tap][h = (k%16)/8][m (M_pad = multiple of 32)][j = k%8]
// = bf16(wt[k][tap][m]), zero padded.  One rounding of the already scaled weight, whatever the kind.
__global__ __launch_bounds__(256) void modconv_weight_to_bf16(const float* __restrict__ wt,
                                                              unsigned short* __restrict__ wtb, int K, int M, int ktaps) {
  const int Mp = (M + 31) / 32 * 32, chunks = (K + 15) / 16;
  const long long total = (long long)chunks * ktaps * 2 * Mp * 4;     // pairs of bf16
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int jp = (int)(idx & 3);
    long long r = idx >> 2;
    const int m = (int)(r % Mp); r /= Mp;
    const int hh = (int)(r & 1); r >>= 1;
    const int t = (int)(r % ktaps);
    const int c = (int)(r / ktaps);
    float v[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int k = c * 16 + hh * 8 + jp * 2 + e;
      v[e] = (m < M && k < K) ? wt[((long long)k * ktaps + t) * M + m] : 0.f;
    }
    reinterpret_cast<unsigned*>(wtb)[idx] = pack_bf16(v[0], v[1]);
  }
}

// ------------------------------------------------------------------ MFMA conv
struct BFParams {
  const float* in; const unsigned short* wt; const float* style; const float* demod; float* out;
  int batch, cin, cout, h, w, oh, ow;
  long long out_plane_stride; int out_row_stride;
  int gh, gw;                                   // position grid (MODE 0/2: outputs; MODE 1: h x w quads)
  int tiles_x, tiles_y, o_tiles, mp;            // mp = padded row count of the prepared weights
  const float* noise; const float* noise_weight; const float* bias;
  int noise_batch, fuse_act; float alpha, act_scale;
};

// LDS-DMA (`buffer_load_dwordx4 ... lds`): 64 lanes x 16 bytes land at lds + lane * 16 (wave-uniform base) from
// rsrc base + voff (per lane, range-checked) + soff (wave-uniform); no VGPR destination.  (Device-pass guard: the host
// pass of hipcc drops the host stub of a template kernel whose body names this builtin directly — see modconv.hip.)
template <typename RSRC>
__device__ __forceinline__ void dma16_to_lds(RSRC rsrc, void* lds, unsigned voff, unsigned soff) {
#if defined(__HIP_DEVICE_COMPILE__)
  typedef __attribute__((address_space(3))) void* lds_void_ptr;
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void_ptr)lds, 16, voff, soff, 0, 0);
#endif
}

template <int MODE, int RM, int RNP>
__global__ __launch_bounds__(256, 2) void modconv_mfma_bf16(const BFParams p) {
  constexpr int BM = 32 * RM, TH = 4 * RNP, TW = 32;
  constexpr int SP = MODE == 2 ? 2 : 1;                 // input step per position
  constexpr int ORG = MODE == 2 ? 0 : 1;                // patch origin = SP * first position - ORG
  constexpr int PH = MODE == 1 ? TH + 1 : SP * (TH - 1) + 3;
  constexpr int PWP = MODE == 1 ? TW + 1 : SP * (TW - 1) + 3;
  constexpr int PLANE = PH * PWP;
  constexpr int NPH = MODE == 1 ? 4 : 1;
  constexpr int NU = (PLANE + 255) / 256;               // patch positions per thread
  constexpr int WQ = 9 * 2 * BM;                        // 16-byte slots of one weight chunk  [tap][h][BM]
  constexpr int WPIECES = WQ / 64;                      // DMA pieces (64 lanes x 16 B)
  static_assert(WQ % 64 == 0, "weight image = whole DMA pieces");
  constexpr int NWP = (WPIECES + 3) / 4;                // pieces per wave
  extern __shared__ u32x4 smem_q[];
  u32x4* Xs = smem_q + 2 * WQ;                          // [2][PLANE], behind the two weight images

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, khalf = lane >> 5;
  unsigned lb = xcd_remap(blockIdx.x, gridDim.x);
  const int o_tile = lb % p.o_tiles; lb /= p.o_tiles;
  const int tx_i = lb % p.tiles_x; lb /= p.tiles_x;
  const int ty_i = lb % p.tiles_y;
  const int b = lb / p.tiles_y;
  const int o0 = o_tile * BM, x0 = tx_i * TW, y0 = ty_i * TH;
  const int hw = p.h * p.w;
  const float* style_b = p.style + (long long)b * p.cin;

  f32x16 acc[RM][RNP][NPH];
#pragma unroll
  for (int a = 0; a < RM; ++a)
#pragma unroll
    for (int g = 0; g < RNP; ++g)
#pragma unroll
      for (int c = 0; c < NPH; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][g][c][r] = 0.f;

  // ---- staging plan, fixed over the K loop.  Buffer loads: per-lane byte offsets are constant (voffset), the chunk
  // moves a scalar (soffset); patch positions outside the image park their voffset out of range and read zeros.
  constexpr unsigned PARK = 0xFFFFFFF0u;
  const unsigned x_bytes = (unsigned)((long long)p.cin * hw * 4);
  const unsigned w_bytes = (unsigned)((long long)(p.cin >> 4) * 18 * p.mp * 16);
  const auto rsrc_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in + (long long)b * p.cin * hw), 0, x_bytes, 0x00020000);
  const auto rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(p.wt), 0, w_bytes, 0x00020000);
  unsigned xvo[NU], wvo[NWP];
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    const int q = tid + 256 * u;
    const int r = q / PWP, c = q - r * PWP;
    const int y = SP * y0 + r - ORG, x = SP * x0 + c - ORG;
    xvo[u] = (q < PLANE && y >= 0 && y < p.h && x >= 0 && x < p.w) ? (unsigned)(y * p.w + x) * 4u : PARK;
  }
#pragma unroll
  for (int k = 0; k < NWP; ++k) {
    const int idx = (4 * k + wave) * 64 + lane;          // slot (tap*2 + h) * BM + o of the chunk image
    const int th = idx / BM, o = idx - th * BM;
    wvo[k] = (unsigned)((th * p.mp + o0 + o) * 16);
  }
  float xv[NU][16];
  auto issue = [&](int i0, int wbuf) {
    const unsigned soff_w = (unsigned)((i0 >> 4) * 18 * p.mp * 16);
#pragma unroll
    for (int k = 0; k < NWP; ++k)
      if (WPIECES % 4 == 0 || 4 * k + wave < WPIECES)
        dma16_to_lds(rsrc_w, smem_q + wbuf * WQ + (4 * k + wave) * 64, wvo[k], soff_w);
#pragma unroll
    for (int kc = 0; kc < 16; ++kc) {
      const unsigned soff = (unsigned)((i0 + kc) * hw * 4);
#pragma unroll
      for (int u = 0; u < NU; ++u)
        xv[u][kc] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc_x, xvo[u], soff, 0));
    }
  };
  auto commit = [&](int i0) {
    float sv[16];
#pragma unroll
    for (int kc = 0; kc < 16; ++kc) sv[kc] = style_b[i0 + kc];     // wave-uniform: scalar loads
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const int q = tid + 256 * u;
      if (q < PLANE) {
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
          u32x4 t;
          t.x = pack_bf16(xv[u][8 * hh + 0] * sv[8 * hh + 0], xv[u][8 * hh + 1] * sv[8 * hh + 1]);
          t.y = pack_bf16(xv[u][8 * hh + 2] * sv[8 * hh + 2], xv[u][8 * hh + 3] * sv[8 * hh + 3]);
          t.z = pack_bf16(xv[u][8 * hh + 4] * sv[8 * hh + 4], xv[u][8 * hh + 5] * sv[8 * hh + 5]);
          t.w = pack_bf16(xv[u][8 * hh + 6] * sv[8 * hh + 6], xv[u][8 * hh + 7] * sv[8 * hh + 7]);
          Xs[hh * PLANE + q] = t;
        }
      }
    }
  };

  // this lane's patch base per position row: row ty = wave*RNP + g, column l31
  int pbase[RNP];
#pragma unroll
  for (int g = 0; g < RNP; ++g) pbase[g] = khalf * PLANE + SP * (wave * RNP + g) * PWP + SP * l31;

  struct Ops { bf16x8 a[RM]; bf16x8 b[RNP]; };
  issue(0, 0);
  int wbuf = 0;
  for (int i0 = 0; i0 < p.cin; i0 += 16, wbuf ^= 1) {
    __builtin_amdgcn_s_waitcnt(0);         // this wave's loads and DMA pieces of chunk i0 have landed
    __syncthreads();                       // ... everyone's; and every wave is done reading the previous chunk
    commit(i0);
    __syncthreads();
    __builtin_amdgcn_sched_barrier(0);
    if (i0 + 16 < p.cin) issue(i0 + 16, wbuf ^ 1);   // in flight during this chunk's MFMAs
    const u32x4* wl = smem_q + wbuf * WQ + khalf * BM + l31;
    if constexpr (MODE != 1) {
      // operands of tap t+1 are fetched while tap t is on the matrix pipe (the sched_barriers keep that order: hipcc
      // otherwise hoists all 9 taps' ds_reads to the top of the chunk — 180 live registers on the 32-channel tile)
      auto fetch = [&](Ops& o, int t) {
        const int ky = t / 3, kx = t - 3 * ky;
#pragma unroll
        for (int m = 0; m < RM; ++m) o.a[m] = __builtin_bit_cast(bf16x8, wl[t * 2 * BM + m * 32]);
#pragma unroll
        for (int g = 0; g < RNP; ++g) o.b[g] = __builtin_bit_cast(bf16x8, Xs[pbase[g] + ky * PWP + kx]);
      };
      auto mma = [&](const Ops& o) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int m = 0; m < RM; ++m)
#pragma unroll
          for (int g = 0; g < RNP; ++g)
            acc[m][g][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(o.a[m], o.b[g], acc[m][g][0], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
      };
      Ops ops[2];
      fetch(ops[0], 0);
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        __builtin_amdgcn_sched_barrier(0);
        if (t + 1 < 9) fetch(ops[(t + 1) & 1], t + 1);
        __builtin_amdgcn_sched_barrier(0);
        mma(ops[t & 1]);
      }
      __builtin_amdgcn_sched_barrier(0);
    } else {
      // (tap = ky*3+kx, input-offset index j -> (ro, co) = (1,1) (1,0) (0,1) (0,0), phase = py*2+px); modconv.hip
      constexpr int T[9][3] = {{0, 0, 0}, {2, 1, 0}, {6, 2, 0}, {8, 3, 0}, {1, 0, 1}, {7, 2, 1}, {3, 0, 2}, {5, 1, 2}, {4, 0, 3}};
      bf16x8 bq[4][RNP];
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int g = 0; g < RNP; ++g)
          bq[j][g] = __builtin_bit_cast(bf16x8, Xs[pbase[g] + (1 - (j >> 1)) * PWP + (1 - (j & 1))]);
      bf16x8 a[2][RM];
#pragma unroll
      for (int m = 0; m < RM; ++m) a[0][m] = __builtin_bit_cast(bf16x8, wl[T[0][0] * 2 * BM + m * 32]);
#pragma unroll
      for (int q = 0; q < 9; ++q) {
        __builtin_amdgcn_sched_barrier(0);
        if (q + 1 < 9) {
#pragma unroll
          for (int m = 0; m < RM; ++m) a[(q + 1) & 1][m] = __builtin_bit_cast(bf16x8, wl[T[q + 1][0] * 2 * BM + m * 32]);
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int m = 0; m < RM; ++m)
#pragma unroll
          for (int g = 0; g < RNP; ++g)
            acc[m][g][T[q][2]] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[q & 1][m], bq[T[q][1]][g], acc[m][g][T[q][2]], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  // ---- epilogue (the fp32 kernel's: loads first, arithmetic in place; stores through a buffer resource — per-lane byte
  // offset fixed per position group and parked for positions outside the grid, channel / phase row as a scalar offset —
  // whenever the tile's channels all exist, the sample's output fits 32-bit offsets and, MODE 1, every quad of the tile is
  // complete; the generic predicated stores otherwise)
  const bool actf = MODE == 0 && p.fuse_act;
  const float nw = (actf && p.noise && p.noise_weight) ? p.noise_weight[0] : 0.f;
  const int orow = o0 + 4 * khalf;          // row r of a 32-row group adds (r&3) + 8*(r>>2)
  float* dst_b = p.out + (long long)b * p.cout * p.out_plane_stride;
  const long long dps = p.out_plane_stride;
  const int drs = p.out_row_stride;
  const long long slab = (long long)p.cout * dps * 4;
  constexpr int TH_ = 4 * RNP;
  bool bstore = o0 + 32 * RM <= p.cout && slab < 0xFFFFFFF0LL;
  if constexpr (MODE == 1) bstore = bstore && y0 + TH_ <= p.h && x0 + 32 <= p.w;
  const auto rsrc_o = __builtin_amdgcn_make_buffer_rsrc(dst_b, 0, bstore ? (unsigned)slab : 0u, 0x00020000);
  unsigned alpha_bits = __float_as_uint(p.alpha), ascale_bits = __float_as_uint(p.act_scale);
  asm volatile("" : "+s"(alpha_bits), "+s"(ascale_bits));
  const float alpha = __uint_as_float(alpha_bits), ascale = __uint_as_float(ascale_bits);
  const unsigned dps4 = (unsigned)(dps * 4), drs4 = (unsigned)drs * 4u;
  auto rows = [&](auto act_c, auto bst_c) {
    constexpr bool ACT = decltype(act_c)::value, BST = decltype(bst_c)::value;
#pragma unroll
    for (int m = 0; m < RM; ++m) {
      float bias_m[16], dm_m[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int oc = min(orow + m * 32 + (r & 3) + 8 * (r >> 2), p.cout - 1);
        bias_m[r] = (ACT && p.bias) ? p.bias[oc] : 0.f;
        dm_m[r] = p.demod ? p.demod[(long long)b * p.cout + oc] : 1.f;
      }
      const unsigned so_m = (unsigned)(o0 + m * 32) * dps4;
#pragma unroll
      for (int g = 0; g < RNP; ++g) {
        const int py_ = y0 + wave * RNP + g, px_ = x0 + l31;       // position
        const bool vg = py_ < p.gh && px_ < p.gw;
        constexpr int SC = MODE == 1 ? 2 : 1;
        const unsigned vof = (BST && vg) ? (unsigned)(((long long)(4 * khalf) * dps + (long long)(SC * py_) * drs + SC * px_) * 4) : 0xFFFFFFF0u;
        if constexpr (MODE != 1) {
          const int pix = vg ? py_ * p.ow + px_ : 0;
          const float nz = (ACT && p.noise) ? __fmul_rn(nw, p.noise[(long long)(p.noise_batch == 1 ? 0 : b) * p.oh * p.ow + pix]) : 0.f;
          float* dpos = dst_b + (long long)(vg ? py_ : 0) * p.out_row_stride + (vg ? px_ : 0);
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int o = orow + m * 32 + (r & 3) + 8 * (r >> 2);
            float v = acc[m][g][0][r] * dm_m[r];
            if constexpr (ACT) {
              v = __fadd_rn(__fadd_rn(v, nz), bias_m[r]);
              v = (v > 0.f ? v : v * alpha) * ascale;
            }
            if constexpr (BST) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rsrc_o, vof, so_m + (unsigned)((r & 3) + 8 * (r >> 2)) * dps4, 0);
            else if (vg && o < p.cout) dpos[(long long)o * p.out_plane_stride] = v;
          }
        } else {
          float* dpos = dst_b + (long long)(vg ? 2 * py_ : 0) * p.out_row_stride + (vg ? 2 * px_ : 0);
          const bool pair = 2 * px_ + 1 < p.ow;          // position n = w owns the last output column only
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int o = orow + m * 32 + (r & 3) + 8 * (r >> 2);
            if constexpr (BST) {
#pragma unroll
              for (int py = 0; py < 2; ++py) {
                u32x2 t;
                t.x = __float_as_uint(acc[m][g][py * 2][r] * dm_m[r]);
                t.y = __float_as_uint(acc[m][g][py * 2 + 1][r] * dm_m[r]);
                __builtin_amdgcn_raw_buffer_store_b64(t, rsrc_o, vof, so_m + (unsigned)((r & 3) + 8 * (r >> 2)) * dps4 + (unsigned)py * drs4, 0);
              }
            } else {
              if (!(vg && o < p.cout)) continue;
              float* dst = dpos + (long long)o * p.out_plane_stride;
#pragma unroll
              for (int py = 0; py < 2; ++py) {
                if (2 * py_ + py >= p.oh) continue;          // position m = h owns the last output row only
                const float v0 = acc[m][g][py * 2][r] * dm_m[r], v1 = acc[m][g][py * 2 + 1][r] * dm_m[r];
                if (pair) {
                  f32x2_u t;
                  t.x = v0; t.y = v1;
                  *reinterpret_cast<f32x2_u*>(dst + (long long)py * p.out_row_stride) = t;
                } else {
                  dst[(long long)py * p.out_row_stride] = v0;
                }
              }
            }
          }
        }
      }
    }
  };
  if (bstore) {
    if (actf) rows(std::true_type{}, std::true_type{}); else rows(std::false_type{}, std::true_type{});
  } else {
    if (actf) rows(std::true_type{}, std::false_type{}); else rows(std::false_type{}, std::false_type{});
  }
}

template <int MODE, int RM, int RNP>
int launch_bf16(BFParams& p, hipStream_t s) {
  constexpr int BM = 32 * RM, TH = 4 * RNP;
  constexpr int SP = MODE == 2 ? 2 : 1;
  constexpr int PH = MODE == 1 ? TH + 1 : SP * (TH - 1) + 3;
  constexpr int PWP = MODE == 1 ? 33 : SP * 31 + 3;
  constexpr size_t lds = (size_t)(2 * 9 * 2 * BM + 2 * PH * PWP) * 16;     // two weight images + the patch
  p.tiles_x = (p.gw + 31) / 32;
  p.tiles_y = (p.gh + TH - 1) / TH;
  p.o_tiles = (p.cout + BM - 1) / BM;
  const long long blocks = (long long)p.o_tiles * p.tiles_x * p.tiles_y * p.batch;
  if (blocks > 0x7fffffffLL) return FMGAN_EOVERFLOW;
  static bool attr = false;     // > 48 KB of dynamic LDS needs the opt-in once per instantiation (idempotent)
  if (lds > 48 * 1024 && !attr) {
    (void)hipFuncSetAttribute((const void*)modconv_mfma_bf16<MODE, RM, RNP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr = true;
  }
  hipLaunchKernelGGL((modconv_mfma_bf16<MODE, RM, RNP>), dim3((unsigned)blocks), dim3(256), lds, s, p);
  return fmgan_check_launch();
}

// ------------------------------------------------------------------ fp32 on the bf16 matrix pipe: split operands
// v_mfma_f32_32x32x16_bf16 does 16x the MACs per cycle of v_mfma_f32_32x32x2_f32.  An fp32 value splits exactly into three
// bf16 pieces, x = hi + mid + lo (8 + 8 + 8 mantissa bits: hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid), every
// difference exact in fp32), every bf16 x bf16 product is exact in fp32, and
//     a * b = ah*bh + ah*bm + am*bh + am*bm + ah*bl + al*bh   + O(2^-24 |a b|)
// — the three dropped cross terms (am*bl, al*bm, al*bl) are at the size of ONE fp32 rounding of the product.  Six bf16
// MFMAs with fp32 accumulation therefore reproduce the fp32 contraction to fp32 accuracy at 16/6 = 2.7x its matrix-pipe
// rate.  Forward modes 0 and 1 only (inference); a LABELLED path — the parity path and every headline number stay on
// v_mfma_f32_32x32x2_f32 (modconv.hip).  Held by tests/test_hip_modconv_bf16.py to the fp32 kernel's own tolerances
// (per layer and end to end against the float64 fixtures).
//   * tile: 32 output channels x (4*RNP rows x 32 columns) positions per block, 4 waves on the same channels;
//     LDS per 16-channel chunk: 3 weight pieces [piece][tap][k-half][32][8] = 27 KB + 3 patch pieces — 60 KB for
//     RNP = 2: two blocks per CU;
//   * weights by LDS-DMA after the chunk's first barrier (single image: a second one would cost the second block),
//     patch through registers one chunk ahead, split into its pieces on the way into LDS (11 VALU ops per pair);
//   * per tap: 3 + 3*RNP operand reads (ds_read_b128) feed 6*RNP MFMAs.
__global__ __launch_bounds__(256) void modconv_weight_to_bf16x3(const float* __restrict__ wt, unsigned* __restrict__ wtb,
                                                                int K, int M, int ktaps) {
  const int Mp = (M + 31) / 32 * 32, chunks = (K + 15) / 16;
  const long long per_piece = (long long)ktaps * 2 * Mp * 4;          // u32 (bf16 pairs) of one piece of one chunk
  const long long total = (long long)chunks * per_piece;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(idx / per_piece);
    long long r = idx - (long long)c * per_piece;
    const int jp = (int)(r & 3); r >>= 2;
    const int m = (int)(r % Mp); r /= Mp;
    const int hh = (int)(r & 1);
    const int t = (int)(r >> 1);
    float v[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int k = c * 16 + hh * 8 + jp * 2 + e;
      v[e] = (m < M && k < K) ? wt[((long long)k * ktaps + t) * M + m] : 0.f;
    }
    const unsigned h01 = pack_bf16(v[0], v[1]);
    const float r0 = v[0] - __uint_as_float(h01 << 16), r1 = v[1] - __uint_as_float(h01 & 0xffff0000u);
    const unsigned m01 = pack_bf16(r0, r1);
    const float s0 = r0 - __uint_as_float(m01 << 16), s1 = r1 - __uint_as_float(m01 & 0xffff0000u);
    const unsigned l01 = pack_bf16(s0, s1);
    const long long base = (long long)c * 3 * per_piece + (idx - (long long)c * per_piece);
    wtb[base] = h01; wtb[base + per_piece] = m01; wtb[base + 2 * per_piece] = l01;
  }
}

template <int MODE, int RNP>
__global__ __launch_bounds__(256, 2) void modconv_mfma_bf16x3(const BFParams p) {
  static_assert(MODE == 0 || MODE == 1, "forward modes only");
  constexpr int BM = 32, TH = 4 * RNP, TW = 32;
  constexpr int PH = MODE == 1 ? TH + 1 : TH + 2;
  constexpr int PWP = MODE == 1 ? TW + 1 : TW + 2;
  constexpr int PLANE = PH * PWP;
  constexpr int NPH = MODE == 1 ? 4 : 1;
  constexpr int NU = (PLANE + 255) / 256;
  constexpr int WQ1 = 9 * 2 * BM;                       // 16-byte slots of one weight piece
  constexpr int WQ = 3 * WQ1;                           // all three pieces of a chunk: 27 DMA pieces of 64 slots
  static_assert(WQ % 64 == 0, "weight image = whole DMA pieces");
  constexpr int WPIECES = WQ / 64, NWP = (WPIECES + 3) / 4;
  extern __shared__ u32x4 smem_q[];
  u32x4* Ws = smem_q;                                   // [3][9][2][32]
  u32x4* Xs = smem_q + WQ;                              // [3][2][PLANE]

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, khalf = lane >> 5;
  unsigned lb = xcd_remap(blockIdx.x, gridDim.x);
  const int o_tile = lb % p.o_tiles; lb /= p.o_tiles;
  const int tx_i = lb % p.tiles_x; lb /= p.tiles_x;
  const int ty_i = lb % p.tiles_y;
  const int b = lb / p.tiles_y;
  const int o0 = o_tile * BM, x0 = tx_i * TW, y0 = ty_i * TH;
  const int hw = p.h * p.w;
  const float* style_b = p.style + (long long)b * p.cin;

  f32x16 acc[RNP][NPH];
#pragma unroll
  for (int g = 0; g < RNP; ++g)
#pragma unroll
    for (int c = 0; c < NPH; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[g][c][r] = 0.f;

  constexpr unsigned PARK = 0xFFFFFFF0u;
  const unsigned x_bytes = (unsigned)((long long)p.cin * hw * 4);
  const unsigned w_bytes = (unsigned)((long long)(p.cin >> 4) * 54 * p.mp * 16);
  const auto rsrc_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in + (long long)b * p.cin * hw), 0, x_bytes, 0x00020000);
  const auto rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(p.wt), 0, w_bytes, 0x00020000);
  unsigned xvo[NU], wvo[NWP];
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    const int q = tid + 256 * u;
    const int r = q / PWP, c = q - r * PWP;
    const int y = y0 + r - 1, x = x0 + c - 1;
    xvo[u] = (q < PLANE && y >= 0 && y < p.h && x >= 0 && x < p.w) ? (unsigned)(y * p.w + x) * 4u : PARK;
  }
#pragma unroll
  for (int k = 0; k < NWP; ++k) {
    const int idx = (4 * k + wave) * 64 + lane;          // slot (piece*18 + tap*2 + h) * 32 + o
    const int pth = idx / BM, o = idx - pth * BM;
    wvo[k] = (unsigned)((pth * p.mp + o0 + o) * 16);
  }
  float xv[NU][16];
  auto issue_x = [&](int i0) {
#pragma unroll
    for (int kc = 0; kc < 16; ++kc) {
      const unsigned soff = (unsigned)((i0 + kc) * hw * 4);
#pragma unroll
      for (int u = 0; u < NU; ++u)
        xv[u][kc] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc_x, xvo[u], soff, 0));
    }
  };
  auto dma_w = [&](int i0) {
    const unsigned soff_w = (unsigned)((i0 >> 4) * 54 * p.mp * 16);
#pragma unroll
    for (int k = 0; k < NWP; ++k)
      if (4 * k + wave < WPIECES) dma16_to_lds(rsrc_w, Ws + (4 * k + wave) * 64, wvo[k], soff_w);
  };
  auto commit_x = [&](int i0) {
    float sv[16];
#pragma unroll
    for (int kc = 0; kc < 16; ++kc) sv[kc] = style_b[i0 + kc];
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const int q = tid + 256 * u;
      if (q < PLANE) {
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
          u32x4 ph, pm, pl;
          unsigned* qh = reinterpret_cast<unsigned*>(&ph);
          unsigned* qm = reinterpret_cast<unsigned*>(&pm);
          unsigned* ql = reinterpret_cast<unsigned*>(&pl);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float v0 = xv[u][8 * hh + 2 * e] * sv[8 * hh + 2 * e], v1 = xv[u][8 * hh + 2 * e + 1] * sv[8 * hh + 2 * e + 1];
            const unsigned h01 = pack_bf16(v0, v1);
            const float r0 = v0 - __uint_as_float(h01 << 16), r1 = v1 - __uint_as_float(h01 & 0xffff0000u);
            const unsigned m01 = pack_bf16(r0, r1);
            const float s0 = r0 - __uint_as_float(m01 << 16), s1 = r1 - __uint_as_float(m01 & 0xffff0000u);
            qh[e] = h01; qm[e] = m01; ql[e] = pack_bf16(s0, s1);
          }
          Xs[(0 * 2 + hh) * PLANE + q] = ph;
          Xs[(1 * 2 + hh) * PLANE + q] = pm;
          Xs[(2 * 2 + hh) * PLANE + q] = pl;
        }
      }
    }
  };

  int pbase[RNP];
#pragma unroll
  for (int g = 0; g < RNP; ++g) pbase[g] = khalf * PLANE + (wave * RNP + g) * PWP + l31;
  const u32x4* wl = Ws + khalf * BM + l31;             // + (piece*18 + tap*2) * BM

  // the six products, smallest first: (a piece, b piece)
  constexpr int PA[6] = {0, 2, 1, 0, 1, 0}, PB[6] = {2, 0, 1, 1, 0, 0};
  issue_x(0);
  for (int i0 = 0; i0 < p.cin; i0 += 16) {
    __builtin_amdgcn_s_waitcnt(0);         // this wave's patch loads of chunk i0 have landed
    __syncthreads();                       // every wave is done reading the previous chunk's LDS images
    dma_w(i0);
    commit_x(i0);
    __builtin_amdgcn_s_waitcnt(0);         // this wave's weight pieces have landed
    __syncthreads();
    __builtin_amdgcn_sched_barrier(0);
    if (i0 + 16 < p.cin) issue_x(i0 + 16); // in flight during this chunk's MFMAs
    if constexpr (MODE == 0) {
      struct Ops { bf16x8 a[3]; bf16x8 b[3][RNP]; };
      auto fetch = [&](Ops& o, int t) {
        const int ky = t / 3, kx = t - 3 * ky;
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) {
          o.a[pc] = __builtin_bit_cast(bf16x8, wl[(pc * 18 + t * 2) * BM]);
#pragma unroll
          for (int g = 0; g < RNP; ++g) o.b[pc][g] = __builtin_bit_cast(bf16x8, Xs[pc * 2 * PLANE + pbase[g] + ky * PWP + kx]);
        }
      };
      Ops ops[2];
      fetch(ops[0], 0);
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        __builtin_amdgcn_sched_barrier(0);
        if (t + 1 < 9) fetch(ops[(t + 1) & 1], t + 1);
        __builtin_amdgcn_sched_barrier(0);
        const Ops& o = ops[t & 1];
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int q = 0; q < 6; ++q)
#pragma unroll
          for (int g = 0; g < RNP; ++g)
            acc[g][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(o.a[PA[q]], o.b[PB[q]][g], acc[g][0], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
      }
      __builtin_amdgcn_sched_barrier(0);
    } else {
      constexpr int T[9][3] = {{0, 0, 0}, {2, 1, 0}, {6, 2, 0}, {8, 3, 0}, {1, 0, 1}, {7, 2, 1}, {3, 0, 2}, {5, 1, 2}, {4, 0, 3}};
      bf16x8 bq[3][4][RNP];
#pragma unroll
      for (int pc = 0; pc < 3; ++pc)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int g = 0; g < RNP; ++g)
            bq[pc][j][g] = __builtin_bit_cast(bf16x8, Xs[pc * 2 * PLANE + pbase[g] + (1 - (j >> 1)) * PWP + (1 - (j & 1))]);
      bf16x8 a[2][3];
#pragma unroll
      for (int pc = 0; pc < 3; ++pc) a[0][pc] = __builtin_bit_cast(bf16x8, wl[(pc * 18 + T[0][0] * 2) * BM]);
#pragma unroll
      for (int q9 = 0; q9 < 9; ++q9) {
        __builtin_amdgcn_sched_barrier(0);
        if (q9 + 1 < 9) {
#pragma unroll
          for (int pc = 0; pc < 3; ++pc) a[(q9 + 1) & 1][pc] = __builtin_bit_cast(bf16x8, wl[(pc * 18 + T[q9 + 1][0] * 2) * BM]);
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int q = 0; q < 6; ++q)
#pragma unroll
          for (int g = 0; g < RNP; ++g)
            acc[g][T[q9][2]] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[q9 & 1][PA[q]], bq[PB[q]][T[q9][1]][g], acc[g][T[q9][2]], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  // ---- epilogue (as modconv_mfma_bf16 with RM = 1)
  const bool actf = MODE == 0 && p.fuse_act;
  const float nw = (actf && p.noise && p.noise_weight) ? p.noise_weight[0] : 0.f;
  const int orow = o0 + 4 * khalf;          // row r of a 32-row group adds (r&3) + 8*(r>>2)
  float* dst_b = p.out + (long long)b * p.cout * p.out_plane_stride;
  const long long dps = p.out_plane_stride;
  const int drs = p.out_row_stride;
  const long long slab = (long long)p.cout * dps * 4;
  constexpr int TH_ = 4 * RNP;
  bool bstore = o0 + 32 <= p.cout && slab < 0xFFFFFFF0LL;
  if constexpr (MODE == 1) bstore = bstore && y0 + TH_ <= p.h && x0 + 32 <= p.w;
  const auto rsrc_o = __builtin_amdgcn_make_buffer_rsrc(dst_b, 0, bstore ? (unsigned)slab : 0u, 0x00020000);
  unsigned alpha_bits = __float_as_uint(p.alpha), ascale_bits = __float_as_uint(p.act_scale);
  asm volatile("" : "+s"(alpha_bits), "+s"(ascale_bits));
  const float alpha = __uint_as_float(alpha_bits), ascale = __uint_as_float(ascale_bits);
  const unsigned dps4 = (unsigned)(dps * 4), drs4 = (unsigned)drs * 4u;
  auto rows = [&](auto act_c, auto bst_c) {
    constexpr bool ACT = decltype(act_c)::value, BST = decltype(bst_c)::value;
#pragma unroll
    for (int m = 0; m < 1; ++m) {
      float bias_m[16], dm_m[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int oc = min(orow + m * 32 + (r & 3) + 8 * (r >> 2), p.cout - 1);
        bias_m[r] = (ACT && p.bias) ? p.bias[oc] : 0.f;
        dm_m[r] = p.demod ? p.demod[(long long)b * p.cout + oc] : 1.f;
      }
      const unsigned so_m = (unsigned)(o0 + m * 32) * dps4;
#pragma unroll
      for (int g = 0; g < RNP; ++g) {
        const int py_ = y0 + wave * RNP + g, px_ = x0 + l31;       // position
        const bool vg = py_ < p.gh && px_ < p.gw;
        constexpr int SC = MODE == 1 ? 2 : 1;
        const unsigned vof = (BST && vg) ? (unsigned)(((long long)(4 * khalf) * dps + (long long)(SC * py_) * drs + SC * px_) * 4) : 0xFFFFFFF0u;
        if constexpr (MODE != 1) {
          const int pix = vg ? py_ * p.ow + px_ : 0;
          const float nz = (ACT && p.noise) ? __fmul_rn(nw, p.noise[(long long)(p.noise_batch == 1 ? 0 : b) * p.oh * p.ow + pix]) : 0.f;
          float* dpos = dst_b + (long long)(vg ? py_ : 0) * p.out_row_stride + (vg ? px_ : 0);
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int o = orow + m * 32 + (r & 3) + 8 * (r >> 2);
            float v = acc[g][0][r] * dm_m[r];
            if constexpr (ACT) {
              v = __fadd_rn(__fadd_rn(v, nz), bias_m[r]);
              v = (v > 0.f ? v : v * alpha) * ascale;
            }
            if constexpr (BST) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rsrc_o, vof, so_m + (unsigned)((r & 3) + 8 * (r >> 2)) * dps4, 0);
            else if (vg && o < p.cout) dpos[(long long)o * p.out_plane_stride] = v;
          }
        } else {
          float* dpos = dst_b + (long long)(vg ? 2 * py_ : 0) * p.out_row_stride + (vg ? 2 * px_ : 0);
          const bool pair = 2 * px_ + 1 < p.ow;          // position n = w owns the last output column only
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int o = orow + m * 32 + (r & 3) + 8 * (r >> 2);
            if constexpr (BST) {
#pragma unroll
              for (int py = 0; py < 2; ++py) {
                u32x2 t;
                t.x = __float_as_uint(acc[g][py * 2][r] * dm_m[r]);
                t.y = __float_as_uint(acc[g][py * 2 + 1][r] * dm_m[r]);
                __builtin_amdgcn_raw_buffer_store_b64(t, rsrc_o, vof, so_m + (unsigned)((r & 3) + 8 * (r >> 2)) * dps4 + (unsigned)py * drs4, 0);
              }
            } else {
              if (!(vg && o < p.cout)) continue;
              float* dst = dpos + (long long)o * p.out_plane_stride;
#pragma unroll
              for (int py = 0; py < 2; ++py) {
                if (2 * py_ + py >= p.oh) continue;          // position m = h owns the last output row only
                const float v0 = acc[g][py * 2][r] * dm_m[r], v1 = acc[g][py * 2 + 1][r] * dm_m[r];
                if (pair) {
                  f32x2_u t;
                  t.x = v0; t.y = v1;
                  *reinterpret_cast<f32x2_u*>(dst + (long long)py * p.out_row_stride) = t;
                } else {
                  dst[(long long)py * p.out_row_stride] = v0;
                }
              }
            }
          }
        }
      }
    }
  };
  if (bstore) {
    if (actf) rows(std::true_type{}, std::true_type{}); else rows(std::false_type{}, std::true_type{});
  } else {
    if (actf) rows(std::true_type{}, std::false_type{}); else rows(std::false_type{}, std::false_type{});
  }
}

template <int MODE, int RNP>
int launch_bf16x3(BFParams& p, hipStream_t s) {
  constexpr int TH = 4 * RNP;
  constexpr int PH = MODE == 1 ? TH + 1 : TH + 2, PWP = MODE == 1 ? 33 : 34;
  constexpr size_t lds = (size_t)(3 * 9 * 2 * 32 + 3 * 2 * PH * PWP) * 16;
  p.tiles_x = (p.gw + 31) / 32;
  p.tiles_y = (p.gh + TH - 1) / TH;
  p.o_tiles = (p.cout + 31) / 32;
  const long long blocks = (long long)p.o_tiles * p.tiles_x * p.tiles_y * p.batch;
  if (blocks > 0x7fffffffLL) return FMGAN_EOVERFLOW;
  static bool attr = false;
  if (lds > 48 * 1024 && !attr) {
    (void)hipFuncSetAttribute((const void*)modconv_mfma_bf16x3<MODE, RNP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr = true;
  }
  hipLaunchKernelGGL((modconv_mfma_bf16x3<MODE, RNP>), dim3((unsigned)blocks), dim3(256), lds, s, p);
  return fmgan_check_launch();
}

}  // namespace

extern "C" long long fmgan_modconv_weight_bf16_bytes(int cin, int cout, int ktaps) {
  if (cout <= 0 || cin <= 0 || ktaps <= 0) return -1;
  return (long long)((cin + 15) / 16) * ktaps * 2 * ((cout + 31) / 32 * 32) * 16;
}

extern "C" int fmgan_modconv_weight_to_bf16(const float* wt, void* wt_bf16, int cin, int cout, int ktaps, void* stream) {
  if (cout <= 0 || cin <= 0 || ktaps <= 0) return FMGAN_EINVAL;
  if (!wt || !wt_bf16) return FMGAN_EINVAL;
  const long long pairs = fmgan_modconv_weight_bf16_bytes(cin, cout, ktaps) / 4;
  long long blocks = (pairs + 255) / 256;
  const long long cap = (long long)FMGAN_NUM_CU * 16;
  if (blocks > cap) blocks = cap;
  hipLaunchKernelGGL(modconv_weight_to_bf16, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, wt,
                     (unsigned short*)wt_bf16, cin, cout, ktaps);
  return fmgan_check_launch();
}

// 1 when fmgan_modconv2d_bf16 serves the shape (host logic only)
extern "C" int fmgan_modconv2d_bf16_supported(int batch, int cin, int cout, int h, int w, int mode) {
  if (batch <= 0 || cin <= 0 || cout <= 0 || h <= 0 || w <= 0 || mode < 0 || mode > 2) return 0;
  if ((cin & 15) || (cout & 31)) return 0;
  int gh = h, gw = w;
  if (mode == 2) { if (h < 3 || w < 3) return 0; gh = (h - 3) / 2 + 1; gw = (w - 3) / 2 + 1; }
  if (gw < 32 || gh < 4) return 0;
  // buffer resources: one sample's input and the whole prepared weight array must be shorter than the parked offset
  if ((long long)cin * h * w * 4 >= 0xFFFFFFF0LL || (long long)(2 * h + 1) * (2 * w + 1) > 0x7fffffffLL) return 0;
  if ((long long)(cin >> 4) * 18 * ((cout + 31) / 32 * 32) * 16 >= 0xFFFFFFF0LL) return 0;
  return 1;
}

extern "C" int fmgan_modconv2d_bf16(const float* in, const void* wt_bf16, const float* style, const float* demod,
                                    float* out, int batch, int cin, int cout, int h, int w, int mode, const float* noise,
                                    const float* noise_weight, const float* bias, int noise_batch, int fuse_act,
                                    float alpha, float act_scale, long long out_plane_stride, int out_row_stride,
                                    void* stream) {
  if (batch < 0 || cin <= 0 || cout <= 0 || h <= 0 || w <= 0) return FMGAN_EINVAL;
  if (mode < 0 || mode > 2) return FMGAN_EUNSUPPORTED;
  if (mode != 0 && fuse_act) return FMGAN_EUNSUPPORTED;
  if (batch == 0) return FMGAN_OK;
  if (!fmgan_modconv2d_bf16_supported(batch, cin, cout, h, w, mode)) return FMGAN_EUNSUPPORTED;
  if (!in || !wt_bf16 || !style || !out) return FMGAN_EINVAL;
  if (fuse_act && noise && noise_batch != 1 && noise_batch != batch) return FMGAN_EINVAL;
  BFParams p{};
  p.in = in; p.wt = (const unsigned short*)wt_bf16; p.style = style; p.demod = demod; p.out = out;
  p.batch = batch; p.cin = cin; p.cout = cout; p.h = h; p.w = w;
  if (mode == 1) { p.oh = 2 * h + 1; p.ow = 2 * w + 1; p.gh = h + 1; p.gw = w + 1; }   // quads (m <= h, n <= w)
  else if (mode == 2) { p.oh = (h - 3) / 2 + 1; p.ow = (w - 3) / 2 + 1; p.gh = p.oh; p.gw = p.ow; }
  else { p.oh = h; p.ow = w; p.gh = h; p.gw = w; }
  if (out_row_stride == 0) out_row_stride = p.ow;
  if (out_plane_stride == 0) out_plane_stride = (long long)p.oh * out_row_stride;
  if (out_row_stride < p.ow || out_plane_stride < (long long)p.oh * out_row_stride) return FMGAN_EINVAL;
  if ((long long)batch * cout * out_plane_stride > (1LL << 40)) return FMGAN_EOVERFLOW;
  p.out_plane_stride = out_plane_stride; p.out_row_stride = out_row_stride;
  p.mp = (cout + 31) / 32 * 32;
  p.noise = noise; p.noise_weight = noise_weight; p.bias = bias; p.noise_batch = noise_batch; p.fuse_act = fuse_act;
  p.alpha = alpha; p.act_scale = act_scale;
  hipStream_t s = (hipStream_t)stream;
  int st;
  if (mode == 0) {
    // (a 128-channel tile needs 2 x 37 KB of weight images: one block per CU; 64 channels x 256 positions fits three)
    if (cout >= 64) st = launch_bf16<0, 2, 2>(p, s);
    else st = launch_bf16<0, 1, 4>(p, s);
  } else if (mode == 2) {
    if (cout >= 64) st = launch_bf16<2, 2, 2>(p, s);
    else st = launch_bf16<2, 1, 2>(p, s);
  } else {
    if (cout >= 64) st = launch_bf16<1, 2, 1>(p, s);
    else st = launch_bf16<1, 1, 2>(p, s);
  }
  return st;
}

// ---- split-operand fp32 ("bf16x3"): see the kernel's comment.  Forward modes 0 / 1.
extern "C" long long fmgan_modconv_weight_bf16x3_bytes(int cin, int cout, int ktaps) {
  const long long one = fmgan_modconv_weight_bf16_bytes(cin, cout, ktaps);
  return one < 0 ? one : 3 * one;
}

extern "C" int fmgan_modconv_weight_to_bf16x3(const float* wt, void* wt_split, int cin, int cout, int ktaps, void* stream) {
  if (cout <= 0 || cin <= 0 || ktaps <= 0) return FMGAN_EINVAL;
  if (!wt || !wt_split) return FMGAN_EINVAL;
  const long long pairs = fmgan_modconv_weight_bf16_bytes(cin, cout, ktaps) / 4;
  long long blocks = (pairs + 255) / 256;
  const long long cap = (long long)FMGAN_NUM_CU * 16;
  if (blocks > cap) blocks = cap;
  hipLaunchKernelGGL(modconv_weight_to_bf16x3, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, wt,
                     (unsigned*)wt_split, cin, cout, ktaps);
  return fmgan_check_launch();
}

extern "C" int fmgan_modconv2d_bf16x3_supported(int batch, int cin, int cout, int h, int w, int mode) {
  if (mode != 0 && mode != 1) return 0;
  if (!fmgan_modconv2d_bf16_supported(batch, cin, cout, h, w, mode)) return 0;
  return (long long)(cin >> 4) * 54 * ((cout + 31) / 32 * 32) * 16 < 0xFFFFFFF0LL ? 1 : 0;
}

extern "C" int fmgan_modconv2d_bf16x3(const float* in, const void* wt_split, const float* style, const float* demod,
                                      float* out, int batch, int cin, int cout, int h, int w, int mode, const float* noise,
                                      const float* noise_weight, const float* bias, int noise_batch, int fuse_act,
                                      float alpha, float act_scale, long long out_plane_stride, int out_row_stride,
                                      void* stream) {
  if (batch < 0 || cin <= 0 || cout <= 0 || h <= 0 || w <= 0) return FMGAN_EINVAL;
  if (mode != 0 && mode != 1) return FMGAN_EUNSUPPORTED;
  if (mode != 0 && fuse_act) return FMGAN_EUNSUPPORTED;
  if (batch == 0) return FMGAN_OK;
  if (!fmgan_modconv2d_bf16x3_supported(batch, cin, cout, h, w, mode)) return FMGAN_EUNSUPPORTED;
  if (!in || !wt_split || !style || !out) return FMGAN_EINVAL;
  if (fuse_act && noise && noise_batch != 1 && noise_batch != batch) return FMGAN_EINVAL;
  BFParams p{};
  p.in = in; p.wt = (const unsigned short*)wt_split; p.style = style; p.demod = demod; p.out = out;
  p.batch = batch; p.cin = cin; p.cout = cout; p.h = h; p.w = w;
  if (mode == 1) { p.oh = 2 * h + 1; p.ow = 2 * w + 1; p.gh = h + 1; p.gw = w + 1; }
  else { p.oh = h; p.ow = w; p.gh = h; p.gw = w; }
  if (out_row_stride == 0) out_row_stride = p.ow;
  if (out_plane_stride == 0) out_plane_stride = (long long)p.oh * out_row_stride;
  if (out_row_stride < p.ow || out_plane_stride < (long long)p.oh * out_row_stride) return FMGAN_EINVAL;
  if ((long long)batch * cout * out_plane_stride > (1LL << 40)) return FMGAN_EOVERFLOW;
  p.out_plane_stride = out_plane_stride; p.out_row_stride = out_row_stride;
  p.mp = (cout + 31) / 32 * 32;
  p.noise = noise; p.noise_weight = noise_weight; p.bias = bias; p.noise_batch = noise_batch; p.fuse_act = fuse_act;
  p.alpha = alpha; p.act_scale = act_scale;
  hipStream_t s = (hipStream_t)stream;
  return mode == 0 ? launch_bf16x3<0, 2>(p, s) : launch_bf16x3<1, 1>(p, s);
}
