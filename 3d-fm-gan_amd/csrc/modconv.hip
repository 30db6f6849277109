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
#include <type_traits>
#include "common.h"
#include <cstdio>
#include <cstdlib>

namespace {

// LDS-DMA: `buffer_load_dword(x4) ... lds` — 64 lanes x SIZE bytes land at lds + lane * SIZE (wave-uniform base), from
// rsrc base + voff (per lane, range-checked against the resource: out of range -> zeros) + soff (wave-uniform).
// (Kept in a helper with a device-pass guard: the host pass of hipcc silently drops the host stub of a template
// kernel whose body names this builtin directly.)
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

template <int SIZE, typename RSRC>
__device__ __forceinline__ void dma_to_lds(RSRC rsrc, float* lds, unsigned voff, unsigned soff) {
#if defined(__HIP_DEVICE_COMPILE__)
  typedef __attribute__((address_space(3))) void* lds_void_ptr;
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void_ptr)lds, SIZE, voff, soff, 0, 0);
#endif
}

// A pointer / word the compiler cannot prove wave-uniform (it came through a runtime-indexed segment table), declared
// uniform: buffer resources built from it stay in SGPRs instead of sending every buffer_load through a waterfall loop.
__device__ __forceinline__ const float* uniform_ptr(const float* q) {
  const unsigned long long v = (unsigned long long)q;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return (const float*)(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ unsigned uniform_u32(unsigned v) { return __builtin_amdgcn_readfirstlane(v); }

// ------------------------------------------------------------------ demod
// PRE: W is the per-(o,i) sum of squared taps [cout][cin] (modconv_wsq_f32, cached with the weight) instead of the
// raw weight — the same fma chains in the same order, so both variants give identical bits.
template <bool PRE>
__global__ __launch_bounds__(256) void modconv_demod_f32(const float* __restrict__ W,
                                                         const float* __restrict__ style,
                                                         float* __restrict__ demod, int batch, int cout, int cin,
                                                         int ktaps, float scale, float eps) {
  const int lane = threadIdx.x & 63;
  const int o = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (o >= cout) return;  // wave-uniform
  const float* wo = W + (long long)o * cin * (PRE ? 1 : ktaps);
  constexpr int MAXJ = 8;
  const bool cached = cin <= 64 * MAXJ;
  float wsq[MAXJ];
  if (cached) {
#pragma unroll
    for (int j = 0; j < MAXJ; ++j) {
      const int i = lane + 64 * j;
      float q = 0.f;
      if (i < cin) {
        if constexpr (PRE) q = wo[i];
        else
          for (int t = 0; t < ktaps; ++t) { const float w = wo[i * ktaps + t]; q = fmaf(w, w, q); }
      }
      wsq[j] = q;
    }
  }
  // PRE (inference): one wave per (output channel, sample) — the grid's y extent covers the batch, so the samples' style
  // loads are independent waves instead of `batch` dependent round trips inside one wave.  The raw-weight form keeps one
  // wave per channel (it squares nine taps per weight; repeating that per sample would cost more than it hides).
  for (int b = blockIdx.y; b < batch; b += gridDim.y) {
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
        if constexpr (PRE) q = wo[i];
        else
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

__global__ __launch_bounds__(256) void modconv_wsq_f32(const float* __restrict__ W, float* __restrict__ wsq,
                                                       long long n, int ktaps) {
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (long long)gridDim.x * blockDim.x) {
    float q = 0.f;
    for (int t = 0; t < ktaps; ++t) { const float w = W[idx * ktaps + t]; q = fmaf(w, w, q); }
    wsq[idx] = q;
  }
}

// ------------------------------------------------------------------ weight prep
// kind 0 (forward):                          wt[i][t][o] = scale * W[o][i][t]
// kind 1 (data-gradient of the plain conv):  wt[o][t][i] = scale * W[o][i][ktaps-1-t]   (taps flipped, roles swapped)
// kind 2 (data-gradient of the transposed):  wt[o][t][i] = scale * W[o][i][t]           (roles swapped)
__global__ __launch_bounds__(256) void modconv_weight_prep_f32(const float* __restrict__ W, float* __restrict__ wt,
                                                               int cout, int cin, int ktaps, float scale, int kind) {
  const long long total = (long long)cout * cin * ktaps;
  const int cols = kind == 0 ? cout : cin;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(idx % cols);
    const long long rt = idx / cols;
    const int t = (int)(rt % ktaps), r = (int)(rt / ktaps);
    float v;
    if (kind == 0) v = W[((long long)c * cin + r) * ktaps + t];
    else v = W[((long long)r * cin + c) * ktaps + (kind == 1 ? ktaps - 1 - t : t)];
    wt[idx] = scale * v;
  }
}

// ------------------------------------------------------------------ MFMA conv
struct MCParams {
  const float* in; const float* wt; const float* style; const float* demod; float* out;
  int batch, cin, cout, h, w, oh, ow;
  long long out_plane_stride; int out_row_stride;   // elements; contiguous: oh*ow and ow
  // Position grid = up to 3 rectangular segments tiled independently but launched together (mode 1: the h x w
  // quad grid plus the last output row and the last output column; mode 0: one segment h x w).
  struct Seg {
    int m_off, n_off, gh, gw;        // sub-grid origin and size (positions)
    int th, nb, tw_log2;             // tile rows per sample, samples per tile, log2(tile width)
    int tiles_x, tiles_y, tiles_b;
    unsigned block_end;              // first logical block id after this segment
    int wide;                        // PIPE 1: the halo patch is staged in 16-byte pieces (see "wide patch" in the kernel)
  } seg[3];
  int nseg, o_tiles;
  int ksplit, cin_per_split;         // split-K over input channels (tiny layers); partials go to `ws`
  float* ws;
  const float* noise; const float* noise_weight; const float* bias;
  int noise_batch, fuse_act; float alpha, act_scale;
  // ToRGB fused into the epilogue (MODE 0, one output-channel tile, no split-K): the block holds every channel of
  // its pixels, so rgb[b,c,pix] = sum_o act[b,o,pix] * wmod[b,c,o] (+ bias + skip) is a reduction over
  // the accumulator rows; `out` may then be null (last layer: the activation has no other consumer).
  const float* rgb_wmod;   // [batch][3][cout] = rgb_scale * W[c,o] * rgb_style[b,o], rows >= rgb_c zero
  const float* rgb_bias; const float* rgb_skip; float* rgb_out;
  int rgb_c;
  // PIPE 1 (all staging by LDS-DMA): floats per LDS buffer = weights + patch region + style region, and the patch
  // region's size (both rounded to 64 floats so every DMA piece of 64 lanes x 4 B stays inside its region)
  int lds_buf_floats, lds_patch_floats;
  int lds_style_floats;              // PIPE 1: > 0 = the tile's whole style slice [cin_per_split] is staged once, behind bias/demod
  // experiments only (FMGAN_MC_DEBUG, tools/bench_conv_variants.py --debug): bit 0 = stage the first chunk only (ablation:
  // what the MFMA + operand-fetch loop alone reaches; results are wrong), bit 1 = no barriers in the K loop (wrong too),
  // bit 2 = PIPE 1: issue all DMA pieces of the next chunk up front instead of spreading them over the MFMA stages.
  int debug;
  // experiments only (FMGAN_MC_CLOCKPTR = device address of 16 x uint64): block 0 adds its shader-clock cycles
  // (s_memtime) and its constant-100-MHz ticks (s_memrealtime): their ratio is the clock the chip held during the kernel
  unsigned long long* dbg_clock;
  int exp_flags;     // experiments only (FMGAN_MC_FLAGS, read per call): bit 0 = set-up and epilogue at wave priority 3
};

#ifdef FMGAN_EXPERIMENTS
#define MC_DEBUG(p) ((p).debug)
#define MC_CLOCK(p) ((p).dbg_clock)
#define MC_FLAGS(p) ((p).exp_flags)
#else                                   // product build: the ablation branches fold away
#define MC_DEBUG(p) 0
#define MC_CLOCK(p) false
#define MC_FLAGS(p) 0
#endif

constexpr int MC_KC = 8;  // input channels per LDS chunk

// demod scale + optional StyledConv epilogue (stylegan2.py:371-373) — shared by the conv epilogue and the split-K finish
__device__ __forceinline__ float mc_epilogue(float v, const MCParams& p, float nw, int b, int o, int pix) {
  if (p.demod) v *= p.demod[(long long)b * p.cout + o];
  if (p.fuse_act) {
    const float n = p.noise ? p.noise[(long long)(p.noise_batch == 1 ? 0 : b) * p.oh * p.ow + pix] : 0.f;
    const float bv = p.bias ? p.bias[o] : 0.f;
    v = __fadd_rn(__fadd_rn(v, __fmul_rn(nw, n)), bv);
    v = (v > 0.f ? v : v * p.alpha) * p.act_scale;
  }
  return v;
}

// MODE 0: plain 3x3, one output pixel per position, 9 taps.
// MODE 2: stride-2 valid 3x3 (the reference's downsample branch, and the data-gradient of MODE 1):
//         out[y,x] = sum w[ky][kx] * in[2y+ky, 2x+kx]; same contraction as MODE 0 over a stride-2 patch.
// MODE 1: transposed stride-2 3x3. Position (m,n) owns the 2x2 output quad (2m+py, 2n+px); its four phases
//         use 4+2+2+1 = 9 taps of the same staged weights and only 4 distinct input offsets:
//         out[2m+py, 2n+px] += w[ky][kx] * in[m - ky/2, n - kx/2],  ky = py (mod 2), kx = px (mod 2).
// (A single-barrier variant — LDS double-buffered, the fill of chunk c+1 sliced between chunk c's MFMA stages — was
// built and measured: 95 vs 121 TFLOP/s on the 128x128 tile.  It halves the co-resident blocks per CU and puts the
// staging instructions of the only remaining wave per SIMD in front of its own MFMAs.  Not kept.
// Persistent tiles for the short-K layers (a block walks 2-8 tiles and loads the next tile's first chunk under the
// current tile's last chunk / epilogue) were also measured: 56-88 vs 68-97 TFLOP/s on the 512^2-1024^2 layers —
// holding the prefetched chunk across the epilogue spills 20-70 VGPRs, and the two co-resident blocks per SIMD
// already overlap one block's prologue/epilogue with the other's MFMAs.  Not kept.
// Two LDS images with ONE barrier per chunk, keeping two blocks per CU (possible for the 64- and 32-channel tiles and
// both transposed tiles: <= 39 KB per image): issue(c+1); MFMAs(c); commit(c+1 -> other image); barrier.  Measured
// slower on every layer it applies to (transposed 128^2..512^2: 860/850/940 vs 812/793/893 us; plain 1024^2: 1713 vs
// 1673 us).  Not kept.
// Interleaving the two 32-channel halves of each 64-channel group in the LDS weight image (so the RM = 2 tiles read
// both A operands of a tap with one ds_read_b64: 25-40 % fewer LDS read instructions, same loads and writes) changed
// nothing measurable (+-1 %): the count of A-operand reads is not what limits the MFMA rate.  Not kept.
// 16-channel chunks for the 64- and 32-channel tiles (twice the MFMAs per pair of barriers): with the same register
// budgets 28-94 VGPRs spill (1964 vs 1603 us on the 1024^2 layer); with budgets that fit (one block per CU less)
// 1820 vs 1603 us, 1550 vs 1400 us (512^2), transposed 1033 vs 890 us: co-resident blocks beat longer chunks.  Not kept.
// The 128 x 128 tile on 8 waves (512 threads, each wave 64 x 32, 4 waves per SIMD in a 128-register budget): 24 VGPRs
// spill and it ties with the 4-wave tile (123.6 vs 125.0 TFLOP/s at 64^2).  Not kept.)
// PIPE 0: register-prefetch staging (global -> VGPR one chunk ahead -> ds_write after a barrier; two barriers per chunk).
// PIPE 1: everything a chunk needs — weights (16-byte pieces), the halo patch and the style slice (4-byte pieces) — is
//         written into LDS by `buffer_load ... lds` (LDS-DMA: no VGPR destination, no ds_write, no commit phase), into
//         the buffer the PREVIOUS chunk is not reading; one barrier per chunk.  The style modulation moves from the
//         staging write to the operand fetch (x * s in a VALU op right before the MFMA: the same fp32 product, so both
//         pipelines give identical bits).  Patch slots outside the image rely on the buffer range check: their
//         voffset is parked out of range and the DMA writes zeros (tools/exp/dma_probe.hip).  Chunks that the range check
//         cannot serve (a partial last chunk, tensors of 4 GB and more, ragged Cout) are filled synchronously instead.
template <int MODE, int RM, int RNP, int WM, int WN, bool RGB = false, int MINB = 2, int KC_ = MC_KC, int PIPE = 0>
__global__ __launch_bounds__(256, MINB) void modconv_mfma_f32(const MCParams p) {
  static_assert(!RGB || MODE == 0, "the RGB epilogue belongs to the plain conv");
  constexpr int KC = KC_;
  static_assert(PIPE == 1 || KC == MC_KC, "the register pipeline is built for 8-channel chunks");
  constexpr int BM = 32 * RM * WM;
  constexpr int NPH = MODE == 1 ? 4 : 1;   // output phases per position
  static_assert(WM * WN == 4, "4 waves per block");
  extern __shared__ float smem[];
  float* Ws = smem;                 // [KC][9][BM]
  float* Xs = smem + KC * 9 * BM;   // [nb][KC][PH][PWP]

  // wave index as an SGPR: LDS-DMA destinations (M0), piece guards and tile offsets derived from it stay scalar
  // (left in a VGPR, every `buffer_load ... lds` sat in a waterfall loop with v_readfirstlane)
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int l31 = lane & 31, khalf = lane >> 5;
  unsigned long long clk0 = 0, rt0 = 0;
  if (MC_CLOCK(p)) { clk0 = __builtin_readcyclecounter(); rt0 = wall_clock64(); }
  if (MC_FLAGS(p) & 1) __builtin_amdgcn_s_setprio(3);

  unsigned lb = xcd_remap(blockIdx.x, gridDim.x);
  int si = 0;
  while (si + 1 < p.nseg && lb >= p.seg[si].block_end) ++si;
  if (si > 0) lb -= p.seg[si - 1].block_end;
  // mode 1: taps (bit ky*3+kx) this block's segment needs — row strip: ky = 2, column strip: kx = 2, main grid: all
  const unsigned tapmask = (MODE == 1 && p.nseg == 3) ? (si == 0 ? 0x1C0u : (si == 1 ? 0x124u : 0x1FFu)) : 0x1FFu;
  const int seg_th = p.seg[si].th, seg_nb = p.seg[si].nb, tw_log2 = p.seg[si].tw_log2;
  const int seg_m_end = p.seg[si].m_off + p.seg[si].gh, seg_n_end = p.seg[si].n_off + p.seg[si].gw;
  constexpr int SP = MODE == 2 ? 2 : 1;          // input step per position
  constexpr int ORG = MODE == 2 ? 0 : 1;         // patch origin = SP * first position - ORG
  // Wide patch (PIPE 1, unit-step modes, 32-wide tiles of images whose rows are whole 16-byte groups; host decides): the
  // patch starts FOUR columns left of the tile instead of one and is TW + 8 wide, so every row is ten aligned 16-byte groups,
  // each either inside the image row or outside it as a whole — one `buffer_load_dwordx4 ... lds` lane moves what four
  // dword lanes moved.  The short-K layers issue ~3.5x fewer vector-memory instructions per chunk: measured with s_memtime
  // stamps (profiles/r03_modconv_block_phases.md) a wave of the 1024^2 layer held the matrix pipe 25 % of its K loop with 2.7
  // waves per SIMD in the loop — it sat in the issue of its DMA pieces, and each epilogue store took ~350 cycles to issue
  // behind them (vector memory is one in-order pipeline per CU).
  const bool wide = PIPE == 1 && SP == 1 && p.seg[si].wide != 0;
  const int XORG = wide ? 4 : ORG;               // patch column c holds image column SP * x0 + c - XORG
  const int TW = 1 << tw_log2, PWP = wide ? TW + 8 : SP * (TW - 1) + 3;
  const int o_tile = lb % p.o_tiles;
  unsigned pt = lb / p.o_tiles;
  const int tx_i = pt % p.seg[si].tiles_x; pt /= p.seg[si].tiles_x;
  const int ty_i = pt % p.seg[si].tiles_y;
  const int tb_i = pt / p.seg[si].tiles_y;
  const int ks = blockIdx.y;
  const int o0 = o_tile * BM, b0 = tb_i * seg_nb;
  const int x0 = p.seg[si].n_off + tx_i * TW, y0 = p.seg[si].m_off + ty_i * seg_th;   // first position of the tile
  const int PH = SP * (seg_th - 1) + 3;
  const int plane = PH * PWP;
  const int samp = KC * plane;

  // this lane's position in each of the wave's 32-position groups
  int pbase[RNP], pos_b[RNP], pos_y[RNP], pos_x[RNP];
#pragma unroll
  for (int g = 0; g < RNP; ++g) {
    const int pos = (wn * RNP + g) * 32 + l31;
    const int tx = pos & (TW - 1), r = pos >> tw_log2;
    const int ty = r % seg_th, nbi = r / seg_th;
    // (a thin segment's tile may hold fewer than BN positions — launch_cfg shrinks it until its patch fits the main
    // segment's LDS image: the rows past nb * th read sample nb - 1's patch and are never stored)
    pbase[g] = min(nbi, seg_nb - 1) * samp + SP * ty * PWP + SP * tx + (XORG - ORG);
    pos_b[g] = nbi < seg_nb ? b0 + nbi : p.batch; pos_y[g] = y0 + ty; pos_x[g] = x0 + tx;
  }

  f32x16 acc[RM][RNP][NPH];
#pragma unroll
  for (int a = 0; a < RM; ++a)
#pragma unroll
    for (int b = 0; b < RNP; ++b)
#pragma unroll
      for (int c = 0; c < NPH; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][b][c][r] = 0.f;

  const bool wvec = (p.cout & 3) == 0;
  const int i_begin = ks * p.cin_per_split;
  const int i_end = min(p.cin, i_begin + p.cin_per_split);
  const int hw = p.h * p.w;

  // ---- staging plan, fixed for the whole K loop.
  // Patch: thread owns up to NU spatial slots (sample, row, col) of the halo patch and moves all KC channels of
  // each; weights: NWV float4 per thread.  Everything a chunk needs is loaded into registers one chunk AHEAD
  // (while the previous chunk is on the matrix pipe) and written to LDS after the barrier.
  constexpr int BNP = 32 * RNP * WN;
  constexpr int NU = ((SP * (BNP / 32 - 1) + 3) * (SP * 31 + 3) + 255) / 256;   // covers a full-width main tile
  constexpr int NWV = (KC * 9 * (BM / 4) + 255) / 256;
  const int spatial = seg_nb * plane;                              // patch slots per channel
  // element offsets are 32-bit, relative to the tile's first sample (host checks nb*cin*h*w < 2^31)
  const float* in_b0 = uniform_ptr(p.in + (long long)b0 * p.cin * hw);
  const float* style_b0 = uniform_ptr(p.style + (long long)b0 * p.cin);
  // PIPE 1, tiles of one sample: the tile's bias and demodulation values go to LDS now, one vector load per wave, and the
  // epilogue reads them back with ds_read_b128 — it used to issue 32 two-address vector loads per wave and row group
  // right when the block's stores want the (in-order, shared) vector-memory pipeline.  Visible to every wave after the
  // K loop's first barrier.
  // (The load is issued here and its value written to LDS after the first chunk's DMA pieces are out, so its latency hides
  // under the wait for that chunk.)
  const bool epi_lds = PIPE == 1 && seg_nb == 1 && p.ksplit == 1;
  float* Es = smem + 2 * p.lds_buf_floats;        // [2][BM]: bias, demod
  float epi_v = 0.f;
  if constexpr (PIPE == 1) {
    if (epi_lds && tid < 2 * BM) {
      const bool isd = tid >= BM;
      const int oc = min(o0 + (isd ? tid - BM : tid), p.cout - 1);
      epi_v = isd ? 1.f : 0.f;
      if (isd) { if (p.demod) epi_v = p.demod[(long long)min(b0, p.batch - 1) * p.cout + oc]; }
      else if (MODE == 0 && p.fuse_act && p.bias) epi_v = p.bias[oc];
    }
  }
  int gofs[NU], lofs[NU], sofs[NU]; bool inb[NU];
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    const int q = tid + 256 * u;
    const int c = q % PWP;
    int t = q / PWP;
    const int r = t % PH, nbi = t / PH;
    const int b = b0 + nbi, y = SP * y0 + r - ORG, x = SP * x0 + c - ORG;
    inb[u] = q < spatial && b < p.batch && y >= 0 && y < p.h && x >= 0 && x < p.w;
    gofs[u] = inb[u] ? (nbi * p.cin * p.h + y) * p.w + x : 0;
    sofs[u] = nbi * p.cin;
    lofs[u] = q < spatial ? nbi * samp + r * PWP + c : -1;
  }
  float xv[NU][KC];
  f32x4 wv[NWV];
  // Fast path: buffer loads.  The per-thread byte offsets are fixed for the whole K loop (voffset), the chunk only
  // moves a scalar (soffset), and the hardware range check returns 0 for patch slots that fall outside the image
  // (their voffset is parked past the end) — no per-load 64-bit address math, no branches.  A partial last chunk
  // (cin not a multiple of 8) takes the guarded path.
  // (An ablation showed the MFMA stages alone reach 140 TFLOP/s and the staging INSTRUCTIONS, not the bytes, are
  // what the co-resident block has to hide.)
  // The range check compares voffset (+4) with num_records and ignores soffset: a slot is "out of range" only through
  // its parked voffset 0xFFFFFFF0, so the buffers must be SHORTER than that (with num_records >= 0xFFFFFFF4 a parked
  // slot would pass the check and read base + 0xFFFFFFF0 + soffset).  Larger tensors take the guarded path.
  constexpr long long PARK = 0xFFFFFFF0LL;
  const bool fastw = wvec && o0 + BM <= p.cout && (long long)p.cin * 9 * p.cout * 4 < PARK;
  const bool fastx = (long long)min(seg_nb, p.batch - b0) * p.cin * hw * 4 < PARK;
  const unsigned w_bytes = uniform_u32((unsigned)((long long)p.cin * 9 * p.cout * 4));
  const unsigned x_bytes = uniform_u32((unsigned)((long long)min(seg_nb, p.batch - b0) * p.cin * hw * 4));
  const auto rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.wt), 0, w_bytes, 0x00020000);
  const auto rsrc_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in_b0), 0, x_bytes, 0x00020000);
  unsigned wofs[NWV], xofs[NU];
#pragma unroll
  for (int j = 0; j < NWV; ++j) {
    const int idx = tid + 256 * j;
    const int row = idx / (BM / 4), c4 = idx % (BM / 4);
    wofs[j] = idx < KC * 9 * (BM / 4) ? (unsigned)((row * p.cout + o0 + c4 * 4) * 4) : 0xFFFFFFF0u;
  }
#pragma unroll
  for (int u = 0; u < NU; ++u) xofs[u] = inb[u] ? (unsigned)gofs[u] * 4u : 0xFFFFFFF0u;
  auto issue = [&](int i0) {
    if (fastw && i0 + KC <= i_end) {
      const unsigned soff = (unsigned)(i0 * 9 * p.cout * 4);
#pragma unroll
      for (int j = 0; j < NWV; ++j) {
        const auto t = __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, wofs[j], soff, 0);
        wv[j] = __builtin_bit_cast(f32x4, t);
      }
    } else {
#pragma unroll
      for (int j = 0; j < NWV; ++j) {
        const int idx = tid + 256 * j;
        const int row = idx / (BM / 4), c4 = idx % (BM / 4);
        const int i = i0 + row / 9, o = o0 + c4 * 4;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (idx < KC * 9 * (BM / 4) && i < i_end) {
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
        wv[j] = v;
      }
    }
    if (fastx && i0 + KC <= i_end) {
#pragma unroll
      for (int kc = 0; kc < KC; ++kc) {
        const unsigned soff = (unsigned)((i0 + kc) * hw * 4);
#pragma unroll
        for (int u = 0; u < NU; ++u)
          xv[u][kc] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc_x, xofs[u], soff, 0));
      }
    } else {
#pragma unroll
      for (int u = 0; u < NU; ++u)
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) {
          const int i = i0 + kc;
          xv[u][kc] = (inb[u] && i < i_end) ? in_b0[gofs[u] + i * hw] : 0.f;
        }
    }
  };
  auto commit = [&](int i0) {
#pragma unroll
    for (int j = 0; j < NWV; ++j) {
      const int idx = tid + 256 * j;
      if (idx < KC * 9 * (BM / 4)) *reinterpret_cast<f32x4*>(Ws + idx * 4) = wv[j];
    }
    // modulate on the way in: x[b,i] * style[b,i] (the style loads are L1 hits, issued ahead of the weight writes)
#pragma unroll
    for (int u = 0; u < NU; ++u)
      if (lofs[u] >= 0) {
        float sv[KC];
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) sv[kc] = (inb[u] && i0 + kc < i_end) ? style_b0[sofs[u] + i0 + kc] : 0.f;
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) Xs[lofs[u] + kc * plane] = xv[u][kc] * sv[kc];
      }
    // patches larger than a main tile's (sample-packed tiny layers, thin edge segments): synchronous remainder
    for (int q = tid + 256 * NU; q < spatial; q += 256) {
      const int c = q % PWP;
      int t = q / PWP;
      const int r = t % PH, nbi = t / PH;
      const int b = b0 + nbi, y = SP * y0 + r - ORG, x = SP * x0 + c - ORG;
      const bool ok = b < p.batch && y >= 0 && y < p.h && x >= 0 && x < p.w;
      for (int kc = 0; kc < KC; ++kc) {
        const int i = i0 + kc;
        float v = 0.f;
        if (ok && i < i_end) {
          const long long ch = (long long)b * p.cin + i;
          v = p.in[(ch * p.h + y) * p.w + x] * p.style[ch];
        }
        Xs[nbi * samp + kc * plane + r * PWP + c] = v;
      }
    }
  };

  // LDS -> registers, one pipeline stage = one row of taps (MODE 0: 3 taps x channel pair) or one channel pair
  // (MODE 1: its 9 taps share 4 input offsets).  Stage s+1 is fetched while stage s is on the matrix pipe.
  constexpr int NTA = MODE == 1 ? 9 : 3;     // A (weight) values per stage per 32-row tile
  constexpr int NTB = MODE == 1 ? 4 : 3;     // B (input) values per stage per position group
  constexpr int NSTAGE = (KC / 2) * (MODE == 1 ? 1 : 3);
  struct Ops { float a[NTA][RM]; float b[NTB][RNP]; float s[RNP]; };
  const float* Wc = Ws;            // LDS image the current chunk is read from (PIPE 1: toggles between two buffers)
  const float* Xc = Xs;
  const float* Sc = nullptr;       // PIPE 1: style slice [nb][KC] of the chunk
  int sbase[RNP];                  // PIPE 1: this lane's sample row in the style slice
#pragma unroll
  for (int g = 0; g < RNP; ++g) sbase[g] = (pbase[g] / samp) * KC;
  auto fetch = [&](Ops& o, int st) {
    const int kk = MODE == 1 ? st : st / 3, ky = MODE == 1 ? 0 : st % 3;
    const int kc = 2 * kk + khalf;
    const float* wrow = Wc + (kc * 9 + ky * 3) * BM + wm * 32 * RM + l31;
    const float* xrow = Xc + kc * plane;
#pragma unroll
    for (int t = 0; t < NTA; ++t)
#pragma unroll
      for (int m = 0; m < RM; ++m) o.a[t][m] = wrow[t * BM + m * 32];
    if constexpr (PIPE == 1) {
      if (MODE == 1 || ky == 0) {
#pragma unroll
        for (int g = 0; g < RNP; ++g) o.s[g] = Sc[sbase[g] + kc];
      }
    }
#pragma unroll
    for (int j = 0; j < NTB; ++j) {
      // MODE 0: offset (ky, kx = j); MODE 1: j -> (ro, co) = (1,1) (1,0) (0,1) (0,0)
      const int ro = MODE == 1 ? 1 - (j >> 1) : ky;
      const int co = MODE == 1 ? 1 - (j & 1) : j;
#pragma unroll
      for (int g = 0; g < RNP; ++g) o.b[j][g] = xrow[pbase[g] + ro * PWP + co];
    }
  };
  // PIPE 1: x * style[b, i] on the way to the matrix pipe (PIPE 0 did it on the way into LDS)
  auto modulate = [&](Ops& o) {
    if constexpr (PIPE == 1) {
#pragma unroll
      for (int j = 0; j < NTB; ++j)
#pragma unroll
        for (int g = 0; g < RNP; ++g) o.b[j][g] *= o.s[g];
    }
  };
  auto mma = [&](const Ops& o, auto all_taps) {
    if constexpr (MODE != 1) {
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int m = 0; m < RM; ++m)
#pragma unroll
          for (int g = 0; g < RNP; ++g)
            acc[m][g][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(o.a[t][m], o.b[t][g], acc[m][g][0], 0, 0, 0);
    } else {
      // (tap = ky*3+kx, input-offset index j, phase = py*2+px)
      constexpr int T[9][3] = {{0, 0, 0}, {2, 1, 0}, {6, 2, 0}, {8, 3, 0}, {1, 0, 1}, {7, 2, 1}, {3, 0, 2}, {5, 1, 2}, {4, 0, 3}};
      // The two thin segments (last output row: positions m = h; last output column: n = w) only ever see the taps
      // whose input lies inside the image — ky = 2 resp. kx = 2: every other product has a zero operand.  Their
      // blocks skip those MFMAs (a wave-uniform bit per tap: same sums, bit for bit) and finish three times sooner.
#pragma unroll
      for (int q = 0; q < 9; ++q)
        if (decltype(all_taps)::value || (tapmask & (1u << T[q][0])))
#pragma unroll
          for (int m = 0; m < RM; ++m)
#pragma unroll
            for (int g = 0; g < RNP; ++g)
              acc[m][g][T[q][2]] =
                  __builtin_amdgcn_mfma_f32_32x32x2f32(o.a[T[q][0]][m], o.b[T[q][1]][g], acc[m][g][T[q][2]], 0, 0, 0);
    }
  };

  // ---- contract: operands of stage st+1 are fetched while stage st is on the matrix pipe.
  // (hipcc otherwise sinks every ds_read next to its MFMA and waits lgkmcnt(0) in front of each group of four:
  // the sched_barriers keep "issue all reads of stage st+1, then run stage st's MFMAs" as written.)
  // hook(st): extra issue work placed in front of stage st's MFMAs (PIPE 1: a slice of the next chunk's DMA pieces)
  // Two operand sets used alternately (NSTAGE is even).  The MFMA cluster of a stage runs at raised wave priority
  // (s_setprio): with three or more waves per SIMD the arbiter otherwise lets the other waves' LDS reads and address
  // VALU in between the matrix instructions — 1024^2 plain 1707 -> 1628 us, transposed 64^2 811 -> 782, the 17 layers
  // 11.87 -> 11.70 ms; little effect on the two-wave 128 x 256 tile.
  static_assert(NSTAGE % 2 == 0, "stages are processed in pairs");
  auto contract = [&](auto&& hook, auto all_taps) {
    Ops o0, o1;
    fetch(o0, 0);
#pragma unroll
    for (int st = 0; st < NSTAGE; st += 2) {
      __builtin_amdgcn_sched_barrier(0);
      hook(st);
      fetch(o1, st + 1);
      if constexpr (PIPE == 1 && MODE != 1) {
        // the style value of a channel pair is read with its first tap row and reused for the other two
        if ((st + 1) % 3 != 0) {
#pragma unroll
          for (int g = 0; g < RNP; ++g) o1.s[g] = o0.s[g];
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      modulate(o0);
      __builtin_amdgcn_s_setprio(1);
      mma(o0, all_taps);
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      hook(st + 1);
      if (st + 2 < NSTAGE) {
        fetch(o0, st + 2);
        if constexpr (PIPE == 1 && MODE != 1) {
          if ((st + 2) % 3 != 0) {
#pragma unroll
            for (int g = 0; g < RNP; ++g) o0.s[g] = o1.s[g];
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      modulate(o1);
      __builtin_amdgcn_s_setprio(1);
      mma(o1, all_taps);
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  unsigned long long clk1 = 0, clk2 = 0, clkw = 0, clke = 0, clkl = 0;
  if (MC_CLOCK(p)) clk1 = __builtin_readcyclecounter();
  if (MC_FLAGS(p) & 1) __builtin_amdgcn_s_setprio(0);
  if constexpr (PIPE == 0) {
    if (i_begin < i_end) issue(i_begin);
    for (int i0 = i_begin; i0 < i_end; i0 += KC) {
      const bool live = !(MC_DEBUG(p) & 1) || i0 == i_begin;
      if (!(MC_DEBUG(p) & 2)) __syncthreads();            // every wave is done reading the previous chunk
      if (live) commit(i0);
      if (!(MC_DEBUG(p) & 2)) __syncthreads();
      __builtin_amdgcn_sched_barrier(0);
      if (live && i0 + KC < i_end) issue(i0 + KC);   // in flight during this chunk's MFMAs
      contract([](int) {}, std::false_type{});
    }
  } else {
    constexpr unsigned PARKED = 0xFFFFFFF0u;
    // floats of one weight image, rounded up to whole 16-byte DMA pieces (64 lanes x 16 B = 256 floats; the lanes past the
    // image are parked and write zeros into the padding)
    constexpr int WSZ = (KC * 9 * BM + 255) / 256 * 256;
    constexpr int WPIECES = WSZ / 256;
    constexpr int NWP = (WPIECES + 3) / 4;                  // per wave
    constexpr int NUP = ((KC * (SP * (BNP / 32 - 1) + 3) * (SP * 31 + 3) + 63) / 64 + 3) / 4;   // patch pieces per wave, main tile
    const int xs_total = seg_nb * samp;                     // patch floats
    const int x_piece_floats = wide ? 256 : 64;             // one DMA piece = 64 lanes x 16 or 4 bytes
    const int x_pieces = (xs_total + x_piece_floats - 1) / x_piece_floats;
    const int s_pieces = (seg_nb * KC + 63) >> 6;
    const int bstride = p.lds_buf_floats;
    const bool fastc = fastw && fastx;                      // DMA-servable tensors (chunk completeness checked per chunk)
    const unsigned s_bytes = (unsigned)(min(seg_nb, p.batch - b0) * p.cin * 4);
    const auto rsrc_s = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(style_b0), 0, s_bytes, 0x00020000);
    // patch element (linear LDS index) -> byte offset from in_b0 of channel i0, or PARKED
    auto x_voff = [&](int e) -> unsigned {
      if (e >= xs_total) return PARKED;
      const int nbi = e / samp, rem = e - nbi * samp;
      const int kc = rem / plane, q = rem - kc * plane;
      const int r = q / PWP, c = q - r * PWP;
      const int b = b0 + nbi, y = SP * y0 + r - ORG, x = SP * x0 + c - XORG;
      const bool ok = b < p.batch && y >= 0 && y < p.h && x >= 0 && x < p.w;
      return ok ? (unsigned)(((nbi * p.cin + kc) * p.h + y) * p.w + x) * 4u : PARKED;
    };
    // wide patch: 16-byte group g16 (four floats, linear LDS index 4 * g16) -> byte offset of its first float, or PARKED;
    // x is a multiple of 4 and so is p.w (host), so a group never straddles the end of an image row
    auto x_voff16 = [&](int g16) -> unsigned { return x_voff(4 * g16); };
    auto s_voff = [&](int e) -> unsigned {
      const int nbi = e / KC, kc = e - nbi * KC;
      return (nbi < seg_nb && b0 + nbi < p.batch) ? (unsigned)(nbi * p.cin + kc) * 4u : PARKED;
    };
    unsigned wvo[NWP], xvo[NUP];
#pragma unroll
    for (int j = 0; j < NWP; ++j) {
      const int idx = (4 * j + wave) * 64 + lane;           // float4 index in the image [KC*9][BM/4]
      const int row = idx / (BM / 4), c4 = idx % (BM / 4);
      wvo[j] = idx < KC * 9 * (BM / 4) ? (unsigned)((row * p.cout + o0 + c4 * 4) * 4) : PARKED;
    }
#pragma unroll
    for (int u = 0; u < NUP; ++u)
      xvo[u] = 4 * u + wave < x_pieces ? (wide ? x_voff16((4 * u + wave) * 64 + lane) : x_voff((4 * u + wave) * 64 + lane)) : PARKED;
    const unsigned svo = s_voff(wave * 64 + lane);          // first style piece of this wave (all of them when nb*KC <= 256)

    // This wave's DMA pieces of one chunk, numbered k = 0 .. NPC-1: weights, patch, then the style slice together with
    // whatever a larger-than-main-tile patch / a sample-packed style slice has left.
    constexpr int NPC = NWP + NUP + 1;
    constexpr int PER_STAGE = (NPC + NSTAGE - 1) / NSTAGE;
    auto dma_piece = [&](int k, int i0, int buf) {
      float* Wb = smem + buf * bstride;
      float* Xb = Wb + WSZ;
      const unsigned soff_x = (unsigned)(i0 * hw * 4);
      if (k < NWP) {
        if (4 * k + wave < WPIECES)
          dma_to_lds<16>(rsrc_w, Wb + (4 * k + wave) * 256, wvo[k < NWP ? k : 0], (unsigned)(i0 * 9 * p.cout * 4));
      } else if (k < NWP + NUP) {
        const int u = k - NWP;
        if (4 * u + wave < x_pieces) {
          if (wide) dma_to_lds<16>(rsrc_x, Xb + (4 * u + wave) * 256, xvo[(u >= 0 && u < NUP) ? u : 0], soff_x);
          else dma_to_lds<4>(rsrc_x, Xb + (4 * u + wave) * 64, xvo[(u >= 0 && u < NUP) ? u : 0], soff_x);
        }
      } else {
        float* Sb = Xb + p.lds_patch_floats;
        for (int pc = 4 * NUP + wave; pc < x_pieces; pc += 4) {     // patches larger than a main tile's
          if (wide) dma_to_lds<16>(rsrc_x, Xb + pc * 256, x_voff16(pc * 64 + lane), soff_x);
          else dma_to_lds<4>(rsrc_x, Xb + pc * 64, x_voff(pc * 64 + lane), soff_x);
        }
        if (wave < s_pieces)
          dma_to_lds<4>(rsrc_s, Sb + wave * 64, svo, (unsigned)(i0 * 4));
        for (int pc = 4 + wave; pc < s_pieces; pc += 4)
          dma_to_lds<4>(rsrc_s, Sb + pc * 64, s_voff(pc * 64 + lane), (unsigned)(i0 * 4));
      }
    };
    auto dma_ok = [&](int i0) { return fastc && i0 + KC <= i_end; };

    auto stage = [&](int i0, int buf) {
      float* Wb = smem + buf * bstride;
      float* Xb = Wb + WSZ;
      float* Sb = Xb + p.lds_patch_floats;
      if (dma_ok(i0)) {
#pragma unroll
        for (int k = 0; k < NPC; ++k) dma_piece(k, i0, buf);
      } else {
        // guarded synchronous fill (same images, zeros where the DMA's range check would have produced them)
        for (int idx = tid; idx < KC * 9 * (BM / 4); idx += 256) {
          const int row = idx / (BM / 4), c4 = idx % (BM / 4);
          const int i = i0 + row / 9, o = o0 + c4 * 4;
          f32x4 v = {0.f, 0.f, 0.f, 0.f};
          if (i < i_end) {
            const float* src = p.wt + ((long long)i * 9 + row % 9) * p.cout + o;
            if (o + 0 < p.cout) v.x = src[0];
            if (o + 1 < p.cout) v.y = src[1];
            if (o + 2 < p.cout) v.z = src[2];
            if (o + 3 < p.cout) v.w = src[3];
          }
          *reinterpret_cast<f32x4*>(Wb + idx * 4) = v;
        }
        for (int e = tid; e < x_pieces * x_piece_floats; e += 256) {
          float v = 0.f;
          if (e < xs_total) {
            const int nbi = e / samp, rem = e - nbi * samp;
            const int kc = rem / plane, q = rem - kc * plane;
            const int r = q / PWP, c = q - r * PWP;
            const int b = b0 + nbi, y = SP * y0 + r - ORG, x = SP * x0 + c - XORG, i = i0 + kc;
            if (b < p.batch && y >= 0 && y < p.h && x >= 0 && x < p.w && i < i_end)
              v = p.in[(((long long)b * p.cin + i) * p.h + y) * p.w + x];
          }
          Xb[e] = v;
        }
        for (int e = tid; e < seg_nb * KC; e += 256) {
          const int nbi = e / KC, kc = e - nbi * KC;
          Sb[e] = (b0 + nbi < p.batch && i0 + kc < i_end) ? p.style[(long long)(b0 + nbi) * p.cin + i0 + kc] : 0.f;
        }
      }
    };

    // Lean K loop for the common case — every chunk complete and DMA-servable, the patch and the style slice within the
    // main tile's piece budget, all nine taps live, no debug switches: the general loop below re-tests all of that per
    // chunk and per stage (spread / dma_ok / debug bits, the larger-than-main-tile piece loops with their div/mod
    // address maths, one wave-uniform branch per tap in MODE 1), ~1300 scalar and vector instructions around the 60-72
    // MFMAs of a chunk on the small tiles.  Same DMA pieces, same MFMA order: identical bits.
    // (FMGAN_MC_DEBUG bit 3 forces the general loop: A/B measurements.  Unrolling the lean loop by two chunks, so that the
    // double buffer's index is a compile-time constant in each half, was measured too: no gain — 12.19 vs 12.10-12.13 ms over
    // the 17 layers — and 24-31 more spilled SGPRs; not kept.)
    const bool lean = fastc && MC_DEBUG(p) == 0 && i_begin < i_end && ((i_end - i_begin) % KC) == 0 && x_pieces <= 4 * NUP &&
                      s_pieces <= 4 && tapmask == 0x1FFu;
    const bool style_once = lean && seg_nb == 1 && p.lds_style_floats > 0;
    float* Ss = Es + 2 * BM;
    if (lean) {
      auto piece = [&](int k, int i0, int bufi) {
        float* Wb = smem + bufi * bstride;
        float* Xb = Wb + WSZ;
        if (k < NWP) {
          if (WPIECES % 4 == 0 || 4 * k + wave < WPIECES)
            dma_to_lds<16>(rsrc_w, Wb + (4 * k + wave) * 256, wvo[k < NWP ? k : 0], (unsigned)(i0 * 9 * p.cout * 4));
        } else if (k < NWP + NUP) {
          const int u = k - NWP;
          if (4 * u + wave < x_pieces) {
            if (wide) dma_to_lds<16>(rsrc_x, Xb + (4 * u + wave) * 256, xvo[(u >= 0 && u < NUP) ? u : 0], (unsigned)(i0 * hw * 4));
            else dma_to_lds<4>(rsrc_x, Xb + (4 * u + wave) * 64, xvo[(u >= 0 && u < NUP) ? u : 0], (unsigned)(i0 * hw * 4));
          }
        } else if (!style_once && wave < s_pieces) {
          dma_to_lds<4>(rsrc_s, Xb + p.lds_patch_floats + wave * 64, svo, (unsigned)(i0 * 4));
        }
      };
      // style slice of the whole K range, once (one DMA piece per wave and chunk less; the loads fly with the first chunk)
      float sty0 = 0.f, sty1 = 0.f;
      if (style_once) {
        const int n = i_end - i_begin;
        if (tid < n) sty0 = style_b0[i_begin + tid];
        if (tid + 256 < n) sty1 = style_b0[i_begin + tid + 256];
      }
#pragma unroll
      for (int k = 0; k < NPC; ++k) piece(k, i_begin, 0);
      if (epi_lds && tid < 2 * BM) Es[tid] = epi_v;
      if (style_once) {
        if (tid < p.lds_style_floats) Ss[tid] = sty0;
        if (tid + 256 < p.lds_style_floats) Ss[tid + 256] = sty1;
      }
      int bufl = 0;
      for (int i0 = i_begin; i0 < i_end; i0 += KC, bufl ^= 1) {
        const unsigned long long tw0 = MC_CLOCK(p) ? __builtin_readcyclecounter() : 0;
        __builtin_amdgcn_s_waitcnt(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (MC_CLOCK(p)) clkw += __builtin_readcyclecounter() - tw0;
        const int i1 = i0 + KC;
        const bool more = i1 < i_end;
        Wc = smem + bufl * bstride;
        Xc = Wc + WSZ;
        Sc = style_once ? Ss + (i0 - i_begin) : Xc + p.lds_patch_floats;
        contract([&](int st) {
          if (more) {
#pragma unroll
            for (int q = 0; q < PER_STAGE; ++q)
              if (st * PER_STAGE + q < NPC) piece(st * PER_STAGE + q, i1, bufl ^ 1);
          }
        }, std::true_type{});
      }
    }
    if (!lean && i_begin < i_end) {
      stage(i_begin, 0);
      if (epi_lds && tid < 2 * BM) Es[tid] = epi_v;
    }
    int buf = 0;
    for (int i0 = lean ? i_end : i_begin; i0 < i_end; i0 += KC, buf ^= 1) {
      // this wave's DMA pieces / LDS stores of chunk i0 have landed; after the barrier everyone's have, and every wave
      // has finished reading the other buffer (its ds_reads were retired before the MFMAs that consumed them)
      __builtin_amdgcn_s_waitcnt(0);
      if (!(MC_DEBUG(p) & 2)) __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      const int i1 = i0 + KC;
      const bool more = i1 < i_end && !(MC_DEBUG(p) & 1);
      // the next chunk's pieces: spread over this chunk's MFMA stages (each DMA issue then hides behind a matrix
      // instruction of the same wave), or all at once for a chunk the DMA cannot serve / with debug bit 2
      const bool spread = more && dma_ok(i1) && !(MC_DEBUG(p) & 4);
      if (more && !spread) stage(i1, buf ^ 1);
      __builtin_amdgcn_sched_barrier(0);
      Wc = smem + buf * bstride;
      Xc = Wc + WSZ;
      Sc = Xc + p.lds_patch_floats;
      contract([&](int st) {
        if (spread) {
#pragma unroll
          for (int q = 0; q < PER_STAGE; ++q)
            if (st * PER_STAGE + q < NPC) dma_piece(st * PER_STAGE + q, i1, buf ^ 1);
        }
      }, std::false_type{});
    }
  }

  if (MC_CLOCK(p)) clk2 = __builtin_readcyclecounter();
  if (MC_FLAGS(p) & 1) __builtin_amdgcn_s_setprio(3);
  // ---- epilogue.  All loads first (bias, demod, noise: clamped indices, no branches — a load inside a divergent
  // `if` costs one full memory round trip per element, 64 of them in a row per thread), then arithmetic on the
  // accumulators in place, then predicated stores.
  const bool partial = p.ksplit > 1;
  const bool actf = MODE == 0 && p.fuse_act && !partial;
  const bool use_demod = p.demod != nullptr && !partial;
  const float nw = (actf && p.noise && p.noise_weight) ? p.noise_weight[0] : 0.f;
  float* dstbase = partial ? p.ws + (long long)ks * p.batch * p.cout * p.oh * p.ow : p.out;
  const long long dps = partial ? (long long)p.oh * p.ow : p.out_plane_stride;
  const int drs = partial ? p.ow : p.out_row_stride;
  const int orow = o0 + wm * 32 * RM + 4 * khalf;     // this lane's first output channel; row r adds (r&3) + 8*(r>>2)
  // per position group: validity, destination, noise term (loaded up front, clamped indices)
  bool vgs[RNP]; float nzs[RNP];
#pragma unroll
  for (int g = 0; g < RNP; ++g) {
    const int b = pos_b[g], bc = min(b, p.batch - 1);
    if constexpr (MODE != 1) {
      const int y = pos_y[g], x = pos_x[g];
      vgs[g] = b < p.batch && y < p.oh && x < p.ow;
      const int pix = vgs[g] ? y * p.ow + x : 0;
      nzs[g] = (actf && p.noise) ? __fmul_rn(nw, p.noise[(long long)(p.noise_batch == 1 ? 0 : bc) * p.oh * p.ow + pix]) : 0.f;
    } else {
      vgs[g] = b < p.batch && pos_y[g] < seg_m_end && pos_x[g] < seg_n_end;
      nzs[g] = 0.f;
    }
  }
  // destination of a position group's first channel (generic store path)
  auto dpos_of = [&](int g) -> float* {
    const int bc = min(pos_b[g], p.batch - 1);
    const int sc = MODE == 1 ? 2 : 1;
    return dstbase + (long long)bc * p.cout * dps + (long long)(vgs[g] ? sc * pos_y[g] : 0) * drs + (vgs[g] ? sc * pos_x[g] : 0);
  };
  // Buffer stores (every tile whose channels all exist and whose samples' output slabs fit 32-bit offsets — every layer of
  // the model): per-lane byte offset fixed per position group (parked for positions outside the grid: the range check
  // drops the store), channel / phase row as a scalar offset.  One SALU op + one store per value instead of a 64-bit
  // multiply-add chain and an exec-mask branch (the generic form below costs ~25 instructions per value).
  const long long slab = (long long)p.cout * dps * 4;                         // bytes of one sample's output
  bool bstore = dstbase != nullptr && o0 + BM <= p.cout && (long long)seg_nb * slab < 0xFFFFFFF0LL;
  if constexpr (MODE == 1) bstore = bstore && 2 * seg_m_end <= p.oh && 2 * seg_n_end <= p.ow;   // every quad complete
  unsigned vofs[RNP];
  const float* obase = dstbase ? uniform_ptr(dstbase + (long long)b0 * p.cout * dps) : nullptr;
  const auto rsrc_o = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(obase), 0,
      uniform_u32(bstore ? (unsigned)((long long)min(seg_nb, p.batch - b0) * slab) : 0u), 0x00020000);
#pragma unroll
  for (int g = 0; g < RNP; ++g) {
    const long long e = (long long)(pos_b[g] - b0) * p.cout * dps + (long long)(4 * khalf) * dps +
                        (long long)(MODE == 1 ? 2 * pos_y[g] : pos_y[g]) * drs + (MODE == 1 ? 2 * pos_x[g] : pos_x[g]);
    vofs[g] = (bstore && vgs[g]) ? (unsigned)(e * 4) : 0xFFFFFFF0u;
  }
  // activation constants pinned in SGPRs (left to the compiler they are re-read from the kernel arguments, with a wait,
  // for every value)
  unsigned alpha_bits = __float_as_uint(p.alpha), ascale_bits = __float_as_uint(p.act_scale);
  asm volatile("" : "+s"(alpha_bits), "+s"(ascale_bits));
  const float alpha = __uint_as_float(alpha_bits), ascale = __uint_as_float(ascale_bits);
  const unsigned dps4 = uniform_u32((unsigned)(dps * 4)), drs4 = uniform_u32((unsigned)drs * 4u);   // used by bstore only
  if (MC_CLOCK(p)) clke = __builtin_readcyclecounter();
  // One 32-channel row group at a time: its bias / demodulation values are loaded together (16 + 16 registers live
  // beside the accumulators instead of 32 * RM + 32 * RM: what lets the 128 x 128 tile fit three blocks per CU).
  auto rows = [&](auto act_c, auto bst_c) {
    constexpr bool ACT = decltype(act_c)::value, BST = decltype(bst_c)::value;
#pragma unroll
    for (int m = 0; m < RM; ++m) {
      float bias_m[16], dm_m[16];
      if (epi_lds) {
        const float* eb = Es + (orow - o0) + m * 32;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          bias_m[r] = ACT ? eb[(r & 3) + 8 * (r >> 2)] : 0.f;
          dm_m[r] = eb[BM + (r & 3) + 8 * (r >> 2)];
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int oc = min(orow + m * 32 + (r & 3) + 8 * (r >> 2), p.cout - 1);
          bias_m[r] = (ACT && p.bias) ? p.bias[oc] : 0.f;
          dm_m[r] = 1.f;
        }
      }
      const unsigned so_m = (unsigned)(o0 + wm * 32 * RM + m * 32) * dps4;     // scalar: first channel of the row group
#pragma unroll
      for (int g = 0; g < RNP; ++g) {
        const int bc = min(pos_b[g], p.batch - 1);
        if (!epi_lds && use_demod && (g == 0 || seg_nb > 1)) {        // tiles of one sample (every large layer) load demod once
#pragma unroll
          for (int r = 0; r < 16; ++r)
            dm_m[r] = p.demod[(long long)bc * p.cout + min(orow + m * 32 + (r & 3) + 8 * (r >> 2), p.cout - 1)];
        }
        const bool vg = vgs[g];
        float* dpos = BST ? nullptr : dpos_of(g);
        if (MC_CLOCK(p) && m == 0 && g == 0) {          // every load of the epilogue has landed (stamps only)
          __builtin_amdgcn_sched_barrier(0);
          __builtin_amdgcn_s_waitcnt(0);
          clkl = __builtin_readcyclecounter();
          __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (MODE != 1) {
          const float nz = nzs[g];
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int o = orow + m * 32 + (r & 3) + 8 * (r >> 2);
            float v = acc[m][g][0][r] * dm_m[r];
            if constexpr (ACT) {
              v = __fadd_rn(__fadd_rn(v, nz), bias_m[r]);
              v = (v > 0.f ? v : v * alpha) * ascale;
            }
            if constexpr (BST) {
              if constexpr (RGB) acc[m][g][0][r] = vg ? v : 0.f;
              __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rsrc_o, vofs[g], so_m + (unsigned)((r & 3) + 8 * (r >> 2)) * dps4, 0);
            } else {
              if constexpr (RGB) acc[m][g][0][r] = (vg && o < p.cout) ? v : 0.f;
              if ((!RGB || dstbase) && vg && o < p.cout) dpos[(long long)o * dps] = v;
            }
          }
        } else {
          const int X = 2 * pos_x[g], Y0 = 2 * pos_y[g];
          const bool pair = X + 1 < p.ow;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int o = orow + m * 32 + (r & 3) + 8 * (r >> 2);
            if constexpr (BST) {
#pragma unroll
              for (int py = 0; py < 2; ++py) {
                u32x2 t;
                t.x = __float_as_uint(acc[m][g][py * 2][r] * dm_m[r]);
                t.y = __float_as_uint(acc[m][g][py * 2 + 1][r] * dm_m[r]);
                __builtin_amdgcn_raw_buffer_store_b64(t, rsrc_o, vofs[g], so_m + (unsigned)((r & 3) + 8 * (r >> 2)) * dps4 + (unsigned)py * drs4, 0);
              }
            } else {
              if (!(vg && o < p.cout)) continue;
              float* dst = dpos + (long long)o * dps;
#pragma unroll
              for (int py = 0; py < 2; ++py) {
                if (Y0 + py >= p.oh) continue;
                const float v0 = acc[m][g][py * 2][r] * dm_m[r], v1 = acc[m][g][py * 2 + 1][r] * dm_m[r];
                if (pair) {
                  f32x2_u t; t.x = v0; t.y = v1;
                  *reinterpret_cast<f32x2_u*>(dst + (long long)py * drs) = t;
                } else {
                  dst[(long long)py * drs] = v0;
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
  if (MC_CLOCK(p) && tid == 0 && (blockIdx.x & 63) == 0) {
    const unsigned long long clk3 = __builtin_readcyclecounter();
    atomicAdd(p.dbg_clock, clk3 - clk0);
    atomicAdd(p.dbg_clock + 1, wall_clock64() - rt0);
    atomicAdd(p.dbg_clock + 2, clk1 - clk0);       // set-up (tile decode, staging plan)
    atomicAdd(p.dbg_clock + 3, clk2 - clk1);       // first chunk's wait + K loop
    atomicAdd(p.dbg_clock + 4, clk3 - clk2);       // epilogue up to the last store's issue
    atomicAdd(p.dbg_clock + 5, 1ULL);
    atomicAdd(p.dbg_clock + 8, clkl - clke);       // of the epilogue: waiting for bias / demod / noise
    __builtin_amdgcn_s_waitcnt(0);
    atomicAdd(p.dbg_clock + 9, __builtin_readcyclecounter() - clk3);   // after the last store's issue: until all are acknowledged
    atomicAdd(p.dbg_clock + 7, clke - clk2);       // of the epilogue: destinations, validity, noise loads issued
    atomicAdd(p.dbg_clock + 6, clkw);               // of the K loop: waiting for the chunk's DMA + barrier (lean loop)
  }
  if constexpr (RGB) {
    // second pass over the activated values (now in the accumulators): rgb = sum over rows of act * wmod, where
    // wmod[b,c,o] = rgb_scale * W[c,o] * rgb_style[b,o]; rows of one position live in both lane halves (khalf) and,
    // with WM = 2, in two waves: fixed-order reduction.
    float rs[RNP][3];
#pragma unroll
    for (int g = 0; g < RNP; ++g) rs[g][0] = rs[g][1] = rs[g][2] = 0.f;
    // (the host only fuses layers whose tiles hold one sample: b0 is the sample of every position of this block)
    const float* wmod = p.rgb_wmod + (long long)min(b0, p.batch - 1) * 3 * p.cout;
#pragma unroll
    for (int m = 0; m < RM; ++m) {
      float rw[3][16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int oc = min(orow + m * 32 + (r & 3) + 8 * (r >> 2), p.cout - 1);
#pragma unroll
        for (int c = 0; c < 3; ++c) rw[c][r] = wmod[c * p.cout + oc];
      }
#pragma unroll
      for (int g = 0; g < RNP; ++g)
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
          for (int c = 0; c < 3; ++c) rs[g][c] = fmaf(acc[m][g][0][r], rw[c][r], rs[g][c]);
    }
#pragma unroll
    for (int g = 0; g < RNP; ++g)
#pragma unroll
      for (int c = 0; c < 3; ++c) rs[g][c] += __shfl_xor(rs[g][c], 32, 64);
    if constexpr (WM > 1) {
      static_assert(WM == 2, "two output-channel waves at most");
      __syncthreads();            // every wave is done with the last chunk's LDS operands
      float* red = smem + ((wn * RNP) * 3) * 32;
      if (wm == 1 && khalf == 0)
#pragma unroll
        for (int g = 0; g < RNP; ++g)
#pragma unroll
          for (int c = 0; c < 3; ++c) red[(g * 3 + c) * 32 + l31] = rs[g][c];
      __syncthreads();
      if (wm == 0 && khalf == 0)
#pragma unroll
        for (int g = 0; g < RNP; ++g)
#pragma unroll
          for (int c = 0; c < 3; ++c) rs[g][c] += red[(g * 3 + c) * 32 + l31];
    }
    if (wm == 0 && khalf == 0) {
#pragma unroll
      for (int g = 0; g < RNP; ++g) {
        const int b = pos_b[g], y = pos_y[g], x = pos_x[g];
        if (b >= p.batch || y >= p.oh || x >= p.ow) continue;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          if (c >= p.rgb_c) continue;
          const long long off = (((long long)b * p.rgb_c + c) * p.oh + y) * p.ow + x;
          float v = rs[g][c];
          if (p.rgb_bias) v = __fadd_rn(v, p.rgb_bias[c]);
          if (p.rgb_skip) v = __fadd_rn(v, p.rgb_skip[off]);
          p.rgb_out[off] = v;
        }
      }
    }
  }
}

// split-K finish: out = epilogue(sum_ks ws[ks])
__global__ __launch_bounds__(256) void modconv_splitk_finish_f32(const MCParams p) {
  const float nw = (p.fuse_act && p.noise && p.noise_weight) ? p.noise_weight[0] : 0.f;
  const int hw = p.oh * p.ow;
  const long long total = (long long)p.batch * p.cout * hw;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    float v = 0.f;
    for (int ks = 0; ks < p.ksplit; ++ks) v += p.ws[ks * total + idx];
    const int pix = (int)(idx % hw);
    const long long bo = idx / hw;
    p.out[bo * p.out_plane_stride + (long long)(pix / p.ow) * p.out_row_stride + pix % p.ow] =
        mc_epilogue(v, p, nw, (int)(bo / p.cout), (int)(bo % p.cout), pix);
  }
}

// Tile width.  Thin column segments of small images use 1- or 2-wide tiles (fewer, fuller blocks: 220 vs 284 us on the
// 16^2 -> 32^2 layer); on large images the taller patch would overflow the prefetch slots and fall to the
// synchronous staging path (+15 %), so they keep 4-wide tiles.
inline int pick_tw_log2(int gw, int gh) {
  if (gh <= 32 && gw <= 2) return gw <= 1 ? 0 : 1;
  return gw <= 4 ? 2 : (gw <= 8 ? 3 : (gw <= 16 ? 4 : 5));
}

// experiments build: FMGAN_MC_WIDE=0 keeps the 4-byte patch pieces (A/B measurements); product: always on
inline bool mc_wide_patch() {
#ifdef FMGAN_EXPERIMENTS
  const char* e = getenv("FMGAN_MC_WIDE");
  return !(e && e[0] == '0');
#else
  return true;
#endif
}

// tile plan of one segment; returns its block count (without the o_tiles factor) and LDS floats for the patch
inline long long plan_segment(MCParams::Seg& sg, int batch, int BN) {
  sg.tw_log2 = pick_tw_log2(sg.gw, sg.gh);
  const int TW = 1 << sg.tw_log2;
  const int rows_total = BN / TW;
  int th = rows_total, nb = 1;
  if (sg.gh < rows_total) {
    th = 1;
    while (th < sg.gh) th <<= 1;
    nb = rows_total / th;
  }
  sg.th = th; sg.nb = nb;
  sg.tiles_x = (sg.gw + TW - 1) / TW;
  sg.tiles_y = (sg.gh + th - 1) / th;
  sg.tiles_b = (batch + nb - 1) / nb;
  return (long long)sg.tiles_x * sg.tiles_y * sg.tiles_b;
}

constexpr size_t MC_LDS_LIMIT = 64 * 1024;   // dynamic LDS a launch may ask for without a function attribute
constexpr size_t MC_LDS_MAX = 160 * 1024;    // LDS of a CU

// Returns FMGAN_OK, an error, or +1 when this variant cannot serve the shape (PIPE 1 with an LDS image over the limit):
// the caller then launches the register-pipeline variant.
template <int MODE, int RM, int RNP, int WM, int WN, int MINB = 2, int KC = MC_KC, int PIPE = 0>
int launch_cfg(MCParams& p, hipStream_t s) {
  if constexpr (MODE != 0) { if (p.rgb_out) return FMGAN_EUNSUPPORTED; }
  constexpr int BM = 32 * RM * WM, BN = 32 * RNP * WN;
  constexpr int SP = MODE == 2 ? 2 : 1;
  p.o_tiles = (p.cout + BM - 1) / BM;
  long long blocks = 0;
  size_t patch = 0, nbmax = 1;
  // LDS floats of a segment's halo patch (wide patch: rows of whole, aligned 16-byte groups — see the kernel)
  auto patch_floats = [&](const MCParams::Seg& sg) {
    size_t f = (size_t)sg.nb * KC * (SP * (sg.th - 1) + 3) * (sg.wide ? (1 << sg.tw_log2) + 8 : SP * ((1 << sg.tw_log2) - 1) + 3);
    return sg.wide ? (f + 255) / 256 * 256 : f;             // whole 64-lane x 16-byte pieces
  };
  long long seg_blocks[3] = {0, 0, 0};
  int main_seg = 0;
  for (int i = 0; i < p.nseg; ++i) {
    seg_blocks[i] = plan_segment(p.seg[i], p.batch, BN);
    if (seg_blocks[i] > seg_blocks[main_seg]) main_seg = i;
    p.seg[i].wide = PIPE == 1 && SP == 1 && p.seg[i].tw_log2 == 5 && (p.seg[i].n_off & 3) == 0 && (p.w & 3) == 0 &&
                    (reinterpret_cast<uintptr_t>(p.in) & 15) == 0 && mc_wide_patch();
  }
  // The LDS image is sized by the largest patch of the launch and decides how many blocks a CU holds.  The thin segments of
  // the transposed conv (one row / one column of quads) pack samples or rows into their tiles and used to need twice the
  // main segment's patch — one block per CU less for the whole layer.  They now take smaller tiles (fewer samples, then
  // fewer rows, per tile: part of the tile's positions stay empty) until their patch fits the main segment's.
  // (Only for launches of several rounds of blocks: there the thin segments are < 1 % of the blocks.  A small layer runs in
  // one round, and four times as many mostly-empty strip blocks with full-length K loops cost it 25 %: 8^2, 16^2 at B = 8.)
  const size_t main_patch = patch_floats(p.seg[main_seg]);
  const bool cap_strips = seg_blocks[main_seg] * p.o_tiles >= 4LL * FMGAN_NUM_CU;
  for (int i = 0; i < p.nseg; ++i) {
    MCParams::Seg& sg = p.seg[i];
    while (cap_strips && i != main_seg && patch_floats(sg) > main_patch && (sg.nb > 1 || sg.th > 1)) {
      if (sg.nb > 1) sg.nb >>= 1; else sg.th >>= 1;
      sg.tiles_y = (sg.gh + sg.th - 1) / sg.th;
      sg.tiles_b = (p.batch + sg.nb - 1) / sg.nb;
      seg_blocks[i] = (long long)sg.tiles_x * sg.tiles_y * sg.tiles_b;
    }
    blocks += seg_blocks[i] * p.o_tiles;
    if (blocks > 0x7fffffffLL) return FMGAN_EOVERFLOW;
    sg.block_end = (unsigned)blocks;
    // 32-bit in-tile element offsets: (samples of a tile that exist) x cin x h x w must fit.  (Slots of a tile's
    // samples beyond the batch are never addressed: their offsets are parked / their loads masked.)
    const int nb_live = sg.nb < p.batch ? sg.nb : p.batch;
    if ((long long)nb_live * p.cin * p.h * p.w >= (1LL << 31)) return FMGAN_EOVERFLOW;
    const size_t f = patch_floats(sg);
    if (f > patch) patch = f;
    if ((size_t)sg.nb > nbmax) nbmax = sg.nb;
  }
  size_t lds = sizeof(float) * ((size_t)KC * 9 * BM + patch);
  if constexpr (PIPE == 1) {
    if (p.cin_per_split % KC != 0) return 1;
    p.lds_patch_floats = (int)((patch + 63) / 64 * 64);
    p.lds_buf_floats = (KC * 9 * BM + 255) / 256 * 256 + p.lds_patch_floats + (int)((nbmax * KC + 63) / 64 * 64);
    // + this tile's bias and demodulation values + (tiles of one sample, <= 512 input channels) its style slice, staged once
    p.lds_style_floats = (nbmax == 1 && p.cin_per_split <= 512) ? p.cin_per_split : 0;
    lds = sizeof(float) * (2 * (size_t)p.lds_buf_floats + 2 * BM + p.lds_style_floats);
    if (lds > (MINB == 1 ? MC_LDS_MAX : MC_LDS_LIMIT)) return 1;
    if (lds > MC_LDS_LIMIT) {
      // one block per CU may use more than the 64 KB a launch gets by default (160 KB per CU on gfx950)
      static bool raised = false;
      if (!raised) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&modconv_mfma_f32<MODE, RM, RNP, WM, WN, false, MINB, KC, PIPE>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)MC_LDS_MAX) != hipSuccess) {
          (void)hipGetLastError();
          return 1;
        }
        raised = true;
      }
    }
  }
  if constexpr (MODE == 0) {
    if (p.rgb_out) {
      if (p.o_tiles != 1 || p.ksplit != 1) return FMGAN_EUNSUPPORTED;
      hipLaunchKernelGGL((modconv_mfma_f32<0, RM, RNP, WM, WN, true, MINB, KC, PIPE>), dim3((unsigned)blocks, 1), dim3(256), lds, s, p);
      return fmgan_check_launch();
    }
  }
  hipLaunchKernelGGL((modconv_mfma_f32<MODE, RM, RNP, WM, WN, false, MINB, KC, PIPE>), dim3((unsigned)blocks, p.ksplit), dim3(256), lds, s, p);
  return fmgan_check_launch();
}

// Which pipeline variant serves (mode, cfg): 'A' = register prefetch, 'B'/'C' = LDS-DMA variants.  Defaults are the
// measured winners (profiles/r02_modconv_variants.md).  Only the experiments build (make experiments:
// -DFMGAN_EXPERIMENTS, tools/exp/lib/libfmgan_hip_exp.so) reads FMGAN_MC_V<mode><cfg>=A|B|C, FMGAN_MC_DEBUG and
// FMGAN_MC_CLOCKPTR; the product library has no environment switch and no ablation code path.
inline char mc_variant(int mode, int cfg) {
#ifndef FMGAN_EXPERIMENTS
  static const char defaults[3][3] = {{'C', 'C', 'E'}, {'A', 'B', 'D'}, {'A', 'A', 'A'}};   // comments: see below
  return defaults[mode][cfg];
#else
  // (read per call: the A/B tools sweep variants inside one process)
  // measured on MI355X (profiles/r02_modconv_variants.md): plain conv, Cout >= 96: 128 x 256 tile by LDS-DMA (C);
  // Cout >= 48: 64 x 256 in 4-channel chunks, three blocks per CU (C; 1470 -> 1390 us at 512^2 against 8-channel chunks
  // at two blocks per CU, possible since the scalar wave index freed ~25 VGPRs); Cout < 48: 32 x 256, three blocks per CU
  // (C: 1701 us at 1024^2; 32 x 128 by LDS-DMA 1809, register pipeline 1760-1790, 32 x 512 1873);
  // transposed conv: LDS-DMA (B) for every width.
  // Round 3 (profiles/r03_modconv_block_phases.md, r03_modconv_layers_ab.md), after the wide patch and the buffer-store
  // epilogue: Cout < 48 plain: 32 x 256 in 4-channel chunks (24 KB of LDS): E = in a 128-register budget, 4 blocks per
  // CU instead of 3, no spills (1446 -> 1361 us at 1024^2, with the fused ToRGB 1852 -> 1796-1800); D = in a 96-register
  // budget, 5 blocks per CU, 55-90 spilled dwords: 1 % behind E (1377 / 1806-1826).  Transposed 32-channel tile in a
  // 128-register budget (D: 4 blocks per CU instead of 3, 7 spilled dwords in the epilogue; 785 -> 757 us at 512^2).
  const char defaults[3][3] = {{'C', 'C', 'E'}, {'A', 'B', 'D'}, {'A', 'A', 'A'}};
  char name[32];
  snprintf(name, sizeof(name), "FMGAN_MC_V%d%d", mode, cfg);
  const char* e = getenv(name);
  return (e && e[0] >= 'A' && e[0] <= 'E') ? e[0] : defaults[mode][cfg];
#endif
}

// Tile configurations (output channels x positions per block; blocks per CU the register budget allows):
//   mode 0: 0 = 128 x 128 (2 per CU), 1 = 64 x 128 (3 per CU), 2 = 32 x 128 (4 per CU)
//   mode 2: 0 = 128 x 128, 1 = 64 x 128, 2 = 32 x 128
//   mode 1: 1 = 64 x 128 (2 per CU; 4 phases = 128 accumulator registers), 2 = 32 x 128 (3 per CU)
// For Cout < 96 larger position tiles (64 x 256, 32 x 512; mode 1: 32 x 256) were the first design: fewer weight
// re-reads per MFMA, but 245-256 VGPRs and only two blocks per CU.  On these short-K layers (4-16 chunks per block) more
// co-resident blocks hide the prologue/epilogue better: 1024^2 plain 1681 -> 1597 us, 1024^2 transposed 980 -> 894 us,
// 512^2 plain 1440 -> 1403 us (B=8).  The same trade for Cout >= 96 loses (64 x 128 x 3: 1283 vs 1233 us at 64^2), and
// 32 x 128 tiles for the transposed conv at Cout >= 48 are a wash (+-3 %, sign depends on batch), so those stay.
inline int pick_cfg(int mode, int cout, long long positions) {
  if (mode == 0) {
    if (positions <= 2048) return 2;          // tiny layers: many small tiles (+ split-K)
    return cout >= 96 ? 0 : (cout >= 48 ? 1 : 2);
  }
  if (mode == 2) return cout >= 96 ? 0 : (cout >= 48 ? 1 : 2);   // stride-2 patches are 4x larger: 128 positions only
  // transposed conv: the 32-channel tile (three blocks per CU) also serves the widest layers once they are large
  // enough (Cout 256-512 at 32^2-64^2, B=8: 518 -> 501 and 850 -> 808 us); a tie at Cout 64-128, slower below 4096 positions
  if (cout >= 192 && positions >= 4096) return 2;
  return cout >= 48 ? 1 : 2;
}

inline void cfg_dims(int mode, int cfg, int& BM, int& BN) {
  (void)mode;
  const int bm[3] = {128, 64, 32};
  BM = bm[cfg]; BN = 128;
}

inline int cfg_blocks_per_cu(int mode, int cfg) {
  if (mode == 1) return cfg == 2 ? 3 : 2;
  return cfg == 0 ? 2 : (cfg == 1 ? (mode == 0 ? 3 : 2) : 4);
}

// blocks a (BM x BN) tiling of this launch would have (all segments)
inline long long blocks_with(const MCParams& p, int BM, int BN) {
  long long blocks = 0;
  for (int i = 0; i < p.nseg; ++i) {
    MCParams::Seg sg = p.seg[i];
    blocks += plan_segment(sg, p.batch, BN);
  }
  return blocks * ((p.cout + BM - 1) / BM);
}

inline int launch_any(int mode, int cfg, MCParams& p, hipStream_t s) {
  char v = mc_variant(mode, cfg);
  // the large-tile variants need enough tiles to fill the chip twice over (2 blocks per CU); smaller launches keep
  // the 128-position tiles (and their split-K plan)
  // (mid-size launches — 16^2 .. 32^2 at B=8 — then take the LDS-DMA pipeline on the 128-position tile: 119 vs 126 us
  // and 380 vs 396 us; below ~1500 positions the register pipeline is as fast or faster)
  if (mode == 0 && v == 'C') {
    const bool small = p.ksplit > 1 || blocks_with(p, cfg == 0 ? 128 : (cfg == 1 ? 64 : 32), 256) < 2LL * FMGAN_NUM_CU;
    if (cfg < 2 && (small || p.rgb_out)) v = (!p.rgb_out && (long long)p.batch * p.h * p.w >= 1536) ? 'B' : 'A';
    else if (cfg == 2 && small) v = 'B';
  }
  if (mode == 0 && (v == 'D' || v == 'E')) {
    const bool small = p.ksplit > 1 || blocks_with(p, 32, 256) < 2LL * FMGAN_NUM_CU;
    if (small) v = 'B';
  }
  int st = 1;
  if (mode == 0) {
    switch (cfg) {
      case 0:
        if (v == 'B') st = launch_cfg<0, 2, 2, 2, 2, 3, 4, 1>(p, s);
        else if (v == 'C') st = launch_cfg<0, 4, 2, 1, 4, 2, 4, 1>(p, s);                      // 128 x 256: 0.75 reads / MFMA
        return st != 1 ? st : launch_cfg<0, 2, 2, 2, 2>(p, s);
      case 1:
        if (v == 'B') st = launch_cfg<0, 2, 1, 1, 4, 3, 8, 1>(p, s);
        else if (v == 'C') st = launch_cfg<0, 2, 2, 1, 4, 3, 4, 1>(p, s);                      // 64 x 256, 4-channel chunks: 3 blocks per CU
        return st != 1 ? st : launch_cfg<0, 2, 1, 1, 4, 3>(p, s);
      default:
        if (v == 'B') st = launch_cfg<0, 1, 1, 1, 4, 4, 8, 1>(p, s);
        else if (v == 'C') st = launch_cfg<0, 1, 2, 1, 4, 3, 8, 1>(p, s);                      // 32 x 256, 3 blocks per CU
        else if (v == 'D') st = launch_cfg<0, 1, 2, 1, 4, 5, 4, 1>(p, s);                      // 32 x 256, 4-channel chunks: 5-6 blocks per CU
        else if (v == 'E') st = launch_cfg<0, 1, 2, 1, 4, 4, 4, 1>(p, s);                      // the same in a 128-register budget (4 blocks per CU, no spills)
        return st != 1 ? st : launch_cfg<0, 1, 1, 1, 4, 4>(p, s);
    }
  }
  if (mode == 2) {
    switch (cfg) {
      case 0: return launch_cfg<2, 2, 2, 2, 2>(p, s);
      case 1: return launch_cfg<2, 2, 1, 1, 4>(p, s);
      default: return launch_cfg<2, 1, 1, 1, 4>(p, s);
    }
  }
  if (cfg == 1) {
    if (v == 'B') st = launch_cfg<1, 2, 1, 1, 4, 2, 8, 1>(p, s);
    else if (v == 'C') st = launch_cfg<1, 1, 2, 1, 4, 2, 8, 1>(p, s);                          // 32 x 256 positions
    return st != 1 ? st : launch_cfg<1, 2, 1, 1, 4>(p, s);
  }
  if (v == 'B') st = launch_cfg<1, 1, 1, 1, 4, 3, 8, 1>(p, s);
  else if (v == 'C') st = launch_cfg<1, 1, 2, 1, 4, 2, 8, 1>(p, s);
  else if (v == 'D') st = launch_cfg<1, 1, 1, 1, 4, 4, 8, 1>(p, s);                            // as B in a 128-register budget: 4 blocks per CU
  return st != 1 ? st : launch_cfg<1, 1, 1, 1, 4, 3>(p, s);
}

// Blocks of one launch (all segments), for a given tile configuration.
inline void out_dims(int mode, int h, int w, int& oh, int& ow) {
  if (mode == 1) { oh = 2 * h + 1; ow = 2 * w + 1; }
  else if (mode == 2) { oh = (h - 3) / 2 + 1; ow = (w - 3) / 2 + 1; }
  else { oh = h; ow = w; }
}

inline long long count_blocks(int mode, int cfg, int batch, int cout, int h, int w) {
  int BM, BN;
  cfg_dims(mode, cfg, BM, BN);
  if (mode == 2) out_dims(2, h, w, h, w);   // positions = outputs
  MCParams::Seg sg[3] = {{0, 0, h, w}, {h, 0, 1, w + 1}, {0, w, h, 1}};   // any order: only the sum matters
  long long blocks = 0;
  for (int i = 0; i < (mode == 1 ? 3 : 1); ++i) blocks += plan_segment(sg[i], batch, BN);
  return blocks * ((cout + BM - 1) / BM);
}

// Split-K factor.  A launch runs in rounds of `slots` co-resident blocks; 608 equal blocks on 512 slots take two
// rounds, i.e. 1.7x their ideal time.  Splitting the input-channel loop ks ways makes the blocks ks times shorter
// (plus a fixed prologue/epilogue per block and a finish pass over ks partial slabs).  Pick the ks that minimises
// the modelled time; layers with many rounds keep ks = 1.
inline int pick_ksplit(int mode, int batch, int cin, int cout, int h, int w) {
  int poh, pow_;
  out_dims(mode == 2 ? 2 : 0, h, w, poh, pow_);   // position grid (mode 1: the input grid)
  const int cfg = pick_cfg(mode, cout, (long long)batch * poh * pow_);
  const long long blocks = count_blocks(mode, cfg, batch, cout, h, w);
  const int chunks = (cin + MC_KC - 1) / MC_KC;
  const int slots = FMGAN_NUM_CU * cfg_blocks_per_cu(mode, cfg);   // co-resident blocks (VGPR-limited)
  if (blocks >= 4LL * slots || chunks < 4) return 1;
  int BM, BN;
  cfg_dims(mode, cfg, BM, BN);
  // time unit: one chunk of one block.  Finish pass: (2*ks + 1) * out_bytes at ~4 TB/s against ~5 us per full-size chunk
  const double mfma_per_chunk = (double)(BM / 32) * (BN / 32) * (mode == 1 ? 9 : 9) * (MC_KC / 2) / 4.0;  // per wave
  const double t_chunk_us = mfma_per_chunk * 64.0 / 2360.0 * 2.0;       // two blocks share each SIMD
  int ooh, oow;
  out_dims(mode, h, w, ooh, oow);
  const double out_mb = (double)batch * cout * ooh * (double)oow * 4e-6;
  int best = 1;
  double best_t = 1e30;
  for (int ks = 1; ks <= 16 && ks <= chunks / 2; ++ks) {
    const long long rounds = (blocks * ks + slots - 1) / slots;
    double t = (double)rounds * ((double)((chunks + ks - 1) / ks) + 2.0) * t_chunk_us;
    if (ks > 1) t += (2.0 * ks + 1.0) * out_mb / 4.0 + 3.0;   // MB / (4 MB per us) + launch
    if (t < best_t * 0.97) { best_t = t; best = ks; }
  }
  return best;
}

// ------------------------------------------------------------------ weight gradient of the plain conv (MFMA)
// gw[o,i,ky,kx] = sum_{b,y,x} (d[b,o] * go[b,o,y,x]) * (s[b,i] * x[b,i,y+ky-1,x+kx-1])
// GEMM with M = Cout (A rows), N = Cin (B columns), K = pixels; the 9 taps are 9 accumulators that share the A
// operand and read B at 9 constant offsets of the staged halo patch.  A block owns a 32(o) x 32(i) tile; its four
// waves take the four 32-pixel quarters of each 128-pixel tile (so a chunk is 16 K-steps x 9 MFMAs per wave) and are
// summed through LDS at the end; the pixel range is split over blocks (fixed-order finish, no atomics).
struct WGParams {
  const float* go; const float* d; const float* x; const float* s; float* partial;
  int batch, cin, cout, h, w;
  int tw_log2, th, tiles_x, tiles_y, ntiles, ksplit, tiles_per_split, o_tiles, i_tiles;
};

__global__ __launch_bounds__(256, 2) void modconv_wgrad_f32(const WGParams p) {
  constexpr int SA = 129;                      // Gz row stride (128 pixels + 1: conflict-free across o)
  extern __shared__ float smem[];
  const int TW = 1 << p.tw_log2, PWP = TW + 2, PH = p.th + 2;
  const int SB = PH * PWP + 1 + ((PH * PWP) & 1);   // odd stride: conflict-free across i
  float* Gz = smem;                            // [32][SA]
  float* Us = smem + 32 * SA;                  // [32][SB]
  // wave index as an SGPR: LDS-DMA destinations (M0), piece guards and tile offsets derived from it stay scalar
  // (left in a VGPR, every `buffer_load ... lds` sat in a waterfall loop with v_readfirstlane)
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, khalf = lane >> 5;
  const unsigned lb = xcd_remap(blockIdx.x, gridDim.x);
  const int o_tile = lb % p.o_tiles;
  const int i_tile = (lb / p.o_tiles) % p.i_tiles;
  const int ks = lb / (p.o_tiles * p.i_tiles);
  const int o0 = o_tile * 32, i0 = i_tile * 32;
  const int hw = p.h * p.w;

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  const int t_begin = ks * p.tiles_per_split, t_end = min(p.ntiles, t_begin + p.tiles_per_split);
  for (int tile = t_begin; tile < t_end; ++tile) {
    const int tx = tile % p.tiles_x;
    const int ty = (tile / p.tiles_x) % p.tiles_y;
    const int b = tile / (p.tiles_x * p.tiles_y);
    const int y0 = ty * p.th, x0 = tx * TW;
    __syncthreads();
    // A: 32 output channels x 128 pixels of d*go
    for (int idx = tid; idx < 32 * 128; idx += 256) {
      const int o = idx >> 7, pix = idx & 127;
      const int y = y0 + (pix >> p.tw_log2), x = x0 + (pix & (TW - 1));
      float v = 0.f;
      if (o0 + o < p.cout && y < p.h && x < p.w) {
        const long long ch = (long long)b * p.cout + o0 + o;
        v = p.go[ch * hw + y * p.w + x];
        if (p.d) v *= p.d[ch];
      }
      Gz[o * SA + pix] = v;
    }
    // B: 32 input channels x halo patch of s*x
    const int patch = PH * PWP;
    for (int idx = tid; idx < 32 * patch; idx += 256) {
      const int i = idx / patch, q = idx - i * patch;
      const int y = y0 + q / PWP - 1, x = x0 + q % PWP - 1;
      float v = 0.f;
      if (i0 + i < p.cin && y >= 0 && y < p.h && x >= 0 && x < p.w) {
        const long long ch = (long long)b * p.cin + i0 + i;
        v = p.x[ch * hw + y * p.w + x] * p.s[ch];
      }
      Us[i * SB + q] = v;
    }
    __syncthreads();
    const float* ga = Gz + l31 * SA + wave * 32 + khalf;
    const float* ub = Us + l31 * SB;
    float a_cur, b_cur[9], a_nxt = 0.f, b_nxt[9];
    auto fetch = [&](float& a, float (&bb)[9], int j) {
      const int pix = wave * 32 + 2 * j + khalf;
      const int off = (pix >> p.tw_log2) * PWP + (pix & (TW - 1));
      a = ga[2 * j];
#pragma unroll
      for (int t = 0; t < 9; ++t) bb[t] = ub[off + (t / 3) * PWP + (t % 3)];
    };
    fetch(a_cur, b_cur, 0);
#pragma unroll 4
    for (int j = 0; j < 16; ++j) {
      __builtin_amdgcn_sched_barrier(0);
      if (j + 1 < 16) fetch(a_nxt, b_nxt, j + 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 0; t < 9; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur, b_cur[t], acc[t], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      a_cur = a_nxt;
#pragma unroll
      for (int t = 0; t < 9; ++t) b_cur[t] = b_nxt[t];
    }
  }
  // sum the four waves through LDS, one tap at a time, and write the block's partial slab [ks][t][o][i]
  float* red = smem;   // [4][32][33]
  float* slab = p.partial + (long long)ks * 9 * p.cout * p.cin;
  for (int t = 0; t < 9; ++t) {
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int o = (r & 3) + 8 * (r >> 2) + 4 * khalf;
      red[(wave * 32 + o) * 33 + l31] = acc[t][r];
    }
    __syncthreads();
    for (int e = tid; e < 32 * 32; e += 256) {
      const int o = e >> 5, i = e & 31;
      const float v = red[o * 33 + i] + red[(32 + o) * 33 + i] + red[(64 + o) * 33 + i] + red[(96 + o) * 33 + i];
      if (o0 + o < p.cout && i0 + i < p.cin) slab[((long long)t * p.cout + o0 + o) * p.cin + i0 + i] = v;
    }
  }
}

// gw[o][i][t] = scale * sum_ks partial[ks][t][o][i]
__global__ __launch_bounds__(256) void modconv_wgrad_finish_f32(const float* __restrict__ partial, float* __restrict__ gw,
                                                                int cout, int cin, int ksplit, float scale) {
  const int total = cout * cin * 9;
  for (int idx = blockIdx.x * 256 + threadIdx.x; idx < total; idx += gridDim.x * 256) {
    const int t = idx % 9, oi = idx / 9;
    float v = 0.f;
    for (int ks = 0; ks < ksplit; ++ks) v += partial[((long long)ks * 9 + t) * cout * cin + oi];
    gw[idx] = v * scale;
  }
}


// ---- 64 x 64 weight-gradient tile (round 2).  The 32 x 32 kernel above spends most of its time filling LDS: every
// 144 MFMAs of a wave need a new 128-pixel tile, and a block re-reads gz / u for a quarter of the output a 64 x 64 tile
// covers.  Here a block owns 64 (o) x 64 (i), its four waves the four 32 x 32 quadrants, and ALL of them walk the same
// 2 x TW pixels per step: 2 TW/2... = TW K-steps x 9 MFMAs = 288 MFMAs per wave between two barriers (TW = 32), with the
// pixel pair of an MFMA (its K = 2) taken from the two rows — so consecutive K-steps move one pixel along x and the
// 3 x 3 window of the shifted operand slides: 3 new LDS reads per step instead of 9 (+1 for gz): 4 reads per 9 MFMAs.
struct WG64Params {
  // R[t][a][b] = sum_{n,y,x} (sa[n,a] * A[n,a,y,x]) * (sb[n,b] * B[n,b, SP*y + ky - ORG, SP*x + kx - ORG])
  //   SP = 1, ORG = 1: plain conv        A = go [cout, h, w],          B = x  [cin, h, w]         gw[o=a][i=b]
  //   SP = 2, ORG = 0: transposed conv   A = x  [cin, h, w],           B = go [cout, 2h+1, 2w+1]  gw[o=b][i=a]
  //   SP = 2, ORG = 0: stride-2 conv     A = go [cout, h', w'],        B = x  [cin, h, w]         gw[o=a][i=b]
  const float* A; const float* sa; const float* B; const float* sb; float* partial;
  int batch, ca, cb, ha, wa, hb, wb;
  int tiles_x, tiles_y, ntiles, ksplit, tiles_per_split, a_tiles, b_tiles;
};

template <int TWL2, int SP>
__global__ __launch_bounds__(256, 2) void modconv_wgrad64_f32(const WG64Params p) {
  constexpr int TW = 1 << TWL2, ORG = SP == 1 ? 1 : 0;
  constexpr int PA = 2 * TW + 1;                          // odd pitches: conflict-free over channels
  constexpr int BR = SP + 3, PWP = SP * (TW - 1) + 3;     // rows / columns of the shifted operand's patch
  constexpr int PB = (BR * PWP) | 1;
  // per thread: float4 of A; of B: float4 of the row interiors + the scalars left over
  constexpr int NA4 = 64 * 2 * TW / 4 / 256;
  constexpr int RV = SP == 1 ? TW / 4 : PWP / 4;          // float4 per patch row (SP 1: the aligned interior x0 .. x0+TW-1)
  constexpr int RS = PWP - 4 * RV;                        // scalars per patch row (SP 1: the two halo columns)
  constexpr int NB4 = (64 * BR * RV + 255) / 256, NBS = (64 * BR * RS + 255) / 256;
  extern __shared__ float smem[];
  float* As = smem;               // [64][PA]
  float* Bs = smem + 64 * PA;     // [64][PB]
  // wave index as an SGPR: LDS-DMA destinations (M0), piece guards and tile offsets derived from it stay scalar
  // (left in a VGPR, every `buffer_load ... lds` sat in a waterfall loop with v_readfirstlane)
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, khalf = lane >> 5;
  const int aq = wave >> 1, bq = wave & 1;
  const unsigned lb = xcd_remap(blockIdx.x, gridDim.x);
  const int a_tile = lb % p.a_tiles;
  const int b_tile = (lb / p.a_tiles) % p.b_tiles;
  const int ks = lb / (p.a_tiles * p.b_tiles);
  const int a0 = a_tile * 64, b0 = b_tile * 64;
  const long long hwa = (long long)p.ha * p.wa, hwb = (long long)p.hb * p.wb;
  // A rows start 16-byte aligned and a float4 never straddles the right edge (B is read with dword-aligned vectors)
  const bool vec = (p.wa & 3) == 0 && (((uintptr_t)p.A) & 15) == 0 && (((uintptr_t)p.B) & 3) == 0;

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  // ---- staging plan: everything a tile step needs is loaded into registers while the previous step is on the
  // matrix pipe and written to LDS after the barrier (the forward kernel's register pipeline)
  f32x4 a4[NA4], b4[NB4];
  float bs[NBS], sa_n = 1.f, sb_n = 1.f, sa_lane = 1.f, sb_lane = 1.f;
  const int my_a = a0 + aq * 32 + l31, my_b = b0 + bq * 32 + l31;
  auto decode = [&](int tile, int& n, int& y0, int& x0) {
    const int tx = tile % p.tiles_x;
    const int ty = (tile / p.tiles_x) % p.tiles_y;
    n = tile / (p.tiles_x * p.tiles_y);
    y0 = ty * 2; x0 = tx * TW;
  };
  // patch (row r, column c) of the shifted operand <-> its pixel
  auto issue = [&](int tile) {
    int n, y0, x0;
    decode(tile, n, y0, x0);
    sa_n = (p.sa && my_a < p.ca) ? p.sa[(long long)n * p.ca + my_a] : (my_a < p.ca ? 1.f : 0.f);
    sb_n = (p.sb && my_b < p.cb) ? p.sb[(long long)n * p.cb + my_b] : (my_b < p.cb ? 1.f : 0.f);
    if (!vec) return;
#pragma unroll
    for (int k = 0; k < NA4; ++k) {
      const int q = tid + 256 * k;
      const int a = q / (2 * TW / 4), rem = q % (2 * TW / 4);
      const int y = y0 + rem / (TW / 4), x = x0 + 4 * (rem % (TW / 4));
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (a0 + a < p.ca && y < p.ha && x < p.wa)
        v = *reinterpret_cast<const f32x4*>(p.A + ((long long)n * p.ca + a0 + a) * hwa + (long long)y * p.wa + x);
      a4[k] = v;
    }
    const int by0 = SP * y0 - ORG, bx0 = SP * x0 - ORG;
#pragma unroll
    for (int k = 0; k < NB4; ++k) {
      const int q = tid + 256 * k;
      const int b = q / (BR * RV), rem = q % (BR * RV);
      const int y = by0 + rem / RV, c = (SP == 1 ? 1 : 0) + 4 * (rem % RV), x = bx0 + c;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (q < 64 * BR * RV && b0 + b < p.cb && y >= 0 && y < p.hb) {
        const float* src = p.B + ((long long)n * p.cb + b0 + b) * hwb + (long long)y * p.wb + x;
        if (x >= 0 && x + 3 < p.wb) {
          const f32x4_u t = *reinterpret_cast<const f32x4_u*>(src);
          v.x = t.x; v.y = t.y; v.z = t.z; v.w = t.w;
        } else {
          if (x + 0 >= 0 && x + 0 < p.wb) v.x = src[0];
          if (x + 1 >= 0 && x + 1 < p.wb) v.y = src[1];
          if (x + 2 >= 0 && x + 2 < p.wb) v.z = src[2];
          if (x + 3 >= 0 && x + 3 < p.wb) v.w = src[3];
        }
      }
      b4[k] = v;
    }
#pragma unroll
    for (int k = 0; k < NBS; ++k) {
      const int q = tid + 256 * k;
      const int b = q / (BR * RS), rem = q % (BR * RS);
      const int r = rem / RS, e = rem % RS;
      // SP 1: the two halo columns 0 and TW+1; SP 2: the column(s) after the last float4
      const int c = SP == 1 ? (e ? TW + 1 : 0) : 4 * RV + e;
      const int y = by0 + r, x = bx0 + c;
      bs[k] = (q < 64 * BR * RS && b0 + b < p.cb && y >= 0 && y < p.hb && x >= 0 && x < p.wb)
                  ? p.B[((long long)n * p.cb + b0 + b) * hwb + (long long)y * p.wb + x] : 0.f;
    }
  };
  auto commit = [&](int tile) {
    sa_lane = sa_n; sb_lane = sb_n;
    if (vec) {
#pragma unroll
      for (int k = 0; k < NA4; ++k) {
        const int q = tid + 256 * k;
        const int a = q / (2 * TW / 4), rem = q % (2 * TW / 4);
        float* dst = As + a * PA + 4 * rem;                // (row r, column 4*c4) = r*TW + 4*c4 = 4*rem
        dst[0] = a4[k].x; dst[1] = a4[k].y; dst[2] = a4[k].z; dst[3] = a4[k].w;
      }
#pragma unroll
      for (int k = 0; k < NB4; ++k) {
        const int q = tid + 256 * k;
        if (q < 64 * BR * RV) {
          const int b = q / (BR * RV), rem = q % (BR * RV);
          float* dst = Bs + b * PB + (rem / RV) * PWP + (SP == 1 ? 1 : 0) + 4 * (rem % RV);
          dst[0] = b4[k].x; dst[1] = b4[k].y; dst[2] = b4[k].z; dst[3] = b4[k].w;
        }
      }
#pragma unroll
      for (int k = 0; k < NBS; ++k) {
        const int q = tid + 256 * k;
        if (q < 64 * BR * RS) {
          const int b = q / (BR * RS), rem = q % (BR * RS);
          const int r = rem / RS, e = rem % RS;
          Bs[b * PB + r * PWP + (SP == 1 ? (e ? TW + 1 : 0) : 4 * RV + e)] = bs[k];
        }
      }
      return;
    }
    // widths that are no multiple of 4 / unaligned tensors: guarded scalar fill, not prefetched
    int n, y0, x0;
    decode(tile, n, y0, x0);
    for (int idx = tid; idx < 64 * 2 * TW; idx += 256) {
      const int a = idx / (2 * TW), rc = idx - a * 2 * TW;
      const int y = y0 + (rc >> TWL2), x = x0 + (rc & (TW - 1));
      float v = 0.f;
      if (a0 + a < p.ca && y < p.ha && x < p.wa) v = p.A[((long long)n * p.ca + a0 + a) * hwa + (long long)y * p.wa + x];
      As[a * PA + rc] = v;
    }
    for (int idx = tid; idx < 64 * BR * PWP; idx += 256) {
      const int b = idx / (BR * PWP), q = idx - b * BR * PWP;
      const int y = SP * y0 - ORG + q / PWP, x = SP * x0 - ORG + q % PWP;
      float v = 0.f;
      if (b0 + b < p.cb && y >= 0 && y < p.hb && x >= 0 && x < p.wb)
        v = p.B[((long long)n * p.cb + b0 + b) * hwb + (long long)y * p.wb + x];
      Bs[b * PB + q] = v;
    }
  };

  const float* ga = As + (aq * 32 + l31) * PA + khalf * TW;
  const float* ub = Bs + (bq * 32 + l31) * PB + SP * khalf * PWP;
  const int t_begin = ks * p.tiles_per_split, t_end = min(p.ntiles, t_begin + p.tiles_per_split);
  if (t_begin < t_end) issue(t_begin);
  for (int tile = t_begin; tile < t_end; ++tile) {
    __syncthreads();                       // every wave is done reading the previous step
    commit(tile);
    __syncthreads();
    __builtin_amdgcn_sched_barrier(0);
    if (tile + 1 < t_end) issue(tile + 1);  // in flight during this step's MFMAs
    // K-step j: A pixel (y0 + khalf, x0 + j); tap (ky, kx) reads patch row SP*khalf + ky, column SP*j + kx.
    // The 3-column window slides by SP columns per step: column c lives in slot c % 3.
    float win[3][3];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int c = 0; c < 3; ++c) win[ky][c] = ub[ky * PWP + c] * sb_lane;
    float a_cur = ga[0] * sa_lane, a_nxt = 0.f;
#pragma unroll
    for (int j = 0; j < TW; ++j) {
      __builtin_amdgcn_sched_barrier(0);
      float nw[3][SP];
      if (j + 1 < TW) {                    // operands of step j + 1 while step j is on the matrix pipe
        a_nxt = ga[j + 1];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
          for (int e = 0; e < SP; ++e) nw[ky][e] = ub[ky * PWP + SP * j + 3 + e];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
          acc[ky * 3 + kx] =
              __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur, win[ky][(SP * j + kx) % 3], acc[ky * 3 + kx], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (j + 1 < TW) {
        a_cur = a_nxt * sa_lane;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
          for (int e = 0; e < SP; ++e) win[ky][(SP * j + e) % 3] = nw[ky][e] * sb_lane;   // columns SP*j .. leave, SP*j+3 .. enter
      }
    }
  }
  float* slab = p.partial + (long long)ks * 9 * p.ca * p.cb;
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int a = a0 + aq * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
      if (a < p.ca && my_b < p.cb) slab[((long long)t * p.ca + a) * p.cb + my_b] = acc[t][r];
    }
}

// gw[o][i][t] = scale * sum_ks partial[ks][t][a][b];  transposed: (a, b) = (i, o), else (o, i)
__global__ __launch_bounds__(256) void modconv_wgrad64_finish_f32(const float* __restrict__ partial, float* __restrict__ gw,
                                                                  int cout, int cin, int ksplit, float scale, int transposed) {
  const int total = cout * cin * 9;
  for (int idx = blockIdx.x * 256 + threadIdx.x; idx < total; idx += gridDim.x * 256) {
    const int t = idx % 9, oi = idx / 9;
    const int o = oi / cin, i = oi - o * cin;
    const long long ab = transposed ? (long long)i * cout + o : (long long)o * cin + i;
    float v = 0.f;
    for (int ks = 0; ks < ksplit; ++ks) v += partial[((long long)ks * 9 + t) * cout * cin + ab];
    gw[idx] = v * scale;
  }
}

// mode 0: plain conv; 1: transposed stride-2 conv (x [h,w], go [2h+1,2w+1]); 2: stride-2 valid conv (x [h,w], go [(h-3)/2+1, ..])
inline bool wgrad64_setup(WG64Params& q, const float* go, const float* demod, const float* x, const float* style,
                          int batch, int cin, int cout, int h, int w, int mode) {
  if (cin < 48 || cout < 48) return false;
  q.batch = batch;
  if (mode == 0) {
    q.A = go; q.sa = demod; q.ca = cout; q.ha = h; q.wa = w;
    q.B = x; q.sb = style; q.cb = cin; q.hb = h; q.wb = w;
  } else if (mode == 1) {
    q.A = x; q.sa = style; q.ca = cin; q.ha = h; q.wa = w;
    q.B = go; q.sb = demod; q.cb = cout; q.hb = 2 * h + 1; q.wb = 2 * w + 1;
  } else {
    if (h < 3 || w < 3) return false;
    q.A = go; q.sa = demod; q.ca = cout; q.ha = (h - 3) / 2 + 1; q.wa = (w - 3) / 2 + 1;
    q.B = x; q.sb = style; q.cb = cin; q.hb = h; q.wb = w;
  }
  if (q.wa < 16) return false;
  const int TW = (mode == 0 && q.wa >= 32) ? 32 : 16;
  q.tiles_x = (q.wa + TW - 1) / TW;
  q.tiles_y = (q.ha + 1) / 2;
  q.ntiles = batch * q.tiles_x * q.tiles_y;
  q.a_tiles = (q.ca + 63) / 64;
  q.b_tiles = (q.cb + 63) / 64;
  const int pairs = q.a_tiles * q.b_tiles;
  int ks = (FMGAN_NUM_CU * 4 + pairs - 1) / pairs;      // ~2 rounds of 2 blocks per CU
  const int max_ks = q.ntiles / 4 > 0 ? q.ntiles / 4 : 1; // at least 4 tile steps per block
  if (ks > max_ks) ks = max_ks;
  if (ks < 1) ks = 1;
  q.tiles_per_split = (q.ntiles + ks - 1) / ks;
  q.ksplit = (q.ntiles + q.tiles_per_split - 1) / q.tiles_per_split;
  return true;
}

template <int TWL2, int SP>
inline void wgrad64_launch(const WG64Params& q, long long nblk, hipStream_t s) {
  constexpr int TW = 1 << TWL2;
  constexpr int PB = ((SP + 3) * (SP * (TW - 1) + 3)) | 1;
  const size_t lds = sizeof(float) * 64 * ((2 * TW + 1) + PB);
  hipLaunchKernelGGL((modconv_wgrad64_f32<TWL2, SP>), dim3((unsigned)nblk), dim3(256), lds, s, q);
}

inline void wgrad_plan(WGParams& p) {
  p.tw_log2 = p.w >= 32 ? 5 : 4;
  const int TW = 1 << p.tw_log2;
  p.th = 128 / TW;
  p.tiles_x = (p.w + TW - 1) / TW;
  p.tiles_y = (p.h + p.th - 1) / p.th;
  p.ntiles = p.batch * p.tiles_x * p.tiles_y;
  p.o_tiles = (p.cout + 31) / 32;
  p.i_tiles = (p.cin + 31) / 32;
  const int pairs = p.o_tiles * p.i_tiles;
  int ks = (FMGAN_NUM_CU * 6 + pairs - 1) / pairs;     // ~3 rounds of 2 blocks per CU
  if (ks > p.ntiles) ks = p.ntiles;
  if (ks < 1) ks = 1;
  p.tiles_per_split = (p.ntiles + ks - 1) / ks;
  p.ksplit = (p.ntiles + p.tiles_per_split - 1) / p.tiles_per_split;
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

// Small images (<= 128^2): too few pixels for one-pixel-group-per-thread to fill the chip and the 512-channel loop is a
// long dependent chain.  Here a block owns 64 pixels; its four waves each reduce a quarter of the input channels and
// the partial sums meet in LDS (112-184 us -> ~15 us per layer at B=8).
__global__ __launch_bounds__(256) void torgb_small_f32(const float* __restrict__ in, const float* __restrict__ weight,
                                                       const float* __restrict__ style, const float* __restrict__ bias,
                                                       const float* __restrict__ skip, float* __restrict__ out, int cin,
                                                       int cout, int hw, float scale) {
  extern __shared__ float ws[];            // [cout][cin] modulated weights, then [4][4][64] partial sums
  const int b = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int idx = threadIdx.x; idx < cout * cin; idx += 256)
    ws[idx] = scale * weight[idx] * style[(long long)b * cin + idx % cin];
  __syncthreads();
  const int pix = blockIdx.x * 64 + lane;
  const int per = (cin + 3) / 4, i_lo = wave * per, i_hi = min(cin, i_lo + per);
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  if (pix < hw) {
    const float* ip = in + ((long long)b * cin + i_lo) * hw + pix;
#pragma unroll 8
    for (int i = i_lo; i < i_hi; ++i, ip += hw) {
      const float v = *ip;
#pragma unroll
      for (int c = 0; c < 4; ++c)
        if (c < cout) acc[c] = fmaf(ws[c * cin + i], v, acc[c]);
    }
  }
  __syncthreads();                          // everyone is done with ws
  float* red = ws;                          // host guarantees room for 4*4*64 floats
#pragma unroll
  for (int c = 0; c < 4; ++c) red[(wave * 4 + c) * 64 + lane] = acc[c];
  __syncthreads();
  if (wave == 0 && pix < hw) {
    for (int c = 0; c < cout; ++c) {
      // same association as the single-pass kernel would give per quarter; quarters added in channel order
      float r = red[(0 * 4 + c) * 64 + lane] + red[(1 * 4 + c) * 64 + lane];
      r += red[(2 * 4 + c) * 64 + lane];
      r += red[(3 * 4 + c) * 64 + lane];
      const long long off = ((long long)b * cout + c) * hw + pix;
      r = __fadd_rn(r, bias ? bias[c] : 0.f);
      if (skip) r = __fadd_rn(r, skip[off]);
      out[off] = r;
    }
  }
}

}  // namespace

extern "C" int fmgan_modconv_demod_f32(const float* weight, const float* style, float* demod, int batch, int cout,
                                       int cin, int ktaps, float scale, float eps, void* stream) {
  if (batch < 0 || cout <= 0 || cin <= 0 || ktaps <= 0) return FMGAN_EINVAL;
  if (batch == 0) return FMGAN_OK;
  if (!weight || !style || !demod) return FMGAN_EINVAL;
  hipLaunchKernelGGL(modconv_demod_f32<false>, dim3((cout + 3) / 4), dim3(256), 0, (hipStream_t)stream, weight, style,
                     demod, batch, cout, cin, ktaps, scale, eps);
  return fmgan_check_launch();
}

extern "C" int fmgan_modconv_wsq_f32(const float* weight, float* wsq, int cout, int cin, int ktaps, void* stream) {
  if (cout <= 0 || cin <= 0 || ktaps <= 0) return FMGAN_EINVAL;
  if (!weight || !wsq) return FMGAN_EINVAL;
  const long long n = (long long)cout * cin;
  long long blocks = (n + 255) / 256;
  if (blocks > FMGAN_NUM_CU * 16) blocks = FMGAN_NUM_CU * 16;
  hipLaunchKernelGGL(modconv_wsq_f32, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, weight, wsq, n, ktaps);
  return fmgan_check_launch();
}

extern "C" int fmgan_modconv_demod_wsq_f32(const float* wsq, const float* style, float* demod, int batch, int cout,
                                           int cin, float scale, float eps, void* stream) {
  if (batch < 0 || cout <= 0 || cin <= 0) return FMGAN_EINVAL;
  if (batch == 0) return FMGAN_OK;
  if (!wsq || !style || !demod) return FMGAN_EINVAL;
  hipLaunchKernelGGL(modconv_demod_f32<true>, dim3((cout + 3) / 4, batch < 64 ? batch : 64), dim3(256), 0, (hipStream_t)stream, wsq, style, demod,
                     batch, cout, cin, 1, scale, eps);
  return fmgan_check_launch();
}

extern "C" int fmgan_modconv_weight_prep_f32(const float* weight, float* wt, int cout, int cin, int ktaps, float scale,
                                             int kind, void* stream) {
  if (cout <= 0 || cin <= 0 || ktaps <= 0) return FMGAN_EINVAL;
  if (kind < 0 || kind > 2) return FMGAN_EUNSUPPORTED;
  if (!weight || !wt) return FMGAN_EINVAL;
  const long long total = (long long)cout * cin * ktaps;
  long long blocks = (total + 255) / 256;
  if (blocks > FMGAN_NUM_CU * 16) blocks = FMGAN_NUM_CU * 16;
  hipLaunchKernelGGL(modconv_weight_prep_f32, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, weight, wt,
                     cout, cin, ktaps, scale, kind);
  return fmgan_check_launch();
}

extern "C" long long fmgan_modconv2d_workspace_bytes(int batch, int cin, int cout, int h, int w, int mode) {
  if (batch <= 0 || cin <= 0 || cout <= 0 || h <= 0 || w <= 0 || mode < 0 || mode > 2) return 0;
  if (mode == 2 && (h < 3 || w < 3)) return 0;
  const int ks = pick_ksplit(mode, batch, cin, cout, h, w);
  if (ks <= 1) return 0;
  int oh, ow;
  out_dims(mode, h, w, oh, ow);
  return (long long)ks * batch * cout * oh * ow * (long long)sizeof(float);
}

namespace {
struct RgbArgs {
  const float* wmod; const float* bias; const float* skip; float* out; int c;
};

bool rgb_fusable(int batch, int cin, int cout, int h, int w) {
  const int cfg = pick_cfg(0, cout, (long long)batch * h * w);
  int BM, BN;
  cfg_dims(0, cfg, BM, BN);
  (void)cin;
  MCParams::Seg sg{0, 0, h, w};
  plan_segment(sg, batch, BN);
  // one output-channel tile, tiles of a single sample; the fused launch runs without split-K
  return cout <= BM && sg.nb == 1;
}

__global__ __launch_bounds__(256) void torgb_weight_mod_f32(const float* __restrict__ W, const float* __restrict__ style,
                                                            float* __restrict__ wmod, int batch, int cout, int rgb_c,
                                                            float scale) {
  const int n = batch * 3 * cout;
  for (int idx = blockIdx.x * 256 + threadIdx.x; idx < n; idx += gridDim.x * 256) {
    const int o = idx % cout, c = (idx / cout) % 3, b = idx / (3 * cout);
    wmod[idx] = c < rgb_c ? scale * W[c * cout + o] * style[(long long)b * cout + o] : 0.f;
  }
}

int modconv2d_impl(const float* in, const float* wt, const float* style, const float* demod,
                   float* out, int batch, int cin, int cout, int h, int w, int mode,
                   const float* noise, const float* noise_weight, const float* bias, int noise_batch,
                   int fuse_act, float alpha, float act_scale, long long out_plane_stride,
                   int out_row_stride, void* workspace, long long workspace_bytes, void* stream, const RgbArgs* rgb) {
  if (batch < 0 || cin <= 0 || cout <= 0 || h <= 0 || w <= 0) return FMGAN_EINVAL;
  if (mode < 0 || mode > 2) return FMGAN_EUNSUPPORTED;
  if (mode != 0 && fuse_act) return FMGAN_EUNSUPPORTED;  // the blur sits between conv and activation
  if (mode == 2 && (h < 3 || w < 3)) return FMGAN_EINVAL;
  if (rgb) {
    if (rgb->c < 1 || rgb->c > 3) return FMGAN_EUNSUPPORTED;
    if (!rgb_fusable(batch, cin, cout, h, w)) return FMGAN_EUNSUPPORTED;
  }
  if (batch == 0) return FMGAN_OK;
  if (!in || !wt || !style || (!out && !rgb)) return FMGAN_EINVAL;
  if (rgb && (!rgb->wmod || !rgb->out)) return FMGAN_EINVAL;
  if (fuse_act && noise && noise_batch != 1 && noise_batch != batch) return FMGAN_EINVAL;
  MCParams p{};
#ifdef FMGAN_EXPERIMENTS
  {
    static int dbg = -1;
    if (dbg < 0) { const char* e = getenv("FMGAN_MC_DEBUG"); dbg = e ? atoi(e) : 0; }
    p.debug = dbg;
    static long long clk = -1;
    if (clk < 0) { const char* e = getenv("FMGAN_MC_CLOCKPTR"); clk = e ? atoll(e) : 0; }
    p.dbg_clock = (unsigned long long*)clk;
    { const char* e = getenv("FMGAN_MC_FLAGS"); p.exp_flags = e ? atoi(e) : 0; }
  }
#endif
  if (rgb) {
    p.rgb_wmod = rgb->wmod; p.rgb_bias = rgb->bias; p.rgb_skip = rgb->skip; p.rgb_out = rgb->out; p.rgb_c = rgb->c;
  }
  p.in = in; p.wt = wt; p.style = style; p.demod = demod; p.out = out;
  p.batch = batch; p.cin = cin; p.cout = cout; p.h = h; p.w = w;
  out_dims(mode, h, w, p.oh, p.ow);
  if (out_row_stride == 0) out_row_stride = p.ow;
  if (out_plane_stride == 0) out_plane_stride = (long long)p.oh * out_row_stride;
  if (out_row_stride < p.ow || out_plane_stride < (long long)p.oh * out_row_stride) return FMGAN_EINVAL;
  p.out_plane_stride = out_plane_stride; p.out_row_stride = out_row_stride;
  p.noise = noise; p.noise_weight = noise_weight; p.bias = bias;
  p.noise_batch = noise_batch; p.fuse_act = fuse_act; p.alpha = alpha; p.act_scale = act_scale;
  // index ranges the kernel relies on: pixel indices (y*ow + x, h*w) and per-chunk weight offsets are 32-bit ints,
  // whole-tensor offsets are 64-bit; per-tile input offsets (nb*cin*h*w) are checked per segment in launch_cfg
  if ((long long)batch * cout * p.oh * p.ow > (1LL << 40)) return FMGAN_EOVERFLOW;
  if ((long long)p.oh * p.ow >= (1LL << 31) || (long long)h * w >= (1LL << 31)) return FMGAN_EOVERFLOW;
  if (cin > (1 << 20) || cout > (1 << 20)) return FMGAN_EOVERFLOW;
  hipStream_t s = (hipStream_t)stream;
  const int cfg = pick_cfg(mode, cout, (long long)batch * (mode == 2 ? p.oh * p.ow : h * w));
  // split-K only when the caller supplied the workspace fmgan_modconv2d_workspace_bytes() asks for
  p.ksplit = pick_ksplit(mode, batch, cin, cout, h, w);
  const long long need = (long long)p.ksplit * batch * cout * p.oh * p.ow * (long long)sizeof(float);
  if (p.ksplit > 1 && (!workspace || workspace_bytes < need)) p.ksplit = 1;
  if (rgb) p.ksplit = 1;   // the RGB reduction needs the finished sums in one block
  p.ws = (float*)workspace;
  const int chunks = (cin + MC_KC - 1) / MC_KC;
  p.cin_per_split = ((chunks + p.ksplit - 1) / p.ksplit) * MC_KC;
  // Main segment: the h x w grid.  Mode 1: quads (2m+py, 2n+px) of m < h, n < w cover Y < 2h, X < 2w; the last
  // output row (m = h) and column (n = w) are two thin segments of the same launch, so they run beside the
  // main tiles instead of serialising their own latency-bound K loops.
  // The thin segments come FIRST in block order: their blocks pack few useful outputs and must not be the tail.
  if (mode == 1) {
    p.seg[0] = {h, 0, 1, w + 1};
    p.seg[1] = {0, w, h, 1};
    p.seg[2] = {0, 0, h, w};
    p.nseg = 3;
  } else {
    p.seg[0] = {0, 0, p.oh, p.ow};      // mode 0: oh = h; mode 2: the output grid
    p.nseg = 1;
  }
  int st = launch_any(mode, cfg, p, s);
  if (st != FMGAN_OK) return st;
  if (p.ksplit > 1) {
    const long long total = (long long)batch * cout * p.oh * p.ow;
    long long blocks = (total + 255) / 256;
    if (blocks > FMGAN_NUM_CU * 16) blocks = FMGAN_NUM_CU * 16;
    hipLaunchKernelGGL(modconv_splitk_finish_f32, dim3((unsigned)blocks), dim3(256), 0, s, p);
    st = fmgan_check_launch();
  }
  return st;
}
}  // namespace

extern "C" int fmgan_modconv2d_f32(const float* in, const float* wt, const float* style, const float* demod,
                                   float* out, int batch, int cin, int cout, int h, int w, int mode,
                                   const float* noise, const float* noise_weight, const float* bias, int noise_batch,
                                   int fuse_act, float alpha, float act_scale, long long out_plane_stride,
                                   int out_row_stride, void* workspace, long long workspace_bytes, void* stream) {
  return modconv2d_impl(in, wt, style, demod, out, batch, cin, cout, h, w, mode, noise, noise_weight, bias, noise_batch,
                        fuse_act, alpha, act_scale, out_plane_stride, out_row_stride, workspace, workspace_bytes, stream,
                        nullptr);
}

extern "C" int fmgan_torgb_weight_mod_f32(const float* weight, const float* style, float* wmod, int batch, int cout,
                                          int rgb_channels, float scale, void* stream) {
  if (batch < 0 || cout <= 0 || rgb_channels < 1 || rgb_channels > 3) return FMGAN_EINVAL;
  if (batch == 0) return FMGAN_OK;
  if (!weight || !style || !wmod) return FMGAN_EINVAL;
  const int n = batch * 3 * cout;
  hipLaunchKernelGGL(torgb_weight_mod_f32, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, weight, style, wmod,
                     batch, cout, rgb_channels, scale);
  return fmgan_check_launch();
}

extern "C" int fmgan_modconv2d_rgb_fusable(int batch, int cin, int cout, int h, int w) {
  if (batch <= 0 || cin <= 0 || cout <= 0 || h <= 0 || w <= 0) return 0;
  return rgb_fusable(batch, cin, cout, h, w) ? 1 : 0;
}

extern "C" int fmgan_modconv2d_rgb_f32(const float* in, const float* wt, const float* style, const float* demod,
                                       float* out, int batch, int cin, int cout, int h, int w,
                                       const float* noise, const float* noise_weight, const float* bias,
                                       int noise_batch, int fuse_act, float alpha, float act_scale,
                                       const float* rgb_wmod, const float* rgb_bias,
                                       const float* rgb_skip, float* rgb_out, int rgb_channels, void* stream) {
  const RgbArgs rgb{rgb_wmod, rgb_bias, rgb_skip, rgb_out, rgb_channels};
  return modconv2d_impl(in, wt, style, demod, out, batch, cin, cout, h, w, 0, noise, noise_weight, bias, noise_batch,
                        fuse_act, alpha, act_scale, 0, 0, nullptr, 0, stream, &rgb);
}

extern "C" long long fmgan_modconv_wgrad_mode_workspace_bytes(int batch, int cin, int cout, int h, int w, int mode) {
  if (batch <= 0 || cin <= 0 || cout <= 0 || h <= 0 || w <= 0 || mode < 0 || mode > 2) return 0;
  WG64Params q{};
  if (wgrad64_setup(q, nullptr, nullptr, nullptr, nullptr, batch, cin, cout, h, w, mode))
    return (long long)q.ksplit * 9 * cout * cin * (long long)sizeof(float);
  if (mode != 0 || w < 16) return 0;
  WGParams p{};
  p.batch = batch; p.cin = cin; p.cout = cout; p.h = h; p.w = w;
  wgrad_plan(p);
  return (long long)p.ksplit * 9 * cout * cin * (long long)sizeof(float);
}

extern "C" long long fmgan_modconv_wgrad_workspace_bytes(int batch, int cin, int cout, int h, int w) {
  return fmgan_modconv_wgrad_mode_workspace_bytes(batch, cin, cout, h, w, 0);
}

extern "C" int fmgan_modconv_wgrad_mode_f32(const float* go, const float* demod, const float* x, const float* style,
                                            float* gw, int batch, int cin, int cout, int h, int w, int mode, float scale,
                                            void* workspace, long long workspace_bytes, void* stream) {
  if (batch <= 0 || cin <= 0 || cout <= 0 || h <= 0 || w <= 0) return FMGAN_EINVAL;
  if (mode < 0 || mode > 2) return FMGAN_EUNSUPPORTED;
  if (!go || !x || !style || !gw || !workspace) return FMGAN_EINVAL;
  // go is the largest tensor of the transposed conv: [batch, cout, 2h+1, 2w+1]
  const long long big_hw = mode == 1 ? (long long)(2 * h + 1) * (2 * w + 1) : (long long)h * w;
  if ((long long)batch * (cin > cout ? cin : cout) * big_hw >= (1LL << 40)) return FMGAN_EOVERFLOW;
  if (big_hw >= (1LL << 31) || cin > (1 << 20) || cout > (1 << 20)) return FMGAN_EOVERFLOW;
  hipStream_t s = (hipStream_t)stream;
  WG64Params q{};
  if (wgrad64_setup(q, go, demod, x, style, batch, cin, cout, h, w, mode)) {
    q.partial = (float*)workspace;
    if (workspace_bytes < (long long)q.ksplit * 9 * cout * cin * (long long)sizeof(float)) return FMGAN_EINVAL;
    const long long nblk = (long long)q.a_tiles * q.b_tiles * q.ksplit;
    if (nblk > 0x7fffffffLL) return FMGAN_EOVERFLOW;
    if (mode == 0) {
      if (q.wa >= 32) wgrad64_launch<5, 1>(q, nblk, s); else wgrad64_launch<4, 1>(q, nblk, s);
    } else {
      wgrad64_launch<4, 2>(q, nblk, s);
    }
    int st = fmgan_check_launch();
    if (st != FMGAN_OK) return st;
    int fb = (cout * cin * 9 + 255) / 256;
    if (fb > FMGAN_NUM_CU * 16) fb = FMGAN_NUM_CU * 16;
    hipLaunchKernelGGL(modconv_wgrad64_finish_f32, dim3(fb), dim3(256), 0, s, (const float*)workspace, gw, cout, cin,
                       q.ksplit, scale, mode == 1 ? 1 : 0);
    return fmgan_check_launch();
  }
  if (mode != 0 || w < 16) return FMGAN_EUNSUPPORTED;   // narrow / tiny layers: the host keeps MIOpen's wgrad
  WGParams p{};
  p.go = go; p.d = demod; p.x = x; p.s = style; p.partial = (float*)workspace;
  p.batch = batch; p.cin = cin; p.cout = cout; p.h = h; p.w = w;
  wgrad_plan(p);
  if (workspace_bytes < (long long)p.ksplit * 9 * cout * cin * (long long)sizeof(float)) return FMGAN_EINVAL;
  const int TW = 1 << p.tw_log2;
  const int patch = (p.th + 2) * (TW + 2);
  const int SB = patch + 1 + (patch & 1);
  size_t lds = sizeof(float) * (32 * 129 + 32 * (size_t)SB);
  if (lds < sizeof(float) * 4 * 32 * 33) lds = sizeof(float) * 4 * 32 * 33;
  const long long blocks = (long long)p.o_tiles * p.i_tiles * p.ksplit;
  if (blocks > 0x7fffffffLL) return FMGAN_EOVERFLOW;
  hipLaunchKernelGGL(modconv_wgrad_f32, dim3((unsigned)blocks), dim3(256), lds, s, p);
  int st = fmgan_check_launch();
  if (st != FMGAN_OK) return st;
  int fb = (cout * cin * 9 + 255) / 256;
  if (fb > FMGAN_NUM_CU * 16) fb = FMGAN_NUM_CU * 16;
  hipLaunchKernelGGL(modconv_wgrad_finish_f32, dim3(fb), dim3(256), 0, s, (const float*)workspace, gw, cout, cin,
                     p.ksplit, scale);
  return fmgan_check_launch();
}

extern "C" int fmgan_modconv_wgrad_f32(const float* go, const float* demod, const float* x, const float* style,
                                       float* gw, int batch, int cin, int cout, int h, int w, float scale,
                                       void* workspace, long long workspace_bytes, void* stream) {
  if (w > 0 && w < 16) return FMGAN_EUNSUPPORTED;           // tiny layers: negligible FLOPs, the host keeps MIOpen's wgrad
  return fmgan_modconv_wgrad_mode_f32(go, demod, x, style, gw, batch, cin, cout, h, w, 0, scale, workspace,
                                      workspace_bytes, stream);
}

extern "C" int fmgan_torgb_f32(const float* in, const float* weight, const float* style, const float* bias,
                               const float* skip, float* out, int batch, int cin, int cout, int hw, float scale,
                               void* stream) {
  if (batch < 0 || cin <= 0 || cout <= 0 || cout > 4 || hw <= 0) return FMGAN_EINVAL;
  if (batch == 0) return FMGAN_OK;
  if (!in || !weight || !style || !out) return FMGAN_EINVAL;
  if (batch > 65535) return FMGAN_EOVERFLOW;
  size_t lds = sizeof(float) * (size_t)cout * cin;
  if (lds > 64 * 1024) return FMGAN_EUNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  if (hw <= 128 * 128) {
    if (lds < sizeof(float) * 4 * 4 * 64) lds = sizeof(float) * 4 * 4 * 64;
    hipLaunchKernelGGL(torgb_small_f32, dim3((hw + 63) / 64, batch), dim3(256), lds, s, in, weight, style, bias, skip, out,
                       cin, cout, hw, scale);
    return fmgan_check_launch();
  }
  const bool vec = (hw & 3) == 0 && ((((uintptr_t)in) | ((uintptr_t)out) | ((uintptr_t)skip)) & 15) == 0;
  const int n = vec ? hw / 4 : hw;
  int gx = (n + 255) / 256;
  const int cap = (FMGAN_NUM_CU * 16 + batch - 1) / batch;
  if (gx > cap) gx = cap;
  if (gx < 1) gx = 1;
  if (vec) hipLaunchKernelGGL(torgb_f32<4>, dim3(gx, batch), dim3(256), lds, s, in, weight, style, bias, skip, out, cin, cout, hw, scale);
  else hipLaunchKernelGGL(torgb_f32<1>, dim3(gx, batch), dim3(256), lds, s, in, weight, style, bias, skip, out, cin, cout, hw, scale);
  return fmgan_check_launch();
}

// ------------------------------------------------------------------ ToRGB backward (HBM-bound, one pass over x)
// out[b,c,p] = sum_i (scale * W[c,i] * s[b,i]) * x[b,i,p]  (stylegan2.py:389-404 without demodulation).  Its backward
// needs  gx[b,i,p] = sum_c (scale * W[c,i] * s[b,i]) * go[b,c,p]  and  M[b,c,i] = sum_p go[b,c,p] * x[b,i,p]  (from which
// gW[c,i] = scale * sum_b s[b,i] M[b,c,i] and gs[b,i] = scale * sum_c W[c,i] M[b,c,i] are [B,3,Cin] algebra).  The autograd
// composite reads x twice and go many times through a grouped 1x1 convolution and its non-reproducible weight
// gradient; here a block owns 16 input channels and a pixel range of one sample, reads its x planes ONCE, writes gx, and
// keeps the 3 x 16 partial sums of M in registers over the whole range — one block-wide reduction at the end, one
// partial row per (pixel split) that the caller sums in a fixed order (bit-reproducible).
constexpr int TB_CT = 16;
__global__ __launch_bounds__(256) void torgb_bwd_f32(const float* __restrict__ x, const float* __restrict__ go,
                                                     const float* __restrict__ weight, const float* __restrict__ style,
                                                     float* __restrict__ gx, float* __restrict__ mpart, int batch, int cin,
                                                     int cout, int hw4, int chunk, float scale) {
  __shared__ float wm[4][TB_CT];
  __shared__ float red[4][4 * TB_CT];
  const int b = blockIdx.z, i0 = blockIdx.y * TB_CT, split = blockIdx.x;
  const int tid = threadIdx.x;
  if (tid < 4 * TB_CT) {
    const int c = tid / TB_CT, j = tid % TB_CT;
    wm[c][j] = (c < cout && i0 + j < cin) ? scale * weight[c * cin + i0 + j] * style[(long long)b * cin + i0 + j] : 0.f;
  }
  __syncthreads();
  float acc[4][TB_CT];
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int j = 0; j < TB_CT; ++j) acc[c][j] = 0.f;
  const f32x4* go4 = reinterpret_cast<const f32x4*>(go) + (long long)b * cout * hw4;
  const f32x4* x4 = reinterpret_cast<const f32x4*>(x) + ((long long)b * cin + i0) * hw4;
  f32x4* gx4 = reinterpret_cast<f32x4*>(gx) + ((long long)b * cin + i0) * hw4;
  const int p_end = min(hw4, (split + 1) * chunk);
  const int nj = min(TB_CT, cin - i0);
  for (int p = split * chunk + tid; p < p_end; p += 256) {
    f32x4 g[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const f32x4 z = {0.f, 0.f, 0.f, 0.f};
      g[c] = c < cout ? go4[(long long)c * hw4 + p] : z;
    }
#pragma unroll
    for (int j = 0; j < TB_CT; ++j) {
      if (j >= nj) break;
      const f32x4 xv = x4[(long long)j * hw4 + p];
      f32x4 r = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float w = wm[c][j];
        r.x = fmaf(w, g[c].x, r.x); r.y = fmaf(w, g[c].y, r.y); r.z = fmaf(w, g[c].z, r.z); r.w = fmaf(w, g[c].w, r.w);
        acc[c][j] = fmaf(g[c].x, xv.x, fmaf(g[c].y, xv.y, fmaf(g[c].z, xv.z, fmaf(g[c].w, xv.w, acc[c][j]))));
      }
      gx4[(long long)j * hw4 + p] = r;
    }
  }
  // block reduction of the 4 x 16 partials: wave butterfly, then the four waves in order
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int j = 0; j < TB_CT; ++j) {
      float v = acc[c][j];
#pragma unroll
      for (int m = 32; m > 0; m >>= 1) v += __shfl_xor(v, m, 64);
      if ((tid & 63) == 0) red[tid >> 6][c * TB_CT + j] = v;
    }
  __syncthreads();
  if (tid < 4 * TB_CT) {
    const int c = tid / TB_CT, j = tid % TB_CT;
    if (c < cout && i0 + j < cin)
      mpart[(((long long)split * batch + b) * cout + c) * cin + i0 + j] = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
  }
}

// pixel splits of the backward launch (= rows of the partial array [splits, batch, cout, cin]); 0: shape not served
extern "C" int fmgan_torgb_backward_splits(int batch, int cin, int hw) {
  if (batch <= 0 || cin <= 0 || hw <= 0 || (hw & 3)) return 0;
  const int hw4 = hw >> 2;
  const long long others = (long long)batch * ((cin + TB_CT - 1) / TB_CT);
  long long s = (8LL * FMGAN_NUM_CU + others - 1) / others;      // ~8 blocks per CU in the grid
  const long long smax = (hw4 + 511) / 512;                       // at least two float4 per lane and split
  if (s > smax) s = smax;
  if (s < 1) s = 1;
  if (s > 1024) s = 1024;
  return (int)s;
}

extern "C" int fmgan_torgb_backward_f32(const float* x, const float* grad_out, const float* weight, const float* style,
                                        float* grad_x, float* m_partial, int batch, int cin, int cout, int hw, float scale,
                                        void* stream) {
  if (batch < 0 || cin <= 0 || cout <= 0 || cout > 4 || hw <= 0) return FMGAN_EINVAL;
  if (batch == 0) return FMGAN_OK;
  if (!x || !grad_out || !weight || !style || !grad_x || !m_partial) return FMGAN_EINVAL;
  const int splits = fmgan_torgb_backward_splits(batch, cin, hw);
  if (splits == 0 || ((((uintptr_t)x) | ((uintptr_t)grad_out) | ((uintptr_t)grad_x)) & 15) != 0) return FMGAN_EUNSUPPORTED;
  if (batch > 65535) return FMGAN_EOVERFLOW;
  const int hw4 = hw >> 2, chunk = (hw4 + splits - 1) / splits;
  hipLaunchKernelGGL(torgb_bwd_f32, dim3(splits, (cin + TB_CT - 1) / TB_CT, batch), dim3(256), 0, (hipStream_t)stream, x,
                     grad_out, weight, style, grad_x, m_partial, batch, cin, cout, hw4, chunk, scale);
  return fmgan_check_launch();
}
