// upfirdn2d for gfx950 (MI355X): upsample - FIR - downsample in one pass.
//
// Replaces upfirdn2d_op / upfirdn2d_kernel<> / upfirdn2d_kernel_large of the
// reference (op/upfirdn2d_kernel.cu:49-105, 107-207, 209-369).  Same maths:
//   out[n,oy,ox,m] = sum_{ky,kx} kflip[ky,kx] * U[n, oy*down_y + ky - pad_y0, ox*down_x + kx - pad_x0, m]
//   U = input zero-stuffed by (up_y, up_x);  kflip[ky,kx] = kernel[kh-1-ky, kw-1-kx]
// accumulated in tap order (ky outer, kx inner, ascending), as the reference's tile kernel does.
//
// Design (not a translation of the CUDA tiling):
//  * This op is pure HBM streaming (<= 4 flop/B).  The hot configuration is the 4x4 blur after
//    every transposed conv, [B*C, 2H+1, 2W+1] -> [B*C, 2H, 2W], rows of 2W+1 floats that are never
//    16-byte aligned.  Path 1 ("row-march") gives each 64-lane wave a 64*VEC-column strip and
//    marches it down TH output rows: one dword-aligned global_load_dwordx4 per lane per input row
//    (1 KiB contiguous per wave-instruction), the 3 halo columns come from the neighbouring lane
//    by wave shuffle, the 4-row vertical window lives in registers (rolling, statically unrolled
//    by 4, two row-loads in flight ahead of the row being filtered) and the 16 taps sit in SGPRs.
//    Every input row is read once per row-tile ((TH+3)/TH amplification, halo rows are L2/MALL hits
//    thanks to the XCD-contiguous block order) and every output row is written once with dwordx4
//    stores (non-temporal when the output cannot stay in the 256 MiB Infinity Cache).
//    Edge lanes do NOT branch: a segment that straddles the row's ends is still one vector load
//    (the bytes belong to the neighbouring row of the same tensor) and row-invariant masks zero the
//    out-of-range columns.  (A per-element guarded path for those lanes cost 15 %: 527 -> 451 us.)
//    fmgan_upfirdn2d_strided lets the producer hand over rows padded to a 128-byte multiple with the
//    first tap column on a 16-byte boundary (op/_native.py::aligned_rows_buffer): 415 us = 5.18 TB/s.
//    Path 1b (aligned-row inputs, i.e. every large blur of the synthesis network) replaces the register staging by an
//    LDS-DMA ring: 3 rows in flight per wave at 8 waves/SIMD, one hand-counted s_waitcnt per step (see ufd_dmaring_f32).
//  * Path 2 ("plane-tile") stages whole small planes (<= ~110x110) in LDS with one flat coalesced
//    copy: at 4..64 px the planes are too narrow for a wave-wide strip.
//  * Path 3 (up=2 polyphase, the ToRGB skip upsample): 2x4 outputs per thread, only the 2x2 taps
//    that meet a non-zero sample of the zero-stuffed input are visited; store-bound.
//  * Path 0 is the generic fallback (any up/down/pad/kernel/minor, f32/f64/f16).
#include <stdlib.h>
#include "common.h"

namespace {

struct UfdParams {
  int major, in_h, in_w, minor, kh, kw;
  int up_x, up_y, down_x, down_y, pad_x0, pad_y0;
  int out_h, out_w;
  long long in_plane_stride;   // elements between planes of the input (== in_h*in_w*minor when contiguous)
  int in_row_stride;           // elements between rows of the input   (== in_w*minor when contiguous)
};

// ---------------------------------------------------------------- path 0: generic
template <typename T>
__global__ __launch_bounds__(256) void ufd_generic(const T* __restrict__ in, const T* __restrict__ kern,
                                                   T* __restrict__ out, const UfdParams p, const long long total) {
  using Acc = typename AccT<T>::type;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += stride) {
    const int mi = (int)(idx % p.minor);
    long long t = idx / p.minor;
    const int ox = (int)(t % p.out_w);
    t /= p.out_w;
    const int oy = (int)(t % p.out_h);
    const long long mj = t / p.out_h;
    const int by = oy * p.down_y - p.pad_y0;
    const int bx = ox * p.down_x - p.pad_x0;
    const T* pin = in + mj * p.in_plane_stride + mi;
    Acc v = 0;
    for (int ky = 0; ky < p.kh; ++ky) {
      const int uy = by + ky;
      if (uy < 0 || (uy % p.up_y) != 0) continue;
      const int iy = uy / p.up_y;
      if (iy >= p.in_h) continue;
      for (int kx = 0; kx < p.kw; ++kx) {
        const int ux = bx + kx;
        if (ux < 0 || (ux % p.up_x) != 0) continue;
        const int ix = ux / p.up_x;
        if (ix >= p.in_w) continue;
        v += to_acc<T>(pin[(long long)iy * p.in_row_stride + (long long)ix * p.minor]) *
             to_acc<T>(kern[(p.kh - 1 - ky) * p.kw + (p.kw - 1 - kx)]);
      }
    }
    out[idx] = from_acc<T>(v);
  }
}

// ---------------------------------------------------------------- path 1: row-march (f32, up=down=1, k<=4x4, minor=1)
struct RMParams {
  int planes, in_h, in_w, out_h, out_w, pad_x0, pad_y0, kh, kw;
  long long in_plane_stride; int in_row_stride;
  // optional StyledConv epilogue fused into the store (stylegan2.py:371-373): lrelu((y + nw*noise) + bias[c]) * scale
  const float* noise; const float* noise_weight; const float* bias;
  int fuse, channels, noise_batch; float alpha, act_scale;
  int th;       // output rows per wave, multiple of 4
  int strips;   // 64*VEC-column strips per row
  int tiles_y;  // row tiles per plane
  long long total_waves;
};

template <int VEC> struct Row { float v[VEC + 3]; };

// One segment of VEC floats at column `a` of row `rp`.  The load is a single (dword-aligned) vector load whenever
// its bytes lie inside the tensor's memory [lo, hi) — columns outside [0, in_w) then hold a neighbouring row's data
// and are zeroed later by the row-invariant masks of finish_row.  Only the first/last few floats of the whole
// tensor need the per-element path, so edge lanes do not diverge in steady state.
// NT: streaming (non-temporal) load — on tensors larger than the Infinity Cache the input is read once, and keeping
// it out of L2 leaves room for what IS re-read (the fused epilogue's noise plane, shared by all channels of a
// sample): FETCH_SIZE of the fused 1024^2 blur 723 -> 630 MiB (as reported), 500 -> 493 us.
template <int VEC, bool NT = false>
__device__ __forceinline__ void load_seg(float (&dst)[VEC], const float* __restrict__ rp, bool rowok, int a, int in_w,
                                         const float* lo, const float* hi) {
  const float* q = rp + a;
  if (!rowok || a >= in_w || a + VEC <= 0) {
#pragma unroll
    for (int e = 0; e < VEC; ++e) dst[e] = 0.f;
  } else if (q >= lo && q + VEC <= hi) {
    if constexpr (VEC == 4) {
      const f32x4_u t = NT ? __builtin_nontemporal_load(reinterpret_cast<const f32x4_u*>(q))
                           : *reinterpret_cast<const f32x4_u*>(q);
      dst[0] = t.x; dst[1] = t.y; dst[2] = t.z; dst[3] = t.w;
    } else if constexpr (VEC == 2) {
      const f32x2_u t = *reinterpret_cast<const f32x2_u*>(q);
      dst[0] = t.x; dst[1] = t.y;
    } else {
      dst[0] = q[0];
    }
  } else {
#pragma unroll
    for (int e = 0; e < VEC; ++e) dst[e] = (a + e >= 0 && a + e < in_w) ? q[e] : 0.f;
  }
}

// Row `iy` of the plane, columns a0 .. a0+VEC+2 for this lane, in two halves so the HBM latency can be
// covered: issue_row() only issues the global loads (lane l loads segment l = VEC floats; the segments past
// lane 63 are loaded by lanes 0..NX-1 as an extra segment); finish_row() — called two rows later — zeroes the
// out-of-range columns and rotates the 3 halo columns in from the next lanes by wave shuffle.
template <int VEC> struct RawRow { float prim[VEC], extra[VEC]; };

template <int VEC, bool NT = false>
__device__ __forceinline__ void issue_row(RawRow<VEC>& raw, const float* __restrict__ pin, int iy, bool need,
                                          int in_h, int in_w, int in_rs, int a0, int lane, const float* lo,
                                          const float* hi) {
  constexpr int NX = (3 + VEC - 1) / VEC;
  const bool rowok = need && iy >= 0 && iy < in_h;  // wave-uniform
  const float* rp = pin + (long long)iy * in_rs;
  load_seg<VEC, NT>(raw.prim, rp, rowok, a0, in_w, lo, hi);
  if (lane < NX) {
    load_seg<VEC, NT>(raw.extra, rp, rowok, a0 + 64 * VEC, in_w, lo, hi);
  } else {
#pragma unroll
    for (int e = 0; e < VEC; ++e) raw.extra[e] = 0.f;
  }
}

template <int VEC>
__device__ __forceinline__ void finish_row(Row<VEC>& r, const RawRow<VEC>& raw, int lane, unsigned pmask,
                                           unsigned xmask) {
  float prim[VEC], extra[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) {
    prim[e] = (pmask >> e) & 1 ? raw.prim[e] : 0.f;
    extra[e] = (xmask >> e) & 1 ? raw.extra[e] : 0.f;
  }
#pragma unroll
  for (int e = 0; e < VEC; ++e) r.v[e] = prim[e];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int e = VEC + j;
    const int d = e / VEC;   // lane distance of the segment holding column a0+e
    const int el = e % VEC;  // element inside that segment
    const float t = (lane < d) ? extra[el] : prim[el];
    r.v[e] = __shfl(t, (lane + d) & 63, 64);
  }
}

struct RowEpilogue {   // wave-uniform
  const float* noise_plane;   // noise of this plane's sample (nullptr: none)
  float nw, bv, alpha, scale;
  bool on;
};

template <int VEC, bool NT = false>
__device__ __forceinline__ void emit_row(float* __restrict__ pout, int oy, int oy_end, int out_w, int c0,
                                         const float (&kf)[4][4], const Row<VEC>& r0, const Row<VEC>& r1,
                                         const Row<VEC>& r2, const Row<VEC>& r3, const RowEpilogue& ep) {
  if (oy >= oy_end) return;  // wave-uniform
  float acc[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) {
    float v = 0.f;
#pragma unroll
    for (int kx = 0; kx < 4; ++kx) v = fmaf(r0.v[e + kx], kf[0][kx], v);
#pragma unroll
    for (int kx = 0; kx < 4; ++kx) v = fmaf(r1.v[e + kx], kf[1][kx], v);
#pragma unroll
    for (int kx = 0; kx < 4; ++kx) v = fmaf(r2.v[e + kx], kf[2][kx], v);
#pragma unroll
    for (int kx = 0; kx < 4; ++kx) v = fmaf(r3.v[e + kx], kf[3][kx], v);
    acc[e] = v;
  }
  if (ep.on) {
    // same roundings as fmgan_noise_bias_act_f32: (y + nw*n) + b, select, *alpha, *scale — no FMA contraction
    float nz[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) nz[e] = 0.f;
    if (ep.noise_plane) {   // wave-uniform; one vector load per lane (rows of the output are dword-aligned at least)
      const float* np = ep.noise_plane + (long long)oy * out_w + c0;
      if (c0 + VEC <= out_w) {
        if constexpr (VEC == 4) {
          const f32x4_u t = *reinterpret_cast<const f32x4_u*>(np);
          nz[0] = t.x; nz[1] = t.y; nz[2] = t.z; nz[3] = t.w;
        } else if constexpr (VEC == 2) {
          const f32x2_u t = *reinterpret_cast<const f32x2_u*>(np);
          nz[0] = t.x; nz[1] = t.y;
        } else {
          nz[0] = np[0];
        }
      } else {
#pragma unroll
        for (int e = 0; e < VEC; ++e)
          if (c0 + e < out_w) nz[e] = np[e];
      }
    }
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      const float n = nz[e];
      float v = __fadd_rn(__fadd_rn(acc[e], __fmul_rn(ep.nw, n)), ep.bv);
      acc[e] = __fmul_rn(v > 0.f ? v : __fmul_rn(v, ep.alpha), ep.scale);
    }
  }
  float* op = pout + (long long)oy * out_w + c0;
  if (c0 + VEC <= out_w) {
    if constexpr (VEC == 4) {
      f32x4_u t; t.x = acc[0]; t.y = acc[1]; t.z = acc[2]; t.w = acc[3];
      if constexpr (NT) __builtin_nontemporal_store(t, reinterpret_cast<f32x4_u*>(op));
      else *reinterpret_cast<f32x4_u*>(op) = t;
    } else if constexpr (VEC == 2) {
      f32x2_u t; t.x = acc[0]; t.y = acc[1];
      *reinterpret_cast<f32x2_u*>(op) = t;
    } else {
      op[0] = acc[0];
    }
  } else {
#pragma unroll
    for (int e = 0; e < VEC; ++e)
      if (c0 + e < out_w) op[e] = acc[e];
  }
}

template <int VEC, bool NT = false>
__global__ __launch_bounds__(256) void ufd_rowmarch_f32(const float* __restrict__ in, const float* __restrict__ kern,
                                                        float* __restrict__ out, const RMParams p) {
  const int lane = threadIdx.x & 63;
  const unsigned lb = xcd_remap(blockIdx.x, gridDim.x);
  const long long gw = (long long)lb * 4 + (threadIdx.x >> 6);
  if (gw >= p.total_waves) return;  // wave-uniform
  const int strip = (int)(gw % p.strips);
  const long long t = gw / p.strips;
  const int ty = (int)(t % p.tiles_y);
  const long long plane = t / p.tiles_y;

  // flipped taps, zero-extended to 4x4 (wave-uniform -> SGPRs)
  float kf[4][4];
#pragma unroll
  for (int ky = 0; ky < 4; ++ky)
#pragma unroll
    for (int kx = 0; kx < 4; ++kx)
      kf[ky][kx] = (ky < p.kh && kx < p.kw) ? kern[(p.kh - 1 - ky) * p.kw + (p.kw - 1 - kx)] : 0.f;

  const int c0 = (strip * 64 + lane) * VEC;
  const int a0 = c0 - p.pad_x0;
  const int oy0 = ty * p.th;
  const int oy_end = min(oy0 + p.th, p.out_h);
  const float* pin = in + plane * p.in_plane_stride;
  float* pout = out + plane * (long long)p.out_h * p.out_w;
  const int iy0 = oy0 - p.pad_y0;

  // Rolling 4-row register window (statically unrolled by 4) fed by a 2-deep ring of in-flight row loads:
  // while row r is filtered, rows r+1 and r+2 are on their way from HBM.
  Row<VEC> w0, w1, w2, w3;
  RawRow<VEC> ra, rb;
  // column-validity masks of this lane's primary / extra segment: the same for every row of the march
  unsigned pmask = 0, xmask = 0;
#pragma unroll
  for (int e = 0; e < VEC; ++e) {
    pmask |= (a0 + e >= 0 && a0 + e < p.in_w) ? 1u << e : 0u;
    xmask |= (a0 + 64 * VEC + e >= 0 && a0 + 64 * VEC + e < p.in_w) ? 1u << e : 0u;
  }
  RowEpilogue ep{};
  ep.on = p.fuse != 0;
  if (ep.on) {
    const int ch = (int)(plane % p.channels), smp = (int)(plane / p.channels);
    ep.nw = (p.noise && p.noise_weight) ? p.noise_weight[0] : 0.f;
    ep.bv = p.bias ? p.bias[ch] : 0.f;
    ep.alpha = p.alpha; ep.scale = p.act_scale;
    ep.noise_plane = p.noise ? p.noise + (long long)(p.noise_batch == 1 ? 0 : smp) * p.out_h * p.out_w : nullptr;
  }
  const float* lo = in;
  const float* hi = in + ((long long)(p.planes - 1) * p.in_plane_stride + (long long)(p.in_h - 1) * p.in_row_stride + p.in_w);
#define ISSUE(raw, k, need) issue_row<VEC, NT>(raw, pin, iy0 + (k), need, p.in_h, p.in_w, p.in_row_stride, a0, lane, lo, hi)
#define FINISH(w, raw) finish_row<VEC>(w, raw, lane, pmask, xmask)
  ISSUE(ra, 0, true); ISSUE(rb, 1, true);
  FINISH(w0, ra); ISSUE(ra, 2, true);
  FINISH(w1, rb); ISSUE(rb, 3, true);
  FINISH(w2, ra); ISSUE(ra, 4, oy0 + 1 < oy_end);
  // invariant at loop top (r): w0..w2 = rows r..r+2; rb = row r+3 in flight, ra = row r+4 in flight
  for (int r = 0; r < p.th; r += 4) {
    const int oy = oy0 + r;
    if (oy >= oy_end) break;  // wave-uniform
    FINISH(w3, rb); ISSUE(rb, r + 5, oy + 2 < oy_end);
    emit_row<VEC, NT>(pout, oy + 0, oy_end, p.out_w, c0, kf, w0, w1, w2, w3, ep);
    FINISH(w0, ra); ISSUE(ra, r + 6, oy + 3 < oy_end);
    emit_row<VEC, NT>(pout, oy + 1, oy_end, p.out_w, c0, kf, w1, w2, w3, w0, ep);
    FINISH(w1, rb); ISSUE(rb, r + 7, oy + 4 < oy_end && r + 4 < p.th);
    emit_row<VEC, NT>(pout, oy + 2, oy_end, p.out_w, c0, kf, w2, w3, w0, w1, ep);
    FINISH(w2, ra); ISSUE(ra, r + 8, oy + 5 < oy_end && r + 4 < p.th);
    emit_row<VEC, NT>(pout, oy + 3, oy_end, p.out_w, c0, kf, w3, w0, w1, w2, ep);
  }
#undef ISSUE
#undef FINISH
}


// ---------------------------------------------------------------- path 1b: row-march through an LDS-DMA ring
// Same blocking as path 1 (a wave marches TH rows of a 256-column strip), for inputs in the aligned-row layout
// (op/_native.py::aligned_rows_buffer: the first tap column of every row on a 16-byte boundary, pitch % 4 == 0):
//  * a row segment goes HBM -> LDS with `buffer_load_dwordx4 ... lds` (+ one dword DMA for the 3 columns past lane 63):
//    no staging VGPRs, no address VALU (the lane offsets are row-invariant, the row is the SGPR soffset), rows outside
//    the image are zero-filled by the buffer range check.  4 slots per wave = 3 rows in flight ahead of the row being
//    filtered, at 46-52 VGPRs (8 waves/SIMD; path 1 holds 2 rows in flight at 114 VGPRs, 4 waves/SIMD);
//  * the wave synchronises with ITSELF only: `s_waitcnt vmcnt(N)`, N = the number of VMEM operations issued after the
//    wanted row's DMAs (returns are in order), a compile-time constant of the step because every step issues the same
//    operations (noise request, 2 DMAs, store) — partial tiles, whose steps skip stores, wait for vmcnt(0) instead;
//  * a lane's 7-column window is two ds_read_b128: the halo columns are simply the next lane's first floats;
//  * the noise row of the fused epilogue is requested 3 steps ahead, BEFORE the DMAs of the row that completes the same
//    output row, so the one wait per step covers it.  The request is inline asm on purpose: the compiler's waitcnt
//    model merges control-flow paths conservatively and puts a near-zero vmcnt before every use of a loaded VGPR,
//    which drains the ring (measured: no gain over path 1 with a compiler-visible load).
// Taps are applied in the same order as path 1 (bit-identical results).  Measured on the headline blur
// [256,1025,1025] -> [256,1024,1024] (tools/exp/blur_probe.hip, five boxes): fused 495-509 -> 404-434 us on the same box,
// plain 408-432 -> 392-420 us; a linear copy of the same bytes takes 337-364 us on those boxes.
struct DRParams {
  const float* in0;   // position 0 of row 0 of plane 0 (= logical column -pad_x0), 16-byte aligned
  float* out; const float* kern; const float* noise; const float* noise_weight; const float* bias;
  int planes, channels, noise_batch, in_h, in_w, out_h, out_w, rs, pad_x0, pad_y0, kh, kw;
  long long ps;
  int th, strips, tiles_y; long long total_waves;
  float alpha, act_scale;
};

template <int SIZE, int AUX, typename RSRC>   // AUX: cache policy bits of the load (gfx940+: 2 = nt, streaming)
__device__ __forceinline__ void ufd_dma_to_lds(RSRC rsrc, float* lds, unsigned voff, unsigned soff) {
#if defined(__HIP_DEVICE_COMPILE__)   // (a kernel template naming the builtin directly loses its host stub)
  typedef __attribute__((address_space(3))) void* lds_void_ptr;
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void_ptr)lds, SIZE, voff, soff, 0, AUX);
#endif
}

template <int N> __device__ __forceinline__ void ufd_wait_vm() {
#if defined(__HIP_DEVICE_COMPILE__)
  // gfx9 encoding: vmcnt = imm[3:0] | imm[15:14] << 4; expcnt imm[6:4] and lgkmcnt imm[11:8] left at "no wait"
  __builtin_amdgcn_s_waitcnt((N & 0xF) | ((N >> 4) << 14) | (0x7 << 4) | (0xF << 8));
#endif
}

// VMEM operations issued after stage(k)'s two DMAs and before step k's wait.  Step s issues, in this order:
// [noise request (EPI)] [2 DMAs of stage(s+3)] [wait] ... [store of output row s-3, s >= 3]; stages 0..2 come first.
__host__ __device__ constexpr int ufd_ring_wait(int k, bool epi) {
  int n = 0;
  if (k >= 3) { if (k - 3 >= 3) ++n; }          // the store of the step that issued stage(k)
  else n += 2 * (2 - k);                        // prologue stages k+1..2
  for (int s = (k - 2 > 0 ? k - 2 : 0); s <= k; ++s) { n += 2 + (epi ? 1 : 0); if (s < k && s >= 3) ++n; }
  return n;
}

template <bool EPI, bool NT, bool REMAP>
__global__ __launch_bounds__(256) void ufd_dmaring_f32(const DRParams p) {
  extern __shared__ __attribute__((aligned(16))) float ufd_ring[];
  constexpr int SLOT = 320;   // floats per slot: 256 (64 lanes x 4) + 64 (the halo dwords; lanes >= 4 write zeros there)
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned lb = REMAP ? xcd_remap(blockIdx.x, gridDim.x) : blockIdx.x;
  const long long gw = (long long)lb * 4 + wv;
  if (gw >= p.total_waves) return;  // wave-uniform
  const int strip = (int)(gw % p.strips);
  const long long t = gw / p.strips;
  const int ty = (int)(t % p.tiles_y);
  const long long plane = t / p.tiles_y;
  float* ring = ufd_ring + wv * (4 * SLOT);

  // flipped taps, zero-extended to 4x4 (wave-uniform -> SGPRs)
  float kf[4][4];
#pragma unroll
  for (int ky = 0; ky < 4; ++ky)
#pragma unroll
    for (int kx = 0; kx < 4; ++kx)
      kf[ky][kx] = (ky < p.kh && kx < p.kw) ? p.kern[(p.kh - 1 - ky) * p.kw + (p.kw - 1 - kx)] : 0.f;

  const int c0 = (strip * 64 + lane) * 4;          // first output column = first window position of this lane
  const unsigned PARK = 0xFFFFFFF0u;               // out of every buffer's range: the DMA writes zeros
  const unsigned vmain = (c0 + 4 <= p.rs) ? (unsigned)c0 * 4u : PARK;
  const int xe = (strip + 1) * 256 + lane;
  const unsigned vext = (lane < 4 && xe < p.rs) ? (unsigned)xe * 4u : PARK;
  unsigned mask = 0;                               // which of the 7 window positions are columns of the image
#pragma unroll
  for (int i = 0; i < 7; ++i) {
    const int col = c0 + i - p.pad_x0;
    mask |= (col >= 0 && col < p.in_w) ? 1u << i : 0u;
  }
  const bool lane_on = c0 < p.out_w;               // out_w % 4 == 0 (host)
  const int oy0 = ty * p.th, oy_end = min(oy0 + p.th, p.out_h);
  const bool full = oy0 + p.th <= p.out_h;         // wave-uniform: the static counts assume every step >= 3 stores
  const int iy0 = oy0 - p.pad_y0;
  const int nsteps = p.th + 3;
  float* pin = const_cast<float*>(p.in0 + plane * p.ps);
  float* pout = p.out + plane * (long long)p.out_h * p.out_w;

  float nw = 0.f, bv = 0.f;
  const float* nplane = nullptr;
  if (EPI) {
    const int ch = (int)(plane % p.channels), smp = (int)(plane / p.channels);
    nw = p.noise_weight ? p.noise_weight[0] : 0.f;
    bv = p.bias ? p.bias[ch] : 0.f;
    nplane = p.noise + (p.noise_batch == 1 ? 0LL : (long long)smp * p.out_h * p.out_w);
  }
  const unsigned nvoff = lane_on ? (unsigned)c0 * 4u : 0u;
  f32x4 nz[4];   // noise rows in flight (row k requested at step k, used at step k+3)
  auto noise_req = [&](f32x4& dst, int row) {
    const float* rowp = nplane + (long long)row * p.out_w;   // wave-uniform
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(nvoff), "s"(rowp) : "memory");
#else
    (void)rowp; (void)nvoff; (void)dst;
#endif
  };
  auto noise_ack = [&](f32x4& v) {   // after the wait that covers the request: from here on the value may be used
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(v) : : "memory");
#endif
  };
  auto issue = [&](int k, int slot) {
    const int iy = iy0 + k;
    const bool ok = k < nsteps && iy >= 0 && iy < p.in_h;   // wave-uniform; else: num_records 0 -> zeros
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(pin, 0, ok ? p.in_h * p.rs * 4 : 0, 0x00020000);
    const unsigned soff = ok ? (unsigned)iy * (unsigned)p.rs * 4u : 0u;
    ufd_dma_to_lds<16, NT ? 2 : 0>(rsrc, ring + slot * SLOT, vmain, soff);
    ufd_dma_to_lds<4, NT ? 2 : 0>(rsrc, ring + slot * SLOT + 256, vext, soff);
  };
  float W[4][7];   // the 4-row window of this lane's 7 columns
  auto take = [&](int slot, int wi) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(ring + slot * SLOT + lane * 4);
    const f32x4 b = *reinterpret_cast<const f32x4*>(ring + slot * SLOT + lane * 4 + 4);
    const float x[7] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z};
#pragma unroll
    for (int i = 0; i < 7; ++i) W[wi][i] = (mask >> i) & 1 ? x[i] : 0.f;
  };
  auto emit = [&](int oy, int w0, int w1, int w2, int w3, const f32x4& nzv) {
    if (oy >= oy_end) return;  // wave-uniform (partial tiles only)
    float acc[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float v = 0.f;
#pragma unroll
      for (int kx = 0; kx < 4; ++kx) v = fmaf(W[w0][e + kx], kf[0][kx], v);
#pragma unroll
      for (int kx = 0; kx < 4; ++kx) v = fmaf(W[w1][e + kx], kf[1][kx], v);
#pragma unroll
      for (int kx = 0; kx < 4; ++kx) v = fmaf(W[w2][e + kx], kf[2][kx], v);
#pragma unroll
      for (int kx = 0; kx < 4; ++kx) v = fmaf(W[w3][e + kx], kf[3][kx], v);
      acc[e] = v;
    }
    if constexpr (EPI) {
      // same roundings as fmgan_noise_bias_act_f32: (y + nw*n) + b, select, *alpha, *scale — no FMA contraction
      const float n[4] = {nzv.x, nzv.y, nzv.z, nzv.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float v = __fadd_rn(__fadd_rn(acc[e], __fmul_rn(nw, n[e])), bv);
        acc[e] = __fmul_rn(v > 0.f ? v : __fmul_rn(v, p.alpha), p.act_scale);
      }
    }
    const f32x4 o = {acc[0], acc[1], acc[2], acc[3]};
    if (lane_on) __builtin_nontemporal_store(o, reinterpret_cast<f32x4*>(pout + (long long)oy * p.out_w + c0));
  };

#define UFD_STEP(K, J, NW)                                                                            \
  {                                                                                                   \
    if (EPI) noise_req(nz[(J) & 3], min(oy0 + (K), p.out_h - 1));                                     \
    issue((K) + 3, ((J) + 3) & 3);                                                                    \
    __builtin_amdgcn_sched_barrier(0);                                                                \
    if (full) ufd_wait_vm<NW>(); else ufd_wait_vm<0>();                                               \
    __builtin_amdgcn_sched_barrier(0);                                                                \
    if (EPI && (K) >= 3) noise_ack(nz[((J) + 1) & 3]);                                                \
    take((J) & 3, (J) & 3);                                                                           \
    if ((K) >= 3) emit(oy0 + (K) - 3, ((J) + 1) & 3, ((J) + 2) & 3, ((J) + 3) & 3, (J) & 3, nz[((J) + 1) & 3]); \
  }
  issue(0, 0); issue(1, 1); issue(2, 2);
  UFD_STEP(0, 0, ufd_ring_wait(0, EPI)) UFD_STEP(1, 1, ufd_ring_wait(1, EPI))
  UFD_STEP(2, 2, ufd_ring_wait(2, EPI)) UFD_STEP(3, 3, ufd_ring_wait(3, EPI))
  UFD_STEP(4, 0, ufd_ring_wait(4, EPI)) UFD_STEP(5, 1, ufd_ring_wait(5, EPI))
  UFD_STEP(6, 2, ufd_ring_wait(6, EPI)) UFD_STEP(7, 3, ufd_ring_wait(7, EPI))
  constexpr int NS = ufd_ring_wait(8, EPI);
  for (int kb = 8; kb < nsteps; kb += 4) {
    UFD_STEP(kb, 0, NS)
    if (kb + 1 >= nsteps) break;
    UFD_STEP(kb + 1, 1, NS)
    if (kb + 2 >= nsteps) break;
    UFD_STEP(kb + 2, 2, NS)
    if (kb + 3 >= nsteps) break;
    UFD_STEP(kb + 3, 3, NS)
  }
#undef UFD_STEP
  ufd_wait_vm<0>();   // nothing of this wave's ring (or its noise requests) is in flight when it ends
}

// ---------------------------------------------------------------- path 2: plane-tile (f32, up=down=1, k<=4x4, small planes)
// Planes of 4..65 px are too narrow for a wave-wide strip.  A block stages PB whole planes (contiguous in memory,
// so the copy is one flat coalesced stream) in LDS and each thread filters consecutive outputs from there.
struct PTParams {
  int planes, in_h, in_w, out_h, out_w, pad_x0, pad_y0, kh, kw, pb;
  long long in_plane_stride; int in_row_stride;      // elements; contiguous planes: in_h*in_w and in_w
  // optional StyledConv epilogue in the store (EPI): lrelu((y + nw*noise) + bias[c]) * scale, the roundings of
  // fmgan_noise_bias_act_f32 — the 4^2..32^2 upsampling layers then need one launch instead of blur + epilogue
  const float* noise; const float* noise_weight; const float* bias;
  int channels, noise_batch; float alpha, act_scale;
};

template <bool EPI>
__global__ __launch_bounds__(256) void ufd_planetile_f32(const float* __restrict__ in, const float* __restrict__ kern,
                                                         float* __restrict__ out, const PTParams p) {
  extern __shared__ float tile[];
  float kf[4][4];
#pragma unroll
  for (int ky = 0; ky < 4; ++ky)
#pragma unroll
    for (int kx = 0; kx < 4; ++kx)
      kf[ky][kx] = (ky < p.kh && kx < p.kw) ? kern[(p.kh - 1 - ky) * p.kw + (p.kw - 1 - kx)] : 0.f;
  const int plane0 = blockIdx.x * p.pb;
  const int np = min(p.pb, p.planes - plane0);
  const int in_sz = p.in_h * p.in_w, out_sz = p.out_h * p.out_w;
  const float* src = in + (long long)plane0 * p.in_plane_stride;
  if (p.in_row_stride == p.in_w && p.in_plane_stride == in_sz) {
    for (int i = threadIdx.x; i < np * in_sz; i += 256) tile[i] = src[i];
  } else {                                   // aligned-row layout of the transposed conv's private intermediate
    for (int i = threadIdx.x; i < np * in_sz; i += 256) {
      const int pl = i / in_sz, r = i - pl * in_sz;
      const int y = r / p.in_w, x = r - y * p.in_w;
      tile[i] = src[(long long)pl * p.in_plane_stride + y * p.in_row_stride + x];
    }
  }
  __syncthreads();
  float* dst = out + (long long)plane0 * out_sz;
  const float nw = (EPI && p.noise && p.noise_weight) ? p.noise_weight[0] : 0.f;
  for (int o = threadIdx.x; o < np * out_sz; o += 256) {
    const int pl = o / out_sz, r = o - pl * out_sz;
    const int oy = r / p.out_w, ox = r - oy * p.out_w;
    const float* tp = tile + pl * in_sz;
    float v = 0.f;
#pragma unroll
    for (int ky = 0; ky < 4; ++ky) {
      const int iy = oy + ky - p.pad_y0;
      const bool yok = iy >= 0 && iy < p.in_h;
#pragma unroll
      for (int kx = 0; kx < 4; ++kx) {
        const int ix = ox + kx - p.pad_x0;
        const float x = (yok && ix >= 0 && ix < p.in_w) ? tp[iy * p.in_w + ix] : 0.f;
        v = fmaf(x, kf[ky][kx], v);
      }
    }
    if constexpr (EPI) {
      const int plane = plane0 + pl;
      const int c = plane % p.channels, b = plane / p.channels;
      const float n = p.noise ? p.noise[(long long)(p.noise_batch == 1 ? 0 : b) * out_sz + r] : 0.f;
      const float bv = p.bias ? p.bias[c] : 0.f;
      const float t = __fadd_rn(__fadd_rn(v, __fmul_rn(nw, n)), bv);
      v = __fmul_rn(t > 0.f ? t : __fmul_rn(t, p.alpha), p.act_scale);
    }
    dst[o] = v;
  }
}

// ---------------------------------------------------------------- path 3: up=2 polyphase (f32, down=1, k<=4x4)
// ToRGB skip upsample: out is 4x the input, so the kernel is store-bound.  A thread owns a 2-row x 4-column output
// block (two dwordx4 stores); for each output only the taps whose zero-stuffed sample is non-zero are visited
// (ky = (pad_y0 - oy) mod 2, +2: 2x2 of the 4x4 taps), inputs come through L1 (each is reused by ~4 threads).
struct U2Params {
  int planes, in_h, in_w, out_h, out_w, pad_x0, pad_y0, kh, kw, bw, bh;   // bw/bh: 4x2 blocks per row/col
};

// A thread owns a 2 x 4 output block whose origin is even in both directions, so which taps meet a non-zero sample of
// the zero-stuffed input depends only on the PARITY of the padding (template parameters): every tap and input index
// below is a compile-time constant.  The 2 x 4 outputs read a 3 x 4 input neighbourhood (one 16-byte load per row when
// it lies inside the image) and cost 32 FMAs — the first version re-derived the taps per output with 16-way selects
// and issued 32 scalar loads (61 us for [24,512,512] -> [24,1024,1024]; store-bound at ~25 us).
template <int PY, int PX>
__global__ __launch_bounds__(256) void ufd_up2_f32(const float* __restrict__ in, const float* __restrict__ kern,
                                                   float* __restrict__ out, const U2Params p) {
  float kf[4][4];
#pragma unroll
  for (int ky = 0; ky < 4; ++ky)
#pragma unroll
    for (int kx = 0; kx < 4; ++kx)
      kf[ky][kx] = (ky < p.kh && kx < p.kw) ? kern[(p.kh - 1 - ky) * p.kw + (p.kw - 1 - kx)] : 0.f;
  // output (dy, dx) of a block takes taps ky = KY0(dy) + 2a, kx = KX0(dx) + 2c from input (RY(dy,a), CX(dx,c)) relative
  // to (oy0/2 - pad_y0/2, ox0/2 - pad_x0/2); >> is an arithmetic shift (floor), as the zero-stuffed index needs
  constexpr auto KY0 = [](int dy) { return (PY - dy) & 1; };
  constexpr auto KX0 = [](int dx) { return (PX - dx) & 1; };
  constexpr auto RY = [](int dy, int a) { return (dy + ((PY - dy) & 1) + 2 * a - PY) >> 1; };
  constexpr auto CX = [](int dx, int c) { return (dx + ((PX - dx) & 1) + 2 * c - PX) >> 1; };
  constexpr int RMIN = RY(0, 0) < RY(1, 0) ? RY(0, 0) : RY(1, 0);
  constexpr int CMIN = CX(0, 0) < CX(1, 0) ? CX(0, 0) : CX(1, 0);
  static_assert(RY(1, 1) - RMIN <= 2 && RY(0, 1) - RMIN <= 2, "3 input rows");
  static_assert(CX(3, 1) - CMIN <= 3 && CX(2, 1) - CMIN <= 3, "4 input columns");
  const int qy = (p.pad_y0 - PY) / 2, qx = (p.pad_x0 - PX) / 2;      // exact: pad - parity is even (also when negative)
  const long long total = (long long)p.planes * p.bh * p.bw;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const int bx = (int)(idx % p.bw);
    const long long t = idx / p.bw;
    const int by = (int)(t % p.bh);
    const long long pl = t / p.bh;
    const float* pin = in + pl * (long long)p.in_h * p.in_w;
    float* pout = out + pl * (long long)p.out_h * p.out_w;
    const int ox0 = bx * 4, oy0 = by * 2;
    const int iy0 = by - qy + RMIN, ix0 = bx * 2 - qx + CMIN;          // first row / column of the neighbourhood
    float v[3][4];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const int iy = iy0 + r;
      const bool yok = iy >= 0 && iy < p.in_h;
      const float* rp = pin + (long long)(yok ? iy : 0) * p.in_w;
      if (yok && ix0 >= 0 && ix0 + 3 < p.in_w) {
        const f32x4_u q = *reinterpret_cast<const f32x4_u*>(rp + ix0);
        v[r][0] = q.x; v[r][1] = q.y; v[r][2] = q.z; v[r][3] = q.w;
      } else {
#pragma unroll
        for (int c = 0; c < 4; ++c) v[r][c] = (yok && ix0 + c >= 0 && ix0 + c < p.in_w) ? rp[ix0 + c] : 0.f;
      }
    }
#pragma unroll
    for (int dy = 0; dy < 2; ++dy) {
      const int oy = oy0 + dy;
      if (oy >= p.out_h) break;
      float acc[4] = {0.f, 0.f, 0.f, 0.f};
      // same order of accumulation as the one-output-at-a-time form: ky outer, kx inner
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int dx = 0; dx < 4; ++dx)
#pragma unroll
          for (int c = 0; c < 2; ++c)
            acc[dx] = fmaf(v[RY(dy, a) - RMIN][CX(dx, c) - CMIN], kf[KY0(dy) + 2 * a][KX0(dx) + 2 * c], acc[dx]);
      float* op = pout + (long long)oy * p.out_w + ox0;
      if (ox0 + 4 <= p.out_w) {
        f32x4_u q; q.x = acc[0]; q.y = acc[1]; q.z = acc[2]; q.w = acc[3];
        *reinterpret_cast<f32x4_u*>(op) = q;
      } else {
#pragma unroll
        for (int dx = 0; dx < 4; ++dx)
          if (ox0 + dx < p.out_w) op[dx] = acc[dx];
      }
    }
  }
}

template <typename T>
int launch_generic(const void* in, const void* kern, void* out, const UfdParams& p, hipStream_t s) {
  const long long total = (long long)p.major * p.out_h * p.out_w * p.minor;
  if (total == 0) return FMGAN_OK;
  long long blocks = (total + 255) / 256;
  const long long cap = (long long)FMGAN_NUM_CU * 32;
  if (blocks > cap) blocks = cap;
  hipLaunchKernelGGL(ufd_generic<T>, dim3((unsigned)blocks), dim3(256), 0, s, (const T*)in, (const T*)kern, (T*)out, p, total);
  return fmgan_check_launch();
}

bool rowmarch_ok(int dtype, const UfdParams& p) {
  return dtype == FMGAN_F32 && p.minor == 1 && p.up_x == 1 && p.up_y == 1 && p.down_x == 1 && p.down_y == 1 &&
         p.kh <= 4 && p.kw <= 4 && p.out_w >= 64 && p.out_h >= 4;
}

struct UfdEpilogue {
  const float* noise; const float* noise_weight; const float* bias;
  int channels, noise_batch; float alpha, act_scale;
};

// Path 1b serves the aligned-row layout only: position 0 of every row (= logical column -pad_x0) on a 16-byte
// boundary, row / plane pitch multiples of 4 floats, rows of the output 16-byte aligned, a noise plane when fused.
// force_path 4 keeps path 1, 5 insists on path 1b (A/B tests and measurements; no environment switch in the library).
bool dmaring_ok(const void* in, const void* out, const UfdParams& p, const UfdEpilogue* ep) {
  const uintptr_t in0 = (uintptr_t)in - 4u * (unsigned)p.pad_x0;
  if (p.pad_x0 < 0 || p.pad_x0 > 3 || (in0 & 15) || ((uintptr_t)out & 15)) return false;
  if ((p.in_row_stride & 3) || (p.in_plane_stride & 3) || (p.out_w & 3) || p.out_w < 256 || p.out_h < 8) return false;
  if (p.in_row_stride < p.in_w + p.pad_x0) return false;                       // the shifted row must fit its pitch
  if ((long long)p.in_h * p.in_row_stride * 4 > 0x7fffffffLL) return false;    // num_records / soffset are 32-bit
  if (ep && (!ep->noise || ((uintptr_t)ep->noise & 15))) return false;
  return true;
}

int launch_dmaring(const void* in, const void* kern, void* out, const UfdParams& p, hipStream_t s,
                   const UfdEpilogue* ep) {
  DRParams r{};
  r.in0 = (const float*)in - p.pad_x0; r.out = (float*)out; r.kern = (const float*)kern;
  if (ep) {
    r.noise = ep->noise; r.noise_weight = ep->noise_weight; r.bias = ep->bias;
    r.channels = ep->channels; r.noise_batch = ep->noise_batch; r.alpha = ep->alpha; r.act_scale = ep->act_scale;
  }
  r.planes = p.major; r.in_h = p.in_h; r.in_w = p.in_w; r.out_h = p.out_h; r.out_w = p.out_w;
  r.rs = p.in_row_stride; r.ps = p.in_plane_stride; r.pad_x0 = p.pad_x0; r.pad_y0 = p.pad_y0; r.kh = p.kh; r.kw = p.kw;
  r.strips = (p.out_w + 255) / 256;
  // Largest row tile (halo re-read = 3/TH) that divides the height and still leaves >= 32 waves per CU in the grid.
  const long long want = (long long)FMGAN_NUM_CU * 32;
  int th = 64;
  while (th > 8 && ((p.out_h % th) != 0 || (long long)p.major * r.strips * (p.out_h / th) < want)) th >>= 1;
  r.th = th;
  r.tiles_y = (p.out_h + th - 1) / th;
  r.total_waves = (long long)p.major * r.strips * r.tiles_y;
  const long long blocks = (r.total_waves + 3) / 4;
  if (blocks > 0x7fffffffLL) return FMGAN_EOVERFLOW;
  const dim3 g((unsigned)blocks), b(256);
  const size_t lds = 4 * 4 * 320 * sizeof(float);   // 4 waves x 4 slots x (256 + 64) floats
  // Block order and cache policy: hardware order, cached loads.  Blocks are dealt round-robin over the 8 XCDs, so with
  // tiles numbered (strip, row tile, plane) an XCD sees the SAME few (strip, row tile) positions of every plane: its
  // share of the noise plane is a few hundred KB that stays in its L2 for all planes, where the XCD-contiguous order of
  // path 1 re-fetches it per plane (FETCH_SIZE of the fused headline blur, standalone: 1.68x the input with
  // XCD-contiguous order, 1.51x with nt loads on top, 1.12x in hardware order; plain blur 1.08x either way — the
  // <.., NT, XCD> instantiations those numbers came from are no longer built).
  if (ep) hipLaunchKernelGGL((ufd_dmaring_f32<true, false, false>), g, b, lds, s, r);
  else hipLaunchKernelGGL((ufd_dmaring_f32<false, false, false>), g, b, lds, s, r);
  return fmgan_check_launch();
}

// variant: -1 / 1 pick between path 1 and 1b, 4 = path 1 (register row-march), 5 = path 1b (LDS-DMA ring) or refuse
int launch_rowmarch(const void* in, const void* kern, void* out, const UfdParams& p, hipStream_t s,
                    const UfdEpilogue* ep = nullptr, int variant = -1) {
  if (variant == 5 && !dmaring_ok(in, out, p, ep)) return FMGAN_EUNSUPPORTED;
  if (variant != 4 && dmaring_ok(in, out, p, ep)) return launch_dmaring(in, kern, out, p, s, ep);
  RMParams r{};
  if (ep) {
    r.fuse = 1; r.noise = ep->noise; r.noise_weight = ep->noise_weight; r.bias = ep->bias;
    r.channels = ep->channels; r.noise_batch = ep->noise_batch; r.alpha = ep->alpha; r.act_scale = ep->act_scale;
  }
  r.planes = p.major; r.in_h = p.in_h; r.in_w = p.in_w; r.out_h = p.out_h; r.out_w = p.out_w;
  r.pad_x0 = p.pad_x0; r.pad_y0 = p.pad_y0; r.kh = p.kh; r.kw = p.kw;
  r.in_plane_stride = p.in_plane_stride; r.in_row_stride = p.in_row_stride;
  const int vec = p.out_w >= 192 ? 4 : (p.out_w >= 96 ? 2 : 1);
  r.strips = (p.out_w + 64 * vec - 1) / (64 * vec);
  // Largest row tile that still leaves >= 32 waves per CU in the grid (halo re-read = 3/TH).
  const long long want = (long long)FMGAN_NUM_CU * 32;
  int th = 64;
  while (th > 4) {
    const long long waves = (long long)p.major * r.strips * ((p.out_h + th - 1) / th);
    if (waves >= want) break;
    th >>= 1;
  }
  // outputs far larger than the 256 MiB Infinity Cache cannot be re-read from it by the consumer: store them
  // non-temporal (measured +3 % on the 1 GB headline call); small layers keep the default policy.
  const bool env_nt = (long long)p.major * p.out_h * p.out_w * 4 >= (512LL << 20);
  r.th = th;
  r.tiles_y = (p.out_h + th - 1) / th;
  r.total_waves = (long long)p.major * r.strips * r.tiles_y;
  const long long blocks = (r.total_waves + 3) / 4;
  if (blocks > 0x7fffffffLL) return FMGAN_EOVERFLOW;
  const dim3 g((unsigned)blocks), b(256);
  if (vec == 4 && env_nt) hipLaunchKernelGGL((ufd_rowmarch_f32<4, true>), g, b, 0, s, (const float*)in, (const float*)kern, (float*)out, r);
  else if (vec == 4) hipLaunchKernelGGL(ufd_rowmarch_f32<4>, g, b, 0, s, (const float*)in, (const float*)kern, (float*)out, r);
  else if (vec == 2) hipLaunchKernelGGL(ufd_rowmarch_f32<2>, g, b, 0, s, (const float*)in, (const float*)kern, (float*)out, r);
  else hipLaunchKernelGGL(ufd_rowmarch_f32<1>, g, b, 0, s, (const float*)in, (const float*)kern, (float*)out, r);
  return fmgan_check_launch();
}

bool planetile_ok(int dtype, const UfdParams& p) {
  return dtype == FMGAN_F32 && p.minor == 1 && p.up_x == 1 && p.up_y == 1 && p.down_x == 1 && p.down_y == 1 &&
         p.kh <= 4 && p.kw <= 4 && (long long)p.in_h * p.in_w <= 12288 && (long long)p.out_h * p.out_w <= (1 << 20) &&
         (long long)p.in_h * p.in_row_stride <= 0x7fffffffLL;
}

int launch_planetile(const void* in, const void* kern, void* out, const UfdParams& p, hipStream_t s,
                     const UfdEpilogue* ep = nullptr) {
  PTParams t{p.major, p.in_h, p.in_w, p.out_h, p.out_w, p.pad_x0, p.pad_y0, p.kh, p.kw, 1};
  t.in_plane_stride = p.in_plane_stride; t.in_row_stride = p.in_row_stride;
  if (ep) {
    t.noise = ep->noise; t.noise_weight = ep->noise_weight; t.bias = ep->bias; t.channels = ep->channels;
    t.noise_batch = ep->noise_batch; t.alpha = ep->alpha; t.act_scale = ep->act_scale;
  }
  const int in_sz = p.in_h * p.in_w;
  int pb = 8192 / in_sz;                      // ~32 KB of LDS per block
  if (pb < 1) pb = 1;
  if (pb > 64) pb = 64;
  // keep at least ~2 blocks per CU when the layer is small
  while (pb > 1 && (p.major + pb - 1) / pb < 2 * FMGAN_NUM_CU) pb >>= 1;
  t.pb = pb;
  const unsigned blocks = (unsigned)((p.major + pb - 1) / pb);
  if (ep) hipLaunchKernelGGL(ufd_planetile_f32<true>, dim3(blocks), dim3(256), sizeof(float) * (size_t)pb * in_sz, s,
                             (const float*)in, (const float*)kern, (float*)out, t);
  else hipLaunchKernelGGL(ufd_planetile_f32<false>, dim3(blocks), dim3(256), sizeof(float) * (size_t)pb * in_sz, s,
                          (const float*)in, (const float*)kern, (float*)out, t);
  return fmgan_check_launch();
}

bool up2_ok(int dtype, const UfdParams& p) {
  return dtype == FMGAN_F32 && p.minor == 1 && p.up_x == 2 && p.up_y == 2 && p.down_x == 1 && p.down_y == 1 &&
         p.kh <= 4 && p.kw <= 4 && p.in_row_stride == p.in_w && p.in_plane_stride == (long long)p.in_h * p.in_w;
}

int launch_up2(const void* in, const void* kern, void* out, const UfdParams& p, hipStream_t s) {
  U2Params t{p.major, p.in_h, p.in_w, p.out_h, p.out_w, p.pad_x0, p.pad_y0, p.kh, p.kw, (p.out_w + 3) / 4,
             (p.out_h + 1) / 2};
  const long long total = (long long)p.major * t.bw * t.bh;
  long long blocks = (total + 255) / 256;
  const long long cap = (long long)FMGAN_NUM_CU * 32;
  if (blocks > cap) blocks = cap;
  const int par = (p.pad_y0 & 1) * 2 + (p.pad_x0 & 1);
  const dim3 g((unsigned)blocks), b(256);
  switch (par) {
    case 0: hipLaunchKernelGGL((ufd_up2_f32<0, 0>), g, b, 0, s, (const float*)in, (const float*)kern, (float*)out, t); break;
    case 1: hipLaunchKernelGGL((ufd_up2_f32<0, 1>), g, b, 0, s, (const float*)in, (const float*)kern, (float*)out, t); break;
    case 2: hipLaunchKernelGGL((ufd_up2_f32<1, 0>), g, b, 0, s, (const float*)in, (const float*)kern, (float*)out, t); break;
    default: hipLaunchKernelGGL((ufd_up2_f32<1, 1>), g, b, 0, s, (const float*)in, (const float*)kern, (float*)out, t); break;
  }
  return fmgan_check_launch();
}

int pick_path(int dtype, const UfdParams& p) {
  if (rowmarch_ok(dtype, p)) return 1;
  if (planetile_ok(dtype, p)) return 2;
  if (up2_ok(dtype, p)) return 3;
  return 0;
}

int validate(int dtype, int major, int in_h, int in_w, int minor, int kh, int kw, int up_x, int up_y, int down_x,
             int down_y) {
  if (dtype != FMGAN_F32 && dtype != FMGAN_F64 && dtype != FMGAN_F16) return FMGAN_EUNSUPPORTED;
  if (major < 0 || in_h <= 0 || in_w <= 0 || minor <= 0 || kh <= 0 || kw <= 0) return FMGAN_EINVAL;
  if (up_x <= 0 || up_y <= 0 || down_x <= 0 || down_y <= 0) return FMGAN_EINVAL;
  return FMGAN_OK;
}

}  // namespace

extern "C" int fmgan_upfirdn2d_out_size(int in_h, int in_w, int kernel_h, int kernel_w, int up_x, int up_y, int down_x,
                                        int down_y, int pad_x0, int pad_x1, int pad_y0, int pad_y1, int* out_h,
                                        int* out_w) {
  if (!out_h || !out_w || up_x <= 0 || up_y <= 0 || down_x <= 0 || down_y <= 0) return FMGAN_EINVAL;
  // op/upfirdn2d_kernel.cu:237-240
  *out_h = (in_h * up_y + pad_y0 + pad_y1 - kernel_h + down_y) / down_y;
  *out_w = (in_w * up_x + pad_x0 + pad_x1 - kernel_w + down_x) / down_x;
  return FMGAN_OK;
}

extern "C" int fmgan_upfirdn2d_select(int dtype, int major, int in_h, int in_w, int minor, int kernel_h, int kernel_w,
                                      int up_x, int up_y, int down_x, int down_y, int pad_x0, int pad_x1, int pad_y0,
                                      int pad_y1) {
  int st = validate(dtype, major, in_h, in_w, minor, kernel_h, kernel_w, up_x, up_y, down_x, down_y);
  if (st != FMGAN_OK) return st;
  UfdParams p{major, in_h, in_w, minor, kernel_h, kernel_w, up_x, up_y, down_x, down_y, pad_x0, pad_y0, 0, 0,
              (long long)in_h * in_w * minor, in_w * minor};
  fmgan_upfirdn2d_out_size(in_h, in_w, kernel_h, kernel_w, up_x, up_y, down_x, down_y, pad_x0, pad_x1, pad_y0, pad_y1,
                           &p.out_h, &p.out_w);
  if (p.out_h <= 0 || p.out_w <= 0) return FMGAN_EINVAL;
  return pick_path(dtype, p);
}

extern "C" int fmgan_upfirdn2d_strided(int dtype, const void* input, const void* kernel, void* out, int major,
                                       int in_h, int in_w, int minor, long long in_plane_stride, int in_row_stride,
                                       int kernel_h, int kernel_w, int up_x, int up_y, int down_x, int down_y,
                                       int pad_x0, int pad_x1, int pad_y0, int pad_y1, int force_path, void* stream) {
  int st = validate(dtype, major, in_h, in_w, minor, kernel_h, kernel_w, up_x, up_y, down_x, down_y);
  if (st != FMGAN_OK) return st;
  if (in_row_stride < in_w * minor || in_plane_stride < (long long)(in_h - 1) * in_row_stride + (long long)in_w * minor)
    return FMGAN_EINVAL;
  UfdParams p{major, in_h, in_w, minor, kernel_h, kernel_w, up_x, up_y, down_x, down_y, pad_x0, pad_y0, 0, 0,
              in_plane_stride, in_row_stride};
  fmgan_upfirdn2d_out_size(in_h, in_w, kernel_h, kernel_w, up_x, up_y, down_x, down_y, pad_x0, pad_x1, pad_y0, pad_y1,
                           &p.out_h, &p.out_w);
  if (p.out_h <= 0 || p.out_w <= 0) return FMGAN_EINVAL;
  if (major == 0) return FMGAN_OK;
  if (!input || !kernel || !out) return FMGAN_EINVAL;
  if ((long long)in_h * in_row_stride > 0x7fffffffLL || (long long)p.out_h * p.out_w * minor > 0x7fffffffLL)
    return FMGAN_EOVERFLOW;
  hipStream_t s = (hipStream_t)stream;
  int path = force_path;
  if (path < 0) path = pick_path(dtype, p);
  switch (path) {
    case 0:
      if (dtype == FMGAN_F32) return launch_generic<float>(input, kernel, out, p, s);
      if (dtype == FMGAN_F64) return launch_generic<double>(input, kernel, out, p, s);
      return launch_generic<__half>(input, kernel, out, p, s);
    case 1:
    case 4:
    case 5:
      if (!rowmarch_ok(dtype, p)) return FMGAN_EUNSUPPORTED;
      return launch_rowmarch(input, kernel, out, p, s, nullptr, path);
    case 2:
      if (!planetile_ok(dtype, p)) return FMGAN_EUNSUPPORTED;
      return launch_planetile(input, kernel, out, p, s);
    case 3:
      if (!up2_ok(dtype, p)) return FMGAN_EUNSUPPORTED;
      return launch_up2(input, kernel, out, p, s);
    default:
      return FMGAN_EUNSUPPORTED;
  }
}

extern "C" int fmgan_blur_noise_bias_act_path_f32(const float* input, const float* kernel, float* out, int batch,
                                                  int channels, int in_h, int in_w, long long in_plane_stride,
                                                  int in_row_stride, int kernel_h, int kernel_w, int pad_x0, int pad_x1,
                                                  int pad_y0, int pad_y1, const float* noise, const float* noise_weight,
                                                  const float* bias, int noise_batch, float alpha, float act_scale,
                                                  int force_path, void* stream) {
  if (force_path != -1 && force_path != 1 && force_path != 4 && force_path != 5) return FMGAN_EUNSUPPORTED;
  if (batch < 0 || channels <= 0) return FMGAN_EINVAL;
  const long long major = (long long)batch * channels;
  if (major > 0x7fffffffLL) return FMGAN_EOVERFLOW;
  int st = validate(FMGAN_F32, (int)major, in_h, in_w, 1, kernel_h, kernel_w, 1, 1, 1, 1);
  if (st != FMGAN_OK) return st;
  if (in_row_stride < in_w || in_plane_stride < (long long)(in_h - 1) * in_row_stride + in_w) return FMGAN_EINVAL;
  if (noise && noise_batch != 1 && noise_batch != batch) return FMGAN_EINVAL;
  UfdParams p{(int)major, in_h, in_w, 1, kernel_h, kernel_w, 1, 1, 1, 1, pad_x0, pad_y0, 0, 0, in_plane_stride,
              in_row_stride};
  fmgan_upfirdn2d_out_size(in_h, in_w, kernel_h, kernel_w, 1, 1, 1, 1, pad_x0, pad_x1, pad_y0, pad_y1, &p.out_h, &p.out_w);
  if (p.out_h <= 0 || p.out_w <= 0) return FMGAN_EINVAL;
  if (major == 0) return FMGAN_OK;
  if (!input || !kernel || !out) return FMGAN_EINVAL;
  if ((long long)in_h * in_row_stride > 0x7fffffffLL) return FMGAN_EOVERFLOW;
  UfdEpilogue ep{noise, noise_weight, bias, channels, noise_batch, alpha, act_scale};
  if (!rowmarch_ok(FMGAN_F32, p)) {
    // small planes (the 4^2..32^2 upsampling layers): the plane-tile kernel reads the aligned-row layout and applies
    // the epilogue in its store
    if (force_path > 1 || !planetile_ok(FMGAN_F32, p)) return FMGAN_EUNSUPPORTED;
    return launch_planetile(input, kernel, out, p, (hipStream_t)stream, &ep);
  }
  return launch_rowmarch(input, kernel, out, p, (hipStream_t)stream, &ep, force_path);
}

// Which kernel fmgan_blur_noise_bias_act_f32 would run for these arguments (host logic, nothing is launched):
// 5 = LDS-DMA ring (path 1b), 1 = register row-march (path 1), 2 = plane-tile, FMGAN_EUNSUPPORTED = none.
extern "C" int fmgan_blur_noise_bias_act_select(const float* input, const float* out, const float* noise, int batch,
                                                int channels, int in_h, int in_w, long long in_plane_stride,
                                                int in_row_stride, int kernel_h, int kernel_w, int pad_x0, int pad_x1,
                                                int pad_y0, int pad_y1) {
  if (batch <= 0 || channels <= 0) return FMGAN_EINVAL;
  const long long major = (long long)batch * channels;
  if (major > 0x7fffffffLL) return FMGAN_EOVERFLOW;
  int st = validate(FMGAN_F32, (int)major, in_h, in_w, 1, kernel_h, kernel_w, 1, 1, 1, 1);
  if (st != FMGAN_OK) return st;
  UfdParams p{(int)major, in_h, in_w, 1, kernel_h, kernel_w, 1, 1, 1, 1, pad_x0, pad_y0, 0, 0, in_plane_stride,
              in_row_stride};
  fmgan_upfirdn2d_out_size(in_h, in_w, kernel_h, kernel_w, 1, 1, 1, 1, pad_x0, pad_x1, pad_y0, pad_y1, &p.out_h, &p.out_w);
  if (p.out_h <= 0 || p.out_w <= 0) return FMGAN_EINVAL;
  UfdEpilogue ep{noise, nullptr, nullptr, channels, 1, 0.f, 1.f};
  if (rowmarch_ok(FMGAN_F32, p)) return dmaring_ok(input, out, p, &ep) ? 5 : 1;
  return planetile_ok(FMGAN_F32, p) ? 2 : FMGAN_EUNSUPPORTED;
}

extern "C" int fmgan_blur_noise_bias_act_f32(const float* input, const float* kernel, float* out, int batch,
                                             int channels, int in_h, int in_w, long long in_plane_stride,
                                             int in_row_stride, int kernel_h, int kernel_w, int pad_x0, int pad_x1,
                                             int pad_y0, int pad_y1, const float* noise, const float* noise_weight,
                                             const float* bias, int noise_batch, float alpha, float act_scale,
                                             void* stream) {
  return fmgan_blur_noise_bias_act_path_f32(input, kernel, out, batch, channels, in_h, in_w, in_plane_stride,
                                            in_row_stride, kernel_h, kernel_w, pad_x0, pad_x1, pad_y0, pad_y1, noise,
                                            noise_weight, bias, noise_batch, alpha, act_scale, -1, stream);
}

extern "C" int fmgan_upfirdn2d(int dtype, const void* input, const void* kernel, void* out, int major, int in_h,
                               int in_w, int minor, int kernel_h, int kernel_w, int up_x, int up_y, int down_x,
                               int down_y, int pad_x0, int pad_x1, int pad_y0, int pad_y1, int force_path,
                               void* stream) {
  return fmgan_upfirdn2d_strided(dtype, input, kernel, out, major, in_h, in_w, minor, (long long)in_h * in_w * minor,
                                 in_w * minor, kernel_h, kernel_w, up_x, up_y, down_x, down_y, pad_x0, pad_x1, pad_y0,
                                 pad_y1, force_path, stream);
}
