// Headline blur ([256,1025,1025] -> [256,1024,1024], 4x4 FIR, fused noise/bias/lrelu store) design probe (gfx950).
// Compares the product kernel (libfmgan_hip.so) with an LDS-DMA row ring:
//   * each wave owns a 256-column strip and marches TH rows; an input row segment goes HBM -> LDS by
//     `buffer_load ... lds` (no VGPR staging, no address VALU: lane offsets are row-invariant, the row is an SGPR
//     soffset), 4 slots per wave = 3 rows in flight; the wave synchronises with itself by s_waitcnt vmcnt(N) only;
//   * the 7-float window of a lane is two ds_read_b128 (the 3 halo columns are simply the next lane's first floats);
//   * SEP: rank-1 taps -> horizontal pass on arrival (16 FMA / row) + vertical pass per output row (16 FMA) instead of 64.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/exp/blur_probe tools/exp/blur_probe.hip -ldl
// run (GPU box, from the repo root): tools/exp/blur_probe
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <utility>

typedef float f4 __attribute__((ext_vector_type(4)));

#define NUM_XCD 8
__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nblk) {
  const unsigned q = nblk / NUM_XCD, r = nblk % NUM_XCD;
  const unsigned xcd = bid % NUM_XCD, slot = bid / NUM_XCD;
  const unsigned base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + slot;
}

template <int SIZE, int AUX = 0, typename RSRC>
__device__ __forceinline__ void dma_to_lds(RSRC rsrc, float* lds, unsigned voff, unsigned soff) {
#if defined(__HIP_DEVICE_COMPILE__)
  typedef __attribute__((address_space(3))) void* lds_void_ptr;
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void_ptr)lds, SIZE, voff, soff, 0, AUX);
#endif
}

template <int N> __device__ __forceinline__ void wait_vm() {
#if defined(__HIP_DEVICE_COMPILE__)
  // gfx9 encoding: vmcnt = imm[3:0] | imm[15:14]<<4, expcnt imm[6:4], lgkmcnt imm[11:8]
  __builtin_amdgcn_s_waitcnt((N & 0xF) | ((N >> 4) << 14) | (0x7 << 4) | (0xF << 8));
#endif
}

struct V2P {
  const float* in;   // position 0 of row 0 of plane 0, 16-byte aligned; logical column c sits at position c + pad_x0
  float* out; const float* noise; const float* nwp; const float* bias;
  int planes, channels, in_h, in_w, out_h, out_w, rs, pad_x0, pad_y0;
  long long ps;
  int th, strips, tiles_y; long long total_waves;
  float alpha, scale;
  float kh[4], kv[4];   // flipped rank-1 factors: tap[ky][kx] = kv[ky] * kh[kx]
  float k2[16];         // flipped 2-D taps
};

// VMEM operations issued after stage(k)'s two DMAs and before step k's wait (step order: see STEP below).
__host__ __device__ constexpr int nwait(int k, bool epi) {
  int n = 0;
  if (k >= 3) { if (k - 3 >= 3) ++n; }          // the store of the step that issued stage(k)
  else n += 2 * (2 - k);                        // prologue stages k+1..2
  for (int s = (k - 2 > 0 ? k - 2 : 0); s <= k; ++s) { n += 2 + (epi ? 1 : 0); if (s < k && s >= 3) ++n; }
  return n;
}

template <bool SEP, bool EPI, bool NT, int MINB, bool REMAP = true>
__global__ __launch_bounds__(256, MINB) void blur_v2(const V2P p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int S = 4, SLOT = 320;
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned lb = REMAP ? xcd_remap(blockIdx.x, gridDim.x) : blockIdx.x;
  const long long gw = (long long)lb * 4 + wv;
  if (gw >= p.total_waves) return;
  const int strip = (int)(gw % p.strips);
  const long long t = gw / p.strips;
  const int ty = (int)(t % p.tiles_y);
  const long long plane = t / p.tiles_y;
  float* ring = lds + wv * (S * SLOT);

  const int c0 = (strip * 64 + lane) * 4;
  const unsigned PARK = 0xFFFFFFF0u;
  const unsigned vmain = (c0 + 4 <= p.rs) ? (unsigned)c0 * 4u : PARK;
  const int xe = (strip + 1) * 256 + lane;
  const unsigned vext = (lane < 4 && xe < p.rs) ? (unsigned)xe * 4u : PARK;
  unsigned mask = 0;
#pragma unroll
  for (int i = 0; i < 7; ++i) {
    const int col = c0 + i - p.pad_x0;
    mask |= (col >= 0 && col < p.in_w) ? 1u << i : 0u;
  }
  const int oy0 = ty * p.th, oy_end = min(oy0 + p.th, p.out_h);
  const bool full = oy0 + p.th <= p.out_h;   // wave-uniform: the static vmcnt counts assume every step stores
  const int iy0 = oy0 - p.pad_y0;
  const int nsteps = p.th + 3;
  float* pin = const_cast<float*>(p.in + plane * p.ps);
  float* pout = p.out + plane * (long long)p.out_h * p.out_w;

  float nw = 0.f, bv = 0.f;
  if (EPI) { nw = p.nwp[0]; bv = p.bias[(int)(plane % p.channels)]; }

  float W[4][SEP ? 4 : 7];   // SEP: horizontally filtered rows; else the raw 7-column windows
  f4 nz[4];   // noise rows in flight: row k is requested at step k, BEFORE stage(k+3)'s DMAs (returns are in order, so the
             // wait for stage(k+3) at step k+3 covers it), and used at step k+3

  // The noise request is inline asm on purpose: the compiler's waitcnt model merges control-flow paths conservatively
  // and would put a near-zero vmcnt before every use, draining the DMA ring.  The data is covered by the wait for
  // stage(k+3) (in-order returns); noise_ack() after that wait is where the value becomes usable.
  const unsigned nvoff = (unsigned)c0 * 4u;
  auto noise_req = [&](f4& dst, int row) {
    const float* rowp = p.noise + (long long)row * p.out_w;   // wave-uniform
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(nvoff), "s"(rowp) : "memory");
#endif
  };
  auto noise_ack = [&](f4& v) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(v) : : "memory");
#endif
  };
  auto issue = [&](int k, int slot) {
    const int iy = iy0 + k;
    const bool ok = k < nsteps && iy >= 0 && iy < p.in_h;   // wave-uniform
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(pin, 0, ok ? p.in_h * p.rs * 4 : 0, 0x00020000);   // whole plane: the range check covers voffset + soffset
    const unsigned soff = ok ? (unsigned)iy * (unsigned)p.rs * 4u : 0u;
    dma_to_lds<16>(rsrc, ring + slot * SLOT, vmain, soff);
    dma_to_lds<4>(rsrc, ring + slot * SLOT + 256, vext, soff);
  };
  auto take = [&](int slot, int wi) {
    const f4 a = *reinterpret_cast<const f4*>(ring + slot * SLOT + lane * 4);
    const f4 b = *reinterpret_cast<const f4*>(ring + slot * SLOT + lane * 4 + 4);
    float x[7] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z};
#pragma unroll
    for (int i = 0; i < 7; ++i) x[i] = (mask >> i) & 1 ? x[i] : 0.f;
    if constexpr (SEP) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float v = x[e] * p.kh[0];
        v = fmaf(x[e + 1], p.kh[1], v);
        v = fmaf(x[e + 2], p.kh[2], v);
        v = fmaf(x[e + 3], p.kh[3], v);
        W[wi][e] = v;
      }
    } else {
#pragma unroll
      for (int i = 0; i < 7; ++i) W[wi][i] = x[i];
    }
  };
  auto emit = [&](int oy, int w0, int w1, int w2, int w3, const f4& nz_cur) {
    if (oy >= oy_end) return;
    float acc[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if constexpr (SEP) {
        float v = W[w0][e] * p.kv[0];
        v = fmaf(W[w1][e], p.kv[1], v);
        v = fmaf(W[w2][e], p.kv[2], v);
        v = fmaf(W[w3][e], p.kv[3], v);
        acc[e] = v;
      } else {
        float v = 0.f;
#pragma unroll
        for (int kx = 0; kx < 4; ++kx) v = fmaf(W[w0][e + kx], p.k2[0 + kx], v);
#pragma unroll
        for (int kx = 0; kx < 4; ++kx) v = fmaf(W[w1][e + kx], p.k2[4 + kx], v);
#pragma unroll
        for (int kx = 0; kx < 4; ++kx) v = fmaf(W[w2][e + kx], p.k2[8 + kx], v);
#pragma unroll
        for (int kx = 0; kx < 4; ++kx) v = fmaf(W[w3][e + kx], p.k2[12 + kx], v);
        acc[e] = v;
      }
    }
    if constexpr (EPI) {
      const float n[4] = {nz_cur.x, nz_cur.y, nz_cur.z, nz_cur.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float v = __fadd_rn(__fadd_rn(acc[e], __fmul_rn(nw, n[e])), bv);
        acc[e] = __fmul_rn(v > 0.f ? v : __fmul_rn(v, p.alpha), p.scale);
      }
    }
    const f4 o = {acc[0], acc[1], acc[2], acc[3]};
    f4* op = reinterpret_cast<f4*>(pout + (long long)oy * p.out_w + c0);
    if constexpr (NT) __builtin_nontemporal_store(o, op); else *op = o;   // probe: out_w % 256 == 0
  };

  // step k:  request noise row k | DMA stage(k+3) | wait for stage(k) | LDS -> window | output row k-3 (noise row k-3)
#define STEP(K, J, NW)                                                                                   \
  {                                                                                                      \
    if (EPI) noise_req(nz[(J) & 3], min(oy0 + (K), p.out_h - 1));                                        \
    issue((K) + 3, ((J) + 3) & 3);                                                                       \
    __builtin_amdgcn_sched_barrier(0);                                                                   \
    if (full) wait_vm<NW>(); else wait_vm<0>();                                                          \
    __builtin_amdgcn_sched_barrier(0);                                                                   \
    if (EPI && (K) >= 3) noise_ack(nz[((J) + 1) & 3]);                                                   \
    take((J) & 3, (J) & 3);                                                                              \
    if ((K) >= 3) emit(oy0 + (K) - 3, ((J) + 1) & 3, ((J) + 2) & 3, ((J) + 3) & 3, (J) & 3, nz[((J) + 1) & 3]); \
  }

  issue(0, 0); issue(1, 1); issue(2, 2);
  STEP(0, 0, nwait(0, EPI)) STEP(1, 1, nwait(1, EPI)) STEP(2, 2, nwait(2, EPI)) STEP(3, 3, nwait(3, EPI))
  STEP(4, 0, nwait(4, EPI)) STEP(5, 1, nwait(5, EPI)) STEP(6, 2, nwait(6, EPI)) STEP(7, 3, nwait(7, EPI))
  constexpr int NS = nwait(8, EPI);
  for (int kb = 8; kb < nsteps; kb += 4) {
    STEP(kb, 0, NS)
    if (kb + 1 >= nsteps) break;
    STEP(kb + 1, 1, NS)
    if (kb + 2 >= nsteps) break;
    STEP(kb + 2, 2, NS)
    if (kb + 3 >= nsteps) break;
    STEP(kb + 3, 3, NS)
  }
  wait_vm<0>();   // nothing of this wave's ring is in flight when it ends
#undef STEP
}


// ---------------------------------------------------------------------------------------------------------------
// v3: short row tiles (TH rows) so that what the chip touches at any moment is a few whole planes (DRAM-page friendly,
// like a linear copy), made cheap by a CHANNEL LOOP: a wave owns (sample, channel group, row tile, strip) and walks the
// CG channels of the group.  The noise tile (TH x 4 floats per lane) is loaded once into registers and reused by all
// channels; the DMA ring never drains between channels (the look-ahead of the last rows of channel c already fetches
// the first rows of channel c+1), and every vmcnt is a compile-time constant of the step position.
template <int... Js, typename F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, Js...>, F&& f) { (f(std::integral_constant<int, Js>{}), ...); }
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(std::make_integer_sequence<int, N>{}, f); }

struct V3P {
  const float* in; float* out; const float* noise; const float* nwp; const float* bias;
  int batch, channels, cg, groups, in_h, in_w, out_h, out_w, rs, pad_x0, pad_y0, noise_batch;
  long long ps;
  int strips, tiles_y; long long total_waves;
  float alpha, scale;
  float k2[16];
};

template <int TH, int S>
__host__ __device__ constexpr int nwait3(int j) {   // VMEM ops issued after stage(j)'s DMAs, before step j's wait
  constexpr int NSTEP = TH + 3, D = S - 1;
  int n = 0;
  for (int s = j - D; s <= j; ++s) {
    const int sm = ((s % NSTEP) + NSTEP) % NSTEP;
    if (s > j - D) n += 2;                 // the look-ahead DMAs of steps after the issuing one
    if (s < j && sm >= 3) n += 1;          // stores of earlier steps (the issuing step's store comes after its DMAs)
  }
  return n;
}

template <bool EPI, bool NT, int TH, int S, int MINW>
__global__ __launch_bounds__(256, MINW) void blur_v3(const V3P p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int SLOT = 320, NSTEP = TH + 3, D = S - 1;
  static_assert(D <= NSTEP - 3, "look-ahead must stay inside one channel's stored steps");
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned lb = xcd_remap(blockIdx.x, gridDim.x);
  const long long gw = (long long)lb * 4 + wv;
  if (gw >= p.total_waves) return;
  const int strip = (int)(gw % p.strips);
  long long t = gw / p.strips;
  const int ty = (int)(t % p.tiles_y); t /= p.tiles_y;
  const int g = (int)(t % p.groups);
  const int n = (int)(t / p.groups);
  const int ch0 = g * p.cg;
  const long long plane0 = (long long)n * p.channels + ch0;
  float* ring = lds + wv * (S * SLOT);

  const int c0 = (strip * 64 + lane) * 4;
  const unsigned PARK = 0xFFFFFFF0u;
  const unsigned vmain = (c0 + 4 <= p.rs) ? (unsigned)c0 * 4u : PARK;
  const int xe = (strip + 1) * 256 + lane;
  const unsigned vext = (lane < 4 && xe < p.rs) ? (unsigned)xe * 4u : PARK;
  unsigned mask = 0;
#pragma unroll
  for (int i = 0; i < 7; ++i) {
    const int col = c0 + i - p.pad_x0;
    mask |= (col >= 0 && col < p.in_w) ? 1u << i : 0u;
  }
  const int oy0 = ty * TH;
  const bool full = oy0 + TH <= p.out_h;
  const int iy0 = oy0 - p.pad_y0;

  f4 nzr[TH];
  float nw = 0.f;
  if (EPI) {
    nw = p.nwp[0];
    const float* np = p.noise + (p.noise_batch == 1 ? 0LL : (long long)n * p.out_h * p.out_w);
#pragma unroll
    for (int r = 0; r < TH; ++r)
      nzr[r] = *reinterpret_cast<const f4*>(np + (long long)min(oy0 + r, p.out_h - 1) * p.out_w + c0);
  }

  // bias of the group's channels: lane l holds channel ch0+l (cg <= 64); read per channel with v_readlane, so the
  // channel loop has no memory operation besides the ring's DMAs and the stores (the counts stay exact)
  float bvec = 0.f;
  if (EPI && lane < p.cg) bvec = p.bias[ch0 + lane];
  int slot_w = 0, slot_r = 0;   // wave-uniform ring cursors
  auto issue = [&](long long plane, bool plane_ok, int j) {
    const int iy = iy0 + j;
    const bool ok = plane_ok && iy >= 0 && iy < p.in_h;
    float* pin = const_cast<float*>(p.in + plane * p.ps);
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(pin, 0, ok ? p.in_h * p.rs * 4 : 0, 0x00020000);
    const unsigned soff = ok ? (unsigned)iy * (unsigned)p.rs * 4u : 0u;
    dma_to_lds<16>(rsrc, ring + slot_w * SLOT, vmain, soff);
    dma_to_lds<4>(rsrc, ring + slot_w * SLOT + 256, vext, soff);
    slot_w = slot_w + 1 == S ? 0 : slot_w + 1;
  };

#pragma unroll
  for (int j = 0; j < D; ++j) issue(plane0, true, j);

  for (int ci = 0; ci < p.cg; ++ci) {
    const long long plane = plane0 + ci;
    float* pout = p.out + plane * (long long)p.out_h * p.out_w;
    const float bv = EPI ? __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, bvec), ci)) : 0.f;
    const bool steady = ci > 0 && full;   // the static counts need the previous channel's stores in the queue
    float W[4][7];
    static_for<NSTEP>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      if (j + D < NSTEP) issue(plane, true, j + D);
      else issue(plane + 1, ci + 1 < p.cg, j + D - NSTEP);
      __builtin_amdgcn_sched_barrier(0);
      if (steady) wait_vm<nwait3<TH, S>(j)>(); else wait_vm<0>();
      __builtin_amdgcn_sched_barrier(0);
      {
        const float* sp = ring + slot_r * SLOT + lane * 4;
        const f4 a = *reinterpret_cast<const f4*>(sp);
        const f4 b = *reinterpret_cast<const f4*>(sp + 4);
        slot_r = slot_r + 1 == S ? 0 : slot_r + 1;
        const float x[7] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z};
#pragma unroll
        for (int i = 0; i < 7; ++i) W[j & 3][i] = (mask >> i) & 1 ? x[i] : 0.f;
      }
      if (j >= 3) {
        const int oy = oy0 + j - 3;
        if (oy < p.out_h) {
          const int w0 = (j + 1) & 3, w1 = (j + 2) & 3, w2 = (j + 3) & 3, w3 = j & 3;
          float acc[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float v = 0.f;
#pragma unroll
            for (int kx = 0; kx < 4; ++kx) v = fmaf(W[w0][e + kx], p.k2[0 + kx], v);
#pragma unroll
            for (int kx = 0; kx < 4; ++kx) v = fmaf(W[w1][e + kx], p.k2[4 + kx], v);
#pragma unroll
            for (int kx = 0; kx < 4; ++kx) v = fmaf(W[w2][e + kx], p.k2[8 + kx], v);
#pragma unroll
            for (int kx = 0; kx < 4; ++kx) v = fmaf(W[w3][e + kx], p.k2[12 + kx], v);
            acc[e] = v;
          }
          if constexpr (EPI) {
            const f4 nzv = nzr[j - 3];
            const float nn[4] = {nzv.x, nzv.y, nzv.z, nzv.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float v = __fadd_rn(__fadd_rn(acc[e], __fmul_rn(nw, nn[e])), bv);
              acc[e] = __fmul_rn(v > 0.f ? v : __fmul_rn(v, p.alpha), p.scale);
            }
          }
          const f4 o = {acc[0], acc[1], acc[2], acc[3]};
          f4* op = reinterpret_cast<f4*>(pout + (long long)oy * p.out_w + c0);
          if constexpr (NT) __builtin_nontemporal_store(o, op); else *op = o;
        }
      }
    });
  }
  wait_vm<0>();
}


// v2b (plain blur only): ring depth and burstiness experiment.  S slots; BURST = 1: one stage per step, S-1 rows ahead;
// BURST = 4: every 4th step issues the next 4 rows at once (S = 8).  Waits are exact in steady state and vmcnt(0)
// while the queue is still filling (k < S + 4).
template <int S, int BURST, int MINW, int AUX = 0, bool NOEXT = false>
__global__ __launch_bounds__(256, MINW) void blur_v2b(const V2P p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int SLOT = 320, D = BURST == 1 ? S - 1 : 4;
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned lb = xcd_remap(blockIdx.x, gridDim.x);
  const long long gw = (long long)lb * 4 + wv;
  if (gw >= p.total_waves) return;
  const int strip = (int)(gw % p.strips);
  const long long t = gw / p.strips;
  const int ty = (int)(t % p.tiles_y);
  const long long plane = t / p.tiles_y;
  float* ring = lds + wv * (S * SLOT);
  const int c0 = (strip * 64 + lane) * 4;
  const unsigned PARK = 0xFFFFFFF0u;
  const unsigned vmain = (c0 + 4 <= p.rs) ? (unsigned)c0 * 4u : PARK;
  const int xe = (strip + 1) * 256 + lane;
  const unsigned vext = (lane < 4 && xe < p.rs) ? (unsigned)xe * 4u : PARK;
  unsigned mask = 0;
#pragma unroll
  for (int i = 0; i < 7; ++i) {
    const int col = c0 + i - p.pad_x0;
    mask |= (col >= 0 && col < p.in_w) ? 1u << i : 0u;
  }
  const int oy0 = ty * p.th, oy_end = min(oy0 + p.th, p.out_h);
  const bool full = oy0 + p.th <= p.out_h;
  const int iy0 = oy0 - p.pad_y0;
  const int nsteps = p.th + 3;
  float* pin = const_cast<float*>(p.in + plane * p.ps);
  float* pout = p.out + plane * (long long)p.out_h * p.out_w;
  int slot_w = 0, slot_r = 0;
  auto issue = [&](int k) {
    const int iy = iy0 + k;
    const bool ok = k < nsteps && iy >= 0 && iy < p.in_h;
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(pin, 0, ok ? p.in_h * p.rs * 4 : 0, 0x00020000);
    const unsigned soff = ok ? (unsigned)iy * (unsigned)p.rs * 4u : 0u;
    dma_to_lds<16, AUX>(rsrc, ring + slot_w * SLOT, vmain, soff);
    if constexpr (!NOEXT) dma_to_lds<4, AUX>(rsrc, ring + slot_w * SLOT + 256, vext, soff);
    slot_w = slot_w + 1 == S ? 0 : slot_w + 1;
  };
  float W[4][7];
#pragma unroll 1
  for (int k = 0; k < D; ++k) issue(k);
  for (int kb = 0; kb < nsteps; kb += 4) {
    const bool steady = full && kb >= S + 4;
    static_for<4>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      const int k = kb + j;
      if (k < nsteps) {
        if constexpr (BURST == 1) issue(k + D);
        else if constexpr (j == 0) { issue(k + 4); issue(k + 5); issue(k + 6); issue(k + 7); }
        __builtin_amdgcn_sched_barrier(0);
        if (steady) wait_vm<(BURST == 1 ? (NOEXT ? 2 : 3) * D : 18 - j)>(); else wait_vm<0>();
        __builtin_amdgcn_sched_barrier(0);
        {
          const float* sp = ring + slot_r * SLOT + lane * 4;
          const f4 a = *reinterpret_cast<const f4*>(sp);
          const f4 b = *reinterpret_cast<const f4*>(sp + 4);
          slot_r = slot_r + 1 == S ? 0 : slot_r + 1;
          const float x[7] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z};
#pragma unroll
          for (int i = 0; i < 7; ++i) W[j][i] = (mask >> i) & 1 ? x[i] : 0.f;
        }
        const int oy = oy0 + k - 3;
        if (k >= 3 && oy < oy_end) {
          constexpr int w0 = (j + 1) & 3, w1 = (j + 2) & 3, w2 = (j + 3) & 3, w3 = j & 3;
          float acc[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float v = 0.f;
#pragma unroll
            for (int kx = 0; kx < 4; ++kx) v = fmaf(W[w0][e + kx], p.k2[0 + kx], v);
#pragma unroll
            for (int kx = 0; kx < 4; ++kx) v = fmaf(W[w1][e + kx], p.k2[4 + kx], v);
#pragma unroll
            for (int kx = 0; kx < 4; ++kx) v = fmaf(W[w2][e + kx], p.k2[8 + kx], v);
#pragma unroll
            for (int kx = 0; kx < 4; ++kx) v = fmaf(W[w3][e + kx], p.k2[12 + kx], v);
            acc[e] = v;
          }
          const f4 o = {acc[0], acc[1], acc[2], acc[3]};
          __builtin_nontemporal_store(o, reinterpret_cast<f4*>(pout + (long long)oy * p.out_w + c0));
        }
      }
    });
  }
  wait_vm<0>();
}


// v4: short tiles with the WHOLE input tile in flight at once, in registers: a wave owns TH output rows of a
// 256-column strip, issues its TH+3 row loads (one dwordx4 per lane per row, plus one scalar s_load_dwordx4 per row for
// the 3 columns past lane 63) and its TH noise rows back to back, and only then starts to filter.  No marching, no
// intra-wave pipeline: the bytes in flight per CU are (TH+3 [+TH]) KB x resident waves, several times what an LDS ring
// can hold, and what the chip touches at any moment is a compact window (like a linear copy).
template <bool EPI, int TH, bool REMAP, int MINW>
__global__ __launch_bounds__(256, MINW) void blur_v4(const V2P p) {
  constexpr int NR = TH + 3;
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned lb = REMAP ? xcd_remap(blockIdx.x, gridDim.x) : blockIdx.x;
  const long long gw = (long long)lb * 4 + wv;
  if (gw >= p.total_waves) return;
  const int strip = (int)(gw % p.strips);
  const long long t = gw / p.strips;
  const int ty = (int)(t % p.tiles_y);
  const long long plane = t / p.tiles_y;
  const int c0 = (strip * 64 + lane) * 4;
  unsigned mask = 0;
#pragma unroll
  for (int i = 0; i < 7; ++i) {
    const int col = c0 + i - p.pad_x0;
    mask |= (col >= 0 && col < p.in_w) ? 1u << i : 0u;
  }
  const int oy0 = ty * TH;
  const int iy0 = oy0 - p.pad_y0;
  const float* pin = p.in + plane * p.ps;
  float* pout = p.out + plane * (long long)p.out_h * p.out_w;
  const int xe = min((strip + 1) * 256, p.rs - 4);      // first position past this strip (clamped: masked anyway)

  f4 raw[NR]; f4 ext[NR]; f4 nz[TH];
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    const int iy = min(max(iy0 + r, 0), p.in_h - 1);     // clamped; rows outside the image are zeroed below
    const float* rp = pin + (long long)iy * p.rs;
    raw[r] = __builtin_nontemporal_load(reinterpret_cast<const f4*>(rp + min(c0, p.rs - 4)));
    ext[r] = *reinterpret_cast<const f4*>(rp + xe);     // wave-uniform address -> scalar load
  }
  float nw = 0.f, bv = 0.f;
  if (EPI) {
    nw = p.nwp[0]; bv = p.bias[(int)(plane % p.channels)];
#pragma unroll
    for (int r = 0; r < TH; ++r)
      nz[r] = *reinterpret_cast<const f4*>(p.noise + (long long)min(oy0 + r, p.out_h - 1) * p.out_w + c0);
  }
  float X[NR][7];
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    const bool rowok = iy0 + r >= 0 && iy0 + r < p.in_h;     // wave-uniform
    const float a[4] = {raw[r].x, raw[r].y, raw[r].z, raw[r].w};
    const float e[3] = {ext[r].x, ext[r].y, ext[r].z};
#pragma unroll
    for (int i = 0; i < 4; ++i) X[r][i] = (rowok && ((mask >> i) & 1)) ? a[i] : 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const float nb = __shfl(a[i], (lane + 1) & 63, 64);
      const float v = lane == 63 ? e[i] : nb;
      X[r][4 + i] = (rowok && ((mask >> (4 + i)) & 1)) ? v : 0.f;
    }
  }
#pragma unroll
  for (int r = 0; r < TH; ++r) {
    const int oy = oy0 + r;
    if (oy < p.out_h) {
      float acc[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float v = 0.f;
#pragma unroll
        for (int ky = 0; ky < 4; ++ky)
#pragma unroll
          for (int kx = 0; kx < 4; ++kx) v = fmaf(X[r + ky][e + kx], p.k2[ky * 4 + kx], v);
        acc[e] = v;
      }
      if constexpr (EPI) {
        const float nn[4] = {nz[r].x, nz[r].y, nz[r].z, nz[r].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float v = __fadd_rn(__fadd_rn(acc[e], __fmul_rn(nw, nn[e])), bv);
          acc[e] = __fmul_rn(v > 0.f ? v : __fmul_rn(v, p.alpha), p.scale);
        }
      }
      const f4 o = {acc[0], acc[1], acc[2], acc[3]};
      __builtin_nontemporal_store(o, reinterpret_cast<f4*>(pout + (long long)oy * p.out_w + c0));
    }
  }
}

// ---- the access shape without arithmetic (movement bound of this blocking)
template <int DEPTH, int STAG, bool REMAP = true>
__global__ __launch_bounds__(256) void rowmarch_copy(const float* __restrict__ in, float* __restrict__ out, int planes,
                                                     int RS, long long PS, int TH) {
  const int lane = threadIdx.x & 63;
  const long long gw = (long long)(REMAP ? xcd_remap(blockIdx.x, gridDim.x) : blockIdx.x) * 4 + (threadIdx.x >> 6);
  const int strips = 4, tiles_y = 1024 / TH;
  if (gw >= (long long)planes * strips * tiles_y) return;
  const int strip = (int)(gw % strips);
  const long long t = gw / strips;
  const int ty = (int)(t % tiles_y);
  const long long plane = t / tiles_y;
  const float* src = in + plane * PS + (long long)(ty * TH) * RS + strip * 256 + lane * 4;
  float* dst = out + plane * 1024LL * 1024 + (long long)(ty * TH) * 1024 + strip * 256 + lane * 4;
  f4 ring[DEPTH];
  // STAG: each tile starts its march at a different row (and wraps), so that concurrently resident tiles do not all
  // touch the same offset of their 256 KB regions at the same time
  const int r0 = STAG ? (int)((t * STAG) % TH) : 0;
#define ROW(r) ((r0 + (r)) & (TH - 1))
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) ring[d] = __builtin_nontemporal_load(reinterpret_cast<const f4*>(src + (long long)ROW(d) * RS));
  for (int r = 0; r < TH; r += DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      const f4 v = ring[d];
      if (r + d + DEPTH < TH) ring[d] = __builtin_nontemporal_load(reinterpret_cast<const f4*>(src + (long long)ROW(r + d + DEPTH) * RS));
      __builtin_nontemporal_store(v, reinterpret_cast<f4*>(dst + (long long)ROW(r + d) * 1024));
    }
  }
#undef ROW
}

// linear grid-stride copy (the best streaming pattern measured on this chip: tools/exp/hbm_copy.hip)
__global__ __launch_bounds__(256) void copy_k(const f4* __restrict__ in, f4* __restrict__ out, long long n) {
  long long i = ((long long)blockIdx.x * 256 * 4) + threadIdx.x;
  const long long stride = (long long)gridDim.x * 256 * 4;
  for (; i < n; i += stride) {
    f4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) if (i + u * 256 < n) v[u] = __builtin_nontemporal_load(in + i + u * 256);
#pragma unroll
    for (int u = 0; u < 4; ++u) if (i + u * 256 < n) __builtin_nontemporal_store(v[u], out + i + u * 256);
  }
}

// linear copy variants: REMAP = XCD-contiguous block order; PITCH = read 4 KB rows at the blur's row pitch
template <bool REMAP, bool PITCH>
__global__ __launch_bounds__(256) void copy2_k(const f4* __restrict__ in, f4* __restrict__ out, long long n, int rs4) {
  const unsigned b = REMAP ? xcd_remap(blockIdx.x, gridDim.x) : blockIdx.x;
  long long i = ((long long)b * 256 * 4) + threadIdx.x;
  const long long stride = (long long)gridDim.x * 256 * 4;
  for (; i < n; i += stride) {
    f4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long long j = i + u * 256;                       // row = j / 256 (256 f4 = 4 KB per row)
      const long long src = PITCH ? (j >> 8) * rs4 + (j & 255) : j;
      if (j < n) v[u] = __builtin_nontemporal_load(in + src);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) if (i + u * 256 < n) __builtin_nontemporal_store(v[u], out + i + u * 256);
  }
}

__global__ void fill_k(float* p, long long n, unsigned seed, int rs, int in_w, int off) {
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long stride = (long long)gridDim.x * 256;
  for (; i < n; i += stride) {
    unsigned h = (unsigned)(i * 2654435761u) ^ seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    float v = (float)(int)(h & 0xFFFF) / 32768.f - 1.f;
    if (rs > 0) { const int pos = (int)(i % rs); if (pos < off || pos >= in_w + off) v = __builtin_nanf(""); }   // padding must never matter
    p[i] = v;
  }
}

__global__ void diff_k(const float* a, const float* b, long long n, unsigned* maxbits, unsigned long long* nbad) {
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long stride = (long long)gridDim.x * 256;
  float m = 0.f; unsigned long long bad = 0;
  for (; i < n; i += stride) {
    const float d = fabsf(a[i] - b[i]);
    if (!(d == d)) ++bad; else m = fmaxf(m, d);
  }
  atomicMax(maxbits, __float_as_uint(m));
  if (bad) atomicAdd(nbad, bad);
}

template <typename F>
double time_us(F launch, int iters = 20) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 3; ++i) launch();
  hipDeviceSynchronize();
  hipEventRecord(a);
  for (int i = 0; i < iters; ++i) launch();
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  hipEventDestroy(a); hipEventDestroy(b);
  return ms / iters * 1e3;
}

typedef int (*blur_fn)(const float*, const float*, float*, int, int, int, int, long long, int, int, int, int, int, int, int,
                       const float*, const float*, const float*, int, float, float, void*);
typedef int (*ufd_strided_fn)(int, const void*, const void*, void*, int, int, int, int, long long, int, int, int, int, int,
                              int, int, int, int, int, int, int, void*);

int main(int argc, char** argv) {
  const char* libpath = argc > 1 ? argv[1] : "3d-fm-gan_amd/csrc/libfmgan_hip.so";
  void* lib = dlopen(libpath, RTLD_NOW);
  if (!lib) { printf("cannot load %s: %s\n", libpath, dlerror()); return 1; }
  blur_fn product = (blur_fn)dlsym(lib, "fmgan_blur_noise_bias_act_f32");
  ufd_strided_fn product_plain = (ufd_strided_fn)dlsym(lib, "fmgan_upfirdn2d_strided");
  if (!product || !product_plain) { printf("symbols missing\n"); return 1; }

  const int B = 8, C = 32, planes = B * C, in_h = 1025, in_w = 1025, out_h = 1024, out_w = 1024, pad0 = 1;
  const int rs = (in_w + pad0 + 31) / 32 * 32;
  const long long ps = (long long)in_h * rs;
  const long long n_in = planes * ps, n_out = (long long)planes * out_h * out_w;
  float *in, *o_ref, *o_new, *noise, *bias, *nwp, *kern;
  unsigned* maxbits; unsigned long long* nbad;
  hipMalloc(&in, n_in * 4); hipMalloc(&o_ref, n_out * 4); hipMalloc(&o_new, n_out * 4);
  hipMalloc(&noise, (long long)out_h * out_w * 4); hipMalloc(&bias, C * 4); hipMalloc(&nwp, 4); hipMalloc(&kern, 64);
  hipMalloc(&maxbits, 4); hipMalloc(&nbad, 8);
  hipLaunchKernelGGL(fill_k, dim3(8192), dim3(256), 0, 0, in, n_in, 0x1234u, rs, in_w, pad0);
  hipLaunchKernelGGL(fill_k, dim3(1024), dim3(256), 0, 0, noise, (long long)out_h * out_w, 0x77u, 0, 0, 0);
  hipLaunchKernelGGL(fill_k, dim3(1), dim3(256), 0, 0, bias, (long long)C, 0x99u, 0, 0, 0);
  const float k1[4] = {1.f, 3.f, 3.f, 1.f};
  float k2h[16]; for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) k2h[i * 4 + j] = k1[i] * k1[j] / 64.f * 4.f;
  const float nwh = 0.37f;
  hipMemcpy(kern, k2h, 64, hipMemcpyHostToDevice); hipMemcpy(nwp, &nwh, 4, hipMemcpyHostToDevice);
  hipDeviceSynchronize();

  V2P p{};
  p.in = in; p.noise = noise; p.nwp = nwp; p.bias = bias;
  p.planes = planes; p.channels = C; p.in_h = in_h; p.in_w = in_w; p.out_h = out_h; p.out_w = out_w; p.rs = rs;
  p.pad_x0 = pad0; p.pad_y0 = pad0; p.ps = ps; p.alpha = 0.2f; p.scale = sqrtf(2.f);
  for (int i = 0; i < 4; ++i) { p.kh[i] = k1[3 - i] / 4.f; p.kv[i] = k1[3 - i] / 4.f; }
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) p.k2[i * 4 + j] = k2h[(3 - i) * 4 + (3 - j)];
  const double gb = 4.0 * planes * ((double)in_h * in_w + (double)out_h * out_w) / 1e9;

  auto check = [&](const char* name) {
    hipMemset(maxbits, 0, 4); hipMemset(nbad, 0, 8);
    hipLaunchKernelGGL(diff_k, dim3(4096), dim3(256), 0, 0, o_ref, o_new, n_out, maxbits, nbad);
    unsigned mb; unsigned long long nb;
    hipMemcpy(&mb, maxbits, 4, hipMemcpyDeviceToHost); hipMemcpy(&nb, nbad, 8, hipMemcpyDeviceToHost);
    float m; memcpy(&m, &mb, 4);
    printf("    %-34s max|diff vs product| = %.3e, NaN elements %llu\n", name, m, nb);
  };

  const bool copy_only = argc > 2 && !strcmp(argv[2], "copy");
  for (int epi = 1; epi >= 0 && !copy_only; --epi) {
    printf("== %s\n", epi ? "fused epilogue (noise + bias + lrelu)" : "plain blur");
    auto run_product = [&](float* o) {
      if (epi) product(in + pad0, kern, o, B, C, in_h, in_w, ps, rs, 4, 4, pad0, 1, pad0, 1, noise, nwp, bias, 1, 0.2f, sqrtf(2.f), nullptr);
      else product_plain(0 /*f32*/, in + pad0, kern, o, planes, in_h, in_w, 1, ps, rs, 4, 4, 1, 1, 1, 1, pad0, 1, pad0, 1, -1, nullptr);
    };
    run_product(o_ref);
    hipDeviceSynchronize();
    double t = time_us([&] { run_product(o_ref); });
    printf("  product kernel                        : %7.1f us  %5.2f TB/s\n", t, gb / t * 1e3);
    for (int th : {8, 64}) {
      p.th = th; p.strips = (out_w + 255) / 256; p.tiles_y = (out_h + th - 1) / th;
      p.total_waves = (long long)planes * p.strips * p.tiles_y;
      const unsigned blocks = (unsigned)((p.total_waves + 3) / 4);
      const size_t ldsb = 4 * 4 * 320 * 4;
      p.out = o_new;
#define RUN(SEPV, EPIV, NTV, MINB, label) RUN2(SEPV, EPIV, NTV, MINB, true, label)
#define RUN2(SEPV, EPIV, NTV, MINB, RM, label)                                                                \
      {                                                                                                       \
        hipMemset(o_new, 0xFF, n_out * 4);                                                                    \
        auto l = [&] { hipLaunchKernelGGL((blur_v2<SEPV, EPIV, NTV, MINB, RM>), dim3(blocks), dim3(256), ldsb, 0, p); }; \
        l(); hipError_t e = hipDeviceSynchronize();                                                           \
        if (e != hipSuccess) { printf("  %s failed: %s\n", label, hipGetErrorString(e)); return 1; }          \
        double tt = time_us(l);                                                                               \
        printf("  %-30s TH %3d  : %7.1f us  %5.2f TB/s\n", label, th, tt, gb / tt * 1e3);              \
        check(label);                                                                                         \
      }
      if (epi) {
        RUN(false, true, true, 4, "v2 direct  nt occ4")
        RUN(true, true, true, 4, "v2 sep     nt occ4")
        RUN2(true, true, true, 6, false, "v2 sep     nt raw order")
        RUN(true, true, false, 6, "v2 sep        occ6")
      } else {
        RUN(false, false, true, 4, "v2 direct  nt occ4")
        RUN(true, false, true, 4, "v2 sep     nt occ4")
        RUN2(true, false, true, 6, false, "v2 sep     nt raw order")
      }
#undef RUN
#undef RUN2
    }

    {
      V3P q{};
      q.in = in; q.out = o_new; q.noise = noise; q.nwp = nwp; q.bias = bias;
      q.batch = B; q.channels = C; q.in_h = in_h; q.in_w = in_w; q.out_h = out_h; q.out_w = out_w; q.rs = rs;
      q.pad_x0 = pad0; q.pad_y0 = pad0; q.noise_batch = 1; q.ps = ps; q.alpha = 0.2f; q.scale = sqrtf(2.f);
      for (int i = 0; i < 16; ++i) q.k2[i] = p.k2[i];
      q.strips = (out_w + 255) / 256;
#define RUN3(EPIV, THV, SV, MINW, CGV, label)                                                                 \
      {                                                                                                       \
        q.cg = CGV; q.groups = C / CGV; q.tiles_y = (out_h + THV - 1) / THV;                                  \
        q.total_waves = (long long)B * q.groups * q.tiles_y * q.strips;                                       \
        const unsigned blocks = (unsigned)((q.total_waves + 3) / 4);                                          \
        const size_t ldsb = 4 * SV * 320 * 4;                                                                 \
        hipMemset(o_new, 0xFF, n_out * 4);                                                                    \
        auto l = [&] { hipLaunchKernelGGL((blur_v3<EPIV, true, THV, SV, MINW>), dim3(blocks), dim3(256), ldsb, 0, q); }; \
        l(); hipError_t e = hipDeviceSynchronize();                                                           \
        if (e != hipSuccess) { printf("  %s failed: %s\n", label, hipGetErrorString(e)); return 1; }          \
        double tt = time_us(l);                                                                               \
        printf("  %-30s blocks %5u : %7.1f us  %5.2f TB/s\n", label, blocks, tt, gb / tt * 1e3);             \
        check(label);                                                                                         \
      }
      if (epi) {
        RUN3(true, 8, 4, 4, 32, "v3 TH8 S4 cg32")
        RUN3(true, 8, 6, 4, 32, "v3 TH8 S6 cg32")
        RUN3(true, 8, 8, 4, 32, "v3 TH8 S8 cg32")
        RUN3(true, 8, 6, 4, 16, "v3 TH8 S6 cg16")
        RUN3(true, 8, 6, 4, 8, "v3 TH8 S6 cg8")
        RUN3(true, 16, 6, 4, 32, "v3 TH16 S6 cg32")
        RUN3(true, 16, 6, 4, 16, "v3 TH16 S6 cg16")
        RUN3(true, 16, 8, 4, 8, "v3 TH16 S8 cg8")
        RUN3(true, 4, 4, 4, 32, "v3 TH4 S4 cg32")
      } else {
        RUN3(false, 8, 6, 4, 32, "v3 TH8 S6 cg32")
        RUN3(false, 8, 6, 4, 16, "v3 TH8 S6 cg16")
        RUN3(false, 16, 6, 4, 16, "v3 TH16 S6 cg16")
      }
#undef RUN3
    }

    {
      p.out = o_new; p.strips = (out_w + 255) / 256;
#define RUN4(EPIV, THV, RM, MINW, label)                                                                      \
      {                                                                                                       \
        p.th = THV; p.tiles_y = (out_h + THV - 1) / THV;                                                      \
        p.total_waves = (long long)planes * p.strips * p.tiles_y;                                             \
        const unsigned blocks = (unsigned)((p.total_waves + 3) / 4);                                          \
        hipMemset(o_new, 0xFF, n_out * 4);                                                                    \
        auto l = [&] { hipLaunchKernelGGL((blur_v4<EPIV, THV, RM, MINW>), dim3(blocks), dim3(256), 0, 0, p); }; \
        l(); hipError_t e = hipDeviceSynchronize();                                                           \
        if (e != hipSuccess) { printf("  %s failed: %s\n", label, hipGetErrorString(e)); return 1; }          \
        double tt = time_us(l);                                                                               \
        printf("  %-30s blocks %6u : %7.1f us  %5.2f TB/s\n", label, blocks, tt, gb / tt * 1e3);             \
        check(label);                                                                                         \
      }
      if (epi) {
        RUN4(true, 4, true, 4, "v4 TH4 remap")
        RUN4(true, 8, true, 4, "v4 TH8 remap")
        RUN4(true, 8, false, 4, "v4 TH8 raw")
        RUN4(true, 8, true, 3, "v4 TH8 remap occ3")
        RUN4(true, 16, true, 2, "v4 TH16 remap occ2")
      } else {
        RUN4(false, 4, true, 4, "v4 TH4 remap")
        RUN4(false, 8, true, 4, "v4 TH8 remap")
        RUN4(false, 8, false, 4, "v4 TH8 raw")
        RUN4(false, 8, true, 6, "v4 TH8 remap occ6")
        RUN4(false, 16, true, 3, "v4 TH16 remap occ3")
      }
#undef RUN4
    }

    if (!epi) {
      for (int th : {64}) {
        p.th = th; p.strips = (out_w + 255) / 256; p.tiles_y = (out_h + th - 1) / th;
        p.total_waves = (long long)planes * p.strips * p.tiles_y;
        const unsigned blocks = (unsigned)((p.total_waves + 3) / 4);
        p.out = o_new;
#define RUNB(SV, BV, MINW, label) RUNB2(SV, BV, MINW, 0, false, label)
#define RUNB2(SV, BV, MINW, AUXV, NX, label)                                                                  \
        {                                                                                                     \
          const size_t ldsb = 4 * SV * 320 * 4;                                                               \
          hipMemset(o_new, 0xFF, n_out * 4);                                                                  \
          auto l = [&] { hipLaunchKernelGGL((blur_v2b<SV, BV, MINW, AUXV, NX>), dim3(blocks), dim3(256), ldsb, 0, p); };\
          l(); hipError_t e = hipDeviceSynchronize();                                                         \
          if (e != hipSuccess) { printf("  %s failed: %s\n", label, hipGetErrorString(e)); return 1; }        \
          double tt = time_us(l);                                                                             \
          printf("  %-30s TH %3d  : %7.1f us  %5.2f TB/s\n", label, th, tt, gb / tt * 1e3);                  \
          check(label);                                                                                       \
        }
        RUNB(4, 1, 4, "v2b S4 (3 ahead)")
        RUNB(6, 1, 4, "v2b S6 (5 ahead)")
        RUNB(8, 1, 4, "v2b S8 (7 ahead)")
        RUNB(12, 1, 2, "v2b S12 (11 ahead)")
        RUNB2(8, 1, 4, 2, false, "v2b S8 aux=2 (nt)")
        RUNB2(8, 1, 4, 1, false, "v2b S8 aux=1 (sc0)")
        RUNB2(8, 1, 4, 16, false, "v2b S8 aux=16 (sc1)")
        RUNB2(8, 1, 4, 0, true, "v2b S8 no halo DMA (wrong)")
        RUNB2(8, 1, 4, 2, true, "v2b S8 nt, no halo DMA (wrong)")
#undef RUNB
#undef RUNB2
      }
    }
  }
  {
    const long long PS = ps;
    for (int TH : {16, 64, 128}) {
      const unsigned blocks = (unsigned)(((long long)planes * 4 * (1024 / TH) + 3) / 4);
      double t2 = time_us([&] { hipLaunchKernelGGL((rowmarch_copy<2, 0>), dim3(blocks), dim3(256), 0, 0, (const float*)in, o_new, planes, rs, PS, TH); });
      double t4 = time_us([&] { hipLaunchKernelGGL((rowmarch_copy<4, 0>), dim3(blocks), dim3(256), 0, 0, (const float*)in, o_new, planes, rs, PS, TH); });
      double t8 = time_us([&] { hipLaunchKernelGGL((rowmarch_copy<8, 0>), dim3(blocks), dim3(256), 0, 0, (const float*)in, o_new, planes, rs, PS, TH); });
      double s1 = time_us([&] { hipLaunchKernelGGL((rowmarch_copy<4, 1>), dim3(blocks), dim3(256), 0, 0, (const float*)in, o_new, planes, rs, PS, TH); });
      double s7 = time_us([&] { hipLaunchKernelGGL((rowmarch_copy<4, 7>), dim3(blocks), dim3(256), 0, 0, (const float*)in, o_new, planes, rs, PS, TH); });
      double s13 = time_us([&] { hipLaunchKernelGGL((rowmarch_copy<8, 13>), dim3(blocks), dim3(256), 0, 0, (const float*)in, o_new, planes, rs, PS, TH); });
      printf("rowmarch copy staggered start rows TH %3d: depth4 stag1 %6.1f us | depth4 stag7 %6.1f us | depth8 stag13 %6.1f us\n", TH, s1, s7, s13);
      printf("rowmarch copy (no arithmetic) TH %3d: depth2 %6.1f us | depth4 %6.1f us | depth8 %6.1f us  (%.2f TB/s at depth8)\n", TH, t2, t4, t8,
             4.0 * planes * 2.0 * 1024 * 1024 / 1e9 / t8 * 1e3);
    }
  }
  {
    const long long n4 = n_out / 4;
    for (int blocks : {4096, 8192, 16384}) {
      double t = time_us([&] { hipLaunchKernelGGL(copy_k, dim3(blocks), dim3(256), 0, 0, (const f4*)in, (f4*)o_new, n4); });
      printf("linear nt copy of %.2f GB, %5d blocks: %6.1f us  %.2f TB/s   (same bytes as the blur: %.1f us)\n", 8.0 * n4 * 4 / 1e9, blocks, t,
             8.0 * n4 * 4 / 1e9 / t * 1e3, t * gb / (8.0 * n4 * 4 / 1e9));
    }
  }
  {
    const long long n4 = (long long)planes * 1024 * 256;   // rows of 4 KB
    for (int blocks : {8192, 16384}) {
      double a = time_us([&] { hipLaunchKernelGGL((copy2_k<true, false>), dim3(blocks), dim3(256), 0, 0, (const f4*)in, (f4*)o_new, n4, rs / 4); });
      double b = time_us([&] { hipLaunchKernelGGL((copy2_k<false, true>), dim3(blocks), dim3(256), 0, 0, (const f4*)in, (f4*)o_new, n4, rs / 4); });
      double c = time_us([&] { hipLaunchKernelGGL((copy2_k<true, true>), dim3(blocks), dim3(256), 0, 0, (const f4*)in, (f4*)o_new, n4, rs / 4); });
      printf("linear copy variants %5d blocks: xcd-remap %6.1f us | pitched rows %6.1f us | both %6.1f us\n", blocks, a, b, c);
    }
    for (int TH : {4, 8, 16, 32, 64}) {
      const unsigned blocks = (unsigned)(((long long)planes * 4 * (1024 / TH) + 3) / 4);
      double a = time_us([&] { hipLaunchKernelGGL((rowmarch_copy<4, 0, false>), dim3(blocks), dim3(256), 0, 0, (const float*)in, o_new, planes, rs, ps, TH); });
      double b = time_us([&] { hipLaunchKernelGGL((rowmarch_copy<4, 0, true>), dim3(blocks), dim3(256), 0, 0, (const float*)in, o_new, planes, rs, ps, TH); });
      double c = TH >= 8 ? time_us([&] { hipLaunchKernelGGL((rowmarch_copy<8, 0, false>), dim3(blocks), dim3(256), 0, 0, (const float*)in, o_new, planes, rs, ps, TH); }) : 0.0;
      double d = TH >= 16 ? time_us([&] { hipLaunchKernelGGL((rowmarch_copy<16, 0, false>), dim3(blocks), dim3(256), 0, 0, (const float*)in, o_new, planes, rs, ps, TH); }) : 0.0;
      printf("rowmarch copy TH %3d: depth4 raw %6.1f us | depth4 xcd-remap %6.1f us | depth8 raw %6.1f us | depth16 raw %6.1f us\n", TH, a, b, c, d);
    }
  }
  return 0;
}
