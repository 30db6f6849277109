// Derived weights of a whole network, re-derived from the LIVE parameters in ONE launch per forward.
//
// The reference recomputes `weight * scale` / `bias * lr_mul` of every EqualLinear and the modulated weights of
// every ModulatedConv2d on every call (stylegan2.py:165-175, 257-262) — ~90 tiny elementwise launches per 1024^2
// forward.  Caching the derived tensors is fast but goes stale without any signal when the parameters are updated
// in place through `.data` (exactly what the reference's EMA `accumulate` does, train_3_encoder.py:195-200).  On an
// MI355X re-deriving everything costs ~120 MB of reads + 120 MB of writes = tens of microseconds: so the table of
// (source parameter -> derived buffer) is walked by ONE kernel at the start of each inference forward and there is
// nothing to invalidate.
//   kind 0  dst[k] = src[k] * scale                                  (EqualLinear weight*scale, bias*lr_mul)
//   kind 1  wt[i][t][o] = scale * W[o][i][t]   and   wsq[o][i] = sum_t W[o][i][t]^2
//           (MFMA A-operand layout of ModulatedConv2d + the demodulation sums; same arithmetic, same order as
//           modconv_weight_prep_f32 / modconv_wsq_f32 in modconv.hip -> identical bits)
// Kind 1 is an LDS-tiled transpose: a block owns 64 output channels x 16 input channels x ktaps; reads are runs of
// 16*ktaps contiguous floats per output channel (576 B for 3x3), writes are runs of 64 contiguous floats (256 B).
// (32 x 8 tiles — 288-byte reads, 128-byte writes — ran the refresh of Generator(1024) at ~0.8 TB/s.)
#include "common.h"

namespace {

constexpr int LW_TO = 64, LW_TI = 16, LW_MAXT = 9;
constexpr int LW_ELEMS0 = 2048;   // kind 0: elements per block

__global__ __launch_bounds__(256) void live_weights_f32(const fmgan_refresh_entry* __restrict__ table, int n_entries) {
  __shared__ float tile[LW_TO][LW_TI * LW_MAXT + 1];
  // entry of this block: the last one whose block_begin <= blockIdx.x (uniform binary search)
  int lo = 0, hi = n_entries - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (table[mid].block_begin <= blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const fmgan_refresh_entry e = table[lo];
  const unsigned lb = blockIdx.x - e.block_begin;
  const float* __restrict__ src = (const float*)e.src;
  float* __restrict__ dst = (float*)e.dst;
  if (e.kind == 0) {
    const long long base = (long long)lb * LW_ELEMS0;
#pragma unroll
    for (int k = 0; k < LW_ELEMS0 / 256; ++k) {
      const long long idx = base + threadIdx.x + 256 * k;
      if (idx < e.n) dst[idx] = src[idx] * e.scale;
    }
    return;
  }
  const int kt = e.ktaps, run = LW_TI * kt;
  const int i_tiles = (e.cin + LW_TI - 1) / LW_TI;
  const int o0 = (lb / i_tiles) * LW_TO, i0 = (lb % i_tiles) * LW_TI;
  const int i_n = min(LW_TI, e.cin - i0);
  for (int idx = threadIdx.x; idx < LW_TO * run; idx += 256) {
    const int o = idx / run, j = idx - o * run;
    float v = 0.f;
    if (o0 + o < e.cout && j < i_n * kt) v = src[((long long)(o0 + o) * e.cin + i0) * kt + j];
    tile[o][j] = v;
  }
  __syncthreads();
  for (int idx = threadIdx.x; idx < LW_TO * run; idx += 256) {
    const int j = idx / LW_TO, o = idx - j * LW_TO;       // j = i_local * kt + t
    if (o0 + o < e.cout && j < i_n * kt) dst[((long long)i0 * kt + j) * e.cout + o0 + o] = e.scale * tile[o][j];
  }
  float* __restrict__ wsq = (float*)e.dst2;
  if (wsq) {
    for (int idx = threadIdx.x; idx < LW_TO * LW_TI; idx += 256) {
      const int o = idx / LW_TI, i = idx % LW_TI;
      if (o0 + o < e.cout && i < i_n) {
        float q = 0.f;
        for (int t = 0; t < kt; ++t) { const float w = tile[o][i * kt + t]; q = fmaf(w, w, q); }
        wsq[(long long)(o0 + o) * e.cin + i0 + i] = q;
      }
    }
  }
}

}  // namespace

extern "C" long long fmgan_weight_refresh_blocks(int kind, int cout, int cin, int ktaps, long long n) {
  if (kind == 0) return n > 0 ? (n + LW_ELEMS0 - 1) / LW_ELEMS0 : -1;
  if (kind == 1) {
    if (cout <= 0 || cin <= 0 || ktaps <= 0 || ktaps > LW_MAXT) return -1;
    return (long long)((cout + LW_TO - 1) / LW_TO) * ((cin + LW_TI - 1) / LW_TI);
  }
  return -1;
}

extern "C" int fmgan_weight_refresh_f32(const fmgan_refresh_entry* table_dev, int n_entries, long long total_blocks,
                                        void* stream) {
  if (n_entries < 0 || total_blocks < 0) return FMGAN_EINVAL;
  if (n_entries == 0 || total_blocks == 0) return FMGAN_OK;
  if (!table_dev) return FMGAN_EINVAL;
  if (total_blocks > 0x7fffffffLL) return FMGAN_EOVERFLOW;
  hipLaunchKernelGGL(live_weights_f32, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream, table_dev,
                     n_entries);
  return fmgan_check_launch();
}
