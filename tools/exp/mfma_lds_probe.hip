// What caps the fp32 MFMA rate of a "ds_read operands -> v_mfma_f32_32x32x2_f32" loop on gfx950?
// Kernel P<RM, RNP, TAPS, WIDE>: the inner loop of modconv_mfma_f32 without any staging: per stage TAPS x RM A-operand reads
// and TAPS x RNP B-operand reads from a static LDS image, TAPS x RM x RNP MFMAs, software-pipelined one stage ahead.
// WIDE = 1 reads the RM A values / RNP B values of a tap with one ds_read_b64 (needs the image interleaved that way).
// Run with 1..3 blocks per CU (4 waves each) to see the two-waves-per-SIMD effect.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int RM, int RNP, int TAPS, int WIDE, int MINB>
__global__ __launch_bounds__(256, MINB) void P(float* out, int iters, unsigned long long* clk = nullptr) {
  extern __shared__ float smem[];
  const unsigned long long c0 = __builtin_readcyclecounter(), r0 = wall_clock64();
  for (int i = threadIdx.x; i < 8192; i += 256) smem[i] = 1e-3f * (i & 255);
  __syncthreads();
  const int lane = threadIdx.x & 63, l31 = lane & 31, kh = lane >> 5;
  f32x16 acc[RM][RNP];
  for (int m = 0; m < RM; ++m) for (int g = 0; g < RNP; ++g) for (int r = 0; r < 16; ++r) acc[m][g][r] = 0.f;
  struct Ops { float a[TAPS][RM]; float b[TAPS][RNP]; };
  const float* Ab = smem + kh * 1024 + l31 * (WIDE ? RM : 1);
  const float* Bb = smem + 4096 + kh * 1024 + l31 * (WIDE ? RNP : 1);
  auto fetch = [&](Ops& o, int st) {
#pragma unroll
    for (int t = 0; t < TAPS; ++t) {
      if constexpr (WIDE && RM == 2) {
        f32x2 v = *reinterpret_cast<const f32x2*>(Ab + (st & 3) * 256 + t * 64);
        o.a[t][0] = v.x; o.a[t][1] = v.y;
      } else {
#pragma unroll
        for (int m = 0; m < RM; ++m) o.a[t][m] = Ab[(st & 3) * 256 + t * 64 + m * 32];
      }
      if constexpr (WIDE && RNP == 2) {
        f32x2 v = *reinterpret_cast<const f32x2*>(Bb + (st & 3) * 256 + t * 64);
        o.b[t][0] = v.x; o.b[t][1] = v.y;
      } else {
#pragma unroll
        for (int g = 0; g < RNP; ++g) o.b[t][g] = Bb[(st & 3) * 256 + t * 64 + g * 32];
      }
    }
  };
  auto mma = [&](const Ops& o) {
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
      for (int m = 0; m < RM; ++m)
#pragma unroll
        for (int g = 0; g < RNP; ++g) acc[m][g] = __builtin_amdgcn_mfma_f32_32x32x2f32(o.a[t][m], o.b[t][g], acc[m][g], 0, 0, 0);
  };
  for (int it = 0; it < iters; ++it) {
    Ops cur, nxt;
    fetch(cur, 0);
#pragma unroll
    for (int st = 0; st < 12; ++st) {
      __builtin_amdgcn_sched_barrier(0);
      if (st + 1 < 12) fetch(nxt, st + 1);
      __builtin_amdgcn_sched_barrier(0);
      mma(cur);
      __builtin_amdgcn_sched_barrier(0);
      if (st + 1 < 12) cur = nxt;
    }
  }
  float s = 0.f;
  for (int m = 0; m < RM; ++m) for (int g = 0; g < RNP; ++g) for (int r = 0; r < 16; ++r) s += acc[m][g][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (clk && threadIdx.x == 0 && (blockIdx.x & 63) == 0) {
    atomicAdd(clk, __builtin_readcyclecounter() - c0);
    atomicAdd(clk + 1, wall_clock64() - r0);
  }
}

template <typename K>
void run(const char* name, K kern, int mfma_per_stage, float* out) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int bpc = 1; bpc <= 3; ++bpc) {
    const int blocks = 256 * bpc, iters = 400;
    kern<<<blocks, 256, 8192 * 4>>>(out, 10, nullptr);
    hipDeviceSynchronize();
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0); kern<<<blocks, 256, 8192 * 4>>>(out, iters, nullptr); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (ms < best) best = ms;
    }
    double flop = (double)blocks * 4 * iters * 12.0 * mfma_per_stage * 4096.0;
    unsigned long long* clk; hipMalloc(&clk, 16); hipMemset(clk, 0, 16);
    kern<<<blocks, 256, 8192 * 4>>>(out, iters, clk);
    unsigned long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost); hipFree(clk);
    printf("%-34s blocks/CU %d: %7.2f ms  %6.1f TFLOP/s  shader clock %.2f GHz\n", name, bpc, best, flop / best / 1e9,
           (double)h[0] / (double)(h[1] ? h[1] : 1) * 0.1);
  }
}

int main() {
  float* out; hipMalloc(&out, 1024 * 256 * 4);
  run("128x128 tile  RM2 RNP2 T3 b32", P<2, 2, 3, 0, 3>, 12, out);
  run("128x128 tile  RM2 RNP2 T3 b64", P<2, 2, 3, 1, 3>, 12, out);
  run("64x128 tile   RM2 RNP1 T3 b32", P<2, 1, 3, 0, 3>, 6, out);
  run("32x128 tile   RM1 RNP1 T3 b32", P<1, 1, 3, 0, 3>, 3, out);
  run("mode1-like    RM2 RNP1 T9 b32", P<2, 1, 9, 0, 3>, 18, out);
  run("wide          RM4 RNP2 T3 b32", P<4, 2, 3, 0, 2>, 24, out);
  run("wide          RM2 RNP4 T3 b32", P<2, 4, 3, 0, 2>, 24, out);
  return 0;
}
// (clock check: run with argument "clock" to print the shader clock held during the 128x128-like loop)
