// Achievable HBM bandwidth on this box: copy / read-only / write-only streaming kernels over ~1 GiB buffers.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/exp/hbm_copy tools/exp/hbm_copy.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));

template <bool NTL, bool NTS, int UNR>
__global__ __launch_bounds__(256) void copy_k(const f4* __restrict__ in, f4* __restrict__ out, long long n) {
  long long i = ((long long)blockIdx.x * 256 * UNR) + threadIdx.x;
  const long long stride = (long long)gridDim.x * 256 * UNR;
  for (; i < n; i += stride) {
    f4 v[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const long long j = i + u * 256;
      if (j < n) v[u] = NTL ? __builtin_nontemporal_load(in + j) : in[j];
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const long long j = i + u * 256;
      if (j < n) { if (NTS) __builtin_nontemporal_store(v[u], out + j); else out[j] = v[u]; }
    }
  }
}

template <int UNR>
__global__ __launch_bounds__(256) void read_k(const f4* __restrict__ in, float* __restrict__ sink, long long n) {
  long long i = ((long long)blockIdx.x * 256 * UNR) + threadIdx.x;
  const long long stride = (long long)gridDim.x * 256 * UNR;
  f4 acc = {0, 0, 0, 0};
  for (; i < n; i += stride) {
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const long long j = i + u * 256;
      if (j < n) acc += __builtin_nontemporal_load(in + j);
    }
  }
  if (acc.x + acc.y + acc.z + acc.w == 12345.678f) sink[0] = acc.x;
}

template <int UNR>
__global__ __launch_bounds__(256) void write_k(f4* __restrict__ out, long long n) {
  long long i = ((long long)blockIdx.x * 256 * UNR) + threadIdx.x;
  const long long stride = (long long)gridDim.x * 256 * UNR;
  const f4 v = {1.f, 2.f, 3.f, 4.f};
  for (; i < n; i += stride) {
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const long long j = i + u * 256;
      if (j < n) __builtin_nontemporal_store(v, out + j);
    }
  }
}

// The blur's access shape without its arithmetic: a wave owns a 256-column strip of a [1025 x 1025] plane and marches
// TH rows; per row one 16-byte load per lane (row pitch RS floats, never 16-byte aligned for RS = 1025) with DEPTH row
// loads in flight, one 16-byte store per lane into a contiguous [1024 x 1024] plane.
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
template <int DEPTH, bool NT>
__global__ __launch_bounds__(256) void rowmarch_copy(const float* __restrict__ in, float* __restrict__ out, int planes,
                                                     int RS, long long PS, int TH) {
  const int lane = threadIdx.x & 63;
  const long long gw = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int strips = 4, tiles_y = 1024 / TH;
  if (gw >= (long long)planes * strips * tiles_y) return;
  const int strip = (int)(gw % strips);
  const long long t = gw / strips;
  const int ty = (int)(t % tiles_y);
  const long long plane = t / tiles_y;
  const float* src = in + plane * PS + (long long)(ty * TH) * RS + strip * 256 + lane * 4;
  float* dst = out + plane * 1024LL * 1024 + (long long)(ty * TH) * 1024 + strip * 256 + lane * 4;
  f4u ring[DEPTH];
#pragma unroll
  for (int d = 0; d < DEPTH; ++d)
    ring[d] = NT ? __builtin_nontemporal_load(reinterpret_cast<const f4u*>(src + (long long)d * RS))
                 : *reinterpret_cast<const f4u*>(src + (long long)d * RS);
  for (int r = 0; r < TH; r += DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      const f4u v = ring[d];
      if (r + d + DEPTH < TH)
        ring[d] = NT ? __builtin_nontemporal_load(reinterpret_cast<const f4u*>(src + (long long)(r + d + DEPTH) * RS))
                     : *reinterpret_cast<const f4u*>(src + (long long)(r + d + DEPTH) * RS);
      f4 w = {v.x, v.y, v.z, v.w};
      if (NT) __builtin_nontemporal_store(w, reinterpret_cast<f4*>(dst + (long long)(r + d) * 1024));
      else *reinterpret_cast<f4*>(dst + (long long)(r + d) * 1024) = w;
    }
  }
}

template <typename F>
double time_ms(F launch, int iters = 10) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 3; ++i) launch();
  hipDeviceSynchronize();
  hipEventRecord(a);
  for (int i = 0; i < iters; ++i) launch();
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  return ms / iters;
}

int main() {
  const long long n = (1LL << 30) / 16 * 1;          // 1 GiB per buffer in float4
  f4 *in, *out; float* sink;
  hipMalloc(&in, n * 16); hipMalloc(&out, n * 16); hipMalloc(&sink, 4);
  hipMemset(in, 0, n * 16); hipMemset(out, 0, n * 16);
  const double gb = 2.0 * n * 16 / 1e9;
  for (int blocks : {256 * 4, 256 * 8, 256 * 16, 256 * 32, 256 * 64, 1 << 20}) {
    double t;
    t = time_ms([&] { hipLaunchKernelGGL((copy_k<false, false, 1>), dim3(blocks), dim3(256), 0, 0, in, out, n); });
    printf("copy  plain        unr1 blocks %8d: %7.1f us %7.1f GB/s\n", blocks, t * 1e3, gb / t);
    t = time_ms([&] { hipLaunchKernelGGL((copy_k<true, true, 1>), dim3(blocks), dim3(256), 0, 0, in, out, n); });
    printf("copy  nt ld+st     unr1 blocks %8d: %7.1f us %7.1f GB/s\n", blocks, t * 1e3, gb / t);
    t = time_ms([&] { hipLaunchKernelGGL((copy_k<true, true, 4>), dim3(blocks), dim3(256), 0, 0, in, out, n); });
    printf("copy  nt ld+st     unr4 blocks %8d: %7.1f us %7.1f GB/s\n", blocks, t * 1e3, gb / t);
    t = time_ms([&] { hipLaunchKernelGGL((copy_k<false, true, 4>), dim3(blocks), dim3(256), 0, 0, in, out, n); });
    printf("copy  nt st only   unr4 blocks %8d: %7.1f us %7.1f GB/s\n", blocks, t * 1e3, gb / t);
    t = time_ms([&] { hipLaunchKernelGGL((read_k<4>), dim3(blocks), dim3(256), 0, 0, in, sink, n); });
    printf("read  nt           unr4 blocks %8d: %7.1f us %7.1f GB/s\n", blocks, t * 1e3, gb / 2 / t);
    t = time_ms([&] { hipLaunchKernelGGL((write_k<4>), dim3(blocks), dim3(256), 0, 0, out, n); });
    printf("write nt           unr4 blocks %8d: %7.1f us %7.1f GB/s\n", blocks, t * 1e3, gb / 2 / t);
  }
  {
    const int planes = 255;                              // 255 x 1025 x 1025 floats fit the 1 GiB input buffer
    const double gbr = 4.0 * planes * (1024.0 * 1024 + 1024.0 * 1024) / 1e9;
    for (int RS : {1025, 1056}) {
      const long long PS = (long long)1025 * RS;
      if ((long long)planes * PS * 4 > n * 16) continue;
      for (int TH : {16, 64, 256}) {
        const unsigned blocks = (unsigned)(((long long)planes * 4 * (1024 / TH) + 3) / 4);
        double t1 = time_ms([&] { hipLaunchKernelGGL((rowmarch_copy<2, true>), dim3(blocks), dim3(256), 0, 0, (const float*)in, (float*)out, planes, RS, PS, TH); });
        double t2 = time_ms([&] { hipLaunchKernelGGL((rowmarch_copy<4, true>), dim3(blocks), dim3(256), 0, 0, (const float*)in, (float*)out, planes, RS, PS, TH); });
        double t3 = time_ms([&] { hipLaunchKernelGGL((rowmarch_copy<8, true>), dim3(blocks), dim3(256), 0, 0, (const float*)in, (float*)out, planes, RS, PS, TH); });
        double t4 = time_ms([&] { hipLaunchKernelGGL((rowmarch_copy<4, false>), dim3(blocks), dim3(256), 0, 0, (const float*)in, (float*)out, planes, RS, PS, TH); });
        printf("rowmarch copy pitch %4d TH %3d: depth2 %6.1f us %4.2f TB/s | depth4 %6.1f us %4.2f | depth8 %6.1f us %4.2f | depth4 no-nt %6.1f us %4.2f\n",
               RS, TH, t1 * 1e3, gbr / t1, t2 * 1e3, gbr / t2, t3 * 1e3, gbr / t3, t4 * 1e3, gbr / t4);
      }
    }
  }
  double t = time_ms([&] { hipMemcpyAsync(out, in, n * 16, hipMemcpyDeviceToDevice, 0); });
  printf("hipMemcpy D2D: %7.1f us %7.1f GB/s\n", t * 1e3, gb / t);
  return 0;
}
