// EqualLinear at inference batch sizes: out[b,n] = sum_k x[b,k] * w[n,k] (+ bias[n]).
//
// Replaces, for this path, `F.linear(input, weight * scale, bias=bias * lr_mul)` of the reference's EqualLinear
// (/root/reference/stylegan2.py:146-180) where it is the modulation of a ModulatedConv2d (stylegan2.py:226: one 512-wide
// linear per StyledConv / ToRGB and forward, batch = the pairs of one rank): the BLAS library serves that 8 x 512 x 512
// product with a 16 x 16 x 64 macro-tile GEMM in 19 us; as a matrix-vector product per sample it is one read of the weight.
// HBM/L2-bound: bytes = 4 * (N*K + B*K + B*N); at N = K = 512, B = 8 about 1 MB.
//
// One wave per (output feature n, sample b): lane l holds w[n, l + 64 j]; fixed summation order (lane-strided partial
// sums, then a butterfly) — bit-reproducible, independent of the batch size.
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void equal_linear_f32(const float* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ bias, float* __restrict__ out,
                                                        int batch, int n_out, int k_in) {
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= n_out) return;   // wave-uniform
  const float* wn = w + (long long)n * k_in;
  for (int b = blockIdx.y; b < batch; b += gridDim.y) {
    const float* xb = x + (long long)b * k_in;
    float acc = 0.f;
    for (int k = lane; k < k_in; k += 64) acc = fmaf(wn[k], xb[k], acc);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if (lane == 0) out[(long long)b * n_out + n] = bias ? acc + bias[n] : acc;
  }
}

}  // namespace

extern "C" int fmgan_equal_linear_f32(const float* x, const float* weight, const float* bias, float* out, int batch,
                                      int n_out, int k_in, void* stream) {
  if (batch < 0 || n_out <= 0 || k_in <= 0) return FMGAN_EINVAL;
  if (batch == 0) return FMGAN_OK;
  if (!x || !weight || !out) return FMGAN_EINVAL;
  hipLaunchKernelGGL(equal_linear_f32, dim3((n_out + 3) / 4, batch < 64 ? batch : 64), dim3(256), 0, (hipStream_t)stream,
                     x, weight, bias, out, batch, n_out, k_in);
  return fmgan_check_launch();
}
