// LDS-DMA probe (gfx950): what does `buffer_load_dword(x4) ... lds` write for lanes whose voffset is out of range?
// Expect: zeros (the kernel design relies on it for halo / padding slots).  Prints PASS/FAIL lines.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void* lds_ptr;
__global__ void probe(const float* p, int n, float* out) {
  extern __shared__ float smem[];
  for (int i = threadIdx.x; i < 2048; i += 256) smem[i] = 7.0f;      // sentinel
  __syncthreads();
  auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, n * 4, 0x00020000);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // 4-byte pieces: odd lanes parked out of range; soffset moves the window by 64 floats
  unsigned vo4 = (lane & 1) ? 0xFFFFFFF0u : (unsigned)((wave * 64 + lane) * 4);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr)(smem + wave * 64), 4, vo4, 64 * 4, 0, 0);
  // 16-byte pieces: lanes >= 48 parked
  unsigned vo16 = lane >= 48 ? 0xFFFFFFF0u : (unsigned)((wave * 64 + lane) * 16);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr)(smem + 1024 + wave * 256), 16, vo16, 0, 0, 0);
  __builtin_amdgcn_s_waitcnt(0x0f70);   // vmcnt(0)
  __syncthreads();
  for (int i = threadIdx.x; i < 2048; i += 256) out[i] = smem[i];
}
int main() {
  const int n = 4096;
  std::vector<float> h(n), o(2048);
  for (int i = 0; i < n; ++i) h[i] = 100.0f + i;
  float *d, *dout;
  hipMalloc(&d, n * 4); hipMalloc(&dout, 2048 * 4);
  hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(256), 2048 * 4, 0, d, n, dout);
  hipMemcpy(o.data(), dout, 2048 * 4, hipMemcpyDeviceToHost);
  int bad4 = 0, bad16 = 0, zero4 = 0, keep4 = 0, zero16 = 0, keep16 = 0;
  for (int i = 0; i < 256; ++i) {
    if (i & 1) { if (o[i] == 0.0f) ++zero4; else if (o[i] == 7.0f) ++keep4; else ++bad4; }
    else if (o[i] != 100.0f + i + 64) ++bad4;
  }
  for (int w = 0; w < 4; ++w) for (int l = 0; l < 64; ++l) for (int e = 0; e < 4; ++e) {
    float v = o[1024 + w * 256 + l * 4 + e];
    if (l >= 48) { if (v == 0.0f) ++zero16; else if (v == 7.0f) ++keep16; else ++bad16; }
    else if (v != 100.0f + (w * 64 + l) * 4 + e) ++bad16;
  }
  printf("b32: in-range wrong %d; out-of-range lanes: zero %d, untouched %d\n", bad4, zero4, keep4);
  printf("b128: in-range wrong %d; out-of-range lanes: zero %d, untouched %d\n", bad16, zero16, keep16);
  printf("%s\n", (bad4 == 0 && bad16 == 0 && keep4 == 0 && keep16 == 0) ? "PASS: OOB lanes write zeros" : "NOTE: see counts");
  return 0;
}
