// Shared helpers for the libfmgan_hip.so sources (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include "fmgan_hip.h"

#define FMGAN_WAVE 64
#define FMGAN_NUM_XCD 8

// Compute units of the current device, asked once per device (MI355X: 256); feeds the grid caps and the split-K
// model.  Without a device (host-logic calls in the CPU test-suite) the MI355X value is assumed, so host-only entry
// points such as fmgan_modconv2d_workspace_bytes() answer the same on both boxes.
static inline int fmgan_num_cu() {
  static int cached[16];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) {
    (void)hipGetLastError();
    return 256;
  }
  int v = __atomic_load_n(&cached[dev], __ATOMIC_RELAXED);
  if (v > 0) return v;
  if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) {
    (void)hipGetLastError();
    v = 256;
  }
  __atomic_store_n(&cached[dev], v, __ATOMIC_RELAXED);
  return v;
}
#define FMGAN_NUM_CU fmgan_num_cu()

static inline int fmgan_check_launch() {
  return hipGetLastError() == hipSuccess ? FMGAN_OK : FMGAN_ELAUNCH;
}

// Blocks are dealt round-robin over the 8 XCDs (b and b+8 share an L2).  Map the
// hardware block id to a logical id so each XCD walks one contiguous chunk of the
// logical grid: tiles that share halo rows / operand panels then share an L2.
// Bijective for any grid size (cdna_hip_programming.md §5 "XCD swizzle must be bijective").
__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nblk) {
  const unsigned q = nblk / FMGAN_NUM_XCD, r = nblk % FMGAN_NUM_XCD;
  const unsigned xcd = bid % FMGAN_NUM_XCD, slot = bid / FMGAN_NUM_XCD;
  const unsigned base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + slot;
}

template <typename T> struct AccT { using type = float; };
template <> struct AccT<double> { using type = double; };

template <typename T> __device__ __forceinline__ typename AccT<T>::type to_acc(T v) {
  return static_cast<typename AccT<T>::type>(v);
}
template <> __device__ __forceinline__ float to_acc<__half>(__half v) { return __half2float(v); }

template <typename T> __device__ __forceinline__ T from_acc(typename AccT<T>::type v) {
  return static_cast<T>(v);
}
template <> __device__ __forceinline__ __half from_acc<__half>(float v) { return __float2half(v); }

// 4-byte-aligned vector types: global_load/store_dwordx{2,4} need only dword alignment
// on gfx950, and rows of 2H+1 floats are never 16-byte aligned.
typedef float f32x4_u __attribute__((ext_vector_type(4), aligned(4)));
typedef float f32x2_u __attribute__((ext_vector_type(2), aligned(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
