// sk_common.h -- shared helpers for libsepkern (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/sepkern.h"

#define SK_WAVE 64

// thread-local message for the last failing call (sk_last_error)
char* sk_errbuf();
int sk_fail(int code, const char* fmt, ...);
// build-option bits of the translation units that carry diagnostic switches (sk_build_flags)
unsigned sk_gemm_build_flags();
unsigned sk_lstm_build_flags();

#define SK_CHECK_ARG(cond, ...)                          \
  do {                                                   \
    if (!(cond)) return sk_fail(SK_EINVAL, __VA_ARGS__); \
  } while (0)

#define SK_CHECK_LAUNCH(what)                                                       \
  do {                                                                              \
    hipError_t e__ = hipGetLastError();                                             \
    if (e__ != hipSuccess) return sk_fail(SK_ELAUNCH, "%s: %s", what, hipGetErrorString(e__)); \
  } while (0)

#define SK_CHECK_HIP(expr)                                                                   \
  do {                                                                                       \
    hipError_t e__ = (expr);                                                                 \
    if (e__ != hipSuccess) return sk_fail(SK_ELAUNCH, "%s: %s", #expr, hipGetErrorString(e__)); \
  } while (0)

static inline int64_t sk_cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline size_t sk_align(size_t x, size_t a) { return (x + a - 1) / a * a; }

// ---- device helpers ------------------------------------------------------------------------
__device__ __forceinline__ float sk_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Block-wide sum for blockDim.x == 256 (4 waves); result valid in every thread.
__device__ __forceinline__ float sk_block_sum256(float v, float* red /* >= 4 floats of LDS */) {
  v = sk_wave_sum(v);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}

__device__ __forceinline__ float sk_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }
