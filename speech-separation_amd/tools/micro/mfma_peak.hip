// Diagnostic: sustained matrix-core rate of one MI355X with no memory traffic at all (the clock the chip
// actually holds under MFMA load bounds every GEMM).  hipcc --offload-arch=gfx950 -O3 mfma_peak.hip -o mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int KIND>
__global__ __launch_bounds__(256) void burn(float* out, int iters) {
  f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
  float x = threadIdx.x * 1e-3f, y = 1.0f + x;
  bf16x8 bx, by;
  for (int i = 0; i < 8; ++i) { bx[i] = (__bf16)x; by[i] = (__bf16)y; }
  for (int i = 0; i < iters; ++i) {
    if (KIND == 0) {
      a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a1, 0, 0, 0);
      a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a2, 0, 0, 0);
      a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a3, 0, 0, 0);
    } else {
      a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bx, by, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bx, by, a1, 0, 0, 0);
      a2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bx, by, a2, 0, 0, 0);
      a3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bx, by, a3, 0, 0, 0);
    }
  }
  float s = 0;
  for (int i = 0; i < 16; ++i) s += a0[i] + a1[i] + a2[i] + a3[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
  float* out;
  hipMalloc(&out, 4096 * 256 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int kind = 0; kind < 2; ++kind)
    for (int blocks : {256, 512, 1024}) {
      const int iters = kind ? 400000 : 50000;
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        if (kind == 0) hipLaunchKernelGGL(burn<0>, dim3(blocks), dim3(256), 0, 0, out, iters);
        else hipLaunchKernelGGL(burn<1>, dim3(blocks), dim3(256), 0, 0, out, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double flop = (double)blocks * 4 * iters * 4 * (kind ? 32.0 * 32 * 16 * 2 : 32.0 * 32 * 2 * 2);
        if (rep) printf("%s blocks=%4d  %8.2f ms  %8.1f TFLOP/s\n", kind ? "bf16 32x32x16" : "f32  32x32x2 ", blocks, ms, flop / ms / 1e9);
      }
    }
  return 0;
}
