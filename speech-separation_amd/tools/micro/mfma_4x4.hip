// Diagnostic: can v_mfma_f32_4x4x1_16B_f32 (16 independent 4x4 rank-1 updates per instruction) carry the
// recurrence's product at the rate of v_mfma_f32_16x16x4_f32?  Both are 64 FLOP/clk/SIMD on paper; the 4x4 form
// lets a workgroup work on 4 or 8 batch rows at a time (two independent half-streams per workgroup).
// Geometry of the real kernels: 512-thread workgroups, one per CU, 112 A-operand registers per wave (the W_hh
// slice), B operand from LDS by ds_read_b128 (one read feeds 4 k values).
//   hipcc --offload-arch=gfx950 -O3 mfma_4x4.hip -o mfma_4x4
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int NQ = 28;  // ds_read_b128 per wave per (half-)step: 112 k values

// KIND 0: 16x16x4, 112 MFMAs per step (16 batch rows).  KIND 1: 4x4x1, 2 x 224 MFMAs per step (two 8-row halves,
// each 2 column blocks of 4 rows).  NACC independent accumulator chains per column block.
template <int KIND, int NACC>
__global__ __launch_bounds__(512, 2) void burn(const float* __restrict__ w, float* out, long long* cyc, int steps) {
  __shared__ __attribute__((aligned(16))) float img[2][NQ * 256];  // two halves x 28 KB
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  float wreg[4 * NQ];
#pragma unroll
  for (int i = 0; i < 4 * NQ; ++i) wreg[i] = w[(size_t)(wv * 4 * NQ + i) * 64 + lane];
  for (int i = tid; i < 2 * NQ * 256; i += 512) (&img[0][0])[i] = 1e-3f * (float)(i & 1023);
  __syncthreads();
  f32x4 acc[2][NACC];
#pragma unroll
  for (int n = 0; n < 2; ++n)
#pragma unroll
    for (int a = 0; a < NACC; ++a) acc[n][a] = f32x4{0.f, 0.f, 0.f, 0.f};
  const long long c0 = clock64();
  for (int s = 0; s < steps; ++s) {
    asm volatile("" ::: "memory");  // the image is re-read every step (as in the kernels, where it changes)
    if (KIND == 0) {
      const float* hp = &img[0][lane * 4];
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const float4 hb = *reinterpret_cast<const float4*>(hp + q * 256);
        acc[0][(4 * q + 0) % NACC] = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[4 * q + 0], hb.x, acc[0][(4 * q + 0) % NACC], 0, 0, 0);
        acc[0][(4 * q + 1) % NACC] = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[4 * q + 1], hb.y, acc[0][(4 * q + 1) % NACC], 0, 0, 0);
        acc[0][(4 * q + 2) % NACC] = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[4 * q + 2], hb.z, acc[0][(4 * q + 2) % NACC], 0, 0, 0);
        acc[0][(4 * q + 3) % NACC] = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[4 * q + 3], hb.w, acc[0][(4 * q + 3) % NACC], 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        asm volatile("" ::: "memory");
        // lane (j = lane>>2, i = lane&3): B value h[k][4 nb + i], the same for all 16 blocks j: 4 distinct addresses
        const float* hp = &img[half][(lane & 3) * 4];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
          const float4 h0 = *reinterpret_cast<const float4*>(hp + q * 256);
          const float4 h1 = *reinterpret_cast<const float4*>(hp + q * 256 + 16);
          acc[0][(4 * q + 0) % NACC] = __builtin_amdgcn_mfma_f32_4x4x1f32(wreg[4 * q + 0], h0.x, acc[0][(4 * q + 0) % NACC], 0, 0, 0);
          acc[1][(4 * q + 0) % NACC] = __builtin_amdgcn_mfma_f32_4x4x1f32(wreg[4 * q + 0], h1.x, acc[1][(4 * q + 0) % NACC], 0, 0, 0);
          acc[0][(4 * q + 1) % NACC] = __builtin_amdgcn_mfma_f32_4x4x1f32(wreg[4 * q + 1], h0.y, acc[0][(4 * q + 1) % NACC], 0, 0, 0);
          acc[1][(4 * q + 1) % NACC] = __builtin_amdgcn_mfma_f32_4x4x1f32(wreg[4 * q + 1], h1.y, acc[1][(4 * q + 1) % NACC], 0, 0, 0);
          acc[0][(4 * q + 2) % NACC] = __builtin_amdgcn_mfma_f32_4x4x1f32(wreg[4 * q + 2], h0.z, acc[0][(4 * q + 2) % NACC], 0, 0, 0);
          acc[1][(4 * q + 2) % NACC] = __builtin_amdgcn_mfma_f32_4x4x1f32(wreg[4 * q + 2], h1.z, acc[1][(4 * q + 2) % NACC], 0, 0, 0);
          acc[0][(4 * q + 3) % NACC] = __builtin_amdgcn_mfma_f32_4x4x1f32(wreg[4 * q + 3], h0.w, acc[0][(4 * q + 3) % NACC], 0, 0, 0);
          acc[1][(4 * q + 3) % NACC] = __builtin_amdgcn_mfma_f32_4x4x1f32(wreg[4 * q + 3], h1.w, acc[1][(4 * q + 3) % NACC], 0, 0, 0);
        }
      }
    }
  }
  const long long c1 = clock64();
  float sum = 0;
#pragma unroll
  for (int n = 0; n < 2; ++n)
#pragma unroll
    for (int a = 0; a < NACC; ++a) sum += acc[n][a][0] + acc[n][a][1] + acc[n][a][2] + acc[n][a][3];
  out[blockIdx.x * 512 + tid] = sum;
  if (tid == 0) cyc[blockIdx.x] = c1 - c0;
}

template <int KIND, int NACC>
void run(const char* name, const float* w, float* out, long long* cyc, int blocks, int steps) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((burn<KIND, NACC>), dim3(blocks), dim3(512), 0, 0, w, out, cyc, steps);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    long long c;
    hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    if (rep)
      printf("%-28s blocks=%3d  %7.3f us/step  %8.0f s_memtime ticks/step (ideal 7168 SIMD cycles)  %6.1f TFLOP/s chip-wide\n", name,
             blocks, 1e3 * ms / steps, (double)c / steps, 2.0 * 64 * 16 * 896 * blocks * (double)steps / ms / 1e9);
  }
}

int main() {
  float *w, *out;
  long long* cyc;
  hipMalloc(&w, 8 * 4 * NQ * 64 * 4);
  hipMalloc(&out, 256 * 512 * 4);
  hipMalloc(&cyc, 256 * 8);
  {
    static float hw[8 * 4 * NQ * 64];
    unsigned x = 12345u;
    for (int i = 0; i < 8 * 4 * NQ * 64; ++i) { x = x * 1664525u + 1013904223u; hw[i] = ((x >> 8) & 0xffff) / 65536.0f - 0.5f; }
    hipMemcpy(w, hw, sizeof(hw), hipMemcpyHostToDevice);
  }
  const int steps = 20000;
  for (int blocks : {1, 224}) {
    run<0, 2>("16x16x4, 2 chains", w, out, cyc, blocks, steps);
    run<0, 4>("16x16x4, 4 chains", w, out, cyc, blocks, steps);
    run<1, 1>("4x4x1_16B, 1 chain/blk", w, out, cyc, blocks, steps);
    run<1, 2>("4x4x1_16B, 2 chains/blk", w, out, cyc, blocks, steps);
    run<1, 4>("4x4x1_16B, 4 chains/blk", w, out, cyc, blocks, steps);
  }
  return 0;
}
