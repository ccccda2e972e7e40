// Diagnostic: how long do D back-to-back LDS-DMAs (global_load_lds_dwordx4 sc1, 1 KB each) of ONE wave take until
// s_waitcnt vmcnt(0), with W waves of a 512-thread workgroup pulling at the same time and 224 workgroups on the chip
// (the geometry of the recurrence's hand-off pull: 56 pieces of 1 KB per workgroup and step)?
//   hipcc --offload-arch=gfx950 -O3 dma_depth.hip -o dma_depth
#include <hip/hip_runtime.h>
#include <stdio.h>

__device__ __forceinline__ void dma_1k(const float* src, unsigned lds_addr) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off sc1" ::"v"(src), "s"(lds_addr) : "memory");
}

template <int D>
__global__ __launch_bounds__(512) void pull(const float* __restrict__ src, long long* out, int W, int reps) {
  __shared__ __attribute__((aligned(1024))) char img[64 * 1024];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const unsigned base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)img;
  // every workgroup reads the SAME 56 KB stream image of its "stream" (4 streams), as the recurrence's consumers do
  const float* s = src + (size_t)(blockIdx.x & 3) * 16384 + lane * 4;
  long long acc = 0;
  for (int r = 0; r < reps; ++r) {
    __syncthreads();
    const long long t0 = wall_clock64();
    if (w < W) {
#pragma unroll
      for (int i = 0; i < D; ++i) {
        const int p = (w + W * i) % 56;
        dma_1k(s + p * 256, __builtin_amdgcn_readfirstlane(base + p * 1024));
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    const long long t1 = wall_clock64();
    acc += t1 - t0;
  }
  if (lane == 0 && w == 0) out[blockIdx.x] = acc;
}

template <int D>
void run(const float* src, long long* out, int W, int blocks) {
  const int reps = 2000;
  hipLaunchKernelGGL((pull<D>), dim3(blocks), dim3(512), 0, 0, src, out, W, reps);
  (void)hipDeviceSynchronize();
  hipLaunchKernelGGL((pull<D>), dim3(blocks), dim3(512), 0, 0, src, out, W, reps);
  (void)hipDeviceSynchronize();
  long long h[256];
  (void)hipMemcpy(h, out, blocks * 8, hipMemcpyDeviceToHost);
  double s = 0;
  for (int i = 0; i < blocks; ++i) s += (double)h[i];
  printf("blocks=%3d  waves=%d  DMAs/wave=%2d  (%2d KB per workgroup)  %7.1f ns per pull (wave 0, 100 MHz clock)\n", blocks, W, D, W * D,
         s / blocks / reps * 10.0);
}

int main() {
  float* src;
  long long* out;
  (void)hipMalloc(&src, 4 * 16384 * 4 + 4096);
  (void)hipMemset(src, 0, 4 * 16384 * 4 + 4096);
  (void)hipMalloc(&out, 256 * 8);
  for (int blocks : {1, 224}) {
    run<1>(src, out, 8, blocks);
    run<2>(src, out, 8, blocks);
    run<4>(src, out, 8, blocks);
    run<7>(src, out, 8, blocks);
    run<14>(src, out, 4, blocks);
    run<14>(src, out, 8, blocks);
    run<28>(src, out, 2, blocks);
    run<7>(src, out, 4, blocks);
    run<7>(src, out, 1, blocks);
  }
  return 0;
}
