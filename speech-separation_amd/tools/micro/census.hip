// census.hip -- where does the dispatcher put the workgroups of a co-resident grid?  (speed only: no kernel here
// depends on the answer for correctness.)  Launches NB workgroups of NT threads that all stay resident for ~50 us
// and records XCC / SE / CU of each, then prints which block ids share a CU.
//   hipcc --offload-arch=gfx950 -O2 census.hip -o census && ./census 448 256 61440
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <map>
#include <vector>

__global__ void census(unsigned* out, long long hold_ticks) {
  extern __shared__ float lds[];
  if (threadIdx.x == 0) {
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);    // HW_REG_HW_ID
    const unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);  // HW_REG_XCC_ID
    out[2 * blockIdx.x] = hw;
    out[2 * blockIdx.x + 1] = xcc;
    lds[0] = (float)hw;
  }
  const long long t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < hold_ticks) __builtin_amdgcn_s_sleep(8);
}

int main(int argc, char** argv) {
  const int nb = argc > 1 ? atoi(argv[1]) : 448, nt = argc > 2 ? atoi(argv[2]) : 256;
  const int lds = argc > 3 ? atoi(argv[3]) : 61440;
  unsigned* d;
  hipMalloc(&d, nb * 8);
  hipFuncSetAttribute((const void*)census, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(census, dim3(nb), dim3(nt), lds, 0, d, 5000LL);  // 50 us at 100 MHz
    hipDeviceSynchronize();
  }
  std::vector<unsigned> h(2 * nb);
  hipMemcpy(h.data(), d, nb * 8, hipMemcpyDeviceToHost);
  std::map<unsigned, std::vector<int>> cu;
  int xcd_rr = 0;
  for (int b = 0; b < nb; ++b) {
    const unsigned hw = h[2 * b], xcc = h[2 * b + 1] & 15;
    const unsigned cuid = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
    cu[(xcc << 16) | (se << 8) | (sh << 4) | cuid].push_back(b);
    if ((int)xcc == b % 8) ++xcd_rr;
  }
  printf("grid %d x %d threads, %d B LDS: %zu distinct CUs; xcc == block %% 8 for %d blocks\n", nb, nt, lds, cu.size(), xcd_rr);
  std::map<int, int> hist, delta;
  for (auto& kv : cu) {
    hist[(int)kv.second.size()]++;
    if (kv.second.size() == 2) delta[kv.second[1] - kv.second[0]]++;
  }
  for (auto& kv : hist) printf("  CUs holding %d blocks: %d\n", kv.first, kv.second);
  for (auto& kv : delta) printf("  pair id distance %d: %d CUs\n", kv.first, kv.second);
  int shown = 0;
  for (auto& kv : cu) {
    if (shown++ >= 12) break;
    printf("  xcc %u se %u sh %u cu %u :", kv.first >> 16, (kv.first >> 8) & 255, (kv.first >> 4) & 15, kv.first & 15);
    for (int b : kv.second) printf(" %d", b);
    printf("\n");
  }
  return 0;
}
