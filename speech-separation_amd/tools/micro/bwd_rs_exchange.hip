// Micro-benchmark: what does ONE time step of the backward recurrence's cross-workgroup exchange cost
// in its two possible forms, at the product's geometry (H = 896: 4 streams x 56 workgroups of 512 threads, one per CU,
// streams dealt to XCD pairs as lstm.hip's map 1 does), with the product's primitives (write-through `sc1` stores, one sc1
// flag per workgroup, one polling wave, LDS-DMA `global_load_lds ... sc1` pulls of 1 KB pieces)?
//
//   form A  "all-gather" (what lstm_bwd_kernel does): a workgroup publishes ITS dG tile (16 units x 4 gates x 16 batch rows =
//           4 KB, one 1 KB store per owner wave), raises its flag, waits for the 56 flags of its stream, pulls the stream's
//           whole dG image (56 x 4 KB = 224 pieces of 1 KB, 28 per wave) and reduces 8 per-wave partial tiles through LDS.
//   form R  "reduce-scatter": a workgroup multiplies its OWN dG tile with its 64 gate rows of W_hh (all H output units:
//           56 tiles of 16 units x 16 rows), publishes the 56 partial tiles (57 KB: 7 pieces of 1 KB per wave), raises its flag,
//           waits for the 56 flags, gathers the 56 partials of ITS OWN output tile (56 x 1 KB, 7 per wave) and sums them.
//
// No cell update and no bulk stores; the product is either left out (exchange only) or emulated with the real number of
// v_mfma_f32_16x16x4_f32 on register operands (112 per wave and step in both forms: `mfma=1`), placed where the form has
// it: A overlaps the pull (ring of sub-blocks in the kernel; here: issued behind the pulls, before their wait), R has it in
// front of the publish.  Every step's published values depend on the previous step's gathered sum (no step can be elided
// or reordered); every spin is bounded (0.2 s) and raises an abort word all workgroups leave on.
//
//   hipcc --offload-arch=gfx950 -O3 bwd_rs_exchange.hip -o bwd_rs_exchange && ./bwd_rs_exchange
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int NUG = 56, NSTREAM = 4, NWG = NUG * NSTREAM, NT = 512;
constexpr long long SPIN = 20000000LL;  // 0.2 s of the 100 MHz wall clock

#define CHECK(x)                                                                   \
  do {                                                                             \
    hipError_t e__ = (x);                                                          \
    if (e__ != hipSuccess) {                                                       \
      fprintf(stderr, "%s:%d: %s\n", __FILE__, __LINE__, hipGetErrorString(e__)); \
      exit(2);                                                                     \
    }                                                                              \
  } while (0)

__device__ __forceinline__ void dma_1k(const float* base, unsigned byte_off, unsigned lds_addr, int lane) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 sc1" ::"v"(byte_off + (unsigned)lane * 16u), "s"(base),
               "s"(lds_addr)
               : "memory");
}
__device__ __forceinline__ void store_1k_sc1(float* base, unsigned byte_off, f32x4 v, unsigned bytes, int lane) {
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(base, 0, (int)bytes, 0x00020000);
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs, byte_off + (unsigned)lane * 16u, 0, 16 /* sc1 */);
}

struct Args {
  float* xbuf;        // form A: [parity][stream][56 x 4 KB]; form R: [parity][stream][dst tile 56][src wg 56][1 KB]
  unsigned* flags;    // [stream][56], one per 128-byte line
  unsigned* abort_w;
  long long* ticks;   // per workgroup: wall-clock ticks of the timed steps
  float* sink;
  const float* wsrc;  // 112 floats per thread: the register-resident W slice
  int steps, mfma;
};

// stream / unit group of a workgroup: streams dealt to XCD pairs (linear block id L lands on XCD L % 8)
__device__ __forceinline__ void decode(int L, int& stream, int& ug) {
  const int x = L & 7, j = L >> 3;
  stream = x >> 1;
  ug = j * 2 + (x & 1);
}

__device__ __forceinline__ bool wait_all(const unsigned* flags, unsigned target, unsigned* abort_w, int lane) {
  const long long t0 = wall_clock64();
  for (unsigned it = 0;; ++it) {
    bool ok = lane >= NUG || __hip_atomic_load(flags + (size_t)lane * 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target;
    if (__all(ok)) return true;
    if ((it & 63u) == 63u) {
      if (__hip_atomic_load(abort_w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return false;
      if (wall_clock64() - t0 > SPIN) {
        if (lane == 0) __hip_atomic_store(abort_w, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return false;
      }
    }
    __builtin_amdgcn_s_sleep(1);
  }
}

// 112 MFMAs of one wave (the per-wave product of a step at H = 896 in either form), two accumulators
__device__ __forceinline__ f32x4 product(const float (&W)[112], f32x4 b, f32x4 acc) {
  f32x4 a0 = acc, a1 = acc;
#pragma unroll
  for (int i = 0; i < 112; i += 4) {
    a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(W[i + 0], b[0], a0, 0, 0, 0);
    a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(W[i + 1], b[1], a1, 0, 0, 0);
    a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(W[i + 2], b[2], a0, 0, 0, 0);
    a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(W[i + 3], b[3], a1, 0, 0, 0);
  }
  return a0 + a1;
}

template <bool RS>
__global__ __launch_bounds__(NT, 2) void exchange(Args a) {
  // form A: the stream image, 224 KB does not fit: pulled through per-wave slots of 7 KB, 4 rounds (the kernel's ring);
  // form R: 56 KB landing zone.  Both: 8 KB reduce scratch.
  __shared__ __attribute__((aligned(1024))) float land[56 * 256];
  __shared__ float red[8][256];
  __shared__ int s_abort;
  int stream, ug;
  decode((int)blockIdx.x, stream, ug);
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned land_lds = __builtin_amdgcn_readfirstlane((int)(unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)land);
  float W[112];
#pragma unroll
  for (int i = 0; i < 112; ++i) W[i] = a.wsrc[(size_t)i * NT + tid];
  unsigned* const myflags = a.flags + (size_t)stream * NUG * 32;
  const size_t blkA = (size_t)NUG * 1024;          // floats per (parity, stream) block, form A: 56 x 4 KB
  const size_t blkR = (size_t)NUG * NUG * 256;     // form R: 56 x 56 x 1 KB
  const size_t blk = RS ? blkR : blkA;
  if (tid == 0) s_abort = 0;
  f32x4 cell = {1.f + lane * 1e-3f, 0.5f, 0.25f, 0.125f};   // what the step publishes; fed by the previous step's sum
  __syncthreads();
  long long t_start = 0;
  const int warm = 20;
  for (int s = 0; s < a.steps + warm; ++s) {
    if (s == warm) t_start = wall_clock64();
    float* const xw = a.xbuf + ((size_t)(s & 1) * NSTREAM + stream) * blk;
    // ---- publish
    if (RS) {
      // own dG x own W rows -> 7 partial tiles per wave (the product sits in FRONT of the publish)
      f32x4 p = cell;
      if (a.mfma) p = product(W, cell, cell);
#pragma unroll
      for (int i = 0; i < 7; ++i) {
        const int dst_tile = w * 7 + i;
        f32x4 v = p;
        v[0] += (float)i;                           // (7 different tiles)
        store_1k_sc1(xw, (unsigned)((((size_t)dst_tile * NUG + ug) * 256) * 4), v, (unsigned)(blk * 4), lane);
      }
    } else if (w < 4) {
      store_1k_sc1(xw, (unsigned)((((size_t)ug * 4 + w) * 256) * 4), cell, (unsigned)(blk * 4), lane);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) __hip_atomic_store(myflags + (size_t)ug * 32, (unsigned)(s + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // ---- wait for the stream
    if (w == 0 && !wait_all(myflags, (unsigned)(s + 1), a.abort_w, lane) && lane == 0) s_abort = 1;
    __syncthreads();
    if (s_abort) return;
    // ---- gather
    float v = 0.f;
    if (RS) {
      const size_t mine = (size_t)ug * NUG * 256;   // my output tile: 56 contiguous 1 KB partials
#pragma unroll
      for (int i = 0; i < 7; ++i)
        dma_1k(xw, (unsigned)((mine + (size_t)(w * 7 + i) * 256) * 4), land_lds + (unsigned)(w * 7 + i) * 1024u, lane);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      // sum of 56 partials per cell: thread (half, c) takes 28 of them
      const int c = tid & 255, half = tid >> 8;
      float acc = 0.f;
#pragma unroll
      for (int k = 0; k < 28; ++k) acc += land[(half * 28 + k) * 256 + c];
      red[half][c] = acc;
      __syncthreads();
      v = red[0][c] + red[1][c];
    } else {
      // the stream's whole image: 224 pieces, 28 per wave, through a 7 KB slot per wave in 4 rounds of 7 (counted waits in
      // the kernel; here a full wait per round keeps the model simple and slightly pessimistic), product overlapped
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      for (int r = 0; r < 4; ++r) {
#pragma unroll
        for (int i = 0; i < 7; ++i)
          dma_1k(xw, (unsigned)(((size_t)(w * 28 + r * 7 + i) * 256) * 4), land_lds + (unsigned)(w * 7 + i) * 1024u, lane);
        if (a.mfma && r == 0) acc = product(W, cell, acc);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        acc[0] += land[(w * 7 + (r & 3)) * 256 + lane * 4];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) red[w][lane * 4 + q] = acc[q];
      __syncthreads();
      const int c = tid & 255;
#pragma unroll
      for (int k = 0; k < 8; ++k) v += red[k][c];
    }
    __syncthreads();
    cell[0] = 0.5f * cell[0] + 1e-6f * v;            // the next step's values depend on this step's sum
  }
  const long long t1 = wall_clock64();
  if (tid == 0) a.ticks[blockIdx.x] = t1 - t_start;
  if (cell[0] == 12345.678f) a.sink[tid] = cell[0];
}

template <bool RS>
double run(Args a, const char* name) {
  CHECK(hipMemset(a.flags, 0, (size_t)NSTREAM * NUG * 32 * 4));
  CHECK(hipMemset(a.abort_w, 0, 4));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  CHECK(hipEventRecord(e0, 0));
  hipLaunchKernelGGL((exchange<RS>), dim3(NWG), dim3(NT), 0, 0, a);
  CHECK(hipEventRecord(e1, 0));
  CHECK(hipDeviceSynchronize());
  unsigned ab = 0;
  CHECK(hipMemcpy(&ab, a.abort_w, 4, hipMemcpyDeviceToHost));
  float ms = 0.f;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  static long long h[NWG];
  CHECK(hipMemcpy(h, a.ticks, sizeof(h), hipMemcpyDeviceToHost));
  double s = 0, mx = 0;
  for (int i = 0; i < NWG; ++i) {
    s += (double)h[i];
    if ((double)h[i] > mx) mx = (double)h[i];
  }
  const double us = s / NWG / a.steps * 0.01;
  printf("%-44s mfma=%d  %7.3f us per step (mean over workgroups; slowest %.3f; kernel %.3f ms for %d + 20 steps)%s\n", name, a.mfma,
         us, mx / a.steps * 0.01, ms, a.steps, ab ? "  ABORTED (a bounded wait gave up)" : "");
  return ab ? -1.0 : us;
}

int main(int argc, char** argv) {
  Args a;
  a.steps = argc > 1 ? atoi(argv[1]) : 2000;
  const size_t xbytes = (size_t)2 * NSTREAM * NUG * NUG * 1024;   // form R's size covers form A's
  CHECK(hipMalloc(&a.xbuf, xbytes));
  CHECK(hipMemset(a.xbuf, 0, xbytes));
  CHECK(hipMalloc(&a.flags, (size_t)NSTREAM * NUG * 32 * 4));
  CHECK(hipMalloc(&a.abort_w, 4));
  CHECK(hipMalloc(&a.ticks, NWG * 8));
  CHECK(hipMalloc(&a.sink, NT * 4));
  {
    static float hw[112 * NT];
    for (int i = 0; i < 112 * NT; ++i) hw[i] = 1e-3f * (float)((i * 7) & 31) - 0.015f;
    float* d;
    CHECK(hipMalloc(&d, sizeof(hw)));
    CHECK(hipMemcpy(d, hw, sizeof(hw), hipMemcpyHostToDevice));
    a.wsrc = d;
  }
  int dev = 0, cus = 0;
  CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  if (cus < NWG) {
    fprintf(stderr, "%d CUs < %d workgroups: the grid would not be co-resident\n", cus, NWG);
    return 3;
  }
  printf("backward-recurrence exchange, H = 896 geometry: %d streams x %d workgroups x %d threads, %d CUs\n", NSTREAM, NUG, NT, cus);
  for (int rep = 0; rep < 2; ++rep) {
    for (int mfma = 0; mfma < 2; ++mfma) {
      a.mfma = mfma;
      if (run<false>(a, "A all-gather (4 KB out, 224 KB in)") < 0) return 1;
      if (run<true>(a, "R reduce-scatter (57 KB out, 56 KB in)") < 0) return 1;
    }
  }
  return 0;
}
