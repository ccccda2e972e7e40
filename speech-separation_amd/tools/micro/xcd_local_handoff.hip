// Micro-benchmark: what would the per-step hand-off of the bf16 forward recurrence cost if a stream's workgroups sat on ONE XCD
// and exchanged h_t through that XCD's L2 instead of through write-through (`sc1`) stores and `sc1` loads?
//
// The shipped geometry (lstm.hip) at H = 896, B = 32: 4 streams (2 directions x 2 batch groups of 16 rows) x 56 workgroups of 16
// hidden units; a stream's 56 workgroups sit on an XCD pair, and because gfx950's eight L2s are not coherent with each other every
// payload store, flag store, poll and LDS-DMA pull carries `sc1`.  In bf16 the W_hh slice of 32 units fits the registers that 16
// units take in fp32, so a stream could be 28 workgroups x 32 units x 8 rows: 8 streams (2 directions x 4 batch groups of 8 rows),
// one per XCD (the dispatcher puts linear block b on XCD b % 8: checked here with HW_REG_XCC_ID, reported, and required for the L2
// form).  Within one XCD the L2 is the coherence point: plain stores (the L1 is write-through) and `sc0` loads (L1 bypassed).
//
// Forms timed (no MFMA, no cell math: the exchange chain alone; every step's payload depends on the previous step's gathered
// data; every spin is bounded at 0.2 s and raises an abort word all workgroups leave on; the gathered stamps are checked):
//   A  4 streams x 56 workgroups on XCD pairs, sc1 everywhere, 512 B published and 28 KB pulled per workgroup   (= shipped)
//   B  8 streams x 28 workgroups, one XCD each, sc1 everywhere, 512 B published and 14 KB pulled
//   C  geometry B, plain stores, sc0 polls and pulls   (measured: the polls never see the flags -- sc0 does not bypass the L1)
//   D  geometry B, plain stores, plain loads behind `buffer_inv sc0` (L1 invalidate): one per poll, one per wave before its pulls
//   E  the same with `buffer_inv sc1`
//   F  geometry B, plain stores (no write-through), sc1 polls and pulls
// A form that aborts (a bounded wait gave up) or gathers stale cells is reported as such and the next one runs.
// `hold`: the first poll is held back this long after the workgroup's own flag store (the shipped kernel: 0.8 us).
//
//   hipcc --offload-arch=gfx950 -O3 xcd_local_handoff.hip -o bin/xcd_local_handoff && bin/xcd_local_handoff [steps]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int NT = 512, NWG = 224;
constexpr long long SPIN = 20000000LL;  // 0.2 s of the 100 MHz wall clock

#define CHECK(x)                                                                   \
  do {                                                                             \
    hipError_t e__ = (x);                                                          \
    if (e__ != hipSuccess) {                                                       \
      fprintf(stderr, "%s:%d: %s\n", __FILE__, __LINE__, hipGetErrorString(e__)); \
      exit(2);                                                                     \
    }                                                                              \
  } while (0)

struct Args {
  unsigned* xbuf;     // [parity][stream][NUG x 512 B]
  unsigned* flags;    // [stream][NUG], one per 128-byte line
  unsigned* abort_w;
  unsigned* errors;   // gathered stamps that were not the step's
  unsigned* xcc;      // per workgroup: the XCD it ran on
  long long* ticks;   // per workgroup: wall-clock ticks of the timed steps
  int steps, hold_ticks;
};

// LDS-DMA of 1 KB (64 lanes x 16 B) from base + byte_off to LDS address lds_addr
template <int MODE>
__device__ __forceinline__ void dma_1k(const unsigned* base, unsigned byte_off, unsigned lds_addr, int lane) {
  if (MODE == 2 || MODE == 3)
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(byte_off + (unsigned)lane * 16u), "s"(base),
                 "s"(lds_addr)
                 : "memory");
  else if (MODE == 0 || MODE == 4)
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 sc1" ::"v"(byte_off + (unsigned)lane * 16u), "s"(base),
                 "s"(lds_addr)
                 : "memory");
  else
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 sc0" ::"v"(byte_off + (unsigned)lane * 16u), "s"(base),
                 "s"(lds_addr)
                 : "memory");
}
template <int MODE>
__device__ __forceinline__ void store16(unsigned* base, unsigned byte_off, u32x4 v) {
  if (MODE == 0)
    asm volatile("global_store_dwordx4 %0, %1, %2 sc1" ::"v"(byte_off), "v"(v), "s"(base) : "memory");
  else
    asm volatile("global_store_dwordx4 %0, %1, %2" ::"v"(byte_off), "v"(v), "s"(base) : "memory");
}
template <int MODE>
__device__ __forceinline__ void store_flag(unsigned* p, unsigned v) {
  if (MODE == 0)
    asm volatile("global_store_dword %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
  else
    asm volatile("global_store_dword %0, %1, off" ::"v"(p), "v"(v) : "memory");
}
template <int MODE>
__device__ __forceinline__ unsigned load_flag(const unsigned* p) {
  unsigned v;
  if (MODE == 2)
    asm volatile("buffer_inv sc0\n\tglobal_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  else if (MODE == 3)
    asm volatile("buffer_inv sc1\n\tglobal_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  else if (MODE == 0 || MODE == 4)
    asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  else
    asm volatile("global_load_dword %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}

template <int NUG, int MODE>
__device__ __forceinline__ bool wait_all(const unsigned* flags, unsigned target, unsigned* abort_w, int lane) {
  const long long t0 = wall_clock64();
  for (unsigned it = 0;; ++it) {
    const bool ok = lane >= NUG || load_flag<MODE>(flags + (size_t)lane * 32) >= target;
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

// NUG workgroups per stream; 224 / NUG streams.  NUG = 56: streams on XCD pairs (lstm.hip's map 1); NUG = 28: one XCD each.
template <int NUG, int MODE>
__global__ __launch_bounds__(NT, 2) void handoff(Args a) {
  constexpr int NSTREAM = NWG / NUG;
  constexpr int NPIECE = NUG / 2;                   // 1 KB pieces of a stream's image (512 B per workgroup)
  __shared__ __attribute__((aligned(1024))) unsigned land[NPIECE * 256];
  __shared__ int s_abort;
  __shared__ unsigned s_sum[8];
  const int L = (int)blockIdx.x, x = L & 7, j = L >> 3;
  const int stream = NUG == 56 ? (x >> 1) : x;
  const int ug = NUG == 56 ? (j * 2 + (x & 1)) : j;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned land_lds = __builtin_amdgcn_readfirstlane((int)(unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned*)land);
  if (tid == 0) {
    a.xcc[L] = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 15u;  // HW_REG_XCC_ID[3:0]
    s_abort = 0;
  }
  unsigned* const myflags = a.flags + (size_t)stream * NUG * 32;
  const size_t blk = (size_t)NUG * 128;             // dwords per (parity, stream) block
  unsigned carry = 0, bad = 0;
  __syncthreads();
  long long t_start = 0;
  const int warm = 20;
  for (int s = 0; s < a.steps + warm; ++s) {
    if (s == warm) t_start = wall_clock64();
    unsigned* const xw = a.xbuf + ((size_t)(s & 1) * NSTREAM + stream) * blk;
    // ---- publish 512 B: wave 0, 32 lanes x 16 B; dword 0 of every 16 B = the step's stamp (+ what the last gather fed back)
    if (w == 0 && lane < 32) {
      const u32x4 v = {(unsigned)(s + 1), carry, (unsigned)ug, (unsigned)lane};
      store16<MODE>(xw, (unsigned)(ug * 512 + lane * 16), v);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    long long t_flag = 0;
    if (tid == 0) {
      store_flag<MODE>(myflags + (size_t)ug * 32, (unsigned)(s + 1));
      t_flag = wall_clock64();
    }
    // ---- wait for the stream (one polling wave; first poll held back)
    if (w == 0) {
      t_flag = __shfl(t_flag, 0);
      while (wall_clock64() - t_flag < a.hold_ticks) __builtin_amdgcn_s_sleep(1);
      if (!wait_all<NUG, MODE>(myflags, (unsigned)(s + 1), a.abort_w, lane) && lane == 0) s_abort = 1;
    }
    __syncthreads();
    if (s_abort) return;
    // ---- gather the stream's image: NPIECE pieces of 1 KB dealt over the 8 waves (modes 2 / 3: behind an L1 invalidate)
    if (MODE == 2) asm volatile("buffer_inv sc0" ::: "memory");
    if (MODE == 3) asm volatile("buffer_inv sc1" ::: "memory");
    for (int p = w; p < NPIECE; p += 8) dma_1k<MODE>(xw, (unsigned)(p * 1024), land_lds + (unsigned)p * 1024u, lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // every 16-byte cell's stamp must be this step's; the sum of the ug fields feeds the next step's payload
    unsigned sum = 0;
    for (int c = tid; c < NPIECE * 64; c += NT) {
      bad += land[c * 4] != (unsigned)(s + 1);
      sum += land[c * 4 + 2];
    }
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    if (lane == 0) s_sum[w] = sum;
    __syncthreads();
    carry = 0;
    for (int k = 0; k < 8; ++k) carry += s_sum[k];
    __syncthreads();
  }
  const long long t1 = wall_clock64();
  if (tid == 0) a.ticks[L] = t1 - t_start;
  if (bad) atomicAdd(a.errors, bad);
}

// Form H (hybrid, the shipped geometry: 4 streams x 56 workgroups on XCD pairs).  A producer publishes its piece TWICE: plain
// stores into a buffer only its own XCD reads + a plain flag, and write-through (sc1) stores into a buffer the partner XCD reads +
// an sc1 flag behind them.  A consumer's waves 0-3 wait for the 28 producers of their OWN XCD (sc1 loads: L1 bypassed, served by
// the XCD's L2) and pull that half of the image; waves 4-7 wait for the partner XCD's 28 producers and pull the other half.  The
// halves meet at the step's end barrier.  Timed: when each half has landed, relative to the step's start -- the gap is what a
// kernel that multiplies the local half first (waves w and w + 4 share a SIMD) could hide.  KB = KB published per workgroup.
template <int KB>
__global__ __launch_bounds__(NT, 2) void hybrid(Args a, unsigned* bufL, unsigned* bufR, unsigned* flagL, unsigned* flagR,
                                                long long* t_half) {
  constexpr int HALF = 28, NPIECE = HALF * KB;      // 1 KB pieces of one half image
  __shared__ __attribute__((aligned(1024))) unsigned land[2 * NPIECE * 256];
  __shared__ int s_abort;
  __shared__ unsigned s_ready[2];
  __shared__ unsigned s_sum[8];
  const int L = (int)blockIdx.x, x = L & 7, j = L >> 3;
  const int stream = x >> 1, par = x & 1;           // this workgroup: producer j of parity par in its stream
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = w >> 2, wq = w & 3;              // half 0: the local one
  const unsigned land_lds = __builtin_amdgcn_readfirstlane((int)(unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned*)land);
  if (tid == 0) {
    a.xcc[L] = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 15u;
    s_abort = 0;
    s_ready[0] = s_ready[1] = 0;
  }
  const size_t blk = (size_t)HALF * KB * 256;       // dwords per (step parity, stream, XCD parity) block
  unsigned* const fL = flagL + ((size_t)stream * 2 + par) * HALF * 32;          // my XCD's flags (I raise [j], I poll all)
  unsigned* const fRmine = flagR + ((size_t)stream * 2 + par) * HALF * 32;      // what the partner polls: I raise [j]
  unsigned* const fRtheirs = flagR + ((size_t)stream * 2 + (par ^ 1)) * HALF * 32;
  unsigned carry = 0, bad = 0;
  long long accL = 0, accR = 0;
  __syncthreads();
  long long t_start = 0;
  const int warm = 20;
  for (int s = 0; s < a.steps + warm; ++s) {
    if (s == warm) {
      t_start = wall_clock64();
      accL = accR = 0;
    }
    const long long t0 = wall_clock64();
    unsigned* const xL = bufL + (((size_t)(s & 1) * 4 + stream) * 2 + par) * blk;          // my XCD's local buffer
    unsigned* const xRmine = bufR + (((size_t)(s & 1) * 4 + stream) * 2 + par) * blk;      // what I publish for the partner
    unsigned* const xRtheirs = bufR + (((size_t)(s & 1) * 4 + stream) * 2 + (par ^ 1)) * blk;
    // ---- publish KB KB twice: waves 0 .. KB-1, plain first, write-through second
    if (w < KB) {
      const u32x4 v = {(unsigned)(s + 1), carry, (unsigned)j, (unsigned)lane};
      store16<1>(xL, (unsigned)((j * KB + w) * 1024 + lane * 16), v);
      store16<0>(xRmine, (unsigned)((j * KB + w) * 1024 + lane * 16), v);
      asm volatile("s_waitcnt vmcnt(1)" ::: "memory");   // the plain store is through (stores retire in order)
    }
    __syncthreads();
    if (tid == 0) store_flag<1>(fL + (size_t)j * 32, (unsigned)(s + 1));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) store_flag<0>(fRmine + (size_t)j * 32, (unsigned)(s + 1));
    // ---- each half: its first wave polls 28 flags, the other three wait on an LDS word; then the half's pieces are pulled
    if (wq == 0) {
      const bool ok = wait_all<HALF, 0>(half == 0 ? fL : fRtheirs, (unsigned)(s + 1), a.abort_w, lane);
      if (lane == 0) {
        if (!ok) s_abort = 1;
        __hip_atomic_store(&s_ready[half], (unsigned)(s + 1), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    } else {
      while (__hip_atomic_load(&s_ready[half], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != (unsigned)(s + 1))
        __builtin_amdgcn_s_sleep(0);
    }
    if (!s_abort) {
      const unsigned* src = half == 0 ? xL : xRtheirs;
      for (int p = wq; p < NPIECE; p += 4) dma_1k<0>(src, (unsigned)(p * 1024), land_lds + (unsigned)(half * NPIECE + p) * 1024u, lane);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    const long long t_in = wall_clock64();
    if (wq == 0 && lane == 0) (half == 0 ? accL : accR) += t_in - t0;
    __syncthreads();
    if (s_abort) return;
    unsigned sum = 0;
    for (int c = tid; c < 2 * NPIECE * 64; c += NT) {
      bad += land[c * 4] != (unsigned)(s + 1);
      sum += land[c * 4 + 2];
    }
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    if (lane == 0) s_sum[w] = sum;
    __syncthreads();
    carry = 0;
    for (int k = 0; k < 8; ++k) carry += s_sum[k];
    __syncthreads();
  }
  const long long t1 = wall_clock64();
  if (tid == 0) {
    a.ticks[L] = t1 - t_start;
    t_half[2 * L] = accL;
  }
  if (tid == 256) t_half[2 * L + 1] = accR;
  if (bad) atomicAdd(a.errors, bad);
}

template <int KB>
void run_hybrid(Args a) {
  unsigned *bufL, *bufR, *flagL, *flagR;
  long long* t_half;
  const size_t bb = (size_t)2 * 4 * 2 * 28 * KB * 1024;
  CHECK(hipMalloc(&bufL, bb));
  CHECK(hipMalloc(&bufR, bb));
  CHECK(hipMemset(bufL, 0, bb));
  CHECK(hipMemset(bufR, 0, bb));
  CHECK(hipMalloc(&flagL, (size_t)NWG * 32 * 4));
  CHECK(hipMalloc(&flagR, (size_t)NWG * 32 * 4));
  CHECK(hipMemset(flagL, 0, (size_t)NWG * 32 * 4));
  CHECK(hipMemset(flagR, 0, (size_t)NWG * 32 * 4));
  CHECK(hipMalloc(&t_half, NWG * 16));
  CHECK(hipMemset(t_half, 0, NWG * 16));
  CHECK(hipMemset(a.abort_w, 0, 4));
  CHECK(hipMemset(a.errors, 0, 4));
  hipLaunchKernelGGL((hybrid<KB>), dim3(NWG), dim3(NT), 0, 0, a, bufL, bufR, flagL, flagR, t_half);
  CHECK(hipDeviceSynchronize());
  unsigned ab = 0, err = 0;
  CHECK(hipMemcpy(&ab, a.abort_w, 4, hipMemcpyDeviceToHost));
  CHECK(hipMemcpy(&err, a.errors, 4, hipMemcpyDeviceToHost));
  static long long h[NWG], th[2 * NWG];
  static unsigned xcc[NWG];
  CHECK(hipMemcpy(h, a.ticks, sizeof(h), hipMemcpyDeviceToHost));
  CHECK(hipMemcpy(th, t_half, sizeof(th), hipMemcpyDeviceToHost));
  CHECK(hipMemcpy(xcc, a.xcc, sizeof(xcc), hipMemcpyDeviceToHost));
  int off_xcd = 0;
  double s = 0, sl = 0, sr = 0;
  for (int i = 0; i < NWG; ++i) {
    off_xcd += xcc[i] != (unsigned)(i & 7);
    s += (double)h[i];
    sl += (double)th[2 * i];
    sr += (double)th[2 * i + 1];
  }
  printf("H hybrid, 4 x 56 on XCD pairs, %d KB per workgroup: %7.3f us per step; local half in at %.3f us, partner's half at %.3f us"
         " after the step's start  stale cells %u  off-XCD workgroups %d%s\n", KB, s / NWG / a.steps * 0.01, sl / NWG / a.steps * 0.01,
         sr / NWG / a.steps * 0.01, err, off_xcd, ab ? "  ABORTED (a bounded wait gave up)" : "");
  CHECK(hipFree(bufL));
  CHECK(hipFree(bufR));
  CHECK(hipFree(flagL));
  CHECK(hipFree(flagR));
  CHECK(hipFree(t_half));
}

template <int NUG, int MODE>
double run(Args a, const char* name, bool need_local) {
  CHECK(hipMemset(a.flags, 0, (size_t)NWG * 32 * 4));
  CHECK(hipMemset(a.abort_w, 0, 4));
  CHECK(hipMemset(a.errors, 0, 4));
  hipLaunchKernelGGL((handoff<NUG, MODE>), dim3(NWG), dim3(NT), 0, 0, a);
  CHECK(hipDeviceSynchronize());
  unsigned ab = 0, err = 0;
  CHECK(hipMemcpy(&ab, a.abort_w, 4, hipMemcpyDeviceToHost));
  CHECK(hipMemcpy(&err, a.errors, 4, hipMemcpyDeviceToHost));
  static long long h[NWG];
  static unsigned xcc[NWG];
  CHECK(hipMemcpy(h, a.ticks, sizeof(h), hipMemcpyDeviceToHost));
  CHECK(hipMemcpy(xcc, a.xcc, sizeof(xcc), hipMemcpyDeviceToHost));
  int off_xcd = 0;   // workgroups that did not run on XCD (block & 7)
  for (int i = 0; i < NWG; ++i) off_xcd += xcc[i] != (unsigned)(i & 7);
  double s = 0, mx = 0;
  for (int i = 0; i < NWG; ++i) {
    s += (double)h[i];
    if ((double)h[i] > mx) mx = (double)h[i];
  }
  const double us = s / NWG / a.steps * 0.01;
  printf("%-58s hold %.1f us  %7.3f us per step (slowest workgroup %.3f)  stale cells %u  off-XCD workgroups %d%s%s\n", name,
         a.hold_ticks * 0.01, us, mx / a.steps * 0.01, err, off_xcd, ab ? "  ABORTED (a bounded wait gave up)" : "",
         (need_local && off_xcd) ? "  (placement assumption broken: the L2 form is not valid on this run)" : "");
  return ab ? -1.0 : us;
}

int main(int argc, char** argv) {
  Args a;
  a.steps = argc > 1 ? atoi(argv[1]) : 2000;
  CHECK(hipMalloc(&a.xbuf, (size_t)2 * NWG * 512));
  CHECK(hipMemset(a.xbuf, 0, (size_t)2 * NWG * 512));
  CHECK(hipMalloc(&a.flags, (size_t)NWG * 32 * 4));
  CHECK(hipMalloc(&a.abort_w, 4));
  CHECK(hipMalloc(&a.errors, 4));
  CHECK(hipMalloc(&a.xcc, NWG * 4));
  CHECK(hipMalloc(&a.ticks, NWG * 8));
  int cus = 0;
  CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
  if (cus < NWG) {
    fprintf(stderr, "%d CUs < %d workgroups: the grid would not be co-resident\n", cus, NWG);
    return 3;
  }
  printf("forward hand-off chain alone (bf16 payload sizes), %d workgroups x %d threads, %d CUs, %d steps\n", NWG, NT, cus, a.steps);
  const int holds[3] = {0, 40, 80};
  for (int rep = 0; rep < 2; ++rep)
    for (int hi = 0; hi < 3; ++hi) {
      a.hold_ticks = holds[hi];
      run<56, 0>(a, "A 4 x 56 on XCD pairs, sc1 (shipped protocol)", false);
      run<28, 0>(a, "B 8 x 28, one XCD each, sc1", false);
      if (rep == 0 && hi == 0) run<28, 1>(a, "C 8 x 28, one XCD each, plain stores, sc0 loads", true);
      run<28, 2>(a, "D 8 x 28, one XCD each, plain stores, buffer_inv sc0 + plain loads", true);
      run<28, 3>(a, "E 8 x 28, one XCD each, plain stores, buffer_inv sc1 + plain loads", true);
      run<28, 4>(a, "F 8 x 28, one XCD each, plain stores, sc1 loads", true);
    }
  for (int rep = 0; rep < 2; ++rep) {
    run_hybrid<1>(a);
    run_hybrid<2>(a);
  }
  return 0;
}
