// gemm.hip -- fp32-in / fp32-accumulate matrix-core GEMM for gfx950.
//
// C[M,N] = act(opA(A) * opB(B) + bias + (accumulate ? C : 0)).  Used for everything on the uPIT
// path that is a plain matrix product: the LSTM input projections for all T*B rows at once,
// nn.Linear, and their dgrad / wgrad passes (reference archs/uPIT.py:132,141 via torch).
// Two families of kernels, both fp32 products with fp32 accumulation (results match an fp32 reference to
// rounding-order differences only):
//   * fp32 MFMA (v_mfma_f32_32x32x2_f32: bit for bit a k-ordered fmaf chain): gemm_f32_kernel (register-staged, any
//     alignment), _dma / _dma256 / _streamk (LDS-DMA staging);
//   * SPLIT products on the bf16 matrix pipe (r05, the default wherever operands are aligned): both operands cut exactly into
//     three bf16 pieces, six exact piece products per element pair -- gemm_f32_kernel_split3, gemm_f32_kernel_planes;
//     the bf16 pipe is 16x the fp32 one, so six products still leave 2.7x its rate.
// The fp32-MFMA kernel's design notes follow.
//
// Tiling: 128x128 block tile, 4 waves as 2x2, each wave 64x64 = 2x2 MFMA tiles of 32x32
// (64 accumulator VGPRs); K step 16, LDS double-buffered, operands kept k-major in LDS so the
// per-lane A/B fragment reads are conflict-free ds_read_b32 of 32 consecutive dwords.  BK = 16 keeps a
// block at 33 KB of LDS and 142 registers, so THREE blocks (3 waves per SIMD) are resident per CU and fill
// each other's barrier / staging gaps: measured 115 / 111 / 117 TFLOP/s on the large NT / NN / TN shapes
// against 110 / 107 / 111 with BK = 32 (two blocks per CU).
#include "sk_common.h"
#include <type_traits>

namespace {
thread_local int t_last_kernel = 0;  // which kernel this thread's last launch took: sk_gemm_last_kernel()
}

namespace {

constexpr int BM = 128, BN = 128, BK = 16;
constexpr int GROUP_M = 8;  // tile rows per L2 working-set group (see the kernels' tile order)
// LDS row stride (floats) of a k-major operand image [BK][LD]: 132 keeps float4 rows 16-byte aligned for
// operands that are k-major in memory; 129 makes the 4x4 transposing stash of [dim][K] operands
// conflict-free (bank = 4q + r + row).  Fragment reads (32 consecutive floats) are conflict-free for both.
constexpr int LD_K = 132, LD_T = 129;
constexpr int PIECES = BK * BM / 4 / 256;  // float4 pieces per thread per operand tile
typedef float f32x16 __attribute__((ext_vector_type(16)));

struct GemmArgs {
  const float* A;
  const float* B;
  float* C;
  const float* bias;
  int M, N, K, lda, ldb, ldc;
  int accumulate, act, vecA, vecB;
  int tilesN;
  int splitk, kchunk;  // split-K: blockIdx.y = slice, K range [y*kchunk, min(K,(y+1)*kchunk)), partials to slabs
  float* slabs;        // (batch, splitk, M, N) dense fp32 partial products when splitk > 1
  unsigned* counters;  // one ticket counter per (batch, tile), zero between launches: the fp32 kernels reduce the slabs
                       // themselves (finish_splitk); nullptr = the separate splitk_reduce_kernel does
  int64_t sA, sB, sC, sbias;
  int sk_tiles, sk_full;  // stream-K kernel: tiles of the product, data-parallel rounds before the stream-K region
  int flip_q;             // split-product kernels: quarter period of the sign phases in K steps of BK (0: none), see SignPhase
  // gemm_f32_kernel_pl3 (operands that arrive split): the three bf16 planes of each operand, K-major ([K][dim], lda / ldb
  // elements between k rows), planeA / planeB elements from one plane to the next; A / B above are unused
  const __bf16* Apl;
  const __bf16* Bpl;
  int64_t planeA, planeB;
};
constexpr size_t COUNTER_BYTES = 65536;  // head of a split-K workspace: 16384 counters

// Fetch this thread's two float4 pieces of an operand tile.  Two branch-free forms, selected by a
// block-uniform condition (per-element "load or zero" branches make hipcc wait vmcnt(0) per load):
//  fast : tile fully inside the matrix and 16-byte aligned -> unconditional float4 loads
//  slow : every element loaded from a CLAMPED in-range address and zeroed by a select
//  KMAJOR == false: operand stored [tile dim (M or N)][K]  (transposed into k-major LDS by stash)
//  KMAJOR == true : operand stored [K][tile dim]            (already k-major)
//  VEC == false (operands whose rows are not 16-byte aligned, e.g. the F = 257-edged ones): the fast form is
//  four unconditional dword loads instead of one float4
template <bool KMAJOR, bool VEC = true>
__device__ __forceinline__ void fetch(const float* __restrict__ P, int ld, int dim0, int dimLimit, int k0, int K,
                                      bool fast, int tid, float4 (&r)[PIECES]) {
#pragma unroll
  for (int i = 0; i < PIECES; ++i) {
    const int idx = tid + 256 * i;
    int row, col, rowLimit, colLimit;  // element (row, col..col+3) of the stored matrix
    if (KMAJOR) {
      row = k0 + (idx >> 5);
      col = dim0 + (idx & 31) * 4;
      rowLimit = K;
      colLimit = dimLimit;
    } else {
      row = dim0 + idx / (BK / 4);
      col = k0 + (idx % (BK / 4)) * 4;
      rowLimit = dimLimit;
      colLimit = K;
    }
    if (fast) {
      const float* q = P + (int64_t)row * ld + col;
      if (VEC)
        r[i] = *reinterpret_cast<const float4*>(q);
      else
        r[i] = make_float4(q[0], q[1], q[2], q[3]);
    } else {
      const bool rok = row < rowLimit;
      const float* q = P + (int64_t)min(row, rowLimit - 1) * ld;
      const int cmax = colLimit - 1;
      const float v0 = q[min(col + 0, cmax)], v1 = q[min(col + 1, cmax)];
      const float v2 = q[min(col + 2, cmax)], v3 = q[min(col + 3, cmax)];
      r[i].x = (rok && col + 0 < colLimit) ? v0 : 0.f;
      r[i].y = (rok && col + 1 < colLimit) ? v1 : 0.f;
      r[i].z = (rok && col + 2 < colLimit) ? v2 : 0.f;
      r[i].w = (rok && col + 3 < colLimit) ? v3 : 0.f;
    }
  }
}

template <bool KMAJOR, int LD>
__device__ __forceinline__ void stash(float (*S)[LD], int tid, const float4 (&r)[PIECES]) {
#pragma unroll
  for (int i = 0; i < PIECES; ++i) {
    const int idx = tid + 256 * i;
    if (KMAJOR) {
      const int kr = idx >> 5, c4 = (idx & 31) * 4;
      *reinterpret_cast<float4*>(&S[kr][c4]) = r[i];
    } else {
      const int row = idx / (BK / 4), k4 = (idx % (BK / 4)) * 4;
      S[k4 + 0][row] = r[i].x;
      S[k4 + 1][row] = r[i].y;
      S[k4 + 2][row] = r[i].z;
      S[k4 + 3][row] = r[i].w;
    }
  }
}

// Epilogue of one wave's 64 x 64 sub-tile at (row0, col0).  C/D layout of the 32x32 MFMA: col = lane&31,
// row = (r&3) + 8*(r>>2) + 4*(lane>>5).
__device__ __forceinline__ void store_tile(const GemmArgs& g, const f32x16 (&acc)[2][2], float* C, int ldc, const float* bias,
                                           bool partial, int row0, int col0, int lane) {
  const int kh = lane >> 5, l31 = lane & 31;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = col0 + j * 32 + l31;
      if (col >= g.N) continue;
      const float bv = bias ? bias[col] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = row0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
        if (row < g.M) {
          float* cp = C + (int64_t)row * ldc + col;
          float v = acc[i][j][r] + bv;
          if (!partial) {
            if (g.accumulate) v += *cp;
            if (g.act == 1) v = sk_sigmoid(v);
            *cp = v;
          } else {
            __hip_atomic_store(cp, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // write-through: read by another XCD
          }
        }
      }
    }
}

// Split-K epilogue inside the product kernel (no second launch): every slice has written its partial tile to its
// slab with write-through stores; when those have landed the block draws a ticket from the tile's counter, and the
// block that draws the LAST ticket adds the tile's slabs in slice order (deterministic: the same sums as
// splitk_reduce_kernel) and applies bias / accumulate / act.  It also puts the counter back to zero, so a workspace
// that starts zeroed stays usable launch after launch.  Cross-XCD visibility as in the recurrence kernels: sc1 stores,
// vmcnt(0) before the ticket, sc1 loads after it.
__device__ __forceinline__ void finish_splitk(const GemmArgs& g, int z, int row0, int col0, int tid) {
  __shared__ int s_last;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) {
    unsigned* cnt = g.counters + (size_t)z * gridDim.x + blockIdx.x;
    const unsigned old = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int last = old == (unsigned)g.splitk - 1u;
    if (last) __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_last = last;
  }
  __syncthreads();
  if (!s_last) return;
  const int lane = tid & 63, kh = lane >> 5, l31 = lane & 31;
  const int64_t mn = (int64_t)g.M * g.N;
  const float* sl = g.slabs + (int64_t)z * g.splitk * mn;
  float* C = g.C + z * g.sC;
  const float* bias = g.bias ? g.bias + z * g.sbias : nullptr;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = col0 + j * 32 + l31;
      if (col >= g.N) continue;
      const float bv = bias ? bias[col] : 0.f;
      float a[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = min(row0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh, g.M - 1);
        a[r] = __hip_atomic_load(sl + (int64_t)row * g.N + col, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      for (int k = 1; k < g.splitk; ++k) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = min(row0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh, g.M - 1);
          a[r] += __hip_atomic_load(sl + k * mn + (int64_t)row * g.N + col, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = row0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
        if (row < g.M) {
          float* cp = C + (int64_t)row * g.ldc + col;
          float x = a[r] + bv;
          if (g.accumulate) x += *cp;
          if (g.act == 1) x = sk_sigmoid(x);
          *cp = x;
        }
      }
    }
}

// VEC == false: the variant for operands with unaligned rows (both operands then use dword loads).
template <bool TA, bool TB, bool VEC = true>
__global__ __launch_bounds__(256, 4) void gemm_f32_kernel(GemmArgs g) {
  constexpr int LDA = TA ? LD_K : LD_T, LDB = TB ? LD_T : LD_K;
  __shared__ __attribute__((aligned(16))) float As[2][BK][LDA];
  __shared__ __attribute__((aligned(16))) float Bs[2][BK][LDB];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  // XCD-aware tile order: blocks b and b+8 share an XCD (and its L2), so give each XCD a
  // contiguous run of tiles (n fastest): neighbours in a run share their A panel, runs share B.
  int tile = blockIdx.x;
  {
    const int nt = gridDim.x, q = nt >> 3, rem = nt & 7, x = tile & 7, j = tile >> 3;
    tile = (x < rem ? x * (q + 1) : rem * (q + 1) + (x - rem) * q) + j;
  }
  // within a run, walk groups of GROUP_M tile rows column by column: the ~128 tiles an XCD has in flight
  // then cover about GROUP_M x 16 tiles and share 8 A panels and 16 B panels in its 4 MB L2, instead of
  // 2-3 A panels and every B panel of the matrix
  int m0, n0;
  {
    const int tilesM = gridDim.x / g.tilesN, per = GROUP_M * g.tilesN;
    const int grp = tile / per, rem2 = tile - grp * per, first = grp * GROUP_M;
    const int gsz = min(GROUP_M, tilesM - first);
    m0 = (first + rem2 % gsz) * BM;
    n0 = (rem2 / gsz) * BN;
  }
  const bool fullA = (g.vecA || !VEC) && (m0 + BM <= g.M), fullB = (g.vecB || !VEC) && (n0 + BN <= g.N);
  const int z = blockIdx.z, ks = blockIdx.y;
  const float* A = g.A + z * g.sA;
  const float* B = g.B + z * g.sB;
  const bool partial = g.splitk > 1;
  float* C = partial ? g.slabs + ((int64_t)z * g.splitk + ks) * g.M * g.N : g.C + z * g.sC;
  const int ldc = partial ? g.N : g.ldc;
  const float* bias = (g.bias && !partial) ? g.bias + z * g.sbias : nullptr;
  const int kbeg = ks * g.kchunk, kend = min(g.K, kbeg + g.kchunk);  // kchunk is a multiple of BK

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nk = (kend - kbeg + BK - 1) / BK;
  float4 ra[PIECES], rb[PIECES];
  // A is k-major in memory when TA (stored K x M); B is k-major when !TB (stored K x N)
  fetch<TA, VEC>(A, g.lda, m0, g.M, kbeg, kend, fullA && kbeg + BK <= kend, tid, ra);
  fetch<!TB, VEC>(B, g.ldb, n0, g.N, kbeg, kend, fullB && kbeg + BK <= kend, tid, rb);
  stash<TA, LDA>(As[0], tid, ra);
  stash<!TB, LDB>(Bs[0], tid, rb);
  __syncthreads();

  const int kh = lane >> 5, l31 = lane & 31;
  int cur = 0;
  for (int kt = 0; kt < nk; ++kt) {
    const bool more = kt + 1 < nk;
    if (more) {
      const int k0 = kbeg + (kt + 1) * BK;
      const bool kfull = k0 + BK <= kend;  // block-uniform
      fetch<TA, VEC>(A, g.lda, m0, g.M, k0, kend, fullA && kfull, tid, ra);
      fetch<!TB, VEC>(B, g.ldb, n0, g.N, k0, kend, fullB && kfull, tid, rb);
    }
#pragma unroll
    for (int kk = 0; kk < BK; kk += 2) {
      const float a0 = As[cur][kk + kh][wm * 64 + l31];
      const float a1 = As[cur][kk + kh][wm * 64 + 32 + l31];
      const float b0 = Bs[cur][kk + kh][wn * 64 + l31];
      const float b1 = Bs[cur][kk + kh][wn * 64 + 32 + l31];
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
    }
    if (more) {
      stash<TA, LDA>(As[cur ^ 1], tid, ra);
      stash<!TB, LDB>(Bs[cur ^ 1], tid, rb);
    }
    __syncthreads();
    cur ^= 1;
  }

  store_tile(g, acc, C, ldc, bias, partial, m0 + wm * 64, n0 + wn * 64, lane);
  if (partial && g.counters) finish_splitk(g, z, m0 + wm * 64, n0 + wn * 64, tid);
}

// ------------------------------------------------------------------------------------------------------
// The same product with both operand tiles DMA'd from global memory STRAIGHT INTO LDS (global_load_lds_dwordx4: no
// staging registers, no ds_write), for operands whose rows are 16-byte aligned and a K range that is a multiple of 16.
// Same block / wave / MFMA tiling and the same epilogue as gemm_f32_kernel, so the two are interchangeable per launch.
// A DMA writes lane-linearly (lane l -> LDS base + 16 l), so the LDS image is shaped on the SOURCE side:
//   operand stored [dim][K] (A of NN / NT, B of NT): piece q = tile rows 16q .. 16q+15, lane l fetches k chunk l>>4
//     (4 floats) of row l&15 (rotated by 8 in odd chunks, see dma_src) -> image offset (row>>4) * 1024 + chunk * 256 +
//     slot * 16: a wave instruction still
//     covers 16 rows x 64 contiguous bytes of memory, and a fragment is ONE ds_read_b128 per lane (16 consecutive
//     rows x one chunk = 16 distinct 16-byte bank groups).  The four floats of a chunk feed four MFMA steps: the K
//     index of step (c', i) is 8c' + 4 (lane>>5) + i -- any K order is a valid K order as long as both operands agree.
//   operand stored [K][dim] (A of TN, B of NN / TN): piece q = k rows 2q, 2q+1, 128 dims each; in rows with bit 2 of
//     k set the two 32-dim halves of every 64 are swapped (lane fetches dim ^ 32), so that the two half-waves of a
//     fragment read (k and k+4, same dims) fall on opposite halves of the 64 banks: conflict-free ds_read_b32.
// Rows / dims past the matrix edge are clamped to valid addresses: their products land in rows / columns never stored.
template <bool KMAJOR, int KBIT = 2>
__device__ __forceinline__ const float* dma_src(const float* P, int ld, int dim0, int dimLimit, int k0, int piece, int lane) {
  if (KMAJOR) {
    const int krow = 2 * piece + (lane >> 5);
    const int dim = (4 * (lane & 31)) ^ (((krow >> KBIT) & 1) << 5);  // KBIT: the k bit the two half-waves of a read differ in
    return P + (int64_t)(k0 + krow) * ld + min(dim0 + dim, dimLimit - 4);
  } else {
    // slot s of chunk c holds row (s - 8 * bit) & 15, bit = the chunk-index bit in which the two half-waves of a fragment
    // read differ (bit 0 for the fp32 kernel, bit 1 for the split kernel): lanes l and l + 32 then sit 128 bytes apart
    // (mod 256), so the read is conflict-free whichever 16 lanes the LDS serves together
    const int c = lane >> 4, row = 16 * piece + (((lane & 15) - 8 * ((c >> (KBIT - 2)) & 1)) & 15);
    return P + (int64_t)min(dim0 + row, dimLimit - 1) * ld + k0 + 4 * c;
  }
}

// byte offset inside an 8 KB operand image of this lane's fragment data for the 32-row fragment starting at row d0
// (a multiple of 32), before the per-step immediates (dim-major: + 512 c'; k-major: + 512 (8c' + i))
template <bool KMAJOR>
__device__ __forceinline__ int frag_base(int d0, int lane) {
  const int l31 = lane & 31, kh = lane >> 5;
  if (KMAJOR) return 4 * kh * 512 + ((d0 ^ (kh << 5)) + l31) * 4;
  return ((d0 + l31) >> 4) * 1024 + kh * 256 + (((l31 & 15) + 8 * kh) & 15) * 16;
}

template <bool KMAJOR>
__device__ __forceinline__ void frag_load(const char* img, int base, int cp, float (&v)[4]) {
  if (KMAJOR) {
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = *reinterpret_cast<const float*>(img + base + (8 * cp + i) * 512);
  } else {
    const float4 q = *reinterpret_cast<const float4*>(img + base + cp * 512);
    v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
  }
}

// One DMA instruction (64 lanes x 16 bytes -> 1 KB of LDS at lds_addr) as inline assembly: hipcc puts an
// s_waitcnt vmcnt(0) in front of every LDS read that follows an LDS-DMA BUILTIN it cannot prove disjoint, which
// serialises the next tile's fetch with this tile's products.  The kernel orders DMA and reads itself (vmcnt(0) +
// barrier at the top of each K step).  M0 = LDS base of the instruction (wave-uniform).
__device__ __forceinline__ void dma_1k(const float* src, unsigned lds_addr) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(lds_addr) : "memory");
}

template <bool TA, bool TB>
__global__ __launch_bounds__(256, 3) void gemm_f32_kernel_dma(GemmArgs g) {
  constexpr int TILE = BM * BK * 4;  // 8 KB per operand image
  // LDS stages: the DMA runs NST - 1 K steps ahead of the products.  A third stage (48 KB of LDS) measured +4 % on the
  // N/N and +1..4 % on the T/N products stand-alone, -4 % on N/T, and nothing on the training step: two are shipped.
  constexpr int NST = 2;
  __shared__ __attribute__((aligned(1024))) char lds[NST][2 * TILE];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  int tile = blockIdx.x;
  {
    const int nt = gridDim.x, q = nt >> 3, rem = nt & 7, x = tile & 7, j = tile >> 3;
    tile = (x < rem ? x * (q + 1) : rem * (q + 1) + (x - rem) * q) + j;
  }
  int m0, n0;
  {
    const int tilesM = gridDim.x / g.tilesN, per = GROUP_M * g.tilesN;
    const int grp = tile / per, rem2 = tile - grp * per, first = grp * GROUP_M;
    const int gsz = min(GROUP_M, tilesM - first);
    m0 = (first + rem2 % gsz) * BM;
    n0 = (rem2 / gsz) * BN;
  }
  const int z = blockIdx.z, ks = blockIdx.y;
  const float* A = g.A + z * g.sA;
  const float* B = g.B + z * g.sB;
  const bool partial = g.splitk > 1;
  float* C = partial ? g.slabs + ((int64_t)z * g.splitk + ks) * g.M * g.N : g.C + z * g.sC;
  const int ldc = partial ? g.N : g.ldc;
  const float* bias = (g.bias && !partial) ? g.bias + z * g.sbias : nullptr;
  const int kbeg = ks * g.kchunk, kend = min(g.K, kbeg + g.kchunk);
  const int nk = (kend - kbeg) / BK;  // the host sends only K ranges that are multiples of BK here

  // A is k-major in memory when TA (stored K x M); B is k-major when !TB (stored K x N).  Wave w DMAs pieces 2w, 2w+1.
  const float* srcA[2];
  const float* srcB[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    srcA[i] = dma_src<TA>(A, g.lda, m0, g.M, kbeg, 2 * wave + i, lane);
    srcB[i] = dma_src<!TB>(B, g.ldb, n0, g.N, kbeg, 2 * wave + i, lane);
  }
  const int64_t stepA = TA ? (int64_t)BK * g.lda : BK, stepB = !TB ? (int64_t)BK * g.ldb : BK;
  const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)&lds[0][0];
  const unsigned my_pieces = __builtin_amdgcn_readfirstlane(lds_base + 2 * wave * 1024);
  auto stage = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      dma_1k(srcA[i], my_pieces + buf * 2 * TILE + i * 1024);
#ifndef SK_ABL_HALFDMA  // (SK_ABL_*: ablation builds for tools/gemm_bench.py, `make gemm_variant`: results wrong by construction)
      dma_1k(srcB[i], my_pieces + buf * 2 * TILE + TILE + i * 1024);
#endif
      srcA[i] += stepA;
      srcB[i] += stepB;
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int fa0 = frag_base<TA>(wm * 64, lane), fa1 = frag_base<TA>(wm * 64 + 32, lane);
  const int fb0 = frag_base<!TB>(wn * 64, lane), fb1 = frag_base<!TB>(wn * 64 + 32, lane);

  if (nk > 0) stage(0);
  if (NST == 3 && nk > 1) stage(1);
  int cur = 0;
  for (int kt = 0; kt < nk; ++kt) {
#ifndef SK_ABL_NOBAR
    if (NST == 3 && kt + 1 < nk)
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");  // step kt has landed; the 4 instructions of step kt + 1 may still fly
    else
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's pieces of step kt have landed ...
    __syncthreads();                                   // ... and everybody's; all reads of the buffer refilled next are done
#endif
#ifndef SK_ABL_NODMA
    if (kt + NST - 1 < nk) stage(NST == 3 ? (cur + 2) % 3 : cur ^ 1);
#endif
    const char* ai = lds[cur];
    const char* bi = lds[cur] + TILE;
#pragma unroll
    for (int cp = 0; cp < 2; ++cp) {
      float a0[4], a1[4], b0[4], b1[4];
      frag_load<TA>(ai, fa0, cp, a0);
      frag_load<TA>(ai, fa1, cp, a1);
      frag_load<!TB>(bi, fb0, cp, b0);
      frag_load<!TB>(bi, fb1, cp, b1);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[i], b0[i], acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[i], b1[i], acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[i], b0[i], acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[i], b1[i], acc[1][1], 0, 0, 0);
      }
    }
    cur = NST == 3 ? (cur + 1) % 3 : cur ^ 1;
  }
  store_tile(g, acc, C, ldc, bias, partial, m0 + wm * 64, n0 + wn * 64, lane);
  if (partial && g.counters) finish_splitk(g, z, m0 + wm * 64, n0 + wn * 64, tid);
}

// ------------------------------------------------------------------------------------------------------
// The LDS-DMA kernel with a 256 x 128 block tile and EIGHT waves (4 x 2, each still 64 x 64): the B tile is fetched
// once per 256 rows instead of once per 128, i.e. 0.75 x the bytes moved global -> LDS per flop (the quantity the
// ablation in DESIGN.md 4b shows the kernel's loss to be proportional to), at the same waves per SIMD and the same
// registers per wave as the 128 x 128 kernel.  Same images, K order and epilogue; the A image is 16 pieces (256 rows),
// a k-major A row is one whole DMA instruction.
template <bool KMAJOR>
__device__ __forceinline__ const float* dma_src_w256(const float* P, int ld, int dim0, int dimLimit, int k0, int piece, int lane) {
  if (KMAJOR) {  // piece = k row; lane fetches 4 dims; the 32-dim halves of every 64 swapped where bit 2 of k is set
    const int dim = (4 * lane) ^ (((piece >> 2) & 1) << 5);
    return P + (int64_t)(k0 + piece) * ld + min(dim0 + dim, dimLimit - 4);
  }
  return dma_src<false>(P, ld, dim0, dimLimit, k0, piece, lane);  // 16 pieces of 16 rows
}

template <bool KMAJOR, int DIM>  // DIM = tile dimension of the image (row bytes of a k-major image = 4 DIM)
__device__ __forceinline__ int frag_base_w(int d0, int lane) {
  const int l31 = lane & 31, kh = lane >> 5;
  if (KMAJOR) return 4 * kh * (4 * DIM) + ((d0 ^ (kh << 5)) + l31) * 4;
  return ((d0 + l31) >> 4) * 1024 + kh * 256 + (((l31 & 15) + 8 * kh) & 15) * 16;
}

template <bool KMAJOR, int DIM>
__device__ __forceinline__ void frag_load_w(const char* img, int base, int cp, float (&v)[4]) {
  if (KMAJOR) {
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = *reinterpret_cast<const float*>(img + base + (8 * cp + i) * (4 * DIM));
  } else {
    const float4 q = *reinterpret_cast<const float4*>(img + base + cp * 512);
    v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
  }
}

template <bool TA, bool TB>
__global__ __launch_bounds__(512, 2) void gemm_f32_kernel_dma256(GemmArgs g) {
  constexpr int BMW = 256;
  constexpr int TILE_A = BMW * BK * 4, TILE_B = BN * BK * 4, STAGE = TILE_A + TILE_B;  // 16 + 8 KB
  __shared__ __attribute__((aligned(1024))) char lds[2][STAGE];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  int tile = blockIdx.x;
  {
    const int nt = gridDim.x, q = nt >> 3, rem = nt & 7, x = tile & 7, j = tile >> 3;
    tile = (x < rem ? x * (q + 1) : rem * (q + 1) + (x - rem) * q) + j;
  }
  int m0, n0;
  {
    constexpr int GM = GROUP_M / 2;  // the same 1024 rows per group
    const int tilesM = gridDim.x / g.tilesN, per = GM * g.tilesN;
    const int grp = tile / per, rem2 = tile - grp * per, first = grp * GM;
    const int gsz = min(GM, tilesM - first);
    m0 = (first + rem2 % gsz) * BMW;
    n0 = (rem2 / gsz) * BN;
  }
  const int z = blockIdx.z, ks = blockIdx.y;
  const float* A = g.A + z * g.sA;
  const float* B = g.B + z * g.sB;
  const bool partial = g.splitk > 1;
  float* C = partial ? g.slabs + ((int64_t)z * g.splitk + ks) * g.M * g.N : g.C + z * g.sC;
  const int ldc = partial ? g.N : g.ldc;
  const float* bias = (g.bias && !partial) ? g.bias + z * g.sbias : nullptr;
  const int kbeg = ks * g.kchunk, kend = min(g.K, kbeg + g.kchunk);
  const int nk = (kend - kbeg) / BK;

  // wave w DMAs A pieces 2w, 2w+1 (of 16) and B piece w (of 8)
  const float* srcA[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) srcA[i] = dma_src_w256<TA>(A, g.lda, m0, g.M, kbeg, 2 * wave + i, lane);
  const float* srcB = dma_src<!TB>(B, g.ldb, n0, g.N, kbeg, wave, lane);
  const int64_t stepA = TA ? (int64_t)BK * g.lda : BK, stepB = !TB ? (int64_t)BK * g.ldb : BK;
  const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)&lds[0][0];
  const unsigned pa = __builtin_amdgcn_readfirstlane(lds_base + 2 * wave * 1024);
  const unsigned pb = __builtin_amdgcn_readfirstlane(lds_base + TILE_A + wave * 1024);
  auto stage = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      dma_1k(srcA[i], pa + buf * STAGE + i * 1024);
      srcA[i] += stepA;
    }
    dma_1k(srcB, pb + buf * STAGE);
    srcB += stepB;
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int fa0 = frag_base_w<TA, BMW>(wm * 64, lane), fa1 = frag_base_w<TA, BMW>(wm * 64 + 32, lane);
  const int fb0 = frag_base_w<!TB, BN>(wn * 64, lane), fb1 = frag_base_w<!TB, BN>(wn * 64 + 32, lane);

  if (nk > 0) stage(0);
  int cur = 0;
  for (int kt = 0; kt < nk; ++kt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (kt + 1 < nk) stage(cur ^ 1);
    const char* ai = lds[cur];
    const char* bi = lds[cur] + TILE_A;
#pragma unroll
    for (int cp = 0; cp < 2; ++cp) {
      float a0[4], a1[4], b0[4], b1[4];
      frag_load_w<TA, BMW>(ai, fa0, cp, a0);
      frag_load_w<TA, BMW>(ai, fa1, cp, a1);
      frag_load_w<!TB, BN>(bi, fb0, cp, b0);
      frag_load_w<!TB, BN>(bi, fb1, cp, b1);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[i], b0[i], acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[i], b1[i], acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[i], b0[i], acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[i], b1[i], acc[1][1], 0, 0, 0);
      }
    }
    cur ^= 1;
  }
  store_tile(g, acc, C, ldc, bias, partial, m0 + wm * 64, n0 + wn * 64, lane);
  if (partial && g.counters) finish_splitk(g, z, m0 + wm * 64, n0 + wn * 64, tid);
}

// ------------------------------------------------------------------------------------------------------
// The three-way bf16 split of an fp32 operand (used by gemm_f32_kernel_split3 and gemm_f32_kernel_planes below; the arithmetic is
// described there): x = hi + mid + lo exactly, three bf16 pieces.
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
struct Split3 {
  bf16x8_t hi, mid, lo;
};

// Pieces by ROUND-TO-NEAREST (v_cvt_pk_bf16_f32): hi = bf16(x), r = x - hi (exact: a multiple of ulp(x); bf16 has 8 significand
// bits, so |r| <= 2^-8 |x|), mid = bf16(r), lo = r - mid (exact, |lo| <= 2^-8 |r| <= 2^-16 |x|, at most 8 significant bits: IS a
// bf16).  (r04 cut by
// truncation: one instruction fewer per pair, but then |mid| < 2^-7 |x|, |lo| < 2^-15 |x| and all pieces share x's sign -- the
// products left out below were up to 2^-21 of a b and a systematic shrink; rounded pieces make them <= 2^-23 and signless.)
// Operands beyond bf16's finite range (|x| > 3.39e38) round to inf.
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ Split3 split3(const float (&v)[8]) {
#ifdef SK_SPLIT_FREE  // TIMING-ONLY diagnostic (wrong numerics): the pieces cost nothing -- an upper bound for a kernel that finds them ready in LDS
  typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));
  Split3 f;
  f.hi = __builtin_bit_cast(bf16x8_t, (u32x4_){__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])});
  f.mid = __builtin_bit_cast(bf16x8_t, (u32x4_){__float_as_uint(v[4]), __float_as_uint(v[5]), __float_as_uint(v[6]), __float_as_uint(v[7])});
  f.lo = __builtin_bit_cast(bf16x8_t, (u32x4_){__float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[5]), __float_as_uint(v[6])});
  return f;
#endif
  unsigned h[4], m[4], l[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const f32x2 x = {v[2 * j], v[2 * j + 1]};
    h[j] = __builtin_bit_cast(unsigned, __builtin_convertvector(x, bf16x2_t));  // element 2j in the low half
    const f32x2 xh = {__uint_as_float(h[j] << 16), __uint_as_float(h[j] & 0xffff0000u)};
    const f32x2 r = x - xh;  // exact
    m[j] = __builtin_bit_cast(unsigned, __builtin_convertvector(r, bf16x2_t));
    const f32x2 rh = {__uint_as_float(m[j] << 16), __uint_as_float(m[j] & 0xffff0000u)};
    const f32x2 q = r - rh;  // exact, <= 8 significant bits
    l[j] = __builtin_bit_cast(unsigned, __builtin_convertvector(q, bf16x2_t));
  }
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  Split3 o;
  o.hi = __builtin_bit_cast(bf16x8_t, (u32x4){h[0], h[1], h[2], h[3]});
  o.mid = __builtin_bit_cast(bf16x8_t, (u32x4){m[0], m[1], m[2], m[3]});
  o.lo = __builtin_bit_cast(bf16x8_t, (u32x4){l[0], l[1], l[2], l[3]});
  return o;
}

__device__ __forceinline__ void mma6(f32x16& acc, const Split3& a, const Split3& b) {
  // the SIX piece products, small terms first (it is one fp32 accumulator either way); SK_SPLIT_NINE: all nine (diagnostic build)
#ifdef SK_SPLIT_NINE
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.lo, b.lo, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.lo, b.mid, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.mid, b.lo, acc, 0, 0, 0);
#endif
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.lo, b.hi, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.hi, b.lo, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.mid, b.mid, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.mid, b.hi, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.hi, b.mid, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.hi, b.hi, acc, 0, 0, 0);
}

// Sign phases of the split-product kernels.  The bf16 MFMA does not round the alignment of its addends to nearest: it TRUNCATES
// TOWARDS MINUS INFINITY (measured, profiles/r05_planes_flip.txt: on all-positive operands the fp32-MFMA kernels' mean signed error
// is 1e-10 of the result, a split kernel's -3.7e-8 at K = 1792, -1.5e-7 at K = 7168, -2.7e-7 at K = 12800) -- a DC offset, the same
// in every element, far below the kernel's rel-L2 error against fp64 but COHERENT: whatever integrates a product's result over
// many elements or steps (the recurrence below a data gradient, r05: layer-0 / 1 gradients 5.9e-6 from a float64 step instead of
// 6.2e-7) sees it.  Cure without a second accumulator: the accumulators hold +sum in some stretches of K and -sum in others (at a
// boundary acc = -acc, and within a negative stretch the B operand's values are negated before they are split: the pieces of -x
// are the negated pieces of x, rounding to nearest is symmetric), so that truncation pulls the sum down in one and up in the next.
// Stretches of q K steps, signs + - - + + - - + ...: over a period of 4 q the offsets cancel both for a sum of constant size and
// for one that grows linearly with k (all-positive operands), and the sign changes at every second boundary only.  q (GemmArgs::
// flip_q, chosen by the launcher from the WHOLE K: periods of about 64 steps, a whole number of them) is counted in GLOBAL K
// steps, so the slices of a split-K product continue one pattern.  r05 had this in the N/N planes kernel only (+ - + -, 32 steps);
// r06: every split-product kernel and form (tests/test_gpu_signed_error.py pins the mean signed error of each).
struct SignPhase {
  int q, left, p;
  unsigned mask;  // 0 or the sign bit: the sign of the stretch the tracked K step lies in
  __device__ __forceinline__ void init(int q_, int kstep0) {
    q = q_; p = 0; left = 0x7fffffff; mask = 0u;
    if (q > 0) {
      p = kstep0 / q;
      left = q - (kstep0 - p * q);
      mask = ((p + 1) & 2) ? 0x80000000u : 0u;
    }
  }
  __device__ __forceinline__ void advance() {
    if (--left == 0) {
      left = q;
      ++p;
      mask = ((p + 1) & 2) ? 0x80000000u : 0u;
    }
  }
};


// ------------------------------------------------------------------------------------------------------
// 256 x 256 block tiles (eight waves of 64 x 128: per flop HALF the bytes global -> LDS of the 128 x 128 kernels and 3/4 of
// their fragment bytes) as a PERSISTENT stream-K kernel (r03, variant 6; unsplit, unbatched products): one workgroup per
// CU, P = gridDim.x of them.  With one workgroup per CU a partial last round of tiles costs a whole round (1400 tiles on
// 256 CUs: 6 rounds for 5.47 of work), which is what kept the 256 x 256 kernel off the training step.  Here workgroup b
// first takes the tiles b, P + b, ... of `sk_full` whole rounds in the usual XCD-aware order, then its share of the
// REMAINING r = tiles - sk_full P tiles: their r nk K steps are laid end to end and cut into P equal ranges, so a
// workgroup's range covers pieces of at most two tiles.  A piece goes to the workgroup's slab in the workspace
// (write-through stores), a ticket is drawn from the tile's counter, and the workgroup that draws the LAST ticket adds
// the tile's pieces in workgroup order (deterministic), applies bias / accumulate / act and stores: nobody ever waits for
// anybody, so the grid need not be co-resident.  The K steps of consecutive segments form ONE software pipeline: the
// next segment's first two stages are in flight while a tile is stored.
// (r05 also carried a split-product form of this kernel, variant 7; since the planes kernel it served no launch of any
// configuration and was retired in r06.)
template <bool TA, bool TB>
__global__ __launch_bounds__(512, 2) void gemm_f32_kernel_streamk(GemmArgs g) {
  constexpr int BMW = 256, BNW = 256;
  constexpr int TILE = BMW * BK * 4, STAGE = 2 * TILE;
  constexpr int NST = 3;
  constexpr int SLAB = BMW * BNW;  // floats per piece
  __shared__ __attribute__((aligned(1024))) char lds[NST][STAGE];
  __shared__ int s_fix[4];  // last ticket?, first and last contributor, which piece of the first

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int P = gridDim.x, b = blockIdx.x;
  const int nk = g.K / BK, nt = g.sk_tiles, full = g.sk_full;
  const int64_t R = (int64_t)(nt - full * P) * nk;
  const int64_t beg = b * R / P, end = (b + 1) * R / P;
  const int t0 = (int)(beg / nk);                                        // first stream-K tile this workgroup touches
  const int nseg = full + (end > beg ? (int)((end - 1) / nk) - t0 + 1 : 0);  // + 0, 1 or 2 pieces

  // segment i of this workgroup: position v in the tile order, K steps [kb, ke)
  auto segment = [&](int i, int& v, int& kb, int& ke) {
    if (i < full) {
      v = i * P + b; kb = 0; ke = nk;
    } else {
      const int t = t0 + (i - full);
      const int64_t lo = (int64_t)t * nk;
      v = full * P + t;
      kb = (int)(max(beg, lo) - lo);
      ke = (int)(min(end, lo + nk) - lo);
    }
  };
  // the tile order of the other kernels (each XCD walks its own run of the list; 8-row groups column-wise inside it)
  auto origin = [&](int v, int& m0, int& n0) {
    const int q = nt >> 3, rem = nt & 7, x = v & 7, j = v >> 3;
    const int tile = (x < rem ? x * (q + 1) : rem * (q + 1) + (x - rem) * q) + j;
    constexpr int GM = GROUP_M / 2;
    const int tilesM = nt / g.tilesN, per = GM * g.tilesN;
    const int grp = tile / per, rem2 = tile - grp * per, first = grp * GM;
    const int gsz = min(GM, tilesM - first);
    m0 = (first + rem2 % gsz) * BMW;
    n0 = (rem2 / gsz) * BNW;
  };

  // DMA cursor: two K steps ahead of the products, across segment boundaries
  const int64_t stepA = TA ? (int64_t)BK * g.lda : BK, stepB = !TB ? (int64_t)BK * g.ldb : BK;
  const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)&lds[0][0];
  const unsigned pa = __builtin_amdgcn_readfirstlane(lds_base + 2 * wave * 1024);
  const float* srcA[2];
  const float* srcB[2];
  int di = 0, dk = 0, dke = 0;
  auto dma_open = [&]() {
    int v, kb, ke, m0, n0;
    segment(di, v, kb, ke);
    origin(v, m0, n0);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      srcA[i] = dma_src_w256<TA>(g.A, g.lda, m0, g.M, kb * BK, 2 * wave + i, lane);
      srcB[i] = dma_src_w256<!TB>(g.B, g.ldb, n0, g.N, kb * BK, 2 * wave + i, lane);
    }
    dk = kb;
    dke = ke;
  };
  auto fetch = [&](int buf) {
    if (di >= nseg) return;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      dma_1k(srcA[i], pa + buf * STAGE + i * 1024);
      dma_1k(srcB[i], pa + buf * STAGE + TILE + i * 1024);
      srcA[i] += stepA;
      srcB[i] += stepB;
    }
    if (++dk == dke && ++di < nseg) dma_open();
  };

  f32x16 acc[2][4];
  auto clear = [&]() {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  };
  auto store = [&](int m0, int n0) {  // the wave's two 64-column halves through the 64 x 64 epilogue
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const f32x16 part[2][2] = {{acc[0][2 * h], acc[0][2 * h + 1]}, {acc[1][2 * h], acc[1][2 * h + 1]}};
      store_tile(g, part, g.C, g.ldc, g.bias, false, m0 + wm * 64, n0 + wn * 128 + 64 * h, lane);
    }
  };
  clear();

  int fa[2], fb[4];
#pragma unroll
  for (int i = 0; i < 2; ++i) fa[i] = frag_base_w<TA, BMW>(wm * 64 + 32 * i, lane);
#pragma unroll
  for (int j = 0; j < 4; ++j) fb[j] = frag_base_w<!TB, BNW>(wn * 128 + 32 * j, lane);

  const int64_t F = (int64_t)full * nk + (end - beg);  // K steps of this workgroup, all segments
  if (nseg > 0) dma_open();
  fetch(0);
  fetch(1);
  int ci = 0, cv = 0, ck = 0, cke = 0;
  if (nseg > 0) segment(0, cv, ck, cke);
  int cur = 0;
  for (int64_t f = 0; f < F; ++f) {
    // step f has landed when at most the 4 instructions of step f + 1 are in flight (loads return in order; stores of an
    // epilogue in between only make the wait conservative)
    if (f + 1 < F)
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    fetch((cur + 2) % NST);
    const char* ai = lds[cur];
    const char* bi = lds[cur] + TILE;
    {
#pragma unroll
      for (int cp = 0; cp < 2; ++cp) {
        float a[2][4], bb[4][4];
#pragma unroll
        for (int i = 0; i < 2; ++i) frag_load_w<TA, BMW>(ai, fa[i], cp, a[i]);
#pragma unroll
        for (int j = 0; j < 4; ++j) frag_load_w<!TB, BNW>(bi, fb[j], cp, bb[j]);
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][k], bb[j][k], acc[i][j], 0, 0, 0);
      }
    }
    cur = (cur + 1) % NST;
    if (++ck < cke) continue;

    // ---- end of a segment
    int m0, n0;
    origin(cv, m0, n0);
    if (ci < full) {
      store(m0, n0);
    } else {
      const int t = cv - full * P;
      float* mine = g.slabs + (size_t)(2 * b + (t - t0)) * SLAB;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            __hip_atomic_store(mine + ((i * 4 + j) * 16 + r) * 512 + tid, acc[i][j][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) {
        // the workgroup whose range holds K step i of the region: the largest w with floor(w R / P) <= i
        const int w0 = (int)((((int64_t)t * nk + 1) * P - 1) / R), w1 = (int)((((int64_t)(t + 1) * nk) * P - 1) / R);
        const unsigned old = __hip_atomic_fetch_add(g.counters + t, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = old == (unsigned)(w1 - w0);
        if (last) __hip_atomic_store(g.counters + t, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_fix[0] = last; s_fix[1] = w0; s_fix[2] = w1;
        s_fix[3] = (w0 * R / P < (int64_t)t * nk) ? 1 : 0;  // the first contributor's piece is its second one unless its range starts here
      }
      __syncthreads();
      if (s_fix[0]) {
        // memory -> memory: element q * 512 + tid of a piece is accumulator register q of thread tid.  Piece of workgroup w:
        // slab 2 w (+ 1 for the first contributor when the tile is the second one of its range)
        const int w0 = s_fix[1], w1 = s_fix[2];
        const float* sl0 = g.slabs + (size_t)(2 * w0 + s_fix[3]) * SLAB + tid;
        const int kh = lane >> 5, col = n0 + wn * 128 + (lane & 31), row = m0 + wm * 64 + 4 * kh;
        // 32 values per thread and round trip: the accumulators are dead here, and the tile waits on latency, not bandwidth
        for (int q0 = 0; q0 < 128; q0 += 32) {
          float v[32];
#pragma unroll
          for (int u = 0; u < 32; ++u) v[u] = __hip_atomic_load(sl0 + (q0 + u) * 512, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          for (int w = w0 + 1; w <= w1; ++w) {
            const float* sl = g.slabs + (size_t)(2 * w) * SLAB + tid;
#pragma unroll
            for (int u = 0; u < 32; ++u) v[u] += __hip_atomic_load(sl + (q0 + u) * 512, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
#pragma unroll
          for (int u = 0; u < 32; ++u) {
            const int q = q0 + u, i = q >> 6, j = (q >> 4) & 3, r = q & 15;
            const int c = col + 32 * j, rw = row + 32 * i + (r & 3) + 8 * (r >> 2);
            if (c < g.N && rw < g.M) {
              float* cp = g.C + (int64_t)rw * g.ldc + c;
              float x = v[u];
              if (g.bias) x += g.bias[c];
              if (g.accumulate) x += *cp;
              if (g.act == 1) x = sk_sigmoid(x);
              *cp = x;
            }
          }
        }
      }
    }
    clear();
    if (++ci < nseg) segment(ci, cv, ck, cke);
  }
}

// ------------------------------------------------------------------------------------------------------
// fp32 product on the bf16 matrix pipe by an EXACT three-way split of both operands (variant 2 of sk_gemm_f32_splitk; what
// variant 0 chooses for every product with aligned operands -- r05).
// Every fp32 operand element x is cut into three bf16 pieces, x = hi + mid + lo exactly (24 significand bits = 3 x 8; pieces by
// rounding to nearest, see split3() above: |mid| <= 2^-8 |x|, |lo| <= 2^-16 |x|).  a*b is the sum of the nine piece products,
// each exact in fp32 (8 x 8 bits), of relative sizes 1, 2^-8 (two), 2^-16 (three), 2^-24 (two), 2^-32.  The kernel forms the
// SIX of size >= 2^-16 and the matrix cores add them into the same fp32 accumulators as always; the three it leaves out are
// together <= 2^-23 |a||b| in the worst case -- one ulp of the product: a SINGLE product may be off by about an ulp where an fp32
// FMA is exact (typical pieces: a quarter of that), which is at the level of the rounding of
// the fp32 accumulation that follows.  So this is an fp32 GEMM in another summation order, not a lower-precision one (tests:
// error against fp64 not above the fp32-MFMA kernel's on operands spanning 2^+-20; single products within 1 ulp, exact when
// both factors have <= 16 significant bits).  Six v_mfma_f32_32x32x16_bf16 (8 passes each, K = 16) replace eight
// v_mfma_f32_32x32x2_f32 (16 passes each): 48 passes instead of 128 per 32 x 32 x 16 block, for 4.5 VALU instructions per
// operand element to split it (other waves' products run meanwhile).  Measured (one MI355X, stand-alone): 160-168 TFLOP/s
// fp32-equivalent on the training step's large products against 124-135 of the fp32-MFMA kernels (whose pipe peaks at 157);
// sustained it is POWER-bound: 181 TFLOP/s at 1.9 GHz and 1375 W of the 1400 W cap (the fp32-MFMA kernels: 126-130 at 2.39 GHz,
// 1140-1260 W).  (-DSK_SPLIT_NINE builds all nine products: 130 TFLOP/s, the r03 form.)  Operand tiles are DMA'd into LDS as
// fp32 exactly as in gemm_f32_kernel_dma; the K order is the natural one (lane holds k = 8 (lane>>5) .. +7 of its row, the
// bf16 MFMA's fragment shape).  Non-finite inputs: x = +-inf splits into (inf, nan, nan): such a product is NaN where an fp32
// FMA gives +-inf; |x| > 3.39e38 rounds to inf.
// this lane's 8 consecutive-k fp32 values of its row of the 32-row fragment starting at d0
template <bool KMAJOR>
__device__ __forceinline__ void frag8_load(const char* img, int d0, int lane, float (&v)[8]) {
  const int l31 = lane & 31, kh = lane >> 5;
  if (KMAJOR) {
    const int base = 8 * kh * 512 + ((d0 ^ (kh << 5)) + l31) * 4;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = *reinterpret_cast<const float*>(img + base + j * 512);
  } else {
    const int base = ((d0 + l31) >> 4) * 1024 + 2 * kh * 256 + (((l31 & 15) + 8 * kh) & 15) * 16;
    const float4 p = *reinterpret_cast<const float4*>(img + base), q = *reinterpret_cast<const float4*>(img + base + 256);
    v[0] = p.x; v[1] = p.y; v[2] = p.z; v[3] = p.w;
    v[4] = q.x; v[5] = q.y; v[6] = q.z; v[7] = q.w;
  }
}

// LDS stages (the DMA runs NST - 1 K steps ahead of the products) and resident blocks per CU: tuning constants.  Measured (r05,
// profiles/r05_gemm_split_tuning.txt): stand-alone 160-167 TFLOP/s whatever the stages (2, 3, 4) or the minimum occupancy (2, 3)
// -- at 122 VGPRs and 32 KB of LDS four blocks are resident per CU and the waves of different blocks fill each other's VALU /
// MFMA phases; in the training step more stages LOSE (48-64 KB of LDS per block: fewer blocks fit beside a recurrence workgroup,
// 30.6-32.0 vs 29.7 ms).  A form software-pipelined across K steps (next step's fragment reads and splits in the shadow of this
// step's MFMAs; 170 VGPRs, 3-4 stages) was built, passed the tests and was removed: 157-165 TFLOP/s, 30.6-32.0 ms in the step
// (profiles/r05_gemm_split_pipe.txt) -- the kernel is bound by the work it issues (sustained: power), not by latency.
#ifndef SK_SPLIT_NST
#define SK_SPLIT_NST 2
#endif
#ifndef SK_SPLIT_OCC
#define SK_SPLIT_OCC 2
#endif
template <bool TA, bool TB>
__global__ __launch_bounds__(256, SK_SPLIT_OCC) void gemm_f32_kernel_split3(GemmArgs g) {
  constexpr int TILE = BM * BK * 4;  // 8 KB per operand image
  constexpr int NST = SK_SPLIT_NST;
  static_assert(NST >= 2 && NST <= 4, "2..4 LDS stages");
  __shared__ __attribute__((aligned(1024))) char lds[NST][2 * TILE];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  int tile = blockIdx.x;
  {
    const int nt = gridDim.x, q = nt >> 3, rem = nt & 7, x = tile & 7, j = tile >> 3;
    tile = (x < rem ? x * (q + 1) : rem * (q + 1) + (x - rem) * q) + j;
  }
  int m0, n0;
  {
    const int tilesM = gridDim.x / g.tilesN, per = GROUP_M * g.tilesN;
    const int grp = tile / per, rem2 = tile - grp * per, first = grp * GROUP_M;
    const int gsz = min(GROUP_M, tilesM - first);
    m0 = (first + rem2 % gsz) * BM;
    n0 = (rem2 / gsz) * BN;
  }
  const int z = blockIdx.z, ks = blockIdx.y;
  const float* A = g.A + z * g.sA;
  const float* B = g.B + z * g.sB;
  const bool partial = g.splitk > 1;
  float* C = partial ? g.slabs + ((int64_t)z * g.splitk + ks) * g.M * g.N : g.C + z * g.sC;
  const int ldc = partial ? g.N : g.ldc;
  const float* bias = (g.bias && !partial) ? g.bias + z * g.sbias : nullptr;
  const int kbeg = ks * g.kchunk, kend = min(g.K, kbeg + g.kchunk);
  const int nk = (kend - kbeg) / BK;

  const float* srcA[2];
  const float* srcB[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    srcA[i] = dma_src<TA, 3>(A, g.lda, m0, g.M, kbeg, 2 * wave + i, lane);
    srcB[i] = dma_src<!TB, 3>(B, g.ldb, n0, g.N, kbeg, 2 * wave + i, lane);
  }
  const int64_t stepA = TA ? (int64_t)BK * g.lda : BK, stepB = !TB ? (int64_t)BK * g.ldb : BK;
  const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)&lds[0][0];
  const unsigned my_pieces = __builtin_amdgcn_readfirstlane(lds_base + 2 * wave * 1024);
  auto stage = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      dma_1k(srcA[i], my_pieces + buf * 2 * TILE + i * 1024);
      dma_1k(srcB[i], my_pieces + buf * 2 * TILE + TILE + i * 1024);
      srcA[i] += stepA;
      srcB[i] += stepB;
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

#pragma unroll
  for (int i = 0; i < NST - 1; ++i)
    if (i < nk) stage(i);
  int cur = 0;
  // sign phases (SignPhase above): this slice's steps are the global K steps kbeg / BK ...; `held` = the sign the accumulators carry
  SignPhase ph;
  ph.init(g.flip_q, kbeg / BK);
  unsigned held = 0u;
  // One K step; sg = -1 in a negative stretch: the B fragments' values are multiplied by it before they are split (exact; eight
  // v_pk_mul_f32 per wave and step beside the ~150 of the splits.  Two copies of the body, one with the negation folded into the
  // split's source modifiers, were tried first: 180 instead of 122 VGPRs -- the kernel no longer fits beside a recurrence).
  auto kstep = [&](const float sg) {
    const char* ai = lds[cur];
    const char* bi = lds[cur] + TILE;
    float va0[8], va1[8], vb0[8], vb1[8];
    frag8_load<TA>(ai, wm * 64, lane, va0);
    frag8_load<!TB>(bi, wn * 64, lane, vb0);
    frag8_load<TA>(ai, wm * 64 + 32, lane, va1);
    frag8_load<!TB>(bi, wn * 64 + 32, lane, vb1);
#pragma unroll
    for (int j = 0; j < 8; j += 2) {
      const f32x2 s2 = {sg, sg};
      const f32x2 p0 = (f32x2){vb0[j], vb0[j + 1]} * s2, p1 = (f32x2){vb1[j], vb1[j + 1]} * s2;
      vb0[j] = p0[0]; vb0[j + 1] = p0[1];
      vb1[j] = p1[0]; vb1[j + 1] = p1[1];
    }
#ifdef SK_SPLIT_FREE_TN  // TIMING-ONLY diagnostic (wrong, finite numerics): the T/N form (weight gradients) finds its pieces for free
    auto sp = [&](const float (&v)[8]) {
      if constexpr (TA && !TB) {
        typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));
        Split3 f;
        f.hi = __builtin_bit_cast(bf16x8_t, (u32x4_){__float_as_uint(v[0]) & 0x3f7f3f7fu, __float_as_uint(v[1]) & 0x3f7f3f7fu, __float_as_uint(v[2]) & 0x3f7f3f7fu, __float_as_uint(v[3]) & 0x3f7f3f7fu});
        f.mid = __builtin_bit_cast(bf16x8_t, (u32x4_){__float_as_uint(v[4]) & 0x3f7f3f7fu, __float_as_uint(v[5]) & 0x3f7f3f7fu, __float_as_uint(v[6]) & 0x3f7f3f7fu, __float_as_uint(v[7]) & 0x3f7f3f7fu});
        f.lo = f.hi;
        return f;
      } else {
        return split3(v);
      }
    };
#else
    auto sp = [&](const float (&v)[8]) { return split3(v); };
#endif
    const Split3 a0 = sp(va0), b0 = sp(vb0);
    mma6(acc[0][0], a0, b0);
    const Split3 a1 = sp(va1);
    mma6(acc[1][0], a1, b0);
    const Split3 b1 = sp(vb1);
    mma6(acc[0][1], a0, b1);
    mma6(acc[1][1], a1, b1);
  };
  auto negate_acc = [&]() {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = -acc[i][j];
  };
  for (int kt = 0; kt < nk; ++kt) {
    // step kt has landed when only the DMAs of the (up to NST - 2) stages issued after it may still fly: 4 instructions each
    const int later = min(NST - 2, nk - 1 - kt);
    if (later >= 2)
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (later == 1)
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // everybody's pieces of step kt have landed; all reads of the buffer refilled next are done
    if (kt + NST - 1 < nk) stage((cur + NST - 1) % NST);
    if (ph.mask != held) {  // a stretch of the other sign begins: the accumulators change sign with the products
      negate_acc();
      held = ph.mask;
    }
    kstep(held ? -1.0f : 1.0f);
    ph.advance();
    cur = (cur + 1) % NST;
  }
  if (held) negate_acc();  // the last stretch held -sum
  store_tile(g, acc, C, ldc, bias, partial, m0 + wm * 64, n0 + wn * 64, lane);
  if (partial && g.counters) finish_splitk(g, z, m0 + wm * 64, n0 + wn * 64, tid);
}

// ------------------------------------------------------------------------------------------------------
// Split products with the split done ONCE per element (variant 9; unsplit, unbatched N/T, N/N and T/N products).
// gemm_f32_kernel_split3 / the S6 stream-K form stage fp32 tiles by LDS-DMA and every WAVE splits the fragments it reads -- 2 to 4
// waves split the same element, and the split (4.5 VALU instructions per element) is ~22 % of those kernels' time and energy
// (profiles/r05_gemm_split_free_upper_bound.txt).  Here a 256 x 256 x 16 tile is staged through registers: every thread fetches
// 16 operand values (four float4: 4 consecutive k of a row), splits them once and writes the three bf16 PLANES of the tile to
// LDS; the waves read ready-made bf16 fragments (ds_read_b128: 8 consecutive k of a row per lane and plane) and issue the six
// MFMAs per fragment pair.  Per wave and K step: 72 VALU + 12 ds_write_b64 + 18 ds_read_b128 + 48 MFMAs.
// LDS image of one plane of one operand: [256 rows][16 k] bf16 = 32 B per row; the two 16-byte k-halves of a row are swapped
// where bit 3 of the row is set, so that the 16 rows a fragment read serves together hit 16 different 16-byte bank groups, and
// the four 8-byte pieces of a row written by neighbouring lanes stay contiguous.  Two stages (96 KB), one workgroup per CU.
// The global loads of step k + 2 are issued before the products of step k and land during them.
struct Pl4 {
  unsigned h[2], m[2], l[2];  // hi / mid / lo pieces of 4 consecutive-k values (two packed pairs each)
};
__device__ __forceinline__ Pl4 split4(const float4& x4) {
  Pl4 o;
  const float v[4] = {x4.x, x4.y, x4.z, x4.w};
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const f32x2 x = {v[2 * j], v[2 * j + 1]};
    o.h[j] = __builtin_bit_cast(unsigned, __builtin_convertvector(x, bf16x2_t));
    const f32x2 xh = {__uint_as_float(o.h[j] << 16), __uint_as_float(o.h[j] & 0xffff0000u)};
    const f32x2 r = x - xh;
    o.m[j] = __builtin_bit_cast(unsigned, __builtin_convertvector(r, bf16x2_t));
    const f32x2 rh = {__uint_as_float(o.m[j] << 16), __uint_as_float(o.m[j] & 0xffff0000u)};
    const f32x2 q = r - rh;
    o.l[j] = __builtin_bit_cast(unsigned, __builtin_convertvector(q, bf16x2_t));
  }
  return o;
}

// 256 x 128 tile, 8 waves (4 x 2) of 64 x 64; TWO sets of staging registers: the fetch runs two K steps ahead of the split, and the
// split of the next step is interleaved with this step's MFMA groups.  AKM / BKM: the operand is K-major in memory (A stored
// [K][M] = transA; B stored [K][N] = !transB): its planes are then kept K-major in LDS too ([16 k][DIM] bf16 per plane; a thread
// stages 4 consecutive DIMS of one k row) and the fragments are gathered by ds_read_b64_tr_b16 (see kmaj_frag below): within a k
// row the 32-byte chunk c (16 dims) sits at chunk c ^ 2 (k & 3), so that the 8 segments a half-wave reads in one instruction (two
// adjacent 16-dim blocks x 4 k rows) fall on 8 different 32-byte bank groups.
#ifndef SK_SPLIT_FLIP
#define SK_SPLIT_FLIP 1  // sign phases of the split-product kernels (SignPhase above; 0: none -- `make gemm_variant DEFS=-DSK_SPLIT_FLIP=0`)
#endif
#ifndef SK_PLANES_SCHED
#define SK_PLANES_SCHED 4  // VALU instructions stated behind every MFMA of a K step (0: the scheduler's own choice)
#endif
template <bool AKM, bool BKM>
__global__ __launch_bounds__(512, 2) void gemm_f32_kernel_planes(GemmArgs g) {
  constexpr int BMW = 256, BNW = 128, NJ = 2;
  constexpr int NL = 3;                    // float4 per thread and K step: 2 of the A tile, 1 of the B tile
  constexpr int PLANE_A = BMW * 32, PLANE_B = BNW * 32;  // bytes of one plane of an operand tile (16 k x DIM bf16)
  constexpr int OPER_A = 3 * PLANE_A, OPER_B = 3 * PLANE_B;
  constexpr int STAGE = OPER_A + OPER_B;
  __shared__ __attribute__((aligned(1024))) char lds[2][STAGE];
  typedef unsigned u32x2_ __attribute__((ext_vector_type(2)));

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  int tile = blockIdx.x;
  {
    const int nt = gridDim.x, q = nt >> 3, rem = nt & 7, x = tile & 7, j = tile >> 3;
    tile = (x < rem ? x * (q + 1) : rem * (q + 1) + (x - rem) * q) + j;
  }
  int m0, n0;
  {
    constexpr int GM = GROUP_M / 2;
    const int tilesM = gridDim.x / g.tilesN, per = GM * g.tilesN;
    const int grp = tile / per, rem2 = tile - grp * per, first = grp * GM;
    const int gsz = min(GM, tilesM - first);
    m0 = (first + rem2 % gsz) * BMW;
    n0 = (rem2 / gsz) * BNW;
  }
  // unsplit, unbatched products only (the launcher sees to it)
  const int nk = g.K / BK;

  // ---- staging.  dim-major operand: thread (r = tid >> 2, q = tid & 3) fetches float4 q (4 consecutive k) of row r (A: and r + 128).
  // K-major operand: thread fetches 4 consecutive dims of k row tid >> 6 (A: and + 8) / tid >> 5 (B).
  const float* src[NL];
  int64_t adv[NL];
  int wo[NL];  // byte offset of the thread's 8-byte piece inside a plane
  {
    const int sr = tid >> 2, sq = tid & 3;
    auto dm_off = [&](int row) { return row * 32 + (((sq >> 1) ^ ((row >> 3) & 1)) << 4) + ((sq & 1) << 3); };
    auto km_off = [&](int k, int d, int rowb) { return k * rowb + ((((d >> 4) ^ (2 * (k & 3)))) << 5) + (((d >> 2) & 3) << 3); };
    if (AKM) {
      const int k = tid >> 6, d = 4 * (tid & 63);
      const int dc = min(m0 + d, g.M - 4);
      src[0] = g.A + (int64_t)k * g.lda + dc;
      src[1] = g.A + (int64_t)(k + 8) * g.lda + dc;
      adv[0] = adv[1] = (int64_t)BK * g.lda;
      wo[0] = km_off(k, d, 2 * BMW);
      wo[1] = km_off(k + 8, d, 2 * BMW);
    } else {
      src[0] = g.A + (int64_t)min(m0 + sr, g.M - 1) * g.lda + 4 * sq;
      src[1] = g.A + (int64_t)min(m0 + sr + 128, g.M - 1) * g.lda + 4 * sq;
      adv[0] = adv[1] = BK;
      wo[0] = dm_off(sr);
      wo[1] = dm_off(sr + 128);
    }
    if (BKM) {
      const int k = tid >> 5, d = 4 * (tid & 31);
      src[2] = g.B + (int64_t)k * g.ldb + min(n0 + d, g.N - 4);
      adv[2] = (int64_t)BK * g.ldb;
      wo[2] = km_off(k, d, 2 * BNW);
    } else {
      src[2] = g.B + (int64_t)min(n0 + sr, g.N - 1) * g.ldb + 4 * sq;
      adv[2] = BK;
      wo[2] = dm_off(sr);
    }
  }
  float4 X[NL], Y[NL];
  auto load = [&](float4* ld) {
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      ld[i] = *reinterpret_cast<const float4*>(src[i]);
      src[i] += adv[i];
    }
  };
  auto put = [&](char* base, int plane, int off, const Pl4& p) {
    *reinterpret_cast<u32x2_*>(base + off) = (u32x2_){p.h[0], p.h[1]};
    *reinterpret_cast<u32x2_*>(base + plane + off) = (u32x2_){p.m[0], p.m[1]};
    *reinterpret_cast<u32x2_*>(base + 2 * plane + off) = (u32x2_){p.l[0], p.l[1]};
  };
  // sm: the sign mask (0 or the sign bit) of the step the pieces belong to (SignPhase above): applied to the B tile's four values
  // before they are split -- the pieces of -x are the negated pieces of x (rounding to nearest is symmetric)
  auto put_piece = [&](int buf, int j, const float4* ld, unsigned sm) {  // j = 0, 1: the A tile's two pieces; 2: the B tile's
    if (j < 2) {
      put(lds[buf], PLANE_A, wo[j], split4(ld[j]));
    } else {
      float4 b = ld[2];
      b.x = __uint_as_float(__float_as_uint(b.x) ^ sm);
      b.y = __uint_as_float(__float_as_uint(b.y) ^ sm);
      b.z = __uint_as_float(__float_as_uint(b.z) ^ sm);
      b.w = __uint_as_float(__float_as_uint(b.w) ^ sm);
      put(lds[buf] + OPER_A, PLANE_B, wo[2], split4(b));
    }
  };
  // Sign phases (SignPhase above; r05: the N/N form only, r06: all three).  ph_cur follows the step whose products are formed,
  // ph_nx the step being staged (one ahead); `held` is the sign the accumulators carry.  Cost: 4 v_xor per thread and K step + 64
  // at every second stretch boundary -- nothing once the loop's interleave is stated (SK_PLANES_SCHED below; left to itself the
  // compiler scheduled the r05 flipped instantiation's loop 6 % slower than the plain one).
  SignPhase ph_cur, ph_nx;
  ph_cur.init(g.flip_q, 0);
  ph_nx.init(g.flip_q, 1);
  unsigned held = 0u;
  // ---- fragments of 32 dims starting at d0: lane (l31 = dim, kh = k half)
  const int l31 = lane & 31, kh = lane >> 5;
  auto dm_frag = [&](int row) { return row * 32 + ((kh ^ ((row >> 3) & 1)) << 4); };
  auto km_frag = [&](int d0, int rowb) {  // address of this lane's transposed read: k row 8 kh + q, piece p of 16-dim block b
    const int bq = (lane >> 4) & 1, q = (lane & 15) >> 2, pp = lane & 3;
    return (8 * kh + q) * rowb + ((((d0 >> 4) + bq) ^ (2 * q)) << 5) + 8 * pp;
  };
  int fa[2], fb[NJ];
#pragma unroll
  for (int i = 0; i < 2; ++i) fa[i] = AKM ? km_frag(wm * 64 + 32 * i, 2 * BMW) : dm_frag(wm * 64 + 32 * i + l31);
#pragma unroll
  for (int j = 0; j < NJ; ++j) fb[j] = BKM ? km_frag(wn * 64 + 32 * j, 2 * BNW) : dm_frag(wn * 64 + 32 * j + l31);
  auto rd = [&](const char* pl, int off, bool km, int rowb) -> bf16x8_t {
    if (km) {
      typedef short s16x4 __attribute__((ext_vector_type(4)));
      typedef short s16x8 __attribute__((ext_vector_type(8)));
      const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(pl + off));
      const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(pl + off + 4 * rowb));
      const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      return __builtin_bit_cast(bf16x8_t, v);
    }
    return *reinterpret_cast<const bf16x8_t*>(pl + off);
  };
  auto frag = [&](const char* oper, int plane, int off, bool km, int rowb) {
    Split3 f;
    f.hi = rd(oper, off, km, rowb);
    f.mid = rd(oper + plane, off, km, rowb);
    f.lo = rd(oper + 2 * plane, off, km, rowb);
    return f;
  };

  f32x16 acc[2][NJ];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // the products of the step in stage `buf`; STORE: the next step's values (in nx) are split and written to the other stage
  // BETWEEN the groups of twelve MFMAs -- no dependence between the two, one basic block
  auto step = [&](int buf, int kt, const float4* nx, auto store) {
    const char* ai = lds[buf];
    const char* bi = lds[buf] + OPER_A;
    (void)kt;
    if (ph_cur.mask != held) {  // a stretch of the other sign begins: the accumulators change sign with the products
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = -acc[i][j];
      held = ph_cur.mask;
    }
    const unsigned sg_next = ph_nx.mask;
    ph_cur.advance();
    ph_nx.advance();
    Split3 sa[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) sa[i] = frag(ai, PLANE_A, fa[i], AKM, 2 * BMW);
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const Split3 sb = frag(bi, PLANE_B, fb[j], BKM, 2 * BNW);
#pragma unroll
      for (int i = 0; i < 2; ++i) mma6(acc[i][j], sa[i], sb);
      if constexpr (decltype(store)::value) {  // three pieces over two groups
        put_piece(buf ^ 1, j, nx, sg_next);
        if (j == 1) put_piece(buf ^ 1, 2, nx, sg_next);
      }
    }
    // The interleaving is STATED, not left to the scheduler: one MFMA, then SK_PLANES_SCHED VALU instructions of the next step's
    // split, 24 times (a step has 24 MFMAs and ~70 VALU instructions; LDS reads / writes and the fetches go where the scheduler
    // likes).  Left to itself the compiler found a good interleave for some instantiations and long VALU runs with the MFMA pipe
    // idle for others (the FL form lost 6 % to that, not to its 4 extra instructions).  Measured, one device
    // (profiles/r05_planes_sched.txt): N/N 181 -> 186-188 TFLOP/s WITH the sign phases (172 without the statement), N/T 204.5 ->
    // 210-212, T/N 187 -> 195, 8192^3 215 -> 222; 3 and 4 are equal within the run-to-run spread, 2 loses; a group of 8-24 VALU
    // instructions stated in FRONT of the step's first MFMA (in the shadow of its first fragment reads) loses 1-4 % too.
    if constexpr (SK_PLANES_SCHED > 0 && decltype(store)::value) {
#pragma unroll
      for (int i = 0; i < 24; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, SK_PLANES_SCHED, 0);
      }
    }
  };
  using Yes = std::integral_constant<bool, true>;
  using No = std::integral_constant<bool, false>;

  if (nk > 0) {
    load(X);  // step 0
#pragma unroll
    for (int j = 0; j < NL; ++j) put_piece(0, j, X, 0u);
    if (nk > 1) load(X);  // step 1
    if (nk > 2) load(Y);  // step 2
    __syncthreads();
    int cur = 0, kt = 0;
    // invariant at the top: stage cur holds step kt, X step kt + 1, Y step kt + 2
    while (kt + 4 < nk) {
      step(cur, kt, X, Yes());
      load(X);  // step kt + 3
      __syncthreads();
      cur ^= 1;
      step(cur, kt + 1, Y, Yes());
      load(Y);  // step kt + 4
      __syncthreads();
      cur ^= 1;
      kt += 2;
    }
    for (; kt < nk; ++kt) {  // the last (up to four) steps
      if (kt + 1 < nk) {
        step(cur, kt, X, Yes());
#pragma unroll
        for (int j = 0; j < NL; ++j) X[j] = Y[j];
        if (kt + 3 < nk) load(Y);
      } else {
        step(cur, kt, X, No());
      }
      __syncthreads();
      cur ^= 1;
    }
    if (held) {  // the last stretch held -sum
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = -acc[i][j];
    }
  }
  store_tile(g, acc, g.C, g.ldc, g.bias, false, m0 + wm * 64, n0 + wn * 64, lane);
}


// ------------------------------------------------------------------------------------------------------
// Operands that ARRIVE split (r06; sk_gemm_pl3_tn): the T/N product C[M][N] = sum_k A[k][m] B[k][n] -- every weight gradient of
// the training step: both factors are activation / gradient matrices whose ROWS are the contraction index -- on operands their
// PRODUCERS already cut into the three bf16 pieces (three planes [K][dim] per operand: the backward recurrence writes dgx's with
// the fp32 values, sk_split_rows / sk_hprev_rows the layer inputs' and the recurrent inputs').  gemm_f32_kernel_split3 stages fp32
// tiles and every wave splits the fragments it reads -- each element twice per workgroup, 4.5 VALU instructions each time, ~22 %
// of that kernel's time and, beside a recurrence, VALU slots and matrix-pipe time taken from its host
// (profiles/r05_hosted_split_free_upper_bound.txt: -0.9 ms per training step with the pieces for free).  Here the plane tiles are
// DMA'd into LDS as they lie (24 KB per K step instead of 16: 6 bytes per element instead of 4) and the waves read ready-made
// bf16x8 fragments by the transposed LDS read (ds_read_b64_tr_b16, the planes kernel's K-major image and swizzle) and issue the
// six MFMAs per fragment pair: NO VALU work in the K loop but the sign phases' 24 v_xor in negative stretches.  Same pieces, same
// products, same K order, same sign phases as gemm_f32_kernel_split3: bit-identical results (tests).  128 x 128 tiles, 4 waves of
// 64 x 64, two stages (48 KB of LDS), <= 128 VGPRs: it fits on a CU beside a persistent recurrence workgroup.  Split-K slices and
// batches (the two directions of dW_hh) as in the other kernels.
__global__ __launch_bounds__(256, 2) void gemm_f32_kernel_pl3(GemmArgs g) {
  constexpr int PTILE = BK * BM * 2;  // 4 KB: one plane of one operand tile, [16 k][128 dims] bf16
  constexpr int OPER = 3 * PTILE, STAGE = 2 * OPER;
  constexpr int NST = 2;
  constexpr int ROWB = 2 * BM;        // bytes of a k row in an LDS plane
  __shared__ __attribute__((aligned(1024))) char lds[NST][STAGE];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  int tile = blockIdx.x;
  {
    const int nt = gridDim.x, q = nt >> 3, rem = nt & 7, x = tile & 7, j = tile >> 3;
    tile = (x < rem ? x * (q + 1) : rem * (q + 1) + (x - rem) * q) + j;
  }
  int m0, n0;
  {
    const int tilesM = gridDim.x / g.tilesN, per = GROUP_M * g.tilesN;
    const int grp = tile / per, rem2 = tile - grp * per, first = grp * GROUP_M;
    const int gsz = min(GROUP_M, tilesM - first);
    m0 = (first + rem2 % gsz) * BM;
    n0 = (rem2 / gsz) * BN;
  }
  const int z = blockIdx.z, ks = blockIdx.y;
  const bool partial = g.splitk > 1;
  float* C = partial ? g.slabs + ((int64_t)z * g.splitk + ks) * g.M * g.N : g.C + z * g.sC;
  const int ldc = partial ? g.N : g.ldc;
  const float* bias = (g.bias && !partial) ? g.bias + z * g.sbias : nullptr;
  const int kbeg = ks * g.kchunk, kend = min(g.K, kbeg + g.kchunk);
  const int nk = (kend - kbeg) / BK;

  // ---- staging.  One DMA instruction = 1 KB = four k rows of one plane tile (16 lanes x 16 B per row).  Lane (kr = lane >> 4,
  // j = lane & 15) fetches, for k row 4 kq + kr, the 16 bytes that belong at position j of the row's LDS image: the image keeps
  // the 32-byte chunk c (16 dims) of k row r at chunk c ^ 2 (r & 3) (the transposed fragment reads of a half-wave then fall on 8
  // different bank groups), and r & 3 = kr whatever kq is -- so a lane's offset inside a piece is the same for all 24 pieces of
  // an operand and the piece itself (plane, k quad, K step) is a wave-uniform base address.  Dims past the operand's edge are
  // clamped into its padded row (their products land in columns the epilogue masks).
  const int kr = lane >> 4, jp = lane & 15;
  const int dim = 16 * ((jp >> 1) ^ (2 * kr)) + 8 * (jp & 1);
  const unsigned offA = (unsigned)((kr * g.lda + min(m0 + dim, ((g.M + 7) & ~7) - 8)) * 2);
  const unsigned offB = (unsigned)((kr * g.ldb + min(n0 + dim, ((g.N + 7) & ~7) - 8)) * 2);
  const __bf16* baseA = g.Apl + z * g.sA + (int64_t)kbeg * g.lda;
  const __bf16* baseB = g.Bpl + z * g.sB + (int64_t)kbeg * g.ldb;
  const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)&lds[0][0];
  // this wave's six pieces of a stage: p = wave + 4 i -> operand p / 12, plane (p % 12) / 4, k quad p % 4
  auto stage = [&](int buf, int kt) {
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const int p = wave + 4 * i, op = p / 12, pl = (p % 12) >> 2, kq = p & 3;
      const __bf16* src = (op ? baseB + pl * g.planeB + ((int64_t)kt * BK + 4 * kq) * g.ldb
                              : baseA + pl * g.planeA + ((int64_t)kt * BK + 4 * kq) * g.lda);
      const unsigned dst = lds_base + buf * STAGE + op * OPER + pl * PTILE + kq * 1024;
      asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(op ? offB : offA), "s"(src), "s"(dst) : "memory");
    }
  };

  // ---- fragments: the planes kernel's K-major read (k row 8 kh + q, piece pp of 16-dim block bq; + 4 rows for the upper half)
  const int kh = lane >> 5;
  auto km_frag = [&](int d0) {
    const int bq = (lane >> 4) & 1, q = (lane & 15) >> 2, pp = lane & 3;
    return (8 * kh + q) * ROWB + ((((d0 >> 4) + bq) ^ (2 * q)) << 5) + 8 * pp;
  };
  int fa[2], fb[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    fa[i] = km_frag(wm * 64 + 32 * i);
    fb[i] = km_frag(wn * 64 + 32 * i);
  }
  auto rd = [&](const char* pl, int off) -> bf16x8_t {
    typedef short s16x4 __attribute__((ext_vector_type(4)));
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(pl + off));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(pl + off + 4 * ROWB));
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8_t, v);
  };
  auto frag = [&](const char* oper, int off, unsigned sm) {
    typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));
    Split3 f;
    f.hi = rd(oper, off);
    f.mid = rd(oper + PTILE, off);
    f.lo = rd(oper + 2 * PTILE, off);
    if (sm) {  // a negative stretch (wave-uniform): the B operand's pieces change sign
      const u32x4_ m4 = {sm, sm, sm, sm};
      f.hi = __builtin_bit_cast(bf16x8_t, __builtin_bit_cast(u32x4_, f.hi) ^ m4);
      f.mid = __builtin_bit_cast(bf16x8_t, __builtin_bit_cast(u32x4_, f.mid) ^ m4);
      f.lo = __builtin_bit_cast(bf16x8_t, __builtin_bit_cast(u32x4_, f.lo) ^ m4);
    }
    return f;
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  auto negate_acc = [&]() {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = -acc[i][j];
  };

  if (nk > 0) stage(0, 0);
  SignPhase ph;
  ph.init(g.flip_q, kbeg / BK);
  unsigned held = 0u;
  int cur = 0;
  for (int kt = 0; kt < nk; ++kt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // everybody's pieces of step kt have landed; all reads of the buffer refilled next are done
    if (kt + 1 < nk) stage(cur ^ 1, kt + 1);
    if (ph.mask != held) {
      negate_acc();
      held = ph.mask;
    }
    const char* ai = lds[cur];
    const char* bi = lds[cur] + OPER;
    const unsigned sm = held ? 0x80008000u : 0u;
    const Split3 a0 = frag(ai, fa[0], 0u), b0 = frag(bi, fb[0], sm);
    mma6(acc[0][0], a0, b0);
    const Split3 a1 = frag(ai, fa[1], 0u);
    mma6(acc[1][0], a1, b0);
    const Split3 b1 = frag(bi, fb[1], sm);
    mma6(acc[0][1], a0, b1);
    mma6(acc[1][1], a1, b1);
    ph.advance();
    cur ^= 1;
  }
  if (held) negate_acc();
  store_tile(g, acc, C, ldc, bias, partial, m0 + wm * 64, n0 + wn * 64, lane);
  if (partial && g.counters) finish_splitk(g, z, m0 + wm * 64, n0 + wn * 64, tid);
}

// The three bf16 planes of an fp32 matrix (sk_split_rows): dst[p][r][c], p = 0 hi, 1 mid, 2 lo, the pieces split3() makes (round
// to nearest even); rows R .. R_pad - 1 and columns C .. ld_dst - 1 are written as zeros (a K-major factor is read in whole K
// steps; a clamped tile edge reads the padding).  HBM-bound: 4 bytes read, 6 written per element.
__global__ __launch_bounds__(256) void split_rows_kernel(const float* __restrict__ src, int R, int C, int ld_src, __bf16* __restrict__ dst,
                                                         int ld_dst, int R_pad, int64_t plane) {
  const int64_t n4 = (int64_t)R_pad * (ld_dst / 4);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const int r = (int)(i / (ld_dst / 4)), c = 4 * (int)(i % (ld_dst / 4));
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r < R) {
      const float* sp = src + (int64_t)r * ld_src + c;
      if (c + 3 < C && (((uintptr_t)sp) & 15) == 0) {
        v = *reinterpret_cast<const float4*>(sp);
      } else {
        if (c < C) v.x = sp[0];
        if (c + 1 < C) v.y = sp[1];
        if (c + 2 < C) v.z = sp[2];
        if (c + 3 < C) v.w = sp[3];
      }
    }
    const Pl4 p = split4(v);
    typedef unsigned u32x2_ __attribute__((ext_vector_type(2)));
    __bf16* d = dst + (int64_t)r * ld_dst + c;
    *reinterpret_cast<u32x2_*>(d) = (u32x2_){p.h[0], p.h[1]};
    *reinterpret_cast<u32x2_*>(d + plane) = (u32x2_){p.m[0], p.m[1]};
    *reinterpret_cast<u32x2_*>(d + 2 * plane) = (u32x2_){p.l[0], p.l[1]};
  }
}

// ------------------------------------------------------------------------------------------------------
// bf16-input variant (BASELINE configs[3]: "bf16 MFMA inputs, fp32 accumulate").  Operands stay fp32 in
// HBM -- no second copy of weights or activations exists -- and are rounded to bf16 (RNE) on their way
// into LDS; the matrix cores run v_mfma_f32_32x32x16_bf16 with fp32 accumulators, and the epilogue, the
// split-K slabs and their fixed-order reduction are the fp32 kernel's.  Same 128x128 block / 2x2 waves /
// 64x64 per wave; K step 32.  Both operands are kept [tile row][32 k] in LDS (64 B per row), which is the
// fragment shape of the bf16 MFMA (lane: row l&31, eight consecutive k at 8*(l>>5)); a k-major operand
// (transA / !transB) is transposed in registers by the thread that fetched its 4 k x 4 row patch.
// LDS image swizzle (no padding): row r, 16-byte chunk c  ->  256 B * (r>>2) + 64 B * ((r&3) ^ ((r>>4)&3))
// + 16 B * (c ^ ((r>>2)&3)): fragment reads (16 rows x one chunk) and row-wise writes are conflict-free,
// patch-transposed writes are 2-way.
namespace bf {

constexpr int BK = 32;
constexpr int PIECES = BK * BM / 4 / 256;  // 4 float4 per thread per operand tile
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int lds_off(int r, int c) {  // byte offset of chunk c of row r
  return ((r >> 2) << 8) + ((((r & 3) ^ ((r >> 4) & 3))) << 6) + ((c ^ ((r >> 2) & 3)) << 4);
}

__device__ __forceinline__ bf16x4 pack4(float a, float b, float c, float d) {
  bf16x4 v;
  v[0] = (__bf16)a;
  v[1] = (__bf16)b;
  v[2] = (__bf16)c;
  v[3] = (__bf16)d;
  return v;
}

// KMAJOR == false: operand stored [tile dim][K]: piece i = row idx/8, k (idx%8)*4 .. +3, idx = tid + 256 i
// KMAJOR == true : operand stored [K][tile dim]: piece i = k 4*(tid>>5) + i, rows (tid&31)*4 .. +3
template <bool KMAJOR>
__device__ __forceinline__ void fetch(const float* __restrict__ P, int ld, int dim0, int dimLimit, int k0, int K,
                                      bool fast, int tid, float4 (&r)[PIECES]) {
#pragma unroll
  for (int i = 0; i < PIECES; ++i) {
    int row, col, rowLimit, colLimit;  // element (row, col..col+3) of the stored matrix
    if (KMAJOR) {
      row = k0 + 4 * (tid >> 5) + i;
      col = dim0 + (tid & 31) * 4;
      rowLimit = K;
      colLimit = dimLimit;
    } else {
      const int idx = tid + 256 * i;
      row = dim0 + (idx >> 3);
      col = k0 + (idx & 7) * 4;
      rowLimit = dimLimit;
      colLimit = K;
    }
    if (fast) {
      r[i] = *reinterpret_cast<const float4*>(P + (int64_t)row * ld + col);
    } else {
      const bool rok = row < rowLimit;
      const float* q = P + (int64_t)min(row, rowLimit - 1) * ld;
      const int cmax = colLimit - 1;
      const float v0 = q[min(col + 0, cmax)], v1 = q[min(col + 1, cmax)];
      const float v2 = q[min(col + 2, cmax)], v3 = q[min(col + 3, cmax)];
      r[i].x = (rok && col + 0 < colLimit) ? v0 : 0.f;
      r[i].y = (rok && col + 1 < colLimit) ? v1 : 0.f;
      r[i].z = (rok && col + 2 < colLimit) ? v2 : 0.f;
      r[i].w = (rok && col + 3 < colLimit) ? v3 : 0.f;
    }
  }
}

template <bool KMAJOR>
__device__ __forceinline__ void stash(char* S, int tid, const float4 (&r)[PIECES]) {
  if (KMAJOR) {
    const int kq = tid >> 5, r0 = (tid & 31) * 4;  // k = 4 kq .. +3 -> chunk kq>>1, half kq&1
    *reinterpret_cast<bf16x4*>(S + lds_off(r0 + 0, kq >> 1) + (kq & 1) * 8) = pack4(r[0].x, r[1].x, r[2].x, r[3].x);
    *reinterpret_cast<bf16x4*>(S + lds_off(r0 + 1, kq >> 1) + (kq & 1) * 8) = pack4(r[0].y, r[1].y, r[2].y, r[3].y);
    *reinterpret_cast<bf16x4*>(S + lds_off(r0 + 2, kq >> 1) + (kq & 1) * 8) = pack4(r[0].z, r[1].z, r[2].z, r[3].z);
    *reinterpret_cast<bf16x4*>(S + lds_off(r0 + 3, kq >> 1) + (kq & 1) * 8) = pack4(r[0].w, r[1].w, r[2].w, r[3].w);
  } else {
#pragma unroll
    for (int i = 0; i < PIECES; ++i) {
      const int idx = tid + 256 * i, row = idx >> 3, q = idx & 7;
      *reinterpret_cast<bf16x4*>(S + lds_off(row, q >> 1) + (q & 1) * 8) = pack4(r[i].x, r[i].y, r[i].z, r[i].w);
    }
  }
}

template <bool TA, bool TB>
__global__ __launch_bounds__(256, 3) void gemm_bf16_kernel(GemmArgs g) {
  constexpr int TILE = BM * BK * 2;  // bytes of one operand image
  __shared__ __attribute__((aligned(256))) char As[2][TILE];
  __shared__ __attribute__((aligned(256))) char Bs[2][TILE];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  int tile = blockIdx.x;
  {
    const int nt = gridDim.x, q = nt >> 3, rem = nt & 7, x = tile & 7, j = tile >> 3;
    tile = (x < rem ? x * (q + 1) : rem * (q + 1) + (x - rem) * q) + j;
  }
  // within a run, walk groups of GROUP_M tile rows column by column: the ~128 tiles an XCD has in flight
  // then cover about GROUP_M x 16 tiles and share 8 A panels and 16 B panels in its 4 MB L2, instead of
  // 2-3 A panels and every B panel of the matrix
  int m0, n0;
  {
    const int tilesM = gridDim.x / g.tilesN, per = GROUP_M * g.tilesN;
    const int grp = tile / per, rem2 = tile - grp * per, first = grp * GROUP_M;
    const int gsz = min(GROUP_M, tilesM - first);
    m0 = (first + rem2 % gsz) * BM;
    n0 = (rem2 / gsz) * BN;
  }
  const bool fullA = g.vecA && (m0 + BM <= g.M), fullB = g.vecB && (n0 + BN <= g.N);
  const int z = blockIdx.z, ks = blockIdx.y;
  const float* A = g.A + z * g.sA;
  const float* B = g.B + z * g.sB;
  const bool partial = g.splitk > 1;
  float* C = partial ? g.slabs + ((int64_t)z * g.splitk + ks) * g.M * g.N : g.C + z * g.sC;
  const int ldc = partial ? g.N : g.ldc;
  const float* bias = (g.bias && !partial) ? g.bias + z * g.sbias : nullptr;
  const int kbeg = ks * g.kchunk, kend = min(g.K, kbeg + g.kchunk);  // kchunk is a multiple of BK

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nk = (kend - kbeg + BK - 1) / BK;
  float4 ra[PIECES], rb[PIECES];
  fetch<TA>(A, g.lda, m0, g.M, kbeg, kend, fullA && kbeg + BK <= kend, tid, ra);
  fetch<!TB>(B, g.ldb, n0, g.N, kbeg, kend, fullB && kbeg + BK <= kend, tid, rb);
  stash<TA>(As[0], tid, ra);
  stash<!TB>(Bs[0], tid, rb);
  __syncthreads();

  const int kh = lane >> 5, l31 = lane & 31;
  // fragment byte offsets: k step s uses chunk 2 s + kh; the swizzle makes step 1 = step 0 ^ 32
  const int oa0 = lds_off(wm * 64 + l31, kh), oa1 = lds_off(wm * 64 + 32 + l31, kh);
  const int ob0 = lds_off(wn * 64 + l31, kh), ob1 = lds_off(wn * 64 + 32 + l31, kh);
  int cur = 0;
  for (int kt = 0; kt < nk; ++kt) {
    const bool more = kt + 1 < nk;
    if (more) {
      const int k0 = kbeg + (kt + 1) * BK;
      const bool kfull = k0 + BK <= kend;  // block-uniform
      fetch<TA>(A, g.lda, m0, g.M, k0, kend, fullA && kfull, tid, ra);
      fetch<!TB>(B, g.ldb, n0, g.N, k0, kend, fullB && kfull, tid, rb);
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(As[cur] + (oa0 ^ (s << 5)));
      const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(As[cur] + (oa1 ^ (s << 5)));
      const bf16x8 b0 = *reinterpret_cast<const bf16x8*>(Bs[cur] + (ob0 ^ (s << 5)));
      const bf16x8 b1 = *reinterpret_cast<const bf16x8*>(Bs[cur] + (ob1 ^ (s << 5)));
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[1][1], 0, 0, 0);
    }
    if (more) {
      stash<TA>(As[cur ^ 1], tid, ra);
      stash<!TB>(Bs[cur ^ 1], tid, rb);
    }
    __syncthreads();
    cur ^= 1;
  }

#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = n0 + wn * 64 + j * 32 + l31;
      if (col >= g.N) continue;
      const float bv = bias ? bias[col] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
        if (row < g.M) {
          float* cp = C + (int64_t)row * ldc + col;
          float v = acc[i][j][r] + bv;
          if (!partial) {
            if (g.accumulate) v += *cp;
            if (g.act == 1) v = sk_sigmoid(v);
            *cp = v;
          } else {
            __hip_atomic_store(cp, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // write-through: read by another XCD
          }
        }
      }
    }
}

}  // namespace bf

// ------------------------------------------------------------------------------------------------------
// bf16 operands IN MEMORY (r02).  C[M,N] = act(A[M,K] * B[N,K]^T + bias (+ C)) with A and B stored as bf16, both
// K-contiguous ("NT"): the producers of the operands (sk_cast_bf16 / sk_cast_bf16_t; weights once per step) write
// bf16 copies -- transposed where the product needs it, so that every product of the network is this one form --
// and the GEMM moves half the bytes per operand element through L2 -> LDS, which is what capped the kernel above
// (fp32 operands rounded on the way into LDS: one ds_read_b128 per MFMA and 2x the L2 traffic).
// Tile 256 x BN (BN = 256 or 128), K step 64, 8 waves as 2 (M) x 4 (N), wave tile 128 x BN/4 = 8 x BN/64 MFMA tiles
// of v_mfma_f32_16x16x32_bf16; operands go global -> LDS by LDS-DMA (global_load_lds_dwordx4: no VGPR staging), two
// LDS buffers, one barrier per K step: the DMA of step k+1 flies under the MFMAs of step k.
// LDS image of a tile (rows of 64 bf16 = 128 B = 8 chunks of 16 B): chunk c of row r sits at
//     256 * (r >> 1) + 16 * ((8 * (r & 1) + c) ^ ((r >> 1) & 7))
// i.e. two tile rows share a 256-B bank row and the 16 slots of a bank row are XOR-rotated by the bank-row index, so
// the 16 rows x one chunk that a ds_read_b128 fragment read touches fall into 16 different slots.  LDS-DMA writes
// LDS lane-linearly, so the permutation is applied to the per-lane SOURCE address (cdna_hip_programming.md rule 21).
namespace bf2 {

constexpr int BM = 256, BK = 64, NT = 512;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct Args {
  const __bf16* A;
  const __bf16* B;
  float* C;
  const float* bias;
  int M, N, K, lda, ldb, ldc;
  int accumulate, act;
  int tilesN;
  int splitk, kchunk;  // K range of slice y: [y * kchunk, min(K, (y + 1) * kchunk)), kchunk a multiple of BK
  float* slabs;
  int64_t sA, sB, sC, sbias;
  unsigned* counters;     // stream-K kernel: one ticket counter per tile of the cut (head of the workspace)
  int sk_tiles, sk_full;  // stream-K kernel: tiles of the product, data-parallel rounds before the stream-K region
};

// byte offset of (row r, 16-byte chunk c) in a tile image
__device__ __forceinline__ int img(int r, int c) { return ((r >> 1) << 8) + (((((r & 1) << 3) | c) ^ ((r >> 1) & 7)) << 4); }

// One 1 KB piece (8 tile rows) of an operand tile: which (row, chunk) lane `lane` must fetch so that the lane-linear
// LDS write of the DMA produces the image above.
__device__ __forceinline__ void piece_src(int piece, int lane, int& row, int& chunk) {
  const int r2 = piece * 4 + (lane >> 4);  // bank row = tile row pair
  const int v = (lane & 15) ^ (r2 & 7);
  row = 2 * r2 + (v >> 3);
  chunk = v & 7;
}

// ---- K-major operands ([K][dim] in memory, dim contiguous: an activation or gradient matrix used as the TRANSPOSED
// factor of a weight-gradient product, or a weight matrix in a data-gradient product) need no transposed copy: the tile
// is DMA'd as it lies -- 64 k rows of DIM elements -- and the MFMA fragments (8 consecutive k of one dim per lane) are
// gathered by ds_read_b64_tr_b16, the hardware-transposed LDS read: a 16-lane group reads a block of 4 k rows x 16 dims
// and lane i gets dim i of the 4 rows; two such reads make one bf16x8 operand (same LDS cycles as one ds_read_b128).
// Image: row r of the tile at r * 2 DIM bytes; inside a row the 32-byte chunk c (16 dims) sits at chunk c ^ f(r),
// f(r) = (r & 3) | (((r >> 3) & 1) << 2): the 8 row segments a half-wave reads in one instruction (k rows 8o + q of the
// two lane groups o, q = 0..3, same 16 dims) then fall on the 8 different 32-byte bank groups (conflict-free).  The DMA
// writes lane-linearly, so the permutation is applied to the per-lane SOURCE address.
template <int DIM>
__device__ __forceinline__ void kmaj_piece_src(int piece, int lane, int& row, int& dim) {
  constexpr int ROWB = 2 * DIM;
  const int off = piece * 1024 + 16 * lane;
  row = off / ROWB;
  const int c16 = (off % ROWB) >> 4;
  const int c32 = (c16 >> 1) ^ ((row & 3) | (((row >> 3) & 1) << 2));
  dim = c32 * 16 + (c16 & 1) * 8;
}
// byte offset of this lane's transposed-read address for the 16-dim block `c32` of the tile, k rows 8 (lane>>4) + q
// (add 32 s * ROWB for the K half s and 4 * ROWB for the second read of a fragment)
template <int DIM>
__device__ __forceinline__ int kmaj_frag_addr(int c32, int lane) {
  const int o = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
  const int f = q | ((o & 1) << 2);
  return (8 * o + q) * (2 * DIM) + ((c32 ^ f) << 5) + 8 * p;
}
__device__ __forceinline__ bf16x8 kmaj_frag(const char* tile, int addr, int rowb) {
  typedef short s16x4 __attribute__((ext_vector_type(4)));
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(tile + addr));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(tile + addr + 4 * rowb));
  const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, v);
}

// AKM / BKM: the operand is K-major in memory (A stored [K][M], lda = row stride of a k row; B stored [K][N]).
template <int BN, bool AKM = false, bool BKM = false>
__global__ __launch_bounds__(NT, 2) void gemm_bf16_nt_kernel(Args g) {
  constexpr int TM = 8, TN = BN / 64;            // MFMA tiles per wave
  constexpr int PA = BM / 8 / 8, PB = BN / 8 / 8;  // 1 KB pieces per wave per K step (A: 4, B: 4 or 2)
  constexpr int ABYTES = BM * BK * 2, BBYTES = BN * BK * 2;
  __shared__ __attribute__((aligned(1024))) char lds[2 * (ABYTES + BBYTES)];

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int wm = w >> 2, wn = w & 3;
  int tile = blockIdx.x;
  {
    const int nt = gridDim.x, q = nt >> 3, rem = nt & 7, x = tile & 7, j = tile >> 3;
    tile = (x < rem ? x * (q + 1) : rem * (q + 1) + (x - rem) * q) + j;
  }
  int m0, n0;
  {
    const int tilesM = gridDim.x / g.tilesN, per = GROUP_M * g.tilesN;
    const int grp = tile / per, rem2 = tile - grp * per, first = grp * GROUP_M;
    const int gsz = min(GROUP_M, tilesM - first);
    m0 = (first + rem2 % gsz) * BM;
    n0 = (rem2 / gsz) * BN;
  }
  const int z = blockIdx.z, ks = blockIdx.y;
  const __bf16* A = g.A + z * g.sA;
  const __bf16* B = g.B + z * g.sB;
  const bool partial = g.splitk > 1;
  float* C = partial ? g.slabs + ((int64_t)z * g.splitk + ks) * g.M * g.N : g.C + z * g.sC;
  const int ldc = partial ? g.N : g.ldc;
  const float* bias = (g.bias && !partial) ? g.bias + z * g.sbias : nullptr;
  const int kbeg = ks * g.kchunk, kend = min(g.K, kbeg + g.kchunk);
  const int nk = (kend - kbeg) / BK;

  // per-lane DMA sources (rows past the matrix edge are clamped: their products land in rows/cols never stored)
  const __bf16* srcA[PA];
  const __bf16* srcB[PB];
#pragma unroll
  for (int i = 0; i < PA; ++i) {
    int r, c;
    if (AKM) {  // r = k row of the tile, c = first of 8 dims; dims past the padded width are clamped (never stored)
      kmaj_piece_src<BM>(w * PA + i, lane, r, c);
      srcA[i] = A + (int64_t)(kbeg + r) * g.lda + min(m0 + c, g.lda - 8);
    } else {
      piece_src(w * PA + i, lane, r, c);
      srcA[i] = A + (int64_t)min(m0 + r, g.M - 1) * g.lda + kbeg + c * 8;
    }
  }
#pragma unroll
  for (int i = 0; i < PB; ++i) {
    int r, c;
    if (BKM) {
      kmaj_piece_src<BN>(w * PB + i, lane, r, c);
      srcB[i] = B + (int64_t)(kbeg + r) * g.ldb + min(n0 + c, g.ldb - 8);
    } else {
      piece_src(w * PB + i, lane, r, c);
      srcB[i] = B + (int64_t)min(n0 + r, g.N - 1) * g.ldb + kbeg + c * 8;
    }
  }
  const int64_t stepA = AKM ? (int64_t)BK * g.lda : BK, stepB = BKM ? (int64_t)BK * g.ldb : BK;
  // K-major instantiations issue the DMA as inline assembly: behind the LDS-DMA BUILTIN hipcc puts an s_waitcnt vmcnt(0) in
  // front of the transposed LDS reads (it cannot prove them disjoint from the DMA's destination), which serialises the next
  // K step's fetch with this step's products (measured: 20-30 % slower).  The kernel orders DMA and reads itself (vmcnt(0)
  // + barrier at the top of every K step).  The NT instantiation keeps the builtin (its plain loads are proven disjoint).
  const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)lds;
  const unsigned pieceA = __builtin_amdgcn_readfirstlane(lds_base + w * PA * 1024);
  const unsigned pieceB = __builtin_amdgcn_readfirstlane(lds_base + ABYTES + w * PB * 1024);
  auto stage = [&](int buf, int kt) {
    char* a = lds + buf * (ABYTES + BBYTES);
    char* b = a + ABYTES;
#pragma unroll
    for (int i = 0; i < PA; ++i) {
      if (AKM || BKM)
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(srcA[i] + kt * stepA),
                     "s"(pieceA + buf * (ABYTES + BBYTES) + i * 1024)
                     : "memory");
      else
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(srcA[i] + kt * stepA),
                                         (__attribute__((address_space(3))) void*)(a + (w * PA + i) * 1024), 16, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < PB; ++i) {
      if (AKM || BKM)
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(srcB[i] + kt * stepB),
                     "s"(pieceB + buf * (ABYTES + BBYTES) + i * 1024)
                     : "memory");
      else
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(srcB[i] + kt * stepB),
                                         (__attribute__((address_space(3))) void*)(b + (w * PB + i) * 1024), 16, 0, 0);
    }
  };

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // fragment addresses: lane holds row (lane & 15) of a 16-row tile, 8 consecutive k at chunk (lane >> 4) + 4 * kstep;
  // a tile further down adds 16 rows = 2048 B (the rotation repeats every 16 rows), kstep 1 flips slot bit 2 (^ 64 B)
  const int fa = img(wm * 128 + (lane & 15), lane >> 4);
  const int fb = img(wn * (BN / 4) + (lane & 15), lane >> 4);
  // K-major operands: one transposed-read address per 16-dim block of the wave's tile (the chunk rotation is per block)
  int ka[AKM ? TM : 1], kb[BKM ? TN : 1];
  if (AKM) {
#pragma unroll
    for (int i = 0; i < TM; ++i) ka[i] = kmaj_frag_addr<BM>(wm * TM + i, lane);
  }
  if (BKM) {
#pragma unroll
    for (int j = 0; j < TN; ++j) kb[j] = kmaj_frag_addr<BN>(wn * TN + j, lane);
  }

  if (nk > 0) stage(0, 0);
  int cur = 0;
  for (int kt = 0; kt < nk; ++kt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's pieces of step kt have landed ...
    __syncthreads();                                   // ... and everybody's; all reads of the other buffer are done
    if (kt + 1 < nk) stage(cur ^ 1, kt + 1);
    const char* a = lds + cur * (ABYTES + BBYTES);
    const char* b = a + ABYTES;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8 af[TM], bfr[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        if (AKM)
          af[i] = kmaj_frag(a, ka[i] + s * 32 * (2 * BM), 2 * BM);
        else
          af[i] = *reinterpret_cast<const bf16x8*>(a + ((fa ^ (s << 6)) + i * 2048));
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        if (BKM)
          bfr[j] = kmaj_frag(b, kb[j] + s * 32 * (2 * BN), 2 * BN);
        else
          bfr[j] = *reinterpret_cast<const bf16x8*>(b + ((fb ^ (s << 6)) + j * 2048));
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    }
    cur ^= 1;
  }

  // epilogue: C/D layout of the 16x16 MFMA: col = lane & 15, row = 4 * (lane >> 4) + reg
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int col = n0 + wn * (BN / 4) + j * 16 + (lane & 15);
    if (col >= g.N) continue;
    const float bv = bias ? bias[col] : 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = m0 + wm * 128 + i * 16 + 4 * (lane >> 4) + r;
        if (row < g.M) {
          float* cp = C + (int64_t)row * ldc + col;
          float v = acc[i][j][r] + bv;
          if (!partial) {
            if (g.accumulate) v += *cp;
            if (g.act == 1) v = sk_sigmoid(v);
          }
          *cp = v;
        }
      }
  }
}

// The 256 x 256 instantiation as a PERSISTENT stream-K kernel (r03): the schedule, piece / ticket protocol and fix-up of
// gemm_f32_kernel_streamk above (one workgroup per CU; whole rounds of tiles data-parallel, the K steps of the remaining
// tiles laid end to end and cut into gridDim.x equal ranges; pieces through slabs, last ticket adds them in workgroup
// order), on this kernel's images, fragments and two LDS buffers.  It replaces split-K for the unbatched products with
// few tiles (data gradients: 350 tiles, weight gradients: 196): no K slabs of the whole matrix, no second launch.
template <bool AKM, bool BKM>
__global__ __launch_bounds__(NT, 2) void gemm_bf16_streamk_kernel(Args g) {
  constexpr int BN = 256, TM = 8, TN = 4, PA = 4, PB = 4;
  constexpr int ABYTES = BM * BK * 2, BBYTES = BN * BK * 2, STAGE = ABYTES + BBYTES;
  constexpr int SLAB = BM * BN;
  __shared__ __attribute__((aligned(1024))) char lds[2 * STAGE];
  __shared__ int s_fix[4];

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int wm = w >> 2, wn = w & 3;
  const int P = gridDim.x, b = blockIdx.x;
  const int nk = g.K / BK, nt = g.sk_tiles, full = g.sk_full;
  const int64_t R = (int64_t)(nt - full * P) * nk;
  const int64_t beg = b * R / P, end = (b + 1) * R / P;
  const int t0 = (int)(beg / nk);
  const int nseg = full + (end > beg ? (int)((end - 1) / nk) - t0 + 1 : 0);

  auto segment = [&](int i, int& v, int& kb, int& ke) {
    if (i < full) {
      v = i * P + b; kb = 0; ke = nk;
    } else {
      const int t = t0 + (i - full);
      const int64_t lo = (int64_t)t * nk;
      v = full * P + t;
      kb = (int)(max(beg, lo) - lo);
      ke = (int)(min(end, lo + nk) - lo);
    }
  };
  auto origin = [&](int v, int& m0, int& n0) {
    const int q = nt >> 3, rem = nt & 7, x = v & 7, j = v >> 3;
    const int tile = (x < rem ? x * (q + 1) : rem * (q + 1) + (x - rem) * q) + j;
    const int tilesM = nt / g.tilesN, per = GROUP_M * g.tilesN;
    const int grp = tile / per, rem2 = tile - grp * per, first = grp * GROUP_M;
    const int gsz = min(GROUP_M, tilesM - first);
    m0 = (first + rem2 % gsz) * BM;
    n0 = (rem2 / gsz) * BN;
  };

  // DMA cursor: one K step ahead of the products, across segment boundaries
  const int64_t stepA = AKM ? (int64_t)BK * g.lda : BK, stepB = BKM ? (int64_t)BK * g.ldb : BK;
  const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)lds;
  const unsigned pieceA = __builtin_amdgcn_readfirstlane(lds_base + w * PA * 1024);
  const unsigned pieceB = __builtin_amdgcn_readfirstlane(lds_base + ABYTES + w * PB * 1024);
  const __bf16* srcA[PA];
  const __bf16* srcB[PB];
  int di = 0, dk = 0, dke = 0;
  auto dma_open = [&]() {
    int v, kb, ke, m0, n0;
    segment(di, v, kb, ke);
    origin(v, m0, n0);
    const int kbeg = kb * BK;
#pragma unroll
    for (int i = 0; i < PA; ++i) {
      int r, c;
      if (AKM) {
        kmaj_piece_src<BM>(w * PA + i, lane, r, c);
        srcA[i] = g.A + (int64_t)(kbeg + r) * g.lda + min(m0 + c, g.lda - 8);
      } else {
        piece_src(w * PA + i, lane, r, c);
        srcA[i] = g.A + (int64_t)min(m0 + r, g.M - 1) * g.lda + kbeg + c * 8;
      }
    }
#pragma unroll
    for (int i = 0; i < PB; ++i) {
      int r, c;
      if (BKM) {
        kmaj_piece_src<BN>(w * PB + i, lane, r, c);
        srcB[i] = g.B + (int64_t)(kbeg + r) * g.ldb + min(n0 + c, g.ldb - 8);
      } else {
        piece_src(w * PB + i, lane, r, c);
        srcB[i] = g.B + (int64_t)min(n0 + r, g.N - 1) * g.ldb + kbeg + c * 8;
      }
    }
    dk = kb;
    dke = ke;
  };
  auto fetch = [&](int buf) {
    if (di >= nseg) return;
#pragma unroll
    for (int i = 0; i < PA; ++i) {
      asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(srcA[i]), "s"(pieceA + buf * STAGE + i * 1024)
                   : "memory");
      srcA[i] += stepA;
    }
#pragma unroll
    for (int i = 0; i < PB; ++i) {
      asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(srcB[i]), "s"(pieceB + buf * STAGE + i * 1024)
                   : "memory");
      srcB[i] += stepB;
    }
    if (++dk == dke && ++di < nseg) dma_open();
  };

  f32x4 acc[TM][TN];
  auto clear = [&]() {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  };
  clear();

  const int fa = img(wm * 128 + (lane & 15), lane >> 4);
  const int fb = img(wn * (BN / 4) + (lane & 15), lane >> 4);
  int ka[AKM ? TM : 1], kb_[BKM ? TN : 1];
  if (AKM) {
#pragma unroll
    for (int i = 0; i < TM; ++i) ka[i] = kmaj_frag_addr<BM>(wm * TM + i, lane);
  }
  if (BKM) {
#pragma unroll
    for (int j = 0; j < TN; ++j) kb_[j] = kmaj_frag_addr<BN>(wn * TN + j, lane);
  }

  const int64_t F = (int64_t)full * nk + (end - beg);
  if (nseg > 0) dma_open();
  fetch(0);
  int ci = 0, cv = 0, ck = 0, cke = 0;
  if (nseg > 0) segment(0, cv, ck, cke);
  int cur = 0;
  for (int64_t f = 0; f < F; ++f) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's pieces of step f have landed (and an epilogue's stores) ...
    __syncthreads();                                   // ... and everybody's; all reads of the other buffer are done
    fetch(cur ^ 1);
    const char* a = lds + cur * STAGE;
    const char* bt = a + ABYTES;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8 af[TM], bfr[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        if (AKM)
          af[i] = kmaj_frag(a, ka[i] + s * 32 * (2 * BM), 2 * BM);
        else
          af[i] = *reinterpret_cast<const bf16x8*>(a + ((fa ^ (s << 6)) + i * 2048));
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        if (BKM)
          bfr[j] = kmaj_frag(bt, kb_[j] + s * 32 * (2 * BN), 2 * BN);
        else
          bfr[j] = *reinterpret_cast<const bf16x8*>(bt + ((fb ^ (s << 6)) + j * 2048));
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    }
    cur ^= 1;
    if (++ck < cke) continue;

    // ---- end of a segment.  C/D layout of the 16x16 MFMA: col = lane & 15, row = 4 * (lane >> 4) + reg
    int m0, n0;
    origin(cv, m0, n0);
    const int col0 = n0 + wn * (BN / 4) + (lane & 15), row0 = m0 + wm * 128 + 4 * (lane >> 4);
    if (ci < full) {
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int col = col0 + j * 16;
        if (col >= g.N) continue;
        const float bv = g.bias ? g.bias[col] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = row0 + i * 16 + r;
            if (row < g.M) {
              float* cp = g.C + (int64_t)row * g.ldc + col;
              float v = acc[i][j][r] + bv;
              if (g.accumulate) v += *cp;
              if (g.act == 1) v = sk_sigmoid(v);
              *cp = v;
            }
          }
      }
    } else {
      const int t = cv - full * P;
      float* mine = g.slabs + (size_t)(2 * b + (t - t0)) * SLAB + tid;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            __hip_atomic_store(mine + ((i * TN + j) * 4 + r) * 512, acc[i][j][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) {
        const int w0 = (int)((((int64_t)t * nk + 1) * P - 1) / R), w1 = (int)((((int64_t)(t + 1) * nk) * P - 1) / R);
        const unsigned old = __hip_atomic_fetch_add(g.counters + t, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = old == (unsigned)(w1 - w0);
        if (last) __hip_atomic_store(g.counters + t, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_fix[0] = last; s_fix[1] = w0; s_fix[2] = w1;
        s_fix[3] = (w0 * R / P < (int64_t)t * nk) ? 1 : 0;
      }
      __syncthreads();
      if (s_fix[0]) {
        const int w0 = s_fix[1], w1 = s_fix[2];
        const float* sl0 = g.slabs + (size_t)(2 * w0 + s_fix[3]) * SLAB + tid;
        for (int q0 = 0; q0 < 128; q0 += 32) {
          float v[32];
#pragma unroll
          for (int u = 0; u < 32; ++u) v[u] = __hip_atomic_load(sl0 + (q0 + u) * 512, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          for (int ww = w0 + 1; ww <= w1; ++ww) {
            const float* sl = g.slabs + (size_t)(2 * ww) * SLAB + tid;
#pragma unroll
            for (int u = 0; u < 32; ++u) v[u] += __hip_atomic_load(sl + (q0 + u) * 512, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
#pragma unroll
          for (int u = 0; u < 32; ++u) {
            const int q = q0 + u, i = q >> 4, j = (q >> 2) & 3, r = q & 3;
            const int col = col0 + j * 16, row = row0 + i * 16 + r;
            if (col < g.N && row < g.M) {
              float* cp = g.C + (int64_t)row * g.ldc + col;
              float x = v[u];
              if (g.bias) x += g.bias[col];
              if (g.accumulate) x += *cp;
              if (g.act == 1) x = sk_sigmoid(x);
              *cp = x;
            }
          }
        }
      }
    }
    clear();
    if (++ci < nseg) segment(ci, cv, ck, cke);
  }
}

// fp32 (R, C) -> bf16 copy with leading dimension ldd >= C, columns C..ldd-1 zero (row-major), RNE
__global__ __launch_bounds__(256) void cast_kernel(const float* __restrict__ src, int R, int C, int lds_, __bf16* __restrict__ dst,
                                                   int ldd, int Rpad) {
  const int64_t n8 = (int64_t)Rpad * (ldd / 8);  // rows R .. Rpad-1 of the copy are zero
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
    const int r = (int)(i / (ldd / 8)), c = (int)(i % (ldd / 8)) * 8;
    const float* s = src + (int64_t)r * lds_ + c;
    bf16x8 v;
    if (r >= R) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = (__bf16)0.f;
    } else if (c + 8 <= C && ((lds_ & 3) == 0) && (((uintptr_t)src & 15) == 0)) {
      const float4 a = *reinterpret_cast<const float4*>(s), b = *reinterpret_cast<const float4*>(s + 4);
      v[0] = (__bf16)a.x; v[1] = (__bf16)a.y; v[2] = (__bf16)a.z; v[3] = (__bf16)a.w;
      v[4] = (__bf16)b.x; v[5] = (__bf16)b.y; v[6] = (__bf16)b.z; v[7] = (__bf16)b.w;
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = (__bf16)(c + e < C ? s[e] : 0.f);
    }
    *reinterpret_cast<bf16x8*>(dst + (int64_t)r * ldd + c) = v;
  }
}

}  // namespace bf2

// C = act(sum_ks slabs[z][ks] + bias (+ C)), slices added in fixed order (deterministic)
__global__ __launch_bounds__(256) void splitk_reduce_kernel(GemmArgs g) {
  const int z = blockIdx.z;
  const int64_t mn = (int64_t)g.M * g.N;
  const float* sl = g.slabs + (int64_t)z * g.splitk * mn;
  float* C = g.C + z * g.sC;
  const float* bias = g.bias ? g.bias + z * g.sbias : nullptr;
  for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; i < mn; i += (int64_t)gridDim.x * 1024) {
    // N % 4 == 0 is required by the host for the vector path; otherwise fall back to scalars
    if ((g.N & 3) == 0) {
      float4 a = *reinterpret_cast<const float4*>(sl + i);
      for (int k = 1; k < g.splitk; ++k) {
        const float4 b = *reinterpret_cast<const float4*>(sl + k * mn + i);
        a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
      }
      const int row = (int)(i / g.N), col = (int)(i - (int64_t)row * g.N);
      float v[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float* cp = C + (int64_t)row * g.ldc + col + e;
        float x = v[e] + (bias ? bias[col + e] : 0.f);
        if (g.accumulate) x += *cp;
        if (g.act == 1) x = sk_sigmoid(x);
        *cp = x;
      }
    } else {
      for (int e = 0; e < 4 && i + e < mn; ++e) {
        float a = sl[i + e];
        for (int k = 1; k < g.splitk; ++k) a += sl[k * mn + i + e];
        const int row = (int)((i + e) / g.N), col = (int)((i + e) - (int64_t)row * g.N);
        float* cp = C + (int64_t)row * g.ldc + col;
        float x = a + (bias ? bias[col] : 0.f);
        if (g.accumulate) x += *cp;
        if (g.act == 1) x = sk_sigmoid(x);
        *cp = x;
      }
    }
  }
}

}  // namespace

extern "C" int sk_gemm_workspace_init(void* ws, sk_stream_t stream) {
  SK_CHECK_ARG(ws, "sk_gemm_workspace_init: null pointer");
  SK_CHECK_HIP(hipMemsetAsync(ws, 0, COUNTER_BYTES, (hipStream_t)stream));
  return SK_OK;
}

namespace {
int gemm_cus() {
  static int n = 0;
  if (!n) {
    int dev = 0;
    hipDeviceProp_t p;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) n = p.multiProcessorCount;
  }
  return n;
}
// workgroups of the stream-K kernel: one per CU, a multiple of 8 (the tile order counts on blockIdx & 7 = XCD)
int streamk_wgs() { return gemm_cus() & ~7; }
}  // namespace

extern "C" size_t sk_gemm_streamk_workspace_bytes(void) {
  return COUNTER_BYTES + (size_t)2 * streamk_wgs() * 256 * 256 * sizeof(float);
}

extern "C" size_t sk_gemm_workspace_bytes(int M, int N, int batch, int splitk) {
  if (splitk <= 1) return 0;
  return COUNTER_BYTES + sk_align((size_t)M * N * batch * splitk * sizeof(float), 256);
}

extern "C" int sk_gemm_f32(const float* A, const float* B, float* C, const float* bias, int M, int N, int K, int lda,
                           int ldb, int ldc, int transA, int transB, int accumulate, int act, int batch, int64_t sA,
                           int64_t sB, int64_t sC, int64_t sbias, sk_stream_t stream) {
  return sk_gemm_f32_splitk(A, B, C, bias, M, N, K, lda, ldb, ldc, transA, transB, accumulate, act, batch, sA, sB, sC,
                            sbias, 1, nullptr, 0, stream);
}

namespace {

// The LDS-DMA kernel takes a launch when every operand row is 16-byte aligned, every K slice is a multiple of its
// K step, a k-major operand's tile dimension is a multiple of 4 (whole float4 pieces) and it is not the T/T form.
// SEPKERN_GEMM_DMA=0 (diagnostics) keeps everything on the register-staged kernel.
bool dma_ok(const GemmArgs& g, int transA, int transB, bool choose = false) {
  static const bool enabled = [] {
    const char* e = getenv("SEPKERN_GEMM_DMA");
    return !(e && e[0] == '0');
  }();
  if (!enabled || !g.vecA || !g.vecB || (transA && transB)) return false;
  if (g.K % BK != 0 || g.kchunk % BK != 0) return false;
  if (transA && (g.M % 4 != 0 || g.M < 4)) return false;
  if (!transB && (g.N % 4 != 0 || g.N < 4)) return false;
  // measured (sustained, one MI355X): the large N/T products run 3-4 % faster register-staged (125.8 vs 121.4 TFLOP/s at
  // 12800 x 7168 x 1792), everything else faster by DMA (N/N +5 %, T/N +2..5 %, N = 514 N/T +25 %)
  if (choose && !transA && transB && g.M >= 1024 && g.N >= 1024) return false;
  return true;
}

int gemm_launch(bool bf16, const float* A, const float* B, float* C, const float* bias, int M, int N, int K, int lda,
                int ldb, int ldc, int transA, int transB, int accumulate, int act, int batch, int64_t sA, int64_t sB,
                int64_t sC, int64_t sbias, int splitk, void* ws, int variant, sk_stream_t stream) {
  SK_CHECK_ARG(A && B && C, "sk_gemm: null pointer");
  SK_CHECK_ARG(variant >= 0 && variant <= 9 && variant != 5 && variant != 7, "sk_gemm: unknown variant %d (5 and 7 are retired)", variant);
  SK_CHECK_ARG(splitk >= 1 && splitk <= 64 && (splitk == 1 || ws), "sk_gemm: bad splitk %d / missing workspace", splitk);
  SK_CHECK_ARG(M > 0 && N > 0 && K > 0 && batch > 0 && batch <= 65535, "sk_gemm: bad sizes M=%d N=%d K=%d batch=%d", M, N, K, batch);
  SK_CHECK_ARG(lda >= (transA ? M : K) && ldb >= (transB ? K : N) && ldc >= N, "sk_gemm: leading dimension too small");
  SK_CHECK_ARG(act == 0 || act == 1, "sk_gemm: unknown activation %d", act);
  const int bk = bf16 ? bf::BK : BK;
  GemmArgs g;
  g.A = A; g.B = B; g.C = C; g.bias = bias;
  g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
  g.accumulate = accumulate; g.act = act;
  g.vecA = ((uintptr_t)A % 16 == 0) && (lda % 4 == 0) && (sA % 4 == 0);
  g.vecB = ((uintptr_t)B % 16 == 0) && (ldb % 4 == 0) && (sB % 4 == 0);
  g.tilesN = (int)sk_cdiv(N, BN);
  g.sA = sA; g.sB = sB; g.sC = sC; g.sbias = sbias;
  g.sk_tiles = 0; g.sk_full = 0; g.flip_q = 0;
  g.kchunk = (int)(sk_cdiv(sk_cdiv(K, splitk), bk) * bk);
  splitk = (int)sk_cdiv(K, g.kchunk);  // slices that actually hold work
  g.splitk = splitk;
  // ---- which kernel.  fp32 products run by default (variant 0) on the bf16 matrix pipe by the three-way split of both operands
  // with six piece products (gemm_f32_kernel_planes / gemm_f32_kernel_split3) wherever the LDS-DMA conditions hold: 160-212 TFLOP/s
  // fp32-equivalent on the training step's large products against 124-135 of the fp32-MFMA kernels (stand-alone, one MI355X).
  // Variant 8 = the choice among the fp32-MFMA kernels (the reference's literal arithmetic; SEPKERN_GEMM_SPLIT=0 makes variant 0
  // that); operands with unaligned rows or K % 16 != 0 take the fp32-MFMA kernels always.
  static const bool split_on = [] { const char* e = getenv("SEPKERN_GEMM_SPLIT"); return !(e && e[0] == '0'); }();
  if (variant == 0 && !split_on) variant = 8;
  const bool split = !bf16 && (variant == 0 || variant == 2 || variant == 9) && dma_ok(g, transA, transB);
  // split ONCE per element while staging (gemm_f32_kernel_planes): unsplit, unbatched products.  With enough 256 x 128 tiles to
  // fill the chip it is what variant 0 takes (SEPKERN_GEMM_PLANES=0: never): 186-212 TFLOP/s on the projection / data-gradient /
  // unsplit weight-gradient shapes against 160-168 of the 128 x 128 kernel; 158 vs 100-124 on the N = 514 Linear product.  It
  // needs 72 KB of LDS and 200 VGPRs: callers that run a product BESIDE a persistent recurrence pass variant 2 (sepkern/engine.py).
  static const bool planes_on = [] { const char* e = getenv("SEPKERN_GEMM_PLANES"); return !(e && e[0] == '0'); }();
  const bool planes = split && splitk == 1 && batch == 1 && M >= 256 && N >= 128 &&
                      (variant == 9 || (variant == 0 && planes_on && sk_cdiv(M, 256) * sk_cdiv(N, 128) >= 192));
  // Sign phases of the split products (SignPhase): stretches of q K steps, signs + - - +, a whole number of periods of about 64
  // steps over the WHOLE K (the slices of a split-K product continue one pattern: K = 1792: q = 14, 7168: 16, 12800: 17 -- a sign
  // change per ~32 steps, the r05 N/N kernel's rate).  What is left of the offset is at most one stretch's worth inside the
  // result's last binade: measured on all-positive operands -5e-10 ... +3e-9 of the result where the plain form has -3.6e-8
  // (K = 1792) ... -2.7e-7 (K = 12800) and the fp32-MFMA kernels -4e-10 ... -1e-9 (profiles/r06_signed_error.txt).  Products
  // shorter than 48 steps (K < 768: the layer-0 projection's K = 272; offset -5e-9) keep the plain form -- three uneven
  // stretches would over-correct it.
  if (split && SK_SPLIT_FLIP) {
    const int nks = K / BK;
    if (nks >= 48) {
      const int periods = (nks + 32) / 64;
      g.flip_q = (nks + 4 * periods - 1) / (4 * periods);
    }
  }
  const bool mfma_choose = variant == 8 || (variant == 0 && !split);  // the r04 policy among the fp32-MFMA kernels
  // 256 x 128 block tiles, 8 waves (fp32 MFMA): variant 4, or chosen for the large unsplit N/T and N/N products -- measured
  // +2 % / +5 % on them stand-alone.  SEPKERN_GEMM_WIDE=0 (diagnostics): never chosen.
  static const bool wide_ok = [] { const char* e = getenv("SEPKERN_GEMM_WIDE"); return !(e && e[0] == '0'); }();
  const bool wide = !bf16 && !split && M >= 256 && dma_ok(g, transA, transB) &&
                    (variant == 4 || variant == 6 || (mfma_choose && wide_ok && !transA && M >= 4096 && N >= 1024 && splitk == 1));
  // 256 x 256 tiles, persistent, with a stream-K cut of the last partial round (fp32 MFMA): variant 6, or chosen under 8 for
  // large unsplit products when the caller passes the workspace of sk_gemm_streamk_workspace_bytes().
  // SEPKERN_GEMM_STREAMK=0 (diagnostics): never chosen.
  static const bool streamk_ok = [] { const char* e = getenv("SEPKERN_GEMM_STREAMK"); return !(e && e[0] == '0'); }();
  const bool sk_shape = ws && batch == 1 && M >= 256 && N >= 256 && splitk == 1 && dma_ok(g, transA, transB);
  bool streamk = !bf16 && !split && sk_shape && (variant == 6 || (mfma_choose && streamk_ok && !transA && M >= 4096 && N >= 1024));
  if (streamk) {
    const int P = streamk_wgs();
    const int64_t nt = sk_cdiv(M, 256) * sk_cdiv(N, 256), nk = K / BK;
    g.sk_tiles = (int)nt;
    g.sk_full = (int)(nt / P);
    const int64_t R = (nt - (int64_t)g.sk_full * P) * nk;
    // a remainder too short to give every workgroup a K step goes to whole tiles (the plain kernel's last round)
    if (P < 8 || nt >= (1 << 24) || nk < 8 || (R > 0 && R < P) || nt - (int64_t)g.sk_full * P > 16384) streamk = false;
  }
  if (streamk) g.tilesN = (int)sk_cdiv(N, 256);
  if (planes) g.tilesN = (int)sk_cdiv(N, 128);
  const int64_t tiles = sk_cdiv(M, (wide || streamk || planes) ? 256 : BM) * g.tilesN;
  SK_CHECK_ARG(tiles < (1ll << 31), "sk_gemm: too many tiles");
  // workspace = [ticket counters | slabs]; the fp32 kernels reduce in-kernel when the counters cover every (batch, tile)
  g.slabs = ws ? (float*)((char*)ws + COUNTER_BYTES) : nullptr;
  const bool inkernel = !bf16 && splitk > 1 && tiles * batch * sizeof(unsigned) <= COUNTER_BYTES;
  g.counters = inkernel ? (unsigned*)ws : nullptr;
  dim3 grid((unsigned)tiles, (unsigned)splitk, (unsigned)batch);
  hipStream_t st = (hipStream_t)stream;
  t_last_kernel = bf16 ? 9 : planes ? 10 : streamk ? 6 : wide ? 4 : split ? 2 : (variant != 1 && dma_ok(g, transA, transB, mfma_choose)) ? 3 : 1;
  if (bf16) {
    if (!transA && !transB)
      hipLaunchKernelGGL((bf::gemm_bf16_kernel<false, false>), grid, dim3(256), 0, st, g);
    else if (!transA && transB)
      hipLaunchKernelGGL((bf::gemm_bf16_kernel<false, true>), grid, dim3(256), 0, st, g);
    else if (transA && !transB)
      hipLaunchKernelGGL((bf::gemm_bf16_kernel<true, false>), grid, dim3(256), 0, st, g);
    else
      hipLaunchKernelGGL((bf::gemm_bf16_kernel<true, true>), grid, dim3(256), 0, st, g);
  } else if (streamk) {
    g.counters = (unsigned*)ws;
    const dim3 pgrid((unsigned)streamk_wgs());
    if (!transA && !transB)
      hipLaunchKernelGGL((gemm_f32_kernel_streamk<false, false>), pgrid, dim3(512), 0, st, g);
    else if (!transA && transB)
      hipLaunchKernelGGL((gemm_f32_kernel_streamk<false, true>), pgrid, dim3(512), 0, st, g);
    else
      hipLaunchKernelGGL((gemm_f32_kernel_streamk<true, false>), pgrid, dim3(512), 0, st, g);
  } else if (wide) {
    if (!transA && !transB)
      hipLaunchKernelGGL((gemm_f32_kernel_dma256<false, false>), grid, dim3(512), 0, st, g);
    else if (!transA && transB)
      hipLaunchKernelGGL((gemm_f32_kernel_dma256<false, true>), grid, dim3(512), 0, st, g);
    else
      hipLaunchKernelGGL((gemm_f32_kernel_dma256<true, false>), grid, dim3(512), 0, st, g);
  } else if (planes) {
    if (!transA && transB)
      hipLaunchKernelGGL((gemm_f32_kernel_planes<false, false>), grid, dim3(512), 0, st, g);
    else if (!transA && !transB)
      hipLaunchKernelGGL((gemm_f32_kernel_planes<false, true>), grid, dim3(512), 0, st, g);  // (data gradients)
    else
      hipLaunchKernelGGL((gemm_f32_kernel_planes<true, true>), grid, dim3(512), 0, st, g);
  } else if (split) {
    if (!transA && !transB)
      hipLaunchKernelGGL((gemm_f32_kernel_split3<false, false>), grid, dim3(256), 0, st, g);
    else if (!transA && transB)
      hipLaunchKernelGGL((gemm_f32_kernel_split3<false, true>), grid, dim3(256), 0, st, g);
    else
      hipLaunchKernelGGL((gemm_f32_kernel_split3<true, false>), grid, dim3(256), 0, st, g);
  } else if (variant != 1 && dma_ok(g, transA, transB, mfma_choose)) {
    if (!transA && !transB)
      hipLaunchKernelGGL((gemm_f32_kernel_dma<false, false>), grid, dim3(256), 0, st, g);
    else if (!transA && transB)
      hipLaunchKernelGGL((gemm_f32_kernel_dma<false, true>), grid, dim3(256), 0, st, g);
    else
      hipLaunchKernelGGL((gemm_f32_kernel_dma<true, false>), grid, dim3(256), 0, st, g);
  } else if (g.vecA && g.vecB) {
    if (!transA && !transB)
      hipLaunchKernelGGL((gemm_f32_kernel<false, false>), grid, dim3(256), 0, st, g);
    else if (!transA && transB)
      hipLaunchKernelGGL((gemm_f32_kernel<false, true>), grid, dim3(256), 0, st, g);
    else if (transA && !transB)
      hipLaunchKernelGGL((gemm_f32_kernel<true, false>), grid, dim3(256), 0, st, g);
    else
      hipLaunchKernelGGL((gemm_f32_kernel<true, true>), grid, dim3(256), 0, st, g);
  } else {  // an operand with unaligned rows (F = 257): dword-load variant
    if (!transA && !transB)
      hipLaunchKernelGGL((gemm_f32_kernel<false, false, false>), grid, dim3(256), 0, st, g);
    else if (!transA && transB)
      hipLaunchKernelGGL((gemm_f32_kernel<false, true, false>), grid, dim3(256), 0, st, g);
    else if (transA && !transB)
      hipLaunchKernelGGL((gemm_f32_kernel<true, false, false>), grid, dim3(256), 0, st, g);
    else
      hipLaunchKernelGGL((gemm_f32_kernel<true, true, false>), grid, dim3(256), 0, st, g);
  }
  SK_CHECK_LAUNCH("sk_gemm");
  if (splitk > 1 && !inkernel) {
    const int64_t quads = sk_cdiv((int64_t)M * N, 4);
    const unsigned nb = (unsigned)(sk_cdiv(quads, 256) > 2048 ? 2048 : sk_cdiv(quads, 256));
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(nb, 1, (unsigned)batch), dim3(256), 0, st, g);
    SK_CHECK_LAUNCH("splitk_reduce_kernel");
  }
  return SK_OK;
}

}  // namespace

extern "C" int sk_gemm_f32_splitk(const float* A, const float* B, float* C, const float* bias, int M, int N, int K, int lda,
                           int ldb, int ldc, int transA, int transB, int accumulate, int act, int batch, int64_t sA,
                           int64_t sB, int64_t sC, int64_t sbias, int splitk, void* ws, int variant, sk_stream_t stream) {
  return gemm_launch(false, A, B, C, bias, M, N, K, lda, ldb, ldc, transA, transB, accumulate, act, batch, sA, sB, sC,
                     sbias, splitk, ws, variant, stream);
}

extern "C" int sk_gemm_bf16_splitk(const float* A, const float* B, float* C, const float* bias, int M, int N, int K, int lda,
                           int ldb, int ldc, int transA, int transB, int accumulate, int act, int batch, int64_t sA,
                           int64_t sB, int64_t sC, int64_t sbias, int splitk, void* ws, sk_stream_t stream) {
  return gemm_launch(true, A, B, C, bias, M, N, K, lda, ldb, ldc, transA, transB, accumulate, act, batch, sA, sB, sC,
                     sbias, splitk, ws, 0, stream);
}

// ---------------------------------------------------------------- bf16 operands in memory
extern "C" int sk_gemm_bf16_nt(const void* A, const void* B, float* C, const float* bias, int M, int N, int K, int lda,
                               int ldb, int ldc, int accumulate, int act, int batch, int64_t sA, int64_t sB, int64_t sC,
                               int64_t sbias, int splitk, void* ws, sk_stream_t stream) {
  return sk_gemm_bf16_mm(A, B, C, bias, M, N, K, lda, ldb, ldc, 0, 0, accumulate, act, batch, sA, sB, sC, sbias, splitk, ws,
                         stream);
}

extern "C" int sk_gemm_bf16_mm(const void* A, const void* B, float* C, const float* bias, int M, int N, int K, int lda,
                               int ldb, int ldc, int a_kmajor, int b_kmajor, int accumulate, int act, int batch, int64_t sA,
                               int64_t sB, int64_t sC, int64_t sbias, int splitk, void* ws, sk_stream_t stream) {
  SK_CHECK_ARG(A && B && C, "sk_gemm_bf16_mm: null pointer");
  SK_CHECK_ARG(M > 0 && N > 0 && K > 0 && batch > 0 && batch <= 65535, "sk_gemm_bf16_mm: bad sizes M=%d N=%d K=%d batch=%d", M, N, K, batch);
  SK_CHECK_ARG(K % bf2::BK == 0, "sk_gemm_bf16_mm: K = %d must be a multiple of %d (pad the bf16 copies with zeros)", K, bf2::BK);
  SK_CHECK_ARG(ldc >= N && lda % 8 == 0 && ldb % 8 == 0 && sA % 8 == 0 && sB % 8 == 0,
               "sk_gemm_bf16_mm: leading dimensions / batch strides must be multiples of 8 elements");
  SK_CHECK_ARG(lda >= (a_kmajor ? (M + 7) / 8 * 8 : K) && ldb >= (b_kmajor ? (N + 7) / 8 * 8 : K),
               "sk_gemm_bf16_mm: leading dimension too small (K-major operands: the padded width of a k row)");
  SK_CHECK_ARG(((uintptr_t)A % 16) == 0 && ((uintptr_t)B % 16) == 0, "sk_gemm_bf16_mm: operands must be 16-byte aligned");
  SK_CHECK_ARG(splitk >= 1 && splitk <= 64 && (splitk == 1 || ws), "sk_gemm_bf16_mm: bad splitk %d / missing workspace", splitk);
  SK_CHECK_ARG(act == 0 || act == 1, "sk_gemm_bf16_mm: unknown activation %d", act);
  bf2::Args g;
  g.A = (const __bf16*)A; g.B = (const __bf16*)B; g.C = C; g.bias = bias;
  g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
  g.accumulate = accumulate; g.act = act;
  g.sA = sA; g.sB = sB; g.sC = sC; g.sbias = sbias;
  g.kchunk = (int)(sk_cdiv(sk_cdiv(K, splitk), bf2::BK) * bf2::BK);
  splitk = (int)sk_cdiv(K, g.kchunk);
  g.splitk = splitk;
  g.slabs = ws ? (float*)((char*)ws + COUNTER_BYTES) : nullptr;  // (the head of a workspace belongs to the fp32 kernels' counters)
  g.counters = (unsigned*)ws; g.sk_tiles = 0; g.sk_full = 0;
  // splitk = 1 WITH a workspace (>= sk_gemm_streamk_workspace_bytes, zero-filled before its first use): the persistent
  // stream-K kernel where it applies (unbatched, 256-wide column tiles, at least 8 K steps) -- r03
  bool streamk = splitk == 1 && ws && batch == 1 && M >= 256 && (N % 256 == 0 || N > 1024);
  if (streamk) {
    const int P = streamk_wgs();
    const int64_t nt = sk_cdiv(M, bf2::BM) * sk_cdiv(N, 256), nk = K / bf2::BK;
    g.sk_tiles = (int)nt;
    g.sk_full = (int)(nt / P);
    const int64_t rem = nt - (int64_t)g.sk_full * P, R = rem * nk;
    if (P < 8 || nt >= (1 << 24) || nk < 8 || (R > 0 && R < P) || rem > 16384) streamk = false;
  }
  // 256-wide column tiles unless N is small enough that they would leave most of the chip idle
  const int64_t tiles256 = sk_cdiv(M, bf2::BM) * sk_cdiv(N, 256) * splitk * batch;
  const bool wide = streamk || ((N % 256 == 0 || N > 1024) && tiles256 >= 2 * 256);
  const int bn = wide ? 256 : 128;
  g.tilesN = (int)sk_cdiv(N, bn);
  const int64_t tiles = sk_cdiv(M, bf2::BM) * g.tilesN;
  SK_CHECK_ARG(tiles < (1ll << 31), "sk_gemm_bf16_mm: too many tiles");
  dim3 grid((unsigned)tiles, (unsigned)splitk, (unsigned)batch);
  hipStream_t st = (hipStream_t)stream;
  t_last_kernel = streamk ? 13 : wide ? 12 : 11;  // (sk_gemm_last_kernel: the bf16-operand kernels)
#define SK_BF2_LAUNCH(BNV, AK, BKV) hipLaunchKernelGGL((bf2::gemm_bf16_nt_kernel<BNV, AK, BKV>), grid, dim3(bf2::NT), 0, st, g)
#define SK_BF2_STREAMK(AK, BKV) \
  hipLaunchKernelGGL((bf2::gemm_bf16_streamk_kernel<AK, BKV>), dim3((unsigned)streamk_wgs()), dim3(bf2::NT), 0, st, g)
  if (streamk) {
    if (!a_kmajor && !b_kmajor) SK_BF2_STREAMK(false, false);
    else if (!a_kmajor) SK_BF2_STREAMK(false, true);
    else if (!b_kmajor) SK_BF2_STREAMK(true, false);
    else SK_BF2_STREAMK(true, true);
  } else if (wide) {
    if (!a_kmajor && !b_kmajor) SK_BF2_LAUNCH(256, false, false);
    else if (!a_kmajor) SK_BF2_LAUNCH(256, false, true);
    else if (!b_kmajor) SK_BF2_LAUNCH(256, true, false);
    else SK_BF2_LAUNCH(256, true, true);
  } else {
    if (!a_kmajor && !b_kmajor) SK_BF2_LAUNCH(128, false, false);
    else if (!a_kmajor) SK_BF2_LAUNCH(128, false, true);
    else if (!b_kmajor) SK_BF2_LAUNCH(128, true, false);
    else SK_BF2_LAUNCH(128, true, true);
  }
#undef SK_BF2_LAUNCH
#undef SK_BF2_STREAMK
  SK_CHECK_LAUNCH("sk_gemm_bf16_nt");
  if (splitk > 1) {
    GemmArgs r;
    r.A = nullptr; r.B = nullptr; r.C = C; r.bias = bias;
    r.M = M; r.N = N; r.K = K; r.lda = 0; r.ldb = 0; r.ldc = ldc;
    r.accumulate = accumulate; r.act = act; r.vecA = r.vecB = 0; r.tilesN = g.tilesN;
    r.splitk = splitk; r.kchunk = g.kchunk; r.slabs = g.slabs; r.counters = nullptr;
    r.sA = 0; r.sB = 0; r.sC = sC; r.sbias = sbias;
    const int64_t quads = sk_cdiv((int64_t)M * N, 4);
    const unsigned nb = (unsigned)(sk_cdiv(quads, 256) > 2048 ? 2048 : sk_cdiv(quads, 256));
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(nb, 1, (unsigned)batch), dim3(256), 0, st, r);
    SK_CHECK_LAUNCH("splitk_reduce_kernel");
  }
  return SK_OK;
}

// ---------------------------------------------------------------- operands that arrive split
extern "C" int sk_split_rows(const float* src, int R, int C, int ld_src, void* dst, int ld_dst, int R_pad, int64_t plane,
                             sk_stream_t stream) {
  SK_CHECK_ARG(src && dst && R > 0 && C > 0 && ld_src >= C && ld_dst >= C && ld_dst % 8 == 0 && ((uintptr_t)dst % 16) == 0 &&
                   R_pad >= R && plane >= (int64_t)R_pad * ld_dst && plane % 8 == 0,
               "sk_split_rows: bad arguments");
  const int64_t n4 = (int64_t)R_pad * (ld_dst / 4);
  const unsigned nb = (unsigned)(sk_cdiv(n4, 256) > 8192 ? 8192 : sk_cdiv(n4, 256));
  hipLaunchKernelGGL(split_rows_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, src, R, C, ld_src, (__bf16*)dst, ld_dst, R_pad, plane);
  SK_CHECK_LAUNCH("sk_split_rows");
  return SK_OK;
}

extern "C" int sk_gemm_pl3_tn(const void* Apl, const void* Bpl, float* C, int M, int N, int K, int lda, int ldb, int ldc,
                              int64_t planeA, int64_t planeB, int accumulate, int batch, int64_t sA, int64_t sB, int64_t sC,
                              int splitk, void* ws, sk_stream_t stream) {
  SK_CHECK_ARG(Apl && Bpl && C, "sk_gemm_pl3_tn: null pointer");
  SK_CHECK_ARG(M > 0 && N > 0 && K > 0 && batch > 0 && batch <= 65535, "sk_gemm_pl3_tn: bad sizes M=%d N=%d K=%d batch=%d", M, N, K, batch);
  SK_CHECK_ARG(K % BK == 0, "sk_gemm_pl3_tn: K = %d must be a multiple of %d (the planes' zero tail rows)", K, BK);
  SK_CHECK_ARG(ldc >= N && lda % 8 == 0 && ldb % 8 == 0 && (batch == 1 || (sA % 8 == 0 && sB % 8 == 0)) && planeA % 8 == 0 && planeB % 8 == 0,
               "sk_gemm_pl3_tn: leading dimensions, plane and batch strides must be multiples of 8 elements");
  SK_CHECK_ARG(lda >= (batch - 1) * sA + ((M + 7) & ~7) && ldb >= (batch - 1) * sB + ((N + 7) & ~7),
               "sk_gemm_pl3_tn: leading dimension smaller than the operand's padded width");
  SK_CHECK_ARG(((uintptr_t)Apl % 16) == 0 && ((uintptr_t)Bpl % 16) == 0, "sk_gemm_pl3_tn: planes must be 16-byte aligned");
  SK_CHECK_ARG(splitk >= 1 && splitk <= 64 && (splitk == 1 || ws), "sk_gemm_pl3_tn: bad splitk %d / missing workspace", splitk);
  GemmArgs g;
  g.A = nullptr; g.B = nullptr; g.C = C; g.bias = nullptr;
  g.Apl = (const __bf16*)Apl; g.Bpl = (const __bf16*)Bpl; g.planeA = planeA; g.planeB = planeB;
  g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
  g.accumulate = accumulate; g.act = 0; g.vecA = g.vecB = 1;
  g.tilesN = (int)sk_cdiv(N, BN);
  g.sA = sA; g.sB = sB; g.sC = sC; g.sbias = 0;
  g.sk_tiles = 0; g.sk_full = 0; g.flip_q = 0;
  g.kchunk = (int)(sk_cdiv(sk_cdiv(K, splitk), BK) * BK);
  splitk = (int)sk_cdiv(K, g.kchunk);
  g.splitk = splitk;
  if (SK_SPLIT_FLIP) {  // the sign-phase rule of gemm_launch: one pattern over the whole K
    const int nks = K / BK;
    if (nks >= 48) {
      const int periods = (nks + 32) / 64;
      g.flip_q = (nks + 4 * periods - 1) / (4 * periods);
    }
  }
  const int64_t tiles = sk_cdiv(M, BM) * g.tilesN;
  SK_CHECK_ARG(tiles < (1ll << 31), "sk_gemm_pl3_tn: too many tiles");
  g.slabs = ws ? (float*)((char*)ws + COUNTER_BYTES) : nullptr;
  const bool inkernel = splitk > 1 && tiles * batch * sizeof(unsigned) <= COUNTER_BYTES;
  g.counters = inkernel ? (unsigned*)ws : nullptr;
  dim3 grid((unsigned)tiles, (unsigned)splitk, (unsigned)batch);
  hipStream_t st = (hipStream_t)stream;
  t_last_kernel = 14;
  hipLaunchKernelGGL(gemm_f32_kernel_pl3, grid, dim3(256), 0, st, g);
  SK_CHECK_LAUNCH("sk_gemm_pl3_tn");
  if (splitk > 1 && !inkernel) {
    const int64_t quads = sk_cdiv((int64_t)M * N, 4);
    const unsigned nb = (unsigned)(sk_cdiv(quads, 256) > 2048 ? 2048 : sk_cdiv(quads, 256));
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(nb, 1, (unsigned)batch), dim3(256), 0, st, g);
    SK_CHECK_LAUNCH("splitk_reduce_kernel");
  }
  return SK_OK;
}

extern "C" int sk_cast_bf16_rows(const float* src, int R, int C, int ld_src, void* dst, int ld_dst, int R_pad,
                                 sk_stream_t stream) {
  SK_CHECK_ARG(src && dst && R > 0 && C > 0 && ld_src >= C && ld_dst >= C && ld_dst % 8 == 0 && ((uintptr_t)dst % 16) == 0 &&
                   R_pad >= R,
               "sk_cast_bf16: bad arguments");
  const int64_t n8 = (int64_t)R_pad * (ld_dst / 8);
  const unsigned nb = (unsigned)(sk_cdiv(n8, 256) > 4096 ? 4096 : sk_cdiv(n8, 256));
  hipLaunchKernelGGL(bf2::cast_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, src, R, C, ld_src, (__bf16*)dst, ld_dst, R_pad);
  SK_CHECK_LAUNCH("sk_cast_bf16");
  return SK_OK;
}

extern "C" int sk_gemm_last_kernel(void) { return t_last_kernel; }

// The numerics- or timing-changing macros this translation unit was built with (sk_build_flags, include/sepkern.h)
unsigned sk_gemm_build_flags() {
  unsigned f = 0;
#if defined(SK_SPLIT_FREE) || defined(SK_SPLIT_FREE_TN) || defined(SK_ABL_HALFDMA) || defined(SK_ABL_NOBAR) || defined(SK_ABL_NODMA)
  f |= SK_BUILD_TIMING_ONLY;
#endif
#ifdef SK_SPLIT_NINE
  f |= SK_BUILD_ARITH;
#endif
  if (SK_SPLIT_FLIP != 1) f |= SK_BUILD_ARITH;
  if (SK_SPLIT_NST != 2 || SK_SPLIT_OCC != 2 || SK_PLANES_SCHED != 4) f |= SK_BUILD_TUNING;
  return f;
}
