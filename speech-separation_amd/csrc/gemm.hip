// gemm.hip -- fp32-in / fp32-accumulate MFMA GEMM for gfx950 (v_mfma_f32_32x32x2_f32).
//
// C[M,N] = act(opA(A) * opB(B) + bias + (accumulate ? C : 0)).  Used for everything on the uPIT
// path that is a plain matrix product: the LSTM input projections for all T*B rows at once,
// nn.Linear, and their dgrad / wgrad passes (reference archs/uPIT.py:132,141 via torch).
// fp32 MFMA is bit-for-bit a k-ordered fmaf chain, so results match an fp32 reference to
// rounding-order differences only.
//
// Tiling: 128x128 block tile, 4 waves as 2x2, each wave 64x64 = 2x2 MFMA tiles of 32x32
// (64 accumulator VGPRs); K step 16, LDS double-buffered, operands kept k-major in LDS so the
// per-lane A/B fragment reads are conflict-free ds_read_b32 of 32 consecutive dwords.
#include "sk_common.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 16, LD = 132;
typedef float f32x16 __attribute__((ext_vector_type(16)));

struct GemmArgs {
  const float* A;
  const float* B;
  float* C;
  const float* bias;
  int M, N, K, lda, ldb, ldc;
  int accumulate, act, vecA, vecB;
  int tilesN;
  int64_t sA, sB, sC, sbias;
};

// Fetch this thread's two float4 pieces of an operand tile.  Two branch-free forms, selected by a
// block-uniform condition (per-element "load or zero" branches make hipcc wait vmcnt(0) per load):
//  fast : tile fully inside the matrix and 16-byte aligned -> unconditional float4 loads
//  slow : every element loaded from a CLAMPED in-range address and zeroed by a select
//  KMAJOR == false: operand stored [tile dim (M or N)][K]  (transposed into k-major LDS by stash)
//  KMAJOR == true : operand stored [K][tile dim]            (already k-major)
template <bool KMAJOR>
__device__ __forceinline__ void fetch(const float* __restrict__ P, int ld, int dim0, int dimLimit, int k0, int K,
                                      bool fast, int tid, float4 (&r)[2]) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int idx = tid + 256 * i;
    int row, col, rowLimit, colLimit;  // element (row, col..col+3) of the stored matrix
    if (KMAJOR) {
      row = k0 + (idx >> 5);
      col = dim0 + (idx & 31) * 4;
      rowLimit = K;
      colLimit = dimLimit;
    } else {
      row = dim0 + (idx >> 2);
      col = k0 + (idx & 3) * 4;
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
__device__ __forceinline__ void stash(float (*S)[LD], int tid, const float4 (&r)[2]) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int idx = tid + 256 * i;
    if (KMAJOR) {
      const int kr = idx >> 5, c4 = (idx & 31) * 4;
      *reinterpret_cast<float4*>(&S[kr][c4]) = r[i];
    } else {
      const int row = idx >> 2, k4 = (idx & 3) * 4;
      S[k4 + 0][row] = r[i].x;
      S[k4 + 1][row] = r[i].y;
      S[k4 + 2][row] = r[i].z;
      S[k4 + 3][row] = r[i].w;
    }
  }
}

template <bool TA, bool TB>
__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmArgs g) {
  __shared__ __attribute__((aligned(16))) float As[2][BK][LD];
  __shared__ __attribute__((aligned(16))) float Bs[2][BK][LD];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  // XCD-aware tile order: blocks b and b+8 share an XCD (and its L2), so give each XCD a
  // contiguous run of tiles (n fastest): neighbours in a run share their A panel, runs share B.
  int tile = blockIdx.x;
  {
    const int nt = gridDim.x, q = nt >> 3, rem = nt & 7, x = tile & 7, j = tile >> 3;
    tile = (x < rem ? x * (q + 1) : rem * (q + 1) + (x - rem) * q) + j;
  }
  const int m0 = (tile / g.tilesN) * BM, n0 = (tile % g.tilesN) * BN;
  const bool fullA = g.vecA && (m0 + BM <= g.M), fullB = g.vecB && (n0 + BN <= g.N);
  const int z = blockIdx.z;
  const float* A = g.A + z * g.sA;
  const float* B = g.B + z * g.sB;
  float* C = g.C + z * g.sC;
  const float* bias = g.bias ? g.bias + z * g.sbias : nullptr;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nk = (g.K + BK - 1) / BK;
  float4 ra[2], rb[2];
  // A is k-major in memory when TA (stored K x M); B is k-major when !TB (stored K x N)
  fetch<TA>(A, g.lda, m0, g.M, 0, g.K, fullA && BK <= g.K, tid, ra);
  fetch<!TB>(B, g.ldb, n0, g.N, 0, g.K, fullB && BK <= g.K, tid, rb);
  stash<TA>(As[0], tid, ra);
  stash<!TB>(Bs[0], tid, rb);
  __syncthreads();

  const int kh = lane >> 5, l31 = lane & 31;
  int cur = 0;
  for (int kt = 0; kt < nk; ++kt) {
    const bool more = kt + 1 < nk;
    if (more) {
      const bool kfull = (kt + 2) * BK <= g.K;  // block-uniform
      fetch<TA>(A, g.lda, m0, g.M, (kt + 1) * BK, g.K, fullA && kfull, tid, ra);
      fetch<!TB>(B, g.ldb, n0, g.N, (kt + 1) * BK, g.K, fullB && kfull, tid, rb);
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
      stash<TA>(As[cur ^ 1], tid, ra);
      stash<!TB>(Bs[cur ^ 1], tid, rb);
    }
    __syncthreads();
    cur ^= 1;
  }

  // epilogue: C/D layout of 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
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
          float* cp = C + (int64_t)row * g.ldc + col;
          float v = acc[i][j][r] + bv;
          if (g.accumulate) v += *cp;
          if (g.act == 1) v = sk_sigmoid(v);
          *cp = v;
        }
      }
    }
}

}  // namespace

extern "C" int sk_gemm_f32(const float* A, const float* B, float* C, const float* bias, int M, int N, int K, int lda,
                           int ldb, int ldc, int transA, int transB, int accumulate, int act, int batch, int64_t sA,
                           int64_t sB, int64_t sC, int64_t sbias, sk_stream_t stream) {
  SK_CHECK_ARG(A && B && C, "sk_gemm_f32: null pointer");
  SK_CHECK_ARG(M > 0 && N > 0 && K > 0 && batch > 0 && batch <= 65535, "sk_gemm_f32: bad sizes M=%d N=%d K=%d batch=%d", M, N, K, batch);
  SK_CHECK_ARG(lda >= (transA ? M : K) && ldb >= (transB ? K : N) && ldc >= N, "sk_gemm_f32: leading dimension too small");
  SK_CHECK_ARG(act == 0 || act == 1, "sk_gemm_f32: unknown activation %d", act);
  GemmArgs g;
  g.A = A; g.B = B; g.C = C; g.bias = bias;
  g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
  g.accumulate = accumulate; g.act = act;
  g.vecA = ((uintptr_t)A % 16 == 0) && (lda % 4 == 0) && (sA % 4 == 0);
  g.vecB = ((uintptr_t)B % 16 == 0) && (ldb % 4 == 0) && (sB % 4 == 0);
  g.tilesN = (int)sk_cdiv(N, BN);
  g.sA = sA; g.sB = sB; g.sC = sC; g.sbias = sbias;
  const int64_t tiles = sk_cdiv(M, BM) * g.tilesN;
  SK_CHECK_ARG(tiles < (1ll << 31), "sk_gemm_f32: too many tiles");
  dim3 grid((unsigned)tiles, 1, (unsigned)batch);
  hipStream_t st = (hipStream_t)stream;
  if (!transA && !transB)
    hipLaunchKernelGGL((gemm_f32_kernel<false, false>), grid, dim3(256), 0, st, g);
  else if (!transA && transB)
    hipLaunchKernelGGL((gemm_f32_kernel<false, true>), grid, dim3(256), 0, st, g);
  else if (transA && !transB)
    hipLaunchKernelGGL((gemm_f32_kernel<true, false>), grid, dim3(256), 0, st, g);
  else
    hipLaunchKernelGGL((gemm_f32_kernel<true, true>), grid, dim3(256), 0, st, g);
  SK_CHECK_LAUNCH("sk_gemm_f32");
  return SK_OK;
}
