// pit.hip -- utterance-level permutation-invariant MSE (uPIT) loss, forward and backward.
//
// Reference: compute_loss, archs/uPIT.py:181-197,206.  The reference evaluates S! full
// element-wise passes; here ONE streaming pass forms the S x S pairwise per-utterance SSE matrix
//   pair[b][s][r] = sum_{t,f} (mask[t,b,s,f] * mix[t,b,f] - src_r[t,b,f])^2
// by wavefront reduction (HBM-bound: (2S+1)*F*4 bytes per frame), and the S! permutation sums,
// the per-utterance arg-min and the scalar loss are formed from it by a tiny finalize kernel.
// Reductions use fixed-order partial sums (no atomics), so results are run-to-run reproducible.
//
// Row layouts: padded time-major (row of (t, b) = t*B + b, zeros past an utterance's end) or, with an offset table
// `offs` (T+1 entries), PACKED rows exactly as torch's PackedSequence.data holds them (archs/uPIT.py:46,167: row of
// (t, b) = offs[t] + b for t < lens[b], lens sorted descending) -- the layout the reference's collator produces.
#include "sk_common.h"

namespace {

constexpr int MAXS = 4;
constexpr int TCH = 16;  // frames per block in the pairwise pass
constexpr int RB = 4;    // (frame, utterance) rows per block in the backward pass

struct SrcPtrs {
  const float* p[MAXS];
};

template <int S>
__global__ __launch_bounds__(256) void pit_pair_kernel(const float* __restrict__ mask, const float* __restrict__ mix,
                                                       SrcPtrs src, const int32_t* __restrict__ lens,
                                                       const int32_t* __restrict__ offs, int T, int B, int F,
                                                       float* __restrict__ partial /* (B, nch, S*S) */) {
  __shared__ float red[4];
  const int b = blockIdx.y, ch = blockIdx.x, nch = gridDim.x;
  float acc[S][S];
#pragma unroll
  for (int s = 0; s < S; ++s)
#pragma unroll
    for (int r = 0; r < S; ++r) acc[s][r] = 0.f;
  const int Tb = offs ? min(T, lens[b]) : T;  // packed rows end with the utterance
  const int t0 = ch * TCH, nfr = max(0, min(Tb, t0 + TCH) - t0);
  auto element = [&](int dt, int f) {
    const int64_t row = (offs ? (int64_t)offs[t0 + dt] : (int64_t)(t0 + dt) * B) + b;
    const float mx = mix[row * F + f];
    float sv[S];
#pragma unroll
    for (int r = 0; r < S; ++r) sv[r] = src.p[r][row * F + f];
#pragma unroll
    for (int s = 0; s < S; ++s) {
      const float mm = mask[row * (int64_t)(S * F) + s * F + f] * mx;
#pragma unroll
      for (int r = 0; r < S; ++r) {
        const float d = mm - sv[r];
        acc[s][r] += d * d;
      }
    }
  };
  // r06: whole 256-bin sweeps with the thread's bin fixed and the chunk's frames as the (unrolled) inner loop -- no index
  // arithmetic per element and (2S + 1) x 4 independent loads in flight per thread; r05 walked the chunk as one flat range and
  // divided every element's index by the run-time F (more instructions than the element's arithmetic, one load batch in flight:
  // 26.5 us for 65.8 MB).  The F - 256 floor(F / 256) left-over bins (F = 257: one) x the chunk's frames are dealt to the first
  // threads as a flat range, so that they do not cost a nearly empty sweep.
  const int fullc = F & ~255;
  for (int f0 = 0; f0 < fullc; f0 += 256) {
    const int f = f0 + threadIdx.x;
#pragma unroll 4
    for (int dt = 0; dt < nfr; ++dt) element(dt, f);
  }
  const int tail = F - fullc;
  for (int i = threadIdx.x; i < nfr * tail; i += 256) {
    const int dt = i / tail;
    element(dt, fullc + i - dt * tail);
  }
#pragma unroll
  for (int s = 0; s < S; ++s)
#pragma unroll
    for (int r = 0; r < S; ++r) {
      const float v = sk_block_sum256(acc[s][r], red);
      if (threadIdx.x == 0) partial[((int64_t)b * nch + ch) * (S * S) + s * S + r] = v;
    }
}

// Lexicographic permutation number `idx` of {0..S-1} (itertools.permutations order).
__device__ __forceinline__ void nth_perm(int idx, int S, int* perm) {
  int avail[MAXS];
  for (int i = 0; i < S; ++i) avail[i] = i;
  int fact = 1;
  for (int i = 2; i < S; ++i) fact *= i;  // (S-1)!
  for (int i = 0; i < S; ++i) {
    const int q = idx / fact;
    idx -= q * fact;
    perm[i] = avail[q];
    for (int j = q; j < S - 1 - i; ++j) avail[j] = avail[j + 1];
    if (S - 1 - i > 0) fact /= (S - 1 - i);
  }
}

__global__ __launch_bounds__(256) void pit_finalize_kernel(const float* __restrict__ partial, int nch,
                                                           const int32_t* __restrict__ lens, int B, int F, int S,
                                                           const float* __restrict__ norm_dev, float* __restrict__ pair_sse,
                                                           float* __restrict__ perm_loss, int32_t* __restrict__ best_perm,
                                                           float* __restrict__ out) {
  __shared__ float red[4];
  int nperm = 1;
  for (int i = 2; i <= S; ++i) nperm *= i;
  float my_min = 0.f, my_len = 0.f;
  for (int b = threadIdx.x; b < B; b += 256) {
    float pr[MAXS * MAXS];
    for (int q = 0; q < S * S; ++q) {
      float a = 0.f;
      for (int c = 0; c < nch; ++c) a += partial[((int64_t)b * nch + c) * (S * S) + q];
      pr[q] = a;
      pair_sse[(int64_t)b * S * S + q] = a;
    }
    float best = 0.f;
    int bi = 0;
    for (int p = 0; p < nperm; ++p) {
      int perm[MAXS];
      nth_perm(p, S, perm);
      float l = 0.f;
      for (int s = 0; s < S; ++s) l += pr[s * S + perm[s]];
      perm_loss[(int64_t)p * B + b] = l;
      if (p == 0 || l < best) {
        best = l;
        bi = p;
      }
    }
    best_perm[b] = bi;
    my_min += best;
    my_len += (float)lens[b];
  }
  const float tot = sk_block_sum256(my_min, red);
  const float len = sk_block_sum256(my_len, red);
  if (threadIdx.x == 0) {
    const float norm = norm_dev ? norm_dev[0] : len * (float)F;
    const float lsum = tot / (float)S;
    out[0] = lsum / norm;
    out[1] = norm;
    out[2] = lsum;
  }
}

template <int S>
__global__ __launch_bounds__(256) void pit_bwd_kernel(const float* __restrict__ mask, const float* __restrict__ mix,
                                                      SrcPtrs src, const int32_t* __restrict__ best_perm,
                                                      const float* __restrict__ out, const float* __restrict__ gscale,
                                                      const int32_t* __restrict__ offs, int64_t nrows, int T, int B, int F,
                                                      float* __restrict__ dmask) {
  __shared__ int perm[RB][MAXS];
  const int64_t row0 = (int64_t)blockIdx.x * RB;  // padded: row = t*B + b; packed: row = offs[t] + b
  if (threadIdx.x < RB && row0 + threadIdx.x < nrows) {
    const int64_t row = row0 + threadIdx.x;
    int b;
    if (offs) {
      int lo = 0, hi = T;  // the t with offs[t] <= row < offs[t+1]
      while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if ((int64_t)offs[mid] <= row) lo = mid; else hi = mid;
      }
      b = (int)(row - offs[lo]);
    } else {
      b = (int)(row % B);
    }
    nth_perm(best_perm[b], S, perm[threadIdx.x]);
  }
  __syncthreads();
  const float k = gscale[0] * 2.0f / ((float)S * out[1]);
  const int nel = (int)min((int64_t)RB, nrows - row0) * F;
  for (int i = threadIdx.x; i < nel; i += 256) {
    const int dr = i / F, f = i - dr * F;
    const int64_t row = row0 + dr;
    const float mx = mix[row * F + f];
#pragma unroll
    for (int s = 0; s < S; ++s) {
      const int64_t o = row * (int64_t)(S * F) + s * F + f;
      const float sv = src.p[perm[dr][s]][row * F + f];
      dmask[o] = k * (mask[o] * mx - sv) * mx;
    }
  }
}

}  // namespace

extern "C" size_t sk_pit_workspace_bytes(int T, int B, int S) {
  return sk_align((size_t)B * sk_cdiv(T, TCH) * S * S * sizeof(float), 256);
}

extern "C" int sk_pit_mse_fwd(const float* mask, const float* mix, const float* const* src_host, const int32_t* lens,
                              const int32_t* offs, int T, int B, int F, int S, const float* norm_dev, float* pair_sse,
                              float* perm_loss, int32_t* best_perm, float* out, void* ws, sk_stream_t stream) {
  SK_CHECK_ARG(mask && mix && src_host && lens && pair_sse && perm_loss && best_perm && out && ws,
               "sk_pit_mse_fwd: null pointer");
  SK_CHECK_ARG(S >= 1 && S <= MAXS, "sk_pit_mse_fwd: num_spk %d outside 1..%d", S, MAXS);
  SK_CHECK_ARG(T > 0 && B > 0 && B <= 65535 && F > 0, "sk_pit_mse_fwd: bad sizes");
  SrcPtrs sp;
  for (int s = 0; s < MAXS; ++s) sp.p[s] = s < S ? src_host[s] : nullptr;
  const int nch = (int)sk_cdiv(T, TCH);
  dim3 grid((unsigned)nch, (unsigned)B);
  float* partial = (float*)ws;
  hipStream_t st = (hipStream_t)stream;
  switch (S) {
    case 1: hipLaunchKernelGGL(pit_pair_kernel<1>, grid, dim3(256), 0, st, mask, mix, sp, lens, offs, T, B, F, partial); break;
    case 2: hipLaunchKernelGGL(pit_pair_kernel<2>, grid, dim3(256), 0, st, mask, mix, sp, lens, offs, T, B, F, partial); break;
    case 3: hipLaunchKernelGGL(pit_pair_kernel<3>, grid, dim3(256), 0, st, mask, mix, sp, lens, offs, T, B, F, partial); break;
    default: hipLaunchKernelGGL(pit_pair_kernel<4>, grid, dim3(256), 0, st, mask, mix, sp, lens, offs, T, B, F, partial); break;
  }
  SK_CHECK_LAUNCH("pit_pair_kernel");
  hipLaunchKernelGGL(pit_finalize_kernel, dim3(1), dim3(256), 0, st, partial, nch, lens, B, F, S, norm_dev,
                     pair_sse, perm_loss, best_perm, out);
  SK_CHECK_LAUNCH("pit_finalize_kernel");
  return SK_OK;
}

extern "C" int sk_pit_mse_bwd(const float* mask, const float* mix, const float* const* src_host,
                              const int32_t* best_perm, const float* out, const float* gscale, const int32_t* offs,
                              int64_t nrows, int T, int B, int F, int S, float* dmask, sk_stream_t stream) {
  SK_CHECK_ARG(mask && mix && src_host && best_perm && out && gscale && dmask, "sk_pit_mse_bwd: null pointer");
  SK_CHECK_ARG(S >= 1 && S <= MAXS, "sk_pit_mse_bwd: num_spk %d outside 1..%d", S, MAXS);
  SK_CHECK_ARG(T > 0 && B > 0 && F > 0, "sk_pit_mse_bwd: bad sizes");
  if (!offs) nrows = (int64_t)T * B;
  SK_CHECK_ARG(nrows > 0 && nrows <= (int64_t)T * B, "sk_pit_mse_bwd: %lld packed rows for T=%d B=%d", (long long)nrows, T, B);
  SrcPtrs sp;
  for (int s = 0; s < MAXS; ++s) sp.p[s] = s < S ? src_host[s] : nullptr;
  dim3 grid((unsigned)sk_cdiv(nrows, RB));
  hipStream_t st = (hipStream_t)stream;
  switch (S) {
    case 1: hipLaunchKernelGGL(pit_bwd_kernel<1>, grid, dim3(256), 0, st, mask, mix, sp, best_perm, out, gscale, offs, nrows, T, B, F, dmask); break;
    case 2: hipLaunchKernelGGL(pit_bwd_kernel<2>, grid, dim3(256), 0, st, mask, mix, sp, best_perm, out, gscale, offs, nrows, T, B, F, dmask); break;
    case 3: hipLaunchKernelGGL(pit_bwd_kernel<3>, grid, dim3(256), 0, st, mask, mix, sp, best_perm, out, gscale, offs, nrows, T, B, F, dmask); break;
    default: hipLaunchKernelGGL(pit_bwd_kernel<4>, grid, dim3(256), 0, st, mask, mix, sp, best_perm, out, gscale, offs, nrows, T, B, F, dmask); break;
  }
  SK_CHECK_LAUNCH("pit_bwd_kernel");
  return SK_OK;
}
