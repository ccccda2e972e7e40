// rsh.hip -- kernels of the recurrent-selective-hearing arch (reference archs/RSH.py) that are not
// shared with uPIT: the per-pass greedy source-assignment loss and the attention update.
//
//   loss   (archs/RSH.py:225-244): masked = mask * mix; sse[r][b] = sum_{t,f} (masked - src_r)^2;
//          sources already taken by row b are excluded (set to +inf); the minimum is taken, its
//          source marked used, and sum_b min / num_spk is this pass's loss term.
//   update (archs/RSH.py:254-257 train, :278-281 test): combos = relu(combos - [0 | mask])
//          (no relu at test time); combos = [mixture | attention], (T,B,2F).
// Streaming, HBM-bound; fixed-order reductions (no atomics).
#include "sk_common.h"

namespace {

constexpr int RMAXS = 8;
constexpr int RTCH = 8;  // frames per block in the SSE pass

struct RSrc {
  const float* p[RMAXS];
};

template <int S>
__global__ __launch_bounds__(256) void rsh_sse_kernel(const float* __restrict__ mask, const float* __restrict__ x,
                                                      int ldx, RSrc src, int T, int B, int F,
                                                      float* __restrict__ partial /* (B, nch, S) */) {
  __shared__ float red[4];
  const int b = blockIdx.y, ch = blockIdx.x, nch = gridDim.x;
  float acc[S];
#pragma unroll
  for (int r = 0; r < S; ++r) acc[r] = 0.f;
  const int tend = min(T, (ch + 1) * RTCH);
  for (int t = ch * RTCH; t < tend; ++t) {
    const int64_t row = (int64_t)t * B + b;
    for (int f = threadIdx.x; f < F; f += 256) {
      const float mm = mask[row * F + f] * x[row * ldx + f];
#pragma unroll
      for (int r = 0; r < S; ++r) {
        const float d = mm - src.p[r][row * F + f];
        acc[r] += d * d;
      }
    }
  }
#pragma unroll
  for (int r = 0; r < S; ++r) {
    const float v = sk_block_sum256(acc[r], red);
    if (threadIdx.x == 0) partial[((int64_t)b * nch + ch) * S + r] = v;
  }
}

__global__ __launch_bounds__(256) void rsh_select_kernel(const float* __restrict__ partial, int nch,
                                                         const int32_t* __restrict__ lens, int B, int F, int S,
                                                         int32_t* __restrict__ used, float* __restrict__ sse,
                                                         int32_t* __restrict__ sel, float* __restrict__ out) {
  __shared__ float red[4];
  float my_min = 0.f, my_len = 0.f;
  for (int b = threadIdx.x; b < B; b += 256) {
    float best = 0.f;
    int bi = 0;
    for (int r = 0; r < S; ++r) {
      float a = 0.f;
      for (int c = 0; c < nch; ++c) a += partial[((int64_t)b * nch + c) * S + r];
      sse[(int64_t)r * B + b] = a;
      const float v = used[(int64_t)r * B + b] ? __builtin_inff() : a;  // archs/RSH.py:229-231
      if (r == 0 || v < best) {
        best = v;
        bi = r;
      }
    }
    sel[b] = bi;
    used[(int64_t)bi * B + b] = 1;
    my_min += best;
    my_len += (float)lens[b];
  }
  const float tot = sk_block_sum256(my_min, red);
  const float len = sk_block_sum256(my_len, red);
  if (threadIdx.x == 0) {
    out[0] = tot / (float)S;
    out[1] = len * (float)F;
  }
}

__global__ __launch_bounds__(256) void rsh_bwd_kernel(const float* __restrict__ mask, const float* __restrict__ x, int ldx,
                                                      RSrc src, const int32_t* __restrict__ sel,
                                                      const float* __restrict__ gscale, int B, int F, int S,
                                                      float* __restrict__ dmask) {
  const int64_t row = blockIdx.x;  // t*B + b
  const int b = (int)(row % B);
  const float* sp = src.p[sel[b]];
  const float k = gscale[0] * 2.0f / (float)S;
  for (int f = threadIdx.x; f < F; f += 256) {
    const float mx = x[row * ldx + f];
    dmask[row * F + f] = k * (mask[row * F + f] * mx - sp[row * F + f]) * mx;
  }
}

// x_out = act(x_in - [0 | mask]) over rows of 2F; act = relu (training) or identity (test)
__global__ __launch_bounds__(256) void att_update_kernel(const float* __restrict__ xin, const float* __restrict__ mask,
                                                         float* __restrict__ xout, int64_t R, int F, int relu) {
  const int64_t total = R * 2 * F;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / (2 * F);
    const int c = (int)(i - r * 2 * F);
    float v = xin[i];
    if (c >= F) v -= mask[r * F + (c - F)];
    xout[i] = (relu && v < 0.f) ? 0.f : v;
  }
}

// dx_in = dx_out * gate; dmask = -(dx_out * gate)[attention half]; gate = (x_out > 0) with relu, 1 without
__global__ __launch_bounds__(256) void att_update_bwd_kernel(const float* __restrict__ dxout, const float* __restrict__ xout,
                                                             float* __restrict__ dxin, float* __restrict__ dmask,
                                                             int64_t R, int F, int relu) {
  const int64_t total = R * 2 * F;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / (2 * F);
    const int c = (int)(i - r * 2 * F);
    const float g = (relu && !(xout[i] > 0.f)) ? 0.f : dxout[i];
    dxin[i] = g;
    if (c >= F) dmask[r * F + (c - F)] = -g;
  }
}

inline unsigned rsh_blocks(int64_t n) {
  int64_t b = sk_cdiv(n, 256);
  return (unsigned)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

}  // namespace

extern "C" size_t sk_rsh_workspace_bytes(int T, int B, int S) {
  return sk_align((size_t)B * sk_cdiv(T, RTCH) * S * sizeof(float), 256);
}

extern "C" int sk_rsh_loss_fwd(const float* mask, const float* x, int ldx, const float* const* src_host,
                               const int32_t* lens, int T, int B, int F, int S, int32_t* used, float* sse, int32_t* sel,
                               float* out, void* ws, sk_stream_t stream) {
  SK_CHECK_ARG(mask && x && src_host && lens && used && sse && sel && out && ws, "sk_rsh_loss_fwd: null pointer");
  SK_CHECK_ARG(S >= 1 && S <= RMAXS, "sk_rsh_loss_fwd: num_spk %d outside 1..%d", S, RMAXS);
  SK_CHECK_ARG(T > 0 && B > 0 && B <= 65535 && F > 0 && ldx >= F, "sk_rsh_loss_fwd: bad sizes");
  RSrc sp;
  for (int s = 0; s < RMAXS; ++s) sp.p[s] = s < S ? src_host[s] : nullptr;
  const int nch = (int)sk_cdiv(T, RTCH);
  dim3 grid((unsigned)nch, (unsigned)B);
  float* partial = (float*)ws;
  hipStream_t st = (hipStream_t)stream;
#define SK_RSH_CASE(N) \
  case N: hipLaunchKernelGGL(rsh_sse_kernel<N>, grid, dim3(256), 0, st, mask, x, ldx, sp, T, B, F, partial); break;
  switch (S) {
    SK_RSH_CASE(1) SK_RSH_CASE(2) SK_RSH_CASE(3) SK_RSH_CASE(4) SK_RSH_CASE(5) SK_RSH_CASE(6) SK_RSH_CASE(7)
    default: hipLaunchKernelGGL(rsh_sse_kernel<8>, grid, dim3(256), 0, st, mask, x, ldx, sp, T, B, F, partial); break;
  }
#undef SK_RSH_CASE
  SK_CHECK_LAUNCH("rsh_sse_kernel");
  hipLaunchKernelGGL(rsh_select_kernel, dim3(1), dim3(256), 0, st, partial, nch, lens, B, F, S, used, sse, sel, out);
  SK_CHECK_LAUNCH("rsh_select_kernel");
  return SK_OK;
}

extern "C" int sk_rsh_loss_bwd(const float* mask, const float* x, int ldx, const float* const* src_host,
                               const int32_t* sel, const float* gscale, int T, int B, int F, int S, float* dmask,
                               sk_stream_t stream) {
  SK_CHECK_ARG(mask && x && src_host && sel && gscale && dmask, "sk_rsh_loss_bwd: null pointer");
  SK_CHECK_ARG(S >= 1 && S <= RMAXS && T > 0 && B > 0 && F > 0 && ldx >= F, "sk_rsh_loss_bwd: bad sizes");
  RSrc sp;
  for (int s = 0; s < RMAXS; ++s) sp.p[s] = s < S ? src_host[s] : nullptr;
  hipLaunchKernelGGL(rsh_bwd_kernel, dim3((unsigned)((int64_t)T * B)), dim3(256), 0, (hipStream_t)stream, mask, x, ldx, sp,
                     sel, gscale, B, F, S, dmask);
  SK_CHECK_LAUNCH("rsh_bwd_kernel");
  return SK_OK;
}

extern "C" int sk_att_update(const float* x_in, const float* mask, float* x_out, int64_t rows, int F, int relu,
                             sk_stream_t stream) {
  SK_CHECK_ARG(x_in && mask && x_out && rows > 0 && F > 0, "sk_att_update: bad arguments");
  hipLaunchKernelGGL(att_update_kernel, dim3(rsh_blocks(rows * 2 * F)), dim3(256), 0, (hipStream_t)stream, x_in, mask,
                     x_out, rows, F, relu);
  SK_CHECK_LAUNCH("sk_att_update");
  return SK_OK;
}

extern "C" int sk_att_update_bwd(const float* dx_out, const float* x_out, float* dx_in, float* dmask, int64_t rows, int F,
                                 int relu, sk_stream_t stream) {
  SK_CHECK_ARG(dx_out && x_out && dx_in && dmask && rows > 0 && F > 0, "sk_att_update_bwd: bad arguments");
  hipLaunchKernelGGL(att_update_bwd_kernel, dim3(rsh_blocks(rows * 2 * F)), dim3(256), 0, (hipStream_t)stream, dx_out,
                     x_out, dx_in, dmask, rows, F, relu);
  SK_CHECK_LAUNCH("sk_att_update_bwd");
  return SK_OK;
}
