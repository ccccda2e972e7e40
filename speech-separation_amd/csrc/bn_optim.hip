// bn_optim.hip -- streaming (HBM-bound) kernels around the BLSTM: BatchNorm1d statistics /
// apply / backward over a (rows, C) matrix, column sums (bias gradients), sigmoid backward,
// global grad-norm and the fused clip + Adam update on one flat parameter buffer.
//
// References: nn.BatchNorm1d at archs/uPIT.py:119,138 (statistics over all B*T_max rows, padding
// included); clip_grad_norm_(0.25) + Adam.step() at steps/train_qsub.py:121-122.
// Every reduction is a fixed-order two-level sum (per-block partials, then one finalize block),
// so results are reproducible run to run.
#include "sk_common.h"

namespace {

constexpr int RCH = 256;  // rows per block in column reductions

// Column-wise partial reduction over a chunk of rows.  Block = 64 columns x 4 row-lanes.
//  MODE 0: sum x            MODE 1: sum (x-mean)^2
//  MODE 2: sum dy, sum dy*xhat (two outputs)
template <int MODE>
__global__ __launch_bounds__(256) void colred_kernel(const float* __restrict__ x, const float* __restrict__ aux,
                                                     const float* __restrict__ mean, const float* __restrict__ var,
                                                     float eps, int R, int C, int ld,
                                                     float* __restrict__ part0, float* __restrict__ part1) {
  __shared__ float s0[4][64], s1[4][64];
  const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cx;
  const int r0 = blockIdx.y * RCH, r1 = min(R, r0 + RCH);
  float a0 = 0.f, a1 = 0.f;
  if (c < C) {
    float mu = 0.f, rs = 0.f;
    if (MODE >= 1) mu = mean[c];
    if (MODE == 2) rs = 1.0f / sqrtf(var[c] + eps);
    for (int r = r0 + ry; r < r1; r += 4) {
      const float v = x[(int64_t)r * ld + c];
      if (MODE == 0) {
        a0 += v;
      } else if (MODE == 1) {
        const float d = v - mu;
        a0 += d * d;
      } else {
        const float dy = aux[(int64_t)r * ld + c];
        a0 += dy;
        a1 += dy * ((v - mu) * rs);
      }
    }
  }
  s0[ry][cx] = a0;
  s1[ry][cx] = a1;
  __syncthreads();
  if (ry == 0 && c < C) {
    part0[(int64_t)blockIdx.y * C + c] = (s0[0][cx] + s0[1][cx]) + (s0[2][cx] + s0[3][cx]);
    if (MODE == 2) part1[(int64_t)blockIdx.y * C + c] = (s1[0][cx] + s1[1][cx]) + (s1[2][cx] + s1[3][cx]);
  }
}

// out[c] = scale * sum_chunks part[chunk][c]  (+ out[c] if accumulate)
__global__ __launch_bounds__(256) void colfin_kernel(const float* __restrict__ part, int nch, int C, float scale,
                                                     int accumulate, float* __restrict__ out) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  float a = 0.f;
  for (int k = 0; k < nch; ++k) a += part[(int64_t)k * C + c];
  a *= scale;
  out[c] = accumulate ? out[c] + a : a;
}

// var[c] = (sum over the R stored rows of (x - mean)^2  +  (count - R) * mean^2) / count: the biased variance over `count`
// rows of which count - R are zero rows that are not stored (packed sequences: the reference normalises over the
// zero-padded (B, T_max) grid, archs/uPIT.py:135-138)
__global__ __launch_bounds__(256) void colfin_var_kernel(const float* __restrict__ part, int nch, int C, float missing,
                                                         float inv_count, const float* __restrict__ mean,
                                                         float* __restrict__ var) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  float a = 0.f;
  for (int k = 0; k < nch; ++k) a += part[(int64_t)k * C + c];
  const float mu = mean[c];
  var[c] = (a + missing * mu * mu) * inv_count;
}

// guard (may be NULL): the recurrence's sticky status word -- non-zero after a launch whose bounded wait gave up, whose
// outputs (and therefore these batch statistics) are garbage: the running statistics are then left alone
__global__ __launch_bounds__(256) void bn_running_kernel(const float* __restrict__ mean, const float* __restrict__ var,
                                                         float* __restrict__ rmean, float* __restrict__ rvar, int64_t R,
                                                         int C, float momentum, const unsigned* __restrict__ guard) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  if (guard && guard[0] != 0u) return;
  const float unbiased = var[c] * ((float)R / (float)(R - 1));
  rmean[c] = (1.0f - momentum) * rmean[c] + momentum * mean[c];
  rvar[c] = (1.0f - momentum) * rvar[c] + momentum * unbiased;
}

__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ x, const float* __restrict__ mean,
                                                       const float* __restrict__ var, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, float* __restrict__ out,
                                                       int64_t total, int C, float eps) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % C);
    const float rs = 1.0f / sqrtf(var[c] + eps);
    out[i] = (x[i] - mean[c]) * rs * gamma[c] + beta[c];
  }
}

// BatchNorm folded into the Linear layer that follows it (reference archs/uPIT.py:138-141: lin(bn(x))):
//   bn(x)[c] = x[c] * s[c] + t[c],  s = gamma / sqrt(var + eps),  t = beta - mean * s
//   lin(bn(x))[o] = sum_c x[c] * (W[o][c] * s[c]) + (b[o] + sum_c W[o][c] * t[c])
// One workgroup per output row o: writes the folded row Wf[o][:] and the folded bias bf[o]; workgroup 0 also writes
// s and t (the backward pass needs them).  The normalised activations are never materialised.
__global__ __launch_bounds__(256) void bn_fold_kernel(const float* __restrict__ W, const float* __restrict__ b,
                                                      const float* __restrict__ mean, const float* __restrict__ var,
                                                      const float* __restrict__ gamma, const float* __restrict__ beta,
                                                      float eps, int C, float* __restrict__ Wf, int ldf,
                                                      float* __restrict__ bf, float* __restrict__ s_out,
                                                      float* __restrict__ t_out) {
  __shared__ float red[256];
  const int o = blockIdx.x;
  float acc = 0.f;
  for (int c = threadIdx.x; c < C; c += 256) {
    const float sc = gamma[c] / sqrtf(var[c] + eps);
    const float tc = beta[c] - mean[c] * sc;
    const float w = W[(size_t)o * C + c];
    Wf[(size_t)o * ldf + c] = w * sc;
    acc += w * tc;
    if (o == 0) {
      s_out[c] = sc;
      t_out[c] = tc;
    }
  }
  for (int c = C + threadIdx.x; c < ldf; c += 256) Wf[(size_t)o * ldf + c] = 0.f;  // padding columns of the folded copy
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int h = 128; h > 0; h >>= 1) {  // fixed-order tree: reproducible
    if (threadIdx.x < h) red[threadIdx.x] += red[threadIdx.x + h];
    __syncthreads();
  }
  if (threadIdx.x == 0) bf[o] = (b ? b[o] : 0.f) + red[0];
}

// Gradient of the UNFOLDED weight from G = dz^T x (the product against the raw activations):
//   dW[o][c] (+)= G[o][c] * s[c] + dzsum[o] * t[c]        (since dz^T bn(x) = (dz^T x) diag(s) + colsum(dz) t^T)
__global__ __launch_bounds__(256) void bn_unfold_grad_kernel(const float* __restrict__ G, int ldg, const float* __restrict__ dzsum,
                                                             const float* __restrict__ s, const float* __restrict__ t,
                                                             float* __restrict__ dW, int O, int C, int accumulate) {
  const int64_t total = (int64_t)O * C;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int o = (int)(i / C), c = (int)(i - (int64_t)o * C);
    const float v = G[(size_t)o * ldg + c] * s[c] + dzsum[o] * t[c];
    dW[i] = accumulate ? dW[i] + v : v;
  }
}

// dx = gamma * rstd / N * (N*dy - dbeta - xhat * dgamma), N = rows the sums dbeta / dgamma (and mean / var) cover
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                           const float* __restrict__ mean, const float* __restrict__ var,
                                                           const float* __restrict__ gamma,
                                                           const float* __restrict__ dgamma,
                                                           const float* __restrict__ dbeta, float* __restrict__ dx,
                                                           int64_t total, float count, int C, float eps) {
  const float invR = 1.0f / count;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % C);
    const float rs = 1.0f / sqrtf(var[c] + eps);
    const float xh = (x[i] - mean[c]) * rs;
    dx[i] = gamma[c] * rs * (dy[i] - invR * (dbeta[c] + xh * dgamma[c]));
  }
}

__global__ __launch_bounds__(256) void sigmoid_bwd_kernel(const float* __restrict__ dm, const float* __restrict__ m,
                                                          float* __restrict__ dz, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float v = m[i];
    dz[i] = dm[i] * v * (1.0f - v);
  }
}

// dst (R, ld_dst) <- src (R, C) with row stride ld_src; columns C..ld_dst-1 zero.  One pass (the engine pads the
// F = 257 input features to 260 columns so that rows of both GEMM operands are 16-byte aligned).
__global__ __launch_bounds__(256) void pad_rows_kernel(const float* __restrict__ src, float* __restrict__ dst, int64_t R,
                                                       int64_t R_pad, int C, int ld_src, int ld_dst) {
  const int64_t total = R_pad * ld_dst;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / ld_dst;
    const int c = (int)(i - r * ld_dst);
    dst[i] = (c < C && r < R) ? src[r * ld_src + c] : 0.f;
  }
}

constexpr int NORM_BLOCKS = 1024;

__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ g, int64_t n, float* __restrict__ part) {
  __shared__ float red[4];
  float a = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float v = g[i];
    a += v * v;
  }
  const float s = sk_block_sum256(a, red);
  if (threadIdx.x == 0) part[blockIdx.x] = s;
}

// scal[0] = total norm, scal[1] = clip coefficient, scal[2] = 1 if this step is to be SKIPPED, scal[3] = number of
// skipped steps so far.  guard (may be NULL): a device word that is non-zero when the gradients of this step cannot be
// trusted (a persistent recurrence launch timed out on some rank: sepkern/engine.py puts the LSTM workspace's sticky
// status word behind the flat gradient, where the data-parallel all-reduce sums it over the ranks).
__global__ __launch_bounds__(256) void norm_fin_kernel(const float* __restrict__ part, int nb, float max_norm,
                                                       const float* __restrict__ guard, float* __restrict__ scal) {
  __shared__ float red[4];
  float a = 0.f;
  for (int i = threadIdx.x; i < nb; i += 256) a += part[i];
  const float s = sk_block_sum256(a, red);
  if (threadIdx.x == 0) {
    const float norm = sqrtf(s);
    const float coef = max_norm / (norm + 1e-6f);
    const bool skip = guard && guard[0] != 0.f;
    scal[0] = norm;
    scal[1] = coef < 1.0f ? coef : 1.0f;
    scal[2] = skip ? 1.f : 0.f;
    if (skip) scal[3] += 1.f;
  }
}

__global__ __launch_bounds__(256) void clip_adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                        float* __restrict__ m, float* __restrict__ v, int64_t n,
                                                        const float* __restrict__ scal, float lr, float beta1,
                                                        float beta2, float eps, int step) {
  if (scal[2] != 0.f) return;  // untrusted gradients never reach the weights or the moments (grid-uniform)
  const float coef = scal[1];
  // bias corrections in double, as torch.optim.Adam's scalar path does, for the number of updates actually APPLIED:
  // the caller counts calls, scal[3] counts the calls that were skipped (this one is not)
  const double applied = (double)(step - (int)scal[3]);
  const float bc1 = (float)(1.0 - pow((double)beta1, applied));
  const float bc2_sqrt = (float)sqrt(1.0 - pow((double)beta2, applied));
  const float step_size = lr / bc1;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float gi = g[i] * coef;
    const float mi = beta1 * m[i] + (1.0f - beta1) * gi;
    const float vi = beta2 * v[i] + (1.0f - beta2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] -= step_size * (mi / denom);
  }
}

inline unsigned stream_blocks(int64_t n) {
  int64_t b = sk_cdiv(n, 256);
  return (unsigned)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

}  // namespace

extern "C" size_t sk_bn_workspace_bytes(int R, int C) {
  return sk_align(2 * (size_t)sk_cdiv(R, RCH) * C * sizeof(float), 256);
}

static int colreduce(int mode, const float* x, const float* aux, const float* mean, const float* var, float eps, int R,
                     int C, int ld, float* p0, float* p1, hipStream_t st) {
  dim3 grid((unsigned)sk_cdiv(C, 64), (unsigned)sk_cdiv(R, RCH));
  if (mode == 0)
    hipLaunchKernelGGL(colred_kernel<0>, grid, dim3(256), 0, st, x, aux, mean, var, eps, R, C, ld, p0, p1);
  else if (mode == 1)
    hipLaunchKernelGGL(colred_kernel<1>, grid, dim3(256), 0, st, x, aux, mean, var, eps, R, C, ld, p0, p1);
  else
    hipLaunchKernelGGL(colred_kernel<2>, grid, dim3(256), 0, st, x, aux, mean, var, eps, R, C, ld, p0, p1);
  SK_CHECK_LAUNCH("colred_kernel");
  return SK_OK;
}

extern "C" int sk_bn_stats(const float* x, int R, int C, int64_t count, float* mean, float* var, void* ws,
                           sk_stream_t stream) {
  SK_CHECK_ARG(x && mean && var && ws && R > 0 && C > 0 && count >= R && count > 1, "sk_bn_stats: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  const int nch = (int)sk_cdiv(R, RCH);
  float* part = (float*)ws;
  const float inv = (float)(1.0 / (double)count);
  int rc = colreduce(0, x, nullptr, nullptr, nullptr, 0.f, R, C, C, part, nullptr, st);
  if (rc) return rc;
  hipLaunchKernelGGL(colfin_kernel, dim3((unsigned)sk_cdiv(C, 256)), dim3(256), 0, st, part, nch, C, inv, 0, mean);
  rc = colreduce(1, x, nullptr, mean, nullptr, 0.f, R, C, C, part, nullptr, st);
  if (rc) return rc;
  hipLaunchKernelGGL(colfin_var_kernel, dim3((unsigned)sk_cdiv(C, 256)), dim3(256), 0, st, part, nch, C, (float)(count - R), inv,
                     mean, var);
  SK_CHECK_LAUNCH("sk_bn_stats");
  return SK_OK;
}

extern "C" int sk_bn_update_running(const float* mean, const float* var, float* running_mean, float* running_var,
                                    int64_t count, int C, float momentum, const void* guard, sk_stream_t stream) {
  SK_CHECK_ARG(mean && var && running_mean && running_var && count > 1 && C > 0, "sk_bn_update_running: bad arguments");
  hipLaunchKernelGGL(bn_running_kernel, dim3((unsigned)sk_cdiv(C, 256)), dim3(256), 0, (hipStream_t)stream, mean, var,
                     running_mean, running_var, count, C, momentum, (const unsigned*)guard);
  SK_CHECK_LAUNCH("sk_bn_update_running");
  return SK_OK;
}

extern "C" int sk_bn_apply(const float* x, const float* mean, const float* var, const float* gamma, const float* beta,
                           float* out, int R, int C, float eps, sk_stream_t stream) {
  SK_CHECK_ARG(x && mean && var && gamma && beta && out && R > 0 && C > 0, "sk_bn_apply: bad arguments");
  const int64_t total = (int64_t)R * C;
  hipLaunchKernelGGL(bn_apply_kernel, dim3(stream_blocks(total)), dim3(256), 0, (hipStream_t)stream, x, mean, var,
                     gamma, beta, out, total, C, eps);
  SK_CHECK_LAUNCH("sk_bn_apply");
  return SK_OK;
}

extern "C" int sk_bn_fold(const float* W, const float* b, const float* mean, const float* var, const float* gamma,
                          const float* beta, float eps, int O, int C, float* Wf, int ldf, float* bf, float* s, float* t,
                          sk_stream_t stream) {
  SK_CHECK_ARG(W && mean && var && gamma && beta && Wf && bf && s && t && O > 0 && C > 0 && ldf >= C, "sk_bn_fold: bad arguments");
  hipLaunchKernelGGL(bn_fold_kernel, dim3((unsigned)O), dim3(256), 0, (hipStream_t)stream, W, b, mean, var, gamma, beta, eps,
                     C, Wf, ldf, bf, s, t);
  SK_CHECK_LAUNCH("sk_bn_fold");
  return SK_OK;
}

extern "C" int sk_bn_unfold_grad(const float* G, int ldg, const float* dzsum, const float* s, const float* t, float* dW,
                                 int O, int C, int accumulate, sk_stream_t stream) {
  SK_CHECK_ARG(G && dzsum && s && t && dW && O > 0 && C > 0 && ldg >= C, "sk_bn_unfold_grad: bad arguments");
  const int64_t total = (int64_t)O * C;
  hipLaunchKernelGGL(bn_unfold_grad_kernel, dim3(stream_blocks(total)), dim3(256), 0, (hipStream_t)stream, G, ldg, dzsum, s,
                     t, dW, O, C, accumulate);
  SK_CHECK_LAUNCH("sk_bn_unfold_grad");
  return SK_OK;
}

extern "C" int sk_bn_bwd_sums(const float* dout, const float* x, const float* mean, const float* var, float* dgamma,
                              float* dbeta, void* ws, int R, int C, float eps, sk_stream_t stream) {
  SK_CHECK_ARG(dout && x && mean && var && dgamma && dbeta && ws && R > 0 && C > 0, "sk_bn_bwd_sums: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  const int nch = (int)sk_cdiv(R, RCH);
  float* p0 = (float*)ws;
  float* p1 = p0 + (size_t)nch * C;
  int rc = colreduce(2, x, dout, mean, var, eps, R, C, C, p0, p1, st);
  if (rc) return rc;
  hipLaunchKernelGGL(colfin_kernel, dim3((unsigned)sk_cdiv(C, 256)), dim3(256), 0, st, p0, nch, C, 1.0f, 0, dbeta);
  hipLaunchKernelGGL(colfin_kernel, dim3((unsigned)sk_cdiv(C, 256)), dim3(256), 0, st, p1, nch, C, 1.0f, 0, dgamma);
  SK_CHECK_LAUNCH("sk_bn_bwd_sums");
  return SK_OK;
}

extern "C" int sk_bn_bwd_apply(const float* dout, const float* x, const float* mean, const float* var,
                               const float* gamma, const float* dgamma, const float* dbeta, float* dx, int R, int C,
                               double count, float eps, sk_stream_t stream) {
  SK_CHECK_ARG(dout && x && mean && var && gamma && dgamma && dbeta && dx && R > 0 && C > 0 && count >= 1.0,
               "sk_bn_bwd_apply: bad arguments");
  const int64_t total = (int64_t)R * C;
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(stream_blocks(total)), dim3(256), 0, (hipStream_t)stream, dout, x, mean,
                     var, gamma, dgamma, dbeta, dx, total, (float)count, C, eps);
  SK_CHECK_LAUNCH("sk_bn_bwd_apply");
  return SK_OK;
}

extern "C" int sk_bn_bwd(const float* dout, const float* x, const float* mean, const float* var, const float* gamma,
                         float* dx, float* dgamma, float* dbeta, void* ws, int R, int C, float eps,
                         sk_stream_t stream) {
  SK_CHECK_ARG(R > 1, "sk_bn_bwd: bad arguments");
  int rc = sk_bn_bwd_sums(dout, x, mean, var, dgamma, dbeta, ws, R, C, eps, stream);
  if (rc) return rc;
  return sk_bn_bwd_apply(dout, x, mean, var, gamma, dgamma, dbeta, dx, R, C, (double)R, eps, stream);
}

extern "C" int sk_colsum(const float* x, int R, int C, int ld, float* out, int accumulate, void* ws,
                         sk_stream_t stream) {
  SK_CHECK_ARG(x && out && ws && R > 0 && C > 0 && ld >= C, "sk_colsum: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  const int nch = (int)sk_cdiv(R, RCH);
  float* part = (float*)ws;
  int rc = colreduce(0, x, nullptr, nullptr, nullptr, 0.f, R, C, ld, part, nullptr, st);
  if (rc) return rc;
  hipLaunchKernelGGL(colfin_kernel, dim3((unsigned)sk_cdiv(C, 256)), dim3(256), 0, st, part, nch, C, 1.0f, accumulate, out);
  SK_CHECK_LAUNCH("sk_colsum");
  return SK_OK;
}

extern "C" int sk_sigmoid_bwd(const float* dmask, const float* m, float* dz, int64_t n, sk_stream_t stream) {
  SK_CHECK_ARG(dmask && m && dz && n > 0, "sk_sigmoid_bwd: bad arguments");
  hipLaunchKernelGGL(sigmoid_bwd_kernel, dim3(stream_blocks(n)), dim3(256), 0, (hipStream_t)stream, dmask, m, dz, n);
  SK_CHECK_LAUNCH("sk_sigmoid_bwd");
  return SK_OK;
}

extern "C" int sk_pad_rows(const float* src, int64_t R, int C, int ld_src, float* dst, int ld_dst, int64_t R_pad,
                           sk_stream_t stream) {
  SK_CHECK_ARG(src && dst && src != dst && R > 0 && C > 0 && ld_src >= C && ld_dst >= C && R_pad >= R, "sk_pad_rows: bad arguments");
  hipLaunchKernelGGL(pad_rows_kernel, dim3(stream_blocks(R_pad * ld_dst)), dim3(256), 0, (hipStream_t)stream, src, dst, R, R_pad,
                     C, ld_src, ld_dst);
  SK_CHECK_LAUNCH("sk_pad_rows");
  return SK_OK;
}

extern "C" size_t sk_optim_workspace_bytes(int64_t n) {
  (void)n;
  return sk_align(NORM_BLOCKS * sizeof(float), 256);
}

extern "C" int sk_grad_norm(const float* g, int64_t n, float max_norm, const float* guard, float* scal, void* ws,
                            sk_stream_t stream) {
  SK_CHECK_ARG(g && scal && ws && n > 0, "sk_grad_norm: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  const int nb = (int)(sk_cdiv(n, 256) < NORM_BLOCKS ? sk_cdiv(n, 256) : NORM_BLOCKS);
  hipLaunchKernelGGL(sumsq_kernel, dim3((unsigned)nb), dim3(256), 0, st, g, n, (float*)ws);
  hipLaunchKernelGGL(norm_fin_kernel, dim3(1), dim3(256), 0, st, (const float*)ws, nb, max_norm, guard, scal);
  SK_CHECK_LAUNCH("sk_grad_norm");
  return SK_OK;
}

extern "C" int sk_clip_adam(float* p, const float* g, float* m, float* v, int64_t n, const float* scal, float lr,
                            float beta1, float beta2, float eps, int step, sk_stream_t stream) {
  SK_CHECK_ARG(p && g && m && v && scal && n > 0 && step >= 1, "sk_clip_adam: bad arguments");
  hipLaunchKernelGGL(clip_adam_kernel, dim3(stream_blocks(n)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, scal,
                     lr, beta1, beta2, eps, step);
  SK_CHECK_LAUNCH("sk_clip_adam");
  return SK_OK;
}
