// packed.hip -- row movers for PACKED sequence batches (HBM-bound copies, no arithmetic).
//
// Reference: the collator hands the network a torch PackedSequence (archs/uPIT.py:46,132,135): only the
// sum(lens) valid frames of a length-sorted batch exist, time-major -- row of (t, j) is offs[t] + j for
// j < n_t, n_t = offs[t+1] - offs[t] = number of utterances longer than t.  The engine keeps every
// activation in that layout (no product, statistic or store ever touches a padded frame), so:
//   pack   : zero-padded (T, B, C) -> packed (R, C)   (callers that hold padded tensors: RSH arch, tools)
//   unpack : packed (R, C) -> zero-padded (T, B, C)
//   hprev  : the recurrent input of every row -- the layer output one step earlier in processing order,
//            or h0 for a row's first step -- gathered into (R, 2H) so that the recurrent weight gradient
//            dW_hh[d] = dG[:, d]^T hprev[:, d] is ONE plain product over the packed rows (in the padded
//            layout the shift was a constant B rows; packed, it is n_{t-1} / n_t rows and varies with t).
// `perm` (B entries or NULL): sorted position j holds the caller's utterance perm[j] (callers whose padded
// batch is not length-sorted).
#include "sk_common.h"

namespace {

constexpr int ROWS = 8;  // sorted positions per block

// PACK: dst[offs[t] + j][c] = src[(t B + perm[j]) C + c];  !PACK: the other way, fill[c] (or 0) at j >= n_t
template <bool PACK>
__global__ __launch_bounds__(256) void pack_rows_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                        const int32_t* __restrict__ offs, const int32_t* __restrict__ perm,
                                                        const float* __restrict__ fill, int B, int C, int ld_packed) {
  const int t = blockIdx.x, j0 = blockIdx.y * ROWS;
  const int o = offs[t], n = offs[t + 1] - o;
  const int jend = min(B, j0 + ROWS);
  for (int j = j0; j < jend; ++j) {
    const int b = perm ? perm[j] : j;
    const size_t prow = ((size_t)t * B + b) * C, krow = (size_t)(o + j) * ld_packed;
    if (PACK) {
      if (j >= n) break;  // sorted: nothing valid behind the first invalid position
      for (int c = threadIdx.x; c < C; c += 256) dst[krow + c] = src[prow + c];
    } else {
      for (int c = threadIdx.x; c < C; c += 256) dst[prow + c] = j < n ? src[krow + c] : (fill ? fill[c] : 0.f);
    }
  }
}

typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

template <bool BF>
__global__ __launch_bounds__(256) void hprev_rows_kernel(const float* __restrict__ y, int ldy, const float* __restrict__ h0,
                                                         const int32_t* __restrict__ offs, int T, int B, int H,
                                                         void* __restrict__ out, int ldo) {
  const int t = blockIdx.x, j0 = blockIdx.y * ROWS;
  const int o = offs[t], n = offs[t + 1] - o;
  const int n_next = t + 1 < T ? offs[t + 2] - offs[t + 1] : 0;
  const int jend = min(n, j0 + ROWS);
  const int q = H >> 2;  // float4 per direction
  for (int j = j0; j < jend; ++j) {
    // forward direction: the output of step t-1 (h0 at t = 0); reverse: the output of step t+1 (h0 at the row's last frame)
    const float* s0 = t > 0 ? y + (size_t)(offs[t - 1] + j) * ldy : h0 + (size_t)j * H;
    const float* s1 = j < n_next ? y + (size_t)(offs[t + 1] + j) * ldy + H : h0 + ((size_t)B + j) * H;
    for (int i = threadIdx.x; i < 2 * q; i += 256) {
      const float4 v = *reinterpret_cast<const float4*>((i < q ? s0 : s1 - H) + 4 * i);
      if (BF) {
        bf16x4 pk;
        pk[0] = (__bf16)v.x; pk[1] = (__bf16)v.y; pk[2] = (__bf16)v.z; pk[3] = (__bf16)v.w;
        *reinterpret_cast<bf16x4*>(reinterpret_cast<__bf16*>(out) + (size_t)(o + j) * ldo + 4 * i) = pk;
      } else {
        *reinterpret_cast<float4*>(reinterpret_cast<float*>(out) + (size_t)(o + j) * ldo + 4 * i) = v;
      }
    }
  }
}

int check_rows(const char* fn, const void* src, const void* dst, const int32_t* offs, int T, int B, int C) {
  SK_CHECK_ARG(src && dst && offs && src != dst, "%s: null pointer", fn);
  SK_CHECK_ARG(T > 0 && T <= 65535 * 32 && B > 0 && B <= 65535 * ROWS && C > 0, "%s: bad sizes T=%d B=%d C=%d", fn, T, B, C);
  return SK_OK;
}

}  // namespace

extern "C" int sk_pack_rows(const float* padded, const int32_t* offs, const int32_t* perm, int T, int B, int C,
                            float* packed, int ld_packed, sk_stream_t stream) {
  int rc = check_rows("sk_pack_rows", padded, packed, offs, T, B, C);
  if (rc) return rc;
  SK_CHECK_ARG(ld_packed >= C, "sk_pack_rows: leading dimension %d < %d", ld_packed, C);
  hipLaunchKernelGGL(pack_rows_kernel<true>, dim3((unsigned)T, (unsigned)sk_cdiv(B, ROWS)), dim3(256), 0, (hipStream_t)stream,
                     padded, packed, offs, perm, (const float*)nullptr, B, C, ld_packed);
  SK_CHECK_LAUNCH("sk_pack_rows");
  return SK_OK;
}

extern "C" int sk_unpack_rows(const float* packed, int ld_packed, const int32_t* offs, const int32_t* perm, int T, int B,
                              int C, const float* fill, float* padded, sk_stream_t stream) {
  int rc = check_rows("sk_unpack_rows", packed, padded, offs, T, B, C);
  if (rc) return rc;
  SK_CHECK_ARG(ld_packed >= C, "sk_unpack_rows: leading dimension %d < %d", ld_packed, C);
  hipLaunchKernelGGL(pack_rows_kernel<false>, dim3((unsigned)T, (unsigned)sk_cdiv(B, ROWS)), dim3(256), 0, (hipStream_t)stream,
                     packed, padded, offs, perm, fill, B, C, ld_packed);
  SK_CHECK_LAUNCH("sk_unpack_rows");
  return SK_OK;
}

extern "C" int sk_hprev_rows(const float* y, int ldy, const float* h0, const int32_t* offs, int T, int B, int H, void* out,
                             int ld_out, int out_bf16, sk_stream_t stream) {
  int rc = check_rows("sk_hprev_rows", y, out, offs, T, B, 2 * H);
  if (rc) return rc;
  SK_CHECK_ARG(h0 && H % 4 == 0 && ldy >= 2 * H && ldy % 4 == 0 && ld_out >= 2 * H && ld_out % 4 == 0,
               "sk_hprev_rows: H %% 4, leading dimensions >= 2H and %% 4");
  SK_CHECK_ARG(((uintptr_t)y % 16) == 0 && ((uintptr_t)h0 % 16) == 0 && ((uintptr_t)out % 16) == 0, "sk_hprev_rows: 16-byte alignment");
  const dim3 grid((unsigned)T, (unsigned)sk_cdiv(B, ROWS));
  if (out_bf16)
    hipLaunchKernelGGL(hprev_rows_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, y, ldy, h0, offs, T, B, H, out, ld_out);
  else
    hipLaunchKernelGGL(hprev_rows_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, y, ldy, h0, offs, T, B, H, out, ld_out);
  SK_CHECK_LAUNCH("sk_hprev_rows");
  return SK_OK;
}
