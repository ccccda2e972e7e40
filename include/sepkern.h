/* sepkern.h -- C ABI of libsepkern.so: the MI355X (gfx950) kernels under the uPIT hot path.
 *
 * The reference (mmaciej2/speech-separation) has no FFI: its hot path is library calls made
 * from Python (torch / librosa).  Each entry point below replaces the library call(s) cited
 * next to it and is what the host-side mirror of archs/uPIT.py, steps/extract_feats.py and
 * steps/reconstruct_sources.py binds through ctypes.
 *
 * Conventions
 *   - every function returns 0 on success, a negative SK_E* code otherwise; nothing throws
 *     across the boundary; sk_last_error() returns a thread-local message for the last failure
 *   - all pointers are DEVICE pointers unless the name ends in _host; the caller owns every
 *     buffer (the library allocates nothing); workspace sizes are queried
 *   - all work is enqueued asynchronously on `stream` (a hipStream_t passed as void*)
 *   - tensors are row-major contiguous fp32; sequence tensors are time-major, either zero-padded (T, B, C) or
 *     PACKED as torch's PackedSequence.data holds them (see "packed rows" below): every sequence entry point takes an
 *     offset table `offs` (NULL = padded)
 *   - LSTM weight layout is torch's per-parameter layout (gate rows i,f,g,o), so a
 *     reference state_dict round-trips unchanged
 */
#ifndef SEPKERN_H
#define SEPKERN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SK_VERSION 131

#define SK_OK 0
#define SK_EINVAL (-1)   /* bad argument / unsupported shape */
#define SK_ELAUNCH (-2)  /* HIP launch or runtime error      */
#define SK_ETIMEOUT (-3) /* a persistent kernel's bounded spin gave up (see sk_lstm_status) */

typedef void* sk_stream_t; /* hipStream_t */

int sk_version(void);
const char* sk_last_error(void);
/* Which numerics- or timing-changing build options this library was compiled with: 0 for the product build (plain `make`).
 * The kernel sources carry ablation / timing-only / tuning switches for the measurement scripts under profiles/ (`make variant`,
 * `make gemm_variant` build them into libsepkern_<name>.so next to the product library); a library built with any of them
 * reports it here, and the ctypes loader refuses it unless SEPKERN_ALLOW_DIAGNOSTIC_LIB=1 (sepkern/_lib.py), bench.py unless
 * --diagnostic -- a headline can then not come from a wrong-numerics build by accident. */
#define SK_BUILD_TIMING_ONLY 0x1 /* results WRONG by construction (work skipped or faked to bound a kernel's time)        */
#define SK_BUILD_ARITH 0x2       /* another arithmetic than documented (nine piece products, no sign phases, ...)          */
#define SK_BUILD_TUNING 0x4      /* same results, other tuning constants (LDS stages, ring depths, poll cadence, ...)      */
#define SK_BUILD_STAMPS 0x8      /* per-phase clock stamps in the recurrence kernels                                       */
unsigned sk_build_flags(void);
/* number of CUs / LDS bytes per workgroup of the current device */
int sk_device_info(int* num_cu, int* lds_bytes);

/* ---------------------------------------------------------------- STFT front end
 * Replaces librosa.core.load's PCM scaling + librosa.core.stft(n_fft=512, hop=128) [+ np.abs]
 * (reference steps/extract_feats.py:85-89 train, :104-105 test).
 * Utterance u: samples wav[wav_offs[u] .. +nsamp[u]) (float32, or int16 PCM scaled by 1/32768
 * when pcm16 != 0); T_u = 1 + nsamp[u]/hop frames, reflect-padded by n_fft/2, periodic Hann.
 * Output element (t, f) of utterance u goes to out[out_offs[u] + t*stride_t[u] + f*stride_f[u]]
 * (units: elements; float32 magnitude when want_complex == 0, interleaved complex64 otherwise).
 * (stride_t = F, stride_f = 1) is the (T,F) training layout; (1, T_u) is the reference's
 * on-disk (F, T) layout.  frame_major != 0 promises stride_f[u] == 1 for every utterance (the strides live on
 * the device): frames are then stored straight from registers, coalesced, with no LDS staging.
 * max_frames >= max_u T_u sizes the grid.  n_fft must be 512. */
int sk_stft(const void* wav, int pcm16, const int64_t* wav_offs, const int32_t* nsamp, int nutt,
            int n_fft, int hop, int want_complex, void* out, const int64_t* out_offs,
            const int64_t* stride_t, const int64_t* stride_f, int frame_major, int max_frames, sk_stream_t stream);

/* ---------------------------------------------------------------- mask-apply + iSTFT back end
 * Replaces np.multiply(mix_spec, mask) + librosa.core.istft(hop_length=128) + (*32767).astype(int16)
 * (reference steps/reconstruct_sources.py:39-42).  For utterance u and source s:
 *   spec (f,t) = mix[mix_offs[u] + f*mix_sf[u] + t*mix_st[u]] (complex64)
 *   mask (f,t) = mask[mask_offs[u*S+s] + f*mask_sf[u] + t*mask_st[u]] (float32; NULL mask = all ones)
 * Output 128*(T_u-1) samples at wav_out / pcm_out + out_offs[u*S+s] (either may be NULL).
 * int16 conversion truncates toward zero and WRAPS (no saturation), as the reference does. */
int sk_mask_istft(const void* mix_c64, const int64_t* mix_offs, const int64_t* mix_st, const int64_t* mix_sf,
                  const float* mask, const int64_t* mask_offs, const int64_t* mask_st, const int64_t* mask_sf,
                  const int32_t* nframes, int nutt, int S, int n_fft, int hop,
                  float* wav_out, int16_t* pcm_out, const int64_t* out_offs, int max_frames,
                  sk_stream_t stream);

/* ---------------------------------------------------------------- fp32 GEMM (matrix cores)
 * C[M,N] (ldc) = act( opA(A) * opB(B) + bias[n] + (accumulate ? C : 0) ).
 * transA == 0: A is M x K row-major (lda); transA != 0: A is stored K x M row-major (lda).
 * transB == 0: B is K x N row-major (ldb); transB != 0: B is stored N x K row-major (ldb).
 * act: 0 none, 1 sigmoid.  bias may be NULL.  batch > 1 runs `batch` independent problems with
 * pointer strides sA/sB/sC/sbias (elements).  Replaces the matrix products inside nn.LSTM's input
 * projection, nn.Linear and their backward passes (reference archs/uPIT.py:132,141). */
int sk_gemm_f32(const float* A, const float* B, float* C, const float* bias, int M, int N, int K,
                int lda, int ldb, int ldc, int transA, int transB, int accumulate, int act,
                int batch, int64_t sA, int64_t sB, int64_t sC, int64_t sbias, sk_stream_t stream);
/* Same product with K split into `splitk` slices (for weight gradients: few output tiles, K = T*B rows):
 * slices write dense partial slabs into ws (>= sk_gemm_workspace_bytes); the block that finishes a tile's LAST slice
 * (a ticket counter per tile at the head of ws) adds the slabs in fixed slice order and applies bias / accumulate /
 * act -- deterministic, no floating-point atomics, no second launch (the bf16 kernels use a second kernel for the same
 * sums).  ws must be ZERO-FILLED before its first use; every launch leaves its head (the counters) zeroed again.
 * variant (which kernel; results agree to fp32 summation order -- every variant is an fp32 product with fp32 accumulation):
 *   0  choose (the default everywhere).  Products whose operands allow LDS-DMA staging (every operand row 16-byte aligned, K a
 *      multiple of 16, not the T/T form) run on the BF16 MATRIX PIPE by the three-way split of both fp32 operands: x = hi + mid + lo
 *      EXACTLY, three bf16 pieces made by rounding to nearest (bf16 has 8 significand bits: |x - hi| <= 2^-8 |x|, so |mid| <= 2^-8 |x|
 *      and |lo| <= 2^-16 |x|; typical values are half of these bounds).  Of the nine piece products per element pair -- each exact in
 *      fp32 (8 x 8 bits), of relative sizes 1, 2^-8 (two), 2^-16 (three), 2^-24 (two), 2^-32 -- the six of size >= 2^-16 are added
 *      into fp32 accumulators by v_mfma_f32_32x32x16_bf16; the three smallest, together at most 2^-23 |a||b| in the worst case (one
 *      fp32 ulp of the product; 2^-25 for typical pieces), are not formed: a SINGLE product may therefore be off by about one ulp
 *      where an fp32 FMA is exact (tests: <= 2^-22 relative).  What justifies calling this an fp32 GEMM is measured, not this
 *      bound: on sums the error against fp64 is not above the fp32-MFMA kernels' (tests, operands spanning 2^+-20; full-size
 *      training step in four arithmetics against one oracle step).  160-212 TFLOP/s fp32-equivalent on the training step's large
 *      products against 124-135 (one MI355X, stand-alone; the fp32-MFMA pipe's own peak is 157).  Kernels: 9 for unsplit,
 *      unbatched products of at least 192 tiles of 256 x 128 (the split is done once per element while the tile is staged:
 *      186-212 TFLOP/s; 72 KB of LDS, 200 VGPRs -- a caller that runs a product beside a persistent recurrence passes 2), else 2
 *      (128 x 128 tiles; any splitk / batch).
 *      SIGN PHASES: the bf16 MFMA truncates the alignment of its addends towards minus infinity, so a plain split-product result
 *      carries a DC offset (about -2e-11 of the result per K element on all-positive operands: -1.5e-7 at K = 7168) that anything
 *      integrating the result amplifies (r05: the recurrence below a data gradient).  Both split kernels, in every form, therefore
 *      keep -sum instead of +sum in their accumulators over stretches of K (signs + - - + per period of about 128 K steps, the B
 *      operand negated before it is split): truncation then pulls down and up in turn and the offsets cancel
 *      (tests/test_gpu_signed_error.py: mean signed error at the fp32-MFMA kernels' level).
 *      SEPKERN_GEMM_PLANES=0: never 9.  Other operands (F = 257 columns, K = 514): the fp32-MFMA kernels as under 8.  Operands
 *      beyond bf16's finite range (|x| > 3.39e38) round to inf.  SEPKERN_GEMM_SPLIT=0 makes 0 mean 8.
 *   1  the register-staged fp32-MFMA kernel (v_mfma_f32_32x32x2_f32), any alignment
 *   2  the 128 x 128-tile split kernel wherever the LDS-DMA conditions hold (else as 8)
 *   3 / 4  the 128 x 128 / 256 x 128-tile fp32-MFMA LDS-DMA kernels wherever they apply (diagnostics)
 *   9  256 x 128 tiles, split once per element while staging (unsplit, unbatched products; else as 2)
 *   6  256 x 256 tiles, one PERSISTENT workgroup per CU with a stream-K cut of the last partial round of tiles, fp32 MFMA:
 *      unsplit, unbatched products; splitk = 1 and ws >= sk_gemm_streamk_workspace_bytes(), zero-filled before its first use and
 *      left with zeroed counters by every launch (without ws: 4).  Tiles of the cut are summed piece by piece in a fixed order:
 *      deterministic.
 *   8  choose among the fp32-MFMA kernels only -- the reference's literal arithmetic (fp32 products, fp32 accumulation rounded to
 *      nearest): LDS-DMA where the operands allow, stream-K (6) for the large unsplit N/T and N/N products when a ws is given, the
 *      register-staged kernel otherwise.  bench.py times the whole step on it as secondary.fp32_mfma.
 *   (5: the 256 x 256 tile without the stream-K cut, retired in r04; 7: the split-product form of 6, retired in r06 -- since the
 *   planes kernel it served no launch of any configuration.  Both are SK_EINVAL.) */
size_t sk_gemm_workspace_bytes(int M, int N, int batch, int splitk);
/* Which kernel the calling thread's LAST sk_gemm_f32[_splitk] / sk_gemm_bf16_splitk launch took (profiling: bench.py prices a
 * launch against the peak of the matrix pipe it ran on; tests/test_gpu_census.py generates DESIGN.md's kernel census from it):
 * 1 register-staged fp32 MFMA, 3 / 4 / 6 the 128 x 128 / 256 x 128 / stream-K fp32-MFMA LDS-DMA kernels, 2 / 10 the 128 x 128 /
 * 256 x 128 split-while-staging SPLIT kernels (bf16 pipe, six piece products), 9 bf16 inputs (fp32 operands rounded on the way
 * in); sk_gemm_bf16_nt / _mm: 11 / 12 the 256 x 128 / 256 x 256-tile bf16-operand kernel, 13 its stream-K form; sk_gemm_pl3_tn: 14;
 * 0 before the first launch.
 * A thread-local read-back: with the error string of sk_last_error() the library's only mutable state that is not a caller's
 * buffer (SURVEY 8b's rule has these two exceptions, both thread-local and neither read by any kernel or launch decision). */
int sk_gemm_last_kernel(void);
size_t sk_gemm_streamk_workspace_bytes(void);
/* Zero the ticket counters at the head of a split-K workspace (once, before its first use; a buffer that was allocated
 * zero-filled needs no call). */
int sk_gemm_workspace_init(void* ws, sk_stream_t stream);
int sk_gemm_f32_splitk(const float* A, const float* B, float* C, const float* bias, int M, int N, int K,
                       int lda, int ldb, int ldc, int transA, int transB, int accumulate, int act,
                       int batch, int64_t sA, int64_t sB, int64_t sC, int64_t sbias, int splitk, void* ws,
                       int variant, sk_stream_t stream);
/* Same contract, bf16 matrix-core inputs (BASELINE configs[3]: "bf16"): A and B stay fp32 in memory and are
 * rounded to bf16 (round-to-nearest-even) on the way into the matrix cores; products are exact and are
 * accumulated in fp32; C, bias, slabs are fp32.  Equals an fp32 GEMM of the bf16-rounded operands up to
 * summation order. */
int sk_gemm_bf16_splitk(const float* A, const float* B, float* C, const float* bias, int M, int N, int K,
                        int lda, int ldb, int ldc, int transA, int transB, int accumulate, int act,
                        int batch, int64_t sA, int64_t sB, int64_t sC, int64_t sbias, int splitk, void* ws,
                        sk_stream_t stream);

/* bf16 operands IN MEMORY, both K-contiguous ("NT"): C[M,N] = act(A[M,K] B[N,K]^T + bias (+ C)), A and B bf16
 * (row-major, leading dimensions lda, ldb in elements), C / bias / slabs fp32.  The form every product of the bf16
 * configuration is brought to by writing row-major bf16 copies of the operands (sk_cast_bf16_rows; a factor whose rows
 * are the contraction index enters sk_gemm_bf16_mm K-major, so no transposed copy exists).  Requirements: K % 64 == 0 (pad the copies with zero columns), lda, ldb, sA, sB
 * multiples of 8, A and B 16-byte aligned; rows are read up to K.  M and N are arbitrary.  splitk / ws / batch /
 * accumulate / act as sk_gemm_f32_splitk. */
int sk_gemm_bf16_nt(const void* A, const void* B, float* C, const float* bias, int M, int N, int K, int lda, int ldb,
                    int ldc, int accumulate, int act, int batch, int64_t sA, int64_t sB, int64_t sC, int64_t sbias,
                    int splitk, void* ws, sk_stream_t stream);
/* The same product with either operand optionally K-MAJOR in memory: a_kmajor: A is stored [K][M] (element (m, k) at
 * A[k * lda + m], i.e. the TRANSPOSE of an (K, M) row-major matrix is the factor), likewise b_kmajor for B stored [K][N].
 * This is what the weight-gradient products (both factors are activation / gradient matrices whose ROWS are the
 * contraction index) and the data-gradient products (the weight matrix (N, K') is the K-major B of dout (R, N) x W) need:
 * with it they read the same row-major bf16 copies as the forward products and no transposed copy is ever made (the
 * tiles are DMA'd as they lie and the MFMA fragments are gathered by the transposed LDS read ds_read_b64_tr_b16).
 * K-major operand: ld = elements between consecutive k rows, >= its dimension rounded up to 8 and a multiple of 8; the
 * (ld - dim) padding elements of a row must be readable (finite or not: their products are never stored); K % 64 == 0
 * as always -- rows [K', K) of a zero-padded factor must be zeros in at least one of the two factors.
 * splitk = 1 WITH ws != NULL (>= sk_gemm_streamk_workspace_bytes(), zero-filled before its first use; r03): the product
 * may run as the persistent stream-K form of the 256 x 256-tile kernel (unbatched, N % 256 == 0 or N > 1024, K >= 512):
 * one workgroup per CU, the last partial round of tiles cut along K, pieces added in a fixed order (deterministic) --
 * instead of K slabs of the whole matrix and a reduce launch.  Not for products meant to run beside a recurrence. */
int sk_gemm_bf16_mm(const void* A, const void* B, float* C, const float* bias, int M, int N, int K, int lda, int ldb,
                    int ldc, int a_kmajor, int b_kmajor, int accumulate, int act, int batch, int64_t sA, int64_t sB,
                    int64_t sC, int64_t sbias, int splitk, void* ws, sk_stream_t stream);
/* ---------------------------------------------------------------- operands that arrive split (r06)
 * The weight gradients of the training step are T/N products of activation / gradient matrices whose ROWS are the contraction
 * index.  As fp32 split products (sk_gemm_f32_splitk variant 2) every wave of the GEMM cuts the fragments it reads into their
 * three bf16 pieces -- each element twice per workgroup, and beside a persistent recurrence that VALU work and the longer launch
 * are taken from the recurrence too.  Here the PRODUCERS cut each operand once: sk_split_rows (layer inputs, recurrent inputs),
 * sk_lstm_bwd's plane_bf16 (dgx); sk_gemm_pl3_tn multiplies the planes.
 * sk_split_rows: dst plane p (p = 0 hi, 1 mid, 2 lo; `plane` elements apart), row r, column c = piece p of src[r][c] (pieces by
 * rounding to nearest even, exactly the pieces the split kernels make); zeros for C <= c < ld_dst and for rows R .. R_pad - 1.
 * ld_dst % 8 == 0, plane >= R_pad * ld_dst and % 8 == 0, dst 16-byte aligned. */
int sk_split_rows(const float* src, int R, int C, int ld_src, void* dst, int ld_dst, int R_pad, int64_t plane, sk_stream_t stream);
/* C[M,N] (+)= A^T B on operands that arrive split: A = three bf16 planes [K][lda] (element (k, m) of plane p at
 * Apl[p * planeA + k * lda + m]), B likewise [K][ldb]; fp32 accumulation of the six piece products per element pair with the sign
 * phases of the split kernels -- bit for bit sk_gemm_f32_splitk variant 2 on the fp32 matrices the planes were cut from.  K % 16 ==
 * 0 (zero tail rows in the planes), lda / ldb / planeA / planeB / sA / sB multiples of 8 elements, lda >= M rounded up to 8 (+ the
 * batch offset), planes 16-byte aligned; batch, splitk / ws (zero-filled once) as sk_gemm_f32_splitk.  128 x 128 tiles, 120
 * VGPRs, 48 KB of LDS: meant to run beside a persistent recurrence (and alone). */
int sk_gemm_pl3_tn(const void* Apl, const void* Bpl, float* C, int M, int N, int K, int lda, int ldb, int ldc, int64_t planeA,
                   int64_t planeB, int accumulate, int batch, int64_t sA, int64_t sB, int64_t sC, int splitk, void* ws,
                   sk_stream_t stream);
/* dst[r][c] = bf16(src[r][c]) (round to nearest even) for r < R, c < C; 0 for C <= c < ld_dst and for the rows
 * R .. R_pad-1 (R_pad >= R): a copy that also serves as a K-MAJOR factor of sk_gemm_bf16_mm (its rows are then the
 * contraction index, read in whole K steps of 64).  ld_dst % 8 == 0. */
int sk_cast_bf16_rows(const float* src, int R, int C, int ld_src, void* dst, int ld_dst, int R_pad, sk_stream_t stream);

/* ---------------------------------------------------------------- packed rows
 * The reference feeds the network a torch PackedSequence (archs/uPIT.py:46,132,135): of a length-sorted batch only the
 * R = sum(lens) valid frames exist, time-major: row of (t, j) is offs[t] + j for j < n_t, where n_t = offs[t+1] - offs[t]
 * is the number of utterances longer than t (offs has T+1 int32 entries, offs[0] = 0, offs[T] = R; lens[j] sorted
 * descending, lens[0] = T).  Entry points that take `offs` work on that layout; offs = NULL means zero-padded (T, B, C).
 * sk_pack_rows / sk_unpack_rows convert (perm, may be NULL: sorted position j is the caller's utterance perm[j]);
 * unpack writes zeros at the padded positions, or the row `fill` (C floats; may be NULL) -- the value the reference's
 * network shows at a zero-padded frame, sigmoid(lin(bn(0))), archs/uPIT.py:135-144.  ld_packed >= C is the packed
 * matrix's leading dimension. */
int sk_pack_rows(const float* padded, const int32_t* offs, const int32_t* perm, int T, int B, int C, float* packed,
                 int ld_packed, sk_stream_t stream);
int sk_unpack_rows(const float* packed, int ld_packed, const int32_t* offs, const int32_t* perm, int T, int B, int C,
                   const float* fill, float* padded, sk_stream_t stream);
/* The recurrent INPUT of every packed row of one BLSTM layer: out[r][0:H] = forward-direction output of the same
 * utterance one frame earlier (h0[0][j] at t = 0), out[r][H:2H] = reverse-direction output one frame later (h0[1][j] at the
 * utterance's last frame); y (R, ldy >= 2H) is the layer output, h0 (2, B, H).  With it the recurrent weight gradient is a
 * plain product over the packed rows, dW_hh[d] = dG[:, d]^T out[:, d-half] (in the padded layout the shift was a constant
 * B rows; packed it varies with t).  out_bf16 != 0: out is bf16 (the operand copy of the bf16 configuration). */
int sk_hprev_rows(const float* y, int ldy, const float* h0, const int32_t* offs, int T, int B, int H, void* out,
                  int ld_out, int out_bf16, sk_stream_t stream);

/* ---------------------------------------------------------------- BLSTM recurrence
 * One bidirectional LSTM layer's time recurrence (the part of nn.LSTM, reference
 * archs/uPIT.py:115,132, that cannot be batched over time), with packed-sequence semantics:
 * for row b the state is frozen at t >= lens[b]; the reverse direction starts from (h0,c0) at t = lens[b]-1.
 * Sequence tensors below are written (T,B,..) for the padded layout (offs = NULL; y is then 0 at t >= lens[b]); with
 * `offs` they are the R packed rows ("packed rows" above; lens sorted descending) and positions past a row's end do
 * not exist -- nothing is read or written there.
 *   gx    (T,B,2,4H)  input projections x*W_ih^T + b_ih + b_hh for both directions, GATE-INTERLEAVED: within a
 *                     direction's 4H values, element 4u + g is gate g (i,f,g,o) of hidden unit u -- the four gates
 *                     of a cell are one 16-byte access (torch keeps the rows of W_ih gate-major, g H + u:
 *                     sk_gate_rows reorders a weight matrix / bias once so that a plain GEMM produces this layout)
 *   whh   (2,4H,H)    recurrent weights (torch layout, gate rows i,f,g,o)
 *   h0,c0 (2,B,H)     initial state of this layer;  hn,cn (2,B,H) final state (may be NULL)
 *   y     (T,B,2H)    layer output [fwd | bwd]
 *   gates (T,B,2,4H)  post-activation i,f,g,o (gate-interleaved like gx) and cs (T,B,2,H) cell states, saved
 *                     for the backward pass (both NULL for inference)
 *   ws    workspace of sk_lstm_workspace_bytes(); zeroed by the call itself
 * mode (low byte): 0 auto, 1 persistent (one launch, flag-synchronised time loop), 2 one launch per step.
 * mode bits 8..15: minimum number of 16-row batch groups a workgroup carries (0/1 = as few as fit): a larger
 * value shrinks the persistent grid, leaving CUs free for kernels running concurrently on other streams.
 * Speed-only variants of the persistent kernels (never the arithmetic, except where noted): bit 16 bf16 matrix-core
 * inputs (this one IS arithmetic: BASELINE configs[3]); bit 17 retired in r05 (8-unit / 256-thread workgroups, two per CU: slower at every shape; setting it is SK_EINVAL); bits 18..19
 * block id -> stream map; bit 20 one polling wave per workgroup; bit 21 flags replicated per XCD; bit 22 one flag per
 * 128-byte line; bits 23..27 hold-back of a step's first poll in units of 0.1 us (0 = the library's choice, 31 = none);
 * bit 28 (fp32 forward, H <= 896): the product h W_hh^T by the EXACT three-way bf16 split of both fp32
 * operands on the bf16 matrix pipe -- x = hi + mid + lo with three bf16 pieces (24 significand bits = 3 x 8); of the nine piece
 * products per element pair (each exact in fp32) the six of relative size >= 2^-16 are added into fp32 accumulators by
 * v_mfma_f32_16x16x32_bf16, the three of size <= 2^-24 -- together at most 2^-23 of |w||h|, see sk_gemm_f32_splitk variant 0 --
 * are not formed: an fp32 product in another summation order (error against fp64 not above the fp32-MFMA kernel's, results
 * within 2.2e-6 of it, no operand perturbed), 96 instead of 256 matrix-pipe cycles per 32 k.  W_hh is split once per launch,
 * h by its producer (three bf16 images, flags hand-off).  Sign phases (r06): the two K halves of a workgroup accumulate with
 * opposite signs (one on -W_hh), so that the bf16 MFMA's truncation towards minus infinity pulls one partial sum down and the
 * other up -- the cell state integrates over the sequence what is left of it (tests/test_gpu_signed_error.py: 400 steps
 * against an fp64 recurrence).  The engine ships it for the fp32 forward recurrence (bench.py's config.numerics
 * names it);
 * bit 29 (fp32 forward; not together with bit 28): "the data is the flag" -- every exchanged h word carries the step's epoch
 * in its two low mantissa bits, producers publish without drain / barrier / flag, consumers hold back, pull, check every word
 * and pull again what was not complete; the next step's product runs on the tagged words (<= 3 ulp = 3.6e-7 relative),
 * everything stored (y, gates, cs, states) is exact.  The r03 default; still the engine's choice for H > 896;
 * bit 30 (bf16 forward with bit 16, persistent launches, 608 < H <= 896, B <= 32, a device of 8 XCDs x 32 CUs; ignored otherwise;
 * r06): XCD-LOCAL streams of 8 rows -- every (direction, 8-row batch group) stream is 28 workgroups of 32 hidden units on ONE
 * XCD, h_t published with plain stores and a plain flag (within an XCD the L2 is the coherence point; polls and pulls stay sc1),
 * 14 KB instead of 28 KB pulled per workgroup and step.  Same arithmetic bit for bit.  A workgroup joins the stream of the XCD it
 * runs on (HW_REG_XCC_ID + one counter per XCD): no dependence on the order in which blocks are dealt; an XCD that received fewer
 * than 28 workgroups ends in the bounded-spin status word (sk_lstm_status: SK_ETIMEOUT, the step is skipped). */
size_t sk_lstm_workspace_bytes(int T, int B, int H);
int sk_lstm_fwd(const float* gx, const float* whh, const float* h0, const float* c0, const int32_t* lens,
                const int32_t* offs, float* y, float* gates, float* cs, float* hn, float* cn, void* ws,
                int T, int B, int H, int mode, sk_stream_t stream);
/* Backward of the recurrence.  mode as sk_lstm_fwd (bits 0..7, 8..15, 16, 18..19, 22, 23..27, 30; bit 29: read by timing-only
 * diagnostic builds alone); bit 17 (speed only, r06): EXCLUSIVE -- the instantiation whose LDS footprint (127 KB) leaves no room
 * for a workgroup of sk_gemm_pl3_tn beside it, so that products enqueued on other streams take the CUs the grid leaves free instead
 * of sharing the recurrence's (without the bit and with one batch group per workgroup: 113 KB, such a workgroup fits).
 * dy (T,B,2H) is the gradient of the layer output, dhn / dcn (2,B,H; either may be NULL = 0)
 * the gradient wrt the final state (the RSH arch carries the hidden state from pass to pass, reference archs/RSH.py:172);
 * produces dgx (T,B,2,4H) = gradient of the gate pre-activations (gate-interleaved like gx; padded layout: zero at padded
 * positions), from which the caller forms dW_ih, dW_hh (with sk_hprev_rows), db and dx with the GEMMs, and dh0/dc0
 * (2,B,H; may be NULL).  gates / cs are what sk_lstm_fwd saved; dgx may alias gates (each cell is read, then
 * overwritten, by the same lane).  Optional by-products that spare the caller a pass over dgx each:
 *   dbias    (ceil(B/16), 2, 4H)  partial sums of dG over (t, b), one row per 16-row batch group block of the
 *            grid (unused rows are 0): their column sum is the gradient of b_ih and of b_hh;
 *   dgx_bf16 (the bf16 configuration) dgx a second time as bf16, dgx_bf16[row ld_bf16 + d 4H + 4u + g]: the row-major
 *            operand copy the layer's data- and weight-gradient products read, so that no cast pass over dgx runs between
 *            the recurrence and those products.  ld_bf16 >= 8H, a multiple of 4; only the rows x 8H entries are written:
 *            padding columns / rows a product expects to be zero are the caller's.
 *            plane_bf16 = 0: that one rounded copy.  plane_bf16 > 0 (the fp32 configuration, r06): dgx_bf16 receives THREE planes
 *            plane_bf16 elements apart -- the exact hi / mid / lo bf16 pieces of every dgx value (x = hi + mid + lo, see
 *            sk_gemm_f32_splitk variant 0), split ONCE here by the lane that computed the value: the operand sk_gemm_pl3_tn reads. */
int sk_lstm_bwd(const float* dy, const float* dhn, const float* dcn, const float* whh, const float* gates,
                const float* cs, const float* c0, const int32_t* lens, const int32_t* offs, float* dgx, float* dh0,
                float* dc0, float* dbias, void* dgx_bf16, int ld_bf16, int64_t plane_bf16, void* ws, int T, int B, int H,
                int mode, sk_stream_t stream);
/* Reorder the rows of a (nblk * 4H, C) matrix between torch's gate-major order (row g H + u inside each block of 4H
 * rows) and the gate-interleaved order of gx / gates / dgx (row 4u + g).  back = 0: dst[4u+g] = src[gH+u] (weights,
 * biases -> interleaved); back = 1: dst[gH+u] (+)= src[4u+g] (weight gradients back to the parameter order,
 * optionally accumulating).  Rows are ld_src / ld_dst floats apart (>= C); without `accumulate`, columns C..ld_dst-1
 * of the destination are zeroed, so a copy with a padded leading dimension (257 -> 260: 16-byte aligned rows for
 * the GEMMs) is made in the same pass.  dbias of sk_lstm_bwd is already gate-major. */
int sk_gate_rows(const float* src, float* dst, int nblk, int H, int C, int ld_src, int ld_dst, int back, int accumulate,
                 sk_stream_t stream);
/* Word 0 of the workspace is a STICKY status word: a launch whose bounded spin gave up sets it (no launch clears
 * it; allocate the workspace zeroed).  Its address can be handed to sk_grad_norm as `guard` (as a float: any
 * non-zero bit pattern counts) so that a failed launch never reaches the weights, without a host sync per step.
 * sk_lstm_status: 0, or SK_ETIMEOUT if a launch since the last call timed out (reads the word back to the host,
 * synchronises the stream, clears the word). */
int sk_lstm_status(void* ws, sk_stream_t stream);
/* dst (R_pad, ld_dst) = src (R, C; rows ld_src floats apart) with columns C..ld_dst-1 and rows R..R_pad-1 zero: the copy of
 * the F = 257 input features with rows padded to 260 floats (16-byte aligned rows for both operands of the layer-0 products)
 * and the row count rounded up to whole K steps of the weight-gradient product. */
int sk_pad_rows(const float* src, int64_t R, int C, int ld_src, float* dst, int ld_dst, int64_t R_pad, sk_stream_t stream);
/* ---------------------------------------------------------------- BatchNorm1d over (rows, C)
 * Replaces nn.BatchNorm1d(2H) on (B, 2H, T) (reference archs/uPIT.py:119,135-138): statistics over
 * ALL count = B*T_max positions, zero-padded frames included.
 * stats: mean[c], var[c] (biased, two-pass) over `count` rows of which the first R are stored in x and the other
 * count - R are zero rows that are not (packed rows: count = B*T_max, R = sum(lens); padded: count = R);
 * ws >= sk_bn_workspace_bytes(R,C). */
size_t sk_bn_workspace_bytes(int R, int C);
int sk_bn_stats(const float* x, int R, int C, int64_t count, float* mean, float* var, void* ws, sk_stream_t stream);
/* running = (1-momentum)*running + momentum*batch (var unbiased, count/(count-1)), as torch does.  guard (may be NULL):
 * a device word, non-zero = leave the running statistics alone (the recurrence's sticky status word, sk_lstm_status:
 * after a timed-out launch the batch statistics are garbage and must not reach a checkpoint). */
int sk_bn_update_running(const float* mean, const float* var, float* running_mean, float* running_var,
                         int64_t count, int C, float momentum, const void* guard, sk_stream_t stream);
/* out = (x - mean) / sqrt(var + eps) * gamma + beta */
int sk_bn_apply(const float* x, const float* mean, const float* var, const float* gamma, const float* beta,
                float* out, int R, int C, float eps, sk_stream_t stream);
/* BatchNorm folded into the Linear layer that follows it (lin(bn(x)), reference archs/uPIT.py:138-141; SURVEY 2.3 K4:
 * "normalize fused into the Linear prologue"): with s = gamma / sqrt(var + eps), t = beta - mean * s,
 *   Wf[o][c] = W[o][c] * s[c]  (leading dimension ldf >= C, extra columns zero),  bf[o] = b[o] + sum_c W[o][c] * t[c],
 * so that lin(bn(x)) = x Wf^T + bf and the normalised activations are never written.  s and t (C each) are returned for
 * sk_bn_unfold_grad: the weight gradient dW (+)= (dz^T x) diag(s) + colsum(dz) t^T from G = dz^T x (O x C, ld ldg). */
int sk_bn_fold(const float* W, const float* b, const float* mean, const float* var, const float* gamma, const float* beta,
               float eps, int O, int C, float* Wf, int ldf, float* bf, float* s, float* t, sk_stream_t stream);
int sk_bn_unfold_grad(const float* G, int ldg, const float* dzsum, const float* s, const float* t, float* dW, int O, int C,
                      int accumulate, sk_stream_t stream);
/* training-mode backward: dgamma, dbeta, dx from dout, x and the batch statistics */
int sk_bn_bwd(const float* dout, const float* x, const float* mean, const float* var, const float* gamma,
              float* dx, float* dgamma, float* dbeta, void* ws, int R, int C, float eps, sk_stream_t stream);
/* The same in two halves, for BatchNorm statistics shared by several devices (data-parallel "sync" option): local
 * column sums dbeta = sum dy, dgamma = sum dy*xhat, then -- after the caller has summed them over the devices --
 * dx from the global sums; `count` = rows that mean / var and the sums cover (all devices). */
int sk_bn_bwd_sums(const float* dout, const float* x, const float* mean, const float* var, float* dgamma,
                   float* dbeta, void* ws, int R, int C, float eps, sk_stream_t stream);
int sk_bn_bwd_apply(const float* dout, const float* x, const float* mean, const float* var, const float* gamma,
                    const float* dgamma, const float* dbeta, float* dx, int R, int C, double count, float eps,
                    sk_stream_t stream);

/* out[c] (+)= sum_r x[r*ld + c]   (bias gradients); ws >= sk_bn_workspace_bytes(R,C) */
int sk_colsum(const float* x, int R, int C, int ld, float* out, int accumulate, void* ws, sk_stream_t stream);
/* dz = dmask * m * (1 - m)   (sigmoid backward, reference archs/uPIT.py:144) */
int sk_sigmoid_bwd(const float* dmask, const float* m, float* dz, int64_t n, sk_stream_t stream);

/* ---------------------------------------------------------------- PIT-MSE loss
 * Replaces the loss body of compute_loss (reference archs/uPIT.py:181-197,206).
 *   mask (T,B,S*F), mix (T,B,F), src_host[s] -> (T,B,F) device pointers (host array of S), lens (B);
 *   with offs != NULL all of them are packed rows (R, .) -- PackedSequence.data as the collator built it
 *   pair_sse (B,S,S): pair[b][s][r] = sum_{t,f} (mask[t,b,s,f]*mix[t,b,f] - src_r[t,b,f])^2
 *   perm_loss (S!,B) in itertools.permutations order; best_perm (B) = argmin (first minimum)
 *   out[0] = loss/norm, out[1] = norm = sum(lens)*F, out[2] = sum_b min loss / S
 * norm_dev (device scalar, may be NULL) replaces norm when given: data-parallel training divides by the
 * GLOBAL norm, all-reduced on the device without a host round trip. */
size_t sk_pit_workspace_bytes(int T, int B, int S);
int sk_pit_mse_fwd(const float* mask, const float* mix, const float* const* src_host, const int32_t* lens,
                   const int32_t* offs, int T, int B, int F, int S, const float* norm_dev, float* pair_sse,
                   float* perm_loss, int32_t* best_perm, float* out, void* ws, sk_stream_t stream);
/* dmask = gscale[0] * 2 * (mask*mix - src_{best_perm[b][s]}) * mix / (S * norm), norm = out[1]; nrows = packed rows R
 * (ignored when offs == NULL) */
int sk_pit_mse_bwd(const float* mask, const float* mix, const float* const* src_host,
                   const int32_t* best_perm, const float* out, const float* gscale, const int32_t* offs, int64_t nrows,
                   int T, int B, int F, int S, float* dmask, sk_stream_t stream);

/* ---------------------------------------------------------------- RSH arch (reference archs/RSH.py)
 * One pass of the greedy source-assignment loss (archs/RSH.py:225-244): mask (T,B,F); x (T,B,ldx) whose
 * first F columns are the mixture; src_host[r] -> (T,B,F), r < S (host array of device pointers).
 *   sse (S,B) raw per-source SSE of mask*mix; used (S,B) int32 in/out: sources already assigned to row b are
 *   excluded; sel (B) = chosen source (first minimum), which is then marked used;
 *   out[0] = sum_b min / S (this pass's loss term), out[1] = sum(lens)*F (its norm term). */
size_t sk_rsh_workspace_bytes(int T, int B, int S);
int sk_rsh_loss_fwd(const float* mask, const float* x, int ldx, const float* const* src_host, const int32_t* lens,
                    int T, int B, int F, int S, int32_t* used, float* sse, int32_t* sel, float* out, void* ws,
                    sk_stream_t stream);
/* dmask = gscale[0] * (2/S) * (mask*mix - src_{sel[b]}) * mix */
int sk_rsh_loss_bwd(const float* mask, const float* x, int ldx, const float* const* src_host, const int32_t* sel,
                    const float* gscale, int T, int B, int F, int S, float* dmask, sk_stream_t stream);
/* Attention update between passes (archs/RSH.py:254-257 / :278-281): x_out = act(x_in - [0 | mask]) on rows of
 * 2F = [mixture | attention]; act = relu when relu != 0 (training), identity otherwise (test). */
int sk_att_update(const float* x_in, const float* mask, float* x_out, int64_t rows, int F, int relu, sk_stream_t stream);
/* dx_in = dx_out * gate, dmask = -(dx_out * gate)[attention half], gate = (x_out > 0) with relu, 1 without */
int sk_att_update_bwd(const float* dx_out, const float* x_out, float* dx_in, float* dmask, int64_t rows, int F,
                      int relu, sk_stream_t stream);

/* ---------------------------------------------------------------- clip_grad_norm_ + Adam
 * Replaces torch.nn.utils.clip_grad_norm_(params, max_norm) + torch.optim.Adam.step()
 * (reference steps/train_qsub.py:121-122) on ONE flat fp32 buffer holding every parameter.
 *   scal[0] = total grad L2 norm, scal[1] = clip coefficient min(1, max_norm/(norm+1e-6)),
 *   scal[2] = 1 when this step is skipped, scal[3] = count of skipped steps (caller zeroes scal once).
 * guard (may be NULL): device float; non-zero means "the gradients of this step are not to be trusted" (a
 * persistent recurrence launch timed out, on this rank or -- summed by the data-parallel all-reduce -- on any
 * rank): sk_clip_adam then leaves parameters and moments untouched.
 * step is the 1-based count of sk_clip_adam calls; the bias corrections use step - scal[3], the number of updates
 * actually applied. ws >= sk_optim_workspace_bytes(n). */
size_t sk_optim_workspace_bytes(int64_t n);
int sk_grad_norm(const float* g, int64_t n, float max_norm, const float* guard, float* scal, void* ws,
                 sk_stream_t stream);
int sk_clip_adam(float* p, const float* g, float* m, float* v, int64_t n, const float* scal,
                 float lr, float beta1, float beta2, float eps, int step, sk_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif
