// lstm.hip -- the time recurrence of one bidirectional LSTM layer on gfx950, forward and backward.
//
// Reference: nn.LSTM(257, H, L, bidirectional=True) on a PackedSequence (archs/uPIT.py:115,132).
// The input projections x*W_ih^T for all T*B rows are a plain GEMM (gemm.hip); what remains is
// the sequential part  gates_t = gx_t + h_{t-1} W_hh^T  ->  cell update,  T steps per direction.
//
// Design (MI355X-first):
//  * One 512-thread workgroup per CU (8 waves, 2 per SIMD) owns 16 hidden units x 16 batch rows
//    of one direction.  Its slice of W_hh (64 gate rows x H, 224 KB at H=896) lives in the VGPRs
//    of its waves for the WHOLE sequence (4*KS/2 registers per lane) and feeds
//    v_mfma_f32_16x16x4_f32 as the A operand: recurrent weights are read once per layer, not per
//    step.  Two waves per SIMD let the hardware overlap one wave's LDS reads with the other's MFMAs.
//  * The MFMA is issued transposed (rows = gate columns, cols = batch) with gate rows
//    interleaved (4*unit + gate), so a lane ends up holding i,f,g,o of ONE (unit,batch) cell in its
//    4 accumulator registers: the cell update is lane-local and c_t never leaves registers.
//  * Per step the workgroups of a (direction, batch group) exchange h_t (forward) or dG_t
//    (backward) through a small buffer laid out exactly as the MFMA B-operand image
//    ([k/4][16 rows][4], 1 KB per 16-k chunk): producers write it with write-through (sc1) stores
//    (fp32 forward: one dword per cell, 256 B per owner wave; bf16 and backward: 8 / 16 B per cell)
//    and raise one sc1 flag per workgroup; consumers poll the flags with sc1 loads and pull the
//    image into LDS with LDS-DMA (global_load_lds ... sc1: no VGPRs, L1 bypassed), so the hand-off
//    is correct for any workgroup->XCD placement (gfx950's per-XCD L2s are not coherent).
//  * Forward: the 16 x H image is shared by all 8 waves (4 gate-row tiles x 2 K halves).
//    Backward: K = 4H is split 8 ways; each wave streams its own eighth through a 2-deep ring of
//    DMA sub-blocks behind counted s_waitcnt vmcnt(N); the 8 partial tiles are summed through LDS.
//  * The hand-off store and flag go out BEFORE the bulk stores (y / gates / dgx) of the step.
//  * Every spin is bounded by a wall-clock timeout that raises a status word (sk_lstm_status).
//  * mode 2 runs the same kernel one step per launch (state through the workspace); it is the
//    fallback when the grid cannot be co-resident (more workgroups than CUs).
#include "sk_common.h"
#include <type_traits>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

// bf16 variant (mode bit 16; BASELINE configs[3]): W_hh and the exchanged operand (h_t forward, dG_t backward)
// are rounded to bf16 on their way into v_mfma_f32_16x16x32_bf16, fp32 accumulate; cell state, gates and all
// stored tensors stay fp32.  The exchange image is then the bf16 B-operand image [k/32][4 k-octets][16 rows][8]
// (still 1 KB per chunk, now 32 k deep), the register slice of W halves, and a step's MFMA phase shrinks 16x.
__device__ __forceinline__ bf16x8 pack8(const float4& a, const float4& b) {
  bf16x8 v;
  v[0] = (__bf16)a.x; v[1] = (__bf16)a.y; v[2] = (__bf16)a.z; v[3] = (__bf16)a.w;
  v[4] = (__bf16)b.x; v[5] = (__bf16)b.y; v[6] = (__bf16)b.z; v[7] = (__bf16)b.w;
  return v;
}
// element offset (bf16 units) of k, row b in the bf16 image
__device__ __forceinline__ int bf_img(int k, int b) { return (k >> 5) * 512 + ((((k >> 3) & 3) * 16 + b) << 3) + (k & 7); }

// S3 variant of the forward kernel (mode bit 28): the fp32 product h W_hh^T on the bf16 matrix pipe by an EXACT three-way
// split of both operands.  An fp32 value x is cut into three bf16 pieces, x = hi + mid + lo exactly (24 significand bits =
// 3 x 8; by rounding to nearest: hi = bf16(x), x - hi is exact and <= 2^-8 |x|; again for mid; what is left IS a bf16, <= 2^-16 |x|).
// w h is the sum of nine piece products, each exact in fp32 (8 x 8 bits), of relative sizes 1 (hi hi), 2^-8 (hi mid, mid hi),
// 2^-16 (hi lo, mid mid, lo hi), 2^-24 (mid lo, lo mid) and 2^-32 (lo lo).  The kernel forms the SIX of size >= 2^-16 and
// adds them into fp32 accumulators with v_mfma_f32_16x16x32_bf16 (16 cycles each, K = 32): 96 matrix-pipe cycles per 32 k
// instead of the 256 of eight v_mfma_f32_16x16x4_f32.  The three it leaves out are together <= 2^-23 |w||h| in the worst case (one
// ulp of that product; typical pieces: a quarter of it) -- at the level of the rounding of the fp32
// accumulation that follows: against an fp64 sum (K = 896, operands of the recurrence's scale) the six-product form has the error of the
// nine-product form (rms 2.21e-7 vs 2.19e-7) and less than the fp32-MFMA kernel's own (2.50e-7), and differs from the nine-
// product form by 7x less than that differs from the fp32-MFMA kernel (tests/test_gpu_kernels.py pins this on the device).
// So this is an fp32 product in another summation order; no operand is perturbed.  W_hh is split once per launch (three
// register pieces: 168 instead of 112 VGPRs), h by its producer before it publishes (three bf16 images: 6 instead of 4 bytes
// per cell).  (r04 shipped all nine products: 144 cycles per 32 k.)
__device__ __forceinline__ void split3(float x, unsigned& hi, unsigned& mid, unsigned& lo) {
  // pieces by round-to-nearest: hi = bf16(x); r = x - hi is exact, |r| <= 2^-8 |x|; mid = bf16(r); lo = r - mid is exact, has
  // at most 8 significant bits (IS a bf16) and |lo| <= 2^-16 |x|
  const __bf16 h = (__bf16)x;
  const float r = x - (float)h;
  const __bf16 m = (__bf16)r;
  const float q = r - (float)m;
  hi = (unsigned)__builtin_bit_cast(unsigned short, h);
  mid = (unsigned)__builtin_bit_cast(unsigned short, m);
  lo = __builtin_bit_cast(unsigned, q) >> 16;
}
// 8 consecutive-k fp32 values -> the three bf16x8 piece vectors.  Packed form (v_cvt_pk_bf16_f32 rounds and packs two values,
// v_pk_add_f32 subtracts two): 9 instructions per pair of values, 36 per call -- the scalar form above costs 60, and the
// backward S3 kernel calls this 14 times per wave and step.  Same pieces, bit for bit (both round to nearest even).
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split3x8(const float4& a, const float4& b, bf16x8& hi, bf16x8& mid, bf16x8& lo) {
  const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
  u32x4 H, M, L;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const f32x2_t x = {v[2 * j], v[2 * j + 1]};
    H[j] = __builtin_bit_cast(unsigned, __builtin_convertvector(x, bf16x2_t));  // element 2j in the low half
    const f32x2_t xh = {__uint_as_float(H[j] << 16), __uint_as_float(H[j] & 0xffff0000u)};
    const f32x2_t r = x - xh;  // exact
    M[j] = __builtin_bit_cast(unsigned, __builtin_convertvector(r, bf16x2_t));
    const f32x2_t rh = {__uint_as_float(M[j] << 16), __uint_as_float(M[j] & 0xffff0000u)};
    const f32x2_t q = r - rh;  // exact, <= 8 significant bits: IS a bf16
    L[j] = __builtin_bit_cast(unsigned, __builtin_convertvector(q, bf16x2_t));
  }
  hi = __builtin_bit_cast(bf16x8, H);
  mid = __builtin_bit_cast(bf16x8, M);
  lo = __builtin_bit_cast(bf16x8, L);
}

constexpr long long SPIN_TICKS = 200000000LL;  // 2 s of the 100 MHz wall clock
// Poll cadence (units of 64 clocks, s_sleep): a poll is a write-through-coherent load that competes with the CU's own
// publish traffic and with everybody's flag lines; diagnostic builds vary it (make -C csrc variant NAME=.. DEFS=..)
#ifndef SK_BF_NSB
#define SK_BF_NSB 4
#endif
#ifndef SK_BF_DEPTH
#define SK_BF_DEPTH 2
#endif
#ifndef SK_F32_NSB
#define SK_F32_NSB 8
#endif
#ifndef SK_F32_DEPTH
#define SK_F32_DEPTH 3
#endif
#ifndef SK_POLL_SLEEP
#define SK_POLL_SLEEP 1
#endif
#ifndef SK_BWD_RING
#define SK_BWD_RING 1  // fp32 backward recurrence: the next chunk's fragment read issued before this chunk's MFMAs (0: compiler's order)
#endif
#ifndef SK_FWD_RING
#define SK_FWD_RING 2  // split forward recurrence: fragment reads in flight ahead of their products (0: the compiler's order)
#endif
#ifndef SK_POLL_DELAY
#define SK_POLL_DELAY 0
#endif
#ifndef SK_FWD_S3_FLIP
#define SK_FWD_S3_FLIP 1  // split forward recurrence: the two K halves accumulate with opposite signs (0: both positive, the r05 form)
#endif
constexpr int NTHREADS = 512;
constexpr int NREP = 8;  // flag replicas (one per XCD label) when replication is on
constexpr int FSPREAD = 32;  // option: one flag per 128-byte line (stride in dwords) instead of 32 flags per line

#define SK_RLX __ATOMIC_RELAXED
#define SK_AGENT __HIP_MEMORY_SCOPE_AGENT
#define SK_GLOBAL_PTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define SK_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

struct WsLayout {
  size_t sticky, ctrl, flags, xbuf, state, total;
  int KS, NBG;
};

inline int pick_ks(int H, bool bf = false) {
  const int need = (H + 15) / 16;
  const int opts[4] = {20, bf ? 40 : 38, 56, 64};  // bf16: 32-k chunks split over 8 waves need KS % 4 == 0
  for (int i = 0; i < 4; ++i)
    if (opts[i] >= need) return opts[i];
  return 0;
}

inline WsLayout ws_layout(int B, int H, bool bf = false) {
  WsLayout w;
  w.KS = pick_ks(H, bf);
  w.NBG = (B + 15) / 16;
  const size_t Hp = 16 * (size_t)w.KS;
  w.sticky = 0;  // word 0: set by any launch whose bounded wait gave up; cleared only by sk_lstm_status
  w.ctrl = 256;  // per-launch status word (+ diagnostic stamps); zeroed with the flags by every call
  w.flags = 512;
  const size_t nflags = (size_t)FSPREAD * 2 * w.NBG * 2 * w.KS;  // up to 2 KS unit groups; NREP replicas or one flag per line
  w.xbuf = w.flags + sk_align(nflags * 4, 256);
  const size_t xbytes = 2 * 2 * (size_t)w.NBG * Hp * 64 * 4;  // backward exchange is the larger one
  w.state = w.xbuf + sk_align(xbytes, 256);
  const size_t sbytes = 2 * 2 * (size_t)w.NBG * 16 * Hp * 4;  // two per-cell state arrays
  w.total = w.state + sk_align(sbytes, 256);
  return w;
}

struct FwdArgs {
  const float* gx;
  const float* whh;
  const float* h0;
  const float* c0;
  const int* lens;
  const int* offs;  // packed rows: row of (t, b) is offs[t] + b for t < lens[b] (lens sorted descending); NULL: padded, t * B + b
  float* y;
  float* gates;
  float* cs;
  float* hn;
  float* cn;
  float* xbuf;
  float* state;
  unsigned* flags;
  unsigned* ctrl;
  unsigned* sticky;
  int T, B, H, NBG, G, s_begin, s_end;
  int map, nby;  // block id -> (unit group, batch-group block, direction) assignment (speed only), grid y extent
  int opt;       // bit 0: one polling wave per workgroup; bit 1: flags replicated per XCD label
  int poll_delay;  // single poller: first poll of a step not before own flag store + poll_delay x 0.1 us (wait_flags)
};

struct BwdArgs {
  const float* dy;
  const float* whh;
  const float* gates;
  const float* cs;
  const float* c0;
  const int* lens;
  const int* offs;  // as FwdArgs::offs
  float* dgx;
  float* dh0;
  float* dc0;
  const float* dhn;  // gradient wrt the final state (2,B,H), may be NULL (= 0)
  const float* dcn;
  float* dbias;      // (grid y, 2, 4H) per-workgroup-row partial sums of dG over (t, b), may be NULL
  float* xbuf;
  float* state;
  unsigned* flags;
  unsigned* ctrl;
  unsigned* sticky;
  int T, B, H, NBG, G, s_begin, s_end, final_mm;
  int map, nby, poll_delay;
  __bf16* dgx_bf;  // optional bf16 twin of dgx (rows (t, b), ld_bf elements apart), written with the fp32 values; may be NULL
  int ld_bf;
  long long pl_bf;  // 0: one bf16 copy (rounded: the bf16 configuration's operand); > 0 (fp32 kernel): THREE planes this many elements
                    // apart -- the exact hi / mid / lo pieces of dgx, the operand sk_gemm_pl3_tn reads ("operands that arrive split")
  int fast;  // mode bit 29: read by the timing-only build -DSK_BWD_BOUND38 alone
  int exclusive;  // mode bit 17: the instantiation with the larger LDS footprint (no GEMM workgroup fits beside it), see launch_bwd
};

// Flag replication (opt bit 1): every producer raises its flag in NREP copies with ONE store instruction (NREP lanes,
// NREP different lines) and a consumer polls the copy selected by its XCD id, so that the pollers of a stream
// are spread over NREP memory channels instead of hammering one or two lines.  Which copy a consumer reads is
// irrelevant for correctness (all copies are written behind the same drain + barrier).
__device__ __forceinline__ int flag_replica(int opt) {
  return (opt & 2) ? (int)(__builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u) : 0;  // HW_REG_XCC_ID[3:0]
}
__device__ __forceinline__ void raise_flag(unsigned* flags0, size_t rep_stride, int opt, int tid, unsigned value) {
  const int n = (opt & 2) ? NREP : 1;
  if (tid < n) __hip_atomic_store(flags0 + (size_t)tid * rep_stride, value, SK_RLX, SK_AGENT);
}

// Wave 0 waits until every flag of its (direction, batch group) has reached `target`.
// `not_before` (100 MHz ticks, 0 = none): the first poll is held back until that time.  All workgroups of a stream publish
// at about the same moment, so polls issued right after one's own flag store only queue on the flag lines in front of
// everybody's flag STORES (and in the CU's own memory queue): measured at H = 896, B = 32, a forward step takes 6.61 us
// with immediate polling and 6.10 us with the first poll ~0.45 us later (bf16: 3.89 -> 3.19 us at ~0.7 us); later
// than that the hold-back adds itself to the step.  The caller passes its own flag-store time plus a fixed allowance.
__device__ __forceinline__ bool wait_flags(const unsigned* flags, int n, unsigned target, unsigned* ctrl, int lane,
                                           long long not_before = 0, int fs = 1) {
  const long long t0 = wall_clock64();
  if (SK_POLL_DELAY > 0) __builtin_amdgcn_s_sleep(SK_POLL_DELAY);
  if (not_before) {
    while (wall_clock64() - not_before < 0) __builtin_amdgcn_s_sleep(1);
  }
  for (unsigned it = 0;; ++it) {
    bool ok = true;
    for (int i = lane; i < n; i += 64) ok = ok && (__hip_atomic_load(flags + (size_t)i * fs, SK_RLX, SK_AGENT) >= target);
    if (__all(ok)) return true;
    if ((it & 63u) == 63u) {
      if (__hip_atomic_load(ctrl, SK_RLX, SK_AGENT) != 0u) return false;  // another workgroup gave up
      if (wall_clock64() - t0 > SPIN_TICKS) {
        if (lane == 0) {
          __hip_atomic_store(ctrl, 1u, SK_RLX, SK_AGENT);
          __hip_atomic_store(ctrl - 64, 1u, SK_RLX, SK_AGENT);  // the sticky word (workspace word 0), see ws_layout
        }
        return false;
      }
    }
    __builtin_amdgcn_s_sleep(SK_POLL_SLEEP);
  }
}

// Every lane of the calling wave waits for ITS flag (idx < 0: none) to reach `target`; returns false on abort.
__device__ __forceinline__ bool wait_flags_sel(const unsigned* flags, int idx, unsigned target, unsigned* ctrl, int lane) {
  const long long t0 = wall_clock64();
  for (unsigned it = 0;; ++it) {
    const bool ok = idx < 0 || __hip_atomic_load(flags + idx, SK_RLX, SK_AGENT) >= target;
    if (__all(ok)) return true;
    if ((it & 63u) == 63u) {
      if (__hip_atomic_load(ctrl, SK_RLX, SK_AGENT) != 0u) return false;  // another workgroup gave up
      if (wall_clock64() - t0 > SPIN_TICKS) {
        if (lane == 0) {
          __hip_atomic_store(ctrl, 1u, SK_RLX, SK_AGENT);
          __hip_atomic_store(ctrl - 64, 1u, SK_RLX, SK_AGENT);  // the sticky word (workspace word 0), see ws_layout
        }
        return false;
      }
    }
    __builtin_amdgcn_s_sleep(SK_POLL_SLEEP);
  }
}

// Cell non-linearities on the hand-off critical path: v_exp_f32 / v_rcp_f32 forms (abs error ~1e-7, inside the
// parity tolerances) instead of the IEEE-exact library expf / division / tanhf.
__device__ __forceinline__ float fast_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float fast_tanh(float x) {
  const float e = __expf(-2.0f * fabsf(x));  // in (0, 1]: no overflow
  const float t = (1.0f - e) * __builtin_amdgcn_rcpf(1.0f + e);
  return copysignf(t, x);
}

// One 1 KB piece (one MFMA chunk's B-operand image) global -> LDS, write-through-coherent (sc1).
__device__ __forceinline__ void dma_piece(const float* gsrc_piece, float* lds_piece, int lane) {
  __builtin_amdgcn_global_load_lds(SK_GLOBAL_PTR(gsrc_piece + lane * 4), SK_LDS_PTR(lds_piece), 16, 0, 16 /* sc1 */);
}
// The same as inline assembly, for the backward kernel's ring of sub-blocks (r03): behind an LDS-DMA BUILTIN hipcc puts
// an `s_waitcnt vmcnt(0)` in front of the next LDS read it cannot prove disjoint from the DMA's destination, which turned
// every third counted wait of the ring (`vmcnt(8)`: this sub-block has landed, two more may fly) into a wait for ALL
// sub-blocks in flight (`s_waitcnt vmcnt(8)` / `s_waitcnt vmcnt(0)` pairs in the ISA).  The kernel orders DMA and reads
// itself (counted vmcnt waits).  Scalar base (a kernel argument) + one VGPR of byte offsets: the builtin's 64-bit VGPR
// addresses also cost the kernel 20 registers (192 -> 172).  Measured, one call: 7.56 -> 7.20 us per step fp32, 4.38 ->
// 4.14 bf16, 37.5 -> 36.9 ms per training step.  The forward kernels wait for all their pieces anyway and keep the
// builtin (the extra scalar moves cost the bf16 forward chain 0.3 us per step).
__device__ __forceinline__ void dma_piece_s(const float* base, unsigned byte_off, unsigned lds_addr, int lane) {
#ifndef SK_DMA_BUILTIN
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 sc1" ::"v"(byte_off + (unsigned)lane * 16u), "s"(base),
               "s"(lds_addr)
               : "memory");
#else
  __builtin_amdgcn_global_load_lds(SK_GLOBAL_PTR(reinterpret_cast<const char*>(base) + byte_off + lane * 16),
                                   (__attribute__((address_space(3))) void*)(uintptr_t)lds_addr, 16, 0, 16 /* sc1 */);
#endif
}

// Diagnostic build only (-DSK_LSTM_STAMPS, libsepkern_stamps.so): wave 0 of workgroup (0,0,0) accumulates the
// 100 MHz wall-clock ticks spent in each phase of the step into words 8.. of the workspace's control block.
// No stamp executes in the shipped library.
#ifdef SK_LSTM_STAMPS
#define SK_STAMP_DECL long long st_t = wall_clock64(), st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define SK_STAMP(i)                                   \
  do {                                                \
    const long long n__ = wall_clock64();             \
    st_acc[i] += n__ - st_t;                          \
    st_t = n__;                                       \
  } while (0)
#define SK_STAMP_FLUSH(ctrl)                                                                   \
  do {                                                                                         \
    if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0)            \
      for (int i__ = 0; i__ < 8; ++i__) ((long long*)(ctrl))[4 + i__] = st_acc[i__];           \
  } while (0)
#else
#define SK_STAMP_DECL
#define SK_STAMP(i)
#define SK_STAMP_FLUSH(ctrl)
#endif

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// ------------------------------------------------------------------------------------ forward
// Waves: w = mt + 4*kh; mt = gate-row tile (4 units x 4 gates), kh = K half.  Cells live in the
// kh == 0 waves: lane (u = lane>>4, b = lane&15) of wave mt <-> unit 4*mt+u, batch row b.
// A workgroup carries G batch groups (blockIdx.y*G .. +G-1) that share its register-resident W slice and
// are processed in turn every step, so that any batch size fits a co-resident grid; while one group's
// h_t is in flight to the other workgroups the next group computes.
constexpr int GMAX = 8;

// Which (unit group, batch-group block, direction) a workgroup takes.  Any bijection is correct (the hand-off is
// placement-independent); the choice only decides which workgroups share an XCD's L2 / a CU.  Observed dispatch:
// linear block id L lands on XCD L % 8, so
//   map 0: direction-major, batch-group-major, unit group fastest (every stream spread over all 8 XCDs);
//   map 1: the streams (direction x batch-group block) are dealt to XCD groups: with S streams and 8 % S == 0 a
//          stream's workgroups all sit on 8/S XCDs, so its h_t exchange crosses fewer L2s;
//   map 2: as 1 for pairs of streams: XCD group g hosts streams 2g and 2g+1, the first half of every XCD's run one
//          stream and the second half the other (two workgroups that share a CU then belong to different streams).
__device__ __forceinline__ void decode_block(int L, int NUG, int nby, int map, int& ug, int& by, int& dir) {
  map &= 3;
  const int S = nby * 2, total = NUG * S;
  int stream, u;
  const int x = L & 7, j = L >> 3, per_xcd = total >> 3;
  if (map == 1 && (total & 7) == 0 && (8 % S) == 0 && (NUG % (8 / S)) == 0) {
    const int g = 8 / S;                 // XCDs per stream
    stream = x / g;
    u = j * g + (x % g);                 // per_xcd * g == NUG
  } else if (map == 2 && (total & 7) == 0 && (S & 1) == 0 && (8 % (S / 2)) == 0 && ((2 * NUG) % (8 / (S / 2))) == 0 &&
             (per_xcd & 1) == 0) {
    const int g = 8 / (S / 2);           // XCDs per pair of streams
    const int half = per_xcd >> 1;       // workgroups of one stream on one XCD
    stream = 2 * (x / g) + (j >= half ? 1 : 0);
    u = (j % half) * g + (x % g);        // half * g == NUG
  } else {
    stream = L / NUG;
    u = L - stream * NUG;
  }
  ug = u;
  by = stream % nby;
  dir = stream / nby;
}

// NW = waves per workgroup: 8 -- 16 hidden units per workgroup (4 gate-row tiles x 2 K halves), one workgroup per CU.  (A 4-wave
// form -- 8 units per workgroup, two workgroups of different streams per CU -- measured slower at every shape and was retired in
// r05; the kernel body stays written in terms of NW.)
// PK: the sequence tensors are PACKED rows (FwdArgs::offs).  A template parameter, not a run-time test of a.offs: the padded
// instantiation is then the r03 kernel instruction for instruction (one process measured the run-time form 0.1-0.25 us per
// step slower on padded batches: the row-base select and the length table in front of the step's gx fetch).
template <int KS, bool BF, int NW, bool S3 = false, bool PK = false>
__global__ __launch_bounds__(NW * 64, 2) void lstm_fwd_kernel(FwdArgs a) {
  static_assert(!(BF && S3) && (!S3 || NW == 8), "S3 is an fp32 variant of the 8-wave kernel");
  constexpr bool B16 = BF || S3;          // the exchanged operand travels as bf16 image(s)
  constexpr int HP = 16 * KS;
  constexpr int MT = NW / 2;              // gate-row tiles (4 units each) per workgroup
  constexpr int NT = NW * 64;
  constexpr int NUG = HP / (4 * MT);      // unit groups = workgroups per (direction, batch-group block)
  constexpr int NCH = B16 ? HP / 32 : KS; // 1 KB chunks of ONE h image (16 k each in fp32, 32 k in bf16)
  constexpr int NQ = NCH / 2;             // chunks per wave (one K half)
  constexpr int NP = S3 ? 3 : 1;          // bf16 piece images (S3: hi, mid, lo)
  constexpr int PIECE = 256 * NCH;        // floats per bf16 piece image (NCH KB)
  constexpr int NPC = NP * NCH;           // 1 KB pieces a workgroup pulls per step
  static_assert(NCH % 2 == 0, "image chunks must split into two K halves");
  __shared__ __attribute__((aligned(16))) float hs[S3 ? 3 * PIECE : 16 * HP];  // B-operand image(s) of h_{s-1}
  __shared__ __attribute__((aligned(16))) float red[MT][64][4];
  __shared__ float st_c[GMAX][64 * MT], st_h[GMAX][64 * MT];  // per-group cell state of the owner lanes
  __shared__ long long st_tpub[GMAX];                         // wave 0: when this workgroup raised the group's flag
  __shared__ int s_abort;
  __shared__ int s_len[GMAX * 16];  // lengths of this workgroup's batch rows (loop-invariant: fetched once, not per step)

  int ug, by, dir;
  decode_block((int)blockIdx.x, NUG, a.nby, a.map, ug, by, dir);
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int mt = w % MT, kh = w / MT;
  const int T = a.T, B = a.B, H = a.H, NBG = a.NBG, G = a.G;

  // ---- W_hh slice -> registers.  MFMA A operand: lane l supplies A[i = l&15][k = l>>4];
  //      row i = 4*unit_local + gate; k of (chunk q, r) = 16 (kh*NQ + q) + 4 (l>>4) + r.
  //      bf16: lane l supplies A[i = l&15][k = 32 chunk + 8 (l>>4) + 0..7] as 8 bf16.
  float wreg[B16 ? 1 : 4 * NQ];
  bf16x8 wb[BF ? NQ : 1];
  bf16x8 w1[S3 ? NQ : 1], w2[S3 ? NQ : 1], w3[S3 ? NQ : 1];  // S3: the hi / mid / lo pieces of the slice
  {
    const int i = lane & 15, kq = lane >> 4;
    const int unit_i = ug * (4 * MT) + 4 * mt + (i >> 2), g_i = i & 3;
    const bool rowok = unit_i < H;
    const float* wrow = a.whh + ((size_t)dir * 4 * H + (size_t)g_i * H + unit_i) * H;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      if (B16) {
        const int k = 32 * (kh * NQ + q) + 8 * kq;
        float4 v0 = make_float4(0.f, 0.f, 0.f, 0.f), v1 = v0;
        if (rowok && k < H) v0 = *reinterpret_cast<const float4*>(wrow + k);
        if (rowok && k + 4 < H) v1 = *reinterpret_cast<const float4*>(wrow + k + 4);
        if (S3) {
          if (SK_FWD_S3_FLIP && kh == 1) {  // sign phases (see the reduce below): this K half holds -W, its partial sum is -S
            v0 = make_float4(-v0.x, -v0.y, -v0.z, -v0.w);
            v1 = make_float4(-v1.x, -v1.y, -v1.z, -v1.w);
          }
          split3x8(v0, v1, w1[q], w2[q], w3[q]);
        } else {
          wb[q] = pack8(v0, v1);
        }
      } else {
        const int k = 16 * (kh * NQ + q) + 4 * kq;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (rowok && k < H) v = *reinterpret_cast<const float4*>(wrow + k);
        wreg[4 * q + 0] = v.x;
        wreg[4 * q + 1] = v.y;
        wreg[4 * q + 2] = v.z;
        wreg[4 * q + 3] = v.w;
      }
    }
  }

  const int u_l = lane >> 4, bl = lane & 15;
  const int unit = ug * (4 * MT) + 4 * mt + u_l;
  const bool owner = kh == 0;  // this wave carries cells
  const int oi = mt * 64 + lane;
  const size_t xblk = S3 ? (size_t)3 * PIECE : (size_t)16 * HP;  // floats per (parity, dir, batch group) exchange block
  const int xoff = ((unit >> 2) * 16 + bl) * 4 + (unit & 3);  // fp32 image position of (k = unit, row = bl)
  const size_t hst = (size_t)2 * NBG * 16 * HP;           // second state array (bf16: exact h between step launches)

  if (owner) {
    for (int gi = 0; gi < G; ++gi) {
      const int bg = by * G + gi, b = bg * 16 + bl;
      float c = 0.f, h = 0.f;
      if (bg < NBG && unit < H && b < B) {
        if (a.s_begin == 0) {
          c = a.c0[((size_t)dir * B + b) * H + unit];
          h = a.h0[((size_t)dir * B + b) * H + unit];
        } else {
          // (step launches: c AND h travel through the workspace's state arrays -- the exchange buffer holds h of the stream's
          // last step in ONE parity slot only, and a stream that has run out of frames no longer writes it)
          c = a.state[((size_t)dir * NBG * 16 + (size_t)bg * 16 + bl) * HP + unit];
          h = a.state[hst + ((size_t)dir * NBG * 16 + (size_t)bg * 16 + bl) * HP + unit];
        }
      }
      st_c[gi][oi] = c;
      st_h[gi][oi] = h;
    }
  }
  if (tid == 0) s_abort = 0;
  // (a per-step load of lens[b] sat in front of the step's gx fetch: the two global latencies in series cost the bf16
  // recurrence, whose matrix phase is too short to hide them, 0.5 us per step)
  if (PK && tid < G * 16) s_len[tid] = ((by * G + (tid >> 4)) < NBG && (by * G + (tid >> 4)) * 16 + (tid & 15) < B)
                                           ? a.lens[(by * G + (tid >> 4)) * 16 + (tid & 15)] : 0;
  __syncthreads();
  SK_STAMP_DECL

  const int fs = (a.opt & 4) ? FSPREAD : 1;  // option: every flag in a 128-byte line of its own (flag stores do not serialise on a line)
  bool aborted = false;
  long long t_self = 0;  // tagged hand-off: when this wave was done with the previous step
  // Packed rows: a stream (direction, batch group) runs only the steps in which its group has a frame at all -- Tg = the
  // group's longest utterance (its first row: the batch is length-sorted) -- as ITS steps 0 .. Tg-1 (t = s forward, Tg-1-s
  // reverse); epochs, parities and flags count those.  Streams are independent, so the short groups of a ragged batch leave
  // the grid early instead of idling through the long group's tail (and leave their CUs to whatever runs beside the grid).
  const int s_hi = PK ? min(a.s_end, s_len[0]) : a.s_end;
  for (int s = a.s_begin; s < s_hi && !aborted; ++s) {
    for (int gi = 0; gi < G; ++gi) {
      const int bg = by * G + gi;
      if (bg >= NBG) break;
      const int Tg = PK ? s_len[gi * 16] : T;
      if (s >= Tg) break;  // (later groups are shorter still)
      const int t = dir ? Tg - 1 - s : s;
      // First row of the time step: offs[t] (packed rows) or t * B (padded; callers pass offs = NULL for batches whose lengths
      // are all equal, where the two layouts coincide).  One scalar load, used at once.  Fetching it a step ahead measured WORSE
      // on one device in one call (profiles/r04_lstm_fwd_row_base_fetch_ab.txt: 6.05 us per step as here, 6.19 as a vector load
      // -- its vmcnt wait at the next step's top also waits for this step's bulk stores --, 6.33 as a scalar load -- every LDS
      // wait of the step becomes a wait for lgkmcnt(0)).
      const int rb = PK ? a.offs[t] : t * B;
      const int b = bg * 16 + bl;
      const bool cellok = owner && unit < H && b < B;
      const int len_b = PK ? s_len[gi * 16 + bl] : ((b < B) ? a.lens[b] : 0);
      const bool live = cellok && t < len_b;                       // this lane's cell takes part in step t
      const size_t row = (size_t)rb + b;                           // its row in gx / y / gates / cs
      float* const xb0 = a.xbuf + ((size_t)(0 * 2 + dir) * NBG + bg) * xblk;
      // (derived, not a second product: two independent block addresses were what tipped this loop's uniform values over the
      // scalar register file -- 111 v_readlane reloads per step of the bf16 instantiation instead of 27, +0.4 us per step)
      float* const xb1 = xb0 + (size_t)2 * NBG * xblk;
      const size_t rep_stride = (size_t)2 * NBG * NUG;
      unsigned* const flags0 = a.flags + (size_t)(dir * NBG + bg) * NUG * fs;  // replica 0 of this stream's flags
      const unsigned* const myflags = flags0 + (size_t)flag_replica(a.opt) * rep_stride;
      SK_STAMP(7);
      // 1. this step's input-projection terms (independent of the recurrence: issue early)
      float4 gxv = make_float4(0.f, 0.f, 0.f, 0.f);  // gate-interleaved layout: i,f,g,o of a cell are one 16-byte access
      // (padded layout: the row exists whatever the length says, so the fetch does not wait for it -- the bf16 recurrence's
      // matrix phase is too short to hide a length read in front of this load)
      if (PK ? live : cellok) gxv = *reinterpret_cast<const float4*>(a.gx + (row * 2 + dir) * 4 * H + 4 * (size_t)unit);
      // 2./3. h_{s-1} image (16 rows x HP) -> LDS.  Each consumer wave waits for the flags of exactly the unit
      // groups whose 1 KB pieces it pulls and starts its LDS-DMAs as soon as those are up -- no workgroup
      // barrier in between.  fp32: all 8 waves pull (56 pieces); bf16 (28 pieces, latency-bound): only the four
      // waves that own no cells, so nothing of the hand-off queues behind the owners' bulk stores (measured:
      // the split costs 2 % in fp32 and gains 1.5 % in bf16).
      const bool tagged = !B16 && (a.opt & 8);
      if (tagged && s > 0) {
        // option (mode bit 29, fp32): THE DATA IS THE FLAG.  Every exchanged word carries the step's epoch in its two low
        // mantissa bits ((s + 1) & 3 for h_s: 3 ulp at most, the product then runs on the tagged values), producers publish
        // without drain / barrier / flag, consumers hold back, pull their pieces, check every word's epoch in LDS and pull
        // again whatever was not there yet.  Two buffers by step parity + a 2-bit epoch: what a buffer held two steps ago
        // carries another epoch, and the buffers are zeroed (epoch 0, expected 1 first) when a sequence starts.
        if (s > a.s_begin && a.poll_delay) {
          const long long nb = t_self + 10LL * a.poll_delay;
          while (wall_clock64() - nb < 0) __builtin_amdgcn_s_sleep(1);
        }
        constexpr int NPW = (NCH + NW - 1) / NW;  // pieces per wave
        const unsigned want = (unsigned)s & 3u;
        const float* src = ((s - 1) & 1) ? xb1 : xb0;
        unsigned pend = 0;
#pragma unroll
        for (int i = 0; i < NPW; ++i)
          if (w + NW * i < NCH) pend |= 1u << i;
        const long long t0 = wall_clock64();
        bool ok = true;
        for (unsigned it = 0; pend; ++it) {
#pragma unroll
          for (int i = 0; i < NPW; ++i)
            if (pend & (1u << i)) dma_piece(src + (w + NW * i) * 256, hs + (w + NW * i) * 256, lane);
          wait_vmcnt<0>();
          unsigned still = 0;
#pragma unroll
          for (int i = 0; i < NPW; ++i)
            if (pend & (1u << i)) {
              const u32x4 v = *reinterpret_cast<const u32x4*>(&hs[(w + NW * i) * 256 + lane * 4]);
              const bool fresh = (((v[0] ^ want) | (v[1] ^ want) | (v[2] ^ want) | (v[3] ^ want)) & 3u) == 0u;
              if (!__all(fresh)) still |= 1u << i;
            }
          pend = still;
          if (pend) {
            if ((it & 15u) == 15u) {
              if (__hip_atomic_load(a.ctrl, SK_RLX, SK_AGENT) != 0u) ok = false;
              if (wall_clock64() - t0 > SPIN_TICKS) {
                if (lane == 0) {
                  __hip_atomic_store(a.ctrl, 1u, SK_RLX, SK_AGENT);
                  __hip_atomic_store(a.ctrl - 64, 1u, SK_RLX, SK_AGENT);  // the sticky word
                }
                ok = false;
              }
              if (!ok) break;
            }
            __builtin_amdgcn_s_sleep(2);
          }
        }
        if (!ok && lane == 0) s_abort = 1;
      } else
      if ((a.opt & 1) && s > a.s_begin && s > 0) {
        // option: ONE polling wave per workgroup (8x fewer pollers on the flag lines, one more barrier)
        if (w == 0 && !wait_flags(myflags, NUG, (unsigned)s, a.ctrl, lane, a.poll_delay ? st_tpub[gi] + 10LL * a.poll_delay : 0LL, fs) &&
            lane == 0)
          s_abort = 1;
        __syncthreads();
      }
      if (s == 0) {
        if (B16) {
          for (int i = tid; i < 16 * (HP / 8); i += NT) {
            const int bb = i & 15, c8 = i >> 4;  // row, k/8
            const int brow = bg * 16 + bb, k = 8 * c8;
            float4 v0 = make_float4(0.f, 0.f, 0.f, 0.f), v1 = v0;
            const float* hp0 = a.h0 + ((size_t)dir * B + brow) * H + k;
            if (brow < B && k < H) v0 = *reinterpret_cast<const float4*>(hp0);
            if (brow < B && k + 4 < H) v1 = *reinterpret_cast<const float4*>(hp0 + 4);
            __bf16* img = reinterpret_cast<__bf16*>(hs) + bf_img(k, bb);
            if (S3) {
              bf16x8 p1, p2, p3;
              split3x8(v0, v1, p1, p2, p3);
              *reinterpret_cast<bf16x8*>(img) = p1;
              *reinterpret_cast<bf16x8*>(img + 2 * PIECE) = p2;
              *reinterpret_cast<bf16x8*>(img + 4 * PIECE) = p3;
            } else {
              *reinterpret_cast<bf16x8*>(img) = pack8(v0, v1);
            }
          }
        } else {
          for (int i = tid; i < 16 * (HP / 4); i += NT) {
            const int bb = i & 15, c = i >> 4;  // row, k/4
            const int brow = bg * 16 + bb;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (brow < B && 4 * c < H) v = *reinterpret_cast<const float4*>(a.h0 + ((size_t)dir * B + brow) * H + 4 * c);
            *reinterpret_cast<float4*>(&hs[(c * 16 + bb) * 4]) = v;
          }
        }
      } else if (tagged) {
        // (pulled and validated above)
      } else if (!BF || !owner) {
        constexpr int PP = NUG / NCH;            // unit groups (flags) per 1 KB piece
        constexpr int NCW = BF ? NW / 2 : NW;    // consumer waves
        static_assert(PP * ((NPC + NCW - 1) / NCW) <= 64, "one lane per polled flag");
        const int wq = BF ? w - NW / 2 : w;
        bool ok = true;
        if (s > a.s_begin) {
          if (a.opt & 1) {
            ok = !s_abort;  // wave 0 polled for the workgroup (above)
          } else {
            const int piece = wq + NCW * (lane / PP);  // (S3: piece p of every image comes from the same unit groups)
            const int idx = (lane < PP * ((NPC + NCW - 1) / NCW) && piece < NPC) ? (piece % NCH) * PP + lane % PP : -1;
            ok = wait_flags_sel(myflags, idx < 0 ? idx : idx * fs, (unsigned)s, a.ctrl, lane);
            if (!ok && lane == 0) s_abort = 1;
          }
        }
        if (ok) {
          const float* src = ((s - 1) & 1) ? xb1 : xb0;
          for (int p = wq; p < NPC; p += NCW) dma_piece(src + p * 256, hs + p * 256, lane);
          wait_vmcnt<0>();
        }
      }
      __syncthreads();
      if (s_abort) {
        aborted = true;
        break;
      }
      SK_STAMP(1);
      // 4. gates^T (64 gate rows x 16 batch) = W_slice (64 x HP) * h^T (HP x 16); this wave: 16 rows, half of K
      f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
      {
        const float* hp = &hs[(kh * NQ) * 256 + lane * 4];
#ifdef SK_TAIL_HALF
        const bool tail = PK && B > 16 && s >= a.lens[16];  // see SK_TAIL_HALF at BwdCfg
#endif
#if SK_FWD_RING > 0
        if constexpr (S3) {
          // The 3 NQ fragment reads of the product (piece lo, mid, hi of chunk 0, 1, ...) through a ring of SK_FWD_RING + 1
          // registers: read i + SK_FWD_RING is ISSUED before the products of read i, so that an LDS latency (~100+ clocks) is
          // covered by the 1-3 MFMAs (16 clocks each) of several reads instead of one.  Left to the compiler every read was
          // followed by s_waitcnt lgkmcnt(0) (ISA: r|M r|MM r|MMM ...): one read in flight, the matrix pipe idle behind each.
          // Same products, same order, same accumulators as the loop below: bit-identical results.  The order is stated
          // (sched_group_barrier), the scheduler would sink the early reads again.  Measured (profiles/r05_fwd_ring.txt, three
          // alternations): forward recurrences 6.17 -> 5.83 ms per training step (5.14 -> 4.86 us per time step), step 28.45 ->
          // 28.05 ms; 2 and 3 reads ahead are equal (250 / 254 VGPRs, 4 would spill).  The same ring in the bf16 kernels (one
          // 16-clock MFMA per read; 6 ahead forward, a sub-block's reads together backward) changed nothing there (12.93 vs
          // 12.90 ms, profiles/r05_bf16_rings.txt): not kept.
          constexpr int D = SK_FWD_RING, NR = 3 * NQ;
          bf16x8 ring[D + 1];
          auto rd = [&](int i) { return *reinterpret_cast<const bf16x8*>(hp + (i / 3) * 256 + (2 - i % 3) * PIECE); };
#pragma unroll
          for (int i = 0; i < D && i < NR; ++i) ring[i] = rd(i);
#pragma unroll
          for (int i = 0; i < NR; ++i) {
            if (i + D < NR) ring[(i + D) % (D + 1)] = rd(i + D);
            const bf16x8 hv = ring[i % (D + 1)];
            const int q = i / 3;
#ifdef SK_UNITS12_BOUND  // TIMING-ONLY (wrong numerics): three quarters of the product -- what a 12-unit workgroup would multiply
            if ((q & 3) == 3) continue;
#endif
            if (i % 3 == 0) {
              acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1[q], hv, acc0, 0, 0, 0);  // hi  * lo
            } else if (i % 3 == 1) {
              acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2[q], hv, acc1, 0, 0, 0);  // mid * mid
              acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1[q], hv, acc0, 0, 0, 0);  // hi  * mid
            } else {
              acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w3[q], hv, acc1, 0, 0, 0);  // lo  * hi
              acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2[q], hv, acc0, 0, 0, 0);  // mid * hi
              acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1[q], hv, acc1, 0, 0, 0);  // hi  * hi
            }
          }
          __builtin_amdgcn_sched_group_barrier(0x100, D < NR ? D : NR, 0);
#pragma unroll
          for (int i = 0; i < NR; ++i) {
            if (i + D < NR) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            if (i % 3 == 0)
              __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            else if (i % 3 == 1)
              __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            else
              __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
          }
        } else
#endif
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
#ifdef SK_TAIL_HALF
          if (tail && (q & 1)) continue;
#endif
#ifdef SK_UNITS12_BOUND
          if ((q & 3) == 3) continue;
#endif
          if (S3) {
            // SIX of the nine piece products per 32 k (see split3 above), the small ones first.  The piece that is needed first is
            // fetched first and the products alternate between the two accumulators (no product waits for the one issued just
            // before it): 5.4 vs 5.7 us per step against fetching hi, mid, lo in that order and chaining a chunk's products on
            // one accumulator (an explicit software pipeline of the fetches on top: no change).
            const bf16x8 h3 = *reinterpret_cast<const bf16x8*>(hp + q * 256 + 2 * PIECE);
            const bf16x8 h2 = *reinterpret_cast<const bf16x8*>(hp + q * 256 + PIECE);
            const bf16x8 h1 = *reinterpret_cast<const bf16x8*>(hp + q * 256);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1[q], h3, acc0, 0, 0, 0);  // hi  * lo    ~2^-16
            acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2[q], h2, acc1, 0, 0, 0);  // mid * mid   ~2^-16
            acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1[q], h2, acc0, 0, 0, 0);  // hi  * mid   ~2^-8
            acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w3[q], h1, acc1, 0, 0, 0);  // lo  * hi    ~2^-16
            acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2[q], h1, acc0, 0, 0, 0);  // mid * hi    ~2^-8
            acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1[q], h1, acc1, 0, 0, 0);  // hi  * hi
          } else if (BF) {
            const bf16x8 hb = *reinterpret_cast<const bf16x8*>(hp + q * 256);
            if (q & 1)
              acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[q], hb, acc1, 0, 0, 0);
            else
              acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[q], hb, acc0, 0, 0, 0);
          } else {
            const float4 hb = *reinterpret_cast<const float4*>(hp + q * 256);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[4 * q + 0], hb.x, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[4 * q + 1], hb.y, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[4 * q + 2], hb.z, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[4 * q + 3], hb.w, acc1, 0, 0, 0);
          }
        }
      }
      f32x4 acc = acc0 + acc1;
      SK_STAMP(2);
      if (!owner) *reinterpret_cast<f32x4*>(&red[mt][lane][0]) = acc;
      __syncthreads();
      SK_STAMP(3);
      float y_out = 0.f, c_out = 0.f;
      bool valid = false;
      if (owner) {
        // S3, sign phases: the bf16 MFMA TRUNCATES the alignment of its addends towards minus infinity, so a partial sum formed on
        // it sits a little below the exact one -- the same way in every cell, every step, and the cell state integrates it over
        // the sequence.  The K half of the kh == 1 waves therefore runs on -W (negated when the slice was split: the pieces of -x
        // are the negated pieces of x) and accumulates -S1; the owner forms S0 - (-S1): one half's truncation pulls the sum down,
        // the other's up, by amounts of the same expectation.  No instruction, no register.  (tests/test_gpu_signed_error.py)
        if (S3 && SK_FWD_S3_FLIP)
          acc -= *reinterpret_cast<const f32x4*>(&red[mt][lane][0]);
        else
          acc += *reinterpret_cast<const f32x4*>(&red[mt][lane][0]);
        // 5. cell update: D row = 4*(lane>>4) + reg -> this lane holds gates i,f,g,o of (unit, b)
        float c_reg = st_c[gi][oi], h_reg = st_h[gi][oi];
        const float gi_ = fast_sigmoid(acc[0] + gxv.x);
        const float gf = fast_sigmoid(acc[1] + gxv.y);
        const float gg = fast_tanh(acc[2] + gxv.z);
        const float go = fast_sigmoid(acc[3] + gxv.w);
        const float c_new = gf * c_reg + gi_ * gg;
        const float h_new = go * fast_tanh(c_new);
        valid = live;
        if (valid) {
          c_reg = c_new;
          h_reg = h_new;
          st_c[gi][oi] = c_reg;
          st_h[gi][oi] = h_reg;
        }
        // 6. publish h_s first (write-through) ...
        if (S3) {
          // as the bf16 kernel: lane bl gathers the 4 units of this wave's tile for batch row bl, splits them and stores 4 bf16
          // = 8 bytes per piece image at k = unit0 .. unit0+3
          const float hv = cellok ? h_reg : 0.f;
          const float hq[4] = {hv, __shfl(hv, bl + 16, 64), __shfl(hv, bl + 32, 64), __shfl(hv, bl + 48, 64)};
          if (lane < 16) {
            unsigned p1[4], p2[4], p3[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) split3(hq[j], p1[j], p2[j], p3[j]);
            float* xdst = (s & 1) ? xb1 : xb0;
            __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(xdst, 0, (int)(xblk * 4), 0x00020000);
            const unsigned off = (unsigned)(bf_img(ug * (4 * MT) + 4 * mt, bl) * 2);
            __builtin_amdgcn_raw_buffer_store_b64((u32x2){p1[0] | (p1[1] << 16), p1[2] | (p1[3] << 16)}, rs, off, 0, 16 /* sc1 */);
            __builtin_amdgcn_raw_buffer_store_b64((u32x2){p2[0] | (p2[1] << 16), p2[2] | (p2[3] << 16)}, rs, off + 4u * PIECE, 0, 16);
            __builtin_amdgcn_raw_buffer_store_b64((u32x2){p3[0] | (p3[1] << 16), p3[2] | (p3[3] << 16)}, rs, off + 8u * PIECE, 0, 16);
          }
        } else if (BF) {
          // the 4 units of this wave's tile for batch row bl sit in lanes bl, 16+bl, 32+bl, 48+bl: lane bl
          // gathers them and stores 4 bf16 = 8 bytes at k = unit0 .. unit0+3 of the image
          const float hv = cellok ? h_reg : 0.f;
          const float h1 = __shfl(hv, bl + 16, 64), h2 = __shfl(hv, bl + 32, 64), h3 = __shfl(hv, bl + 48, 64);
          if (lane < 16) {
            bf16x4 pk;
            pk[0] = (__bf16)hv; pk[1] = (__bf16)h1; pk[2] = (__bf16)h2; pk[3] = (__bf16)h3;
            float* xdst = (s & 1) ? xb1 : xb0;
            __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(xdst, 0, (int)(xblk * 4), 0x00020000);
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, pk), rs,
                                                  (unsigned)(bf_img(ug * (4 * MT) + 4 * mt, bl) * 2), 0, 16 /* sc1 */);
          }
        } else if (tagged) {
          const unsigned hb = (__builtin_bit_cast(unsigned, cellok ? h_reg : 0.f) & ~3u) | ((unsigned)(s + 1) & 3u);
          __hip_atomic_store(reinterpret_cast<unsigned*>(((s & 1) ? xb1 : xb0) + xoff), hb, SK_RLX, SK_AGENT);
        } else {
          __hip_atomic_store(((s & 1) ? xb1 : xb0) + xoff, cellok ? h_reg : 0.f, SK_RLX, SK_AGENT);
        }
        SK_STAMP(4);
        if (!tagged) wait_vmcnt<0>();  // tagged words need no ordering: nobody is told anything
        SK_STAMP(5);
        acc[0] = gi_;
        acc[1] = gf;
        acc[2] = gg;
        acc[3] = go;
        y_out = valid ? h_new : 0.f;
        c_out = c_new;
      }
      if (tagged) {
        t_self = wall_clock64();  // every wave holds its next pull back from here (the owners: their publish)
      } else {
        __syncthreads();
        raise_flag(flags0 + (size_t)ug * fs, rep_stride, a.opt, tid, (unsigned)(s + 1));
        if (tid == 0) st_tpub[gi] = wall_clock64();  // read back by this same wave when it polls for the next step
      }
      // 7. ... then the bulk stores of the step, off the critical path
      if (PK ? valid : cellok) {  // padded layout: y = 0 past a row's end; packed: those rows do not exist
        a.y[row * 2 * H + (size_t)dir * H + unit] = y_out;
        if (a.gates && valid) {
          *reinterpret_cast<f32x4*>(a.gates + (row * 2 + dir) * 4 * H + 4 * (size_t)unit) = acc;
          a.cs[(row * 2 + dir) * H + unit] = c_out;
        }
      }
      SK_STAMP(6);
    }
  }
  SK_STAMP_FLUSH(a.ctrl);

  if (!aborted && owner) {
    for (int gi = 0; gi < G; ++gi) {
      const int bg = by * G + gi, b = bg * 16 + bl;
      if (bg >= NBG || unit >= H || b >= B) continue;
      if (a.s_end == T) {
        if (a.hn) a.hn[((size_t)dir * B + b) * H + unit] = st_h[gi][oi];
        if (a.cn) a.cn[((size_t)dir * B + b) * H + unit] = st_c[gi][oi];
      } else {
        a.state[((size_t)dir * NBG * 16 + (size_t)bg * 16 + bl) * HP + unit] = st_c[gi][oi];
        a.state[hst + ((size_t)dir * NBG * 16 + (size_t)bg * 16 + bl) * HP + unit] = st_h[gi][oi];
      }
    }
  }
}

// ------------------------------------------------------------------------------------ forward, bf16: XCD-local streams of 8 rows
// (r06, mode bit 30)  The bf16 recurrences are hand-off chains: 0.19 us of MFMA in a 3.1 us step.  What the chain
// costs was measured stand-alone (tools/micro/xcd_local_handoff.hip, profiles/r05_xcd_local_handoff.txt): 2.45 us in the shipped
// geometry (streams of 56 workgroups on an XCD pair, write-through stores, 28 KB pulled per workgroup and step), 1.34 us for a
// stream of 28 workgroups on ONE XCD that publishes with PLAIN stores (within an XCD the L2 is the coherence point: an sc1 load
// bypasses the L1 and is served by that L2, dirty line or not -- no write-through, no acknowledgement from memory) and pulls 14 KB.
// r05 built such streams from 16-wave workgroups of 32 units x 16 rows and lost on the two things that were not the hand-off (two
// cell-owning waves per SIMD; 28 KB still pulled).  This form changes ONE thing, the stream: 2 directions x ceil(B / 8) batch
// groups of EIGHT rows (B <= 32: at most 8 streams, one per XCD), each 28 workgroups x 32 hidden units -- in bf16 the W_hh slice
// of 32 units takes the registers 16 units take in fp32 (112 VGPRs).  A workgroup stays 8 waves: wave (tp, kh) multiplies the two
// gate-row tiles 2 tp, 2 tp + 1 (8 units) with one K half; the MFMA's 16 batch columns carry the 8 rows TWICE (lanes n and n + 8
// read the same fragment), so a lane simply keeps tile 2 tp's result for n < 8 and tile 2 tp + 1's for n >= 8: the four kh = 0
// waves own all 256 cells with every lane busy -- one cell-owning wave per SIMD, as in the shipped kernel.  Per step a workgroup
// publishes 512 B (its 32 units x 8 rows = chunk `ug` of the stream's image), polls 28 flags, pulls 14 KB.  The matrix pipe does
// twice the work for the same cells (half-empty columns): 0.37 instead of 0.19 us per step, against more than a microsecond of
// chain.  Same products, same K order, same sums as lstm_fwd_kernel<KS, true, 8>: bit-identical results (tests).
// Membership: a workgroup belongs to the stream of the XCD it RUNS on (HW_REG_XCC_ID) and takes the next free unit group there (a
// counter per XCD in the launch's control block) -- the dispatcher deals blocks round-robin over the XCDs but does not start at
// XCD 0 for every dispatch (r05: a fixed block -> stream map failed inside the training step).  The grid is 8 x 32 blocks; the 4
// surplus workgroups of an XCD, and those of an XCD without a stream, leave at once.  An XCD that received fewer than 28 could
// not complete a step: the bounded spins then raise the status word like any failed launch.
// Persistent launches only (mode 2 runs the ordinary kernel); one batch group per workgroup (no G loop: c and h live in registers).
__device__ __forceinline__ int xl_img(int k, int r) { return (k >> 5) * 256 + ((((k >> 3) & 3) * 8 + r) << 3) + (k & 7); }  // bf16 elements

template <int KS, bool PK>
__global__ __launch_bounds__(512, 2) void lstm_fwd_xl8_kernel(FwdArgs a) {
  constexpr int HP = 16 * KS;
  constexpr int NUG = HP / 32;   // workgroups (unit groups of 32) per stream = 512-byte chunks of a stream's image
  constexpr int NQ = NUG / 2;    // chunks per K half
  constexpr int NPIECE = NUG / 2;  // 1 KB DMA pieces (two chunks each)
  static_assert(NUG % 2 == 0 && NUG <= 32, "one XCD per stream");
  __shared__ __attribute__((aligned(1024))) float hs[NUG * 128];   // B-operand image of h_{s-1}: NUG chunks x [4 k-octets][8 rows][8 bf16]
  __shared__ __attribute__((aligned(16))) float red[4][2][64][4];  // the kh = 1 waves' partial tiles
  __shared__ long long st_tpub;
  __shared__ int s_abort, s_slot;
  __shared__ int s_len[8];

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int T = a.T, B = a.B, H = a.H, NBG = a.NBG;  // NBG: batch groups of EIGHT rows
  const int stream = (int)(__builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u);  // HW_REG_XCC_ID[3:0]
  if (tid == 0) {
    s_slot = (stream < 2 * NBG) ? (int)__hip_atomic_fetch_add(a.ctrl + 32 + stream, 1u, SK_RLX, SK_AGENT) : NUG;
    s_abort = 0;
  }
  __syncthreads();
  const int ug = s_slot;
  if (ug >= NUG) return;  // (uniform)
  const int bg = stream % NBG, dir = stream / NBG;
  const int tp = w & 3, kh = w >> 2;

  // ---- W_hh slice -> registers: tiles 2 tp, 2 tp + 1; lane supplies A[i = lane & 15][k = 32 chunk + 8 (lane >> 4) + 0..7]
  bf16x8 wb[2][NQ];
  {
    const int i = lane & 15, kq = lane >> 4;
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
      const int unit_i = ug * 32 + 4 * (2 * tp + tt) + (i >> 2), g_i = i & 3;
      const bool rowok = unit_i < H;
      const float* wrow = a.whh + ((size_t)dir * 4 * H + (size_t)g_i * H + unit_i) * H;
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const int k = 32 * (kh * NQ + q) + 8 * kq;
        float4 v0 = make_float4(0.f, 0.f, 0.f, 0.f), v1 = v0;
        if (rowok && k < H) v0 = *reinterpret_cast<const float4*>(wrow + k);
        if (rowok && k + 4 < H) v1 = *reinterpret_cast<const float4*>(wrow + k + 4);
        wb[tt][q] = pack8(v0, v1);
      }
    }
  }

  // ---- the cell of an owner lane: n < 8: tile 2 tp (units 8 tp + 0..3), row n; n >= 8: tile 2 tp + 1 (units 8 tp + 4..7), row n - 8
  const int u_l = lane >> 4, n = lane & 15, sel = n >> 3, r = n & 7;
  const bool owner = kh == 0;
  const int unit = ug * 32 + 8 * tp + 4 * sel + u_l;
  const int b = bg * 8 + r;
  const bool cellok = owner && unit < H && b < B;
  float c_reg = 0.f, h_reg = 0.f;
  if (cellok) {
    c_reg = a.c0[((size_t)dir * B + b) * H + unit];
    h_reg = a.h0[((size_t)dir * B + b) * H + unit];
  }
  if (tid < 8) s_len[tid] = (bg * 8 + tid < B) ? a.lens[bg * 8 + tid] : 0;
  __syncthreads();

  constexpr size_t XBLK = (size_t)NUG * 128;  // floats per (parity, stream) exchange block
  float* const xb0 = a.xbuf + (size_t)stream * XBLK;
  float* const xb1 = xb0 + (size_t)8 * XBLK;
  unsigned* const flags0 = a.flags + (size_t)stream * NUG * FSPREAD;  // one flag per 128-byte line
  const int len_b = s_len[r];
  const int Tg = PK ? s_len[0] : T;  // the group's longest utterance (lengths sorted descending)
  const int s_hi = min(a.s_end, Tg);
  bool aborted = false;
  for (int s = a.s_begin; s < s_hi; ++s) {
    const int t = dir ? Tg - 1 - s : s;
    const int rb = PK ? a.offs[t] : t * B;
    const bool live = cellok && t < len_b;
    const size_t row = (size_t)rb + b;
    // 1. this step's input-projection terms (independent of the recurrence: issue early)
    float4 gxv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (PK ? live : cellok) gxv = *reinterpret_cast<const float4*>(a.gx + (row * 2 + dir) * 4 * H + 4 * (size_t)unit);
    // 2. h_{s-1}: the stream's image (8 rows x HP) -> LDS
    if (s == 0) {
      for (int i = tid; i < 8 * (HP / 8); i += 512) {
        const int bb = i & 7, c8 = i >> 3;
        const int brow = bg * 8 + bb, k = 8 * c8;
        float4 v0 = make_float4(0.f, 0.f, 0.f, 0.f), v1 = v0;
        const float* hp0 = a.h0 + ((size_t)dir * B + brow) * H + k;
        if (brow < B && k < H) v0 = *reinterpret_cast<const float4*>(hp0);
        if (brow < B && k + 4 < H) v1 = *reinterpret_cast<const float4*>(hp0 + 4);
        *reinterpret_cast<bf16x8*>(reinterpret_cast<__bf16*>(hs) + xl_img(k, bb)) = pack8(v0, v1);
      }
    } else {
      // one polling wave (28 flags, one per lane), held back behind the own flag store as in the shipped kernel; then the four
      // waves that own no cells pull the 14 pieces (nothing of the hand-off queues behind the owners' bulk stores)
      if (w == 0 && !wait_flags(flags0, NUG, (unsigned)s, a.ctrl, lane, a.poll_delay ? st_tpub + 10LL * a.poll_delay : 0LL, FSPREAD) &&
          lane == 0)
        s_abort = 1;
      __syncthreads();
      if (!s_abort && !owner) {
        const float* src = ((s - 1) & 1) ? xb1 : xb0;
        for (int p = w - 4; p < NPIECE; p += 4) dma_piece(src + p * 256, hs + p * 256, lane);
        wait_vmcnt<0>();
      }
    }
    __syncthreads();
    if (s_abort) {
      aborted = true;
      break;
    }
    // 3. gates^T (128 gate rows x 8 rows, carried twice) = W_slice (128 x HP) * h^T; this wave: 32 gate rows, half of K.  Even / odd
    //    chunks on two accumulators per tile, as lstm_fwd_kernel does (the same sums, bit for bit)
    f32x4 acc[2][2] = {{{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}};
    {
      const float* hp = &hs[(kh * NQ) * 128 + ((lane >> 4) * 8 + r) * 4];
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const bf16x8 hb = *reinterpret_cast<const bf16x8*>(hp + q * 128);
        acc[0][q & 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[0][q], hb, acc[0][q & 1], 0, 0, 0);
        acc[1][q & 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[1][q], hb, acc[1][q & 1], 0, 0, 0);
      }
    }
    const f32x4 t0 = acc[0][0] + acc[0][1], t1 = acc[1][0] + acc[1][1];
    if (!owner) {
      *reinterpret_cast<f32x4*>(&red[tp][0][lane][0]) = t0;
      *reinterpret_cast<f32x4*>(&red[tp][1][lane][0]) = t1;
    }
    __syncthreads();
    float y_out = 0.f, c_out = 0.f;
    f32x4 g4 = {0.f, 0.f, 0.f, 0.f};
    bool valid = false;
    if (owner) {
      f32x4 pre = sel ? t1 : t0;
      pre += *reinterpret_cast<const f32x4*>(&red[tp][sel][lane][0]);
      // 4. cell update: this lane holds gates i, f, g, o of (unit, b)
      const float gi_ = fast_sigmoid(pre[0] + gxv.x);
      const float gf = fast_sigmoid(pre[1] + gxv.y);
      const float gg = fast_tanh(pre[2] + gxv.z);
      const float go = fast_sigmoid(pre[3] + gxv.w);
      const float c_new = gf * c_reg + gi_ * gg;
      const float h_new = go * fast_tanh(c_new);
      valid = live;
      if (valid) {
        c_reg = c_new;
        h_reg = h_new;
      }
      // 5. publish h_s first: the 8 units of this wave for row r are in lanes (u_l = 0..3, n = r) and (u_l, n = r + 8); lane n < 16
      //    gathers the four of its half and stores 4 bf16 = 8 bytes of chunk `ug` -- PLAIN store: the consumers sit on this XCD
      const float hv = cellok ? h_reg : 0.f;
      const float h1 = __shfl(hv, n + 16, 64), h2 = __shfl(hv, n + 32, 64), h3 = __shfl(hv, n + 48, 64);
      if (lane < 16) {
        bf16x4 pk;
        pk[0] = (__bf16)hv; pk[1] = (__bf16)h1; pk[2] = (__bf16)h2; pk[3] = (__bf16)h3;
        float* xdst = (s & 1) ? xb1 : xb0;
        *reinterpret_cast<bf16x4*>(reinterpret_cast<char*>(xdst) + ug * 512 + (tp * 8 + r) * 16 + sel * 8) = pk;
      }
      wait_vmcnt<0>();
      g4[0] = gi_; g4[1] = gf; g4[2] = gg; g4[3] = go;
      y_out = valid ? h_new : 0.f;
      c_out = c_new;
    }
    __syncthreads();
    if (tid == 0) {
      __hip_atomic_store(flags0 + (size_t)ug * FSPREAD, (unsigned)(s + 1), SK_RLX, __HIP_MEMORY_SCOPE_WORKGROUP);  // plain store
      st_tpub = wall_clock64();
    }
    // 6. ... then the bulk stores of the step, off the critical path
    if (PK ? valid : cellok) {
      a.y[row * 2 * H + (size_t)dir * H + unit] = y_out;
      if (a.gates && valid) {
        *reinterpret_cast<f32x4*>(a.gates + (row * 2 + dir) * 4 * H + 4 * (size_t)unit) = g4;
        a.cs[(row * 2 + dir) * H + unit] = c_out;
      }
    }
  }
  if (!aborted && cellok) {
    if (a.hn) a.hn[((size_t)dir * B + b) * H + unit] = h_reg;
    if (a.cn) a.cn[((size_t)dir * B + b) * H + unit] = c_reg;
  }
}

// ------------------------------------------------------------------------------------ backward
// dh_{prev}[b][u] = sum_{k'} dG[b][k'] W_hh[row(k')][u],  k' = 4*unit_k + gate (gate-interleaved).
// Transposed MFMA: D[m = out unit][n = batch] = sum_k' A[m][k'] B[k'][n]; wave w takes the k' chunks
// [w*NQ, (w+1)*NQ) (an eighth of K = 4H), streamed through its own 2-deep ring of DMA sub-blocks;
// the 8 partial tiles are summed through LDS.
template <int KS, bool BF>
struct BwdCfg {
  static constexpr int NQ = BF ? KS / 4 : KS / 2;         // 1 KB chunks per wave (16 k' each in fp32, 32 in bf16)
  static constexpr int NSB = BF ? SK_BF_NSB : SK_F32_NSB;          // sub-blocks per step (ring = 2 sub-blocks per wave; fp32: 8 keeps
                                                          // the workgroup at 91 KB of LDS, bf16 measured faster with 4)
  static constexpr int SB = (NQ + NSB - 1) / NSB;         // chunks per sub-block (last may be short)
  static constexpr int DEPTH = BF ? SK_BF_DEPTH : SK_F32_DEPTH;      // sub-blocks in flight per wave (ring slots)
  static constexpr int cnt(int sb) { return (sb * SB >= NQ) ? 0 : ((sb + 1) * SB <= NQ ? SB : NQ - sb * SB); }
};

template <int KS, bool BF, int SBI>
__device__ __forceinline__ void bwd_issue(const float* xbase, unsigned xoff, unsigned ring_lds, int w, int lane) {
  using C = BwdCfg<KS, BF>;
  constexpr int n = C::cnt(SBI);
  const unsigned dst = ring_lds + (SBI % C::DEPTH) * C::SB * 1024;
#pragma unroll
  for (int j = 0; j < n; ++j) dma_piece_s(xbase, xoff + (unsigned)(w * C::NQ + SBI * C::SB + j) * 1024u, dst + j * 1024, lane);
}

// Register slice of W_hh^T of one wave: fp32 4 floats per chunk, bf16 8 bf16 (4 VGPRs) per chunk.
template <int KS, bool BF>
struct BwdW {
  float f[BF ? 1 : 4 * BwdCfg<KS, BF>::NQ];
  bf16x8 b[BF ? BwdCfg<KS, BF>::NQ : 1];
};

// TIMING-ONLY diagnostic (-DSK_TAIL_HALF, `make variant NAME=tailhalf DEFS=-DSK_TAIL_HALF`; never shipped, WRONG numerics): an
// upper bound for re-partitioning the tail of a ragged batch.  Once the short batch group's streams have left the grid (the long
// group's steps s >= lens[16], B = 32), the long group's workgroups issue only every other MFMA chunk -- what a step's product
// would cost if the 112 idle CUs took half of it for free (same launches, same hand-off, same pulls).
#ifdef SK_TAIL_HALF
#define SK_TAIL_SKIP(j) if (tail && ((j) & 1)) continue;
#else
#define SK_TAIL_SKIP(j)
#endif

template <int KS, bool BF, int SBI>
__device__ __forceinline__ void bwd_consume(const BwdW<KS, BF>& W, const float* ring, int lane, f32x4& acc0, f32x4& acc1,
                                            bool tail) {
  using C = BwdCfg<KS, BF>;
  constexpr int n = C::cnt(SBI);
  const float* src = ring + (SBI % C::DEPTH) * C::SB * 256 + lane * 4;
#ifdef SK_BWD_BOUND38
  // TIMING-ONLY diagnostic (-DSK_BWD_BOUND38, never shipped, WRONG numerics): launches that pass mode bit 29 issue three of every
  // eight MFMAs of the product -- the matrix-pipe time six bf16 piece products would take (96 instead of 256 cycles per 32 k'),
  // with the pieces for free: an upper bound for a split-product form of this kernel (profiles/r06_bwd_split_top_layer.txt)
  if (!BF && tail) {
#pragma unroll
    for (int j = 0; j < n; ++j) {
      const float4 db = *reinterpret_cast<const float4*>(src + j * 256);
      const int q = SBI * C::SB + j;
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(W.f[4 * q + 0], db.x, acc0, 0, 0, 0);
      if (!(j & 1)) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(W.f[4 * q + 1], db.y, acc1, 0, 0, 0);
    }
    return;
  }
#endif
#if SK_BWD_RING > 0
  if constexpr (!BF && n > 0 && KS <= 56) {  // (KS = 64: 4 more registers would pass the 192 this kernel must stay under)
    // fp32: the fragment read of chunk j + 1 is issued before the four MFMAs of chunk j (the compiler's order: read, wait, four
    // MFMAs -- the read's latency is covered by the previous chunk's 128 pipe clocks only as long as the LDS answers within them,
    // and beside a hosted GEMM it does not always).  Same products, same order: bit-identical.  Backward recurrences 12.37 ->
    // 12.17 ms per training step, three alternations (profiles/r05_bwd_ring.txt); 178 VGPRs.
    float4 rg[2];
    rg[0] = *reinterpret_cast<const float4*>(src);
#pragma unroll
    for (int j = 0; j < n; ++j) {
      if (j + 1 < n) rg[(j + 1) & 1] = *reinterpret_cast<const float4*>(src + (j + 1) * 256);
      const float4 db = rg[j & 1];
      const int q = SBI * C::SB + j;
#ifdef SK_UNITS12_BOUND
      if ((q & 3) == 3) continue;
#endif
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(W.f[4 * q + 0], db.x, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(W.f[4 * q + 1], db.y, acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(W.f[4 * q + 2], db.z, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(W.f[4 * q + 3], db.w, acc1, 0, 0, 0);
    }
    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
#pragma unroll
    for (int j = 0; j < n; ++j) {
      if (j + 1 < n) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
    }
    return;
  }
#endif
#pragma unroll
  for (int j = 0; j < n; ++j) {
    SK_TAIL_SKIP(j)
    const int q = SBI * C::SB + j;
    if (BF) {
      const bf16x8 db = *reinterpret_cast<const bf16x8*>(src + j * 256);  // dG[b = lane&15][k' = 32 cc + 8 (lane>>4) + 0..7]
      if (q & 1)
        acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(W.b[q], db, acc1, 0, 0, 0);
      else
        acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(W.b[q], db, acc0, 0, 0, 0);
    } else {
      const float4 db = *reinterpret_cast<const float4*>(src + j * 256);  // dG[b = lane&15][k' = 16 cc + 4 (lane>>4) + 0..3]
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(W.f[4 * q + 0], db.x, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(W.f[4 * q + 1], db.y, acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(W.f[4 * q + 2], db.z, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(W.f[4 * q + 3], db.w, acc1, 0, 0, 0);
    }
  }
}

template <int KS, bool BF>
constexpr int bwd_younger(int I) {  // DMAs of the DEPTH-1 sub-blocks after sub-block I
  int n = 0;
  for (int d = 1; d < BwdCfg<KS, BF>::DEPTH; ++d) n += BwdCfg<KS, BF>::cnt(I + d);
  return n;
}
template <int KS, bool BF, int I>
__device__ __forceinline__ void bwd_prologue(const float* xbase, unsigned xoff, unsigned ring_lds, int w, int lane) {
  if constexpr (I < BwdCfg<KS, BF>::DEPTH && I < BwdCfg<KS, BF>::NSB) {
    bwd_issue<KS, BF, I>(xbase, xoff, ring_lds, w, lane);
    bwd_prologue<KS, BF, I + 1>(xbase, xoff, ring_lds, w, lane);
  }
}

// Sub-block I of the DEPTH-deep ring: wait until it has landed (only the DMAs of the next DEPTH-1 sub-blocks may
// still be in flight), multiply, and refill its slot with sub-block I+DEPTH.
template <int KS, bool BF, int I>
__device__ __forceinline__ void bwd_ring(const BwdW<KS, BF>& wreg, const float* xbase, unsigned xoff, float* ring,
                                         unsigned ring_lds, int w, int lane, f32x4& acc0, f32x4& acc1, bool tail) {
  using C = BwdCfg<KS, BF>;
  if constexpr (I < C::NSB) {
    constexpr int younger = bwd_younger<KS, BF>(I);
    wait_vmcnt<younger>();
    bwd_consume<KS, BF, I>(wreg, ring, lane, acc0, acc1, tail);
    if constexpr (I + C::DEPTH < C::NSB) {
      if constexpr (C::cnt(I + C::DEPTH) > 0) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // all reads of this ring slot returned before it is refilled
        __builtin_amdgcn_sched_barrier(0);
        bwd_issue<KS, BF, I + C::DEPTH>(xbase, xoff, ring_lds, w, lane);
      }
    }
    bwd_ring<KS, BF, I + 1>(wreg, xbase, xoff, ring, ring_lds, w, lane, acc0, acc1, tail);
  }
}

// Position of (out unit m, batch n) in a wave's 16 x 16 partial tile.  Rows 2j and 2j+1 share a 32-word line; which 16-word half
// a row takes is (m ^ (m >> 2)) & 1, so that BOTH access patterns are conflict-free: the writers of a 32-lane pass hold rows r and
// r + 4 (opposite halves), the readers rows 4j and 4j + 1 (one line, opposite halves).  The former [16][17] padding was conflict-
// free for the reads only: SQ_LDS_BANK_CONFLICT was 14 % of this kernel's LDS cycles (profiles/r04_lstm_pmc.txt).
__device__ __forceinline__ int red_slot(int m, int n) { return (m >> 1) * 32 + 16 * ((m ^ (m >> 2)) & 1) + n; }

// Returns, for the cell-owning lanes, sum over all k' of dG * W for their (unit, batch).
template <int KS, bool BF>
__device__ __forceinline__ float bwd_matmul(const BwdW<KS, BF>& wreg, const float* xbase, unsigned xoff, float* ring,
                                            float (*red)[256], int w, int lane, bool tail = false) {
  // xbase: the exchange buffer (kernel argument: scalar), xoff: byte offset of this stream's block of the step (uniform)
  const unsigned ring_lds = __builtin_amdgcn_readfirstlane((int)(unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)ring);
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  // The counted waits below assume the DMAs are the YOUNGEST vector-memory operations of this wave:
  // nothing may be scheduled into this region (the cell loads of the step were issued before it).
  __builtin_amdgcn_sched_barrier(0);
  bwd_prologue<KS, BF, 0>(xbase, xoff, ring_lds, w, lane);
  bwd_ring<KS, BF, 0>(wreg, xbase, xoff, ring, ring_lds, w, lane, acc0, acc1, tail);
  __builtin_amdgcn_sched_barrier(0);
  // D row m = 4*(lane>>4) + reg (out unit), col n = lane&15 (batch)
#pragma unroll
  for (int r = 0; r < 4; ++r) red[w][red_slot(4 * (lane >> 4) + r, lane & 15)] = acc0[r] + acc1[r];
  __syncthreads();
  const int m = 4 * (w & 3) + (lane >> 4), n = lane & 15;  // the cell of this lane (owner waves: w < 4)
  float v = 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k) v += red[k][red_slot(m, n)];
  __syncthreads();
  return v;
}

// Keep this kernel at 192 VGPRs or fewer (bias-gradient sums live in LDS for that reason): two waves per SIMD then leave 128 registers per lane for ONE co-resident GEMM wave (the
// weight-gradient GEMMs of the layer above run next to this recurrence on the same CUs, sepkern/engine.py); at 200+
// no GEMM block fits beside it and the co-scheduling is lost (measured: 39.3 -> 40.9 ms per step).
// (r06: a split-product form of this kernel for the top layer's launch -- W_hh^T as three register pieces, dG split by the wave that
// multiplies it, six bf16 piece products -- was built, parity-pinned and measured: its 36 VALU instructions per 32 k' cost what the
// fp32 MFMAs they replace cost; not kept.  profiles/r06_bwd_split_top_layer.txt, the code: profiles/r06_bwd_split_top_layer.patch)
// GM: the batch groups a workgroup may carry (1 or GMAX; a.G <= GM).  The G = 1 instantiation (every BASELINE shape but the
// reference's batch of 100) keeps 14 KB less per-group state in LDS: 113 instead of 127 KB, which leaves the 48 KB a workgroup of
// gemm_f32_kernel_pl3 needs beside it (r06: with 127 KB that kernel was not co-resident at all).
template <int KS, bool BF, int GM = GMAX>
__global__ __launch_bounds__(NTHREADS, 2) __attribute__((amdgpu_num_vgpr(192))) void lstm_bwd_kernel(BwdArgs a) {
  using C = BwdCfg<KS, BF>;
  constexpr int HP = 16 * KS, NQ = C::NQ;
  __shared__ __attribute__((aligned(16))) float ring_all[8][C::DEPTH * C::SB * 256];
  __shared__ float red[8][256];
  __shared__ float st_carry[GM][256], st_dc[GM][256];  // per-group recurrent state of the owner lanes
  __shared__ __attribute__((aligned(16))) float st_db[256][4];  // owner lanes: running sum of their cells' dG (bias gradient)
  __shared__ long long st_tpub[GM];  // wave 0: when this workgroup raised the group's flag (see wait_flags)
  __shared__ int s_abort;
  __shared__ int s_len[GM * 16];  // lengths of this workgroup's batch rows (loop-invariant: fetched once, not per step)

  int ug, by, dir;
  decode_block((int)blockIdx.x, KS, a.nby, a.map, ug, by, dir);
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform: everything derived from it stays scalar
  const int T = a.T, B = a.B, H = a.H, NBG = a.NBG, G = a.G;
  float* const ring = &ring_all[w][0];

  // ---- W_hh^T slice -> registers: A[m = out unit i][k'] = W_hh[gate r * H + unit_k][ug*16 + i]
  //      bf16: chunk cc holds k' = 32 cc + 8 (l>>4) + j, i.e. unit_k = 8 cc + 2 (l>>4) + (j>>2), gate j&3.
  BwdW<KS, BF> wreg;
  {
    const int i = lane & 15, kq = lane >> 4;
    const int uout = ug * 16 + i;
    const float* wbase = a.whh + (size_t)dir * 4 * H * H + uout;
#pragma unroll
    for (int s = 0; s < NQ; ++s) {
      if (BF) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int unit_k = 8 * (w * NQ + s) + 2 * kq + (j >> 2);
          const bool ok = uout < H && unit_k < H;
          wreg.b[s][j] = (__bf16)(ok ? wbase[((size_t)(j & 3) * H + unit_k) * H] : 0.f);
        }
      } else {
        const int unit_k = 4 * (w * NQ + s) + kq;
        const bool ok = uout < H && unit_k < H;
#pragma unroll
        for (int r = 0; r < 4; ++r) wreg.f[4 * s + r] = ok ? wbase[((size_t)r * H + unit_k) * H] : 0.f;
      }
    }
  }

  const int mt = w & 3, u_l = lane >> 4, bl = lane & 15;
  const int unit = ug * 16 + 4 * mt + u_l;
  const bool owner = w < 4;
  const int oi = (w & 3) * 64 + lane;
  const size_t xblk = (size_t)HP * 64;  // floats per (parity, dir, batch group) exchange block
  // image position of k' = 4*unit + 0..3, row bl: fp32 floats / bf16 elements (chunk unit>>3, octet (unit&7)>>1)
  const unsigned xoff = BF ? (unsigned)bf_img(4 * unit, bl) : (unsigned)(unit * 16 + bl) * 4u;
  const size_t st2 = (size_t)2 * NBG * 16 * HP;

  // carry = gradient wrt h that passes straight through a frozen (padded) step
  if (owner) {
    for (int gi = 0; gi < G; ++gi) {
      const int bg = by * G + gi;
      float cy = 0.f, dc = 0.f;
      if (a.s_begin > 0 && bg < NBG) {
        const size_t soff = ((size_t)dir * NBG * 16 + (size_t)bg * 16 + bl) * HP + unit;
        cy = a.state[soff];
        dc = a.state[st2 + soff];
      } else if (bg < NBG && unit < H && bg * 16 + bl < B) {
        // gradient arriving at the final state: it reaches the last valid step through the frozen ones
        const size_t o = ((size_t)dir * B + bg * 16 + bl) * H + unit;
        if (a.dhn) cy = a.dhn[o];
        if (a.dcn) dc = a.dcn[o];
      }
      st_carry[gi][oi] = cy;
      st_dc[gi][oi] = dc;
    }
  }
  if (tid == 0) s_abort = 0;
  if (tid < G * 16) s_len[tid] = ((by * G + (tid >> 4)) < NBG && (by * G + (tid >> 4)) * 16 + (tid & 15) < B)
                                     ? a.lens[(by * G + (tid >> 4)) * 16 + (tid & 15)] : 0;
  __syncthreads();
  SK_STAMP_DECL

  if (owner) *reinterpret_cast<f32x4*>(&st_db[oi][0]) = f32x4{0.f, 0.f, 0.f, 0.f};  // read and written by the same lane only
  bool aborted = false;
  // first row of a time step: offs[t] (packed) or t * B (padded).  The step processed NEXT is the one the forward pass
  // processed BEFORE this one, so its table entry is also the row base of this step's c_{prev}: one scalar load per step,
  // used by this step's c_{prev} fetch and as the next step's row base (measured neutral in this kernel, whose waits inside
  // the step are counted vmcnt waits on its DMA ring)
  auto row_base = [&](int tt) { return (tt < 0 || tt >= T) ? 0 : (a.offs ? a.offs[tt] : tt * B); };
  // (packed rows: a stream runs its group's Tg steps only, as in lstm_fwd_kernel: t = s reverse direction, Tg-1-s forward)
  const int s_hi = a.offs ? min(a.s_end, s_len[0]) : a.s_end;
  for (int s = a.s_begin; s < s_hi && !aborted; ++s) {
    for (int gi = 0; gi < G; ++gi) {
      const int bg = by * G + gi;
      if (bg >= NBG) break;
      const int Tg = a.offs ? s_len[gi * 16] : T;
      if (s >= Tg) break;
      const int t = dir ? s : Tg - 1 - s;  // reverse of the forward processing order
      const int rb = row_base(t);
      const int rb_next = row_base(dir ? t + 1 : t - 1);
      const int b = bg * 16 + bl;
      const bool cellok = owner && unit < H && b < B;
      const int len_b = s_len[gi * 16 + bl];
      float* const xb0 = a.xbuf + ((size_t)(0 * 2 + dir) * NBG + bg) * xblk;
      float* const xb1 = a.xbuf + ((size_t)(1 * 2 + dir) * NBG + bg) * xblk;
      const int fs = (a.map & 4) ? FSPREAD : 1;  // option (bit 2 of the map field): one flag per 128-byte line
      unsigned* const myflags = a.flags + (size_t)(dir * NBG + bg) * KS * fs;
      SK_STAMP(7);
      const bool valid = cellok && t < len_b;
      const size_t row = (size_t)rb + b;  // row of (t, b) in gates / cs / dy / dgx
      // 1. saved activations of this cell (independent of the recurrence: issue early)
      float gi_ = 0.f, gf = 0.f, gg = 0.f, go = 0.f, ct = 0.f, cprev = 0.f, dyv = 0.f;
      if (valid) {
        const float4 gv = *reinterpret_cast<const float4*>(a.gates + (row * 2 + dir) * 4 * H + 4 * (size_t)unit);
        gi_ = gv.x;
        gf = gv.y;
        gg = gv.z;
        go = gv.w;
        ct = a.cs[(row * 2 + dir) * H + unit];
        const bool has_prev = dir ? (t + 1 < len_b) : (t > 0);
        const size_t rowp = (size_t)rb_next + b;  // row of (t +- 1, b): the step the forward pass took before this one
        cprev = has_prev ? a.cs[(rowp * 2 + dir) * H + unit] : a.c0[((size_t)dir * B + b) * H + unit];
        dyv = a.dy[row * 2 * H + (size_t)dir * H + unit];
      }
      // 2. recurrent gradient from the step processed before this one
      float dh_rec = 0.f;
      if (s > 0) {
        const unsigned xo = (unsigned)(((size_t)((((s - 1) & 1) * 2 + dir) * NBG + bg) * xblk) * 4);
        // one wave polls for the whole workgroup: letting every wave wait for just the unit groups of its own
        // eighth of k' (no barrier before the product) measured SLOWER here (fp32 9.4 -> 10.0 ms, bf16 5.6 -> 5.9)
        if (s > a.s_begin && w == 0) {
          if (!wait_flags(myflags, KS, (unsigned)s, a.ctrl, lane, a.poll_delay ? st_tpub[gi] + 10LL * a.poll_delay : 0LL, fs) && lane == 0)
            s_abort = 1;
        }
        __syncthreads();
        if (s_abort) {
          aborted = true;
          break;
        }
        SK_STAMP(0);
#ifdef SK_TAIL_HALF
        dh_rec = bwd_matmul<KS, BF>(wreg, a.xbuf, xo, ring, red, w, lane, a.offs && B > 16 && s >= a.lens[16]);
#elif defined(SK_BWD_BOUND38)
        dh_rec = bwd_matmul<KS, BF>(wreg, a.xbuf, xo, ring, red, w, lane, a.fast != 0);
#else
        dh_rec = bwd_matmul<KS, BF>(wreg, a.xbuf, xo, ring, red, w, lane);
#endif
        SK_STAMP(2);
      }
      // 3. cell backward (owner waves)
      f32x4 dpre = {0.f, 0.f, 0.f, 0.f};
      if (owner) {
        float carry = st_carry[gi][oi], dc_rec = st_dc[gi][oi];
        dh_rec += carry;
        if (valid) {
          const float dh = dyv + dh_rec;
          const float tc = fast_tanh(ct);
          const float dout = dh * tc;
          const float dc = dc_rec + dh * go * (1.0f - tc * tc);
          dpre[0] = dc * gg * gi_ * (1.0f - gi_);
          dpre[1] = dc * cprev * gf * (1.0f - gf);
          dpre[2] = dc * gi_ * (1.0f - gg * gg);
          dpre[3] = dout * go * (1.0f - go);
          dc_rec = dc * gf;
          carry = 0.f;
          if (a.dbias) *reinterpret_cast<f32x4*>(&st_db[oi][0]) += dpre;
        } else {
          carry = dh_rec;  // frozen step: h_t = h_{t-1}
        }
        st_carry[gi][oi] = carry;
        st_dc[gi][oi] = dc_rec;
        // 4. publish dG_s first, in gate-interleaved order: 16 B per cell, 1 KB contiguous per wave ...
        float* xdst = (s & 1) ? xb1 : xb0;
        __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(xdst, 0, (int)(xblk * 4), 0x00020000);
        SK_STAMP(4);
        if (BF) {
          bf16x4 pk;
          pk[0] = (__bf16)dpre[0]; pk[1] = (__bf16)dpre[1]; pk[2] = (__bf16)dpre[2]; pk[3] = (__bf16)dpre[3];
          __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, pk), rs, (unsigned)(xoff * 2), 0, 16 /* sc1 */);
        } else {
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, dpre), rs, (unsigned)(xoff * 4), 0, 16 /* sc1 */);
        }
        wait_vmcnt<0>();
        SK_STAMP(5);
      }
      __syncthreads();
      if (tid == 0) {
        __hip_atomic_store(myflags + (size_t)ug * fs, (unsigned)(s + 1), SK_RLX, SK_AGENT);
        st_tpub[gi] = wall_clock64();
      }
      // 5. ... then the bulk store of the step (dgx, zero at padded positions)
      if (valid || (cellok && !a.offs)) {  // (packed rows: positions past a row's end do not exist)
        *reinterpret_cast<f32x4*>(a.dgx + (row * 2 + dir) * 4 * H + 4 * (size_t)unit) = dpre;
        if (a.dgx_bf) {  // the operand copy the data- and weight-gradient products of the bf16 configuration read (no cast pass)
          __bf16* const tw = a.dgx_bf + row * a.ld_bf + (size_t)dir * 4 * H + 4 * (size_t)unit;
          if (!BF && a.pl_bf) {
            // fp32: the three bf16 planes of the cell's four values -- the weight-gradient products then find their pieces made
            // (18 VALU instructions per cell here instead of 4.5 per element in every wave that reads it)
            const f32x2_t x0 = {dpre[0], dpre[1]}, x1 = {dpre[2], dpre[3]};
            u32x2 Hh, Mm, Ll;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
              const f32x2_t x = j ? x1 : x0;
              Hh[j] = __builtin_bit_cast(unsigned, __builtin_convertvector(x, bf16x2_t));
              const f32x2_t xh = {__uint_as_float(Hh[j] << 16), __uint_as_float(Hh[j] & 0xffff0000u)};
              const f32x2_t r = x - xh;
              Mm[j] = __builtin_bit_cast(unsigned, __builtin_convertvector(r, bf16x2_t));
              const f32x2_t rh = {__uint_as_float(Mm[j] << 16), __uint_as_float(Mm[j] & 0xffff0000u)};
              const f32x2_t q = r - rh;
              Ll[j] = __builtin_bit_cast(unsigned, __builtin_convertvector(q, bf16x2_t));
            }
            *reinterpret_cast<u32x2*>(tw) = Hh;
            *reinterpret_cast<u32x2*>(tw + a.pl_bf) = Mm;
            *reinterpret_cast<u32x2*>(tw + 2 * a.pl_bf) = Ll;
          } else {
            bf16x4 pk;
            pk[0] = (__bf16)dpre[0]; pk[1] = (__bf16)dpre[1]; pk[2] = (__bf16)dpre[2]; pk[3] = (__bf16)dpre[3];
            *reinterpret_cast<bf16x4*>(tw) = pk;
          }
        }
      }
      SK_STAMP(6);
    }
  }
  SK_STAMP_FLUSH(a.ctrl);

  if (aborted) return;
  if (a.dbias && owner) {
    // bias gradient: sum of this workgroup's dG over its batch rows (16 lanes of a unit, fixed order) -- one
    // partial row per grid-y block; the buffer is zeroed by the host, step launches (mode 2) add to it
    f32x4 dbsum = *reinterpret_cast<const f32x4*>(&st_db[oi][0]);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float v = dbsum[r];
      v += __shfl_xor(v, 1, 64);
      v += __shfl_xor(v, 2, 64);
      v += __shfl_xor(v, 4, 64);
      v += __shfl_xor(v, 8, 64);
      dbsum[r] = v;
    }
    if (bl == 0 && unit < H) {
      float* dbp = a.dbias + ((size_t)by * 2 + dir) * 4 * H + unit;
#pragma unroll
      for (int r = 0; r < 4; ++r) dbp[(size_t)r * H] += dbsum[r];
    }
  }
  for (int gi = 0; gi < G; ++gi) {
    const int bg = by * G + gi;
    if (bg >= NBG) break;
    const int b = bg * 16 + bl;
    const bool cellok = owner && unit < H && b < B;
    if (a.final_mm) {
      // gradient wrt the initial state: one more product with the last published dG (the stream's step Tg - 1)
      const int Tg = a.offs ? s_len[gi * 16] : T;
      const int fs = (a.map & 4) ? FSPREAD : 1;
      const unsigned* const myflags = a.flags + (size_t)(dir * NBG + bg) * KS * fs;
      const unsigned xo = (unsigned)(((size_t)((((Tg - 1) & 1) * 2 + dir) * NBG + bg) * xblk) * 4);
      if (Tg > a.s_begin && w == 0) {
        if (!wait_flags(myflags, KS, (unsigned)Tg, a.ctrl, lane, 0LL, fs) && lane == 0) s_abort = 1;
      }
      __syncthreads();
      if (s_abort) return;
      float dh_rec = bwd_matmul<KS, BF>(wreg, a.xbuf, xo, ring, red, w, lane);
      if (cellok) {
        dh_rec += st_carry[gi][oi];
        if (a.dh0) a.dh0[((size_t)dir * B + b) * H + unit] = dh_rec;
        if (a.dc0) a.dc0[((size_t)dir * B + b) * H + unit] = st_dc[gi][oi];
      }
    } else if (owner) {
      const size_t soff = ((size_t)dir * NBG * 16 + (size_t)bg * 16 + bl) * HP + unit;
      a.state[soff] = st_carry[gi][oi];
      a.state[st2 + soff] = st_dc[gi][oi];
    }
  }
}

// ------------------------------------------------------------------------------------ backward, bf16: XCD-local streams of 8 rows
// The backward twin of lstm_fwd_xl8_kernel (r06, mode bit 30): a stream = (direction, 8-row batch group) = 28 workgroups x 32 OUT
// units on one XCD; per step a workgroup publishes the dG of its 256 cells (32 units x 4 gates x 8 rows bf16 = 2 KB: four 512-byte
// chunks of the stream's image [k' / 32][4 octets][8 rows][8]) with PLAIN stores and a plain flag, polls 28 flags and pulls the
// stream's 56 KB (instead of 112 KB): every wave its eighth of k' = 4H, seven 1 KB LDS-DMAs issued together and consumed behind
// counted vmcnt waits.  A wave multiplies both 16-unit tiles of the workgroup with each fragment (the MFMA's 16 columns carry the 8
// rows twice, as in the forward kernel), the 8 x 2 partial tiles meet in LDS, and the four owner waves (one per SIMD) hold all 256
// cells: lane (u_l, n) owns out unit 16 (n >> 3) + 4 w + u_l, row n & 7.  W_hh^T for 32 out units: 112 VGPRs.  Same products, same
// K order, same sums as lstm_bwd_kernel<KS, true>: bit-identical dgx / dh0 / dc0 / dbias.  One batch group per workgroup (carry and
// dc in registers); persistent launches only.
template <int KS>
__global__ __launch_bounds__(NTHREADS, 2) void lstm_bwd_xl8_kernel(BwdArgs a) {
  constexpr int HP = 16 * KS;
  constexpr int NUG = HP / 32;      // workgroups per stream
  constexpr int NCHK = HP * 4 / 32; // 512-byte chunks of a stream's dG image
  constexpr int NQ = NCHK / 8;      // chunks per wave
  constexpr int NDMA = NQ / 2;      // 1 KB LDS-DMAs per wave and step
  static_assert(NCHK % 16 == 0 && NUG <= 32, "eight waves x pairs of chunks; one XCD per stream");
  __shared__ __attribute__((aligned(1024))) float ring_all[8][NQ * 128];
  __shared__ float red[8][2][256];
  __shared__ __attribute__((aligned(16))) float st_db[256][4];
  __shared__ long long st_tpub;
  __shared__ int s_abort, s_slot;
  __shared__ int s_len[8];

  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int T = a.T, B = a.B, H = a.H, NBG = a.NBG;  // NBG: batch groups of EIGHT rows
  const int stream = (int)(__builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u);  // HW_REG_XCC_ID[3:0]
  if (tid == 0) {
    s_slot = (stream < 2 * NBG) ? (int)__hip_atomic_fetch_add(a.ctrl + 32 + stream, 1u, SK_RLX, SK_AGENT) : NUG;
    s_abort = 0;
  }
  __syncthreads();
  const int ug = s_slot;
  if (ug >= NUG) return;  // (uniform)
  const int bg = stream % NBG, dir = stream / NBG;
  float* const ring = &ring_all[w][0];

  // ---- W_hh^T slice -> registers: tiles tt = 0, 1 (out units ug*32 + 16 tt + i); chunk c of this wave holds
  //      k' = 32 (w NQ + c) + 8 (l >> 4) + j, i.e. unit_k = 8 (w NQ + c) + 2 (l >> 4) + (j >> 2), gate j & 3
  bf16x8 wreg[2][NQ];
  {
    const int i = lane & 15, kq = lane >> 4;
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
      const int uout = ug * 32 + 16 * tt + i;
      // Unconditional loads from clamped addresses + a mask (per-element "load or zero" branches serialise the 224 loads of the
      // slice, each behind its own s_waitcnt).  The element offset of (chunk c, slot j) is min(off0_j + c 8 H, lim_j): eight
      // running 32-bit offsets -- computed afresh per load the scheduler hoists all 224 address computations to the top.  As
      // built the kernel still takes all 256 VGPRs, with 40 dwords spilled in THIS prologue only (the loop needs ~190 and does
      // not spill): two waves per SIMD fill the register file, no GEMM wave fits beside this kernel -- the side-stream weight
      // gradients run on the 32 CUs its grid leaves free (profiles/r06_bf16_xcd_local_streams_of_8_rows.txt)
      const float* wbase = a.whh + (size_t)dir * 4 * H * H + min(uout, H - 1);
      unsigned off[8], lim[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        off[j] = (unsigned)(((j & 3) * H + 8 * (w * NQ) + 2 * kq + (j >> 2)) * H);
        lim[j] = (unsigned)(((j & 3) * H + (H - 1)) * H);
      }
#pragma unroll
      for (int c = 0; c < NQ; ++c) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int unit_k = 8 * (w * NQ + c) + 2 * kq + (j >> 2);
          const float x = wbase[min(off[j], lim[j])];
          v[j] = __uint_as_float(__float_as_uint(x) & ((uout < H && unit_k < H) ? 0xffffffffu : 0u));
          off[j] += 8u * (unsigned)H;
        }
        wreg[tt][c] = pack8(make_float4(v[0], v[1], v[2], v[3]), make_float4(v[4], v[5], v[6], v[7]));
        asm volatile("" ::"v"(wreg[tt][c]) : "memory");
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  __builtin_amdgcn_sched_barrier(0);  // (nothing of what follows is to be computed above the weight load and kept alive across it)

  const int u_l = lane >> 4, n = lane & 15, sel = n >> 3, r = n & 7;
  const bool owner = w < 4;
  const int unit = ug * 32 + 16 * sel + 4 * (w & 3) + u_l;
  const int b = bg * 8 + r;
  const bool cellok = owner && unit < H && b < B;
  const int oi = (w & 3) * 64 + lane;
  // carry = gradient wrt h that passes straight through a frozen (padded) step; starts as the gradient arriving at the final state
  float carry = 0.f, dc_rec = 0.f;
  if (cellok) {
    const size_t o = ((size_t)dir * B + b) * H + unit;
    if (a.dhn) carry = a.dhn[o];
    if (a.dcn) dc_rec = a.dcn[o];
  }
  if (tid < 8) s_len[tid] = (bg * 8 + tid < B) ? a.lens[bg * 8 + tid] : 0;
  if (owner) *reinterpret_cast<f32x4*>(&st_db[oi][0]) = f32x4{0.f, 0.f, 0.f, 0.f};
  __syncthreads();

  constexpr unsigned XBYTES = (unsigned)NCHK * 512u;  // bytes per (parity, stream) exchange block
  unsigned* const myflags = a.flags + (size_t)stream * NUG * FSPREAD;
  const unsigned ring_lds = __builtin_amdgcn_readfirstlane((int)(unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)ring);
  const unsigned xpos = (unsigned)xl_img(4 * unit, r) * 2u;  // byte position of this cell's four gate gradients in the image
  const int len_b = s_len[r];
  const int Tg = a.offs ? s_len[0] : T;
  const int s_hi = min(a.s_end, Tg);
  auto row_base = [&](int tt) { return (tt < 0 || tt >= T) ? 0 : (a.offs ? a.offs[tt] : tt * B); };

  // dh from the dG image of parity `par`: returns, for the owner lanes, sum over all k' of dG * W for their (unit, row)
  auto matmul = [&](int par) -> float {
    const unsigned xo = ((unsigned)par * 8u + (unsigned)stream) * XBYTES + (unsigned)w * (NQ * 512u);
    f32x4 acc[2][2] = {{{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}};
    __builtin_amdgcn_sched_barrier(0);  // the counted waits below assume the DMAs are this wave's youngest vector-memory operations
#pragma unroll
    for (int i = 0; i < NDMA; ++i) dma_piece_s(a.xbuf, xo + i * 1024u, ring_lds + i * 1024u, lane);
    const float* src = ring + ((lane >> 4) * 8 + r) * 4;
#define SK_XL8_PIECE(I)                                                                                         \
    if constexpr (I < NDMA) {                                                                                     \
      wait_vmcnt<NDMA - 1 - I>();                                                                                 \
      const bf16x8 f0 = *reinterpret_cast<const bf16x8*>(src + (2 * I) * 128);                                    \
      const bf16x8 f1 = *reinterpret_cast<const bf16x8*>(src + (2 * I + 1) * 128);                                \
      acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wreg[0][2 * I], f0, acc[0][0], 0, 0, 0);               \
      acc[1][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wreg[1][2 * I], f0, acc[1][0], 0, 0, 0);               \
      acc[0][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wreg[0][2 * I + 1], f1, acc[0][1], 0, 0, 0);           \
      acc[1][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wreg[1][2 * I + 1], f1, acc[1][1], 0, 0, 0);           \
    }
    SK_XL8_PIECE(0) SK_XL8_PIECE(1) SK_XL8_PIECE(2) SK_XL8_PIECE(3) SK_XL8_PIECE(4) SK_XL8_PIECE(5) SK_XL8_PIECE(6) SK_XL8_PIECE(7)
#undef SK_XL8_PIECE
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
      for (int q = 0; q < 4; ++q) red[w][tt][red_slot(4 * (lane >> 4) + q, lane & 15)] = acc[tt][0][q] + acc[tt][1][q];
    __syncthreads();
    const int m = 4 * (w & 3) + (lane >> 4);
    float v = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) v += red[k][sel][red_slot(m, lane & 15)];
    __syncthreads();
    return v;
  };

  bool aborted = false;
  for (int s = a.s_begin; s < s_hi; ++s) {
    const int t = dir ? s : Tg - 1 - s;  // reverse of the forward processing order
    const int rb = row_base(t);
    const int rb_next = row_base(dir ? t + 1 : t - 1);
    const bool valid = cellok && t < len_b;
    const size_t row = (size_t)rb + b;
    // 1. saved activations of this cell (independent of the recurrence: issue early)
    float gi_ = 0.f, gf = 0.f, gg = 0.f, go = 0.f, ct = 0.f, cprev = 0.f, dyv = 0.f;
    if (valid) {
      const float4 gv = *reinterpret_cast<const float4*>(a.gates + (row * 2 + dir) * 4 * H + 4 * (size_t)unit);
      gi_ = gv.x; gf = gv.y; gg = gv.z; go = gv.w;
      ct = a.cs[(row * 2 + dir) * H + unit];
      const bool has_prev = dir ? (t + 1 < len_b) : (t > 0);
      const size_t rowp = (size_t)rb_next + b;
      cprev = has_prev ? a.cs[(rowp * 2 + dir) * H + unit] : a.c0[((size_t)dir * B + b) * H + unit];
      dyv = a.dy[row * 2 * H + (size_t)dir * H + unit];
    }
    // 2. recurrent gradient from the step processed before this one
    float dh_rec = 0.f;
    if (s > 0) {
      if (s > a.s_begin && w == 0) {
        if (!wait_flags(myflags, NUG, (unsigned)s, a.ctrl, lane, a.poll_delay ? st_tpub + 10LL * a.poll_delay : 0LL, FSPREAD) && lane == 0)
          s_abort = 1;
      }
      __syncthreads();
      if (s_abort) {
        aborted = true;
        break;
      }
      dh_rec = matmul((s - 1) & 1);
    }
    // 3. cell backward (owner waves)
    f32x4 dpre = {0.f, 0.f, 0.f, 0.f};
    if (owner) {
      dh_rec += carry;
      if (valid) {
        const float dh = dyv + dh_rec;
        const float tc = fast_tanh(ct);
        const float dout = dh * tc;
        const float dc = dc_rec + dh * go * (1.0f - tc * tc);
        dpre[0] = dc * gg * gi_ * (1.0f - gi_);
        dpre[1] = dc * cprev * gf * (1.0f - gf);
        dpre[2] = dc * gi_ * (1.0f - gg * gg);
        dpre[3] = dout * go * (1.0f - go);
        dc_rec = dc * gf;
        carry = 0.f;
        if (a.dbias) *reinterpret_cast<f32x4*>(&st_db[oi][0]) += dpre;
      } else {
        carry = dh_rec;  // frozen step: h_t = h_{t-1}
      }
      // 4. publish dG_s first: 8 bytes per cell into chunk unit >> 3 of the stream's image -- PLAIN store (the consumers sit on this XCD)
      bf16x4 pk;
      pk[0] = (__bf16)dpre[0]; pk[1] = (__bf16)dpre[1]; pk[2] = (__bf16)dpre[2]; pk[3] = (__bf16)dpre[3];
      char* xdst = reinterpret_cast<char*>(a.xbuf) + ((unsigned)(s & 1) * 8u + (unsigned)stream) * XBYTES;
      if (unit < HP) *reinterpret_cast<bf16x4*>(xdst + xpos) = pk;
      wait_vmcnt<0>();
    }
    __syncthreads();
    if (tid == 0) {
      __hip_atomic_store(myflags + (size_t)ug * FSPREAD, (unsigned)(s + 1), SK_RLX, __HIP_MEMORY_SCOPE_WORKGROUP);  // plain store
      st_tpub = wall_clock64();
    }
    // 5. ... then the bulk stores of the step (dgx, zero at padded positions)
    if (valid || (cellok && !a.offs)) {
      *reinterpret_cast<f32x4*>(a.dgx + (row * 2 + dir) * 4 * H + 4 * (size_t)unit) = dpre;
      if (a.dgx_bf) {
        bf16x4 pk;
        pk[0] = (__bf16)dpre[0]; pk[1] = (__bf16)dpre[1]; pk[2] = (__bf16)dpre[2]; pk[3] = (__bf16)dpre[3];
        *reinterpret_cast<bf16x4*>(a.dgx_bf + row * a.ld_bf + (size_t)dir * 4 * H + 4 * (size_t)unit) = pk;
      }
    }
  }
  if (aborted) return;
  if (a.dbias && owner) {
    // bias gradient: this workgroup's dG summed over its 8 rows; the row of (batch-group-of-16 block, direction) receives the sums
    // of its TWO 8-row groups by atomic adds -- two addends: the result does not depend on their order, and equals the 16-row
    // kernel's (whose shuffle tree adds the two halves last)
    f32x4 dbsum = *reinterpret_cast<const f32x4*>(&st_db[oi][0]);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float v = dbsum[q];
      v += __shfl_xor(v, 1, 64);
      v += __shfl_xor(v, 2, 64);
      v += __shfl_xor(v, 4, 64);
      dbsum[q] = v;
    }
    if (r == 0 && unit < H) {
      float* dbp = a.dbias + ((size_t)(bg >> 1) * 2 + dir) * 4 * H + unit;
#pragma unroll
      for (int q = 0; q < 4; ++q) atomicAdd(dbp + (size_t)q * H, dbsum[q]);
    }
  }
  if (a.final_mm) {
    // gradient wrt the initial state: one more product with the last published dG (the stream's step Tg - 1)
    if (Tg > a.s_begin && w == 0) {
      if (!wait_flags(myflags, NUG, (unsigned)Tg, a.ctrl, lane, 0LL, FSPREAD) && lane == 0) s_abort = 1;
    }
    __syncthreads();
    if (s_abort) return;
    float dh_rec = matmul((Tg - 1) & 1);
    if (cellok) {
      dh_rec += carry;
      if (a.dh0) a.dh0[((size_t)dir * B + b) * H + unit] = dh_rec;
      if (a.dc0) a.dc0[((size_t)dir * B + b) * H + unit] = dc_rec;
    }
  }
}


// Rows of a (nblk * 4H, C) matrix between torch's gate-major order (row g*H + u inside each block of 4H rows: the
// order of weight_ih / weight_hh / bias rows, gates i,f,g,o) and the recurrence's gate-interleaved order (row 4u + g).
//   back == 0: dst[4u + g] = src[g H + u]            (weights and biases -> the order gx / gates / dgx are kept in)
//   back == 1: dst[g H + u] (+)= src[4u + g]         (weight gradients -> the order of the parameter tensors)
__global__ __launch_bounds__(256) void gate_rows_kernel(const float* __restrict__ src, float* __restrict__ dst, int H,
                                                        int C, int ld_src, int ld_dst, int back, int accumulate, int nrows) {
  const int rows4 = 4 * H;
  const bool vec = ((C | ld_src | ld_dst) & 3) == 0 && (((uintptr_t)src | (uintptr_t)dst) & 15) == 0;
  // a block walks rows blockIdx.x, + gridDim.x, ...: a few hundred fat blocks instead of one per row (r03: the weight-
  // gradient reorders run on the side stream beside a persistent grid, where 7168 one-row blocks took up to 1.3 ms)
  for (int row = blockIdx.x; row < nrows; row += gridDim.x) {
    const int blk = row / rows4, r = row % rows4;  // r: interleaved index 4u + g
    const size_t ri = (size_t)blk * rows4 + r, rg = (size_t)blk * rows4 + (size_t)(r & 3) * H + (r >> 2);
    const float* s = src + (back ? ri : rg) * ld_src;
    float* d = dst + (back ? rg : ri) * ld_dst;
    if (vec) {
      for (int c = threadIdx.x * 4; c < C; c += 1024) {
        float4 v = *reinterpret_cast<const float4*>(s + c);
        if (accumulate) {
          const float4 o = *reinterpret_cast<const float4*>(d + c);
          v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
        }
        *reinterpret_cast<float4*>(d + c) = v;
      }
    } else {
      for (int c = threadIdx.x; c < C; c += 256) d[c] = accumulate ? d[c] + s[c] : s[c];
    }
    if (!accumulate)  // padding columns of a destination with a wider leading dimension read as zero
      for (int c = C + threadIdx.x; c < ld_dst; c += 256) d[c] = 0.f;
  }
}

template <int KS, bool BF>
int launch_fwd(const FwdArgs& a, int nblocks, hipStream_t st) {
  if (a.offs)
    hipLaunchKernelGGL((lstm_fwd_kernel<KS, BF, 8, false, true>), dim3((unsigned)nblocks), dim3(512), 0, st, a);
  else
    hipLaunchKernelGGL((lstm_fwd_kernel<KS, BF, 8>), dim3((unsigned)nblocks), dim3(512), 0, st, a);
  return 0;
}
template <int KS>
void launch_fwd_s3(const FwdArgs& a, int nblocks, hipStream_t st) {
  if (a.offs)
    hipLaunchKernelGGL((lstm_fwd_kernel<KS, false, 8, true, true>), dim3((unsigned)nblocks), dim3(512), 0, st, a);
  else
    hipLaunchKernelGGL((lstm_fwd_kernel<KS, false, 8, true, false>), dim3((unsigned)nblocks), dim3(512), 0, st, a);
}
template <int KS, bool BF>
int launch_bwd(const BwdArgs& a, dim3 grid, hipStream_t st) {
  // G = 1: the instantiation that keeps 113 KB of LDS -- a workgroup of gemm_f32_kernel_pl3 (48 KB) fits beside it and the weight
  // gradients run CO-RESIDENT with the recurrence.  a.exclusive (mode bit 17) selects the 127 KB one instead: only a 32 KB split3
  // workgroup would fit, the pl3 kernel does not -- the recurrence keeps its CUs to itself and the weight gradients take the CUs the
  // grid leaves free.  The engine asks for that on RAGGED batches: once the short batch group's streams have left the grid, half
  // the chip is idle for the rest of the launch and the weight gradients run there for free, while co-resident they only slow the
  // long group's chain (measured, profiles/r06_wgrad_planes.txt: ragged 28.98 -> 28.48 ms exclusive, 29.40 co-resident; uniform
  // 28.27 -> 27.95 co-resident, 28.11 exclusive).
  if (a.G == 1 && !a.exclusive)
    hipLaunchKernelGGL((lstm_bwd_kernel<KS, BF, 1>), dim3(grid.x * grid.y * grid.z), dim3(NTHREADS), 0, st, a);
  else
    hipLaunchKernelGGL((lstm_bwd_kernel<KS, BF, GMAX>), dim3(grid.x * grid.y * grid.z), dim3(NTHREADS), 0, st, a);
  return 0;
}

void launch_fwd_xl8(const FwdArgs& a, hipStream_t st) {  // bf16, 608 < H <= 896 (KS = 56): 8 XCDs x 32 blocks
  if (a.offs)
    hipLaunchKernelGGL((lstm_fwd_xl8_kernel<56, true>), dim3(256), dim3(512), 0, st, a);
  else
    hipLaunchKernelGGL((lstm_fwd_xl8_kernel<56, false>), dim3(256), dim3(512), 0, st, a);
}

int dispatch_fwd_s3(int KS, const FwdArgs& a, int nblocks, hipStream_t st) {
  switch (KS) {
    case 20: launch_fwd_s3<20>(a, nblocks, st); break;
    case 40: launch_fwd_s3<40>(a, nblocks, st); break;
    default: launch_fwd_s3<56>(a, nblocks, st); break;
  }
  return 0;
}

int dispatch_fwd(int KS, bool bf, const FwdArgs& a, int nblocks, hipStream_t st) {
  if (bf) switch (KS) {
      case 20: return launch_fwd<20, true>(a, nblocks, st);
      case 40: return launch_fwd<40, true>(a, nblocks, st);
      case 56: return launch_fwd<56, true>(a, nblocks, st);
      default: return launch_fwd<64, true>(a, nblocks, st);
    }
  switch (KS) {
    case 20: return launch_fwd<20, false>(a, nblocks, st);
    case 38: return launch_fwd<38, false>(a, nblocks, st);
    case 56: return launch_fwd<56, false>(a, nblocks, st);
    default: return launch_fwd<64, false>(a, nblocks, st);
  }
}
int dispatch_bwd(int KS, bool bf, const BwdArgs& a, dim3 grid, hipStream_t st) {
  if (bf) switch (KS) {
      case 20: return launch_bwd<20, true>(a, grid, st);
      case 40: return launch_bwd<40, true>(a, grid, st);
      case 56: return launch_bwd<56, true>(a, grid, st);
      default: return launch_bwd<64, true>(a, grid, st);
    }
  switch (KS) {
    case 20: return launch_bwd<20, false>(a, grid, st);
    case 38: return launch_bwd<38, false>(a, grid, st);
    case 56: return launch_bwd<56, false>(a, grid, st);
    default: return launch_bwd<64, false>(a, grid, st);
  }
}

int num_cus() {
  static int n = 0;
  if (!n) {
    int dev = 0;
    hipDeviceProp_t p;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) n = p.multiProcessorCount;
  }
  return n;
}

// Smallest number of batch groups per workgroup for which the whole grid (one workgroup per CU) is
// co-resident; 0 if even GMAX groups per workgroup do not fit (then the caller launches per step).
int groups_per_wg(int NUG, int NBG, int gmin, int wg_per_cu = 1) {
  const int cus = num_cus();
  for (int g = (gmin < 1 ? 1 : gmin); g <= GMAX; ++g)
    if (NUG * ((NBG + g - 1) / g) * 2 <= cus * wg_per_cu) return g;
  return 0;
}

int check_common(const char* fn, int T, int B, int H, const float* whh, int mode) {
  SK_CHECK_ARG(T > 0 && B > 0 && H > 0, "%s: bad sizes T=%d B=%d H=%d", fn, T, B, H);
  SK_CHECK_ARG(H % 4 == 0 && H <= 1024, "%s: hidden size %d must be a multiple of 4 and <= 1024", fn, H);
  SK_CHECK_ARG(((uintptr_t)whh % 16) == 0, "%s: whh must be 16-byte aligned", fn);
  SK_CHECK_ARG((mode & 0xff) >= 0 && (mode & 0xff) <= 2 && ((mode >> 8) & 0xff) <= GMAX && (mode >> 31) == 0,
               "%s: unknown mode %d", fn, mode);
  return SK_OK;
}

}  // namespace

// The numerics- or timing-changing macros this translation unit was built with (sk_build_flags, include/sepkern.h)
unsigned sk_lstm_build_flags() {
  unsigned f = 0;
#ifdef SK_TAIL_HALF
  f |= SK_BUILD_TIMING_ONLY;
#endif
#ifdef SK_BWD_BOUND38
  f |= SK_BUILD_TIMING_ONLY;
#endif
#ifdef SK_UNITS12_BOUND
  f |= SK_BUILD_TIMING_ONLY;
#endif
#ifdef SK_LSTM_STAMPS
  f |= SK_BUILD_STAMPS;
#endif
#ifdef SK_DMA_BUILTIN
  f |= SK_BUILD_TUNING;
#endif
  if (SK_BF_NSB != 4 || SK_BF_DEPTH != 2 || SK_F32_NSB != 8 || SK_F32_DEPTH != 3 || SK_POLL_SLEEP != 1 || SK_BWD_RING != 1 ||
      SK_FWD_RING != 2 || SK_POLL_DELAY != 0)
    f |= SK_BUILD_TUNING;
  if (SK_FWD_S3_FLIP != 1) f |= SK_BUILD_ARITH;
  return f;
}

extern "C" size_t sk_lstm_workspace_bytes(int T, int B, int H) {
  (void)T;
  if (B <= 0 || H <= 0 || pick_ks(H) == 0) return 0;
  const size_t a = ws_layout(B, H, false).total, b = ws_layout(B, H, true).total;  // either precision
  return a > b ? a : b;
}

extern "C" int sk_lstm_fwd(const float* gx, const float* whh, const float* h0, const float* c0, const int32_t* lens,
                           const int32_t* offs, float* y, float* gates, float* cs, float* hn, float* cn, void* ws, int T,
                           int B, int H, int mode, sk_stream_t stream) {
  SK_CHECK_ARG(gx && whh && h0 && c0 && lens && y && ws, "sk_lstm_fwd: null pointer");
  SK_CHECK_ARG((gates == nullptr) == (cs == nullptr), "sk_lstm_fwd: gates and cs must be given together");
  SK_CHECK_ARG(((uintptr_t)h0 % 16) == 0, "sk_lstm_fwd: h0 must be 16-byte aligned");
  SK_CHECK_ARG(((uintptr_t)gx % 16) == 0 && ((uintptr_t)gates % 16) == 0, "sk_lstm_fwd: gx / gates must be 16-byte aligned");
  int rc = check_common("sk_lstm_fwd", T, B, H, whh, mode);
  if (rc) return rc;
  const int gmin = (mode >> 8) & 0xff;  // bits 8..15: minimum batch groups per workgroup (frees CUs for concurrent kernels)
  const bool bf = (mode >> 16) & 1;     // bit 16: bf16 matrix-core inputs
  SK_CHECK_ARG(!((mode >> 17) & 1), "sk_lstm_fwd: mode bit 17 (8-unit / 256-thread workgroups) was retired in r05");
  const int map = (mode >> 18) & 3;     // bits 18..19: block id -> stream assignment (speed only)
  int opt = (mode >> 20) & 7;     // bit 20: one polling wave per workgroup; bit 21: flags replicated per XCD;
                                        // bit 22: one flag per 128-byte line
  if (opt & 4) opt &= ~2;              // one flag per line: no replicas on top (the flag block is sized for either)
  // bit 28 (fp32): the product by the exact three-way bf16 split on the bf16 matrix pipe (S3)
  const bool s3 = ((mode >> 28) & 1) && !bf && pick_ks(H, true) != 64;  // (KS = 64: 168 + 88 registers do not fit)
  const bool tagged = ((mode >> 29) & 1) && !bf && !s3;  // bit 29 (fp32)
  if (tagged) opt |= 8;                // the data is the flag (lstm_fwd_kernel)
  int poll_delay = (mode >> 23) & 31;  // bits 23..27: FwdArgs::poll_delay, units of 0.1 us; 0 = choose, 31 = none
  const bool xl8_bit = (mode >> 30) & 1;  // bit 30 (bf16): XCD-local streams of 8 rows (lstm_fwd_xl8_kernel)
  mode &= 0xff;
  const WsLayout L = ws_layout(B, H, bf || s3);  // (S3 exchanges bf16 images: the bf16 unit-group counts)
  hipStream_t st = (hipStream_t)stream;
  char* base = (char*)ws;
  FwdArgs a;
  a.gx = gx; a.whh = whh; a.h0 = h0; a.c0 = c0; a.lens = lens; a.offs = offs;
  a.y = y; a.gates = gates; a.cs = cs; a.hn = hn; a.cn = cn;
  a.xbuf = (float*)(base + L.xbuf); a.state = (float*)(base + L.state);
  a.flags = (unsigned*)(base + L.flags); a.ctrl = (unsigned*)(base + L.ctrl);
  a.T = T; a.B = B; a.H = H; a.NBG = L.NBG;
  const int NUG = L.KS;
  const int G = groups_per_wg(NUG, L.NBG, gmin, 1);
  const bool fits = G > 0;
  a.G = fits ? G : 1;
  const int nby = (L.NBG + a.G - 1) / a.G;
  const int nblocks = NUG * nby * 2;
  // hold-back of the first poll after one's own flag store (wait_flags): 0.8 us on a full grid, 0.4 us on a small one
  // (fewer flag stores to land); measured optimum for H 896/1024 (0.8) and H 256/300 (0.4), no effect at 2x600 / B = 100
  if (poll_delay == 0) poll_delay = nblocks >= 128 ? 8 : 4;
  if (poll_delay == 31) poll_delay = 0;
  a.map = map; a.nby = nby; a.opt = opt; a.poll_delay = poll_delay;
  SK_CHECK_ARG(mode != 1 || fits, "sk_lstm_fwd: persistent mode cannot keep B=%d H=%d co-resident on %d CUs", B, H, num_cus());
  SK_CHECK_HIP(hipMemsetAsync(base + L.ctrl, 0, L.xbuf - L.ctrl, st));  // per-launch status word + flags (not the sticky word)
  if ((opt & 8) && mode != 2)  // tagged words: a new sequence must not find an old one's epochs in the buffers
    SK_CHECK_HIP(hipMemsetAsync(base + L.xbuf, 0, L.state - L.xbuf, st));
  // XCD-local streams of 8 rows: where the shape allows it -- the 56-chunk instantiation (608 < H <= 896: 28 workgroups of 32 units
  // per stream), at most 8 streams (B <= 32), a device of 8 XCDs x 32 CUs, a persistent launch; anything else runs the ordinary form
  const bool xl8 = xl8_bit && bf && L.KS == 56 && B <= 32 && gmin <= 1 && num_cus() >= 256 && (mode == 1 || (mode == 0 && fits));
  if (xl8) {
    a.s_begin = 0; a.s_end = T;
    a.NBG = (B + 7) / 8;  // batch groups of EIGHT rows
    a.G = 1; a.nby = a.NBG;
    if (poll_delay == 8) a.poll_delay = 4;  // (a stream's flag stores land sooner: the small-grid hold-back)
    launch_fwd_xl8(a, st);
  } else if (mode == 1 || (mode == 0 && fits)) {
    a.s_begin = 0; a.s_end = T;
    s3 ? dispatch_fwd_s3(L.KS, a, nblocks, st) : dispatch_fwd(L.KS, bf, a, nblocks, st);
  } else {  // one launch per step: the state travels through the workspace
    for (int s = 0; s < T; ++s) {
      a.s_begin = s; a.s_end = s + 1;
      s3 ? dispatch_fwd_s3(L.KS, a, nblocks, st) : dispatch_fwd(L.KS, bf, a, nblocks, st);
    }
  }
  SK_CHECK_LAUNCH("sk_lstm_fwd");
  return SK_OK;
}

extern "C" int sk_lstm_bwd(const float* dy, const float* dhn, const float* dcn, const float* whh, const float* gates,
                           const float* cs, const float* c0, const int32_t* lens, const int32_t* offs, float* dgx,
                           float* dh0, float* dc0, float* dbias, void* dgx_bf16, int ld_bf16, int64_t plane_bf16, void* ws, int T,
                           int B, int H, int mode, sk_stream_t stream) {
  SK_CHECK_ARG(dy && whh && gates && cs && c0 && lens && dgx && ws, "sk_lstm_bwd: null pointer");
  SK_CHECK_ARG(!dgx_bf16 || (ld_bf16 >= 8 * H && ld_bf16 % 4 == 0 && ((uintptr_t)dgx_bf16 % 8) == 0 && plane_bf16 >= 0 && plane_bf16 % 4 == 0),
               "sk_lstm_bwd: bf16 twin needs ld >= 8H, ld %% 4 == 0, 8-byte alignment, plane stride %% 4");
  SK_CHECK_ARG(((uintptr_t)gates % 16) == 0 && ((uintptr_t)dgx % 16) == 0, "sk_lstm_bwd: gates / dgx must be 16-byte aligned");
  int rc = check_common("sk_lstm_bwd", T, B, H, whh, mode);
  if (rc) return rc;
  const int gmin = (mode >> 8) & 0xff;
  const bool bf = (mode >> 16) & 1;
  // block map as sk_lstm_fwd (+4: one flag per 128-byte line, mode bit 22); flag replication was measured here too
  // (7.59 -> 7.50 us/step) and not kept
  const int map = ((mode >> 18) & 3) | (((mode >> 22) & 1) << 2);
  int poll_delay = (mode >> 23) & 31;  // as sk_lstm_fwd; 0 = none here until measured otherwise
  const int fast = (mode >> 29) & 1;   // (diagnostic builds only, BwdArgs::fast)
  const bool xl8_bit = (mode >> 30) & 1;  // bit 30 (bf16): XCD-local streams of 8 rows (lstm_bwd_xl8_kernel)
  const int exclusive = (mode >> 17) & 1;  // bit 17: keep the CUs to the recurrence (launch_bwd)
  mode &= 0xff;
  const WsLayout L = ws_layout(B, H, bf);
  hipStream_t st = (hipStream_t)stream;
  char* base = (char*)ws;
  BwdArgs a;
  a.dy = dy; a.whh = whh; a.gates = gates; a.cs = cs; a.c0 = c0; a.lens = lens; a.offs = offs;
  a.dgx = dgx; a.dh0 = dh0; a.dc0 = dc0; a.dhn = dhn; a.dcn = dcn;
  a.dbias = dbias;
  a.dgx_bf = (__bf16*)dgx_bf16; a.ld_bf = ld_bf16; a.pl_bf = plane_bf16; a.fast = fast; a.exclusive = exclusive;
  SK_CHECK_ARG(!plane_bf16 || !bf, "sk_lstm_bwd: planes of dgx are the fp32 configuration's (mode bit 16 = bf16 inputs is set)");
  a.xbuf = (float*)(base + L.xbuf); a.state = (float*)(base + L.state);
  a.flags = (unsigned*)(base + L.flags); a.ctrl = (unsigned*)(base + L.ctrl);
  a.T = T; a.B = B; a.H = H; a.NBG = L.NBG;
  const int want_d0 = (dh0 || dc0) ? 1 : 0;
  const int G = groups_per_wg(L.KS, L.NBG, gmin);
  const bool fits = G > 0;
  a.G = fits ? G : 1;
  const int nby = (L.NBG + a.G - 1) / a.G;
  if (poll_delay == 31) poll_delay = 0;
  a.map = map; a.nby = nby; a.poll_delay = poll_delay;
  dim3 grid((unsigned)L.KS, (unsigned)nby, 2);
  SK_CHECK_ARG(mode != 1 || fits, "sk_lstm_bwd: persistent mode cannot keep B=%d H=%d co-resident on %d CUs", B, H, num_cus());
  SK_CHECK_HIP(hipMemsetAsync(base + L.ctrl, 0, L.xbuf - L.ctrl, st));
  if (dbias)  // (the kernels ADD their sums: a sequence advanced in step launches accumulates)
    SK_CHECK_HIP(hipMemsetAsync(dbias, 0, (size_t)L.NBG * 8 * H * sizeof(float), st));  // rows >= grid y stay 0
  const bool xl8 = xl8_bit && bf && L.KS == 56 && B <= 32 && gmin <= 1 && num_cus() >= 256 && (mode == 1 || (mode == 0 && fits));
  if (xl8) {  // (as sk_lstm_fwd's: 28 workgroups of 32 out units per (direction, 8-row group) stream, one XCD each)
    a.s_begin = 0; a.s_end = T; a.final_mm = want_d0;
    a.NBG = (B + 7) / 8;
    a.G = 1; a.nby = a.NBG;
    hipLaunchKernelGGL((lstm_bwd_xl8_kernel<56>), dim3(256), dim3(NTHREADS), 0, st, a);
  } else if (mode == 1 || (mode == 0 && fits)) {
    a.s_begin = 0; a.s_end = T; a.final_mm = want_d0;
    dispatch_bwd(L.KS, bf, a, grid, st);
  } else {
    a.final_mm = 0;  // a step launch never waits on other workgroups
    for (int s = 0; s < T; ++s) {
      a.s_begin = s; a.s_end = s + 1;
      dispatch_bwd(L.KS, bf, a, grid, st);
    }
    if (want_d0) {
      a.s_begin = T; a.s_end = T; a.final_mm = 1;
      dispatch_bwd(L.KS, bf, a, grid, st);
    }
  }
  SK_CHECK_LAUNCH("sk_lstm_bwd");
  return SK_OK;
}

extern "C" int sk_lstm_status(void* ws, sk_stream_t stream) {
  SK_CHECK_ARG(ws, "sk_lstm_status: null workspace");
  unsigned v = 0;
  SK_CHECK_HIP(hipMemcpyAsync(&v, ws, sizeof(v), hipMemcpyDeviceToHost, (hipStream_t)stream));
  SK_CHECK_HIP(hipStreamSynchronize((hipStream_t)stream));
  if (v != 0) {
    SK_CHECK_HIP(hipMemsetAsync(ws, 0, sizeof(v), (hipStream_t)stream));  // reported once
    return sk_fail(SK_ETIMEOUT, "sk_lstm: a workgroup's bounded wait timed out in a launch since the last check "
                                "(grid not co-resident?)");
  }
  return SK_OK;
}

extern "C" int sk_gate_rows(const float* src, float* dst, int nblk, int H, int C, int ld_src, int ld_dst, int back,
                            int accumulate, sk_stream_t stream) {
  SK_CHECK_ARG(src && dst && src != dst && nblk > 0 && H > 0 && C > 0 && ld_src >= C && ld_dst >= C,
               "sk_gate_rows: bad arguments");
  const int nrows = nblk * 4 * H;
  hipLaunchKernelGGL(gate_rows_kernel, dim3((unsigned)(nrows < 512 ? nrows : 512)), dim3(256), 0, (hipStream_t)stream, src, dst,
                     H, C, ld_src, ld_dst, back, accumulate, nrows);
  SK_CHECK_LAUNCH("sk_gate_rows");
  return SK_OK;
}
