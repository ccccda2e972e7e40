// stft.hip -- framed STFT / mask-apply + iSTFT for n_fft = 512, gfx950.
//
// The 512 real samples of a frame are packed into 256 complex points; the 256-point FFT is computed by
// SIXTEEN lanes, each holding 16 points: two in-register 16-point FFTs (radix 4 x 4) with ONE transpose
// through LDS in between (256 = 16 x 16), then the real-FFT split, whose mirrored partner Z[256-k] lives
// in lane (16-j)%16 of the same group and is fetched with a cross-lane shuffle.  A wavefront therefore
// transforms 4 frames at once and a workgroup (4 waves) 16 consecutive frames of one utterance, whose
// 75 %-overlapping samples are read from HBM once, coalesced, into LDS.  Compared with one-frame-per-wave
// radix-4 stages (3 LDS exchanges) this is ~6x less LDS traffic and ~2x fewer twiddle multiplies.
// Both kernels are streaming kernels (HBM roofline):
//   STFT   reads 128 new samples and writes 257 bins per frame,
//   iSTFT  reads 257 complex bins (+257 mask values) and writes 128 samples per frame per source.
//
// Reference semantics restated: librosa.core.stft / istft as used at
// steps/extract_feats.py:85-89,104-105 and steps/reconstruct_sources.py:39-42 (see oracle/stft.py).
#include "sk_common.h"
#include "tables512.inc"

namespace {

constexpr int NFFT = 512;
constexpr int NBIN = 257;
constexpr int HOP = 128;
constexpr int FPB = 16;       // frames per workgroup (STFT and iSTFT): 4 waves x 4 frames
constexpr int XLD = 17;       // padded row of the 16 x 16 transpose (conflict-free column reads)
constexpr int TPB = 5;        // consecutive 16-frame tiles per STFT workgroup (next tile's samples are prefetched)

// Complex numbers are 2-vectors so that additions, scalings and the two halves of a complex product map onto
// the packed fp32 VALU ops (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32, swizzles and signs in their modifiers).
typedef float v2f __attribute__((ext_vector_type(2)));

__device__ __forceinline__ v2f cmul(v2f a, v2f b) {  // (a.x b.x - a.y b.y, a.x b.y + a.y b.x)
  const v2f bs = {-b.y, b.x};
  return a.xx * b + a.yy * bs;
}
__device__ __forceinline__ v2f mul_mi(v2f a) { return (v2f){a.y, -a.x}; }  // a * (-i)
__device__ __forceinline__ v2f conj(v2f a) { return (v2f){a.x, -a.y}; }
__device__ __forceinline__ v2f ld2(const float2* p) { return *reinterpret_cast<const v2f*>(p); }
__device__ __forceinline__ void st2(float2* p, v2f v) { *reinterpret_cast<v2f*>(p) = v; }

// A wave's LDS instructions execute in order, so data exchanged between the lanes of ONE wave needs no
// hardware barrier -- only a compiler fence so that the ds_writes stay ahead of the ds_reads that follow.
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// radix-4 DFT of (a, b, c, d) (forward, e^{-i...})
__device__ __forceinline__ void radix4(v2f& a, v2f& b, v2f& c, v2f& d) {
  const v2f s0 = a + c, s1 = a - c, s2 = b + d, s3 = mul_mi(b - d);
  a = s0 + s2;
  b = s1 + s3;
  c = s0 - s2;
  d = s1 - s3;
}

// In-register 16-point DFT, natural order in and out (16 = 4 x 4).
__device__ __forceinline__ void dft16(v2f (&x)[16]) {
  constexpr float C1 = 0.92387953251128674f, S1 = 0.38268343236508977f, R2 = 0.70710678118654752f;
  // step A: for n2 = 0..3 a radix-4 over n1 of x[4 n1 + n2]  ->  t[k1][n2] kept in x[4 k1 + n2]
#pragma unroll
  for (int n2 = 0; n2 < 4; ++n2) radix4(x[n2], x[4 + n2], x[8 + n2], x[12 + n2]);
  // twiddles W16^(n2 k1): (1,1)=W1 (1,2)=W2 (1,3)=W3 (2,1)=W2 (2,2)=W4 (2,3)=W6 (3,1)=W3 (3,2)=W6 (3,3)=W9
  x[4 + 1] = cmul(x[4 + 1], (v2f){C1, -S1});
  x[4 + 2] = cmul(x[4 + 2], (v2f){R2, -R2});
  x[4 + 3] = cmul(x[4 + 3], (v2f){S1, -C1});
  x[8 + 1] = cmul(x[8 + 1], (v2f){R2, -R2});
  x[8 + 2] = mul_mi(x[8 + 2]);
  x[8 + 3] = cmul(x[8 + 3], (v2f){-R2, -R2});
  x[12 + 1] = cmul(x[12 + 1], (v2f){S1, -C1});
  x[12 + 2] = cmul(x[12 + 2], (v2f){-R2, -R2});
  x[12 + 3] = cmul(x[12 + 3], (v2f){-C1, S1});
  // step B: for k1 = 0..3 a radix-4 over n2  ->  X[k1 + 4 k2] left in x[4 k1 + k2]
#pragma unroll
  for (int k1 = 0; k1 < 4; ++k1) radix4(x[4 * k1 + 0], x[4 * k1 + 1], x[4 * k1 + 2], x[4 * k1 + 3]);
  // transpose the 4 x 4 register tile so that x[k] = X[k]
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = a + 1; b < 4; ++b) {
      const v2f tmp = x[4 * a + b];
      x[4 * a + b] = x[4 * b + a];
      x[4 * b + a] = tmp;
    }
}

// 256-point forward FFT by the 16 lanes of one group (j = lane & 15).  On entry z[n1] = in[16 n1 + j];
// on exit z[k2] = Z[j + 16 k2].  xch: this GROUP's 16 x XLD float2 transpose area in LDS; tw = e^{-2 pi i m/512}.
// The transpose goes through a 16 x XLD FLOAT plane, real parts first, then imaginary parts: half the LDS of a
// complex plane (the STFT workgroup then fits four times per CU instead of three) for twice the LDS instructions.
// t256[16 k1 + j] = W256^(j k1): the inter-stage twiddles laid out so that the 16 lanes of a group read 128 contiguous
// bytes (read from the 512-entry table at (2 j k1) & 511 the even k1 are 2- to 8-way bank conflicts: r03).
__device__ __forceinline__ void fft256_g16(v2f (&z)[16], float* xch, const float2* t256, int j) {
  dft16(z);  // over n1: z[k1] = A[k1][n2 = j]
#pragma unroll
  for (int k1 = 1; k1 < 16; ++k1) z[k1] = cmul(z[k1], ld2(&t256[16 * k1 + j]));  // W256^(j k1)
#pragma unroll
  for (int k1 = 0; k1 < 16; ++k1) xch[k1 * XLD + j] = z[k1].x;
  wave_sync();
  float re[16];
#pragma unroll
  for (int n2 = 0; n2 < 16; ++n2) re[n2] = xch[j * XLD + n2];  // lane j now plays k1 = j
  wave_sync();
#pragma unroll
  for (int k1 = 0; k1 < 16; ++k1) xch[k1 * XLD + j] = z[k1].y;
  wave_sync();
#pragma unroll
  for (int n2 = 0; n2 < 16; ++n2) z[n2] = (v2f){re[n2], xch[j * XLD + n2]};
  wave_sync();
  dft16(z);  // over n2: z[k2] = Z[j + 16 k2]
}

// BINMAJOR == false: every utterance of the launch is written frame-major (stride_f == 1): lanes store their
// bins straight from registers.  BINMAJOR == true: arbitrary strides (the reference's on-disk (257, T)
// layout): results are staged in LDS and written with consecutive threads on consecutive frames.
template <bool BINMAJOR, bool CPLX>
__global__ __launch_bounds__(256, BINMAJOR ? 2 : 4) void stft_kernel(const void* __restrict__ wav, int pcm16,
                                                   const int64_t* __restrict__ wav_offs,
                                                   const int32_t* __restrict__ nsamp,
                                                   void* __restrict__ out, const int64_t* __restrict__ out_offs,
                                                   const int64_t* __restrict__ stride_t,
                                                   const int64_t* __restrict__ stride_f) {
  __shared__ __attribute__((aligned(16))) float smp[NFFT + (FPB - 1) * HOP];
  __shared__ __attribute__((aligned(16))) float win[NFFT];
  __shared__ float2 tw[NFFT];
  __shared__ float2 t256[256];
  __shared__ float xch[16][16 * XLD];  // one transpose plane per 16-lane group
  __shared__ float2 ost[BINMAJOR ? FPB : 1][BINMAJOR ? NBIN : 1];

  const int u = blockIdx.y;
  const int N = nsamp[u];
  const int T = 1 + N / HOP;
  const int tile0 = blockIdx.x * TPB;  // this block transforms tiles tile0 .. tile0+TPB-1 (FPB frames each)
  if (tile0 * FPB >= T) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t woff = wav_offs[u];
  const int64_t ooff = out_offs[u];
  const int64_t st = stride_t[u], sf = stride_f[u];
  constexpr int SPAN = NFFT + (FPB - 1) * HOP, SPT = (SPAN + 255) / 256;  // samples per tile / per thread

  // reflect-padded samples [t0*HOP, t0*HOP + SPAN) of the padded signal (zeros past the last frame of the
  // utterance).  Tiles whose whole span lies inside the utterance (all but the first and the last one or two)
  // take a block-uniform path without the reflect/clamp arithmetic.
  auto fetch = [&](int t0, float (&r)[SPT]) {
    const int nfr = min(FPB, T - t0);
    const int span = NFFT + (nfr - 1) * HOP;
    const int first = t0 * HOP - NFFT / 2;
    if (first >= 0 && first + SPAN <= N) {
      if (pcm16) {
        const int16_t* w = (const int16_t*)wav + woff + first;
#pragma unroll
        for (int q = 0; q < SPT; ++q) {
          const int i = tid + 256 * q;
          r[q] = (i < SPAN) ? (float)w[i] * (1.0f / 32768.0f) : 0.f;
        }
      } else {
        const float* w = (const float*)wav + woff + first;
#pragma unroll
        for (int q = 0; q < SPT; ++q) {
          const int i = tid + 256 * q;
          r[q] = (i < SPAN) ? w[i] : 0.f;
        }
      }
      return;
    }
#pragma unroll
    for (int q = 0; q < SPT; ++q) {
      const int i = tid + 256 * q;
      int src = first + i;
      if (src < 0) src = -src;
      if (src >= N) src = 2 * (N - 1) - src;
      src = max(0, min(src, N - 1));
      float v = pcm16 ? (float)((const int16_t*)wav)[woff + src] * (1.0f / 32768.0f) : ((const float*)wav)[woff + src];
      r[q] = (i < span) ? v : 0.f;
    }
  };
  auto stash = [&](const float (&r)[SPT]) {
#pragma unroll
    for (int q = 0; q < SPT; ++q) {
      const int i = tid + 256 * q;
      if (i < SPAN) smp[i] = r[q];
    }
  };

  for (int i = tid; i < NFFT; i += 256) {
    tw[i] = g_tw512[i];
    win[i] = g_hann512[i];
  }
  t256[tid] = g_tw512[(2 * (tid & 15) * (tid >> 4)) & 511];
  float pre[SPT];
  fetch(tile0 * FPB, pre);
  stash(pre);
  __syncthreads();

  const int j = lane & 15, g = lane >> 4;
  const int fr = 4 * wave + g;  // this group's frame within a tile
  const int partner = (lane & 48) | ((16 - j) & 15);
  for (int ti = 0; ti < TPB; ++ti) {
    const int t0 = (tile0 + ti) * FPB;
    if (t0 >= T) break;  // block-uniform
    const int nfr = min(FPB, T - t0);
    const bool more = ti + 1 < TPB && t0 + FPB < T;
    if (more) fetch(t0 + FPB, pre);  // next tile's samples travel while this tile is transformed
    const bool active = fr < nfr;
    v2f z[16];
#pragma unroll
    for (int n1 = 0; n1 < 16; ++n1) {  // packed point n = 16 n1 + j  <->  samples 2n, 2n+1
      const v2f sm = *reinterpret_cast<const v2f*>(&smp[fr * HOP + 32 * n1 + 2 * j]);
      const v2f w = *reinterpret_cast<const v2f*>(&win[32 * n1 + 2 * j]);
      z[n1] = sm * w;
    }
    fft256_g16(z, xch[4 * wave + g], t256, j);

    // real-FFT split: X[k] = Xe + W^k Xo, Xe = (Z[k] + conj Z[256-k])/2, Xo = -i (Z[k] - conj Z[256-k])/2, k = j + 16 k2.
    // Z[256-k] is register 15-k2 of lane (16-j)%16 of this group (register (16-k2)%16 of lane 0 itself when j == 0).
    // r03: the bins k and 256-k of a pair are formed by ONE lane from shared terms -- Xe[256-k] = conj Xe[k],
    // Xo[256-k] = conj Xo[k], W^(256-k) = -conj W^k, so with A = Xe[k], Bt = W^k Xo[k]:  X[k] = A + Bt,  X[256-k] = conj(A - Bt).
    // Lane j takes k2 = 0..7 (its partner's k2 = 8..15 are the mirrors of lane 16-j's 0..7): 16 shuffles, 8 twiddles and 8
    // complex products per lane instead of 32 / 16 / 16; lane 0 adds the self-paired bin 128 (k2 = 0 gives bins 0 and 256).
    // frame-major output row of this group's frame: lane j writes bins j + 16 k2 and 256 - j - 16 k2 at constant offsets
    constexpr int CW = CPLX ? 2 : 1;
    float* const orow = (float*)out + CW * (ooff + (int64_t)(t0 + fr) * st) + CW * j;
    float* const orow2 = (float*)out + CW * (ooff + (int64_t)(t0 + fr) * st) + CW * (256 - j);
    auto emit = [&](v2f x, int k, float* dstp) {
      if (BINMAJOR)
        st2(&ost[fr][k], x);
      else if (CPLX)
        *reinterpret_cast<v2f*>(dstp) = x;
      else
        *dstp = __builtin_amdgcn_sqrtf(x.x * x.x + x.y * x.y);
    };
    if (active) {  // uniform over the 16-lane group, which is all the shuffles below reach
#pragma unroll
      for (int k2 = 0; k2 < 8; ++k2) {
        v2f zc;
        zc.x = __shfl(z[15 - k2].x, partner, 64);
        zc.y = __shfl(z[15 - k2].y, partner, 64);
        if (j == 0) zc = z[(16 - k2) & 15];
        const v2f zk = z[k2], cz = conj(zc);
        const int k = j + 16 * k2;
        const v2f A = 0.5f * (zk + cz), Bt = cmul(ld2(&tw[k]), 0.5f * mul_mi(zk - cz));
        emit(A + Bt, k, orow + CW * 16 * k2);
        emit(conj(A - Bt), 256 - k, orow2 - CW * 16 * k2);
      }
      if (j == 0) {  // bin 128 pairs with itself
        const v2f zk = z[8], cz = conj(zk);
        emit(0.5f * (zk + cz) + cmul(ld2(&tw[128]), 0.5f * mul_mi(zk - cz)), 128, orow + CW * 128);
      }
    }
    __syncthreads();  // every wave is done with smp (and ost is complete)
    if (BINMAJOR) {
      const int total = nfr * NBIN;
      for (int i = tid; i < total; i += 256) {
        int f2, k;
        if (st == 1) {  // bin-major (F,T): consecutive threads -> consecutive frames
          k = i / nfr;
          f2 = i - k * nfr;
        } else {  // frame-major: consecutive threads -> consecutive bins
          f2 = i / NBIN;
          k = i - f2 * NBIN;
        }
        const float2 x = ost[f2][k];
        const int64_t o = ooff + (int64_t)(t0 + f2) * st + (int64_t)k * sf;
        if (CPLX)
          ((float2*)out)[o] = x;
        else
          ((float*)out)[o] = __builtin_amdgcn_sqrtf(x.x * x.x + x.y * x.y);
      }
    }
    if (more) {
      stash(pre);
      __syncthreads();
    }
  }
}

constexpr int RING = FPB + 3;  // row slots of the iSTFT ring: one tile of frames + the 3 frames before it
// float2 per slot: a 257-bin spectrum, then the FFT's 16 x 17 float transpose plane, then 512 samples.  272 (r03; was 258):
// consecutive slots lie 544 dwords = 32 banks (mod 64) apart, so the two frames a 32-lane group of ds_read_b64 covers fall
// on opposite halves of the bank row; and bin k of frame t sits at index k ^ ((t + 3) & 15) -- a permutation inside each
// aligned group of 16 bins -- so that the 16 FRAMES x one bin a 16-lane group of the tile load writes (rows now a multiple
// of 32 banks apart) land on 16 different bank pairs.  Both were 2-way conflicts (35 % of the kernel's LDS cycles, r02 PMC).
constexpr int RLD = 272;

// Masked spectra of frames [tfirst, tfirst + COUNT) into their ring slots (zeros outside [0, T)).
template <int COUNT>
__device__ __forceinline__ void istft_load(float2 (*rows)[RLD], int tfirst, int T, const float2* __restrict__ mix,
                                           int64_t mo, int64_t mst, int64_t msf, const float* __restrict__ mask,
                                           int64_t ko, int64_t kst, int64_t ksf) {
  for (int i = threadIdx.x; i < COUNT * NBIN; i += 256) {
    int fr, k;
    if (mst == 1) {  // bin-major (257, T): consecutive threads -> consecutive frames
      k = i / COUNT;
      fr = i - k * COUNT;
    } else {
      fr = i / NBIN;
      k = i - fr * NBIN;
    }
    const int t = tfirst + fr;
    float2 x = make_float2(0.f, 0.f);
    if (t >= 0 && t < T) {
      x = mix[mo + (int64_t)t * mst + (int64_t)k * msf];
      if (mask) {
        const float m = mask[ko + (int64_t)t * kst + (int64_t)k * ksf];
        x.x *= m;
        x.y *= m;
      }
    }
    rows[(t + 3) % RING][k ^ ((t + 3) & 15)] = x;
  }
}

// Inverse real FFT of frame t by one 16-lane group; the windowed 512 samples replace the spectrum in the slot.
// key = (t + 3) & 15: the slot's bin permutation (RLD).
__device__ __forceinline__ void istft_frame(float2* row, const float2* tw, const float2* t256, const float* win, int j, int key) {
  v2f z[16];
#pragma unroll
  for (int n1 = 0; n1 < 16; ++n1) {
    const int k = 16 * n1 + j;
    v2f a = ld2(&row[k ^ key]);
    v2f b = ld2(&row[(256 - k) ^ key]);
    if (k == 0) {  // irfft ignores Im X[0] and Im X[256]
      a.y = 0.f;
      b.y = 0.f;
    }
    b = conj(b);  // conj(X[256-k])
    const v2f xe = 0.5f * (a + b), hd = 0.5f * (a - b);
    const v2f xo = cmul(hd, conj(ld2(&tw[k])));  // * conj(W^k)
    // Z = Xe + i Xo ; feed conj(Z) to the forward FFT
    z[n1] = (v2f){xe.x - xo.y, -(xe.y + xo.x)};
  }
  wave_sync();  // the spectrum is in registers: the slot now serves as the transpose area, then as the output
  // transpose plane: the odd one of the two groups that share a 32-lane half sits 16 banks further (slots are a multiple
  // of 32 banks apart), so the two groups' column writes / row reads never meet on a bank
  fft256_g16(z, reinterpret_cast<float*>(row) + 16 * ((threadIdx.x >> 4) & 1), t256, j);
  float* tf = reinterpret_cast<float*>(row);
#pragma unroll
  for (int k2 = 0; k2 < 16; ++k2) {  // z[k2] = Y[n], n = j + 16 k2; x[2n] = Re Y / 256, x[2n+1] = -Im Y / 256
    const int n = j + 16 * k2;
    const v2f w = *reinterpret_cast<const v2f*>(&win[2 * n]);
    *reinterpret_cast<v2f*>(&tf[2 * n]) = w * (z[k2] * (v2f){1.0f / 256.0f, -1.0f / 256.0f});
  }
}

// Bin-major fast path of the tile load (the reference's (257, T) layout: stride_t == 1 for spectrum and mask): thread
// (fr = tid & 15, k0 = tid >> 4) owns frame t0 + fr and bins k0, k0 + 16, ... -- 17 loads at a constant pointer
// stride, no per-element index arithmetic.  fetch() only issues the global loads (into registers: they travel while
// the current tile is transformed and overlap-added), stash() applies the mask and fills the ring slot.
struct TileRegs {
  float2 x[17];
  float m[17];
};

__device__ __forceinline__ void istft_fetch(TileRegs& r, int t0, int T, const float2* __restrict__ mix, int64_t mo, int64_t msf,
                                            const float* __restrict__ mask, int64_t ko, int64_t ksf) {
  const int fr = threadIdx.x & 15, k0 = threadIdx.x >> 4;
  const int t = t0 + fr;
  const bool ok = t >= 0 && t < T;
  const float2* pm = mix + mo + (ok ? t : 0) + (int64_t)k0 * msf;
  const float* pk = mask ? mask + ko + (ok ? t : 0) + (int64_t)k0 * ksf : nullptr;
#pragma unroll
  for (int q = 0; q < 17; ++q) {
    const bool live = ok && (q < 16 || k0 == 0);  // bin 256 = k0 0, q 16
    r.x[q] = live ? pm[(int64_t)q * 16 * msf] : make_float2(0.f, 0.f);
    r.m[q] = (live && pk) ? pk[(int64_t)q * 16 * ksf] : 1.f;
  }
}

__device__ __forceinline__ void istft_stash(const TileRegs& r, float2 (*rows)[RLD], int t0) {
  const int fr = threadIdx.x & 15, k0 = threadIdx.x >> 4;
  float2* row = rows[(t0 + fr + 3) % RING] + (k0 ^ ((t0 + fr + 3) & 15));
#pragma unroll
  for (int q = 0; q < 17; ++q)
    if (q < 16 || k0 == 0) row[16 * q] = make_float2(r.x[q].x * r.m[q], r.x[q].y * r.m[q]);
}

// One workgroup reconstructs `tpb` consecutive 16-hop tiles of one (utterance, source): every frame is read
// and transformed once; the 3 frames that overlap into the next tile stay in the LDS ring.
__global__ __launch_bounds__(256, 3) void istft_kernel(
    const float2* __restrict__ mix, const int64_t* __restrict__ mix_offs, const int64_t* __restrict__ mix_st,
    const int64_t* __restrict__ mix_sf, const float* __restrict__ mask, const int64_t* __restrict__ mask_offs,
    const int64_t* __restrict__ mask_st, const int64_t* __restrict__ mask_sf, const int32_t* __restrict__ nframes,
    int S, float* __restrict__ wav_out, int16_t* __restrict__ pcm_out, const int64_t* __restrict__ out_offs, int tpb) {
  // 48.4 KB: three workgroups per CU.  (Reading the 6 KB of tables through the vector L1 instead would fit four, but
  // the 64-bit gather addresses push the kernel over 128 VGPRs into scratch: measured 2.7 -> 4.9 ms.)
  __shared__ __attribute__((aligned(16))) float2 rows[RING][RLD];
  __shared__ __attribute__((aligned(16))) float win[NFFT];
  __shared__ float2 tw[NFFT];
  __shared__ float2 t256[256];

  const int us = blockIdx.y;
  const int u = us / S;
  const int T = nframes[u];
  const int nout = HOP * (T - 1);
  const int ntiles = T / FPB + 1;  // hops 0 .. T carry output samples
  const int tile0 = blockIdx.x * tpb;
  if (tile0 >= ntiles) return;
  const int tile1 = min(ntiles, tile0 + tpb);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 15, fr = 4 * __builtin_amdgcn_readfirstlane(wave) + (lane >> 4);
  const int64_t mo = mix_offs[u], mst = mix_st[u], msf = mix_sf[u];
  const int64_t ko = mask ? mask_offs[us] : 0, kst = mask ? mask_st[u] : 0, ksf = mask ? mask_sf[u] : 0;
  const int64_t oo = out_offs[us];
  const bool binmajor = mst == 1 && (!mask || kst == 1);  // block-uniform

  TileRegs pre;
  if (binmajor) istft_fetch(pre, tile0 * FPB, T, mix, mo, msf, mask, ko, ksf);
  for (int i = tid; i < NFFT; i += 256) {
    tw[i] = g_tw512[i];
    win[i] = g_hann512[i];
  }
  t256[tid] = g_tw512[(2 * (tid & 15) * (tid >> 4)) & 511];
  if (tile0 > 0) {  // the 3 frames before this block's first tile
    const int tf = tile0 * FPB - 3;
    istft_load<3>(rows, tf, T, mix, mo, mst, msf, mask, ko, kst, ksf);
    __syncthreads();
    if (fr < 3) istft_frame(rows[(tf + fr + 3) % RING], tw, t256, win, j, (tf + fr + 3) & 15);
  }
  // overlap-add roles: thread (m0 = tid & 127, hsel = tid >> 7) produces sample m0 of hops t0 + hsel + 2q, q = 0..7.
  // With all four taps inside [0, T) the window-sum-square is a constant of the thread.
  const int m0 = tid & (HOP - 1), hsel = __builtin_amdgcn_readfirstlane(tid >> 7);  // wave-uniform: the slot arithmetic stays scalar
  float wss_full = 0.f;
  __syncthreads();  // win is complete
#pragma unroll
  for (int q = 3; q >= 0; --q) {
    const float w = win[q * HOP + m0];
    wss_full += w * w;
  }
  const float inv_full = 1.0f / wss_full;

  for (int tile = tile0; tile < tile1; ++tile) {
    const int t0 = tile * FPB;
    __syncthreads();  // the previous tile's overlap-add is done with the slots about to be refilled
    if (binmajor)
      istft_stash(pre, rows, t0);
    else
      istft_load<FPB>(rows, t0, T, mix, mo, mst, msf, mask, ko, kst, ksf);
    __syncthreads();
    if (binmajor && tile + 1 < tile1) istft_fetch(pre, t0 + FPB, T, mix, mo, msf, mask, ko, ksf);  // travels under the FFTs
    istft_frame(rows[(t0 + fr + 3) % RING], tw, t256, win, j, (t0 + fr + 3) & 15);
    __syncthreads();
    // overlap-add in increasing frame order, window-sum-square normalisation, trim, convert
    int slot = (t0 + hsel) % RING;  // ring slot of frame hp - 3 (frame t lives in slot (t + 3) % RING)
#pragma unroll
    for (int q = 0; q < FPB * HOP / 256; ++q) {
      const int hp = t0 + hsel + 2 * q;
      const int n = hp * HOP + m0 - NFFT / 2;
      const int s3 = slot, s2 = slot + 1 >= RING ? slot + 1 - RING : slot + 1, s1 = slot + 2 >= RING ? slot + 2 - RING : slot + 2,
                s0 = slot + 3 >= RING ? slot + 3 - RING : slot + 3;
      slot = slot + 2 >= RING ? slot + 2 - RING : slot + 2;
      if (n < 0 || n >= nout) continue;
      float acc;
      if (hp >= 3 && hp < T) {  // all four frames exist
        acc = reinterpret_cast<const float*>(rows[s3])[3 * HOP + m0];
        acc += reinterpret_cast<const float*>(rows[s2])[2 * HOP + m0];
        acc += reinterpret_cast<const float*>(rows[s1])[HOP + m0];
        acc += reinterpret_cast<const float*>(rows[s0])[m0];
        acc *= inv_full;
      } else {
        acc = 0.f;
        float wss = 0.f;
        const int sl[4] = {s0, s1, s2, s3};
#pragma unroll
        for (int qq = 3; qq >= 0; --qq) {
          const int t = hp - qq;
          if (t >= 0 && t < T) {
            const int m = qq * HOP + m0;
            acc += reinterpret_cast<const float*>(rows[sl[qq]])[m];
            const float w = win[m];
            wss += w * w;
          }
        }
        if (wss > 1.17549435e-38f) acc /= wss;
      }
      if (wav_out) wav_out[oo + n] = acc;
      if (pcm_out) {
        const float sv = acc * 32767.0f;
        pcm_out[oo + n] = (int16_t)(long long)sv;  // truncation toward zero, wrap on overflow
      }
    }
  }
}

}  // namespace

extern "C" int sk_stft(const void* wav, int pcm16, const int64_t* wav_offs, const int32_t* nsamp, int nutt, int n_fft,
                       int hop, int want_complex, void* out, const int64_t* out_offs, const int64_t* stride_t,
                       const int64_t* stride_f, int frame_major, int max_frames, sk_stream_t stream) {
  SK_CHECK_ARG(n_fft == NFFT && hop == HOP, "sk_stft: only n_fft=512, hop=128 are built (got %d, %d)", n_fft, hop);
  SK_CHECK_ARG(wav && wav_offs && nsamp && out && out_offs && stride_t && stride_f, "sk_stft: null pointer");
  SK_CHECK_ARG(nutt > 0 && nutt <= 65535 && max_frames > 0, "sk_stft: bad nutt/max_frames");
  dim3 grid((unsigned)sk_cdiv(max_frames, FPB * TPB), (unsigned)nutt);
#define SK_STFT_LAUNCH(BM, CX)                                                                                 \
  hipLaunchKernelGGL((stft_kernel<BM, CX>), grid, dim3(256), 0, (hipStream_t)stream, wav, pcm16, wav_offs, nsamp, \
                     out, out_offs, stride_t, stride_f)
  if (frame_major) {
    if (want_complex) SK_STFT_LAUNCH(false, true); else SK_STFT_LAUNCH(false, false);
  } else {
    if (want_complex) SK_STFT_LAUNCH(true, true); else SK_STFT_LAUNCH(true, false);
  }
#undef SK_STFT_LAUNCH
  SK_CHECK_LAUNCH("sk_stft");
  return SK_OK;
}

extern "C" int sk_mask_istft(const void* mix_c64, const int64_t* mix_offs, const int64_t* mix_st,
                             const int64_t* mix_sf, const float* mask, const int64_t* mask_offs,
                             const int64_t* mask_st, const int64_t* mask_sf, const int32_t* nframes, int nutt, int S,
                             int n_fft, int hop, float* wav_out, int16_t* pcm_out, const int64_t* out_offs,
                             int max_frames, sk_stream_t stream) {
  SK_CHECK_ARG(n_fft == NFFT && hop == HOP, "sk_mask_istft: only n_fft=512, hop=128 are built");
  SK_CHECK_ARG(mix_c64 && mix_offs && mix_st && mix_sf && nframes && out_offs, "sk_mask_istft: null pointer");
  SK_CHECK_ARG(!mask || (mask_offs && mask_st && mask_sf), "sk_mask_istft: mask given without its strides");
  SK_CHECK_ARG(wav_out || pcm_out, "sk_mask_istft: no output buffer");
  SK_CHECK_ARG(nutt > 0 && S > 0 && (int64_t)nutt * S <= 65535 && max_frames > 1, "sk_mask_istft: bad sizes");
  // tiles of 16 hops per (utterance, source); a workgroup walks `tpb` of them so that every frame is transformed
  // once, as long as that still leaves a few workgroups per CU
  const int ntiles = max_frames / FPB + 1;
  const int64_t total = (int64_t)ntiles * nutt * S;
  const int tpb = (int)std::min<int64_t>(std::max<int64_t>(total / 2048, 2), ntiles);
  dim3 grid((unsigned)sk_cdiv(ntiles, tpb), (unsigned)(nutt * S));
  hipLaunchKernelGGL(istft_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const float2*)mix_c64, mix_offs, mix_st,
                     mix_sf, mask, mask_offs, mask_st, mask_sf, nframes, S, wav_out, pcm_out, out_offs, tpb);
  SK_CHECK_LAUNCH("sk_mask_istft");
  return SK_OK;
}
