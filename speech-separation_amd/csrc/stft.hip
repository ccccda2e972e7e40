// stft.hip -- framed STFT / mask-apply + iSTFT for n_fft = 512, gfx950.
//
// One wavefront owns one frame: the 512 real samples are packed into 256 complex points and
// go through a 4-stage radix-4 Stockham FFT whose stages exchange data through LDS (each of the
// 64 lanes does one radix-4 butterfly per stage), followed by the real-FFT split.  A workgroup
// (4 waves) handles 16 consecutive frames of one utterance so that the 75 %-overlapping samples
// are read from HBM once, coalesced, into LDS.  Both kernels are HBM-bound streaming kernels:
//   STFT   reads 128 new samples and writes 257 bins per frame,
//   iSTFT  reads 257 complex bins (+257 mask values) and writes 128 samples per frame.
//
// Reference semantics restated: librosa.core.stft / istft as used at
// steps/extract_feats.py:85-89,104-105 and steps/reconstruct_sources.py:39-42 (see oracle/stft.py).
#include "sk_common.h"
#include "tables512.inc"

namespace {

constexpr int NFFT = 512;
constexpr int NBIN = 257;
constexpr int HOP = 128;
constexpr int FPB = 16;  // frames per workgroup (STFT)
constexpr int HPB = 16;  // hops of output per workgroup (iSTFT)
constexpr int IFR = HPB + 3;  // frames an iSTFT workgroup must invert

__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
  return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}

// In-place radix-4 DFT of v[0..3] (forward, e^{-i...}).
__device__ __forceinline__ void radix4(float2 (&v)[4]) {
  const float2 a0 = make_float2(v[0].x + v[2].x, v[0].y + v[2].y);
  const float2 a1 = make_float2(v[0].x - v[2].x, v[0].y - v[2].y);
  const float2 a2 = make_float2(v[1].x + v[3].x, v[1].y + v[3].y);
  const float2 d = make_float2(v[1].x - v[3].x, v[1].y - v[3].y);
  const float2 a3 = make_float2(d.y, -d.x);  // d * (-i)
  v[0] = make_float2(a0.x + a2.x, a0.y + a2.y);
  v[1] = make_float2(a1.x + a3.x, a1.y + a3.y);
  v[2] = make_float2(a0.x - a2.x, a0.y - a2.y);
  v[3] = make_float2(a1.x - a3.x, a1.y - a3.y);
}

// 256-point complex forward FFT of one wave.  v holds z[lane + 64 r] on entry; on exit the
// natural-order result is in `res` (Z[k], k = 0..255).  b0/b1 are the wave's two LDS buffers.
// All four waves of the block call this together (it contains block barriers).
__device__ __forceinline__ float2* fft256_wave(float2 (&v)[4], float2* b0, float2* b1, const float2* tw,
                                               int lane) {
  // stage 0: Ns = 1 (twiddles are 1), write b0[4 lane + r]
  radix4(v);
#pragma unroll
  for (int r = 0; r < 4; ++r) b0[4 * lane + r] = v[r];
  __syncthreads();
  // stage 1: Ns = 4
  {
    const int jm = lane & 3;
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = cmul(b0[lane + 64 * r], tw[(2 * r * jm * 16) & 511]);
    radix4(v);
    const int j0 = (lane >> 2) * 16 + jm;
#pragma unroll
    for (int r = 0; r < 4; ++r) b1[j0 + 4 * r] = v[r];
  }
  __syncthreads();
  // stage 2: Ns = 16
  {
    const int jm = lane & 15;
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = cmul(b1[lane + 64 * r], tw[(2 * r * jm * 4) & 511]);
    radix4(v);
    const int j0 = (lane >> 4) * 64 + jm;
#pragma unroll
    for (int r = 0; r < 4; ++r) b0[j0 + 16 * r] = v[r];
  }
  __syncthreads();
  // stage 3: Ns = 64
  {
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = cmul(b0[lane + 64 * r], tw[(2 * r * lane) & 511]);
    radix4(v);
#pragma unroll
    for (int r = 0; r < 4; ++r) b1[lane + 64 * r] = v[r];
  }
  __syncthreads();
  return b1;
}

__global__ __launch_bounds__(256) void stft_kernel(const void* __restrict__ wav, int pcm16,
                                                   const int64_t* __restrict__ wav_offs,
                                                   const int32_t* __restrict__ nsamp, int want_complex,
                                                   void* __restrict__ out, const int64_t* __restrict__ out_offs,
                                                   const int64_t* __restrict__ stride_t,
                                                   const int64_t* __restrict__ stride_f) {
  __shared__ float smp[NFFT + (FPB - 1) * HOP];
  __shared__ float2 tw[NFFT];
  __shared__ float2 fbuf[4][2][256];
  __shared__ float2 ost[FPB][NBIN];

  const int u = blockIdx.y;
  const int N = nsamp[u];
  const int T = 1 + N / HOP;
  const int t0 = blockIdx.x * FPB;
  if (t0 >= T) return;
  const int nfr = min(FPB, T - t0);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

  for (int i = tid; i < NFFT; i += 256) tw[i] = g_tw512[i];
  // reflect-padded samples [t0*HOP, t0*HOP + 512 + (nfr-1)*HOP) of the padded signal
  const int span = NFFT + (nfr - 1) * HOP;
  const int64_t woff = wav_offs[u];
  for (int i = tid; i < span; i += 256) {
    int src = t0 * HOP + i - NFFT / 2;
    if (src < 0) src = -src;
    if (src >= N) src = 2 * (N - 1) - src;
    src = max(0, min(src, N - 1));
    float v;
    if (pcm16)
      v = (float)((const int16_t*)wav)[woff + src] * (1.0f / 32768.0f);
    else
      v = ((const float*)wav)[woff + src];
    smp[i] = v;
  }
  __syncthreads();

  for (int it = 0; it < FPB / 4; ++it) {
    const int fr = it * 4 + wave;
    const bool active = fr < nfr;
    float2 v[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = lane + 64 * r;
      if (active) {
        const float2 w = *reinterpret_cast<const float2*>(&g_hann512[2 * n]);
        v[r] = make_float2(smp[fr * HOP + 2 * n] * w.x, smp[fr * HOP + 2 * n + 1] * w.y);
      } else {
        v[r] = make_float2(0.f, 0.f);
      }
    }
    const float2* Z = fft256_wave(v, fbuf[wave][0], fbuf[wave][1], tw, lane);
    if (active) {
      // real-FFT split: X[k] = Xe + W^k Xo, Xe = (Z[k] + conj Z[256-k])/2, Xo = -i (Z[k] - conj Z[256-k])/2
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int k = lane + 64 * q;
        const float2 zk = Z[k];
        const float2 zc = Z[(256 - k) & 255];
        const float2 xe = make_float2(0.5f * (zk.x + zc.x), 0.5f * (zk.y - zc.y));
        const float2 dd = make_float2(zk.x - zc.x, zk.y + zc.y);
        const float2 xo = make_float2(0.5f * dd.y, -0.5f * dd.x);
        const float2 t = cmul(tw[k], xo);
        ost[fr][k] = make_float2(xe.x + t.x, xe.y + t.y);
      }
      if (lane == 0) ost[fr][256] = make_float2(Z[0].x - Z[0].y, 0.f);
    }
    __syncthreads();
  }

  const int64_t ooff = out_offs[u];
  const int64_t st = stride_t[u], sf = stride_f[u];
  const int total = nfr * NBIN;
  for (int i = tid; i < total; i += 256) {
    int fr, k;
    if (st == 1) {  // bin-major (F,T): consecutive threads -> consecutive frames
      k = i / nfr;
      fr = i - k * nfr;
    } else {  // frame-major: consecutive threads -> consecutive bins
      fr = i / NBIN;
      k = i - fr * NBIN;
    }
    const float2 x = ost[fr][k];
    const int64_t o = ooff + (int64_t)(t0 + fr) * st + (int64_t)k * sf;
    if (want_complex)
      ((float2*)out)[o] = x;
    else
      ((float*)out)[o] = sqrtf(x.x * x.x + x.y * x.y);
  }
}

__global__ __launch_bounds__(256) void istft_kernel(
    const float2* __restrict__ mix, const int64_t* __restrict__ mix_offs, const int64_t* __restrict__ mix_st,
    const int64_t* __restrict__ mix_sf, const float* __restrict__ mask, const int64_t* __restrict__ mask_offs,
    const int64_t* __restrict__ mask_st, const int64_t* __restrict__ mask_sf, const int32_t* __restrict__ nframes,
    int S, float* __restrict__ wav_out, int16_t* __restrict__ pcm_out, const int64_t* __restrict__ out_offs) {
  // rows[fr] holds the masked spectrum of frame fr (257 complex) and is then overwritten by its
  // windowed time-domain frame (512 floats); only the owning wave touches a row in between.
  __shared__ float2 rows[IFR][NBIN + 1];
  __shared__ float2 tw[NFFT];
  __shared__ float2 fbuf[4][2][256];

  const int us = blockIdx.y;
  const int u = us / S;
  const int T = nframes[u];
  const int nout = HOP * (T - 1);
  const int c = blockIdx.x;
  if (c * HPB * HOP >= nout) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tbase = c * HPB - 1;  // first frame that can touch this chunk (may be -1)

  for (int i = tid; i < NFFT; i += 256) tw[i] = g_tw512[i];

  // ---- masked spectra of frames tbase .. tbase+IFR-1 into LDS
  {
    const int64_t mo = mix_offs[u], mst = mix_st[u], msf = mix_sf[u];
    const int64_t ko = mask ? mask_offs[us] : 0, kst = mask ? mask_st[u] : 0, ksf = mask ? mask_sf[u] : 0;
    const int total = IFR * NBIN;
    for (int i = tid; i < total; i += 256) {
      int fr, k;
      if (mst == 1) {
        k = i / IFR;
        fr = i - k * IFR;
      } else {
        fr = i / NBIN;
        k = i - fr * NBIN;
      }
      const int t = tbase + fr;
      float2 x = make_float2(0.f, 0.f);
      if (t >= 0 && t < T) {
        x = mix[mo + (int64_t)t * mst + (int64_t)k * msf];
        if (mask) {
          const float m = mask[ko + (int64_t)t * kst + (int64_t)k * ksf];
          x.x *= m;
          x.y *= m;
        }
      }
      rows[fr][k] = x;
    }
  }
  __syncthreads();

  // ---- inverse real FFT of every frame, windowed, left in rows[fr] as 512 floats
  for (int it = 0; it < (IFR + 3) / 4; ++it) {
    const int fr = it * 4 + wave;
    const bool active = fr < IFR;
    float2 v[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int k = lane + 64 * r;
      if (active) {
        float2 a = rows[fr][k];
        float2 b = rows[fr][256 - k];
        if (k == 0) {  // irfft ignores Im X[0] and Im X[256]
          a.y = 0.f;
          b.y = 0.f;
        }
        b.y = -b.y;  // conj(X[256-k])
        const float2 xe = make_float2(0.5f * (a.x + b.x), 0.5f * (a.y + b.y));
        const float2 hd = make_float2(0.5f * (a.x - b.x), 0.5f * (a.y - b.y));
        const float2 wk = make_float2(tw[k].x, -tw[k].y);  // conj(W^k)
        const float2 xo = cmul(hd, wk);
        // Z = Xe + i Xo ; feed conj(Z) to the forward FFT
        v[r] = make_float2(xe.x - xo.y, -(xe.y + xo.x));
      } else {
        v[r] = make_float2(0.f, 0.f);
      }
    }
    // all reads of rows[fr] above are done by this wave before it overwrites the row below
    const float2* Y = fft256_wave(v, fbuf[wave][0], fbuf[wave][1], tw, lane);
    if (active) {
      float* tf = reinterpret_cast<float*>(&rows[fr][0]);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = lane + 64 * r;
        const float2 y = Y[n];
        const float2 w = *reinterpret_cast<const float2*>(&g_hann512[2 * n]);
        // z[n] = conj(Y[n]) / 256 ; x[2n] = Re z, x[2n+1] = Im z
        *reinterpret_cast<float2*>(&tf[2 * n]) =
            make_float2(w.x * (y.x * (1.0f / 256.0f)), w.y * (-y.y * (1.0f / 256.0f)));
      }
    }
    __syncthreads();
  }

  // ---- overlap-add in increasing frame order, window-sum-square normalisation, trim, convert
  const int64_t oo = out_offs[us];
  for (int i = tid; i < HPB * HOP; i += 256) {
    const int n = c * HPB * HOP + i;
    if (n >= nout) break;
    const int p = n + NFFT / 2;
    const int hp = p >> 7;
    float acc = 0.f, wss = 0.f;
#pragma unroll
    for (int j = 3; j >= 0; --j) {
      const int t = hp - j;
      if (t >= 0 && t < T) {
        const int m = p - t * HOP;
        const float* tf = reinterpret_cast<const float*>(&rows[t - tbase][0]);
        acc += tf[m];
        const float w = g_hann512[m];
        wss += w * w;
      }
    }
    if (wss > 1.17549435e-38f) acc /= wss;
    if (wav_out) wav_out[oo + n] = acc;
    if (pcm_out) {
      const float sv = acc * 32767.0f;
      pcm_out[oo + n] = (int16_t)(long long)sv;  // truncation toward zero, wrap on overflow
    }
  }
}

}  // namespace

extern "C" int sk_stft(const void* wav, int pcm16, const int64_t* wav_offs, const int32_t* nsamp, int nutt, int n_fft,
                       int hop, int want_complex, void* out, const int64_t* out_offs, const int64_t* stride_t,
                       const int64_t* stride_f, int max_frames, sk_stream_t stream) {
  SK_CHECK_ARG(n_fft == NFFT && hop == HOP, "sk_stft: only n_fft=512, hop=128 are built (got %d, %d)", n_fft, hop);
  SK_CHECK_ARG(wav && wav_offs && nsamp && out && out_offs && stride_t && stride_f, "sk_stft: null pointer");
  SK_CHECK_ARG(nutt > 0 && nutt <= 65535 && max_frames > 0, "sk_stft: bad nutt/max_frames");
  dim3 grid((unsigned)sk_cdiv(max_frames, FPB), (unsigned)nutt);
  hipLaunchKernelGGL(stft_kernel, grid, dim3(256), 0, (hipStream_t)stream, wav, pcm16, wav_offs, nsamp, want_complex,
                     out, out_offs, stride_t, stride_f);
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
  dim3 grid((unsigned)sk_cdiv((int64_t)HOP * (max_frames - 1), HPB * HOP), (unsigned)(nutt * S));
  hipLaunchKernelGGL(istft_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const float2*)mix_c64, mix_offs, mix_st,
                     mix_sf, mask, mask_offs, mask_st, mask_sf, nframes, S, wav_out, pcm_out, out_offs);
  SK_CHECK_LAUNCH("sk_mask_istft");
  return SK_OK;
}
