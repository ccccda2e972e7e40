"""CPU restatement of the reference's STFT front end and iSTFT back end.

TEST INFRASTRUCTURE (see oracle/__init__.py).  PARITY UNPINNED: the reference
delegates to librosa (`librosa.core.load/stft/istft`, version not pinned; API
usage matches 0.6.x), which is absent from /root/reference and from this image.
Call sites restated here:

  steps/extract_feats.py:85-89   load(sr=8000) -> stft(n_fft=512, hop=128) -> abs   (train)
  steps/extract_feats.py:104-105 load -> stft (complex kept)                       (test)
  steps/reconstruct_sources.py:39-42  mix_spec*mask -> istft(hop=128) -> *32767 -> int16

librosa semantics restated (documented defaults):
  stft : center=True, pad_mode='reflect', window='hann' == scipy get_window(fftbins=True)
         (periodic Hann), win_length=n_fft, output complex64 (n_fft/2+1, 1+N//hop)
  istft: win_length=n_fft=2*(F-1), same window, per-frame real inverse FFT times the
         window, overlap-add in increasing frame order, divide by the squared-window
         sum where it exceeds float32 tiny, trim n_fft//2 from both ends, float32.
  load : int16 PCM / 32768 -> float32 (no resampling when the file is already 8 kHz)
"""
import numpy as np


def hann_periodic(n_fft):
    """scipy.signal.get_window('hann', n_fft, fftbins=True) restated, float32."""
    n = np.arange(n_fft, dtype=np.float64)
    return (0.5 - 0.5 * np.cos(2.0 * np.pi * n / n_fft)).astype(np.float32)


def num_frames(n_samples, hop=128):
    """T = 1 + N//hop for center=True (steps/extract_feats.py:87 via librosa.stft)."""
    return 1 + n_samples // hop


def pcm16_to_float(pcm):
    """librosa.load of a 16-bit wav: samples / 32768 as float32 (extract_feats.py:85)."""
    return (np.asarray(pcm, dtype=np.int16).astype(np.float32) / np.float32(32768.0))


def stft(y, n_fft=512, hop=128):
    """librosa.core.stft(y, n_fft, hop) restated -> complex64 (n_fft/2+1, T).

    steps/extract_feats.py:76,78,87,89,98,105.
    """
    y = np.asarray(y, dtype=np.float32)
    if y.shape[0] <= n_fft // 2:
        raise ValueError("reflect padding needs more than n_fft//2 samples")
    win = hann_periodic(n_fft)
    ypad = np.pad(y, n_fft // 2, mode="reflect")
    T = 1 + (ypad.shape[0] - n_fft) // hop
    idx = np.arange(n_fft)[None, :] + hop * np.arange(T)[:, None]
    frames = ypad[idx] * win[None, :]                      # float32 product, as librosa
    spec = np.fft.rfft(frames.astype(np.float64), axis=1)  # (T, F)
    return spec.T.astype(np.complex64)


def stft_mag(y, n_fft=512, hop=128):
    """np.abs(stft) as stored for training (steps/extract_feats.py:87,89)."""
    return np.abs(stft(y, n_fft, hop)).astype(np.float32)


def istft(S, hop=128):
    """librosa.core.istft(S, hop_length=hop) restated -> float32 (hop*(T-1),).

    steps/reconstruct_sources.py:40.
    """
    S = np.asarray(S)
    F, T = S.shape
    n_fft = 2 * (F - 1)
    win = hann_periodic(n_fft)
    n = n_fft + hop * (T - 1)
    y = np.zeros(n, dtype=np.float32)
    wss = np.zeros(n, dtype=np.float32)
    win_sq = win * win
    for t in range(T):
        # irfft ignores the imaginary parts of the DC and Nyquist bins, exactly as
        # Re(ifft(hermitian-extended spectrum)) does in librosa.
        ytmp = win * np.fft.irfft(S[:, t].astype(np.complex128), n=n_fft).astype(np.float32)
        y[t * hop:t * hop + n_fft] += ytmp
        wss[t * hop:t * hop + n_fft] += win_sq
    nz = wss > np.finfo(np.float32).tiny
    y[nz] /= wss[nz]
    return y[n_fft // 2:n - n_fft // 2]


def to_int16_wav(s):
    """wav = s*32767; wav.astype('int16') (steps/reconstruct_sources.py:41-42).

    C truncation toward zero, NO clipping: values beyond int16 wrap (numpy's cast of an
    out-of-range float is implementation-defined; on x86-64 it wraps through int32/64).
    The oracle makes the wrap explicit so the expected result is platform independent.
    """
    v = np.asarray(s, dtype=np.float32) * np.float32(32767.0)
    i = np.trunc(v.astype(np.float64)).astype(np.int64)
    return ((i + 32768) % 65536 - 32768).astype(np.int16)


def reconstruct(mix_spec, mask, hop=128):
    """One source of steps/reconstruct_sources.py:39-42: mask-apply + iSTFT + int16."""
    S = np.multiply(mix_spec, mask)
    s = istft(S, hop)
    return s, to_int16_wav(s)
