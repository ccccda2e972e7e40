#!/usr/bin/env python3
"""HBM-roofline check of the streaming kernels at corpus scale (diagnostic): STFT (train layout, PCM in),
mask-apply + iSTFT (PCM out), PIT-MSE forward/backward.  Prints algorithmic GB/s (SURVEY.md 8d byte counts)
against the 8 TB/s spec / 6.3 TB/s achievable HBM rate."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from sepkern import _lib, ops  # noqa: E402


def timeit(fn, n=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    U, T, F = 4096, 400, 257                      # 4096 utterances x 400 frames = 1.6 M frames per launch
    N = 128 * (T - 1) + 64
    dev = "cuda"
    pcm = torch.randint(-20000, 20000, (U * N,), dtype=torch.int16, device=dev)
    i64 = lambda v: torch.tensor(v, dtype=torch.int64, device=dev)     # noqa: E731
    woffs, ns = i64([u * N for u in range(U)]), torch.full((U,), N, dtype=torch.int32, device=dev)
    out = torch.empty(U * T * F, device=dev)
    ooffs, st, sf = i64([u * T * F for u in range(U)]), i64([F] * U), i64([1] * U)
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = lambda t: C.c_void_p(t.data_ptr())       # noqa: E731

    def stft():
        _lib.call("sk_stft", p(pcm), 1, p(woffs), p(ns), U, 512, 128, 0, p(out), p(ooffs), p(st), p(sf), 1, T, stream)
    ms = timeit(stft)
    by = U * T * (128 * 2 + F * 4)
    print("stft_mag   (PCM16 in, (T,F) fp32 out): %8.3f ms  %7.1f M frames/s  %7.1f GB/s algorithmic (%.1f %% of 8 TB/s)"
          % (ms, U * T / ms / 1e3, by / ms / 1e6, by / ms / 1e6 / 80))

    spec = torch.randn(U * F * T, 2, device=dev).view(torch.complex64).reshape(-1)
    mask = torch.rand(U * F * T, device=dev)
    moffs, mst, msf = i64([u * F * T for u in range(U)]), i64([1] * U), i64([T] * U)
    nfr = torch.full((U,), T, dtype=torch.int32, device=dev)
    pcm_out = torch.empty(U * 128 * (T - 1), dtype=torch.int16, device=dev)
    po = i64([u * 128 * (T - 1) for u in range(U)])

    def istft():
        _lib.call("sk_mask_istft", p(spec), p(moffs), p(mst), p(msf), p(mask), p(moffs), p(mst), p(msf), p(nfr), U, 1, 512, 128,
                  None, p(pcm_out), p(po), T, stream)
    ms = timeit(istft)
    by = U * T * (F * 8 + F * 4 + 128 * 2)
    print("mask_istft ((F,T) c64 + mask in, PCM16 out): %6.3f ms  %7.1f M frames/s  %7.1f GB/s algorithmic (%.1f %% of 8 TB/s)"
          % (ms, U * T / ms / 1e3, by / ms / 1e6, by / ms / 1e6 / 80))

    B, S = 512, 2
    m = torch.rand(T, B, S * F, device=dev)
    mix = torch.rand(T, B, F, device=dev)
    srcs = [torch.rand(T, B, F, device=dev) for _ in range(S)]
    lens = torch.full((B,), T, dtype=torch.int32, device=dev)
    res = ops.pit_mse_fwd(m, mix, srcs, lens)
    ms = timeit(lambda: ops.pit_mse_fwd(m, mix, srcs, lens))
    by = T * B * (2 * S + 1) * F * 4
    print("pit_mse_fwd (S=2, %d x %d frames): %8.3f ms  %7.1f GB/s algorithmic (%.1f %% of 8 TB/s)" % (B, T, ms, by / ms / 1e6, by / ms / 1e6 / 80))
    one = torch.ones(1, device=dev)
    ms = timeit(lambda: ops.pit_mse_bwd(m, mix, srcs, res["best_perm"], res["out"], one))
    by = T * B * (2 * S + 1 + S) * F * 4
    print("pit_mse_bwd: %8.3f ms  %7.1f GB/s algorithmic (%.1f %% of 8 TB/s)" % (ms, by / ms / 1e6, by / ms / 1e6 / 80))


if __name__ == "__main__":
    main()
