#!/usr/bin/env python3
"""Diagnostic: how much GEMM work hides behind a persistent BLSTM launch when both are resident on the same
CUs (the recurrence leaves the matrix pipe idle during its hand-offs, ~120 VGPRs per SIMD lane and >60 KB of LDS
free).  Times: recurrence alone, N GEMMs alone, both concurrently on two streams."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from sepkern import ops  # noqa: E402

T, B, H = 400, 32, 896


def main():
    bwd = "--bwd" in sys.argv
    torch.manual_seed(0)
    dev = "cuda"
    gx = torch.randn(T, B, 2, 4 * H, device=dev) * 0.5
    whh = torch.randn(2, 4 * H, H, device=dev) / 30
    h0, c0 = torch.randn(2, B, H, device=dev), torch.randn(2, B, H, device=dev)
    lens = torch.full((B,), T, dtype=torch.int32, device=dev)
    y, cs = torch.empty(T, B, 2 * H, device=dev), torch.empty(T, B, 2, H, device=dev)
    g = gx.clone()
    ops.lstm_fwd(g, whh, h0, c0, lens, y, g, cs, None, None, T, B, H, 1)
    dy = torch.randn(T, B, 2 * H, device=dev)
    g2 = g.clone()
    # GEMM work: chunks of the next layer's input projection (1600 x 7168 x 896), NG of them
    M, N, K = 1600, 8 * H, H
    A = torch.randn(M, 2 * H, device=dev)
    W = torch.randn(N, 2 * H, device=dev)
    C = torch.empty(M, N, device=dev)
    NG = 16

    def rec():
        if bwd:
            g2.copy_(g)
            return ops.lstm_bwd(dy, whh, g2, cs, c0, lens, g2, None, None, T, B, H, 1)
        return ops.lstm_fwd(gx, whh, h0, c0, lens, y, None, None, None, None, T, B, H, 1)

    def gemms():
        for _ in range(NG):
            ops.gemm(A, W, C, M, N, K, 2 * H, 2 * H, N, transB=True)

    side = torch.cuda.Stream()
    main_s = torch.cuda.current_stream()

    def timed(fn, n=3):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n

    def both():
        side.wait_stream(main_s)
        ws = rec()
        with torch.cuda.stream(side):
            gemms()
        main_s.wait_stream(side)
        return ws

    if "--trace" in sys.argv:
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True)
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(NG)]
        er = torch.cuda.Event(enable_timing=True)
        e0.record()
        side.wait_stream(main_s)
        rec()
        er.record()
        with torch.cuda.stream(side):
            for i in range(NG):
                ops.gemm(A, W, C, M, N, K, 2 * H, 2 * H, N, transB=True)
                evs[i].record()
        main_s.wait_stream(side)
        torch.cuda.synchronize()
        print("recurrence done at %.3f ms; GEMM completions: %s" % (e0.elapsed_time(er), " ".join("%.2f" % e0.elapsed_time(e) for e in evs)))
    t_rec = timed(rec)
    t_gemm = timed(gemms)
    t_both = timed(both)
    ops.lstm_status(both())
    fl = 2.0 * M * N * K * NG
    print("%s recurrence alone %.3f ms | %d GEMMs alone %.3f ms (%.1f TFLOP/s) | concurrent %.3f ms  -> hidden %.3f ms of %.3f"
          % ("bwd" if bwd else "fwd", t_rec, NG, t_gemm, fl / t_gemm / 1e9, t_both, t_rec + t_gemm - t_both, t_gemm))


if __name__ == "__main__":
    main()
