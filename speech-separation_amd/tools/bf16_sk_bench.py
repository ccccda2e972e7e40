#!/usr/bin/env python3
"""bf16 stream-K kernel, stand-alone: the step's main-stream shapes and 8192^3, NT and K-major forms (diagnostic).
SEPKERN_BF16_SK_STAGES=2 selects the first kernel (K steps of 64, two LDS stages)."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from sepkern import ops  # noqa: E402

SHAPES = [("proj NT", 12800, 7168, 1792, False, False), ("dgrad A row, B K-major", 12800, 1792, 7168, False, True),
          ("wgrad both K-major", 7168, 1792, 12800, True, True), ("8192^3 NT", 8192, 8192, 8192, False, False),
          ("8192^3 K-major B", 8192, 8192, 8192, False, True)]
for name, M, N, K, ak, bk in SHAPES:
    A = torch.randn((K, M) if ak else (M, K), device="cuda").bfloat16()
    B = torch.randn((K, N) if bk else (N, K), device="cuda").bfloat16()
    C = torch.empty(M, N, device="cuda")
    lda, ldb = A.shape[1], B.shape[1]
    for sk in (True, False):
        kw = dict(a_kmajor=ak, b_kmajor=bk, splitk=0, streamk=sk)
        for _ in range(2):
            ops.gemm_bf16_mm(A, B, C, M, N, K, lda, ldb, N, **kw)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            ops.gemm_bf16_mm(A, B, C, M, N, K, lda, ldb, N, **kw)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print("%-26s M=%6d N=%5d K=%6d %-9s %7.3f ms %7.1f TFLOP/s" % (name, M, N, K, "stream-K" if sk else "split-K", ms, 2.0 * M * N * K / ms / 1e9),
              flush=True)
