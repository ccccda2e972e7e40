#!/usr/bin/env python3
"""Per-shape timing of sk_gemm_f32 on the shapes of the 3x896, 32x400 training step (diagnostic)."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from sepkern import ops  # noqa: E402

H, R, F = 896, 12800, 257
SHAPES = [  # name, M, N, K, transA, transB, batch, splitk
    ("fwd  L1/2  NT x*Wih^T", R, 8 * H, 2 * H, False, True, 1, 1),
    ("fwd  L0    NT K=257", R, 8 * H, F, False, True, 1, 1),
    ("fwd  lin   NT N=514", R, 2 * F, 2 * H, False, True, 1, 1),
    ("dgrad L1/2 NN dgx*Wih", R, 2 * H, 8 * H, False, False, 1, 1),
    ("dgrad lin  NN K=514", R, 2 * H, 2 * F, False, False, 1, 1),
    ("wgrad Wih  TN", 8 * H, 2 * H, R, True, False, 1, 0),
    ("wgrad Wih  TN no split", 8 * H, 2 * H, R, True, False, 1, 1),
    ("wgrad Whh  TN batch2", 4 * H, H, R, True, False, 2, 0),
    ("wgrad Wih0 TN N=257", 8 * H, F, R, True, False, 1, 0),
    ("wgrad lin  TN M=514", 2 * F, 2 * H, R, True, False, 1, 0),
]


def bench_nt():
    """The bf16-operands-in-memory kernel (sk_gemm_bf16_nt) on the same products, every one in NT form."""
    S3 = 771
    shapes = [("fwd  L1/2", R, 8 * H, 2 * H, 1, 1), ("fwd  L0 K=257->320", R, 8 * H, 320, 1, 1), ("fwd  lin N=771", R, S3, 2 * H, 1, 1),
              ("dgrad L1/2", R, 2 * H, 8 * H, 1, 1), ("dgrad lin K=771->832", R, 2 * H, 832, 1, 1),
              ("wgrad Wih", 8 * H, 2 * H, R, 1, 0), ("wgrad Whh batch2", 4 * H, H, R, 2, 0), ("wgrad Wih0 N=257", 8 * H, F, R, 1, 0),
              ("wgrad lin M=771", S3, 2 * H, R, 1, 0), ("square 4096", 4096, 4096, 4096, 1, 1), ("square 8192", 8192, 8192, 8192, 1, 1)]
    for name, M, N, K, batch, sk in shapes:
        A = (torch.randn(M * batch, K, device="cuda")).bfloat16()
        B = (torch.randn(N * batch, K, device="cuda")).bfloat16()
        C = torch.empty(batch, M, N, device="cuda")
        used = ops.pick_splitk_bf16(M, N, K, batch) if sk == 0 else sk
        kw = dict(batch=batch, sA=M * K, sB=N * K, sC=M * N, splitk=used)
        for _ in range(2):
            ops.gemm_bf16_nt(A, B, C, M, N, K, K, K, N, **kw)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 10
        e0.record()
        for _ in range(n):
            ops.gemm_bf16_nt(A, B, C, M, N, K, K, K, N, **kw)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
        print("NT %-22s M=%6d N=%5d K=%6d b=%d splitk=%2d  %8.3f ms  %7.1f TFLOP/s" %
              (name, M, N, K, batch, used, ms, 2.0 * M * N * K * batch / ms / 1e9), flush=True)


def bench_km():
    """sk_gemm_bf16_mm: the backward products of the bf16 configuration on the row-major copies (K-major factors) next to
    the same products in NT form on transposed copies."""
    S3 = 771
    shapes = [("dgrad L1/2", R, 2 * H, 8 * H, 1, 1, False, True), ("dgrad lin K=771->832", R, 2 * H, 832, 1, 1, False, True),
              ("wgrad Wih", 8 * H, 2 * H, R, 1, 0, True, True), ("wgrad Whh batch2", 4 * H, H, R, 2, 0, True, True),
              ("wgrad Wih0 N=257->320", 8 * H, 320, R, 1, 0, True, True), ("wgrad lin M=771->832", 832, 2 * H, R, 1, 0, True, True),
              ("square 4096 TN", 4096, 4096, 4096, 1, 1, True, True), ("square 8192 TN", 8192, 8192, 8192, 1, 1, True, True),
              ("square 8192 NN", 8192, 8192, 8192, 1, 1, False, True)]
    if "--only" in sys.argv:
        shapes = [shapes[int(sys.argv[sys.argv.index("--only") + 1])]]
    for name, M, N, K, batch, sk, akm, bkm in shapes:
        used = ops.pick_splitk_bf16(M, N, K, batch) if sk == 0 else sk
        res = []
        for form in ("km", "nt"):
            ak, bk = (akm, bkm) if form == "km" else (False, False)
            A = torch.randn((batch * K, M) if ak else (batch * M, K), device="cuda").bfloat16()
            B = torch.randn((batch * K, N) if bk else (batch * N, K), device="cuda").bfloat16()
            C = torch.empty(batch, M, N, device="cuda")
            kw = dict(a_kmajor=ak, b_kmajor=bk, batch=batch, sA=M * K, sB=N * K, sC=M * N, splitk=used)
            lda, ldb = (M if ak else K), (N if bk else K)
            for _ in range(2):
                ops.gemm_bf16_mm(A, B, C, M, N, K, lda, ldb, N, **kw)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n = 10
            e0.record()
            for _ in range(n):
                ops.gemm_bf16_mm(A, B, C, M, N, K, lda, ldb, N, **kw)
            e1.record()
            torch.cuda.synchronize()
            res.append(e0.elapsed_time(e1) / n)
        print("%-24s M=%6d N=%5d K=%6d b=%d splitk=%2d  K-major %7.3f ms %7.1f TFLOP/s | NT %7.3f ms %7.1f TFLOP/s" %
              (name, M, N, K, batch, used, res[0], 2.0 * M * N * K * batch / res[0] / 1e9, res[1], 2.0 * M * N * K * batch / res[1] / 1e9),
              flush=True)


def main():
    if "--nt" in sys.argv:
        return bench_nt()
    if "--km" in sys.argv:
        return bench_km()
    bf16 = "--bf16" in sys.argv
    variant = int(sys.argv[sys.argv.index("--variant") + 1]) if "--variant" in sys.argv else 0   # sk_gemm_f32_splitk's variant
    shapes = SHAPES
    if "--shape" in sys.argv:                       # --shape M,N,K,tA,tB[,splitk]  (may repeat)
        shapes = []
        for i, a in enumerate(sys.argv):
            if a == "--shape":
                v = [int(x) for x in sys.argv[i + 1].split(",")]
                shapes.append(("custom", v[0], v[1], v[2], bool(v[3]), bool(v[4]), 1, v[5] if len(v) > 5 else 1))
    for name, M, N, K, tA, tB, batch, sk in shapes:
        A = torch.randn((K, M * batch) if tA else (M, K), device="cuda")
        B = torch.randn((N, K) if tB else (K, N * batch), device="cuda")
        C = torch.empty(batch, M, N, device="cuda")
        lda, ldb = A.shape[1], B.shape[1]
        kw = dict(transA=tA, transB=tB, batch=batch, sA=M if batch > 1 else 0, sB=N if batch > 1 else 0, sC=M * N, splitk=sk, bf16=bf16, variant=variant)
        used = ops.pick_splitk(M, N, K, batch) if sk == 0 else sk
        for _ in range(2):
            ops.gemm(A, B, C, M, N, K, lda, ldb, N, **kw)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 10
        e0.record()
        for _ in range(n):
            ops.gemm(A, B, C, M, N, K, lda, ldb, N, **kw)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
        print("%-26s M=%6d N=%5d K=%6d b=%d splitk=%2d  %8.3f ms  %7.1f TFLOP/s" %
              (name, M, N, K, batch, used, ms, 2.0 * M * N * K * batch / ms / 1e9), flush=True)


if __name__ == "__main__":
    main()
