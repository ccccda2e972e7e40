#!/usr/bin/env python3
"""Diagnostic: error of the fp32 GEMM kernel variants against an fp64 product, relative to sum |a||b| (the scale of an fp32
accumulation's rounding), in the N/T, N/N and T/N forms incl. ragged tile edges.  usage: gemm_check.py [variant ...]  (default 8 2 7 9)"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from sepkern import ops  # noqa: E402


def main():
    variants = [int(v) for v in sys.argv[1:]] or [8, 2, 7, 9]
    torch.manual_seed(0)
    worst = 0.0
    for (M, N, K) in [(256, 128, 16), (256, 128, 64), (512, 384, 1792), (300, 200, 48), (12800, 1792, 1792), (12800, 516, 1792),
                      (1000, 1792, 3584), (260, 132, 32), (7168, 1792, 3200)]:
        for form, tA, tB in (("NT", False, True), ("NN", False, False), ("TN", True, False)):
            A = torch.randn((K, M) if tA else (M, K), device="cuda")
            B = torch.randn((N, K) if tB else (K, N), device="cuda")
            bias = torch.randn(N, device="cuda")
            a64 = A.double().t() if tA else A.double()
            b64 = B.double().t() if tB else B.double()
            ref = a64 @ b64 + bias.double()
            mag = a64.abs() @ b64.abs()
            line = "%s M=%-6d N=%-5d K=%-5d " % (form, M, N, K)
            for v in variants:
                C = torch.full((M, N), float("nan"), device="cuda")
                ops.gemm(A, B, C, M, N, K, A.shape[1], B.shape[1], N, transA=tA, transB=tB, bias=bias, variant=v)
                torch.cuda.synchronize()
                bad = bool(torch.isnan(C).any())
                e = float(((C.double() - ref).abs() / mag).max())
                worst = max(worst, e if not bad else 1.0)
                line += " v%d %.2e%s" % (v, e, " NaN!" if bad else "")
            print(line, flush=True)
    print("worst %.3e" % worst)
    return 0 if worst < 1e-5 else 1


if __name__ == "__main__":
    sys.exit(main())
