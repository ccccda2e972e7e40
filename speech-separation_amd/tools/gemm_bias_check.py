#!/usr/bin/env python3
"""Diagnostic: rel-L2 error and SIGNED mean relative error (bias) of the fp32 GEMM kernel variants against fp64 on data without
cancellation (all-positive operands: every rounding / truncation bias adds up) and on gradient-like data (wide magnitude
range), N/N form, K = 7168 (the data-gradient shape) and K = 12800 (T/N weight-gradient shape)."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from sepkern import ops  # noqa: E402


def run(name, A, B, tA, tB, variants):
    a64 = A.double().t() if tA else A.double()
    b64 = B.double().t() if tB else B.double()
    ref = a64 @ b64
    M, N = ref.shape
    K = a64.shape[1]
    line = "%-34s K=%-6d" % (name, K)
    for v in variants:
        C = torch.empty(M, N, device="cuda")
        ops.gemm(A, B, C, M, N, K, A.shape[1], B.shape[1], N, transA=tA, transB=tB, variant=v)
        torch.cuda.synchronize()
        d = C.double() - ref
        rel = float(d.norm() / ref.norm())
        bias = float((d / ref.abs().clamp_min(1e-300)).mean())
        line += "   v%d relL2 %.2e bias %+.2e" % (v, rel, bias)
    print(line, flush=True)


def main():
    variants = [int(v) for v in sys.argv[1:]] or [8, 2, 9]
    g = torch.Generator(device="cuda").manual_seed(0)
    M, N = 2048, 1792
    for K in (1792, 7168, 12800):
        Ap = torch.rand(M, K, device="cuda", generator=g) + 0.5
        Bp = torch.rand(K, N, device="cuda", generator=g) + 0.5
        run("all positive U(0.5,1.5) NN", Ap, Bp, False, False, variants)
        An = torch.randn(M, K, device="cuda", generator=g)
        Bn = torch.randn(K, N, device="cuda", generator=g) * 0.03
        run("normal x 0.03 normal NN", An, Bn, False, False, variants)
        mag = torch.exp(torch.randn(M, K, device="cuda", generator=g) * 3.0 - 12.0)      # gradient-like: lognormal magnitudes
        Ag = torch.randn(M, K, device="cuda", generator=g).sign() * mag
        run("gradient-like (lognormal) NN", Ag, Bn, False, False, variants)
        At = torch.randn(K, M, device="cuda", generator=g).sign() * torch.exp(torch.randn(K, M, device="cuda", generator=g) * 3.0 - 12.0)
        Bt = torch.tanh(torch.randn(K, N, device="cuda", generator=g))
        run("gradient-like^T x tanh TN", At, Bt, True, False, variants)


if __name__ == "__main__":
    main()
