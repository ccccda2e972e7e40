#!/usr/bin/env python3
"""Diagnostic: the data-gradient product of the top BLSTM layer on the REAL operands of one full-size training step (dgx of layer 2 as
the backward recurrence wrote it, the gate-interleaved W_ih), by the GEMM kernel variants, against fp64: rel-L2 error, the share of
exact zeros / tiny values in dgx, and the error restricted to rows by magnitude."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "speech-separation_amd"), os.path.join(ROOT, "speech-separation_amd", "archs")):
    sys.path.insert(0, p)
import uPIT  # noqa: E402
from sepkern import ops  # noqa: E402


def main():
    H, L, S, B, T = 896, 3, 2, 32, 400
    torch.manual_seed(H + L)
    rng = np.random.default_rng(H)
    model = uPIT.SepDNN(0, num_spk=str(S), hidden_dim=str(H), num_layers=str(L))
    model.cuda()
    model.train()
    lens = sorted([int(v) for v in rng.integers(T // 2, T + 1, B)])
    lens[-1] = T
    samples = []
    for n in lens:
        d = {"mix": np.abs(rng.standard_normal((n, 257))).astype(np.float32)}
        for s in range(S):
            d["source%d" % (s + 1)] = np.abs(rng.standard_normal((n, 257))).astype(np.float32) * 0.6
        samples.append(d)
    captured = []
    real = ops.gemm

    def spy(A, Bm, C, M, N, K, lda, ldb, ldc, **kw):
        if not kw.get("transA") and not kw.get("transB") and K == 8 * H and N == 2 * H and not captured:
            captured.append((A[:M].clone(), Bm.clone(), M, N, K, lda, ldb))
        return real(A, Bm, C, M, N, K, lda, ldb, ldc, **kw)
    ops.gemm = spy
    loss, norm = uPIT.compute_loss(model, 0, uPIT.Collator("mix")(samples))
    loss.backward()
    torch.cuda.synchronize()
    ops.gemm = real
    A, Bm, M, N, K, lda, ldb = captured[0]
    a = A[:, :K]
    print("dgx of the top layer: %d x %d, |x| max %.3e, rms %.3e, exact zeros %.2f %%, |x| < 1e-30: %.2f %%, < 1e-20: %.2f %%, < 1e-12: %.2f %%" % (
        M, K, float(a.abs().max()), float(a.double().pow(2).mean().sqrt()), 100 * float((a == 0).float().mean()),
        100 * float((a.abs() < 1e-30).float().mean()), 100 * float((a.abs() < 1e-20).float().mean()), 100 * float((a.abs() < 1e-12).float().mean())))
    ref = a.double() @ Bm.double()
    for v in (8, 1, 2, 9):
        C = torch.empty(M, N, device="cuda")
        ops.gemm(A, Bm, C, M, N, K, lda, ldb, N, variant=v)
        torch.cuda.synchronize()
        d = C.double() - ref
        rown = ref.norm(dim=1)
        big = rown > rown.median()
        print("variant %d: rel-L2 %.3e   rows above the median norm %.3e, below %.3e   max |err| / max |ref| %.3e" % (
            v, float(d.norm() / ref.norm()), float(d[big].norm() / ref[big].norm()), float(d[~big].norm() / ref[~big].norm()),
            float(d.abs().max() / ref.abs().max())))


if __name__ == "__main__":
    main()
