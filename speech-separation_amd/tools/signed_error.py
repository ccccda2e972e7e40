#!/usr/bin/env python3
"""SIGNED error of the split-product arithmetic against fp64, everywhere it runs (r05 verdict, weak #3).

The bf16 MFMA truncates the alignment of its addends towards minus infinity, so a product formed on it by the three-way split
carries a small DC offset -- far below its rel-L2 error, but coherent.  The kernels cancel it with sign phases (csrc/gemm.hip
SignPhase; csrc/lstm.hip, the two K halves of the split forward recurrence).  This module measures, for one kernel and form,
the MEAN SIGNED relative error and the rel-L2 error against an fp64 host evaluation; tests/test_gpu_signed_error.py gates
them, `python tools/signed_error.py` prints the table (profiles/r06_signed_error*.txt: product build vs builds without the
phases).

    python tools/signed_error.py [--quick] [--cache DIR]     (--cache: keep the fp64 recurrences between runs on two libraries)
"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def gemm_case(ops, form, M, N, K, positive, variant, splitk=1, batch=1, seed=0):
    """One product C = opA(A) opB(B) of the given form ("NT", "NN", "TN") on the device against fp64 on the host.
    positive: operands uniform in [0.5, 1.5) (every truncation pulls the same way) instead of N(0, 1).
    Returns dict(mean_signed, rel_l2, kernel): mean over elements of (C - ref) / (|A| |B|), i.e. relative to the size of the
    sum's terms (for positive operands that is the result itself), and ||C - ref|| / ||ref||."""
    from sepkern import _lib
    g = torch.Generator().manual_seed(1000 * seed + K + 7 * M + N)
    tA, tB = form[0] == "T", form[1] == "T"

    def draw(*shape):
        return (torch.rand(*shape, generator=g) + 0.5) if positive else torch.randn(*shape, generator=g)
    outs, refs, mags = [], [], []
    A = draw(batch, *((K, M) if tA else (M, K)))
    B = draw(batch, *((N, K) if tB else (K, N)))
    C = torch.empty(batch, M, N).cuda()
    Ad, Bd = A.cuda(), B.cuda()
    ops.gemm(Ad, Bd, C, M, N, K, A.shape[2], B.shape[2], N, transA=tA, transB=tB, variant=variant, splitk=splitk, batch=batch,
             sA=A.shape[1] * A.shape[2], sB=B.shape[1] * B.shape[2], sC=M * N, ws_tag="t_signed")
    torch.cuda.synchronize()
    kern = _lib.load().sk_gemm_last_kernel()
    for z in range(batch):
        a64 = A[z].double().t() if tA else A[z].double()
        b64 = B[z].double().t() if tB else B[z].double()
        refs.append(a64 @ b64)
        mags.append(a64.abs() @ b64.abs())
        outs.append(C[z].cpu().double())
    out, ref, mag = torch.stack(outs), torch.stack(refs), torch.stack(mags)
    d = out - ref
    return dict(mean_signed=float((d / mag).mean()), rel_l2=float(d.norm() / ref.norm()), kernel=int(kern))


def lstm_inputs(T, B, H, positive, seed=0):
    """(gx (T,B,2,4H) gate-interleaved, whh (2,4H,H) torch order, h0, c0 (2,B,H)).  positive: W_hh > 0 and a cell whose
    operating point keeps h in (0, 1) without saturating, so that every product h_k W_jk of a step is positive."""
    g = torch.Generator().manual_seed(77 * seed + H + T)
    k = 1.0 / H ** 0.5
    if not positive:
        gx = torch.randn(T, B, 2, 4 * H, generator=g) * 0.5
        whh = (torch.rand(2, 4 * H, H, generator=g) * 2 - 1) * k          # torch's own initial scale (nn.LSTM.reset_parameters)
        h0, c0 = torch.tanh(torch.randn(2, B, H, generator=g)), torch.randn(2, B, H, generator=g)
        return gx, whh, h0, c0
    # W_hh in [0, 0.1 k): the recurrent sums are ~ 1.5 mean(h) -- loop gain of the operating point below one, so the cell
    # settles at h ~ 0.3 instead of running into saturation (where no error would show)
    whh = torch.rand(2, 4 * H, H, generator=g) * k * 0.1
    gx = torch.randn(T, B, 2, H, 4, generator=g) * 0.5
    gx[..., 0] -= 0.45         # i, f, o: pre-activation ~ 0
    gx[..., 1] -= 0.45
    gx[..., 3] -= 0.45
    gx[..., 2] += 0.35         # g: ~ +0.8, tanh > 0
    h0 = torch.rand(2, B, H, generator=g) * 0.5 + 0.2
    c0 = torch.rand(2, B, H, generator=g) * 0.8 + 0.4
    return gx.reshape(T, B, 2, 4 * H), whh, h0, c0


def lstm_fp64(gx, whh, h0, c0):
    """The recurrence in fp64 on the host (equal lengths): y (T, B, 2H), c_T (2, B, H).  gx is gate-interleaved (element
    4u + g of a direction's 4H), whh in torch's gate-major row order; direction 1 runs from t = T - 1 down."""
    T, B, _, H4 = gx.shape
    H = H4 // 4
    # (H, 4, B) per (t, direction): the layout in which W (4H x H) @ h^T (H x B) comes out -- the fast form for a 32-column
    # fp64 product on the host BLAS (the row-vector form h @ W^T is 2-3 x slower there)
    gx64 = gx.double().view(T, B, 2, H, 4).permute(0, 2, 4, 3, 1).contiguous()      # (T, 2, 4, H, B)
    w = whh.double().contiguous()                                                   # (2, 4H, H), rows g H + u
    y = torch.empty(T, B, 2 * H, dtype=torch.float64)
    cT = torch.empty(2, B, H, dtype=torch.float64)
    for d in range(2):
        h, c = h0[d].double().t().contiguous(), c0[d].double().t().contiguous()     # (H, B)
        for s in range(T):
            t = T - 1 - s if d else s
            pre = (w[d] @ h).view(4, H, B) + gx64[t, d]
            i_, f_, g_, o_ = torch.sigmoid(pre[0]), torch.sigmoid(pre[1]), torch.tanh(pre[2]), torch.sigmoid(pre[3])
            c = f_ * c + i_ * g_
            h = o_ * torch.tanh(c)
            y[t, :, d * H:(d + 1) * H] = h.t()
        cT[d] = c.t()
    return y, cT


def lstm_case(ops, gx, whh, h0, c0, ref, bits, bf16=False):
    """The forward recurrence kernel (mode 1 | bits) against `ref` = lstm_fp64's result: signed and rel-L2 errors of y and c_T."""
    T, B, _, H4 = gx.shape
    H = H4 // 4
    lens = torch.full((B,), T, dtype=torch.int32).cuda()
    gg = gx.reshape(T * B, 8 * H).cuda()
    y, cs = torch.zeros(T * B, 2 * H).cuda(), torch.zeros(T * B, 2 * H).cuda()
    hn, cn = torch.zeros(2, B, H).cuda(), torch.zeros(2, B, H).cuda()
    ws = ops.lstm_fwd(gg, whh.cuda(), h0.cuda(), c0.cuda(), lens, y, gg, cs, hn, cn, T, B, H, 1 | bits, bf16=bf16)
    ops.lstm_status(ws)
    y_ref, c_ref = ref
    ey = y.cpu().double().view(T, B, 2 * H) - y_ref
    ec = cn.cpu().double() - c_ref
    return dict(y_mean_signed=float(ey.mean()), y_rel_l2=float(ey.norm() / y_ref.norm()),
                # the second half of the sequence alone: where an integrated offset would have built up
                y_late_mean_signed=float(ey[T // 2:].mean()),
                c_mean_signed=float(ec.mean()), c_rel_l2=float(ec.norm() / c_ref.norm()))


def lstm_bwd_fp64(gx, whh, h0, c0, dy):
    """Gradients of sum(y * dy) through the fp64 host recurrence (autograd): dgx (T, B, 2, 4H) gate-interleaved like gx -- the
    gradient of the gate pre-activations, what sk_lstm_bwd writes --, dh0, dc0 (2, B, H)."""
    g64 = gx.double().requires_grad_(True)
    h64, c64 = h0.double().requires_grad_(True), c0.double().requires_grad_(True)
    T, B, _, H4 = gx.shape
    H = H4 // 4
    gv = g64.view(T, B, 2, H, 4)
    w = whh.double()
    ys = []
    for d in range(2):
        h, c = h64[d], c64[d]
        wt = w[d].t().contiguous()                      # (H, 4H), columns g H + u
        out = [None] * T
        for s in range(T):
            t = T - 1 - s if d else s
            pre = (h @ wt).view(B, 4, H) + gv[t, :, d].permute(0, 2, 1)
            i_, f_, g_, o_ = torch.sigmoid(pre[:, 0]), torch.sigmoid(pre[:, 1]), torch.tanh(pre[:, 2]), torch.sigmoid(pre[:, 3])
            c = f_ * c + i_ * g_
            h = o_ * torch.tanh(c)
            out[t] = h
        ys.append(torch.stack(out))
    y = torch.cat(ys, 2)
    (y * dy.double()).sum().backward()
    return g64.grad.detach(), h64.grad.detach(), c64.grad.detach()


def lstm_bwd_case(ops, gx, whh, h0, c0, dy, ref, bits):
    """sk_lstm_bwd (mode 1 | bits) on the activations saved by the fp32-MFMA forward kernel, against lstm_bwd_fp64's result."""
    T, B, _, H4 = gx.shape
    H = H4 // 4
    lens = torch.full((B,), T, dtype=torch.int32).cuda()
    gg = gx.reshape(T * B, 8 * H).cuda()
    y, cs = torch.zeros(T * B, 2 * H).cuda(), torch.zeros(T * B, 2 * H).cuda()
    whh_d, c0_d = whh.cuda(), c0.cuda()
    ws = ops.lstm_fwd(gg, whh_d, h0.cuda(), c0_d, lens, y, gg, cs, None, None, T, B, H, 1 | ops.lstm_variant_bits(False, 1, True, False, False, 0))
    ops.lstm_status(ws)
    dh0, dc0 = torch.zeros(2, B, H).cuda(), torch.zeros(2, B, H).cuda()
    ws = ops.lstm_bwd(dy.reshape(T * B, 2 * H).cuda(), whh_d, gg, cs, c0_d, lens, gg, dh0, dc0, T, B, H, 1 | bits)
    ops.lstm_status(ws)
    dgx_ref, dh_ref, _ = ref
    eg = gg.cpu().double().view(T, B, 2, 4 * H) - dgx_ref
    eh = dh0.cpu().double() - dh_ref
    scale = float(dgx_ref.abs().mean())
    return dict(dgx_mean_signed=float(eg.mean()) / scale, dgx_rel_l2=float(eg.norm() / dgx_ref.norm()),
                dh0_mean_signed=float(eh.mean()) / float(dh_ref.abs().mean()), dh0_rel_l2=float(eh.norm() / dh_ref.norm()))


# what the training step launches (3 x 896, 32 x 400: R = 12800 rows), reduced in M and N -- the offset is per element
GEMM_CASES = (
    # name, form, M, N, K, variant, splitk, batch
    ("hosted dW_ih (T/N, K = 12800, 128x128 split kernel)", "TN", 512, 384, 12800, 2, 1, 1),
    ("hosted dW_ih, K in 5 slices", "TN", 512, 384, 12800, 2, 5, 1),
    ("hosted dW_hh (T/N, batch 2)", "TN", 512, 384, 12800, 2, 1, 2),
    ("unsplit T/N on the planes kernel", "TN", 512, 384, 12800, 9, 1, 1),
    ("projection (N/T, K = 1792, planes kernel)", "NT", 1024, 512, 1792, 9, 1, 1),
    ("projection on the 128x128 split kernel", "NT", 1024, 512, 1792, 2, 1, 1),
    ("layer-0 projection (N/T, K = 272)", "NT", 1024, 512, 272, 9, 1, 1),
    ("data gradient (N/N, K = 7168, planes kernel)", "NN", 1024, 512, 7168, 9, 1, 1),
    ("data gradient on the 128x128 split kernel", "NN", 1024, 512, 7168, 2, 1, 1),
)


def main():
    from sepkern import ops, _lib
    quick = "--quick" in sys.argv
    torch.set_num_threads(max(1, min(16, len(os.sched_getaffinity(0)))))
    print("library: %s" % _lib.library_info())
    print("GEMM: mean signed error / rel-L2 against fp64 -- split kernel | fp32-MFMA kernels (variant 8)")
    for name, form, M, N, K, variant, splitk, batch in GEMM_CASES:
        for positive in (True, False):
            r = gemm_case(ops, form, M, N, K, positive, variant, splitk, batch)
            m = gemm_case(ops, form, M, N, K, positive, 8, splitk, batch)
            print("  %-52s %-8s kernel %2d: %+.2e / %.2e | %+.2e / %.2e" % (
                name, "positive" if positive else "N(0,1)", r["kernel"], r["mean_signed"], r["rel_l2"], m["mean_signed"], m["rel_l2"]))
    T, B, H = (100 if quick else 400), 32, 896
    print("forward recurrence, T = %d, B = %d, H = %d against an fp64 host recurrence: y mean signed (all / second half) / rel-L2, "
          "c_T mean signed / rel-L2" % (T, B, H))
    cache = sys.argv[sys.argv.index("--cache") + 1] if "--cache" in sys.argv else None
    for positive in (False, True):
        inp = lstm_inputs(T, B, H, positive)
        f = os.path.join(cache, "lstm_fp64_%d_%d_%d_%d.pt" % (T, B, H, positive)) if cache else None
        if f and os.path.exists(f):
            ref = torch.load(f)
        else:
            ref = lstm_fp64(*inp)
            if f:
                torch.save(ref, f)
        print("  %s inputs: mean |y| %.3f, mean y %+.3f" % ("positive" if positive else "N(0,1)", float(ref[0].abs().mean()), float(ref[0].mean())))
        for name, bits, bf in (("split (shipped)", ops.lstm_variant_bits(False, 1, True, False, False, 0, split3=True), False),
                               ("fp32 MFMA", ops.lstm_variant_bits(False, 1, True, False, False, 0), False),
                               ("bf16 inputs", ops.lstm_variant_bits(False, 1, True, False, True, 0), True)):
            r = lstm_case(ops, *inp, ref, bits, bf16=bf)
            print("    %-16s y %+.2e (%+.2e) / %.2e   c_T %+.2e / %.2e" % (
                name, r["y_mean_signed"], r["y_late_mean_signed"], r["y_rel_l2"], r["c_mean_signed"], r["c_rel_l2"]))
        g = torch.Generator().manual_seed(5)
        dy = torch.randn(T, B, 2 * H, generator=g)
        dy = dy.abs() if positive else dy
        f = os.path.join(cache, "lstm_bwd_fp64_%d_%d_%d_%d.pt" % (T, B, H, positive)) if cache else None
        if f and os.path.exists(f):
            bref = torch.load(f)
        else:
            bref = lstm_bwd_fp64(*inp, dy)
            if f:
                torch.save(bref, f)
        for name, bits in (("bwd fp32 MFMA", ops.lstm_variant_bits(False, 1, poll_delay=31)),):
            r = lstm_bwd_case(ops, *inp, dy, bref, bits)
            print("    %-22s dgx %+.2e / %.2e (mean signed / mean |dgx|, rel-L2)   dh0 %+.2e / %.2e" % (
                name, r["dgx_mean_signed"], r["dgx_rel_l2"], r["dh0_mean_signed"], r["dh0_rel_l2"]))


if __name__ == "__main__":
    main()
