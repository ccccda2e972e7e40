"""Parity of every libsepkern kernel against the CPU oracle, through the C ABI (ctypes).

Run on the MI355X box:  python -m pytest tests -m gpu -x -q
Tolerances: fp32 kernels vs an fp64/fp32 CPU evaluation of the same formula; the only
differences are summation order (MFMA k-ordered fmaf chains, wave reductions) and libm.
"""
import itertools

import numpy as np
import pytest
import torch

from oracle import stft as OS
from oracle import upit as OU

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need an MI355X (torch.cuda.is_available() is False)")
    from sepkern import ops as _ops
    return _ops


def dev(t):
    return t.cuda()


# ------------------------------------------------------------------------------------ GEMM
@pytest.mark.parametrize("M,N,K,tA,tB", [
    (300, 200, 257, False, True),      # layer-0 input projection shape class (K=257: unaligned rows)
    (256, 256, 64, False, False), (256, 256, 64, True, False), (256, 256, 64, False, True), (256, 256, 64, True, True),
    (130, 514, 96, False, True), (1000, 72, 1, False, False), (129, 131, 300, True, False), (64, 1792, 514, False, False),
    # shapes the LDS-DMA kernel takes (aligned rows, K % 16 == 0), with ragged tile edges in M and N
    (300, 260, 512, False, True), (300, 260, 528, False, False), (132, 516, 1024, True, False), (1000, 772, 1792, False, True),
    (517, 1792, 3584, False, False), (1028, 132, 2048, True, False), (1, 4, 16, False, True), (4, 4, 16, True, False),
])
def test_gemm_matches_fp64(ops, M, N, K, tA, tB):
    g = torch.Generator().manual_seed(M * 7 + N)
    A = torch.randn((K, M) if tA else (M, K), generator=g)
    B = torch.randn((N, K) if tB else (K, N), generator=g)
    bias = torch.randn(N, generator=g)
    ref = (A.double().t() if tA else A.double()) @ (B.double().t() if tB else B.double()) + bias.double()
    C = torch.full((M, N), float("nan")).cuda()
    ops.gemm(dev(A), dev(B), C, M, N, K, A.shape[1], B.shape[1], N, transA=tA, transB=tB, bias=dev(bias))
    torch.cuda.synchronize()
    np.testing.assert_allclose(C.cpu().numpy(), ref.numpy(), atol=2e-5 * np.sqrt(K) * 4, rtol=1e-5)


@pytest.mark.parametrize("M,N,K,tA,tB", [(300, 260, 512, False, True), (300, 260, 528, False, False), (132, 516, 1024, True, False),
                                         (1000, 772, 1792, False, True), (517, 1792, 3584, False, False), (1028, 132, 2048, True, False),
                                         (4, 4, 16, True, False)])
def test_gemm_exact_bf16_split_is_an_fp32_product(ops, M, N, K, tA, tB):
    """variant 2 of sk_gemm_f32_splitk: both fp32 operands cut exactly into three bf16 pieces (|mid| <= 2^-8 |x|, |lo| <= 2^-16 |x|);
    the six piece products per element pair of relative size >= 2^-16 (each exact) are added on the bf16 matrix pipe into fp32
    accumulators, the three of size <= 2^-24 -- together at most 2^-23 |a||b|, one ulp of the product in the worst case -- are
    not formed.  On SUMS that is an fp32 GEMM in another summation order: it meets the fp32 kernel's tolerance against fp64, and
    its error is not larger than the fp32-MFMA kernel's on the same operands (wide dynamic range: elements scaled by
    2^-20 .. 2^20, so every piece carries weight)."""
    g = torch.Generator().manual_seed(M * 13 + N)
    scale = lambda shape: torch.exp2(torch.randint(-20, 21, shape, generator=g).float())
    A = torch.randn((K, M) if tA else (M, K), generator=g)
    B = torch.randn((N, K) if tB else (K, N), generator=g)
    A = A * scale(A.shape)                     # element-wise: terms of very different size meet in every sum
    B = B * scale(B.shape)
    bias = torch.randn(N, generator=g)
    ref = (A.double().t() if tA else A.double()) @ (B.double().t() if tB else B.double()) + bias.double()
    mag = (A.double().abs().t() if tA else A.double().abs()) @ (B.double().abs().t() if tB else B.double().abs())
    outs = []
    for variant in (2, 0):
        C = torch.full((M, N), float("nan")).cuda()
        ops.gemm(dev(A), dev(B), C, M, N, K, A.shape[1], B.shape[1], N, transA=tA, transB=tB, bias=dev(bias), variant=variant)
        torch.cuda.synchronize()
        outs.append(C.cpu().double())
    err_split = ((outs[0] - ref).abs() / mag.clamp_min(1e-30)).max().item()
    err_f32 = ((outs[1] - ref).abs() / mag.clamp_min(1e-30)).max().item()
    # a K-term fp32 sum: errors of a few ulp relative to sum |a||b|
    assert err_split <= 2.0 ** -23 * 4 * np.sqrt(K), (err_split, err_f32)
    assert err_split <= max(2.0 * err_f32, 2.0 ** -22), (err_split, err_f32)


def test_gemm_exact_bf16_split_pieces_reassemble_single_products(ops):
    """K = 16 with ONE non-zero term per output: C[m, n] = a[m] * b[n] must then be the fp32 product to within 2 ulp = 2^-22 (the
    six piece products formed miss the exact 48-bit product by the three smallest, <= 2^-23 of it together in the worst case --
    where an fp32 FMA would be exact to half an ulp --, and are rounded once at each accumulation), and exact whenever both
    factors have <= 16 significant bits (nothing is left out then)."""
    g = torch.Generator().manual_seed(99)
    M, N, K = 128, 128, 16
    a = torch.randn(M, generator=g) * torch.exp2(torch.randint(-30, 31, (M,), generator=g).float())
    b = torch.randn(N, generator=g) * torch.exp2(torch.randint(-30, 31, (N,), generator=g).float())
    A, B = torch.zeros(M, K), torch.zeros(N, K)
    A[:, 5], B[:, 5] = a, b
    C = torch.empty(M, N).cuda()
    ops.gemm(dev(A), dev(B), C, M, N, K, K, K, N, transB=True, variant=2)
    exact = a.double()[:, None] * b.double()[None, :]
    rel = ((C.cpu().double() - exact).abs() / exact.abs()).max().item()
    assert rel <= 2.0 ** -22, rel
    # powers of two times small integers: representable products must come out exact
    ai = torch.randint(-255, 256, (M,), generator=g).float() * 2.0 ** 7
    bi = torch.randint(-255, 256, (N,), generator=g).float() * 2.0 ** -9
    A[:, 5], B[:, 5] = ai, bi
    ops.gemm(dev(A), dev(B), C, M, N, K, K, K, N, transB=True, variant=2)
    assert torch.equal(C.cpu(), ai[:, None] * bi[None, :])


@pytest.mark.parametrize("M,N,K,tA,tB", [
    (300, 200, 257, False, True), (256, 256, 64, False, False), (256, 256, 64, True, False), (256, 256, 64, False, True),
    (256, 256, 64, True, True), (130, 514, 96, False, True), (1000, 72, 1, False, False), (129, 131, 300, True, False),
    (64, 1792, 514, False, False), (257, 384, 1000, True, False),
])
def test_gemm_bf16_equals_fp32_product_of_rounded_operands(ops, M, N, K, tA, tB):
    """bf16 x bf16 products are exact in fp32, so the bf16-input kernel must agree with an fp64 product of the
    bf16-ROUNDED operands to fp32 summation error -- the same tolerance as the fp32 kernel (BASELINE configs[3])."""
    g = torch.Generator().manual_seed(M * 11 + N)
    A = torch.randn((K, M) if tA else (M, K), generator=g)
    B = torch.randn((N, K) if tB else (K, N), generator=g)
    bias = torch.randn(N, generator=g)
    Ar, Br = A.bfloat16().double(), B.bfloat16().double()        # torch rounds to nearest even, like v_cvt_pk_bf16_f32
    ref = (Ar.t() if tA else Ar) @ (Br.t() if tB else Br) + bias.double()
    C = torch.full((M, N), float("nan")).cuda()
    ops.gemm(dev(A), dev(B), C, M, N, K, A.shape[1], B.shape[1], N, transA=tA, transB=tB, bias=dev(bias), bf16=True)
    torch.cuda.synchronize()
    np.testing.assert_allclose(C.cpu().numpy(), ref.numpy(), atol=2e-5 * np.sqrt(K) * 4, rtol=1e-5)


def test_gemm_bf16_splitk_batch_accumulate(ops):
    g = torch.Generator().manual_seed(77)
    K, M, N = 3000, 140, 257
    A, B = torch.randn(K, 2 * M, generator=g), torch.randn(K, 2 * N, generator=g)
    bias, C0 = torch.randn(N, generator=g), torch.randn(2, M, N, generator=g)
    Ar, Br = A.bfloat16().double(), B.bfloat16().double()
    ref = torch.stack([Ar[:, d * M:(d + 1) * M].t() @ Br[:, d * N:(d + 1) * N] for d in range(2)]) + bias.double() + C0.double()
    outs = []
    for _ in range(2):
        C = dev(C0.clone())
        ops.gemm(dev(A), dev(B), C, M, N, K, 2 * M, 2 * N, N, transA=True, bias=dev(bias), accumulate=True, batch=2,
                 sA=M, sB=N, sC=M * N, splitk=3, bf16=True)
        outs.append(C.cpu())
    np.testing.assert_allclose(outs[0].numpy(), ref.numpy(), atol=2e-5 * np.sqrt(K) * 4, rtol=1e-5)
    assert torch.equal(outs[0], outs[1])


def test_gemm_accumulate_sigmoid_batch_and_ld(ops):
    g = torch.Generator().manual_seed(5)
    # batch of 2 TN products out of strided storage: A (K, 2*M) and B (K, 2*N), like dW_hh
    K, M, N = 70, 40, 24
    A = torch.randn(K, 2 * M, generator=g)
    B = torch.randn(K, 2 * N, generator=g)
    C0 = torch.randn(2, M, N, generator=g)
    C = dev(C0.clone())
    ops.gemm(dev(A), dev(B), C, M, N, K, 2 * M, 2 * N, N, transA=True, accumulate=True, batch=2, sA=M, sB=N, sC=M * N)
    ref = torch.stack([A[:, d * M:(d + 1) * M].double().t() @ B[:, d * N:(d + 1) * N].double() for d in range(2)]) + C0.double()
    np.testing.assert_allclose(C.cpu().numpy(), ref.numpy(), atol=2e-4, rtol=1e-5)
    # sigmoid epilogue
    X, W, b = torch.randn(50, 30, generator=g), torch.randn(20, 30, generator=g), torch.randn(20, generator=g)
    out = torch.empty(50, 20).cuda()
    ops.gemm(dev(X), dev(W), out, 50, 20, 30, 30, 30, 20, transB=True, bias=dev(b), act=1)
    np.testing.assert_allclose(out.cpu().numpy(), torch.sigmoid(X.double() @ W.double().t() + b.double()).numpy(), atol=2e-6)


@pytest.mark.parametrize("M,N,K,S", [(300, 260, 5000, 5), (140, 257, 3000, 3), (129, 64, 4096, 0), (132, 64, 4096, 0),
                                     (300, 260, 4096, 4), (140, 260, 3008, 1)])
def test_gemm_splitk_is_deterministic_and_matches_fp64(ops, M, N, K, S):
    g = torch.Generator().manual_seed(K)
    A, B = torch.randn(K, 2 * M, generator=g), torch.randn(K, 2 * N, generator=g)     # weight-gradient shape (TN)
    bias, C0 = torch.randn(N, generator=g), torch.randn(2, M, N, generator=g)
    ref = torch.stack([A[:, d * M:(d + 1) * M].double().t() @ B[:, d * N:(d + 1) * N].double() for d in range(2)]) \
        + bias.double() + C0.double()
    outs = []
    for _ in range(2):
        C = dev(C0.clone())
        ops.gemm(dev(A), dev(B), C, M, N, K, 2 * M, 2 * N, N, transA=True, bias=dev(bias), accumulate=True, batch=2,
                 sA=M, sB=N, sC=M * N, splitk=S)
        outs.append(C.cpu())
    np.testing.assert_allclose(outs[0].numpy(), ref.numpy(), atol=2e-5 * np.sqrt(K) * 4, rtol=1e-5)
    assert torch.equal(outs[0], outs[1])                     # fixed-order slab reduction: bitwise reproducible


@pytest.mark.parametrize("variant", [3, 4])
@pytest.mark.parametrize("M,N,K,tA,tB,S", [(300, 260, 512, False, True, 1), (700, 260, 528, False, False, 1), (516, 132, 1024, True, False, 1),
                                            (1000, 772, 1792, False, True, 1), (517, 1792, 3584, False, False, 2),
                                            (1028, 132, 4096, True, False, 4), (256, 128, 16, False, True, 1),
                                            (1030, 1796, 528, False, False, 1), (772, 516, 1040, True, False, 1), (256, 256, 32, False, True, 1)])
def test_gemm_dma_kernels_forced(ops, variant, M, N, K, tA, tB, S):
    """sk_gemm_f32_splitk variant 3 (the LDS-DMA kernel wherever it applies; two LDS stages for N/T, three for N/N and T/N),
    4 (256 x 128 block tiles) and 5 (256 x 256 block tiles, unsplit products of at least one tile; r03): ragged edges in M
    and N, bias, accumulate, in-kernel split-K; against fp64 and run-to-run identical."""
    g = torch.Generator().manual_seed(M + 3 * N + variant)
    A = torch.randn((K, M) if tA else (M, K), generator=g)
    B = torch.randn((N, K) if tB else (K, N), generator=g)
    bias, C0 = torch.randn(N, generator=g), torch.randn(M, N, generator=g)
    ref = (A.double().t() if tA else A.double()) @ (B.double().t() if tB else B.double()) + bias.double() + C0.double()
    outs = []
    for _ in range(2):
        C = dev(C0.clone())
        ops.gemm(dev(A), dev(B), C, M, N, K, A.shape[1], B.shape[1], N, transA=tA, transB=tB, bias=dev(bias), accumulate=True,
                 splitk=S, variant=variant)
        torch.cuda.synchronize()
        outs.append(C.cpu())
    np.testing.assert_allclose(outs[0].numpy(), ref.numpy(), atol=2e-5 * np.sqrt(K) * 4, rtol=1e-5)
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("M,N,K,tA,tB", [(1000, 772, 1792, False, True),      # 16 tiles, no whole round: every tile in 16 pieces
                                         (4352, 4096, 256, False, True),      # 272 tiles: one round + 16 tiles cut at every K step
                                         (5000, 3800, 528, False, False),     # 300 tiles, ragged edges, ranges that span two tiles
                                         (3800, 5000, 528, True, False),      # the T/N form
                                         (4096, 4096, 64, False, True),       # a whole number of rounds: no cut at all
                                         (300, 260, 64, False, True)])        # remainder too short to cut: the plain 256 x 256 kernel
def test_gemm_stream_k_kernel(ops, M, N, K, tA, tB):
    """sk_gemm_f32_splitk variant 6 (r03): the persistent 256 x 256-tile kernel with a stream-K cut of the last partial round.
    Against fp64 with bias and accumulate, run-to-run identical (pieces are added in workgroup order), counters left zeroed."""
    from sepkern import _lib
    g = torch.Generator().manual_seed(M + 3 * N)
    A = torch.randn((K, M) if tA else (M, K), generator=g)
    B = torch.randn((N, K) if tB else (K, N), generator=g)
    bias, C0 = torch.randn(N, generator=g), torch.randn(M, N, generator=g)
    ref = (A.double().t() if tA else A.double()) @ (B.double().t() if tB else B.double()) + bias.double() + C0.double()
    outs = []
    for _ in range(2):
        C = dev(C0.clone())
        ops.gemm(dev(A), dev(B), C, M, N, K, A.shape[1], B.shape[1], N, transA=tA, transB=tB, bias=dev(bias), accumulate=True, variant=6,
                 ws_tag="t_sk")
        torch.cuda.synchronize()
        outs.append(C.cpu())
    np.testing.assert_allclose(outs[0].numpy(), ref.numpy(), atol=2e-5 * np.sqrt(K) * 4, rtol=1e-5)
    assert torch.equal(outs[0], outs[1])
    ws = ops.workspace(_lib.load().sk_gemm_streamk_workspace_bytes(), "t_sk_sk")
    assert int(ws[:65536].max()) == 0
    if M == 5000:      # the other epilogue: sigmoid(product + bias), no accumulate, rows of the result 8 floats apart from N
        Cw = torch.full((M, N + 8), 7.0).cuda()
        ops.gemm(dev(A), dev(B), Cw, M, N, K, A.shape[1], B.shape[1], N + 8, transA=tA, transB=tB, bias=dev(bias), act=1, variant=6,
                 ws_tag="t_sk")
        want = torch.sigmoid(ref - C0.double())
        assert float((Cw[:, :N].cpu().double() - want).abs().max()) < 1e-4      # fp32 pre-activations of magnitude sqrt(K)
        assert bool((Cw[:, N:] == 7).all())


@pytest.mark.parametrize("variant", [2, 9])
@pytest.mark.parametrize("M,N,K,tA,tB", [(1400, 1300, 1792, False, True), (1030, 772, 3584, False, False), (1028, 516, 2048, True, False),
                                         (256, 128, 16, False, True), (4352, 4096, 256, False, False), (300, 260, 64, True, False),
                                         (260, 132, 80, False, False), (260, 132, 112, True, False), (516, 260, 96, False, True)])
def test_gemm_split_kernels_of_the_large_products(ops, variant, M, N, K, tA, tB):
    """sk_gemm_f32_splitk variants 2 (128 x 128 tiles, every wave splits the fragments it reads) and 9 (256 x 128, the split done
    once per element while the tile is staged; K-major operands read back by ds_read_b64_tr_b16) in the N/T, N/N and T/N forms
    with ragged tile edges: against fp64 with bias, accumulate and the sigmoid epilogue; run-to-run identical; error not above the
    fp32-MFMA kernels' (variant 8); the two are bit for bit equal: same pieces, same products, same K order, same sign phases
    (csrc/gemm.hip SignPhase; tests/test_gpu_signed_error.py).  (Variant 7, the stream-K split form, was retired in r06.)"""
    g = torch.Generator().manual_seed(M + 3 * N + variant)
    A = torch.randn((K, M) if tA else (M, K), generator=g)
    B = torch.randn((N, K) if tB else (K, N), generator=g)
    bias, C0 = torch.randn(N, generator=g), torch.randn(M, N, generator=g)
    a64 = A.double().t() if tA else A.double()
    b64 = B.double().t() if tB else B.double()
    ref = a64 @ b64 + bias.double() + C0.double()
    mag = a64.abs() @ b64.abs()

    def run(v, **kw):
        C = dev(C0.clone())
        ops.gemm(dev(A), dev(B), C, M, N, K, A.shape[1], B.shape[1], N, transA=tA, transB=tB, bias=dev(bias), accumulate=True,
                 variant=v, ws_tag="t_sp", **kw)
        torch.cuda.synchronize()
        return C.cpu()
    out, again, mfma = run(variant), run(variant), run(8)
    assert torch.equal(out, again)
    err = float(((out.double() - ref).abs() / mag).max())
    err_mfma = float(((mfma.double() - ref).abs() / mag).max())
    assert err <= max(1.25 * err_mfma, 2.0 ** -22), (err, err_mfma)
    if variant == 9:
        assert torch.equal(out, run(2))
    Cw = torch.full((M, N + 8), 7.0).cuda()
    ops.gemm(dev(A), dev(B), Cw, M, N, K, A.shape[1], B.shape[1], N + 8, transA=tA, transB=tB, bias=dev(bias), act=1, variant=variant,
             ws_tag="t_sp")
    want = torch.sigmoid(ref - C0.double())
    assert float((Cw[:, :N].cpu().double() - want).abs().max()) < 1e-4
    assert bool((Cw[:, N:] == 7).all())


def test_gemm_splitk_workspace_from_a_c_caller(ops):
    """A split-K workspace that did NOT come zero-filled (a C caller's own allocation): sk_gemm_workspace_init zeroes the
    ticket counters at its head once; launches then leave them zeroed (two launches in a row give the same, right result)."""
    import ctypes as C
    from sepkern import _lib
    g = torch.Generator().manual_seed(31)
    M, N, K, S = 260, 132, 2048, 4
    A, B = dev(torch.randn(K, M, generator=g)), dev(torch.randn(K, N, generator=g))
    ref = A.cpu().double().t() @ B.cpu().double()
    nbytes = _lib.load().sk_gemm_workspace_bytes(M, N, 1, S)
    ws = torch.full((nbytes,), 0xA5, dtype=torch.uint8, device="cuda")                     # garbage, counters included
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = lambda t: C.c_void_p(t.data_ptr())
    _lib.call("sk_gemm_workspace_init", p(ws), st)
    for _ in range(2):
        out = torch.full((M, N), float("nan")).cuda()
        _lib.call("sk_gemm_f32_splitk", p(A), p(B), p(out), None, M, N, K, M, N, N, 1, 0, 0, 0, 1, 0, 0, 0, 0, S, p(ws), 0, st)
        torch.cuda.synchronize()
        np.testing.assert_allclose(out.cpu().numpy(), ref.numpy(), atol=2e-5 * np.sqrt(K) * 4, rtol=1e-5)
    assert int(ws[:65536].max()) == 0                                                          # counters back at zero


# ------------------------------------------------------------------------------------ STFT / iSTFT
def _sig(n, seed):
    rng = np.random.default_rng(seed)
    return (rng.standard_normal(n) * 0.2).astype(np.float32)


@pytest.mark.parametrize("layout", ["TF", "FT"])
def test_stft_matches_oracle(ops, layout):
    ns = [51072, 24000, 4097, 700, 257]
    ys = [_sig(n, n) for n in ns]
    outs = ops.stft_batch([torch.from_numpy(y).cuda() for y in ys], want_complex=True, layout=layout)
    mags = ops.stft_batch([torch.from_numpy(y).cuda() for y in ys], want_complex=False, layout=layout)
    for y, X, Mg in zip(ys, outs, mags):
        ref = OS.stft(y)                          # (257, T) complex64
        got = X.cpu().numpy() if layout == "FT" else X.cpu().numpy().T
        gm = Mg.cpu().numpy() if layout == "FT" else Mg.cpu().numpy().T
        assert got.shape == ref.shape == (257, 1 + len(y) // 128)
        tol = 1e-5 * np.abs(ref).max()            # stated tolerance: 1e-5 of the largest bin (fp32 FFT)
        np.testing.assert_allclose(got, ref, atol=tol)
        np.testing.assert_allclose(gm, np.abs(ref), atol=tol)


def test_stft_pcm16_input_and_time_major_batch_layout(ops):
    rng = np.random.default_rng(3)
    pcm = [rng.integers(-20000, 20000, n).astype(np.int16) for n in (5000, 3000)]
    Ts = [1 + len(p) // 128 for p in pcm]
    B, F, T = 2, 257, max(Ts)
    out = torch.zeros(T, B, F).cuda()
    ops.stft_batch([torch.from_numpy(p).cuda() for p in pcm], out=out, out_offs=[b * F for b in range(B)],
                   stride_t=[B * F] * B, stride_f=[1] * B)
    got = out.cpu().numpy()
    for b, p in enumerate(pcm):
        ref = OS.stft_mag(OS.pcm16_to_float(p)).T          # (T_b, F)
        np.testing.assert_allclose(got[:Ts[b], b], ref, atol=1e-5 * ref.max())
        assert np.all(got[Ts[b]:, b] == 0)                  # padding untouched


def test_mask_istft_matches_oracle_and_round_trips(ops):
    rng = np.random.default_rng(11)
    ns = [51072, 6400, 1000]
    ys = [_sig(n, 100 + n) for n in ns]
    specs = [OS.stft(y) for y in ys]
    masks = [[rng.uniform(0, 1, s.shape).astype(np.float32) for _ in range(2)] for s in specs]
    wav, pcm = ops.mask_istft([torch.from_numpy(s).cuda() for s in specs],
                              [[torch.from_numpy(m).cuda() for m in ms] for ms in masks])
    for u, s in enumerate(specs):
        for k in range(2):
            ref_f, ref_i = OS.reconstruct(s, masks[u][k])
            got_f, got_i = wav[u][k].cpu().numpy(), pcm[u][k].cpu().numpy()
            assert got_f.shape == ref_f.shape == (128 * (s.shape[1] - 1),)
            np.testing.assert_allclose(got_f, ref_f, atol=3e-6)          # fp32 iFFT + overlap-add
            d = np.abs(got_i.astype(np.int32) - ref_i.astype(np.int32))
            assert d.max() <= 1 and (d > 0).mean() < 2e-3               # truncation boundary cases only
    # no mask: istft(stft(x)) == x on the retained samples
    wav, _ = ops.mask_istft([torch.from_numpy(s).cuda() for s in specs], None, want_pcm=False)
    for u, y in enumerate(ys):
        got = wav[u][0].cpu().numpy()
        np.testing.assert_allclose(got, y[:len(got)], atol=3e-6)


def test_mask_istft_many_tiles_per_workgroup_and_frame_major_strides(ops):
    """(1) Enough (utterance, source) pairs that a workgroup walks MANY consecutive tiles (tpb > 2: the register-prefetch
    pipeline across tiles, ring wrap-around, ragged ends), checked against the oracle on a sample of the pairs.
    (2) The general-stride path of the C ABI: spectrum and mask given frame-major ((T, 257) rows, the model's own output
    order) must give the same samples as the (257, T) layout, bit for bit."""
    import ctypes as C
    from sepkern import _lib
    rng = np.random.default_rng(21)
    U, S = 64, 2                                                     # 65 tiles x 128 pairs -> 4 tiles per workgroup
    ns = [int(v) for v in rng.integers(90000, 140000, U)]            # 700 .. 1090 frames: 45 .. 69 tiles per pair
    ns[0], ns[1] = 128 * 1023 + 5, 128 * 1024 + 127                  # tile counts on both sides of a multiple of 16 frames
    ys = [_sig(n, 500 + i) for i, n in enumerate(ns)]
    specs = [OS.stft(y) for y in ys]
    masks = [[rng.uniform(0, 1, s.shape).astype(np.float32) for _ in range(S)] for s in specs]
    d_specs = [torch.from_numpy(s).cuda() for s in specs]
    d_masks = [[torch.from_numpy(m).cuda() for m in ms] for ms in masks]
    wav, pcm = ops.mask_istft(d_specs, d_masks)
    for u in (0, 1, 2, 17, U - 1):
        for k in range(S):
            ref_f, ref_i = OS.reconstruct(specs[u], masks[u][k])
            got_f, got_i = wav[u][k].cpu().numpy(), pcm[u][k].cpu().numpy()
            assert got_f.shape == ref_f.shape
            np.testing.assert_allclose(got_f, ref_f, atol=3e-6)
            d = np.abs(got_i.astype(np.int32) - ref_i.astype(np.int32))
            assert d.max() <= 1 and (d > 0).mean() < 2e-3
    # (2) frame-major operands through the C ABI, three utterances
    sel = [0, 2, 5]
    Ts = [specs[u].shape[1] for u in sel]
    F = 257
    spec_tf = torch.cat([d_specs[u].t().contiguous().view(-1) for u in sel])                       # (T, F) rows
    mask_tf = torch.cat([torch.cat([d_masks[u][k] for k in range(S)], 0).t().contiguous().view(-1) for u in sel])   # (T, S*F)
    i64 = lambda v: torch.tensor(v, dtype=torch.int64, device="cuda")
    moffs, koffs, ooffs, acc_m, acc_k, acc_o = [], [], [], 0, 0, 0
    for T in Ts:
        moffs.append(acc_m)
        acc_m += T * F
        for k in range(S):
            koffs.append(acc_k + k * F)                                                            # source k: columns k*F ..
            ooffs.append(acc_o)
            acc_o += 128 * (T - 1)
        acc_k += T * S * F
    out = torch.empty(acc_o, dtype=torch.float32, device="cuda")
    p = lambda t: C.c_void_p(t.data_ptr())
    args = [i64(moffs), i64([F] * 3), i64([1] * 3), i64(koffs), i64([S * F] * 3), i64([1] * 3),
            torch.tensor(Ts, dtype=torch.int32, device="cuda"), i64(ooffs)]
    _lib.call("sk_mask_istft", p(spec_tf), p(args[0]), p(args[1]), p(args[2]), p(mask_tf), p(args[3]), p(args[4]), p(args[5]),
              p(args[6]), 3, S, 512, 128, p(out), None, p(args[7]), max(Ts), C.c_void_p(torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    for i, u in enumerate(sel):
        for k in range(S):
            o = ooffs[i * S + k]
            assert torch.equal(out[o:o + 128 * (Ts[i] - 1)], wav[u][k])


def test_istft_int16_wraps_like_reference(ops):
    y = _sig(4096, 9) * 12.0                                   # |y| well above 1.0
    s = OS.stft(y)
    _, pcm = ops.mask_istft([torch.from_numpy(s).cuda()], None, want_float=False)
    ref = OS.to_int16_wav(OS.istft(s))
    d = np.abs(pcm[0][0].cpu().numpy().astype(np.int32) - ref.astype(np.int32))
    d = np.minimum(d, 65536 - d)
    assert d.max() <= 2 and (np.abs(y) > 1.0).any()


# ------------------------------------------------------------------------------------ PIT-MSE
@pytest.mark.parametrize("S", [1, 2, 3])
def test_pit_mse_fwd_bwd_matches_oracle(ops, S):
    torch.manual_seed(S)
    T, B, F = 19, 5, 257
    lens = torch.tensor([19, 17, 12, 12, 3])
    valid = (torch.arange(T)[:, None] < lens[None, :]).float().unsqueeze(2)        # (T,B,1)
    mask = torch.rand(T, B, S * F)
    mix = torch.rand(T, B, F) * valid
    srcs = [torch.rand(T, B, F) * valid for _ in range(S)]
    mo = mask.permute(1, 0, 2).contiguous().requires_grad_(True)
    loss, norm, losses, idx = OU.pit_mse(mo, mix.permute(1, 0, 2).contiguous(),
                                          [s.permute(1, 0, 2).contiguous() for s in srcs], lens, S, F)
    loss.backward()
    res = ops.pit_mse_fwd(dev(mask), dev(mix), [dev(s) for s in srcs], dev(lens.int()))
    out = res["out"].cpu().numpy()
    np.testing.assert_allclose(out[0], float(loss), rtol=2e-6)
    np.testing.assert_allclose(out[1], float(norm), rtol=0)
    np.testing.assert_allclose(res["perm_loss"].cpu().numpy(), losses.detach().numpy(), rtol=2e-6)
    assert res["best_perm"].cpu().tolist() == idx.tolist()
    dm = ops.pit_mse_bwd(dev(mask), dev(mix), [dev(s) for s in srcs], res["best_perm"], res["out"],
                         torch.ones(1).cuda())
    np.testing.assert_allclose(dm.cpu().permute(1, 0, 2).numpy(), mo.grad.numpy(), rtol=1e-5, atol=1e-10)
    # the same on PACKED rows (PackedSequence.data, what the collator hands over): only the valid frames exist
    from sepkern.packing import Packing
    pk = Packing.from_lens(lens.tolist(), "cuda")
    pm, px, ps = pk.pack(dev(mask)), pk.pack(dev(mix)), [pk.pack(dev(s)) for s in srcs]
    assert pm.shape == (pk.Rp, S * F) and pk.R == int(lens.sum())
    res2 = ops.pit_mse_fwd(pm, px, ps, None, packing=pk)
    np.testing.assert_allclose(res2["out"].cpu().numpy(), out, rtol=2e-6)
    np.testing.assert_allclose(res2["perm_loss"].cpu().numpy(), losses.detach().numpy(), rtol=2e-6)
    assert res2["best_perm"].cpu().tolist() == idx.tolist()
    dm2 = ops.pit_mse_bwd(pm, px, ps, res2["best_perm"], res2["out"], torch.ones(1).cuda(), packing=pk)
    np.testing.assert_allclose(pk.unpack(dm2).cpu().permute(1, 0, 2).numpy(), mo.grad.numpy(), rtol=1e-5, atol=1e-10)


# ------------------------------------------------------------------------------------ BN / colsum / sigmoid
def test_bn_stats_apply_backward(ops):
    torch.manual_seed(0)
    R, Cc = 1000, 200
    x = torch.randn(R, Cc) * 0.5 + 0.3
    x[700:] = 0.0                                             # zero-padded frames are part of the statistics
    bn = torch.nn.BatchNorm1d(Cc)
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5)
        bn.bias.normal_()
    xr = x.clone().requires_grad_(True)
    y = bn(xr)
    dy = torch.randn(R, Cc)
    y.backward(dy)
    mean, var = torch.empty(Cc).cuda(), torch.empty(Cc).cuda()
    ops.bn_stats(dev(x), mean, var)
    np.testing.assert_allclose(mean.cpu().numpy(), x.double().mean(0).numpy(), atol=1e-6)
    np.testing.assert_allclose(var.cpu().numpy(), x.double().var(0, unbiased=False).numpy(), rtol=2e-5)
    # packed rows: only the 700 non-zero rows are stored, the statistics still cover all R = 1000 positions
    mean2, var2 = torch.empty(Cc).cuda(), torch.empty(Cc).cuda()
    ops.bn_stats(dev(x[:700].contiguous()), mean2, var2, count=R)
    np.testing.assert_allclose(mean2.cpu().numpy(), mean.cpu().numpy(), atol=1e-6)
    np.testing.assert_allclose(var2.cpu().numpy(), var.cpu().numpy(), rtol=2e-5)
    rm, rv = torch.zeros(Cc).cuda(), torch.ones(Cc).cuda()
    ops.bn_update_running(mean, var, rm, rv, R, 0.1, guard=torch.ones(1, dtype=torch.int32).cuda())
    assert float(rm.abs().sum()) == 0 and bool((rv == 1).all())          # a raised guard word: running statistics untouched
    ops.bn_update_running(mean, var, rm, rv, R, 0.1, guard=torch.zeros(1, dtype=torch.int32).cuda())
    np.testing.assert_allclose(rm.cpu().numpy(), bn.running_mean.numpy(), atol=1e-6)
    np.testing.assert_allclose(rv.cpu().numpy(), bn.running_var.numpy(), rtol=2e-5)
    out = torch.empty(R, Cc).cuda()
    ops.bn_apply(dev(x), mean, var, dev(bn.weight.detach()), dev(bn.bias.detach()), out, 1e-5)
    np.testing.assert_allclose(out.cpu().numpy(), y.detach().numpy(), atol=5e-6)
    dx, dg, db = torch.empty(R, Cc).cuda(), torch.empty(Cc).cuda(), torch.empty(Cc).cuda()
    ops.bn_bwd(dev(dy), dev(x), mean, var, dev(bn.weight.detach()), dx, dg, db, 1e-5)
    np.testing.assert_allclose(dg.cpu().numpy(), bn.weight.grad.numpy(), rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(db.cpu().numpy(), bn.bias.grad.numpy(), rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(dx.cpu().numpy(), xr.grad.numpy(), rtol=1e-4, atol=2e-6)


def test_bn_folded_into_linear_forward_and_weight_gradient(ops):
    """sk_bn_fold / sk_bn_unfold_grad (BatchNorm normalisation fused into the Linear that follows it):
    lin(bn(x)) = x Wf^T + bf, and dW = dz^T bn(x) recovered from G = dz^T x -- against the explicit computation in fp64."""
    g = torch.Generator().manual_seed(8)
    R, Cc, O = 300, 70, 23
    x = torch.randn(R, Cc, generator=g) * 3 + 1
    W, b = torch.randn(O, Cc, generator=g), torch.randn(O, generator=g)
    gamma, beta = torch.rand(Cc, generator=g) + 0.5, torch.randn(Cc, generator=g)
    mean, var = x.mean(0), x.var(0, unbiased=False)
    eps = 1e-5
    Wf, bf, s, t = ops.bn_fold(dev(W), dev(b), dev(mean), dev(var), dev(gamma), dev(beta), eps, ld=72)
    xd = x.double()
    xbn = (xd - mean.double()) / torch.sqrt(var.double() + eps) * gamma.double() + beta.double()
    ref = xbn @ W.double().t() + b.double()
    got = xd @ Wf.cpu().double()[:, :Cc].t() + bf.cpu().double()
    np.testing.assert_allclose(got.numpy(), ref.numpy(), atol=1e-4)
    assert torch.all(Wf[:, Cc:] == 0)
    dz = torch.randn(R, O, generator=g)
    G = (dz.double().t() @ xd).float()
    dW0 = torch.randn(O, Cc, generator=g)
    for acc in (False, True):
        dW = dev(dW0.clone())
        ops.bn_unfold_grad(dev(G), dev(dz.sum(0)), s, t, dW, accumulate=acc)
        want = dz.double().t() @ xbn + (dW0.double() if acc else 0)
        np.testing.assert_allclose(dW.cpu().double().numpy(), want.numpy(), atol=2e-3, rtol=1e-5)


def test_colsum_and_sigmoid_bwd(ops):
    torch.manual_seed(1)
    x = torch.randn(777, 130)
    out = torch.ones(100).cuda()
    ops.colsum(dev(x)[:, 10:], 777, 100, 130, out, accumulate=True)
    np.testing.assert_allclose(out.cpu().numpy(), x[:, 10:110].double().sum(0).numpy() + 1.0, atol=2e-4)
    m, dm = torch.rand(5000), torch.randn(5000)
    dz = torch.empty(5000).cuda()
    ops.sigmoid_bwd(dev(dm), dev(m), dz)
    np.testing.assert_allclose(dz.cpu().numpy(), (dm * m * (1 - m)).numpy(), rtol=1e-6, atol=1e-8)


def test_clip_adam_matches_torch(ops):
    torch.manual_seed(2)
    n = 100_003
    p0 = torch.randn(n)
    pt = torch.nn.Parameter(p0.clone())
    opt = torch.optim.Adam([pt], lr=1e-3)
    p = dev(p0.clone())
    m, v, scal = torch.zeros(n).cuda(), torch.zeros(n).cuda(), torch.zeros(4).cuda()     # scal: 4 floats (sepkern.h)
    for step in range(1, 4):
        g = torch.randn(n) * (0.01 if step == 2 else 1e-4)           # step 2 clips, the others do not
        pt.grad = g.clone()
        tn = torch.nn.utils.clip_grad_norm_([pt], 0.25)
        opt.step()
        ops.grad_norm(dev(g), 0.25, scal)
        ops.clip_adam(p, dev(g), m, v, scal, 1e-3, 0.9, 0.999, 1e-8, step)
        np.testing.assert_allclose(scal[0].item(), float(tn), rtol=1e-5)
        np.testing.assert_allclose(p.cpu().numpy(), pt.detach().numpy(), rtol=1e-5, atol=1e-6)


# ------------------------------------------------------------------------------------ BLSTM recurrence
def _layer_case(T, B, H, I, lens, seed):
    g = torch.Generator().manual_seed(seed)
    k = 1.0 / np.sqrt(H)
    w = [[tuple((torch.rand(s, generator=g) * 2 - 1) * k for s in ((4 * H, I), (4 * H, H), (4 * H,), (4 * H,)))
          for _ in range(2)]]
    x = torch.randn(T, B, I, generator=g)
    for b, n in enumerate(lens):
        x[n:, b] = 0
    h0 = torch.randn(2, B, H, generator=g)
    c0 = torch.randn(2, B, H, generator=g)
    return w, x, h0, c0


class _Rows:
    """The two row layouts of the recurrence entry points behind one face: zero-padded (T, B, .) (offs = NULL) and
    PACKED rows (torch's PackedSequence.data layout, sepkern.packing.Packing)."""

    def __init__(self, layout, T, B, lens):
        from sepkern.packing import Packing
        self.packed, self.T, self.B = layout == "packed", T, B
        self.pk = Packing.from_lens(lens, "cuda")
        self.offs = self.pk.offs if self.packed else None
        self.R = self.pk.Rp if self.packed else T * B
        self.lens = torch.tensor(lens, dtype=torch.int32).cuda()
        self.valid = (torch.arange(T)[:, None] < torch.tensor(lens)[None, :]).cuda()

    def put(self, t):                    # (T, B, ...) host or device tensor -> this layout's rows on the device
        t = t.cuda()
        return self.pk.pack(t.reshape(self.T, self.B, -1)).clone() if self.packed else t.reshape(self.T * self.B, -1).clone()

    def get(self, rows, C):              # rows -> (T, B, C), zeros at padded positions
        rows = rows.reshape(self.R, -1)
        return self.pk.unpack(rows) if self.packed else rows.view(self.T, self.B, C)

    def new(self, C, fill=float("nan")):
        return torch.full((self.R, C), fill).cuda()


@pytest.mark.parametrize("layout", ["padded", "packed"])
@pytest.mark.parametrize("mode", [2, 1])
@pytest.mark.parametrize("T,B,H,I,lens", [
    (5, 3, 8, 6, [5, 3, 1]),
    (7, 20, 300, 33, [7] * 5 + [6] * 5 + [4] * 5 + [1] * 5),
    (6, 32, 600, 40, [6] * 16 + [5] * 8 + [2] * 8),
    (4, 32, 896, 24, [4] * 20 + [3] * 12),
    (5, 100, 600, 20, [5] * 40 + [4] * 30 + [2] * 29 + [1]),     # the reference's default batch: 3 batch groups per workgroup
    (4, 40, 896, 16, [4] * 17 + [3] * 20 + [1] * 3),             # 3 batch groups over 2 workgroup rows, last one ragged
])
def test_lstm_layer_fwd_bwd_matches_oracle(ops, mode, layout, T, B, H, I, lens):
    w, x, h0, c0 = _layer_case(T, B, H, I, lens, seed=T * 100 + H)
    # ---- oracle with autograd
    wr = [[tuple(t.clone().requires_grad_(True) for t in w[0][d]) for d in range(2)]]
    h0r, c0r = h0.clone().requires_grad_(True), c0.clone().requires_grad_(True)
    y_ref, hn_ref, cn_ref = OU.blstm_padded(x, lens, wr, h0r, c0r)
    gd = torch.Generator().manual_seed(1)
    dy = torch.randn(T, B, 2 * H, generator=gd)
    dhn, dcn = torch.randn(2, B, H, generator=gd), torch.randn(2, B, H, generator=gd)   # the final state feeds a later pass (RSH)
    ((y_ref * dy).sum() + (hn_ref * dhn).sum() + (cn_ref * dcn).sum()).backward()
    # ---- kernels
    rw = _Rows(layout, T, B, lens)
    R = rw.R
    wih = torch.stack([w[0][d][0] for d in range(2)]).cuda()          # (2,4H,I)
    whh = torch.stack([w[0][d][1] for d in range(2)]).cuda()          # (2,4H,H)
    bsum = torch.stack([w[0][d][2] + w[0][d][3] for d in range(2)]).reshape(-1).cuda()
    xr = rw.put(x)
    gx = rw.new(8 * H)
    # gx in the recurrence's gate-interleaved order (4u + g): reordered rows of W_ih and of the bias, plain GEMM
    ops.gemm(xr, ops.gate_rows(wih.view(8 * H, I), H), gx, R, 8 * H, I, I, I, 8 * H, transB=True, bias=ops.gate_rows(bsum, H))
    y = rw.new(2 * H)
    cs = rw.new(2 * H)
    hn, cn = torch.empty(2, B, H).cuda(), torch.empty(2, B, H).cuda()
    ws = ops.lstm_fwd(gx, whh, dev(h0), dev(c0), rw.lens, y, gx, cs, hn, cn, T, B, H, mode, offs=rw.offs)
    ops.lstm_status(ws)
    tol = dict(rtol=2e-5, atol=2e-6)
    if rw.packed and rw.pk.Rp > rw.pk.R:
        assert bool(torch.isnan(y[rw.pk.R:]).all())                  # rows past the packed data are nobody's to write
    np.testing.assert_allclose(rw.get(y, 2 * H).cpu().numpy(), y_ref.detach().numpy(), **tol)
    np.testing.assert_allclose(hn.cpu().numpy(), hn_ref.detach().numpy(), **tol)
    np.testing.assert_allclose(cn.cpu().numpy(), cn_ref.detach().numpy(), **tol)
    dh0, dc0 = torch.empty(2, B, H).cuda(), torch.empty(2, B, H).cuda()
    nbg = (B + 15) // 16
    dbias = torch.full((nbg, 2, 4 * H), float("nan")).cuda()         # by-product: bias-gradient partials
    ws = ops.lstm_bwd(rw.put(dy), whh, gx, cs, dev(c0), rw.lens, gx, dh0, dc0, T, B, H, mode, dhn=dev(dhn), dcn=dev(dcn),
                      dbias=dbias, offs=rw.offs)
    ops.lstm_status(ws)
    gtol = dict(rtol=1e-4, atol=2e-6)
    np.testing.assert_allclose(dh0.cpu().numpy(), h0r.grad.numpy(), **gtol)
    np.testing.assert_allclose(dc0.cpu().numpy(), c0r.grad.numpy(), **gtol)
    dgx_pad = rw.get(gx, 8 * H).contiguous()                         # (T, B, 8H), zero at padded positions in either layout
    if rw.packed:
        assert bool((dgx_pad[~rw.valid] == 0).all())
    dgx = ops.gates_interleaved(dgx_pad.view(T * B, 2, 4 * H), H, back=True).cpu().double()   # back to torch's gate-major order
    assert torch.isfinite(dgx).all()
    # dW_hh as the engine forms it: the recurrent inputs of the PACKED rows gathered (sk_hprev_rows), one batched product
    pk = rw.pk
    yp, dgp = pk.pack(rw.get(y, 2 * H).contiguous()), pk.pack(dgx_pad)
    hp = ops.hprev_rows(yp, dev(h0), pk, H, pk.rows(2 * H))
    dwhh_gi = torch.full((2, 4 * H, H), float("nan")).cuda()
    ops.gemm(dgp, hp, dwhh_gi, 4 * H, H, pk.Rp, 8 * H, 2 * H, H, transA=True, batch=2, sA=4 * H, sB=H, sC=4 * H * H, splitk=0)
    dwhh = ops.gate_rows(dwhh_gi, H, back=True)                       # rows come out interleaved, like dgx
    dwhh2 = ops.gate_rows(dwhh_gi, H, back=True, out=dwhh.clone(), accumulate=True)
    db = dbias.cpu().double().sum(0)                                  # (2, 4H)
    for d in range(2):
        w_ih_g, w_hh_g, b_ih_g, b_hh_g = (t.grad for t in wr[0][d])
        np.testing.assert_allclose((dgx[:, d].t() @ x.double().view(T * B, I)).numpy(), w_ih_g.numpy(), rtol=1e-4, atol=2e-5)
        np.testing.assert_allclose(dwhh[d].cpu().numpy(), w_hh_g.numpy(), rtol=1e-4, atol=2e-5)
        np.testing.assert_allclose(dwhh2[d].cpu().numpy(), 2 * w_hh_g.numpy(), rtol=1e-4, atol=4e-5)
        np.testing.assert_allclose(dgx[:, d].sum(0).numpy(), b_ih_g.numpy(), rtol=1e-4, atol=2e-5)
        np.testing.assert_allclose(db[d].numpy(), b_ih_g.numpy(), rtol=1e-4, atol=2e-5)
        np.testing.assert_allclose(b_hh_g.numpy(), b_ih_g.numpy())


@pytest.mark.parametrize("lens", [[9, 7, 7, 3], [5, 5, 5], [6] + [1] * 40, [4, 4, 2, 1], [1]])
def test_pack_unpack_and_recurrent_input_rows(ops, lens):
    """sk_pack_rows / sk_unpack_rows / sk_hprev_rows against torch's own PackedSequence: the packed rows ARE
    pack_padded_sequence(...).data, unpacking restores the padded tensor (or puts a fill row at padded positions), an unsorted
    batch goes through `perm`, and the recurrent-input rows are the output one frame earlier / later or h0."""
    from torch.nn.utils.rnn import pack_padded_sequence
    from sepkern.packing import Packing
    g = torch.Generator().manual_seed(len(lens))
    T, B, C, H = max(lens), len(lens), 13, 8
    x = torch.randn(T, B, C, generator=g)
    for b, n in enumerate(lens):
        x[n:, b] = 0
    pk = Packing.from_lens(lens, "cuda")
    ref = pack_padded_sequence(x, torch.tensor(lens), enforce_sorted=True)
    assert pk.R == ref.data.shape[0] and pk.offs_host[-1] == pk.R
    assert np.array_equal(np.diff(pk.offs_host), ref.batch_sizes.numpy())
    pk2 = Packing.from_batch_sizes(ref.batch_sizes, "cuda")
    assert np.array_equal(pk2.lens_host, np.asarray(lens)) and np.array_equal(pk2.offs_host, pk.offs_host)
    rows = pk.pack(dev(x))
    assert torch.equal(rows[:pk.R].cpu(), ref.data) and float(rows[pk.R:].abs().sum()) == 0
    assert torch.equal(pk.unpack(rows).cpu(), x)
    fill = torch.randn(C, generator=g)
    want = x.clone()
    for b, n in enumerate(lens):
        want[n:, b] = fill
    if not pk.uniform:
        assert torch.equal(pk.unpack(rows, fill=dev(fill)).cpu(), want)
    # an unsorted batch: sorted through perm, restored in the caller's order
    order = torch.randperm(B, generator=g)
    xs, ls = x[:, order].contiguous(), [lens[int(i)] for i in order]
    pks = Packing.from_lens(ls, "cuda")
    rs = pks.pack(dev(xs))
    assert torch.equal(pks.unpack(rs).cpu(), xs)
    order_s = pks.perm_host if pks.perm_host is not None else np.arange(B)      # sorted position -> the caller's utterance
    assert list(pks.lens_host) == sorted(ls, reverse=True) and [ls[int(i)] for i in order_s] == list(pks.lens_host)
    for t in (0, T - 1):
        n_t = int(pks.offs_host[t + 1] - pks.offs_host[t])
        assert torch.equal(rs[pks.offs_host[t]:pks.offs_host[t] + n_t].cpu(), xs[t, torch.as_tensor(order_s[:n_t].astype(np.int64))])
    hs = torch.randn(2, B, H, generator=g)
    assert torch.equal(pks.unsort_batch(pks.sort_batch(dev(hs), 1), 1).cpu(), hs)
    # recurrent inputs
    y = torch.randn(T, B, 2 * H, generator=g)
    h0 = torch.randn(2, B, H, generator=g)
    want = torch.zeros(T, B, 2 * H)
    for b, n in enumerate(lens):
        for t in range(n):
            want[t, b, :H] = y[t - 1, b, :H] if t > 0 else h0[0, b]
            want[t, b, H:] = y[t + 1, b, H:] if t + 1 < n else h0[1, b]
    yp = pk.pack(dev(y))
    hp = ops.hprev_rows(yp, dev(h0), pk, H, pk.rows(2 * H))
    assert torch.equal(pk.unpack(hp).cpu(), want) and float(hp[pk.R:].abs().sum()) == 0
    hb = ops.hprev_rows(yp, dev(h0), pk, H, torch.zeros(pk.Rp + 64, 64, dtype=torch.bfloat16).cuda())
    assert torch.equal(hb[:pk.R, :2 * H].float().cpu(), pk.pack(want.cuda())[:pk.R].cpu().bfloat16().float())
    assert float(hb[pk.R:].float().abs().sum()) == 0 and float(hb[:, 2 * H:].float().abs().sum()) == 0


@pytest.mark.parametrize("layout", ["padded", "packed"])
@pytest.mark.parametrize("mode", [2, 1])
@pytest.mark.parametrize("T,B,H,I,lens", [
    (5, 3, 8, 6, [5, 3, 1]),
    (7, 20, 300, 33, [7] * 5 + [6] * 5 + [4] * 5 + [1] * 5),
    (6, 32, 600, 40, [6] * 16 + [5] * 8 + [2] * 8),              # H=600: 40 unit groups in bf16 (38 in fp32)
    (4, 32, 896, 24, [4] * 20 + [3] * 12),
    (4, 40, 1024, 16, [4] * 17 + [3] * 20 + [1] * 3),
])
def test_lstm_layer_bf16_matches_bf16_oracle(ops, mode, layout, T, B, H, I, lens):
    """The recurrence with bf16 matrix-core inputs (mode bit 16; BASELINE configs[3]) against the CPU computation of
    the same arithmetic (oracle/upit_bf16.py: W_hh, h_{t-1} and dG_t rounded to bf16 inside the products)."""
    from oracle import upit_bf16 as OB
    w, x, h0, c0 = _layer_case(T, B, H, I, lens, seed=T * 100 + H + 1)
    wr = [[tuple(t.clone().requires_grad_(True) for t in w[0][d]) for d in range(2)]]
    h0r, c0r = h0.clone().requires_grad_(True), c0.clone().requires_grad_(True)
    y_ref, hn_ref, cn_ref = OB.blstm_padded(x, lens, wr, h0r, c0r)
    gd = torch.Generator().manual_seed(2)
    dy = torch.randn(T, B, 2 * H, generator=gd)
    dhn, dcn = torch.randn(2, B, H, generator=gd), torch.randn(2, B, H, generator=gd)
    ((y_ref * dy).sum() + (hn_ref * dhn).sum() + (cn_ref * dcn).sum()).backward()
    rw = _Rows(layout, T, B, lens)
    R = rw.R
    wih = torch.stack([w[0][d][0] for d in range(2)]).cuda()
    whh = torch.stack([w[0][d][1] for d in range(2)]).cuda()
    bsum = torch.stack([w[0][d][2] + w[0][d][3] for d in range(2)]).reshape(-1).cuda()
    gx = rw.new(8 * H)
    ops.gemm(rw.put(x), ops.gate_rows(wih.view(8 * H, I), H), gx, R, 8 * H, I, I, I, 8 * H, transB=True, bias=ops.gate_rows(bsum, H),
             bf16=True)
    y = rw.new(2 * H)
    cs = rw.new(2 * H)
    hn, cn = torch.empty(2, B, H).cuda(), torch.empty(2, B, H).cuda()
    ws = ops.lstm_fwd(gx, whh, dev(h0), dev(c0), rw.lens, y, gx, cs, hn, cn, T, B, H, mode, bf16=True, offs=rw.offs)
    ops.lstm_status(ws)
    # a rounding-order difference in fp32 can flip a bf16 rounding of one h (2^-9 relative on one term of H)
    tol = dict(rtol=2e-3, atol=2e-4)
    np.testing.assert_allclose(rw.get(y, 2 * H).cpu().numpy(), y_ref.detach().numpy(), **tol)
    np.testing.assert_allclose(hn.cpu().numpy(), hn_ref.detach().numpy(), **tol)
    np.testing.assert_allclose(cn.cpu().numpy(), cn_ref.detach().numpy(), **tol)
    dh0, dc0 = torch.empty(2, B, H).cuda(), torch.empty(2, B, H).cuda()
    ws = ops.lstm_bwd(rw.put(dy), whh, gx, cs, dev(c0), rw.lens, gx, dh0, dc0, T, B, H, mode, dhn=dev(dhn), dcn=dev(dcn),
                      bf16=True, offs=rw.offs)
    ops.lstm_status(ws)

    def close(a, ref, what):
        err = float((a.double() - ref.double()).norm() / (ref.double().norm() + 1e-30))
        assert err < 3e-3, (what, err)
    close(dh0.cpu(), h0r.grad, "dh0")
    close(dc0.cpu(), c0r.grad, "dc0")
    dgx = ops.gates_interleaved(rw.get(gx, 8 * H).contiguous().view(T * B, 2, 4 * H), H, back=True).cpu()
    assert torch.isfinite(dgx).all()
    for d in range(2):
        close(dgx[:, d].sum(0), wr[0][d][2].grad, "db dir %d" % d)     # column sums of dG = bias gradient


# ----------------------------------------------------------------------------- bf16 operands in memory (r02)
@pytest.mark.parametrize("M,N,K", [(256, 256, 64), (300, 200, 257), (1000, 771, 1792), (513, 1300, 320), (50, 7168, 128)])
def test_gemm_bf16_nt_equals_fp64_product_of_rounded_operands(ops, M, N, K):
    """sk_cast_bf16 + sk_gemm_bf16_nt: bf16 x bf16 products are exact in fp32, so against the fp64 product of the
    ROUNDED operands only the fp32 accumulation order differs (the fp32 kernel's tolerance).  Ragged M / N (clamped
    edge tiles), K padded to 64 with zero columns, bias + sigmoid epilogue."""
    g = torch.Generator().manual_seed(M + N + K)
    A, B = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g)
    bias = torch.randn(N, generator=g)
    Ab, Bb = ops.cast_bf16(dev(A)), ops.cast_bf16(dev(B))
    Kp = Ab.shape[1]
    assert Kp % 64 == 0 and Kp >= K and Ab.dtype == torch.bfloat16
    np.testing.assert_array_equal(Ab[:, :K].float().cpu().numpy(), A.bfloat16().float().numpy())   # RNE, as torch rounds
    assert float(Ab[:, K:].float().abs().sum()) == 0.0
    C = torch.full((M, N), float("nan")).cuda()
    ops.gemm_bf16_nt(Ab, Bb, C, M, N, Kp, Kp, Kp, N, bias=dev(bias))
    ref = A.bfloat16().double() @ B.bfloat16().double().t() + bias.double()
    err = float((C.cpu().double() - ref).abs().max() / ref.abs().max())
    assert err < 2e-6, err
    ops.gemm_bf16_nt(Ab, Bb, C, M, N, Kp, Kp, Kp, N, bias=dev(bias), act=1)
    np.testing.assert_allclose(C.cpu().numpy(), torch.sigmoid(ref).float().numpy(), atol=2e-5)   # pre-activations of O(sqrt(K))


def test_gemm_bf16_nt_splitk_batch_accumulate(ops):
    """The NT form with K split into slabs, a batch of 2 with strides, accumulation into C; bitwise reproducible."""
    g = torch.Generator().manual_seed(7)
    R, M, N = 1000, 300, 140
    X, Y = torch.randn(R, 2 * M, generator=g), torch.randn(R, 2 * N, generator=g)
    Xt, Yt = ops.cast_bf16(dev(X.t().contiguous())), ops.cast_bf16(dev(Y.t().contiguous()))     # (2M, ld), (2N, ld): K-contiguous
    ld = Xt.shape[1]
    assert Xt.shape == (2 * M, ld) and ld % 64 == 0 and ld >= R
    np.testing.assert_array_equal(Xt[:, :R].float().cpu().numpy(), X.t().bfloat16().float().numpy())
    assert float(Xt[:, R:].float().abs().sum()) == 0.0
    C0 = torch.randn(2, M, N, generator=g)
    Kp = ops.pad_to(R, 64)
    outs = []
    for _ in range(2):
        C = dev(C0.clone())
        ops.gemm_bf16_nt(Xt, Yt, C, M, N, Kp, ld, ld, N, accumulate=True, batch=2, sA=M * ld, sB=N * ld, sC=M * N, splitk=3)
        outs.append(C.clone())
    assert torch.equal(outs[0], outs[1])
    for z in range(2):
        ref = C0[z].double() + X[:, z * M:(z + 1) * M].bfloat16().double().t() @ Y[:, z * N:(z + 1) * N].bfloat16().double()
        err = float((outs[0][z].cpu().double() - ref).abs().max() / ref.abs().max())
        assert err < 2e-6, (z, err)


@pytest.mark.parametrize("akm,bkm", [(False, True), (True, False), (True, True)])
@pytest.mark.parametrize("M,N,K,batch,splitk", [(512, 512, 256, 1, 1), (771, 1792, 1280, 1, 0), (7168, 257, 1024, 1, 4),
                                                (300, 130, 192, 2, 1), (3584, 896, 640, 2, 2), (1000, 1800, 128, 1, 1)])
def test_gemm_bf16_k_major_operands_equal_fp64_product_of_rounded_operands(ops, M, N, K, batch, splitk, akm, bkm):
    """sk_gemm_bf16_mm: operands that are K-major in memory (activation / gradient matrices as the transposed factors of
    a weight gradient, a weight matrix in a data gradient) are DMA'd as they lie and transposed by ds_read_b64_tr_b16 on
    the way into the matrix cores: same product as the NT form on transposed copies.  Ragged M / N against padded leading
    dimensions (771 -> 776, 257 -> 264), batches, accumulation, split-K; the padding columns hold NaN (never stored)."""
    g = torch.Generator().manual_seed(M + 3 * N + K)
    pad8 = lambda n: (n + 7) // 8 * 8
    A = torch.randn(batch, M, K, generator=g).bfloat16()
    Bm = torch.randn(batch, N, K, generator=g).bfloat16()
    ref = torch.einsum("zmk,znk->zmn", A.double(), Bm.double())
    if akm:
        lda = pad8(M) + 8
        Ad = torch.full((batch, K, lda), float("nan")).bfloat16()
        Ad[:, :, :M] = A.transpose(1, 2)
        sA = K * lda
    else:
        lda, Ad, sA = K, A, M * K
    if bkm:
        ldb = pad8(N)
        Bd = torch.full((batch, K, ldb), float("nan")).bfloat16()
        Bd[:, :, :N] = Bm.transpose(1, 2)
        sB = K * ldb
    else:
        ldb, Bd, sB = K, Bm, N * K
    C0 = torch.randn(batch, M, N, generator=g)
    C = C0.clone().cuda()
    ops.gemm_bf16_mm(Ad.contiguous().cuda(), Bd.contiguous().cuda(), C, M, N, K, lda, ldb, N, a_kmajor=akm, b_kmajor=bkm,
                     accumulate=True, batch=batch, sA=sA, sB=sB, sC=M * N, splitk=splitk)
    err = float((C.cpu().double() - (ref + C0.double())).abs().max() / ref.abs().max())
    assert err < 2e-6, err


@pytest.mark.parametrize("akm,bkm", [(False, False), (False, True), (True, False), (True, True)])
@pytest.mark.parametrize("M,N,K", [(1000, 772 + 512, 1792),      # 24 tiles, no whole round: every tile in ~10 pieces
                                   (4352, 4096, 512),            # 272 tiles: one round + 16 tiles cut at every K step
                                   (5000, 3800, 1088),           # 300 tiles, ragged edges, ranges that span two tiles
                                   (4096, 4096, 512)])           # a whole number of rounds: no cut at all
def test_gemm_bf16_stream_k_kernel(ops, M, N, K, akm, bkm):
    """sk_gemm_bf16_mm with splitk = 1 and a stream-K workspace (r03): the persistent 256 x 256-tile bf16 kernel with the last
    partial round of tiles cut along K.  All four operand forms, bias + accumulate, against the fp64 product of the bf16
    operands; run-to-run identical; ticket counters left zeroed."""
    from sepkern import _lib
    g = torch.Generator().manual_seed(M + 3 * N + K)
    pad8 = lambda n: (n + 7) // 8 * 8
    A = torch.randn(M, K, generator=g).bfloat16()
    Bm = torch.randn(N, K, generator=g).bfloat16()
    bias, C0 = torch.randn(N, generator=g), torch.randn(M, N, generator=g)
    ref = A.double() @ Bm.double().t() + bias.double() + C0.double()
    if akm:
        lda = pad8(M) + 8
        Ad = torch.full((K, lda), float("nan")).bfloat16()
        Ad[:, :M] = A.t()
    else:
        lda, Ad = K, A
    if bkm:
        ldb = pad8(N)
        Bd = torch.full((K, ldb), float("nan")).bfloat16()
        Bd[:, :N] = Bm.t()
    else:
        ldb, Bd = K, Bm
    Ad, Bd, bias_d = Ad.contiguous().cuda(), Bd.contiguous().cuda(), bias.cuda()
    outs = []
    for _ in range(2):
        C = C0.clone().cuda()
        ops.gemm_bf16_mm(Ad, Bd, C, M, N, K, lda, ldb, N, a_kmajor=akm, b_kmajor=bkm, bias=bias_d, accumulate=True, ws_tag="t_bsk",
                         streamk=True)
        torch.cuda.synchronize()
        outs.append(C.cpu())
    err = float((outs[0].double() - ref).abs().max() / ref.abs().max())
    assert err < 2e-6, err
    assert torch.equal(outs[0], outs[1])
    ws = ops.workspace(_lib.load().sk_gemm_streamk_workspace_bytes(), "t_bsk_sk")
    assert int(ws[:65536].max()) == 0


@pytest.mark.parametrize("layout", ["padded", "packed"])
@pytest.mark.parametrize("T,B,H,lens", [(9, 32, 896, [9] * 20 + [7] * 8 + [2] * 4), (7, 40, 600, [7] * 17 + [5] * 20 + [1] * 3),
                                        (6, 16, 64, [6] * 10 + [3] * 6)])
def test_lstm_geometry_and_protocol_variants_are_bitwise_identical(ops, layout, T, B, H, lens):
    """The speed-only variants of the persistent recurrence (include/sepkern.h, mode bits 17..27: 8-unit workgroups, block
    -> stream maps, one polling wave, replicated flags, poll hold-back) change which workgroup computes what and how the
    hand-off is signalled, never the arithmetic: outputs equal the default's bit for bit."""
    g = torch.Generator().manual_seed(H + T)
    rw = _Rows(layout, T, B, lens)
    gx = rw.put(torch.randn(T, B, 2, 4 * H, generator=g) * 0.5)
    whh = (torch.randn(2, 4 * H, H, generator=g) / 30).cuda()
    h0, c0 = torch.randn(2, B, H, generator=g).cuda(), torch.randn(2, B, H, generator=g).cuda()
    dy = rw.put(torch.randn(T, B, 2 * H, generator=g))

    def fwd(bits):
        gg = gx.clone()
        y, cs = rw.new(2 * H, 0.0), rw.new(2 * H, 0.0)
        hn, cn = torch.empty(2, B, H).cuda(), torch.empty(2, B, H).cuda()
        ws = ops.lstm_fwd(gg, whh, h0, c0, rw.lens, y, gg, cs, hn, cn, T, B, H, 1 | bits, offs=rw.offs)
        ops.lstm_status(ws)
        return y, gg, cs, hn, cn

    def bwd(bits, saved):
        y, gates, cs, _, _ = saved
        gg = gates.clone()
        dh0, dc0 = torch.empty(2, B, H).cuda(), torch.empty(2, B, H).cuda()
        dbias = torch.empty((B + 15) // 16, 2, 4 * H).cuda()
        ws = ops.lstm_bwd(dy, whh, gg, cs, c0, rw.lens, gg, dh0, dc0, T, B, H, 1 | bits, dbias=dbias, offs=rw.offs)
        ops.lstm_status(ws)
        return gg, dh0, dc0, dbias

    def rows_of(a):          # gates / cs are defined at valid steps only (padded layout: compare those)
        if a.dim() == 2 and a.shape[0] == rw.R and not rw.packed and a.shape[1] in (8 * H, 2 * H):
            return a.view(T, B, -1)[rw.valid]
        return a

    ref = fwd(ops.lstm_variant_bits(False, 0, False, False, False, 31))                # r01 geometry, no hold-back
    for variant in [(False, 1, True, False, False, 0), (False, 1, True, True, False, 8), (False, 2, False, False, False, 31),
                    (False, 2, True, True, False, 4), (False, 1, True, False, True, 0),
                    (False, 0, False, False, True, 31)]:
        out = fwd(ops.lstm_variant_bits(*variant))
        for a, b in zip(out, ref):
            assert torch.equal(rows_of(a), rows_of(b)), variant
    bref = bwd(ops.lstm_variant_bits(False, 0, False, False, False, 31), ref)
    for variant in [(False, 1, False, False, False, 31), (False, 2, False, False, False, 6), (False, 1, False, False, True, 0)]:
        out = bwd(ops.lstm_variant_bits(*variant), ref)
        for a, b in zip(out, bref):
            assert torch.equal(a, b), variant


@pytest.mark.parametrize("layout", ["padded", "packed"])
@pytest.mark.parametrize("T,B,H,lens,delay", [(12, 32, 896, [12] * 20 + [7] * 8 + [2] * 3 + [1], 0), (40, 27, 896, [40] * 9 + [21] * 14 + [3] * 4, 4),
                                              (9, 20, 704, [9] * 7 + [4] * 13, 31), (33, 8, 896, [33] * 3 + [20] * 5, 0), (7, 32, 600, [7] * 32, 0),
                                              (6, 40, 896, [6] * 40, 0)])
def test_lstm_forward_bf16_xcd_local_streams_of_eight_rows(ops, layout, T, B, H, lens, delay):
    """Mode bit 30 (bf16 forward, 608 < H <= 896, B <= 32, r06): every (direction, 8-row batch group) stream is 28 workgroups of
    32 units on ONE XCD, h_t published with plain stores and a plain flag (the XCD's L2 is the coherence point; sc1 polls and
    pulls), 14 KB pulled per step.  Who computes what and how it is signalled changes, the arithmetic does not: every output equals
    the ordinary bf16 form's bit for bit, run after run, with ragged lengths, B not a multiple of 8, hidden sizes that leave part
    of the last unit group empty; the status word stays clear.  H = 600 and B = 40 take the ordinary form (the bit is ignored)."""
    g = torch.Generator().manual_seed(H + T)
    rw = _Rows(layout, T, B, lens)
    gx = rw.put(torch.randn(T, B, 2, 4 * H, generator=g) * 0.5)
    whh = (torch.randn(2, 4 * H, H, generator=g) / 30).cuda()
    h0, c0 = torch.randn(2, B, H, generator=g).cuda(), torch.randn(2, B, H, generator=g).cuda()

    def fwd(bits):
        gg = gx.clone()
        y, cs = rw.new(2 * H, 0.0), rw.new(2 * H, 0.0)
        hn, cn = torch.empty(2, B, H).cuda(), torch.empty(2, B, H).cuda()
        ws = ops.lstm_fwd(gg, whh, h0, c0, rw.lens, y, gg, cs, hn, cn, T, B, H, 1 | bits, bf16=True, offs=rw.offs)
        ops.lstm_status(ws)
        return y, gg, cs, hn, cn

    def rows_of(a):
        if a.dim() == 2 and a.shape[0] == rw.R and not rw.packed and a.shape[1] in (8 * H, 2 * H):
            return a.view(T, B, -1)[rw.valid]
        return a[:rw.pk.R] if (a.dim() == 2 and a.shape[0] == rw.R) else a

    ref = fwd(ops.lstm_variant_bits(False, 1, True, False, True, 0))
    for rep in range(2):
        out = fwd(ops.lstm_variant_bits(False, 1, True, False, True, delay, xl8=True))
        for a_, b_ in zip(out, ref):
            assert torch.equal(rows_of(a_), rows_of(b_))
    # ... and the backward twin (lstm_bwd_xl8_kernel): dgx, its bf16 copy, dh0, dc0 and the bias-gradient partials bit for bit
    dy = rw.put(torch.randn(T, B, 2 * H, generator=g))
    dhn, dcn = torch.randn(2, B, H, generator=g).cuda(), torch.randn(2, B, H, generator=g).cuda()
    y, gates, cs, hn, cn = ref

    def bwd(bits):
        gg = gates.clone()
        dh0, dc0 = torch.zeros(2, B, H).cuda(), torch.zeros(2, B, H).cuda()
        dbias = torch.full(((B + 15) // 16, 2, 4 * H), float("nan")).cuda()
        twin = torch.zeros(rw.R, 8 * H, dtype=torch.bfloat16).cuda()
        ws = ops.lstm_bwd(dy, whh, gg, cs, c0, rw.lens, gg, dh0, dc0, T, B, H, 1 | bits, dhn=dhn, dcn=dcn, bf16=True, dbias=dbias,
                          dgx_bf16=twin, offs=rw.offs)
        ops.lstm_status(ws)
        return gg, twin, dh0, dc0, dbias
    bref = bwd(ops.lstm_variant_bits(False, 1, False, False, False, 31))
    for rep in range(2):
        out = bwd(ops.lstm_variant_bits(False, 1, False, False, False, 31, xl8=True))
        for a_, b_ in zip(out, bref):
            assert torch.equal(rows_of(a_), rows_of(b_))
    assert float(bref[2].abs().max()) > 0 and torch.isfinite(bref[4]).all()


@pytest.mark.parametrize("layout", ["padded", "packed"])
@pytest.mark.parametrize("T,B,H,lens,delay", [(12, 32, 896, [12] * 20 + [7] * 8 + [2] * 3 + [1], 0), (30, 16, 896, [30] * 9 + [17] * 7, 31),
                                              (9, 100, 600, [9] * 60 + [4] * 40, 4), (7, 20, 300, [7] * 7 + [4] * 13, 8), (11, 3, 64, [11, 5, 1], 0)])
def test_lstm_forward_tagged_hand_off(ops, layout, T, B, H, lens, delay):
    """Mode bit 29 (fp32 forward): the exchanged h carries the step's epoch in its two low mantissa bits, producers publish
    without drain / barrier / flag, consumers pull, check every word and pull again what was not there yet.  The product
    then runs on h with its two low bits replaced (3 ulp): results within 2e-6 of the flag protocol's, bit-reproducible from
    run to run (also with NO hold-back, where first pulls regularly come too early and are repeated), one launch per step
    (mode 2) bit-identical to the persistent launch, stale epochs of an earlier sequence in the workspace never accepted
    (T = 30 after T = 12 on one workspace: 30 % 4 == 2 is the colliding case without the zeroing)."""
    g = torch.Generator().manual_seed(11 * H + T)
    rw = _Rows(layout, T, B, lens)
    gx = rw.put(torch.randn(T, B, 2, 4 * H, generator=g) * 0.5)
    whh = (torch.randn(2, 4 * H, H, generator=g) / 30).cuda()
    h0, c0 = torch.randn(2, B, H, generator=g).cuda(), torch.randn(2, B, H, generator=g).cuda()

    def fwd(bits, mode=1):
        gg = gx.clone()
        y, cs = rw.new(2 * H, 0.0), rw.new(2 * H, 0.0)
        hn, cn = torch.zeros(2, B, H).cuda(), torch.zeros(2, B, H).cuda()
        ws = ops.lstm_fwd(gg, whh, h0, c0, rw.lens, y, gg, cs, hn, cn, T, B, H, mode | bits, offs=rw.offs)
        ops.lstm_status(ws)
        return y, gg, cs, hn, cn

    def rows_of(a):
        if a.dim() == 2 and a.shape[0] == rw.R and not rw.packed:
            return a.view(T, B, -1)[rw.valid]
        return a[:rw.pk.R] if (a.dim() == 2 and a.shape[0] == rw.R) else a

    ref = fwd(ops.lstm_variant_bits(False, 1, True, False, False, 0))
    tg = ops.lstm_variant_bits(False, 1, True, False, False, delay, tagged=True)
    out, again = fwd(tg), fwd(tg)
    steps = fwd(tg, mode=2)
    for a, b, c_, e in zip(out, ref, again, steps):
        a, b, c_, e = rows_of(a), rows_of(b), rows_of(c_), rows_of(e)
        assert torch.isfinite(a).all()
        assert float((a - b).abs().max()) < 2e-6
        assert torch.equal(a, c_) and torch.equal(a, e)


@pytest.mark.parametrize("bf16", [False, True])
@pytest.mark.parametrize("T,B,H,lens", [(9, 32, 896, [9] * 20 + [7] * 8 + [2] * 4), (6, 20, 304, [6] * 7 + [4] * 13), (5, 3, 64, [5, 3, 1])])
def test_lstm_backward_bf16_twin_of_dgx(ops, T, B, H, lens, bf16):
    """The backward recurrence also writes dgx as bf16 into a caller-supplied (rows, ld) matrix (sk_lstm_bwd, dgx_bf16).  The
    twin equals the RNE rounding of the fp32 dgx of the same launch entry by entry (padded layout: zero rows past a sequence's
    end included), what lies outside the rows x 8H block is not touched, and the fp32 results are those of the launch
    without a twin.  Packed rows: the same, on the R valid rows."""
    g = torch.Generator().manual_seed(7 * H + T)
    for layout in ("padded", "packed"):
        rw = _Rows(layout, T, B, lens)
        gates = rw.put(torch.randn(T, B, 2, 4 * H, generator=g) * 0.5)
        whh = (torch.randn(2, 4 * H, H, generator=g) / 30).cuda()
        h0, c0 = torch.randn(2, B, H, generator=g).cuda(), torch.randn(2, B, H, generator=g).cuda()
        dy = rw.put(torch.randn(T, B, 2 * H, generator=g))
        y, cs = rw.new(2 * H, 0.0), rw.new(2 * H, 0.0)
        ops.lstm_status(ops.lstm_fwd(gates, whh, h0, c0, rw.lens, y, gates, cs, None, None, T, B, H, 1, bf16=bf16, offs=rw.offs))
        R = rw.pk.R if rw.packed else T * B
        ld = 8 * H + 64
        outs = []
        for twin in (None, torch.full((R + 5, ld), 7.0, dtype=torch.bfloat16).cuda()):
            gg = gates.clone()
            dh0, dc0 = torch.zeros(2, B, H).cuda(), torch.zeros(2, B, H).cuda()
            ops.lstm_status(ops.lstm_bwd(dy, whh, gg, cs, c0, rw.lens, gg, dh0, dc0, T, B, H, 1, bf16=bf16, dgx_bf16=twin,
                                         offs=rw.offs))
            outs.append((gg, dh0, dc0))
        for a, b in zip(*outs):
            assert torch.equal(a, b)
        dgx = outs[1][0][:R]
        assert torch.equal(twin[:R, :8 * H], dgx.to(torch.bfloat16))
        assert float(dgx.abs().max()) > 0
        assert bool((twin[R:] == 7).all()) and bool((twin[:, 8 * H:] == 7).all())
        if not bf16:
            # fp32 (r06, operands that arrive split): dgx also as its THREE exact bf16 planes -- hi + mid + lo = dgx bit for bit,
            # pieces of the sizes the split promises, zero tail rows left alone, dgx itself unchanged
            pl = ops.Planes.empty(R, 8 * H, "cuda")
            gg = gates.clone()
            dh0, dc0 = torch.zeros(2, B, H).cuda(), torch.zeros(2, B, H).cuda()
            ops.lstm_status(ops.lstm_bwd(dy, whh, gg, cs, c0, rw.lens, gg, dh0, dc0, T, B, H, 1, dgx_bf16=pl, offs=rw.offs))
            assert torch.equal(gg, outs[0][0]) and torch.equal(dh0, outs[0][1])
            hi, mid, lo = (pl.t[i, :R].double() for i in range(3))
            assert torch.equal((hi + mid + lo)[:, :8 * H], dgx.double())
            assert bool((mid.abs() <= hi.abs() * 2.0 ** -8 + 1e-45).all()) and bool((lo.abs() <= hi.abs() * 2.0 ** -16 + 1e-45).all())
            assert bool((pl.t[:, R:] == 0).all())


@pytest.mark.parametrize("M,N,K,batch,splitk", [(512, 272, 1280, 1, 1), (514, 384, 2048, 1, 3), (256, 128, 1024, 2, 2), (300, 260, 64, 1, 1),
                                                (1024, 512, 12800, 1, 5), (3584, 896, 1600, 2, 0)])
def test_gemm_on_operands_that_arrive_split_is_bit_for_bit_the_split_kernel(ops, M, N, K, batch, splitk):
    """sk_split_rows + sk_gemm_pl3_tn (r06): the T/N product on operands cut ONCE into their three bf16 planes equals the 128 x 128
    split kernel (variant 2) on the fp32 operands bit for bit -- same pieces, same six products, same K order, same sign phases,
    same slab sums -- with ragged tile edges (M = 514: clamped edge reads), batches taken out of wider matrices (the two
    directions of dW_hh), K slices, accumulation; the planes reassemble the operand exactly and carry zero tails."""
    g = torch.Generator().manual_seed(M + 7 * N + K)
    R = K - 37 if K > 64 else K                                # valid rows: the planes' tail rows up to K are zero
    A = torch.zeros(K, batch * M)
    B = torch.zeros(K, batch * N)
    A[:R] = torch.randn(R, batch * M, generator=g) * torch.exp2(torch.randint(-12, 13, (R, batch * M), generator=g).float())
    B[:R] = torch.randn(R, batch * N, generator=g)
    Ad, Bd = dev(A), dev(B)
    Apl, Bpl = ops.split_rows(Ad, R), ops.split_rows(Bd, R)
    for pl, X in ((Apl, A), (Bpl, B)):
        assert pl.rows >= R and pl.rows % 64 == 0 and pl.ld % 8 == 0
        parts = pl.t.cpu().double()
        assert torch.equal(parts.sum(0)[:R, :X.shape[1]], X[:R].double())
        assert bool((parts[:, R:] == 0).all()) and bool((parts[:, :, X.shape[1]:] == 0).all())
    Kp = Apl.rows if K > 64 else K
    C0 = torch.randn(batch, M, N, generator=g)
    outs = []
    for planes in (True, False):
        for _ in range(2):
            C = dev(C0.clone())
            if planes:
                ops.gemm_pl3_tn(Apl, Bpl, C, M, N, Kp, accumulate=True, batch=batch, sA=M, sB=N, sC=M * N, splitk=splitk, ws_tag="t_pl3")
            else:
                Az, Bz = torch.zeros(Kp, batch * M).cuda(), torch.zeros(Kp, batch * N).cuda()
                Az[:K], Bz[:K] = Ad, Bd
                ops.gemm(Az, Bz, C, M, N, Kp, batch * M, batch * N, N, transA=True, accumulate=True, batch=batch, sA=M, sB=N, sC=M * N,
                         splitk=splitk, variant=2, ws_tag="t_pl3")
            torch.cuda.synchronize()
            outs.append(C.cpu())
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[2], outs[3])          # run-to-run
    ref = torch.stack([A[:, z * M:(z + 1) * M].double().t() @ B[:, z * N:(z + 1) * N].double() for z in range(batch)]) + C0.double()
    mag = torch.stack([A[:, z * M:(z + 1) * M].double().abs().t() @ B[:, z * N:(z + 1) * N].double().abs() for z in range(batch)])
    assert float(((outs[0].double() - ref).abs() / (mag + 1e-30)).max()) < 2.0 ** -23 * 4 * np.sqrt(K)
    if M % 4 == 0:                                            # (M = 514: the fp32 split kernel's LDS-DMA conditions do not hold, variant 2 falls back)
        assert torch.equal(outs[0], outs[2])


@pytest.mark.parametrize("H,B", [(896, 32), (600, 48), (304, 16)])
def test_lstm_forward_split_product_is_as_close_to_fp64_as_the_fp32_mfma_product(ops, H, B):
    """The split-3 forward product forms six of the nine piece products (those of relative size >= 2^-16; the other three are
    together at most 2^-23 of |w||h|: one ulp of that product in the worst case).  ONE recurrence step from a given h0 against the same cell
    computed in fp64 on the host: the split kernel's error is not larger than the fp32-MFMA kernel's (csrc/lstm.hip: rms
    2.2e-7 vs 2.5e-7 on the pre-activations at K = 896) -- it is an fp32 product, not a reduced-precision one.  The bf16-input
    kernel on the same data is two orders of magnitude further away."""
    g = torch.Generator().manual_seed(5 * H + B)
    T = 1
    gx = torch.randn(T, B, 2, 4 * H, generator=g) * 0.5
    whh = torch.randn(2, 4 * H, H, generator=g) / 30
    h0, c0 = torch.tanh(torch.randn(2, B, H, generator=g)), torch.randn(2, B, H, generator=g)
    lens = torch.full((B,), T, dtype=torch.int32).cuda()
    # fp64 cell on the host; the device keeps gx gate-interleaved (4u + g), whh in torch's gate-major order
    pre = gx.double().view(B, 2, H, 4) + torch.einsum("dbk,dgk->bdg", h0.double(), whh.double()).view(B, 2, 4, H).permute(0, 1, 3, 2)
    i_, f_, g_, o_ = torch.sigmoid(pre[..., 0]), torch.sigmoid(pre[..., 1]), torch.tanh(pre[..., 2]), torch.sigmoid(pre[..., 3])
    c_ref = f_ * c0.double().permute(1, 0, 2) + i_ * g_
    y_ref = (o_ * torch.tanh(c_ref)).reshape(B, 2 * H)

    def run(bits, bf16=False):
        gg = gx.reshape(T * B, 8 * H).cuda()
        y, cs = torch.zeros(T * B, 2 * H).cuda(), torch.zeros(T * B, 2 * H).cuda()
        ws = ops.lstm_fwd(gg, whh.cuda(), h0.cuda(), c0.cuda(), lens, y, gg, cs, None, None, T, B, H, 1 | bits, bf16=bf16)
        ops.lstm_status(ws)
        ey = (y.cpu().double() - y_ref)
        ec = (cs.cpu().double() - c_ref.reshape(B, 2 * H))
        return float((ey ** 2).mean().sqrt()), float((ec ** 2).mean().sqrt())
    mfma = run(ops.lstm_variant_bits(False, 1, True, False, False, 0))
    split = run(ops.lstm_variant_bits(False, 1, True, False, False, 0, split3=True))
    low = run(ops.lstm_variant_bits(False, 1, True, False, True, 0), bf16=True)
    assert split[0] <= 1.1 * mfma[0] and split[1] <= 1.1 * mfma[1], (split, mfma)
    assert mfma[0] < 1e-6 and low[0] > 30 * split[0], (mfma, split, low)


@pytest.mark.parametrize("layout", ["padded", "packed"])
@pytest.mark.parametrize("T,B,H,lens", [(12, 32, 896, [12] * 20 + [7] * 8 + [2] * 3 + [1]), (9, 100, 600, [9] * 60 + [4] * 40),
                                        (7, 20, 300, [7] * 7 + [4] * 13), (11, 3, 64, [11, 5, 1])])
def test_lstm_forward_exact_bf16_split_is_an_fp32_product(ops, layout, T, B, H, lens):
    """Mode bit 28 (fp32 forward): h W_hh^T by the exact three-way bf16 split of both operands on the bf16 matrix pipe -- the
    six piece products per element pair that are not below fp32's resolution of the product, fp32 accumulators: the fp32
    recurrence in another summation order.  Outputs agree with the fp32-MFMA kernel's to 4e-6 (as two fp32 summation orders
    do), are bit-reproducible, equal in per-step launch mode, and 50 x closer to it than the bf16-input recurrence is."""
    g = torch.Generator().manual_seed(13 * H + T)
    rw = _Rows(layout, T, B, lens)
    gx = rw.put(torch.randn(T, B, 2, 4 * H, generator=g) * 0.5)
    whh = (torch.randn(2, 4 * H, H, generator=g) / 30).cuda()
    h0, c0 = torch.randn(2, B, H, generator=g).cuda(), torch.randn(2, B, H, generator=g).cuda()

    def fwd(bits, mode=1, bf16=False):
        gg = gx.clone()
        y, cs = rw.new(2 * H, 0.0), rw.new(2 * H, 0.0)
        hn, cn = torch.zeros(2, B, H).cuda(), torch.zeros(2, B, H).cuda()
        ws = ops.lstm_fwd(gg, whh, h0, c0, rw.lens, y, gg, cs, hn, cn, T, B, H, mode | bits, offs=rw.offs, bf16=bf16)
        ops.lstm_status(ws)
        return y, gg, cs, hn, cn

    def rows_of(a):
        if a.dim() == 2 and a.shape[0] == rw.R and not rw.packed:
            return a.view(T, B, -1)[rw.valid]
        return a[:rw.pk.R] if (a.dim() == 2 and a.shape[0] == rw.R) else a

    ref = fwd(ops.lstm_variant_bits(False, 1, True, False, False, 0))
    s3 = ops.lstm_variant_bits(False, 1, True, False, False, 0, split3=True)
    out, again, steps = fwd(s3), fwd(s3), fwd(s3, mode=2)
    low = fwd(ops.lstm_variant_bits(False, 1, True, False, True, 0), bf16=True)
    for a, b, c_, e, lo in zip(out, ref, again, steps, low):
        a, b, c_, e, lo = rows_of(a), rows_of(b), rows_of(c_), rows_of(e), rows_of(lo)
        assert torch.isfinite(a).all()
        err = float((a - b).abs().max())
        assert err < 4e-6, err             # (gate pre-activations are sums of up to 1024 products of magnitude ~1: 2e-6 is a few ulp)
        assert torch.equal(a, c_) and torch.equal(a, e)
        if H >= 300:
            assert err * 50 < float((lo - b).abs().max())
