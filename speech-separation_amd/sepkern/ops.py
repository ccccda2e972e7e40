"""Tensor-level wrappers over the C ABI (include/sepkern.h).

PyTorch supplies device memory and the current HIP stream; all arithmetic is in libsepkern.so.
Every wrapper takes fp32 CUDA tensors and passes raw device pointers.
"""
import ctypes as C

import torch

from . import _lib

_WS = {}

# Optional per-kernel-class timing with HIP events recorded on the launch stream (bench.py):
# PROF = {} enables it; each timed call appends (class, start_event, end_event, flops).
PROF = None


class _timed:
    def __init__(self, cls, flops=0.0):
        self.cls, self.flops = cls, flops

    def __enter__(self):
        if PROF is not None:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e1 = torch.cuda.Event(enable_timing=True)
            self.e0.record()
        return self

    def __exit__(self, *exc):
        if PROF is not None:
            self.e1.record()
            # launches on a side stream are co-scheduled with other kernels: their durations are kept apart
            side = torch.cuda.current_stream() != torch.cuda.default_stream()
            PROF.setdefault(self.cls + ("@side" if side else ""), []).append((self.e0, self.e1, self.flops))
        return False


def prof_summary():
    """{class: (launches, total_ms, total_flops)} -- call after torch.cuda.synchronize()."""
    out = {}
    for cls, recs in (PROF or {}).items():
        out[cls] = (len(recs), sum(a.elapsed_time(b) for a, b, _ in recs), sum(f for _, _, f in recs))
    return out


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _chk(t, dtype=torch.float32):
    if t is None:
        return
    if not t.is_cuda:
        raise _lib.SepkernError("sepkern ops need CUDA(HIP) tensors; got a CPU tensor (there is no CPU path)")
    if t.dtype != dtype:
        raise _lib.SepkernError("expected dtype %s, got %s" % (dtype, t.dtype))


def workspace(nbytes, tag="default"):
    """A cached per-(device, tag) scratch buffer of at least nbytes (owned by torch's allocator)."""
    key = (torch.cuda.current_device(), tag)
    ws = _WS.get(key)
    if ws is None or ws.numel() < nbytes:
        # zeroed: the LSTM workspace starts with a sticky status word that no launch clears (include/sepkern.h) ...
        old, ws = ws, torch.zeros(max(int(nbytes), 256), dtype=torch.uint8, device="cuda")
        if old is not None and tag == "lstm":
            ws[:4].copy_(old[:4])      # ... and a larger request must not lose it (RSH passes of different (T, B))
        _WS[key] = ws
    return ws


def device_info():
    ncu, lds = C.c_int(0), C.c_int(0)
    _lib.call("sk_device_info", C.byref(ncu), C.byref(lds))
    return ncu.value, lds.value


# ----------------------------------------------------------------------------- GEMM
_NUM_CUS = None
_STREAMK_BF16 = __import__('os').environ.get('SEPKERN_BF16_STREAMK', '1') != '0'
_STREAMK = __import__('os').environ.get('SEPKERN_GEMM_STREAMK', '1') != '0'    # variant 0 may choose the stream-K kernel
_SPLITK_MAX = int(__import__('os').environ.get('SEPKERN_SPLITK_MAX', '32'))   # diagnostic: cap the K slices


def pick_splitk(M, N, K, batch=1):
    """K slices for a product with few output tiles and a long K (weight gradients; the N = 2H data gradient):
    the matrix pipe of a CU is saturated by its resident 128x128 blocks, so time goes with the LARGEST number
    of blocks any CU gets, ceil(blocks / CUs); choose the slice count that minimises that quantisation loss
    plus the cost of writing and re-reading the partial slabs (measured: 1400 tiles on 256 CUs run at 91 %)."""
    global _NUM_CUS
    if _NUM_CUS is None:
        _NUM_CUS = device_info()[0]
    cus = _NUM_CUS
    tiles = ((M + 127) // 128) * ((N + 127) // 128) * batch
    if tiles >= 16 * cus or K < 2048 or _SPLITK_MAX == 1:
        return 1
    work = 2.0 * M * N * K * batch / 140e12                      # seconds at the kernel's un-quantised rate
    best, best_t = 1, None
    for s in range(1, min(33, _SPLITK_MAX + 1)):
        if K // s < 512:
            break
        per_cu = tiles * s / cus
        eff = per_cu / -(-(tiles * s) // cus) * min(1.0, 0.55 + 0.15 * min(per_cu, 3.0))   # <3 blocks/CU: poor overlap
        t = work / eff + (0 if s == 1 else 2.0 * s * M * N * batch * 4 / 4e12)
        if best_t is None or t < best_t * 0.995:
            best, best_t = s, t
    return best


def gemm(A, B, Cout, M, N, K, lda, ldb, ldc, transA=False, transB=False, bias=None, accumulate=False, act=0,
         batch=1, sA=0, sB=0, sC=0, sbias=0, splitk=1, ws_tag="gemm", bf16=False, variant=0):
    """Cout[M,N] = act(opA(A) opB(B) + bias (+ Cout)).  A/B/Cout are tensors whose data_ptr() is the
    first element of the operand (views are fine: leading dimensions are explicit).  splitk > 1 (or 0 =
    choose) splits K into deterministic partial slabs -- for weight gradients.  bf16=True rounds A and B to
    bf16 on the way into the matrix cores (fp32 accumulate; everything in memory stays fp32).  variant (fp32 only,
    sk_gemm_f32_splitk's `variant`): 0 choose -- the three-way bf16 split of both operands on the bf16 matrix pipe (six piece
    products per element pair: fp32 products in another summation order) wherever the operands are aligned, else the
    fp32-MFMA kernels; 1 the register-staged fp32-MFMA kernel; 2 / 9 the 128 x 128 / 256 x 128 split-once-while-staging split
    kernels; 3 / 4 / 6 the 128 x 128 / 256 x 128 / stream-K 256 x 256 fp32-MFMA LDS-DMA kernels; 8 choose among the fp32-MFMA
    kernels only (the reference's literal arithmetic; SEPKERN_GEMM_SPLIT=0 makes 0 mean this)."""
    for t in (A, B, Cout, bias):
        _chk(t)
    if splitk == 0:
        splitk = pick_splitk(M, N, K, batch)
    ws = None
    if splitk > 1:
        ws = workspace(_lib.load().sk_gemm_workspace_bytes(M, N, batch, splitk), ws_tag)
    elif not bf16 and batch == 1 and (variant == 6 or (_STREAMK and M >= 4096 and N >= 1024 and
                                                (variant == 0 or (variant == 8 and not transA)))):
        ws = _streamk_ws()                                                              # pieces of the stream-K cut
    with _timed("gemm_bf16_kernel" if bf16 else "gemm_f32_kernel", 2.0 * M * N * K * batch) as rec:
        args = (_ptr(A), _ptr(B), _ptr(Cout), _ptr(bias), M, N, K, lda, ldb, ldc, int(transA), int(transB), int(accumulate),
                int(act), batch, sA, sB, sC, sbias, int(splitk), _ptr(ws))
        if bf16:
            _lib.call("sk_gemm_bf16_splitk", *args, _stream())
        else:
            _lib.call("sk_gemm_f32_splitk", *args, int(variant), _stream())
            if PROF is not None and _lib.load().sk_gemm_last_kernel() in (2, 10):
                # the launch ran on the bf16 matrix pipe (split products): its own class -- another pipe, another peak
                rec.cls = "gemm_f32_split_kernel"


def _streamk_ws():
    """The stream-K kernels' workspace (ticket counters + piece slabs, 134 MB): ONE per stream -- launches on a stream run
    in order, so they can share it; two streams never do (their tickets and slabs would mix)."""
    return workspace(_lib.load().sk_gemm_streamk_workspace_bytes(), "streamk_%x" % torch.cuda.current_stream().cuda_stream)


def pad_to(n, m):
    return (n + m - 1) // m * m


def cast_bf16(x2d, ld=None, out=None, rows=None):
    """bf16 copy of an fp32 (R, C) matrix (row stride x2d.stride(0)), leading dimension ld >= C (default: C rounded
    up to 64), extra columns zero; rows >= R: that many rows are written, those past R zero (a copy that also serves as
    a K-major factor, gemm_bf16_mm).  Returns the (rows or R, ld) bfloat16 tensor."""
    _chk(x2d)
    R, Cc = x2d.shape
    ld = pad_to(Cc, 64) if ld is None else ld
    rows = R if rows is None else rows
    if out is None:
        out = torch.empty(rows, ld, dtype=torch.bfloat16, device=x2d.device)
    _lib.call("sk_cast_bf16_rows", _ptr(x2d), R, Cc, x2d.stride(0), _ptr(out), ld, rows, _stream())
    return out


class Planes:
    """The three bf16 planes (hi, mid, lo pieces: x = hi + mid + lo exactly) of an fp32 (R, C) matrix: t is a (3, rows, ld)
    bfloat16 tensor, ld = C rounded up to 8, rows >= R rounded up to 64 with zero tail rows ("operands that arrive split",
    include/sepkern.h).  Made by split_rows() or by lstm_bwd(dgx_bf16=Planes.empty(...))."""

    def __init__(self, t, R, C):
        self.t, self.R, self.C = t, R, C
        self.rows, self.ld = t.shape[1], t.shape[2]
        self.plane = t.stride(0)

    @staticmethod
    def empty(R, C, device, zero_tail=True):
        rows, ld = pad_to(R, 64), pad_to(C, 8)
        t = torch.empty(3, rows + 1, ld, dtype=torch.bfloat16, device=device)[:, :rows]     # (+1 row: a clamped edge tile may read past the last row's end)
        if zero_tail and (rows > R or ld > C):
            t[:, R:].zero_()
            if ld > C:
                t[:, :, C:].zero_()
        return Planes(t, R, C)


def split_rows(x2d, R=None):
    """Planes of the first R rows of an fp32 (>= R, C) matrix (sk_split_rows: one pass, 4 bytes read and 6 written per element)."""
    _chk(x2d)
    R = x2d.shape[0] if R is None else R
    Cc = x2d.shape[1]
    pl = Planes.empty(R, Cc, x2d.device, zero_tail=False)
    with _timed("split_rows_kernel", 0.0):
        _lib.call("sk_split_rows", _ptr(x2d), R, Cc, x2d.stride(0), _ptr(pl.t), pl.ld, pl.rows, pl.plane, _stream())
    return pl


def gemm_pl3_tn(A, B, Cout, M, N, K, accumulate=False, batch=1, sA=0, sB=0, sC=0, splitk=1, ws_tag="gemm"):
    """Cout[M, N] (+)= A^T B with A, B Planes whose rows are the contraction index (K <= their row counts, K % 16 == 0): the weight
    gradients on operands that arrive split (sk_gemm_pl3_tn).  batch / strides (in columns) / splitk (0 = choose) as gemm()."""
    _chk(Cout)
    if splitk == 0:
        splitk = pick_splitk(M, N, K, batch)
    ws = workspace(_lib.load().sk_gemm_workspace_bytes(M, N, batch, splitk), ws_tag) if splitk > 1 else None
    with _timed("gemm_f32_split_kernel", 2.0 * M * N * K * batch):
        _lib.call("sk_gemm_pl3_tn", _ptr(A.t), _ptr(B.t), _ptr(Cout), M, N, K, A.ld, B.ld, Cout.stride(-2), A.plane, B.plane,
                  int(accumulate), batch, sA, sB, sC, int(splitk), _ptr(ws), _stream())


def gemm_bf16_nt(A, B, Cout, M, N, K, lda, ldb, ldc, bias=None, accumulate=False, act=0, batch=1, sA=0, sB=0, sC=0,
                 sbias=0, splitk=1, ws_tag="gemm", streamk=False):
    """Cout[M,N] = act(A[M,K] B[N,K]^T + bias (+ Cout)) with A, B bfloat16 tensors (K-contiguous, K % 64 == 0).
    streamk: as gemm_bf16_mm."""
    _chk(A, torch.bfloat16)
    _chk(B, torch.bfloat16)
    _chk(Cout)
    _chk(bias)
    ws = None
    if streamk and _STREAMK_BF16 and batch == 1 and splitk in (0, 1) and M >= 256 and (N % 256 == 0 or N > 1024) and K >= 512:
        splitk = 1
        ws = _streamk_ws()
    if splitk == 0:
        splitk = pick_splitk_bf16(M, N, K, batch)
    if splitk > 1:
        ws = workspace(_lib.load().sk_gemm_workspace_bytes(M, N, batch, splitk), ws_tag)
    with _timed("gemm_bf16_nt_kernel", 2.0 * M * N * K * batch):
        _lib.call("sk_gemm_bf16_nt", _ptr(A), _ptr(B), _ptr(Cout), _ptr(bias), M, N, K, lda, ldb, ldc, int(accumulate),
                  int(act), batch, sA, sB, sC, sbias, int(splitk), _ptr(ws), _stream())


def gemm_bf16_mm(A, B, Cout, M, N, K, lda, ldb, ldc, a_kmajor=False, b_kmajor=False, bias=None, accumulate=False, act=0, batch=1,
                 sA=0, sB=0, sC=0, sbias=0, splitk=1, ws_tag="gemm", streamk=False):
    """Cout[M,N] = act(opA opB + bias (+ Cout)) on bfloat16 operands in memory, either of them optionally K-MAJOR
    (a_kmajor: A stored [K][M] with lda elements between k rows; b_kmajor: B stored [K][N]) -- sk_gemm_bf16_mm.  Row-major
    operands: K-contiguous as in gemm_bf16_nt.  K % 64 == 0.  streamk=True (unbatched products that have the chip to
    themselves): the persistent stream-K kernel instead of K slices where it applies (SEPKERN_BF16_STREAMK=0: never)."""
    _chk(A, torch.bfloat16)
    _chk(B, torch.bfloat16)
    _chk(Cout)
    _chk(bias)
    ws = None
    if streamk and _STREAMK_BF16 and batch == 1 and splitk in (0, 1) and M >= 256 and (N % 256 == 0 or N > 1024) and K >= 512:
        splitk = 1
        ws = _streamk_ws()
    if splitk == 0:
        splitk = pick_splitk_bf16(M, N, K, batch)
    if splitk > 1:
        ws = workspace(_lib.load().sk_gemm_workspace_bytes(M, N, batch, splitk), ws_tag)
    with _timed("gemm_bf16_nt_kernel", 2.0 * M * N * K * batch):
        _lib.call("sk_gemm_bf16_mm", _ptr(A), _ptr(B), _ptr(Cout), _ptr(bias), M, N, K, lda, ldb, ldc, int(a_kmajor), int(b_kmajor),
                  int(accumulate), int(act), batch, sA, sB, sC, sbias, int(splitk), _ptr(ws), _stream())


def pick_splitk_bf16(M, N, K, batch=1):
    """K slices for the 256 x 256-tile bf16 kernel (one block per CU): enough blocks to fill the chip about twice,
    slices of at least 1024."""
    global _NUM_CUS
    if _NUM_CUS is None:
        _NUM_CUS = device_info()[0]
    tiles = ((M + 255) // 256) * ((N + 255) // 256) * batch
    best, best_t = 1, None
    for s in range(1, 17):
        if K // s < 1024 and s > 1:
            break
        rounds = -(-(tiles * s) // _NUM_CUS)
        t = rounds / s + (0.0 if s == 1 else 0.02 * s)       # time ~ rounds x K/s, plus the slab round trip
        if best_t is None or t < best_t * 0.98:
            best, best_t = s, t
    return best


# ----------------------------------------------------------------------------- STFT / iSTFT
def _i64(vals, device):
    return torch.tensor(vals, dtype=torch.int64, device=device)


def stft_batch(wavs, want_complex=False, layout="TF", out=None, out_offs=None, stride_t=None, stride_f=None, lengths=None, repeat=1):
    """STFT (n_fft 512, hop 128, reflect-centred, periodic Hann) of a list of 1-D waveforms.

    wavs: list of 1-D CUDA tensors, float32 in [-1,1) or int16 PCM (scaled by 1/32768 in-kernel) -- or, with
    `lengths` (samples per utterance), ONE 1-D CUDA tensor holding the utterances back to back (a batch that crossed
    PCIe as one copy).
    layout "TF": returns list of (T_u, 257) tensors; "FT": list of (257, T_u) (the reference's npz layout).
    With `out` given, writes element (t,f) of utterance u at out_offs[u] + t*stride_t[u] + f*stride_f[u].
    repeat (bench.py's aux leg): the launch is enqueued that many times back to back (same result) between the profile's events.
    """
    if lengths is not None:
        cat = wavs.contiguous()
        ns = [int(n) for n in lengths]
        dev, pcm16 = cat.device, cat.dtype == torch.int16
        _chk(cat, torch.int16 if pcm16 else torch.float32)
        if cat.dim() != 1 or sum(ns) != cat.numel() or min(ns) <= 256:
            raise _lib.SepkernError("stft needs 1-D waveforms longer than n_fft/2 samples (and lengths that add up)")
    else:
        dev = wavs[0].device
        pcm16 = wavs[0].dtype == torch.int16
        for w in wavs:
            _chk(w, torch.int16 if pcm16 else torch.float32)
            if w.dim() != 1 or w.numel() <= 256:
                raise _lib.SepkernError("stft needs 1-D waveforms longer than n_fft/2 samples")
        ns = [int(w.numel()) for w in wavs]
        cat = torch.cat(wavs) if len(wavs) > 1 else wavs[0].contiguous()
    Ts = [1 + n // 128 for n in ns]
    woffs, acc = [], 0
    for n in ns:
        woffs.append(acc)
        acc += n
    F = 257
    ret = None
    if out is None:
        odt = torch.complex64 if want_complex else torch.float32
        total = sum(Ts) * F
        out = torch.empty(total, dtype=odt, device=dev)
        out_offs, stride_t, stride_f, acc = [], [], [], 0
        for T in Ts:
            out_offs.append(acc)
            stride_t.append(F if layout == "TF" else 1)
            stride_f.append(1 if layout == "TF" else T)
            acc += T * F
        ret = [out[o:o + T * F].view((T, F) if layout == "TF" else (F, T)) for o, T in zip(out_offs, Ts)]
    # descriptor arrays must outlive the (asynchronous) launch call: keep references until it returns
    d_woffs, d_ns = _i64(woffs, dev), torch.tensor(ns, dtype=torch.int32, device=dev)
    d_ooffs, d_st, d_sf = _i64(out_offs, dev), _i64(stride_t, dev), _i64(stride_f, dev)
    frame_major = all(int(v) == 1 for v in stride_f)
    # algorithmic bytes: 128 new samples in, 257 bins out per frame
    with _timed("stft_kernel", repeat * float(sum(Ts)) * (128 * (2 if pcm16 else 4) + 257 * (8 if want_complex else 4))):
        for _ in range(repeat):
            _lib.call("sk_stft", _ptr(cat), int(pcm16), _ptr(d_woffs), _ptr(d_ns), len(ns), 512, 128, int(want_complex),
                      _ptr(out), _ptr(d_ooffs), _ptr(d_st), _ptr(d_sf), int(frame_major), max(Ts), _stream())
    return ret if ret is not None else out


def mask_istft_flat(mixcat, maskcat, Ts, S, want_pcm=True, want_float=True, repeat=1):
    """Mask-apply + iSTFT on buffers that crossed PCIe as ONE copy each: mixcat = the utterances' (257, T_u) complex64
    spectra back to back (flattened), maskcat = None or, per utterance and source (utterance-major), the (257, T_u) float32
    masks back to back.  Returns (wav float32 flat or None, pcm int16 flat or None, offsets): source s of utterance u is
    the 128 (T_u - 1) samples at offsets[u * S + s]."""
    dev = mixcat.device
    nutt, F = len(Ts), 257
    _chk(mixcat, torch.complex64)
    _chk(maskcat)
    if mixcat.numel() != F * sum(Ts) or (maskcat is not None and maskcat.numel() != F * S * sum(Ts)):
        raise _lib.SepkernError("mask_istft: buffer sizes do not match the frame counts")
    moffs, koffs, ooffs, am, ak, ao = [], [], [], 0, 0, 0
    for T in Ts:
        moffs.append(am)
        am += T * F
        for s in range(S):
            koffs.append(ak)
            ak += T * F
            ooffs.append(ao)
            ao += 128 * (T - 1)
    wav = torch.empty(ao, dtype=torch.float32, device=dev) if want_float else None
    pcm = torch.empty(ao, dtype=torch.int16, device=dev) if want_pcm else None
    d_moffs, d_mst, d_msf = _i64(moffs, dev), _i64([1] * nutt, dev), _i64(list(Ts), dev)
    d_koffs = d_kst = d_ksf = None
    if maskcat is not None:
        d_koffs, d_kst, d_ksf = _i64(koffs, dev), d_mst, d_msf
    d_T, d_ooffs = torch.tensor(list(Ts), dtype=torch.int32, device=dev), _i64(ooffs, dev)
    # algorithmic bytes per frame and source: the complex spectrum (read once per source), the mask, 128 samples out
    per = 257 * 8 + (257 * 4 if maskcat is not None else 0) + 128 * ((2 if want_pcm else 0) + (4 if want_float else 0))
    with _timed("istft_kernel", repeat * float(sum(Ts)) * S * per):
        for _ in range(repeat):
            _lib.call("sk_mask_istft", _ptr(mixcat), _ptr(d_moffs), _ptr(d_mst), _ptr(d_msf),
                      _ptr(maskcat), _ptr(d_koffs), _ptr(d_kst), _ptr(d_ksf),
                      _ptr(d_T), nutt, S, 512, 128, _ptr(wav), _ptr(pcm), _ptr(d_ooffs), max(Ts), _stream())
    return wav, pcm, ooffs


def mask_istft(mix_specs, masks=None, want_pcm=True, want_float=True, repeat=1):
    """Mask-apply + iSTFT.  mix_specs: list of (257, T_u) complex64 CUDA tensors (the reference's
    feats_test layout); masks: None or list (per utterance) of lists (per source) of (257, T_u) float32.
    Returns (list of lists of float32 waveforms or None, list of lists of int16 waveforms or None)."""
    nutt = len(mix_specs)
    S = len(masks[0]) if masks is not None else 1
    Ts = [int(m.shape[1]) for m in mix_specs]
    for m in mix_specs:
        _chk(m, torch.complex64)
        if m.shape[0] != 257:
            raise _lib.SepkernError("mask_istft expects (257, T) spectra")
    mixcat = torch.cat([m.contiguous().view(-1) for m in mix_specs])
    maskcat = None
    if masks is not None:
        for u in range(nutt):
            for s in range(S):
                _chk(masks[u][s])
        maskcat = torch.cat([masks[u][s].contiguous().view(-1) for u in range(nutt) for s in range(S)])
    wav, pcm, ooffs = mask_istft_flat(mixcat, maskcat, Ts, S, want_pcm, want_float, repeat=repeat)

    def split(buf):
        if buf is None:
            return None
        return [[buf[ooffs[u * S + s]:ooffs[u * S + s] + 128 * (Ts[u] - 1)] for s in range(S)] for u in range(nutt)]
    return split(wav), split(pcm)


# ----------------------------------------------------------------------------- PIT-MSE
def pit_mse_fwd(mask, mix, srcs, lens, norm_dev=None, packing=None, repeat=1):
    """mask (T,B,S*F), mix (T,B,F), srcs list of S (T,B,F), lens int32 (B), norm_dev: optional device
    scalar replacing sum(lens)*F (the global norm under data parallelism) ->
    dict(out (3,), pair (B,S,S), perm_loss (S!,B), best_perm (B)).
    packing (sepkern.packing.Packing): mask (>= R, S*F), mix and srcs (>= R, F) are PACKED rows (PackedSequence.data)."""
    if packing is not None:
        T, B, F = packing.T, packing.B, mix.shape[1]
        lens = packing.lens
    else:
        T, B, F = mix.shape
    S = len(srcs)
    for t in [mask, mix] + list(srcs):
        _chk(t)
        if not t.is_contiguous():
            raise _lib.SepkernError("pit_mse needs contiguous tensors")
    _chk(lens, torch.int32)
    nperm = 1
    for i in range(2, S + 1):
        nperm *= i
    dev = mix.device
    pair = torch.empty(B, S, S, device=dev)
    perm_loss = torch.empty(nperm, B, device=dev)
    best = torch.empty(B, dtype=torch.int32, device=dev)
    out = torch.empty(3, device=dev)
    ws = workspace(_lib.load().sk_pit_workspace_bytes(T, B, S), "pit")
    sp = (C.c_void_p * S)(*[s.data_ptr() for s in srcs])
    _chk(norm_dev)
    rows = packing.R if packing is not None else T * B
    with _timed("pit_fwd", repeat * float(rows) * (2 * S + 1) * F * 4):       # algorithmic bytes: mask, mixture, S sources
        for _ in range(repeat):
            _lib.call("sk_pit_mse_fwd", _ptr(mask), _ptr(mix), sp, _ptr(lens), _ptr(packing.offs) if packing is not None else None,
                      T, B, F, S, _ptr(norm_dev), _ptr(pair), _ptr(perm_loss), _ptr(best), _ptr(out), _ptr(ws), _stream())
    return dict(out=out, pair=pair, perm_loss=perm_loss, best_perm=best)


def pit_mse_bwd(mask, mix, srcs, best_perm, out, gscale, packing=None, repeat=1):
    S = len(srcs)
    dmask = torch.empty_like(mask)
    sp = (C.c_void_p * S)(*[s.data_ptr() for s in srcs])
    _chk(gscale)
    if packing is not None:
        T, B, F, R, offs = packing.T, packing.B, mix.shape[1], packing.R, packing.offs
        if mask.shape[0] > R:
            dmask[R:].zero_()            # tail rows of an (Rp, .) buffer stay zero
    else:
        (T, B, F), R, offs = mix.shape, 0, None
    rows = R if packing is not None else T * B
    with _timed("pit_bwd", repeat * float(rows) * (3 * S + 1) * F * 4):        # the forward's operands + dmask written
        for _ in range(repeat):
            _lib.call("sk_pit_mse_bwd", _ptr(mask), _ptr(mix), sp, _ptr(best_perm), _ptr(out), _ptr(gscale), _ptr(offs), R, T, B, F, S,
                      _ptr(dmask), _stream())
    return dmask


# ----------------------------------------------------------------------------- RSH loss / attention
def rsh_loss_fwd(mask, x, srcs, lens, used):
    """One greedy-assignment pass.  mask (T,B,F), x (T,B,2F) [mixture | attention], srcs list of S (T,B,F),
    used (S,B) int32 (updated in place) -> dict(out (2,) = [loss term, norm term], sse (S,B), sel (B))."""
    T, B, F = mask.shape
    S = len(srcs)
    for t in [mask, x] + list(srcs):
        _chk(t)
        if not t.is_contiguous():
            raise _lib.SepkernError("rsh_loss needs contiguous tensors")
    _chk(lens, torch.int32)
    _chk(used, torch.int32)
    dev = mask.device
    sse = torch.empty(S, B, device=dev)
    sel = torch.empty(B, dtype=torch.int32, device=dev)
    out = torch.empty(2, device=dev)
    ws = workspace(_lib.load().sk_rsh_workspace_bytes(T, B, S), "rsh")
    sp = (C.c_void_p * S)(*[s.data_ptr() for s in srcs])
    _lib.call("sk_rsh_loss_fwd", _ptr(mask), _ptr(x), x.shape[2], sp, _ptr(lens), T, B, F, S, _ptr(used), _ptr(sse),
              _ptr(sel), _ptr(out), _ptr(ws), _stream())
    return dict(out=out, sse=sse, sel=sel)


def rsh_loss_bwd(mask, x, srcs, sel, gscale):
    T, B, F = mask.shape
    S = len(srcs)
    dmask = torch.empty_like(mask)
    sp = (C.c_void_p * S)(*[s.data_ptr() for s in srcs])
    _chk(gscale)
    _lib.call("sk_rsh_loss_bwd", _ptr(mask), _ptr(x), x.shape[2], sp, _ptr(sel), _ptr(gscale), T, B, F, S, _ptr(dmask),
              _stream())
    return dmask


def att_update(x, mask, relu):
    out = torch.empty_like(x)
    F = mask.shape[-1]
    _lib.call("sk_att_update", _ptr(x), _ptr(mask), _ptr(out), x.numel() // (2 * F), F, int(relu), _stream())
    return out


def att_update_bwd(dx_out, x_out, F, relu):
    dx_in = torch.empty_like(dx_out)
    dmask = torch.empty(dx_out.shape[:-1] + (F,), device=dx_out.device)
    _lib.call("sk_att_update_bwd", _ptr(dx_out), _ptr(x_out), _ptr(dx_in), _ptr(dmask), dx_out.numel() // (2 * F), F,
              int(relu), _stream())
    return dx_in, dmask


# ----------------------------------------------------------------------------- BN / column ops
def bn_ws(R, Ccols, tag="bn"):
    return workspace(_lib.load().sk_bn_workspace_bytes(R, Ccols), tag)


def bn_stats(x2d, mean, var, rows=None, count=None):
    """mean / biased variance per column over `count` rows of which the first `rows` of x2d are stored and the rest are
    zero rows that are not (packed sequences: count = B * T_max); defaults: all of x2d's rows, count = rows."""
    R, Cc = x2d.shape
    R = R if rows is None else rows
    _lib.call("sk_bn_stats", _ptr(x2d), R, Cc, int(R if count is None else count), _ptr(mean), _ptr(var), _ptr(bn_ws(R, Cc)),
              _stream())


def bn_update_running(mean, var, rmean, rvar, count, momentum, guard=None):
    """guard: optional device word (the recurrence's sticky status, lstm_sticky): non-zero = leave the running statistics."""
    _lib.call("sk_bn_update_running", _ptr(mean), _ptr(var), _ptr(rmean), _ptr(rvar), int(count), mean.numel(), float(momentum),
              _ptr(guard), _stream())


def bn_apply(x2d, mean, var, gamma, beta, out, eps):
    R, Cc = x2d.shape
    _lib.call("sk_bn_apply", _ptr(x2d), _ptr(mean), _ptr(var), _ptr(gamma), _ptr(beta), _ptr(out), R, Cc, float(eps),
              _stream())


def bn_fold(W, b, mean, var, gamma, beta, eps, ld=None):
    """BatchNorm folded into the Linear layer behind it (sk_bn_fold): returns (Wf (O, ld), bf (O), s (C), t (C)) with
    lin(bn(x)) = x Wf[:, :C]^T + bf."""
    for t_ in (W, b, mean, var, gamma, beta):
        _chk(t_)
    O, Cc = W.shape
    ld = Cc if ld is None else ld
    Wf = torch.empty(O, ld, device=W.device)
    bf, s, t = torch.empty(O, device=W.device), torch.empty(Cc, device=W.device), torch.empty(Cc, device=W.device)
    _lib.call("sk_bn_fold", _ptr(W), _ptr(b), _ptr(mean), _ptr(var), _ptr(gamma), _ptr(beta), float(eps), O, Cc, _ptr(Wf), ld,
              _ptr(bf), _ptr(s), _ptr(t), _stream())
    return Wf, bf, s, t


def bn_unfold_grad(G, dzsum, s, t, dW, accumulate=False):
    """dW (O, C) (+)= G[:, :C] diag(s) + dzsum t^T (sk_bn_unfold_grad): the gradient of the unfolded Linear weight from
    G = dz^T x, the product against the raw (un-normalised) activations."""
    for t_ in (G, dzsum, s, t, dW):
        _chk(t_)
    O, Cc = dW.shape
    _lib.call("sk_bn_unfold_grad", _ptr(G), G.stride(0), _ptr(dzsum), _ptr(s), _ptr(t), _ptr(dW), O, Cc, int(accumulate), _stream())


def bn_bwd(dout, x2d, mean, var, gamma, dx, dgamma, dbeta, eps):
    R, Cc = x2d.shape
    _lib.call("sk_bn_bwd", _ptr(dout), _ptr(x2d), _ptr(mean), _ptr(var), _ptr(gamma), _ptr(dx), _ptr(dgamma),
              _ptr(dbeta), _ptr(bn_ws(R, Cc)), R, Cc, float(eps), _stream())


def bn_bwd_sums(dout, x2d, mean, var, dgamma, dbeta, eps):
    R, Cc = x2d.shape
    _lib.call("sk_bn_bwd_sums", _ptr(dout), _ptr(x2d), _ptr(mean), _ptr(var), _ptr(dgamma), _ptr(dbeta),
              _ptr(bn_ws(R, Cc)), R, Cc, float(eps), _stream())


def bn_bwd_apply(dout, x2d, mean, var, gamma, dgamma, dbeta, dx, count, eps):
    R, Cc = x2d.shape
    _lib.call("sk_bn_bwd_apply", _ptr(dout), _ptr(x2d), _ptr(mean), _ptr(var), _ptr(gamma), _ptr(dgamma), _ptr(dbeta),
              _ptr(dx), R, Cc, float(count), float(eps), _stream())


def colsum(x, R, Ccols, ld, out, accumulate=False, ws_tag="bn"):
    _lib.call("sk_colsum", _ptr(x), R, Ccols, ld, _ptr(out), int(accumulate), _ptr(bn_ws(R, Ccols, ws_tag)), _stream())


def pad_rows(x2d, ld, rows=None):
    """(rows, ld) copy of an (R, C) matrix with zero columns C..ld-1 and zero rows R..rows-1, one pass (sk_pad_rows)."""
    _chk(x2d)
    R, Cc = x2d.shape
    rows = R if rows is None else rows
    out = torch.empty(rows, ld, device=x2d.device)
    _lib.call("sk_pad_rows", _ptr(x2d), R, Cc, x2d.stride(0), _ptr(out), ld, rows, _stream())
    return out


# ----------------------------------------------------------------------------- packed rows
def pack_rows(padded, pk, out):
    """(T, B, C) zero-padded (caller's utterance order) -> the first R rows of out (>= R, ld): packed rows."""
    _chk(padded)
    _chk(out)
    _, B, Cc = padded.shape                  # (frames past pk.T, if any, are padding)
    _lib.call("sk_pack_rows", _ptr(padded), _ptr(pk.offs), _ptr(pk.perm), pk.T, B, Cc, _ptr(out), out.stride(0), _stream())


def unpack_rows(packed, pk, out, fill=None):
    """The first R rows of packed (>= R, ld) -> out (T, B, C) in the caller's utterance order; padded positions get the row
    `fill` (C floats) or zeros."""
    _chk(packed)
    _chk(out)
    _chk(fill)
    T, B, Cc = out.shape
    _lib.call("sk_unpack_rows", _ptr(packed), packed.stride(0), _ptr(pk.offs), _ptr(pk.perm), T, B, Cc, _ptr(fill), _ptr(out),
              _stream())


def hprev_rows(y2d, h0, pk, H, out):
    """out[r] = [forward-direction output one frame earlier | reverse-direction output one frame later] of packed row r,
    h0 (2, B, H) where a row has no such neighbour (sk_hprev_rows); out fp32 or bfloat16, (>= R, ld >= 2H)."""
    _chk(y2d)
    _chk(h0)
    bf = out.dtype == torch.bfloat16
    _chk(out, torch.bfloat16 if bf else torch.float32)
    _lib.call("sk_hprev_rows", _ptr(y2d), y2d.stride(0), _ptr(h0), _ptr(pk.offs), pk.T, pk.B, H, _ptr(out), out.stride(0), int(bf),
              _stream())
    return out


def sigmoid_bwd(dmask, m, dz):
    _lib.call("sk_sigmoid_bwd", _ptr(dmask), _ptr(m), _ptr(dz), dmask.numel(), _stream())


# ----------------------------------------------------------------------------- LSTM recurrence
def lstm_ws(T, B, H):
    n = _lib.load().sk_lstm_workspace_bytes(T, B, H)
    if n == 0:
        raise _lib.SepkernError("unsupported LSTM shape B=%d H=%d" % (B, H))
    return workspace(n, "lstm")


def lstm_variant_bits(half=False, blockmap=0, poll1=False, repflags=False, spread=False, poll_delay=0, tagged=False, split3=False,
                      xl8=False):
    """Geometry / protocol variants of the persistent recurrence (speed only; include/sepkern.h, mode bits 17..29);
    poll_delay: the forward kernel's polling wave holds its first poll of a step back (units of 0.1 us, 0 = the library's
    choice, 31 = none); tagged (forward, fp32): the exchanged h carries the step's epoch in its two low mantissa bits and
    nothing else is signalled (mode bit 29); split3 (forward, fp32): the product h W_hh^T by the exact three-way bf16 split
    of both operands on the bf16 matrix pipe (mode bit 28; flags hand-off, the tagged one does not combine with it); xl8 (forward,
    bf16, 608 < H <= 896, B <= 32, persistent launches; other shapes run the ordinary form): XCD-local streams of 8 rows x 28
    workgroups of 32 units with a plain-store hand-off (mode bit 30) -- the same arithmetic bit for bit."""
    if half:      # (the first field of SEPKERN_LSTM_FWD / _BWD keeps its place so that recorded switch strings stay readable)
        raise _lib.SepkernError("the 8-unit / 256-thread forward recurrence (field `half`, mode bit 17) was retired in r05: "
                                "measured slower at every shape (DESIGN_HISTORY.md)")
    return (((int(blockmap) & 3) << 18) | (0x100000 if poll1 else 0) |
            (0x200000 if repflags else 0) | (0x400000 if spread else 0) | ((int(poll_delay) & 31) << 23) |
            (0x20000000 if tagged else 0) | (0x10000000 if split3 else 0) | (0x40000000 if xl8 else 0))


def lstm_fwd(gx, whh, h0, c0, lens, y, gates, cs, hn, cn, T, B, H, mode=0, bf16=False, blockmap=0, offs=None, rows=None):
    """bf16=True: W_hh and h_{t-1} enter the matrix cores rounded to bf16 (fp32 accumulate, fp32 state).
    blockmap 0..2: which workgroups share an XCD / a CU (mode bits 18..19; speed only).  offs (int32, T+1): the sequence tensors are PACKED
    rows (lens sorted descending; `rows` of them: the launch's algorithmic work for the profile); None: zero-padded (T, B, .)."""
    ws = lstm_ws(T, B, H)
    mode = int(mode) | (0x10000 if bf16 else 0) | ((int(blockmap) & 3) << 18)
    _chk(offs, torch.int32)
    with _timed("lstm_fwd_kernel", 2.0 * (T * B if rows is None else rows) * 2 * 4 * H * H):
        _lib.call("sk_lstm_fwd", _ptr(gx), _ptr(whh), _ptr(h0), _ptr(c0), _ptr(lens), _ptr(offs), _ptr(y), _ptr(gates), _ptr(cs),
                  _ptr(hn), _ptr(cn), _ptr(ws), T, B, H, mode, _stream())
    return ws


def lstm_bwd(dy, whh, gates, cs, c0, lens, dgx, dh0, dc0, T, B, H, mode=0, dhn=None, dcn=None, bf16=False,
             dbias=None, dgx_bf16=None, offs=None, rows=None):
    """dbias ((B+15)//16, 2, 4H): optional by-product (include/sepkern.h), its column sum is the bias gradient.
    dgx_bf16: a (rows, ld >= 8H) bfloat16 tensor that receives dgx as bf16 as well -- or (fp32 configuration) a Planes object:
    dgx's three exact bf16 pieces, the operand gemm_pl3_tn reads.  offs: as lstm_fwd."""
    plane = 0
    if isinstance(dgx_bf16, Planes):
        if bf16:
            raise _lib.SepkernError("lstm_bwd: planes of dgx belong to the fp32 configuration")
        plane, dgx_bf16 = dgx_bf16.plane, dgx_bf16.t[0]
    ws = lstm_ws(T, B, H)
    mode = int(mode) | (0x10000 if bf16 else 0)
    _chk(dbias)
    _chk(offs, torch.int32)
    _chk(dgx_bf16, torch.bfloat16)
    if dgx_bf16 is not None and (dgx_bf16.dim() != 2 or dgx_bf16.stride(1) != 1):
        raise _lib.SepkernError("lstm_bwd: the bf16 twin must be a row-major (rows, ld) matrix")
    with _timed("lstm_bwd_kernel", 2.0 * (T * B if rows is None else rows) * 2 * 4 * H * H):
        _lib.call("sk_lstm_bwd", _ptr(dy), _ptr(dhn), _ptr(dcn), _ptr(whh), _ptr(gates), _ptr(cs), _ptr(c0),
                  _ptr(lens), _ptr(offs), _ptr(dgx), _ptr(dh0), _ptr(dc0), _ptr(dbias), _ptr(dgx_bf16),
                  0 if dgx_bf16 is None else dgx_bf16.stride(0), int(plane), _ptr(ws), T, B, H, mode, _stream())
    return ws


def gate_rows(src, H, back=False, out=None, accumulate=False, cols=None):
    """Reorder rows of a (nblk * 4H, C) fp32 matrix (or (nblk * 4H,) vector) between torch's gate-major order and the
    gate-interleaved order of gx / gates / dgx (include/sepkern.h, sk_gate_rows).  src / out may have leading dimensions
    larger than the `cols` logical columns (default: the narrower of the two)."""
    _chk(src)
    s2 = src.reshape(-1, 1) if src.dim() == 1 else src.reshape(-1, src.shape[-1])
    if s2.stride(-1) != 1:
        raise _lib.SepkernError("gate_rows needs unit-stride rows")
    nblk = s2.shape[0] // (4 * H)
    if out is None:
        out = torch.empty_like(src)
    o2 = out.reshape(-1, 1) if out.dim() == 1 else out.reshape(-1, out.shape[-1])
    C_ = min(s2.shape[1], o2.shape[1]) if cols is None else cols
    _lib.call("sk_gate_rows", _ptr(s2), _ptr(o2), nblk, H, C_, s2.stride(0), o2.stride(0), int(back), int(accumulate), _stream())
    return out


def gates_interleaved(t, H, back=False):
    """(..., 4H) gate-major <-> gate-interleaved along the LAST axis, by torch view/permute (tests and tools: the
    product path gets the interleaved order for free from reordered weight rows)."""
    shape = t.shape
    if back:
        return t.reshape(shape[:-1] + (H, 4)).transpose(-1, -2).reshape(shape).contiguous()
    return t.reshape(shape[:-1] + (4, H)).transpose(-1, -2).reshape(shape).contiguous()


def lstm_status(ws):
    """Raises SepkernError (SK_ETIMEOUT) if a persistent launch on this workspace timed out since the last call."""
    _lib.call("sk_lstm_status", _ptr(ws), _stream())


def lstm_sticky(ws):
    """The workspace's sticky status word as a 1-element int32 view (device side checks, no sync)."""
    return ws[:4].view(torch.int32)


# ----------------------------------------------------------------------------- optimizer
def grad_norm(g, max_norm, scal, guard=None):
    """scal (4,) <- [norm, clip coefficient, skip this step, skipped so far]; guard: optional 1-element float tensor,
    non-zero = do not apply this step (include/sepkern.h)."""
    ws = workspace(_lib.load().sk_optim_workspace_bytes(g.numel()), "optim")
    _lib.call("sk_grad_norm", _ptr(g), g.numel(), float(max_norm), _ptr(guard), _ptr(scal), _ptr(ws), _stream())


def clip_adam(p, g, m, v, scal, lr, beta1, beta2, eps, step):
    _lib.call("sk_clip_adam", _ptr(p), _ptr(g), _ptr(m), _ptr(v), p.numel(), _ptr(scal), float(lr), float(beta1),
              float(beta2), float(eps), int(step), _stream())
