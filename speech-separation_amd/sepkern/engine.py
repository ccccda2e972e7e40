"""The uPIT network (BLSTM -> BatchNorm1d -> Linear -> sigmoid) as a sequence of libsepkern calls.

Mirrors SepDNN.forward of the reference (archs/uPIT.py:129-147) and the backward torch's autograd
would run for it, on padded time-major (T, B, C) tensors with per-row lengths instead of
PackedSequences.  All parameters live in ONE flat fp32 buffer (and all gradients in another):
  * the LSTM kernels want (2, 4H, *) blocks (both directions of a layer) contiguous,
  * data-parallel training all-reduces one buffer (RCCL, one collective per step),
  * clip_grad_norm_ + Adam run as one fused pass over it (sepkern.optim).
nn.Parameters with the reference's names are views into it, so state_dict() is unchanged.
"""
import os

import torch

from . import dist as skdist
from . import ops
from ._lib import SepkernError

_PER_DIR = ("weight_ih", "weight_hh", "bias_ih", "bias_hh")


def _align(n, a=4):
    return (n + a - 1) // a * a


class ParamLayout:
    """Offsets (in floats) of every parameter block inside the flat buffer.

    Per layer: weight_ih (2,4H,I) | weight_hh (2,4H,H) | bias_ih (2,4H) | bias_hh (2,4H); then
    lin.weight (out_dim, 2H), lin.bias, bn.weight, bn.bias.  Every block starts 16-byte aligned.
    uPIT: in_dim = F, out_dim = S*F; RSH: in_dim = 2F (mixture | attention), out_dim = F.
    """

    def __init__(self, in_dim, out_dim, hidden, layers):
        self.I, self.O, self.H, self.L = in_dim, out_dim, hidden, layers
        self.blocks = {}
        off = 0
        H = hidden
        for l in range(layers):
            I = in_dim if l == 0 else 2 * H
            for name, shape in (("weight_ih", (2, 4 * H, I)), ("weight_hh", (2, 4 * H, H)),
                                ("bias_ih", (2, 4 * H)), ("bias_hh", (2, 4 * H))):
                n = 1
                for s in shape:
                    n *= s
                self.blocks["%s_l%d" % (name, l)] = (off, shape)
                off = _align(off + n)
        for name, shape in (("lin.weight", (out_dim, 2 * H)), ("lin.bias", (out_dim,)),
                            ("bn.weight", (2 * H,)), ("bn.bias", (2 * H,))):
            n = 1
            for s in shape:
                n *= s
            self.blocks[name] = (off, shape)
            off = _align(off + n)
        self.total = off

    GUARD = 4      # floats in front of the gradients in Engine.grad_full (word 0: the recurrence's status, see Engine)

    def grad_chunks(self):
        """[(name, lo, hi)] over Engine.grad_full = [guard words | gradients], in the order in which the backward pass
        completes them: Linear + BatchNorm, then the LSTM layers from the top down; the bottom layer's chunk comes last
        and carries the guard words in front of it (they are written at the very end of the pass).  Contiguous, disjoint,
        covering the buffer: the data-parallel exchange may go chunk by chunk (sepkern.dist.GradReducer)."""
        G = self.GUARD
        start = [self.blocks["weight_ih_l%d" % l][0] for l in range(self.L)] + [self.blocks["lin.weight"][0]]
        out = [("lin+bn", G + start[self.L], G + self.total)]
        for l in range(self.L - 1, 0, -1):
            out.append(("layer%d" % l, G + start[l], G + start[l + 1]))
        out.append(("guard+layer0", 0, G + start[1]))
        return out

    def view(self, flat, name):
        off, shape = self.blocks[name]
        n = 1
        for s in shape:
            n *= s
        return flat[off:off + n].view(shape)


class Engine:
    """Forward / backward of the network on one device."""

    def __init__(self, in_dim, out_dim, hidden, layers, device, precision="fp32", sync_bn=False):
        if precision not in ("fp32", "bf16"):
            raise SepkernError("precision must be 'fp32' or 'bf16' (got %r)" % (precision,))
        # bf16: EVERY matrix product (input projections, Linear, the recurrence h W_hh^T, their data and weight
        # gradients) rounds both operands to bf16 on the way into the matrix cores and accumulates in fp32;
        # parameters, activations, cell state, gradients, BatchNorm, the loss and Adam stay fp32.
        self.precision, self.bf16 = precision, precision == "bf16"
        # bf16 products read bf16 COPIES of their operands (sk_cast_bf16 / sk_cast_bf16_t, transposed where the product
        # needs it so that every product is the one NT form of sk_gemm_bf16_nt); SEPKERN_BF16_NT=0 keeps the r01 kernel
        # that reads fp32 operands and rounds them on the way into LDS (same arithmetic, half the speed)
        self.nt = self.bf16 and os.environ.get("SEPKERN_BF16_NT", "1") == "1"
        # ... and (r03) only ROW-MAJOR copies: a factor whose rows are the contraction index (both factors of a weight
        # gradient, the weight matrix of a data gradient) enters the product K-major (sk_gemm_bf16_mm: the tile is DMA'd as
        # it lies and transposed by ds_read_b64_tr_b16 on the way into the matrix cores), so the 12 transposed copies per
        # step (sk_cast_bf16_t, 0.94 ms) are never made.  SEPKERN_BF16_KMAJOR=0 keeps the r02 path with transposed copies.
        self.kmajor = self.nt and hidden % 8 == 0 and os.environ.get("SEPKERN_BF16_KMAJOR", "1") == "1"
        # ... and the backward recurrence writes the bf16 copy of dgx itself (sk_lstm_bwd_twin); SEPKERN_BF16_TWIN=0: cast pass
        self.bf16_twin = os.environ.get("SEPKERN_BF16_TWIN", "1") == "1"
        # data-parallel runs only: BatchNorm over the GLOBAL batch instead of per rank (sepkern/dist.py)
        self.sync_bn = bool(sync_bn) or os.environ.get("SEPKERN_SYNC_BN", "0") == "1"
        if hidden % 4 != 0 or hidden > 1024:
            raise SepkernError("hidden_dim must be a multiple of 4 and <= 1024 (got %d)" % hidden)
        self.I, self.O, self.H, self.L = in_dim, out_dim, hidden, layers
        self.layout = ParamLayout(in_dim, out_dim, hidden, layers)
        self.device = device
        self.flat = torch.zeros(self.layout.total, device=device)
        # [guard words | gradients]: the recurrence's sticky status (sepkern/ops.lstm_sticky) is copied into word 0 at the
        # end of every backward, so the data-parallel all-reduce of the buffer tells EVERY rank when any rank's
        # persistent launch timed out, and the fused clip+Adam skips that step on the device (no host sync per step).
        # In FRONT of the gradients: the bottom layer's gradients are the last to be complete, so in the chunked exchange
        # (ParamLayout.grad_chunks) the guard travels with the last chunk.
        G = ParamLayout.GUARD
        self.grad_full = torch.zeros(G + self.layout.total, device=device)
        self.grad = self.grad_full[G:]
        self.guard = self.grad_full[0:1]
        self.running_mean = torch.zeros(2 * hidden, device=device)
        self.running_var = torch.ones(2 * hidden, device=device)
        self.eps, self.momentum = 1e-5, 0.1
        self.lstm_mode = int(os.environ.get("SEPKERN_LSTM_MODE", "0"))
        # forward recurrence geometry (speed only): "half,map" -- half=1: 8-unit / 256-thread workgroups, two per CU;
        # map 0..2: block id -> stream assignment (csrc/lstm.hip::decode_block)
        def variant(env, default):
            v = [int(x) for x in os.environ.get(env, default).split(",")]
            v += [0] * (8 - len(v))
            return ops.lstm_variant_bits(bool(v[0]), v[1], bool(v[2]), bool(v[3]), bool(v[4]), v[5], dual=bool(v[6]), tagged=bool(v[7]))
        # "half,map,poll1,repflags,spread,delay[,dual[,tagged]]" (dual: the two-stream forward kernel, tagged:
        # the data-is-the-flag hand-off, both 5c).  Forward: streams dealt to XCD groups, one polling wave,
        # first poll held back (delay 0 = the library's choice); r02: 7.33 -> 6.0 us/step in fp32, 4.0 -> 3.0 in bf16, where
        # one flag per 128-byte line is worth another 5 %.  Backward: the XCD map only (8.00 -> 7.25; it polls later anyway).
        self.fwd_bits = variant("SEPKERN_LSTM_FWD", "0,1,1,0,1,0" if self.bf16 else "0,1,1,0,0,8,0,1")
        # fp32 forward (r03): the data is the flag (tagged = 1, mode bit 29; hold-back 0.8 us): the exchanged h carries the
        # step's epoch in its two low mantissa bits, no drain / barrier / flag / poll: 7.44 -> 7.02 ms of forward recurrences
        # per training step (37.1 -> 36.6 ms), h entering the next step's product perturbed by at most 3 ulp
        self.bwd_bits = variant("SEPKERN_LSTM_BWD", "0,1,0,0,0,31")
        # Weight-gradient GEMMs of layer l run on a side stream while layer l-1's recurrence runs on the main one.
        # SEPKERN_OVERLAP=2 (default): the recurrence keeps its one-workgroup-per-CU grid and the GEMM blocks
        # become CO-RESIDENT on its CUs -- a persistent workgroup leaves >=124 VGPRs per SIMD lane and >=69 KB of
        # LDS free, and the matrix pipe idle during its hand-offs -- : 42.9 -> 40.5 ms/step at 3x896, 32x400 (the
        # co-scheduled GEMMs run at ~1/3 of their stand-alone rate, the recurrence 25 % slower, net +6 %).
        # =1: recurrence on half the CUs (2 batch groups per workgroup), GEMMs on the rest: +1.3 % only, the
        # hand-off chain slows from 9.5 to 15.6 ms/step under the GEMMs' L2/fabric load.  =0: no overlap.
        self.overlap_mode = int(os.environ.get("SEPKERN_OVERLAP", "2"))
        self.overlap = self.overlap_mode in (1, 2)
        # forward: recurrences in two launches with half of the next input projection beside the second (see forward())
        self.fwd_split = os.environ.get("SEPKERN_FWD_SPLIT", "0") == "1"
        # fp32 GEMM kernel per stream (sk_gemm_f32_splitk's variant): "main,side".  Default: choose (LDS-DMA where it
        # applies) on the main stream, the register-staged kernel for products that run beside a recurrence (_wgrad).
        # "2,2": every product by the exact three-way bf16 split on the bf16 matrix pipe (opt-in).
        self.dgrad_unsplit = os.environ.get("SEPKERN_DGRAD_UNSPLIT", "1") == "1"
        # backward recurrences in two launches, half of a layer's own weight-gradient products beside the second
        # (backward(); built, parity-tested, measured 38.7 vs 37.7 ms per step: off)
        self.bwd_split = os.environ.get("SEPKERN_BWD_SPLIT", "0") == "1"
        # BatchNorm folded into the Linear layer (fp32 path; the bf16 arithmetic is DEFINED with bn(y) and W rounded
        # separately, oracle/upit_bf16.py, so that path keeps the explicit normalisation)
        self.bn_fold = os.environ.get("SEPKERN_BN_FOLD", "1") == "1" and not self.bf16
        self.var_main, self.var_side = (int(v) for v in os.environ.get("SEPKERN_GEMM_VARIANTS", "0,1").split(","))
        self.side = None
        self.grads_fresh = True        # True: next backward may overwrite instead of accumulate

    def p(self, name):
        return self.layout.view(self.flat, name)

    def g(self, name):
        return self.layout.view(self.grad, name)

    def zero_grad(self):
        """model.zero_grad(): the next backward overwrites every gradient element, so nothing is memset."""
        self.grads_fresh = True

    def check_status(self):
        """Host-side check (synchronises): raises SepkernError if a persistent recurrence launch timed out since the
        last check.  Training does not need it per step -- see `guard` -- drivers call it at epoch / checkpoint time."""
        ops.lstm_status(ops.workspace(0, "lstm"))

    def sticky(self):
        """The workspace's sticky status word as a device tensor (1 element, int32): non-zero after a timed-out launch."""
        return ops.lstm_sticky(ops.workspace(0, "lstm"))

    # ------------------------------------------------------------------ the three kinds of product
    # fp32: the fp32 MFMA kernel on the fp32 tensors.  bf16: sk_gemm_bf16_nt on bf16 copies; `cache` (one dict per
    # pass) holds the copies already made, keyed by (kind, id of the fp32 tensor), and keeps them alive.
    @staticmethod
    def _copy(cache, kind, t2d):
        """bf16 operand copy of t2d ("row": as stored, "t": transposed), made once per pass.  A copy made on one stream and
        used on the other is ordered by ITS OWN event: the user waits for that cast, not for everything the maker's
        stream has queued behind it."""
        key = (kind, t2d.data_ptr(), tuple(t2d.shape))
        cur = torch.cuda.current_stream()
        ent = cache.get(key)
        if ent is None:
            if kind == "rowk":     # row-major copy that also serves as a K-major factor: whole K steps of zero rows behind it
                c = ops.cast_bf16(t2d, rows=ops.pad_to(t2d.shape[0], 64) + 64)
            else:
                c = ops.cast_bf16(t2d) if kind == "row" else ops.cast_bf16_t(t2d)
            ev = torch.cuda.Event()
            ev.record(cur)
            # the entry holds the fp32 SOURCE too: its address cannot be recycled for another tensor of the same shape
            # while the copy is cached (sources are not written between their uses within a pass)
            cache[key] = ent = (c, ev, cur, t2d)
        elif ent[2] != cur:
            cur.wait_event(ent[1])
            ent[0].record_stream(cur)
        return ent[0]

    def _proj(self, cache, inp2d, w, out2d, bias, act=0):
        """out (R, N) = act(inp (R, K) w (N, K)^T + bias)."""
        R, K = inp2d.shape
        N = w.shape[0]
        if not self.nt:
            ops.gemm(inp2d, w, out2d, R, N, K, inp2d.stride(0), K, N, transB=True, bias=bias, act=act, bf16=self.bf16,
                     variant=self.var_main)
            return
        kind = "rowk" if self.kmajor else "row"            # the same copies serve the backward products K-major
        a, b = self._copy(cache, kind, inp2d), self._copy(cache, kind, w)
        # (not the stream-K kernel: with 1400 tiles of 28 K steps its fix-up costs more than the sixth partial round it saves --
        # main-stream products 3.03 vs 3.11 ms per step; the data gradients' 350 tiles of 112 steps are where it pays)
        ops.gemm_bf16_nt(a, b, out2d, R, N, a.shape[1], a.shape[1], b.shape[1], N, bias=bias, act=act,
                         streamk=os.environ.get("SEPKERN_BF16_PROJ_SK", "0") == "1")

    def _dgrad(self, cache, dout2d, w, out2d, ws_tag):
        """out (R, K) = dout (R, N) w (N, K)."""
        R, N = dout2d.shape
        K = w.shape[1]
        if not self.nt:
            # large data gradients unsplit: the 256 x 128-tile kernel (sk_gemm_f32_splitk picks it for unsplit N/N products)
            # measured 125.5 TFLOP/s against 119-120 for two K slices of 128 x 128 tiles
            sk = 1 if (self.dgrad_unsplit and not self.bf16 and R >= 4096 and K >= 1024 and N % 16 == 0) else 0
            ops.gemm(dout2d, w, out2d, R, K, N, N, K, K, splitk=sk, ws_tag=ws_tag, bf16=self.bf16, variant=self.var_main)
            return
        if self.kmajor:
            # out = dout w: w (N, K) is the K-major B of the product as it lies (contraction over its rows, padded with
            # zero rows up to dout's zero-padded width)
            a, b = self._copy(cache, "rowk", dout2d), self._copy(cache, "rowk", w)
            ops.gemm_bf16_mm(a, b, out2d, R, K, a.shape[1], a.shape[1], b.shape[1], K, b_kmajor=True, splitk=0, ws_tag=ws_tag,
                             streamk=True)
            return
        a, bt = self._copy(cache, "row", dout2d), self._copy(cache, "t", w)      # w^T: (K, N padded)
        ops.gemm_bf16_nt(a, bt, out2d, R, K, a.shape[1], a.shape[1], bt.shape[1], K, splitk=0, ws_tag=ws_tag)

    def _wgrad(self, cache, dout2d, inp2d, gw, acc, ws_tag, beside=False):
        """gw (N, K) [+]= dout (R, N)^T inp (R, K).  beside=True: the product runs co-resident with a recurrence (side
        stream): the register-staged GEMM kernel, which leaves the recurrence more of the matrix pipe than the LDS-DMA
        one does (measured: same step time with either, 2 ms longer recurrences with the latter)."""
        R, N = dout2d.shape
        K = inp2d.shape[1]
        if not self.nt:
            ops.gemm(dout2d, inp2d, gw, N, K, R, N, inp2d.stride(0), K, transA=True, accumulate=acc, splitk=0,
                     ws_tag=ws_tag, bf16=self.bf16, variant=self.var_side if beside else self.var_main)
            return
        if self.kmajor:
            # gw = dout^T inp: both factors K-major as they lie (their rows are the contraction index)
            a, b = self._copy(cache, "rowk", dout2d), self._copy(cache, "rowk", inp2d)
            ops.gemm_bf16_mm(a, b, gw, N, K, ops.pad_to(R, 64), a.shape[1], b.shape[1], K, a_kmajor=True, b_kmajor=True,
                             accumulate=acc, splitk=0, ws_tag=ws_tag, streamk=not beside)
            return
        at, bt = self._copy(cache, "t", dout2d), self._copy(cache, "t", inp2d)    # (N, R padded), (K, R padded)
        ops.gemm_bf16_nt(at, bt, gw, N, K, ops.pad_to(R, 64), at.shape[1], bt.shape[1], K, accumulate=acc, splitk=0,
                         ws_tag=ws_tag)

    def _whh_grad(self, cache, dgx2d, y2d, h0, dg_first, gw, T, B, acc, ws_tag, beside=False):
        """dW_hh (2,4H,H) [+]= sum_t dG_t^T h_prev(t) (ops.lstm_whh_grad); in bf16 from the transposed copies: the time
        shift is an offset of B columns into one of them, and the copies end in >= 64 zero columns."""
        H = self.H
        if not (self.nt and T > 1 and B % 8 == 0):
            ops.lstm_whh_grad(dgx2d, y2d, h0, dg_first, gw, T, B, H, accumulate=acc, bf16=self.bf16, ws_tag=ws_tag,
                              variant=self.var_side if beside else self.var_main)
            return
        if self.kmajor:
            # the time shift is an offset of B ROWS into one of the K-major factors: direction 0 pairs dG rows from t = 1
            # with y rows from t = 0, direction 1 dG rows from t = 0 with y rows from t = 1; the copies end in >= 64 zero
            # rows, which absorb the rounding of K = (T-1) B up to 64
            a, b = self._copy(cache, "rowk", dgx2d), self._copy(cache, "rowk", y2d)   # (R+, 8H), (R+, 2H)
            lda, ldb = a.shape[1], b.shape[1]
            ops.gemm_bf16_mm(a.view(-1)[B * lda:], b, gw, 4 * H, H, ops.pad_to((T - 1) * B, 64), lda, ldb, H, a_kmajor=True,
                             b_kmajor=True, accumulate=acc, batch=2, sA=4 * H - B * lda, sB=B * ldb + H, sC=4 * H * H, splitk=0,
                             ws_tag=ws_tag)
            ops.gemm(dg_first, h0, gw, 4 * H, H, B, 4 * H, H, H, transA=True, accumulate=True, batch=2, sA=B * 4 * H, sB=B * H,
                     sC=4 * H * H, ws_tag=ws_tag, bf16=True)
            return
        at, bt = self._copy(cache, "t", dgx2d), self._copy(cache, "t", y2d)       # (8H, ld), (2H, ld)
        ld = at.shape[1]
        ops.gemm_bf16_nt(at.view(-1)[B:], bt, gw, 4 * H, H, ops.pad_to((T - 1) * B, 64), ld, ld, H, accumulate=acc, batch=2,
                         sA=4 * H * ld - B, sB=H * ld + B, sC=4 * H * H, splitk=0, ws_tag=ws_tag)
        ops.gemm(dg_first, h0, gw, 4 * H, H, B, 4 * H, H, H, transA=True, accumulate=True, batch=2, sA=B * 4 * H, sB=B * H,
                 sC=4 * H * H, ws_tag=ws_tag, bf16=True)

    def _wgrad_half(self, c, dgx, y, inp, gw_hh, gw_ih, T, B, Ip, ws_tag, beside):
        """Half c (1 or 2) of a layer's weight-gradient products in the split backward schedule (fp32).  With S = T/2:
        half 1 = the rows that are final after S steps of the backward recurrence (forward direction t >= S, reverse
        direction t < S), half 2 = the rest, accumulated onto half 1.  Per direction, as one batched launch each:
          dW_hh[d] (+)= sum_t dG_t[d]^T h_prev(t)[d]   (ops.lstm_whh_grad's time-shifted pairs, cut at S)
          dW_ih[d] (+)= sum_t dG_t[d]^T x_t            (rows of gw_ih: direction 0's 4H, then direction 1's)"""
        H, S = self.H, T // 2
        dg, yv, xv = dgx.view(-1), y.view(-1), inp.reshape(-1)
        var = self.var_side if beside else self.var_main
        kw = dict(transA=True, accumulate=(c == 2), batch=2, splitk=0, ws_tag=ws_tag, variant=var)
        if c == 1:     # forward direction: pairs (dG_t, y_{t-1}), t = S..T-1; reverse: (dG_t, y_{t+1}), t = 0..S-1
            ops.gemm(dg[S * B * 8 * H:], yv[(S - 1) * B * 2 * H:], gw_hh, 4 * H, H, S * B, 8 * H, 2 * H, H,
                     sA=4 * H - S * B * 8 * H, sB=H - (S - 2) * B * 2 * H, sC=4 * H * H, **kw)
            ops.gemm(dg[S * B * 8 * H:], xv[S * B * Ip:], gw_ih, 4 * H, Ip, S * B, 8 * H, Ip, Ip,
                     sA=4 * H - S * B * 8 * H, sB=-S * B * Ip, sC=4 * H * Ip, **kw)
        else:          # forward direction: t = 1..S-1 (t = 0 pairs with h0: the caller's rank-B term); reverse: t = S..T-2
            ops.gemm(dg[B * 8 * H:], yv, gw_hh, 4 * H, H, (S - 1) * B, 8 * H, 2 * H, H,
                     sA=S * B * 8 * H + 4 * H - B * 8 * H, sB=(S + 1) * B * 2 * H + H, sC=4 * H * H, **kw)
            ops.gemm(dg, xv, gw_ih, 4 * H, Ip, S * B, 8 * H, Ip, Ip, sA=S * B * 8 * H + 4 * H, sB=S * B * Ip, sC=4 * H * Ip, **kw)

    # ------------------------------------------------------------------ forward
    def forward(self, x, lens, h0, c0, training, save, want_state=False):
        """x (T,B,in_dim) fp32, lens int32 (B) on device, h0/c0 (2L,B,H) ->
        (mask (T,B,out_dim), hn, cn (2L,B,H) or None, ctx or None).  ctx feeds backward(); several may be alive
        (the RSH arch runs the network num_spk times per batch)."""
        T, B, I0 = x.shape
        if I0 != self.I:
            raise SepkernError("input feature dim %d != model input dim %d" % (I0, self.I))
        H, L, O = self.H, self.L, self.O
        R = T * B
        x = x.contiguous()
        dev = x.device
        saved = []
        cache = {}
        inp, I = x, I0
        ws = None
        hn = torch.empty(2 * L, B, H, device=dev) if want_state else None
        cn = torch.empty(2 * L, B, H, device=dev) if want_state else None
        def weights_of(l, I, ws_tag="bn"):
            """(W_ih rows gate-interleaved and padded to Ip columns, summed bias gate-interleaved, Ip) of layer l."""
            wih = self.p("weight_ih_l%d" % l)
            # b_ih + b_hh for both directions: the two bias blocks are adjacent rows of a (2, 8H) matrix
            off_ih, _ = self.layout.blocks["bias_ih_l%d" % l]
            bsum = torch.empty(8 * H, device=dev)
            ops.colsum(self.flat[off_ih:], 2, 8 * H, 8 * H, bsum, ws_tag=ws_tag)
            # the recurrence keeps i,f,g,o of a cell adjacent (one 16-byte access per cell and step instead of four
            # H-strided ones): reorder the rows of W_ih and of the bias once, the GEMM then writes gx in that order
            # ... and in the same pass pads an input width that is no multiple of 4 (F = 257 -> 260) with zero columns,
            # so that the rows of both operands of the layer-0 products are 16-byte aligned (float4 fetches)
            Ip = ops.pad_to(I, 4)
            wih_gi = ops.gate_rows(wih.view(8 * H, I), H, out=torch.empty(8 * H, Ip, device=dev), cols=I)
            return wih_gi, ops.gate_rows(bsum, H), Ip

        # Split recurrences (fp32): a layer that feeds another one runs as TWO launches of T/2 steps.  After the first,
        # the forward half of y is final for t < T/2 and the reverse half for t >= T/2, so half of the next layer's input
        # projection -- those rows times the matching half of W_ih -- is issued on the side stream and runs CO-RESIDENT
        # with the second launch (a recurrence leaves its CU's matrix pipe idle half of the time); the other
        # half of the product follows on the main stream.  Same sums in a different order: fp32 rounding-level changes.
        split = (self.fwd_split and self.overlap and not self.bf16 and self.lstm_mode == 0 and T % 2 == 0 and T >= 16 and L > 1)
        main = torch.cuda.current_stream(dev)
        if split and self.side is None:
            self.side = torch.cuda.Stream(device=dev)
        keep = []
        # The gate-interleaved copies of W_ih and the summed biases of the layers above the first depend on the weights only:
        # they are made on the side stream (when the backward pass has created one) beside layer 0's projection and
        # recurrence instead of in front of each layer's projection on the main stream (r03: 0.13 ms of small kernels per step)
        ahead, ahead_ev = {}, None
        if self.side is not None and self.overlap and L > 1 and not split and os.environ.get("SEPKERN_PREP_AHEAD", "1") == "1":
            self.side.wait_stream(main)
            with torch.cuda.stream(self.side):
                for l in range(1, L):
                    ahead[l] = weights_of(l, 2 * H, ws_tag="bn_side")
                    for t_ in ahead[l][:2]:
                        t_.record_stream(main)
                ahead_ev = torch.cuda.Event()
                ahead_ev.record(self.side)
        gx_ready = None                                  # (gx, wih_gi) of the NEXT layer when its projection was split in
        for l in range(L):
            whh = self.p("weight_hh_l%d" % l)
            if gx_ready is None:
                if l in ahead:
                    if ahead_ev is not None:
                        main.wait_event(ahead_ev)
                        ahead_ev = None
                    wih_gi, bsum, Ip = ahead[l]
                else:
                    wih_gi, bsum, Ip = weights_of(l, I)
                inp2d = inp.view(R, I)
                if Ip != I:
                    inp2d = ops.pad_rows(inp2d, Ip)          # one pass, no memset (F = 257 -> 260)
                gx = torch.empty(T, B, 2, 4 * H, device=dev)
                self._proj(cache, inp2d, wih_gi, gx.view(R, 8 * H), bsum)
            else:
                gx, wih_gi = gx_ready
                inp2d = inp.view(R, I)
                gx_ready = None
            y = torch.empty(T, B, 2 * H, device=dev)
            cs = torch.empty(T, B, 2, H, device=dev) if save else None
            args = (gx, whh, h0[2 * l:2 * l + 2], c0[2 * l:2 * l + 2], lens, y, gx if save else None, cs,
                    hn[2 * l:2 * l + 2] if want_state else None, cn[2 * l:2 * l + 2] if want_state else None,
                    T, B, H, self.lstm_mode | self.fwd_bits)
            if split and l + 1 < L:
                S1, half = T // 2, (T // 2) * B              # steps per launch, rows of y per half
                nw, nb, _ = weights_of(l + 1, 2 * H)         # (8H, 2H) interleaved rows, bias
                ngx = torch.empty(T, B, 2, 4 * H, device=dev)
                ws = ops.lstm_fwd(*args, bf16=False, steps=(0, S1))
                self.side.wait_stream(main)
                with torch.cuda.stream(self.side):
                    # rows t < T/2: forward half of y (columns :H) x W[:, :H]^T;  rows t >= T/2: reverse half x W[:, H:]^T
                    ops.gemm(y.view(R, 2 * H), nw, ngx.view(R, 8 * H), half, 8 * H, H, 2 * H, 2 * H, 8 * H, transB=True, bias=nb,
                             batch=2, sA=half * 2 * H + H, sB=H, sC=half * 8 * H, sbias=0)
                ws = ops.lstm_fwd(*args, bf16=False, steps=(S1, T))
                main.wait_stream(self.side)
                # the other halves, accumulated: rows t < T/2 x W[:, H:]^T of the reverse half, rows t >= T/2 x W[:, :H]^T
                ops.gemm(y.view(-1)[H:], nw.view(-1)[H:], ngx.view(R, 8 * H), half, 8 * H, H, 2 * H, 2 * H, 8 * H, transB=True,
                         accumulate=True, batch=2, sA=half * 2 * H - H, sB=-H, sC=half * 8 * H)
                gx_ready = (ngx, nw)
                keep += [nw, nb, ngx, y]
            else:
                ws = ops.lstm_fwd(*args, bf16=self.bf16)
            saved.append((inp2d, gx, cs, y, wih_gi))
            inp, I = y, 2 * H
        del keep
        if not save and not skdist.is_parallel():
            # inference: the caller copies the masks to the host next, a sync costs nothing.  Under data parallelism a
            # raise on ONE rank would leave the others waiting in their next collective: there the sticky word stays set
            # and the driver reports it on every rank together (steps/train_qsub.py::validation_pass)
            ops.lstm_status(ws)
        y2d = inp.view(R, 2 * H)
        if training:
            mean = torch.empty(2 * H, device=dev)
            var = torch.empty(2 * H, device=dev)
            ops.bn_stats(y2d, mean, var)
            bn_count = float(R)
            if self.sync_bn:                             # statistics of the global batch (one all-gather)
                mean, var, bn_count = skdist.combine_bn_stats(mean, var, R)
            ops.bn_update_running(mean, var, self.running_mean, self.running_var, int(bn_count), self.momentum)
        else:
            mean, var, bn_count = self.running_mean, self.running_var, float(R)
        mask = torch.empty(T, B, O, device=dev)
        xbn = fold = None
        if self.bn_fold:
            # BatchNorm folded into the Linear weights (sk_bn_fold): mask = sigmoid(y Wf^T + bf), the
            # normalised activations are never written (one 92 MB pass less, forward and backward)
            Wf, bf_, s_, t_ = ops.bn_fold(self.p("lin.weight"), self.p("lin.bias"), mean, var, self.p("bn.weight"),
                                          self.p("bn.bias"), self.eps)
            self._proj(cache, y2d, Wf, mask.view(R, O), bf_, act=1)
            fold = (s_, t_)
        else:
            xbn = torch.empty(R, 2 * H, device=dev)
            ops.bn_apply(y2d, mean, var, self.p("bn.weight"), self.p("bn.bias"), xbn, self.eps)
            self._proj(cache, xbn, self.p("lin.weight"), mask.view(R, O), self.p("lin.bias"), act=1)
        ctx = None
        if save:
            ctx = dict(saved=saved, mean=mean, var=var, bn_count=bn_count, xbn=xbn, fold=fold, mask=mask, lens=lens, h0=h0,
                       c0=c0, T=T, B=B, training=training,
                       # bf16, K-major products: the backward pass multiplies the SAME row-major copies of the weights and
                       # of every layer input (none of them is written in between), so they are made once per step
                       cache=cache if self.kmajor else None)
        return mask, hn, cn, ctx

    # ------------------------------------------------------------------ backward
    def backward(self, ctx, dmask, dhn=None, dcn=None, want_dx=False, want_dstate=False, reducer=None):
        """Parameter gradients (into the flat gradient buffer) from dmask (T,B,out_dim) and, optionally, the
        gradient wrt the final state (dhn, dcn (2L,B,H)).  Returns (dx (T,B,in_dim) or None, dh0, dc0 or None).
        reducer (sepkern.dist.GradReducer, data-parallel runs with SEPKERN_DP_OVERLAP=1, last backward of a step only):
        every chunk of ParamLayout.grad_chunks() is handed over as soon as the kernels that complete it are enqueued."""
        if ctx is None:
            raise SepkernError("backward called without a saved forward")
        if not ctx["training"]:
            raise SepkernError("backward through eval-mode BatchNorm is not built")
        T, B, H, L, O, I0 = ctx["T"], ctx["B"], self.H, self.L, self.O, self.I
        R = T * B
        dev = dmask.device
        acc = not self.grads_fresh
        lens, h0, c0 = ctx["lens"], ctx["h0"], ctx["c0"]
        dmask = dmask.contiguous()

        def put(name, val):           # small vectors produced by non-accumulating kernels
            if acc:
                self.g(name).add_(val)
            else:
                self.g(name).copy_(val)

        overlap = self.overlap and L > 1 and self.lstm_mode == 0
        if overlap and self.side is None:
            self.side = torch.cuda.Stream(device=dev)
        main = torch.cuda.current_stream(dev)
        keep = []                                        # tensors used on the side stream stay alive until the join
        dz = torch.empty_like(dmask)
        ops.sigmoid_bwd(dmask, ctx["mask"], dz)
        cache = ctx.get("cache") or {}                   # bf16 operand copies (see _copy); with K-major products the forward's
        dz2d = dz.view(R, O)
        dxbn = torch.empty(R, 2 * H, device=dev)
        self._dgrad(cache, dz2d, self.p("lin.weight"), dxbn, "gemm")
        y_top = ctx["saved"][-1][3].view(R, 2 * H)
        dy = torch.empty(R, 2 * H, device=dev)
        dgamma = torch.empty(2 * H, device=dev)
        dbeta = torch.empty(2 * H, device=dev)
        ops.bn_bwd_sums(dxbn, y_top, ctx["mean"], ctx["var"], dgamma, dbeta, self.eps)
        put("bn.weight", dgamma)                         # local sums: the flat all-reduce adds the ranks up later
        put("bn.bias", dbeta)
        if self.sync_bn:                                 # dx needs the sums over the global batch (one all-reduce)
            dgamma, dbeta = skdist.allreduce_bn_sums(dgamma, dbeta)
        ops.bn_bwd_apply(dxbn, y_top, ctx["mean"], ctx["var"], self.p("bn.weight"), dgamma, dbeta, dy, ctx["bn_count"],
                         self.eps)
        del dxbn
        # the Linear layer's own gradients are needed by nobody before clip+Adam: side stream, next to the top
        # layer's recurrence (enqueued HERE, after the BatchNorm backward on the main stream: issued earlier they ran
        # beside those short critical-path kernels and slowed them by 0.1 ms)
        stream = self.side if overlap else main
        if stream is not main:
            stream.wait_stream(main)
        with torch.cuda.stream(stream):
            tag = "_side" if overlap else ""
            if ctx["fold"] is not None:
                # dW = dz^T bn(y) = (dz^T y) diag(s) + colsum(dz) t^T  (sk_bn_unfold_grad): the product runs against y itself
                G = torch.empty(O, 2 * H, device=dev)
                dzsum = torch.empty(O, device=dev)
                self._wgrad(cache, dz2d, y_top, G, False, "gemm" + tag, beside=overlap)
                ops.colsum(dz, R, O, O, dzsum, ws_tag="bn" + tag)
                ops.bn_unfold_grad(G, dzsum, ctx["fold"][0], ctx["fold"][1], self.g("lin.weight"), accumulate=acc)
                if acc:
                    self.g("lin.bias").add_(dzsum)
                else:
                    self.g("lin.bias").copy_(dzsum)
                keep += [G, dzsum]
            else:
                self._wgrad(cache, dz2d, ctx["xbn"], self.g("lin.weight"), acc, "gemm" + tag, beside=overlap)
                ops.colsum(dz, R, O, O, self.g("lin.bias"), accumulate=acc, ws_tag="bn" + tag)
            keep.append(dz)
        del dz
        chunks = {name: (lo, hi) for name, lo, hi in self.layout.grad_chunks()} if reducer is not None else None
        if reducer is not None:
            reducer.chunk(self.grad_full, *chunks["lin+bn"], stream)       # (bn.weight / bn.bias were put before the side stream forked)
        ws = None
        dh0 = torch.empty(2 * L, B, H, device=dev) if want_dstate else None
        dc0 = torch.empty(2 * L, B, H, device=dev) if want_dstate else None
        dx = None
        for l in range(L - 1, -1, -1):
            inp, gates, cs, y, wih_gi = ctx["saved"][l]          # inp: (R, I padded to a multiple of 4)
            I = I0 if l == 0 else 2 * H
            Ip = inp.shape[1]
            whh = self.p("weight_hh_l%d" % l)
            dgx = gates                                  # overwritten in place, cell by cell
            # with weight-gradient GEMMs in flight on the side stream: SEPKERN_OVERLAP=1 carries 2 batch groups per
            # workgroup (the recurrence on half the CUs, GEMMs on the rest); =2 leaves the recurrence as it is and
            # lets GEMM blocks co-reside on its CUs (it leaves 124 VGPRs per SIMD lane and 69 KB of LDS free)
            mode = self.lstm_mode | self.bwd_bits | ((2 << 8) if (overlap and self.overlap_mode == 1 and l < L - 1) else 0)
            sl = slice(2 * l, 2 * l + 2)
            nbg = (B + 15) // 16
            dbias = torch.empty(nbg, 8 * H, device=dev)      # by-products of the recurrence: bias-gradient partials ...
            dg_first = torch.empty(2, B, 4 * H, device=dev)  # ... and the dG of the steps whose recurrent input is h0
            bargs = (dy, whh, gates, cs, c0[sl], lens, dgx, dh0[sl] if want_dstate else None,
                     dc0[sl] if want_dstate else None, T, B, H, mode)
            bkw = dict(dhn=dhn[sl] if dhn is not None else None, dcn=dcn[sl] if dcn is not None else None, bf16=self.bf16,
                       dbias=dbias, dg_first=dg_first)
            gw_hh = torch.empty(2, 4 * H, H, device=dev)     # rows gate-interleaved, like dgx (sk_gate_rows puts them back)
            gw_ih = torch.empty(8 * H, Ip, device=dev)
            split = self.bwd_split and overlap and not self.bf16 and T % 2 == 0 and T >= 16
            if split:
                # Split schedule: the recurrence in two launches of T/2 steps.  After the first, the forward direction's dgx
                # is final for t >= T/2 and the reverse direction's for t < T/2, so HALF of this layer's own weight-gradient
                # products (those rows, per direction: _wgrad_half) starts on the side stream beside the second launch --
                # the top layer's recurrence then hosts work too, and only half of layer 0's products is left for the end.
                ws = ops.lstm_bwd(*bargs, steps=(0, T // 2), **bkw)
                self.side.wait_stream(main)
                with torch.cuda.stream(self.side):
                    self._wgrad_half(1, dgx, y, inp, gw_hh, gw_ih, T, B, Ip, "gemm_side", True)
                ws = ops.lstm_bwd(*bargs, steps=(T // 2, T), **bkw)
            else:
                # bf16 (r03): the recurrence writes dgx a second time as bf16 -- the operand copy its three products read
                # (data gradient, dW_ih, dW_hh) -- instead of a cast pass over 4 x the bytes between recurrence and products
                twin = None
                if self.kmajor and self.bf16_twin:
                    rows, ld = ops.pad_to(R, 64) + 64, ops.pad_to(8 * H, 64)
                    twin = (torch.empty if ld == 8 * H else torch.zeros)(rows, ld, dtype=torch.bfloat16, device=dev)
                    if ld == 8 * H:
                        twin[R:].zero_()                 # whole K steps of zero rows behind the data (K-major factor)
                ws = ops.lstm_bwd(*bargs, dgx_bf16=twin, **bkw)
                if twin is not None:
                    ev = torch.cuda.Event()
                    ev.record(main)
                    d2 = dgx.view(R, 8 * H)
                    cache[("rowk", d2.data_ptr(), tuple(d2.shape))] = (twin, ev, main, d2)
            # The layer's weight-gradient products need the recurrence's dgx only: with SEPKERN_WGRAD_EARLY=1 the side stream is
            # released BEFORE the data gradient is issued on the main stream, so its blocks fill what that launch leaves free
            # (its tail, the launch gaps) instead of starting behind it.  r03, fp32: 35.85-35.97 vs 36.00-36.07 ms per step,
            # but the side launches then spend 4.5 ms per step queued behind the persistent data-gradient kernel, which
            # bench.py's per-launch events count as theirs (roofline.frac 0.505 instead of 0.57): opt-in.  bf16: slower
            # (14.0 vs 13.55 ms; short products that share operand copies across the two streams).
            early = overlap and l > 0 and not split and not self.bf16 and os.environ.get("SEPKERN_WGRAD_EARLY", "0") == "1"
            if early:
                self.side.wait_stream(main)
            if l > 0 or want_dx:                         # the only product the next recurrence (or the caller) waits for
                dy_next = torch.empty(R, Ip, device=dev)
                self._dgrad(cache, dgx.view(R, 8 * H), wih_gi, dy_next, "gemm_dgrad")
                if l == 0:
                    dx = (dy_next if Ip == I else dy_next[:, :I].contiguous()).view(T, B, I)
            stream = self.side if (overlap and l > 0) else main
            if stream is not main:
                if not early:
                    stream.wait_stream(main)
            elif overlap and split:
                main.wait_stream(self.side)      # layer 0 adds its second half onto the half sums the side stream made
            # (unsplit: layer 0's products share nothing with the side stream's but bf16 operand copies, which carry their
            # own events -- _copy -- so they start as soon as layer 0's recurrence ends)
            with torch.cuda.stream(stream):
                tag = "side" if stream is not main else "main"
                beside = stream is not main
                if split:
                    self._wgrad_half(2, dgx, y, inp, gw_hh, gw_ih, T, B, Ip, "gemm_" + tag, beside)
                    ops.gemm(dg_first, h0[sl], gw_hh, 4 * H, H, B, 4 * H, H, H, transA=True, accumulate=True, batch=2,
                             sA=B * 4 * H, sB=B * H, sC=4 * H * H, ws_tag="gemm_" + tag)      # the steps that start from h0
                else:
                    # dW_hh[d] = sum_t dG_t^T h_prev(t): the layer output shifted by one step in time (+ the h0 steps)
                    self._whh_grad(cache, dgx.view(R, 8 * H), y.view(R, 2 * H), h0[sl], dg_first, gw_hh, T, B, False, "gemm_" + tag, beside)
                    # dW_ih (both directions stacked as (8H, I)) = dgx^T x_in
                    self._wgrad(cache, dgx.view(R, 8 * H), inp, gw_ih, False, "gemm_" + tag, beside)
                ops.gate_rows(gw_hh, H, back=True, out=self.g("weight_hh_l%d" % l), accumulate=acc)
                ops.gate_rows(gw_ih, H, back=True, out=self.g("weight_ih_l%d" % l).view(8 * H, I), accumulate=acc, cols=I)
                db = torch.empty(8 * H, device=dev)
                ops.colsum(dbias, nbg, 8 * H, 8 * H, db, ws_tag="bn_" + tag)       # a few rows: the kernel did the sums
                put("bias_ih_l%d" % l, db.view(2, 4 * H))
                put("bias_hh_l%d" % l, db.view(2, 4 * H))
                keep += [db, dbias, dg_first, dgx, inp, y, gw_hh, gw_ih]
            if reducer is not None and l > 0:
                reducer.chunk(self.grad_full, *chunks["layer%d" % l], stream)
            if l > 0:
                dy = dy_next
        if overlap:
            main.wait_stream(self.side)
        del keep, cache
        self.guard.copy_(ops.lstm_sticky(ws))      # int32 -> float: non-zero = this step's gradients are garbage
        if reducer is not None:
            reducer.chunk(self.grad_full, *chunks["guard+layer0"], main)
        self.grads_fresh = False
        return dx, dh0, dc0
