"""The uPIT network (BLSTM -> BatchNorm1d -> Linear -> sigmoid) as a sequence of libsepkern calls.

Mirrors SepDNN.forward of the reference (archs/uPIT.py:129-147) and the backward torch's autograd
would run for it, on PACKED rows -- the layout of the PackedSequence the reference's collator builds
(archs/uPIT.py:46) and nn.LSTM consumes (:132): only the R = sum(lens) valid frames of a length-sorted
batch exist (sepkern.packing.Packing), so no product, statistic or store touches a padded frame.
All parameters live in ONE flat fp32 buffer (and all gradients in another):
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
        # bf16 products read ROW-MAJOR bf16 copies of their operands (sk_cast_bf16_rows; the backward recurrence writes the
        # copy of dgx itself, sk_hprev_rows the recurrent inputs): a factor whose rows are the contraction index (both factors
        # of a weight gradient, the weight matrix of a data gradient) enters the product K-major (sk_gemm_bf16_mm).  Hidden
        # sizes that are no multiple of 8 (none of BASELINE's) take the kernel that reads fp32 operands and rounds them on
        # the way into LDS -- same arithmetic, half the speed.
        self.nt = self.bf16 and hidden % 8 == 0
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

        # hand-off geometry / protocol of the persistent recurrences (speed only): "half,map,poll1,repflags,
        # spread,delay,tagged,split3".  Forward: streams dealt to XCD groups, one polling wave, first poll held back (delay 0 =
        # the library's choice); bf16: one flag per 128-byte line on top.  fp32 forward (r04): the product h W_hh^T by the EXACT
        # three-way bf16 split of both operands on the bf16 matrix pipe (split3 = 1: the six piece products per element pair that reach fp32's resolution, each exact,
        # fp32 accumulators -- an fp32 product in another summation order, no operand is perturbed; 35.6 vs 36.6 ms per
        # training step against the r03 default).  That r03 default, "the data is the flag" (SEPKERN_LSTM_FWD=0,1,1,0,0,8,1,0:
        # the exchanged h carries the step's epoch in its two low mantissa bits, <= 3 ulp on the operand), remains for hidden
        # sizes whose register slice does not fit the split (H > 896); SEPKERN_LSTM_FWD=0,1,1,0,0,0,0,0 is the plain fp32-MFMA
        # product with flags.  bench.py's config.numerics says which one ran.  Backward: the XCD map only.
        def variant(env, default):
            v = [int(x) for x in os.environ.get(env, default).split(",")]
            v += [0] * (9 - len(v))
            return ops.lstm_variant_bits(bool(v[0]), v[1], bool(v[2]), bool(v[3]), bool(v[4]), v[5], tagged=bool(v[6]), split3=bool(v[7]),
                                         xl8=bool(v[8]))
        # (bf16, r06: ninth field = XCD-local streams of 8 rows where the shape allows them -- 608 < H <= 896, B <= 32; mode bit 30)
        self.fwd_bits = variant("SEPKERN_LSTM_FWD", "0,1,1,0,1,0,0,0,1" if self.bf16 else
                                ("0,1,1,0,0,0,0,1" if hidden <= 896 else "0,1,1,0,0,8,1,0"))
        self.bwd_bits = variant("SEPKERN_LSTM_BWD", "0,1,0,0,0,31,0,0,1" if self.bf16 else "0,1,0,0,0,31,0,0")
        self.split3_fwd = bool(self.fwd_bits & 0x10000000) and not self.bf16 and hidden <= 896
        self.tagged_fwd = bool(self.fwd_bits & 0x20000000) and not self.bf16 and not self.split3_fwd
        # Weight-gradient GEMMs of layer l run on a side stream while layer l-1's recurrence runs on the main one: the
        # recurrence keeps its one-workgroup-per-CU grid and the GEMM blocks become CO-RESIDENT on its CUs -- a persistent
        # workgroup leaves >= 124 VGPRs per SIMD lane and >= 69 KB of LDS free, and the matrix pipe idle during its
        # hand-offs (42.9 -> 40.5 ms per step in r01).  SEPKERN_OVERLAP=0: everything on one stream (tests: bitwise equal).
        self.overlap = os.environ.get("SEPKERN_OVERLAP", "1") != "0"
        # BatchNorm folded into the Linear layer (fp32 path; the bf16 arithmetic is DEFINED with bn(y) and W rounded
        # separately, oracle/upit_bf16.py, so that path keeps the explicit normalisation)
        self.bn_fold = os.environ.get("SEPKERN_BN_FOLD", "1") == "1" and not self.bf16
        # fp32 GEMM kernel per stream (sk_gemm_f32_splitk's variant): "main,side".  Default 0,2.  Main stream: the library chooses --
        # split products on the bf16 matrix pipe (three-way bf16 split, six piece products) wherever the operands allow: the
        # kernel that splits once per element while staging a 256 x 128 tile for the large unsplit products, the 128 x 128 split
        # kernel otherwise.  Beside a recurrence: the 128 x 128 split kernel always (122 VGPRs, 32 KB of LDS: it fits on a CU next
        # to a persistent workgroup, the 256 x 128 one does not), which finishes its products sooner than r04's register-staged
        # fp32-MFMA kernel (12.2 vs 12.8 ms of backward recurrences).  r05: 34.7 -> 30.0 ms per step.  "8,1" = the r04 arrangement.
        v = [int(x) for x in os.environ.get("SEPKERN_GEMM_VARIANTS", "0,2").split(",")]
        self.var_main, self.var_side = v[0], v[1]
        # (diagnostics: optional third / fourth field = the variant of the forward projections / of the data gradients alone)
        self.var_proj = v[2] if len(v) > 2 else None
        self.var_dgrad = v[3] if len(v) > 3 else None
        self.pad_in = int(os.environ.get("SEPKERN_PAD_IN", "16"))   # diagnostics: 4 = the r04 padding of the input width
        # r06, "operands that arrive split" (fp32): the weight gradients -- T/N products of activation / gradient matrices, most of
        # them hosted beside a backward recurrence -- read operands their producers already cut into the three bf16 pieces: the
        # backward recurrence writes dgx's planes with the fp32 values, the layer inputs and recurrent inputs are split once in the
        # forward pass (side stream), and sk_gemm_pl3_tn multiplies the planes with no VALU work in its K loop.  Bit for bit the
        # products of the 128 x 128 split kernel.  SEPKERN_WGRAD_PLANES=0: that kernel on the fp32 operands (the r05 arrangement).
        # (hidden sizes that are no multiple of 8: the second direction's columns of a plane would start off a 16-byte boundary)
        self.wgrad_planes = os.environ.get("SEPKERN_WGRAD_PLANES", "1") != "0" and not self.bf16 and hidden % 8 == 0
        # ... and whether the backward recurrences keep their CUs to themselves while those products run (sk_lstm_bwd mode bit 17):
        # "auto" = on ragged batches -- there half the chip falls idle once the short batch group's streams have left the grid, the
        # weight gradients run on those CUs for free, and co-resident they would only slow the long group's chain (ragged 28.98 ->
        # 28.48 ms exclusive, 29.40 co-resident); on uniform batches nothing falls idle and co-residency wins (28.27 -> 27.95, 28.11
        # exclusive; profiles/r06_wgrad_planes.txt).  "0" / "1": never / always.
        self.bwd_exclusive = os.environ.get("SEPKERN_BWD_EXCLUSIVE", "auto")
        # diagnostics: sk_lstm_bwd mode bit 29 for the top layer's launch (1) or every layer's (2) -- read by timing-only builds
        # of the recurrence alone (csrc/lstm.hip SK_BWD_BOUND38); the product library ignores the bit
        self.bwd_diag = int(os.environ.get("SEPKERN_BWD_DIAG", "0"))
        self.side = None
        self.grads_fresh = True        # True: next backward may overwrite instead of accumulate
        self.version = 0               # bumped by whoever writes the parameters (ClipAdam, load_state_dict): see backward()

    def p(self, name):
        return self.layout.view(self.flat, name)

    def param_version(self):
        """Changes whenever the parameters were written: `version` is bumped by sepkern.optim.ClipAdam (its kernel writes
        through raw pointers), torch's own counter covers in-place torch ops on the flat buffer or its views."""
        return (self.version, self.flat._version)

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

    def pad_row(self):
        """(out_dim,) the mask the reference's network shows at a zero-padded frame with the statistics of the last forward:
        sigmoid(lin(bn(0))) (archs/uPIT.py:135-144 run BatchNorm / Linear / sigmoid over the padded (B, T_max) grid).  The
        engine never computes padded frames; callers that hand out padded masks (SepDNN.forward, run_net) fill them in."""
        mean, var = self._last_bn
        Wf, bf_, _, _ = ops.bn_fold(self.p("lin.weight"), self.p("lin.bias"), mean, var, self.p("bn.weight"), self.p("bn.bias"),
                                    self.eps)
        zero = torch.zeros(1, 2 * self.H, device=self.device)
        row = torch.empty(1, self.O, device=self.device)
        ops.gemm(zero, Wf, row, 1, self.O, 2 * self.H, 2 * self.H, 2 * self.H, self.O, transB=True, bias=bf_, act=1)
        return row.view(-1)

    def _side_stream(self, dev):
        if self.side is None:
            self.side = torch.cuda.Stream(device=dev)
        return self.side

    # ------------------------------------------------------------------ the three kinds of product
    # All of them over packed rows: operands are (Rp, C) row buffers with zero tail rows; row-parallel products write the
    # R valid rows, contractions over rows run over all Rp (whole K steps).
    # fp32: the fp32 MFMA kernels on the fp32 tensors.  bf16: sk_gemm_bf16_nt / _mm on bf16 copies; `cache` (one dict per
    # step) holds the copies already made, keyed by (address, shape) of the fp32 tensor, and keeps them alive.
    @staticmethod
    def _copy(cache, t2d):
        """Row-major bf16 copy of t2d (whole K steps of zero rows behind it: it may serve as a K-major factor), made once per
        step.  A copy made on one stream and used on the other is ordered by ITS OWN event: the user waits for that cast, not
        for everything the maker's stream has queued behind it."""
        key = (t2d.data_ptr(), tuple(t2d.shape))
        cur = torch.cuda.current_stream()
        ent = cache.get(key)
        if ent is None:
            c = ops.cast_bf16(t2d, rows=ops.pad_to(t2d.shape[0], 64) + 64)
            ev = torch.cuda.Event()
            ev.record(cur)
            # the entry holds the fp32 SOURCE too: its address cannot be recycled for another tensor of the same shape
            # while the copy is cached (sources are not written between their uses within a step)
            cache[key] = ent = (c, ev, cur, t2d)
        elif ent[2] != cur:
            cur.wait_event(ent[1])
            ent[0].record_stream(cur)
        return ent[0]

    def _proj(self, cache, inp2d, w, out2d, bias, R, act=0):
        """out[:R] (R, N) = act(inp[:R] (R, K) w (N, K)^T + bias)."""
        K = inp2d.shape[1]
        N = w.shape[0]
        if not self.nt:
            ops.gemm(inp2d, w, out2d, R, N, K, inp2d.stride(0), K, N, transB=True, bias=bias, act=act, bf16=self.bf16,
                     variant=self.var_main if self.var_proj is None else self.var_proj)
            return
        a, b = self._copy(cache, inp2d), self._copy(cache, w)
        # (not the stream-K kernel: with 1400 tiles of 28 K steps its fix-up costs more than the sixth partial round it saves --
        # main-stream products 3.03 vs 3.11 ms per step; the data gradients' 350 tiles of 112 steps are where it pays)
        ops.gemm_bf16_nt(a, b, out2d, R, N, a.shape[1], a.shape[1], b.shape[1], N, bias=bias, act=act)

    def _dgrad(self, cache, dout2d, w, out2d, R, ws_tag):
        """out[:R] (R, K) = dout[:R] (R, N) w (N, K)."""
        N = dout2d.shape[1]
        K = w.shape[1]
        if not self.nt:
            # large data gradients unsplit (stream-K / 256 x 128 tiles: 125.5-134.6 TFLOP/s against 119-120 for two K slices)
            sk = 1 if (not self.bf16 and R >= 4096 and K >= 1024 and N % 16 == 0) else 0
            ops.gemm(dout2d, w, out2d, R, K, N, N, K, K, splitk=sk, ws_tag=ws_tag, bf16=self.bf16,
                     variant=self.var_main if self.var_dgrad is None else self.var_dgrad)
            return
        # w (N, K) is the K-major B of the product as it lies (contraction over its rows, padded with zero rows up to
        # dout's zero-padded width)
        a, b = self._copy(cache, dout2d), self._copy(cache, w)
        ops.gemm_bf16_mm(a, b, out2d, R, K, a.shape[1], a.shape[1], b.shape[1], K, b_kmajor=True, splitk=0, ws_tag=ws_tag,
                         streamk=True)

    def _wgrad(self, cache, dout2d, inp2d, gw, acc, ws_tag, beside=False, batch=1, sA=0, sB=0, sC=0, N=None, K=None):
        """gw (N, K) [+]= dout (Rp, N)^T inp (Rp, K), contraction over all Rp rows (zero tails).  batch = 2 with operand
        strides: the two directions of the recurrent weight gradient in one launch.  beside=True: the product runs
        co-resident with a recurrence (side stream): the register-staged GEMM kernel, which leaves the recurrence more of
        the matrix pipe than the LDS-DMA one does (measured: same step time with either, 2 ms longer recurrences with the
        latter)."""
        if isinstance(dout2d, ops.Planes):              # operands that arrive split (fp32): both factors as planes
            N = dout2d.C if N is None else N
            K = inp2d.C if K is None else K
            ops.gemm_pl3_tn(dout2d, inp2d, gw, N, K, dout2d.rows, accumulate=acc, batch=batch, sA=sA, sB=sB, sC=sC, splitk=0, ws_tag=ws_tag)
            return
        Rp = dout2d.shape[0]
        N = dout2d.shape[1] if N is None else N
        K = inp2d.shape[1] if K is None else K
        if not self.nt:
            ops.gemm(dout2d, inp2d, gw, N, K, Rp, dout2d.stride(0), inp2d.stride(0), K, transA=True, accumulate=acc, splitk=0,
                     batch=batch, sA=sA, sB=sB, sC=sC, ws_tag=ws_tag, bf16=self.bf16, variant=self.var_side if beside else self.var_main)
            return
        # both factors K-major as they lie (their rows are the contraction index)
        a = dout2d if dout2d.dtype == torch.bfloat16 else self._copy(cache, dout2d)
        b = inp2d if inp2d.dtype == torch.bfloat16 else self._copy(cache, inp2d)
        ops.gemm_bf16_mm(a, b, gw, N, K, ops.pad_to(Rp, 64), a.shape[1], b.shape[1], K, a_kmajor=True, b_kmajor=True,
                         accumulate=acc, batch=batch, sA=sA, sB=sB, sC=sC, splitk=0, ws_tag=ws_tag, streamk=not beside and batch == 1)

    def _hprev(self, y, h0l, pk):
        """The recurrent inputs of a layer's packed rows (sk_hprev_rows), as the operand its recurrent weight gradient reads."""
        H = self.H
        if self.nt:
            ld = ops.pad_to(2 * H, 64)
            rows = ops.pad_to(pk.Rp, 64) + 64
            hp = (torch.empty if ld == 2 * H else torch.zeros)(rows, ld, dtype=torch.bfloat16, device=y.device)
            if ld == 2 * H:
                hp[pk.R:].zero_()
        else:
            hp = pk.rows(2 * H)
        return ops.hprev_rows(y, h0l, pk, H, hp)

    # ------------------------------------------------------------------ forward
    def forward(self, x2d, pk, h0, c0, training, save, want_state=False):
        """x2d (>= R, in_dim) packed rows of the batch `pk` (sepkern.packing.Packing), h0/c0 (2L,B,H) in sorted order ->
        (mask (R, out_dim) packed, hn, cn (2L,B,H) or None, ctx or None).  ctx feeds backward(); several may be alive
        (the RSH arch runs the network num_spk times per batch)."""
        if x2d.dim() != 2 or x2d.shape[1] != self.I or x2d.shape[0] < pk.R:
            raise SepkernError("forward: input is %s, expected (>= %d, %d) packed rows" % (tuple(x2d.shape), pk.R, self.I))
        T, B, R, Rp = pk.T, pk.B, pk.R, pk.Rp
        H, L, O, I0 = self.H, self.L, self.O, self.I
        lens = pk.lens
        offs = None if pk.uniform else pk.offs      # equal lengths: packed rows ARE the (T, B) grid, no table to read
        dev = x2d.device
        saved = []
        cache = {}
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
            # ... and in the same pass pads an input width that is no multiple of 16 (F = 257 -> 272; RSH 514 -> 528) with zero
            # columns: the rows of both operands of the layer-0 products are then 16-byte aligned AND K is a whole number of
            # 16-deep K steps, so that the layer-0 projection and weight gradient take the split-product kernels too
            # (r04 padded to 260: float4 fetches, fp32-MFMA kernels).  Layer 0 only: the layers above read y (2H columns, a
            # multiple of 4 by construction) as it lies, and their data gradient IS the next recurrence's dy (2H columns)
            Ip = ops.pad_to(I, self.pad_in if l == 0 else 4)
            wih_gi = ops.gate_rows(wih.view(8 * H, I), H, out=torch.empty(8 * H, Ip, device=dev), cols=I)
            return wih_gi, ops.gate_rows(bsum, H), Ip

        main = torch.cuda.current_stream(dev)
        use_side = self.overlap and self.lstm_mode == 0 and L > 1
        side = self._side_stream(dev) if (use_side and save) else self.side
        # The gate-interleaved copies of W_ih and the summed biases of the layers above the first depend on the weights only:
        # they are made on the side stream beside layer 0's projection and recurrence instead of in front of each layer's
        # projection on the main stream (r03: 0.13 ms of small kernels per step)
        ahead, ahead_ev = {}, None
        if side is not None and use_side:
            side.wait_stream(main)
            with torch.cuda.stream(side):
                for l in range(1, L):
                    ahead[l] = weights_of(l, 2 * H, ws_tag="bn_side")
                    for t_ in ahead[l][:2]:
                        t_.record_stream(main)
                ahead_ev = torch.cuda.Event()
                ahead_ev.record(side)
        inp, I = x2d, I0
        # (planes only under the split arithmetic: with fp32-MFMA variants asked for -- 8 / 1, the reference's literal arithmetic --
        # the weight gradients stay fp32-MFMA products of the fp32 operands)
        planes = save and self.wgrad_planes and self.var_side in (0, 2) and self.var_main in (0, 2, 9)
        inp_pl = None                                       # the planes of the current layer's input (made beside the layer below)
        for l in range(L):
            whh = self.p("weight_hh_l%d" % l)
            if l in ahead:
                if ahead_ev is not None:
                    main.wait_event(ahead_ev)
                    ahead_ev = None
                wih_gi, bsum, Ip = ahead[l]
            else:
                wih_gi, bsum, Ip = weights_of(l, I)
            # The weight gradients contract over all Rp rows of `inp` and rely on ZERO tail rows R..Rp.  The layers above
            # the first read y = pk.rows() (zero tail by construction); the caller's x2d is only known to hold R rows, so
            # it is taken as it is only when there is no tail (R == Rp) -- whatever lies behind row R of a larger buffer
            # (another batch, NaN) never enters a product.
            if Ip != I or inp.shape[0] < Rp or not inp.is_contiguous() or (l == 0 and Rp > R):
                inp = ops.pad_rows(inp[:R], Ip, rows=Rp)   # one pass, no memset (F = 257 -> 260; tail rows zero)
            gx = pk.rows(8 * H)                              # (Rp, 2, 4H): projections -> saved gates -> dgx, in place
            if planes and l == 0:
                # the (padded) network input as planes, for layer 0's weight gradient: on the side stream beside its projection
                if use_side:
                    ev = torch.cuda.Event()
                    ev.record(main)
                    side.wait_event(ev)
                    with torch.cuda.stream(side):
                        inp_pl = ops.split_rows(inp, R)
                    inp_pl.t.record_stream(main)
                    inp.record_stream(side)
                else:
                    inp_pl = ops.split_rows(inp, R)
            self._proj(cache, inp, wih_gi, gx, bsum, R)
            y = pk.rows(2 * H)
            cs = torch.empty(Rp, 2 * H, device=dev) if save else None
            h0l = h0[2 * l:2 * l + 2]
            ws = ops.lstm_fwd(gx, whh, h0l, c0[2 * l:2 * l + 2], lens, y, gx if save else None, cs,
                              hn[2 * l:2 * l + 2] if want_state else None, cn[2 * l:2 * l + 2] if want_state else None,
                              T, B, H, self.lstm_mode | self.fwd_bits, bf16=self.bf16, offs=offs, rows=R)
            hp = hp_ev = y_pl = None
            if save:
                # the recurrent inputs of the layer's rows, for its recurrent weight gradient: gathered HERE, on the side
                # stream beside the next layer's projection, not in the backward pass where the side stream is the bound
                # (fp32, operands that arrive split: cut into their planes there too, and so is y -- the next layer's input, or the
                # Linear layer's: the other factor of their weight gradients)
                def side_work():
                    hp_ = self._hprev(y, h0l, pk)
                    if planes:
                        return ops.split_rows(hp_, R), ops.split_rows(y, R)
                    return hp_, None
                if use_side:
                    ev = torch.cuda.Event()
                    ev.record(main)
                    side.wait_event(ev)
                    with torch.cuda.stream(side):
                        hp, y_pl = side_work()
                        hp_ev = torch.cuda.Event()
                        hp_ev.record(side)
                    for t_ in (hp, y_pl):
                        if t_ is not None:
                            (t_.t if planes else t_).record_stream(main)
                    y.record_stream(side)
                else:
                    hp, y_pl = side_work()
            saved.append((inp, gx, cs, y, wih_gi, hp, hp_ev, inp_pl))
            inp, I, inp_pl = y, 2 * H, y_pl
        if not save and not skdist.is_parallel():
            # inference: the caller copies the masks to the host next, a sync costs nothing.  Under data parallelism a
            # raise on ONE rank would leave the others waiting in their next collective: there the sticky word stays set
            # and the driver reports it on every rank together (steps/train_qsub.py::validation_pass)
            ops.lstm_status(ws)
        y_top = inp
        count = B * T                 # BatchNorm1d sees the zero-padded (B, 2H, T_max) grid (archs/uPIT.py:135-138)
        if training:
            mean = torch.empty(2 * H, device=dev)
            var = torch.empty(2 * H, device=dev)
            ops.bn_stats(y_top, mean, var, rows=R, count=count)
            bn_count = float(count)
            if self.sync_bn:                             # statistics of the global batch (one all-gather)
                mean, var, bn_count = skdist.combine_bn_stats(mean, var, B, T)
            # (guarded: after a timed-out launch y is garbage and must not reach running statistics a checkpoint will hold)
            ops.bn_update_running(mean, var, self.running_mean, self.running_var, int(bn_count), self.momentum,
                                  guard=ops.lstm_sticky(ws))
        else:
            mean, var, bn_count = self.running_mean, self.running_var, float(count)
        self._last_bn = (mean, var)
        mask = torch.empty(Rp, O, device=dev)
        xbn = fold = None
        if self.bn_fold:
            # BatchNorm folded into the Linear weights (sk_bn_fold): mask = sigmoid(y Wf^T + bf), the
            # normalised activations are never written (one 92 MB pass less, forward and backward)
            Wf, bf_, s_, t_ = ops.bn_fold(self.p("lin.weight"), self.p("lin.bias"), mean, var, self.p("bn.weight"),
                                          self.p("bn.bias"), self.eps)
            self._proj(cache, y_top, Wf, mask, bf_, R, act=1)
            fold = (s_, t_)
        else:
            xbn = pk.rows(2 * H)
            ops.bn_apply(y_top[:R], mean, var, self.p("bn.weight"), self.p("bn.bias"), xbn, self.eps)
            self._proj(cache, xbn, self.p("lin.weight"), mask, self.p("lin.bias"), R, act=1)
        ctx = None
        if save:
            ctx = dict(saved=saved, mean=mean, var=var, bn_count=bn_count, xbn=xbn, fold=fold, mask=mask, pk=pk, h0=h0, ytop_pl=inp_pl,
                       c0=c0, training=training, version=self.param_version(),
                       # bf16: the backward pass multiplies the SAME row-major copies of the weights and of every layer
                       # input (none of them is written in between), so they are made once per step
                       cache=cache if self.nt else None)
        return mask[:R], hn, cn, ctx

    # ------------------------------------------------------------------ backward
    def backward(self, ctx, dmask, dhn=None, dcn=None, want_dx=False, want_dstate=False, reducer=None):
        """Parameter gradients (into the flat gradient buffer) from dmask (R, out_dim) packed and, optionally, the
        gradient wrt the final state (dhn, dcn (2L,B,H)).  Returns (dx (R, in_dim) packed or None, dh0, dc0 or None).
        reducer (sepkern.dist.GradReducer, data-parallel runs with SEPKERN_DP_OVERLAP=1, last backward of a step only):
        every chunk of ParamLayout.grad_chunks() is handed over as soon as the kernels that complete it are enqueued."""
        if ctx is None:
            raise SepkernError("backward called without a saved forward")
        if not ctx["training"]:
            raise SepkernError("backward through eval-mode BatchNorm is not built")
        pk = ctx["pk"]
        T, B, R, Rp = pk.T, pk.B, pk.R, pk.Rp
        H, L, O, I0 = self.H, self.L, self.O, self.I
        lens = pk.lens
        offs = None if pk.uniform else pk.offs      # equal lengths: packed rows ARE the (T, B) grid, no table to read
        dev = dmask.device
        acc = not self.grads_fresh
        h0, c0 = ctx["h0"], ctx["c0"]
        dmask = dmask.contiguous()

        def put(name, val):           # small vectors produced by non-accumulating kernels
            if acc:
                self.g(name).add_(val)
            else:
                self.g(name).copy_(val)

        overlap = self.overlap and L > 1 and self.lstm_mode == 0
        main = torch.cuda.current_stream(dev)
        side = self._side_stream(dev) if overlap else None
        keep = []                                        # tensors used on the side stream stay alive until the join
        dz = pk.rows(O)
        ops.sigmoid_bwd(dmask[:R], ctx["mask"][:R], dz)
        # bf16 operand copies (see _copy): the forward's, unless the parameters were written since (a backward of a ctx saved
        # before an optimizer step would otherwise multiply stale bf16 weight copies against the new fp32 weights)
        cache = ctx.get("cache") if ctx.get("version") == self.param_version() else None
        cache = {} if cache is None else cache
        dxbn = torch.empty(Rp, 2 * H, device=dev)
        # (folded or not, the data gradient contracts with the UNFOLDED weight: dxbn is the gradient wrt bn(y))
        self._dgrad(cache, dz, self.p("lin.weight"), dxbn, R, "gemm")
        y_top = ctx["saved"][-1][3]
        dy = torch.empty(Rp, 2 * H, device=dev)
        dgamma = torch.empty(2 * H, device=dev)
        dbeta = torch.empty(2 * H, device=dev)
        ops.bn_bwd_sums(dxbn[:R], y_top[:R], ctx["mean"], ctx["var"], dgamma, dbeta, self.eps)
        put("bn.weight", dgamma)                         # local sums: the flat all-reduce adds the ranks up later
        put("bn.bias", dbeta)
        if self.sync_bn:                                 # dx needs the sums over the global batch (one all-reduce)
            dgamma, dbeta = skdist.allreduce_bn_sums(dgamma, dbeta)
        ops.bn_bwd_apply(dxbn[:R], y_top[:R], ctx["mean"], ctx["var"], self.p("bn.weight"), dgamma, dbeta, dy, ctx["bn_count"],
                         self.eps)
        del dxbn
        # the Linear layer's own gradients are needed by nobody before clip+Adam: side stream, next to the top
        # layer's recurrence (enqueued HERE, after the BatchNorm backward on the main stream: issued earlier they ran
        # beside those short critical-path kernels and slowed them by 0.1 ms)
        stream = side if overlap else main
        if stream is not main:
            stream.wait_stream(main)
        with torch.cuda.stream(stream):
            tag = "_side" if overlap else ""
            if ctx["fold"] is not None:
                # dW = dz^T bn(y) = (dz^T y) diag(s) + colsum(dz) t^T  (sk_bn_unfold_grad): the product runs against y itself
                G = torch.empty(O, 2 * H, device=dev)
                dzsum = torch.empty(O, device=dev)
                if ctx.get("ytop_pl") is not None:
                    dz_pl = ops.split_rows(dz, R)            # (R x O: a tenth of a layer's dgx)
                    self._wgrad(cache, dz_pl, ctx["ytop_pl"], G, False, "gemm" + tag, beside=overlap)
                    keep.append(dz_pl)
                else:
                    self._wgrad(cache, dz, y_top, G, False, "gemm" + tag, beside=overlap)
                ops.colsum(dz, R, O, O, dzsum, ws_tag="bn" + tag)
                ops.bn_unfold_grad(G, dzsum, ctx["fold"][0], ctx["fold"][1], self.g("lin.weight"), accumulate=acc)
                if acc:
                    self.g("lin.bias").add_(dzsum)
                else:
                    self.g("lin.bias").copy_(dzsum)
                keep += [G, dzsum]
            else:
                self._wgrad(cache, dz, ctx["xbn"], self.g("lin.weight"), acc, "gemm" + tag, beside=overlap)
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
            inp, gates, cs, y, wih_gi, hp, hp_ev, inp_pl = ctx["saved"][l]      # inp: (Rp, I padded to a multiple of 4)
            I = I0 if l == 0 else 2 * H
            Ip = inp.shape[1]
            whh = self.p("weight_hh_l%d" % l)
            dgx = gates                                  # overwritten in place, cell by cell
            mode = self.lstm_mode | self.bwd_bits
            if self.bwd_diag == 2 or (self.bwd_diag == 1 and l == L - 1):
                mode |= 0x20000000
            if inp_pl is not None and (self.bwd_exclusive == "1" or (self.bwd_exclusive == "auto" and not pk.uniform and B > 16)):
                mode |= 0x20000                          # exclusive: see __init__
            sl = slice(2 * l, 2 * l + 2)
            nbg = (B + 15) // 16
            dbias = torch.empty(nbg, 8 * H, device=dev)      # by-product of the recurrence: bias-gradient partials
            gw_hh = torch.empty(2, 4 * H, H, device=dev)     # rows gate-interleaved, like dgx (sk_gate_rows puts them back)
            gw_ih = torch.empty(8 * H, Ip, device=dev)
            # bf16: the recurrence writes dgx a second time as bf16 -- the operand copy its three products read (data
            # gradient, dW_ih, dW_hh) -- instead of a cast pass over 4 x the bytes between recurrence and products
            twin = None
            if inp_pl is not None:                       # fp32: dgx also as its three bf16 planes, for the two weight gradients
                twin = ops.Planes.empty(R, 8 * H, dev)
            if self.nt:
                rows, ld = ops.pad_to(Rp, 64) + 64, ops.pad_to(8 * H, 64)
                twin = (torch.empty if ld == 8 * H else torch.zeros)(rows, ld, dtype=torch.bfloat16, device=dev)
                if ld == 8 * H:
                    twin[R:].zero_()                     # whole K steps of zero rows behind the data (K-major factor)
            ws = ops.lstm_bwd(dy, whh, gates, cs, c0[sl], lens, dgx, dh0[sl] if want_dstate else None,
                              dc0[sl] if want_dstate else None, T, B, H, mode,
                              dhn=dhn[sl] if dhn is not None else None, dcn=dcn[sl] if dcn is not None else None, bf16=self.bf16,
                              dbias=dbias, dgx_bf16=twin, offs=offs, rows=R)
            if twin is not None and self.nt:
                ev = torch.cuda.Event()
                ev.record(main)
                cache[(dgx.data_ptr(), tuple(dgx.shape))] = (twin, ev, main, dgx)
            if l > 0 or want_dx:                         # the only product the next recurrence (or the caller) waits for
                dy_next = torch.empty(Rp, Ip, device=dev)
                self._dgrad(cache, dgx, wih_gi, dy_next, R, "gemm_dgrad")
                if l == 0:
                    dx = dy_next[:R] if Ip == I else dy_next[:R, :I].contiguous()
            stream = side if (overlap and l > 0) else main
            if stream is not main:
                stream.wait_stream(main)
            elif hp_ev is not None:
                main.wait_event(hp_ev)                   # (layer 0's products run on the main stream; long since recorded)
            # (layer 0's products share nothing with the side stream's but bf16 operand copies, which carry their own
            # events -- _copy -- so they start as soon as layer 0's recurrence ends)
            with torch.cuda.stream(stream):
                tag = "side" if stream is not main else "main"
                beside = stream is not main
                # dW_hh[d] = sum_rows dG[:, d]^T hprev[:, d-half]: both directions as one batched launch
                pl = inp_pl is not None
                self._wgrad(cache, twin if pl else dgx, hp, gw_hh, False, "gemm_" + tag, beside, batch=2, sA=4 * H, sB=H, sC=4 * H * H,
                            N=4 * H, K=H)
                # dW_ih (both directions stacked as (8H, I)) = dgx^T x_in
                self._wgrad(cache, twin if pl else dgx, inp_pl if pl else inp, gw_ih, False, "gemm_" + tag, beside)
                if pl:
                    keep += [twin, inp_pl]
                ops.gate_rows(gw_hh, H, back=True, out=self.g("weight_hh_l%d" % l), accumulate=acc)
                ops.gate_rows(gw_ih, H, back=True, out=self.g("weight_ih_l%d" % l).view(8 * H, I), accumulate=acc, cols=I)
                db = torch.empty(8 * H, device=dev)
                ops.colsum(dbias, nbg, 8 * H, 8 * H, db, ws_tag="bn_" + tag)       # a few rows: the kernel did the sums
                put("bias_ih_l%d" % l, db.view(2, 4 * H))
                put("bias_hh_l%d" % l, db.view(2, 4 * H))
                keep += [db, dbias, dgx, inp, y, hp, gw_hh, gw_ih]
            if reducer is not None and l > 0:
                reducer.chunk(self.grad_full, *chunks["layer%d" % l], stream)
            if l > 0:
                dy = dy_next
        if overlap:
            main.wait_stream(side)
        del keep, cache
        self.guard.copy_(ops.lstm_sticky(ws))      # int32 -> float: non-zero = this step's gradients are garbage
        if reducer is not None:
            reducer.chunk(self.grad_full, *chunks["guard+layer0"], main)
        self.grads_fresh = False
        return dx, dh0, dc0
