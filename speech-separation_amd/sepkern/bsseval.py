"""BSS Eval (v3) source-level SDR / SIR / SAR, the metric the reference scores with
(`mir_eval.separation.bss_eval_sources`, steps/evaluate_sources.py:57, steps/evaluate_oracle.py:118,143).

mir_eval is a third-party dependency of the reference that is neither vendored under /root/reference nor
installed here (no version pinned by the reference).  This is a restatement of the published algorithm
(Vincent, Gribonval, Fevotte, "Performance measurement in blind audio source separation", IEEE TASLP 2006; the
`bss_eval_sources` variant with a time-invariant distortion filter of 512 taps):

  an estimate e is split, by least-squares projections onto delayed copies of the true sources r_1..r_S, into
      s_target = P_j e            (projection onto the span of r_j delayed by 0..L-1 samples: "allowed distortion")
      e_interf = P_all e - P_j e  (what the other sources explain)
      e_artif  = e - P_all e      (what no source explains)
  SDR = |s_target|^2 / |e_interf + e_artif|^2,  SIR = |s_target|^2 / |e_interf|^2,
  SAR = |s_target + e_interf|^2 / |e_artif|^2   (all in dB; a zero denominator gives +inf),
  and the estimate -> source assignment is the permutation with the highest MEAN SIR.

Organisation (differs from mir_eval's, same numbers): the Gram matrix of the delayed sources depends on the sources
only, so it is built and Cholesky-factorised ONCE per utterance (and once per single source), and every estimate
is then projected with two triangular solves -- mir_eval rebuilds and re-solves the system for each of the S^2
(estimate, source) pairs.  Off the hot path: numpy / scipy on the host.
"""
import itertools

import numpy as np
import scipy.linalg
import scipy.signal

FILTER_TAPS = 512        # bss_eval_sources' fixed allowed-distortion filter length


def _db(num, den):
    if den == 0:
        return np.inf
    return 10.0 * np.log10(num / den)


class _DelayedSpan:
    """Least-squares projector onto span{ r_i delayed by t : i in sources, 0 <= t < taps } for signals of length n
    (zero-padded to n + taps - 1)."""

    def __init__(self, spectra, rows, n, taps, n_fft):
        self.rows, self.n, self.taps, self.n_fft = list(rows), n, taps, n_fft
        self.spectra = spectra
        k = len(self.rows)
        gram = np.empty((k * taps, k * taps))
        for a, i in enumerate(self.rows):
            for b, j in enumerate(self.rows[a:], start=a):
                # c[d] = sum_n r_i[n] r_j[n + d]; entry (t, t') of the block is <r_i(.-t), r_j(.-t')> = c[t - t'] ... with
                # both signals real: <r_i(.-t), r_j(.-t')> = sum_n r_i[n-t] r_j[n-t'] = c_ij[t - t'] where
                # c_ij[d] = sum_m r_i[m] r_j[m + d]
                c = np.fft.irfft(np.conj(spectra[i]) * spectra[j], n_fft)
                col = c[:taps]                              # d = t - t' >= 0 down the first column (t' = 0)
                row = np.concatenate((c[:1], c[:-taps:-1]))  # d = -t' along the first row (t = 0)
                block = scipy.linalg.toeplitz(col, row)
                gram[a * taps:(a + 1) * taps, b * taps:(b + 1) * taps] = block
                if b != a:
                    gram[b * taps:(b + 1) * taps, a * taps:(a + 1) * taps] = block.T
        self.gram = gram
        try:
            self.chol = scipy.linalg.cho_factor(gram, lower=True, check_finite=False)
        except np.linalg.LinAlgError:
            self.chol = None                                # rank-deficient sources (e.g. a silent one): least squares

    def coefficients(self, est_spectrum):
        taps = self.taps
        rhs = np.empty(len(self.rows) * taps)
        for a, i in enumerate(self.rows):
            # <r_i(.-t), e> = sum_n r_i[n - t] e[n] = c_ie[t]
            rhs[a * taps:(a + 1) * taps] = np.fft.irfft(np.conj(self.spectra[i]) * est_spectrum, self.n_fft)[:taps]
        if self.chol is not None:
            return scipy.linalg.cho_solve(self.chol, rhs, check_finite=False)
        return np.linalg.lstsq(self.gram, rhs, rcond=None)[0]

    def project(self, est_spectrum):
        """P e as a time signal of length n + taps - 1."""
        coef = self.coefficients(est_spectrum).reshape(len(self.rows), self.taps)
        out = np.zeros(self.n_fft // 2 + 1, dtype=complex)
        for a, i in enumerate(self.rows):
            out += np.fft.rfft(coef[a], self.n_fft) * self.spectra[i]
        return np.fft.irfft(out, self.n_fft)[:self.n + self.taps - 1]


def bss_eval_sources(reference_sources, estimated_sources, compute_permutation=True, taps=FILTER_TAPS):
    """(sdr, sir, sar, perm), each of length S; entry j scores the estimate assigned to true source j, and
    estimated source perm[j] is the one assigned to true source j (mir_eval's convention).  Inputs (S, n) arrays
    (a 1-D array is one source)."""
    ref = np.atleast_2d(np.asarray(reference_sources, dtype=np.float64))
    est = np.atleast_2d(np.asarray(estimated_sources, dtype=np.float64))
    if ref.shape != est.shape:
        raise ValueError("reference and estimated sources must have the same shape, got %s and %s" % (ref.shape, est.shape))
    S, n = ref.shape
    if not np.all(np.any(ref != 0, axis=1)):
        raise ValueError("all-zero reference source: BSS Eval metrics are undefined")
    if not np.all(np.any(est != 0, axis=1)):
        raise ValueError("all-zero estimated source: BSS Eval metrics are undefined")
    n_fft = 1 << int(np.ceil(np.log2(n + taps)))            # linear (not circular) correlations up to |lag| < taps
    ref_f = [np.fft.rfft(r, n_fft) for r in ref]
    est_f = [np.fft.rfft(e, n_fft) for e in est]
    span_all = _DelayedSpan(ref_f, range(S), n, taps, n_fft)
    span_one = [span_all if S == 1 else _DelayedSpan(ref_f, [j], n, taps, n_fft) for j in range(S)]
    est_pad = np.zeros((S, n + taps - 1))
    est_pad[:, :n] = est

    explained = [span_all.project(est_f[k]) for k in range(S)]          # P_all e_k
    pairs = [(k, j) for k in range(S) for j in range(S)] if compute_permutation else [(j, j) for j in range(S)]
    sdr = np.full((S, S), np.nan)
    sir = np.full((S, S), np.nan)
    sar = np.full((S, S), np.nan)
    for k, j in pairs:                                                   # estimate k scored against true source j
        target = span_one[j].project(est_f[k])
        interf = explained[k] - target
        artif = est_pad[k] - explained[k]
        t2 = float(np.dot(target, target))
        sdr[k, j] = _db(t2, float(np.sum((interf + artif) ** 2)))
        sir[k, j] = _db(t2, float(np.dot(interf, interf)))
        sar[k, j] = _db(float(np.sum((target + interf) ** 2)), float(np.dot(artif, artif)))
    if not compute_permutation:
        idx = np.arange(S)
        return sdr[idx, idx], sir[idx, idx], sar[idx, idx], idx
    best, best_mean = None, None
    for perm in itertools.permutations(range(S)):                        # perm[j] = estimate given to source j
        mean_sir = np.mean([sir[perm[j], j] for j in range(S)])
        if best is None or mean_sir > best_mean:
            best, best_mean = perm, mean_sir
    rows, cols = np.array(best), np.arange(S)
    return sdr[rows, cols], sir[rows, cols], sar[rows, cols], rows
