"""BSS Eval SDR / SIR / SAR (sepkern/bsseval.py; what the reference scores with through mir_eval,
steps/evaluate_sources.py:57): closed-form cases and an independent dense least-squares check.  CPU only."""
import numpy as np
import pytest

from sepkern import bsseval


def _dense_projection(refs, e, taps):
    """P e onto the delayed references by an explicit design matrix and numpy's least squares."""
    n = refs.shape[1]
    cols = []
    for r in refs:
        for t in range(taps):
            c = np.zeros(n + taps - 1)
            c[t:t + n] = r
            cols.append(c)
    A = np.stack(cols, axis=1)
    y = np.concatenate((e, np.zeros(taps - 1)))
    coef = np.linalg.lstsq(A, y, rcond=None)[0]
    return A @ coef


def test_projections_equal_dense_least_squares():
    rng = np.random.default_rng(0)
    refs = rng.standard_normal((2, 300))
    e = 0.7 * refs[0] + 0.2 * np.roll(refs[1], 2) + 0.1 * rng.standard_normal(300)
    est = np.stack([e, refs[1] + 0.05 * rng.standard_normal(300)])
    taps = 8
    sdr, sir, sar, perm = bsseval.bss_eval_sources(refs, est, taps=taps)
    assert perm.tolist() == [0, 1]
    p_all = _dense_projection(refs, e, taps)
    p_one = _dense_projection(refs[:1], e, taps)
    pad = np.concatenate((e, np.zeros(taps - 1)))
    want_sdr = 10 * np.log10(np.sum(p_one ** 2) / np.sum((pad - p_one) ** 2))
    want_sir = 10 * np.log10(np.sum(p_one ** 2) / np.sum((p_all - p_one) ** 2))
    want_sar = 10 * np.log10(np.sum(p_all ** 2) / np.sum((pad - p_all) ** 2))
    np.testing.assert_allclose([sdr[0], sir[0], sar[0]], [want_sdr, want_sir, want_sar], rtol=1e-7)


def test_scaled_and_filtered_copies_are_allowed_distortions():
    rng = np.random.default_rng(1)
    refs = rng.standard_normal((2, 4000))
    refs[1, -40:] = 0                                            # room for the filter's tail inside the signal
    h = rng.standard_normal(40)                                  # any filter shorter than the 512 taps
    est = np.stack([0.3 * refs[0], np.convolve(refs[1], h)[:4000]])
    sdr, sir, sar, perm = bsseval.bss_eval_sources(refs, est)
    assert perm.tolist() == [0, 1]
    assert np.all(sdr > 100) and np.all(sir > 100) and np.all(sar > 100)   # "+inf" in exact arithmetic


def test_orthogonal_interference_gives_the_energy_ratio():
    rng = np.random.default_rng(2)
    a = rng.standard_normal(2000)
    b = rng.standard_normal(2000)
    b -= a * np.dot(a, b) / np.dot(a, a)                         # exactly orthogonal at lag 0
    refs = np.stack([a, b])
    est = np.stack([a + 0.1 * b, b + 0.5 * a])
    sdr, sir, sar, perm = bsseval.bss_eval_sources(refs, est, taps=1)     # taps = 1: plain projections onto a, b
    want0 = 10 * np.log10(np.dot(a, a) / (0.01 * np.dot(b, b)))
    want1 = 10 * np.log10(np.dot(b, b) / (0.25 * np.dot(a, a)))
    np.testing.assert_allclose(sir, [want0, want1], rtol=1e-9)
    np.testing.assert_allclose(sdr, sir, rtol=1e-6)              # nothing unexplained: SDR = SIR, SAR = +inf (or huge)
    assert np.all(sar > 150)
    # with the real 512-tap filter the delayed copies explain a little more of a finite random signal: close, not equal
    a, b = rng.standard_normal(40000), rng.standard_normal(40000)
    b -= a * np.dot(a, b) / np.dot(a, a)
    _, sir512, _, _ = bsseval.bss_eval_sources(np.stack([a, b]), np.stack([a + 0.1 * b, b + 0.5 * a]))
    want = [10 * np.log10(np.dot(a, a) / (0.01 * np.dot(b, b))), 10 * np.log10(np.dot(b, b) / (0.25 * np.dot(a, a)))]
    np.testing.assert_allclose(sir512, want, atol=1.0)


def test_artifacts_and_permutation_search():
    rng = np.random.default_rng(3)
    refs = rng.standard_normal((3, 3000))
    noise = rng.standard_normal((3, 3000))
    clean = refs + 0.1 * noise
    order = [2, 0, 1]                                            # estimate k is source order[k]
    est = clean[order]
    sdr, sir, sar, perm = bsseval.bss_eval_sources(refs, est, taps=16)
    assert perm.tolist() == [1, 2, 0]                            # estimate perm[j] belongs to source j
    s2, i2, a2, p2 = bsseval.bss_eval_sources(refs, clean, taps=16)
    assert p2.tolist() == [0, 1, 2]
    np.testing.assert_allclose(sdr, s2, rtol=1e-9)
    np.testing.assert_allclose(sar, a2, rtol=1e-9)
    assert np.all(np.abs(sar - 20.0) < 1.0)                      # |r|^2 / |0.1 n|^2 = 20 dB, minus what the span absorbs
    same, _, _, fixed = bsseval.bss_eval_sources(refs, est, compute_permutation=False, taps=16)
    assert fixed.tolist() == [0, 1, 2] and np.all(same < 0)      # wrong pairing, no search: negative SDR


def test_rejects_silent_inputs():
    x = np.ones((2, 100))
    with pytest.raises(ValueError):
        bsseval.bss_eval_sources(np.stack([x[0], np.zeros(100)]), x)
    with pytest.raises(ValueError):
        bsseval.bss_eval_sources(x, x[:, :50])
