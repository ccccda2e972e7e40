"""SI-SDR (Le Roux et al. 2019), zero-mean, best speaker permutation.

The reference scores BSS-eval SDR with mir_eval (steps/evaluate_sources.py:57); SI-SDR is the
metric BASELINE.json's parity gate names, computed identically for both sides of a comparison.
"""
import itertools

import numpy as np


def si_sdr(est, ref):
    est = np.asarray(est, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    est = est - est.mean()
    ref = ref - ref.mean()
    alpha = np.dot(est, ref) / (np.dot(ref, ref) + 1e-30)
    target = alpha * ref
    noise = est - target
    return 10.0 * np.log10((np.dot(target, target) + 1e-30) / (np.dot(noise, noise) + 1e-30))


def si_sdr_best_perm(ests, refs):
    S = len(refs)
    return max(float(np.mean([si_sdr(ests[s], refs[p[s]]) for s in range(S)]))
               for p in itertools.permutations(range(S)))
