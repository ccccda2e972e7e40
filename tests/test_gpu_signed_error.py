"""The SIGNED error of the split-product arithmetic, pinned everywhere it runs.

The bf16 MFMA truncates the alignment of its addends towards minus infinity: a product formed on it by the three-way split sits a
little BELOW the exact one -- by far less than its rel-L2 error against fp64, but the same way in every element, and whatever
integrates the result (a cell state over 400 steps, the recurrence below a data gradient) adds it up.  r05 cured that in ONE
kernel (the N/N planes form); r06 gives every split kernel and the split forward recurrence sign phases (csrc/gemm.hip
SignPhase, csrc/lstm.hip SK_FWD_S3_FLIP).  Here: mean signed error and rel-L2 error against fp64 on the HOST, for

  (a) one layer of the split forward recurrence and of the fp32-MFMA one, T = 400, H = 896, B = 32, same inputs -- N(0, 1)-scaled
      and an all-positive h W_hh case that provokes the truncation;
  (b) every GEMM form the training step launches on a split kernel, at the step's own K (and K slicing / batching), on both
      split kernels, against the fp32-MFMA kernels on the same operands.

Gates: |mean signed error| of the split form <= 2 x the fp32-MFMA form's + 8 standard errors of the mean (what a sample mean of
this many elements scatters by) [GEMMs: + a tenth of the plain form's measured offset], and rel-L2 <= 1.1 x.  Measured
(profiles/r06_signed_error.txt; the same builds WITHOUT the phases: profiles/r06_signed_error_noflip.txt): the plain forms sit
5 ... 100 x outside these gates.
"""
import os
import sys

import pytest
import torch

from conftest import PKG

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(PKG, "tools"))


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need an MI355X (torch.cuda.is_available() is False)")
    from sepkern import ops as _ops
    torch.set_num_threads(max(1, min(16, len(os.sched_getaffinity(0)))))
    return _ops


@pytest.mark.parametrize("positive", [False, True])
def test_split_forward_recurrence_over_400_steps_has_the_fp32_mfma_kernels_signed_error(ops, positive):
    import signed_error as SE
    T, B, H = 400, 32, 896
    inp = SE.lstm_inputs(T, B, H, positive)
    y_ref, c_ref = ref = SE.lstm_fp64(*inp)
    if positive:          # the case is what it claims: h > 0 (3 % of the cells dip below), unsaturated
        assert float((y_ref > 0).double().mean()) > 0.9 and 0.05 < float(y_ref.mean()) < 0.6
    split = SE.lstm_case(ops, *inp, ref, ops.lstm_variant_bits(False, 1, True, False, False, 0, split3=True))
    mfma = SE.lstm_case(ops, *inp, ref, ops.lstm_variant_bits(False, 1, True, False, False, 0))
    print("T=400 %s: split y %+.2e (late %+.2e) / %.2e  c_T %+.2e / %.2e | fp32 MFMA y %+.2e (late %+.2e) / %.2e  c_T %+.2e / %.2e" % (
        "positive" if positive else "N(0,1)", split["y_mean_signed"], split["y_late_mean_signed"], split["y_rel_l2"], split["c_mean_signed"],
        split["c_rel_l2"], mfma["y_mean_signed"], mfma["y_late_mean_signed"], mfma["y_rel_l2"], mfma["c_mean_signed"], mfma["c_rel_l2"]))
    # standard error of a mean over n elements whose rms is rel_l2 x rms(reference)
    se_y = split["y_rel_l2"] * float(y_ref.pow(2).mean().sqrt()) / (y_ref.numel() ** 0.5)
    se_c = split["c_rel_l2"] * float(c_ref.pow(2).mean().sqrt()) / (c_ref.numel() ** 0.5)
    assert abs(split["y_mean_signed"]) <= 2 * abs(mfma["y_mean_signed"]) + 8 * se_y, (split, mfma, se_y)
    assert abs(split["y_late_mean_signed"]) <= 2 * abs(mfma["y_late_mean_signed"]) + 12 * se_y, (split, mfma, se_y)
    assert abs(split["c_mean_signed"]) <= 2 * abs(mfma["c_mean_signed"]) + 8 * se_c, (split, mfma, se_c)
    assert split["y_rel_l2"] <= 1.1 * mfma["y_rel_l2"] and split["c_rel_l2"] <= 1.1 * mfma["c_rel_l2"], (split, mfma)
    assert mfma["y_rel_l2"] < 1e-6                       # (both ARE fp32 recurrences: 400 steps end 1.4e-7 from the fp64 one)


def _gemm_cases():
    import signed_error as SE
    return [pytest.param(*c[1:], id=c[0].split(" (")[0].replace(" ", "_").replace(",", "")) for c in SE.GEMM_CASES]


@pytest.mark.parametrize("positive", [False, True])
@pytest.mark.parametrize("form,M,N,K,variant,splitk,batch", _gemm_cases())
def test_split_gemm_forms_have_the_fp32_mfma_kernels_signed_error(ops, form, M, N, K, variant, splitk, batch, positive):
    import signed_error as SE
    split = SE.gemm_case(ops, form, M, N, K, positive, variant, splitk, batch)
    mfma = SE.gemm_case(ops, form, M, N, K, positive, 8, splitk, batch)
    assert split["kernel"] == (10 if variant == 9 else 2) and mfma["kernel"] not in (2, 10), (split, mfma)
    # (error / sum of |terms|) scatters by about rel_l2 x (||ref|| / ||mag||) per element: for positive operands that is rel_l2
    # itself, for N(0, 1) operands 1 / sqrt(0.64 K) of it
    scale = 1.0 if positive else (0.64 * K) ** -0.5 / 0.8
    se = split["rel_l2"] * scale / ((M * N * batch) ** 0.5)
    print("%s %dx%dx%d v%d %s: split %+.2e / %.2e | fp32 MFMA %+.2e / %.2e  (8 se = %.1e)" % (
        form, M, N, K, variant, "positive" if positive else "N(0,1)", split["mean_signed"], split["rel_l2"], mfma["mean_signed"], mfma["rel_l2"], 8 * se))
    assert split["rel_l2"] <= 1.1 * mfma["rel_l2"], (split, mfma)
    # what the same kernels carried WITHOUT the phases (profiles/r06_signed_error_noflip.txt), |mean signed error|:
    plain = {(272, True): 5.3e-9, (1792, True): 3.6e-8, (7168, True): 1.47e-7, (12800, True): 2.71e-7,
             (272, False): 1.5e-9, (1792, False): 3.4e-9, (7168, False): 6.7e-9, (12800, False): 8.8e-9}[(K, positive)]
    if splitk > 1:
        plain *= 0.2 if positive else 0.5                 # (K in 5 slices of 2560: -5.4e-8 / -4.2e-9)
    if K < 768:
        # too short for sign phases (three uneven stretches would over-correct): the plain form and its known, tiny offset
        assert abs(split["mean_signed"]) <= 1.5 * plain, (split, plain)
        return
    # the fp32-MFMA kernels' level (twice their own + what a mean over this many elements scatters by), and at most a tenth of
    # the plain form's offset on top: the + - - + pattern cancels a constant and a linear trend of the sum's size exactly, the
    # curvature that is left (a random walk's sqrt(k)) shows as a few 1e-10 on N(0, 1) operands
    assert abs(split["mean_signed"]) <= 2 * abs(mfma["mean_signed"]) + 8 * se + 0.1 * plain, (split, mfma, se, plain)
