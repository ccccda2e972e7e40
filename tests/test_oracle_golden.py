"""Pins oracle/upit.py to golden vectors produced by the reference's archs/uPIT.py
(tests/golden/make_fixtures.py).  CPU only."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, fixture_samples
from oracle import upit as O


def _check_init(model, fx, prefix="wsum_"):
    for k, v in model.state_dict().items():
        if v.dtype.is_floating_point:
            ref = fx[prefix + k]
            got = np.array([float(v.double().sum()), float(v.double().abs().sum())])
            np.testing.assert_allclose(got, ref, rtol=1e-12, atol=1e-12, err_msg="init drift in " + k)


@pytest.mark.parametrize("tag", ["s2", "s3"])
def test_loss_and_grads_match_reference(tag):
    fx = np.load(os.path.join(GOLDEN, "ref_upit_loss_%s.npz" % tag))
    S = int(fx["num_spk"])
    lens = fx["lens"].tolist()
    torch.manual_seed(int(fx["seed"]))
    model = O.OracleSepDNN(num_spk=S)
    model.train()
    assert list(model.state_dict().keys()) == [
        "blstm.weight_ih_l0", "blstm.weight_hh_l0", "blstm.bias_ih_l0", "blstm.bias_hh_l0",
        "blstm.weight_ih_l0_reverse", "blstm.weight_hh_l0_reverse", "blstm.bias_ih_l0_reverse",
        "blstm.bias_hh_l0_reverse",
        "blstm.weight_ih_l1", "blstm.weight_hh_l1", "blstm.bias_ih_l1", "blstm.bias_hh_l1",
        "blstm.weight_ih_l1_reverse", "blstm.weight_hh_l1_reverse", "blstm.bias_ih_l1_reverse",
        "blstm.bias_hh_l1_reverse",
        "lin.weight", "lin.bias", "bn.weight", "bn.bias", "bn.running_mean", "bn.running_var",
        "bn.num_batches_tracked"]
    samples = fixture_samples(fx, "", len(lens), ["mix"] + ["source%d" % (s + 1) for s in range(S)])
    assert np.array_equal(O.collate_order(lens), fx["order"])
    batch = O.collate(samples)
    hidden = (torch.from_numpy(fx["h0"]), torch.from_numpy(fx["c0"]))
    loss, norm, aux = O.compute_loss(model, batch, hidden)
    loss.backward()
    # the init checksum is taken after the forward in the generator (running stats moved)
    _check_init(model, fx)
    np.testing.assert_allclose(float(norm), float(fx["norm"]), rtol=0)
    np.testing.assert_allclose(float(loss), float(fx["loss"]), rtol=1e-6)
    np.testing.assert_allclose(aux["mask_out"].detach().numpy(), fx["mask_out"], rtol=1e-5, atol=1e-6)
    for k, p in model.named_parameters():
        g = p.grad
        np.testing.assert_allclose(float(g.double().norm()), float(fx["gnorm_" + k]), rtol=1e-4)
        flat = g.flatten()
        sl = flat[:: max(1, flat.numel() // 64)][:64].numpy()
        np.testing.assert_allclose(sl, fx["gslice_" + k], rtol=1e-3, atol=1e-7)


def test_three_train_steps_match_reference():
    fx = np.load(os.path.join(GOLDEN, "ref_upit_train3.npz"))
    torch.manual_seed(int(fx["seed"]))
    model = O.OracleSepDNN()
    model.train()
    opt = torch.optim.Adam(model.parameters(), lr=0.001)
    for step in range(3):
        lens = fx["step%d_lens" % step].tolist()
        samples = fixture_samples(fx, "step%d_" % step, len(lens), ["mix", "source1", "source2"])
        batch = O.collate(samples)
        hidden = (torch.from_numpy(fx["step%d_h0" % step]), torch.from_numpy(fx["step%d_c0" % step]))
        loss, norm, gnorm, _ = O.train_step(model, opt, batch, hidden)
        np.testing.assert_allclose(loss, float(fx["step%d_loss" % step]), rtol=2e-5)
        np.testing.assert_allclose(norm, float(fx["step%d_norm" % step]), rtol=0)
        np.testing.assert_allclose(gnorm, float(fx["step%d_gnorm" % step]), rtol=1e-4)
    for k, v in model.state_dict().items():
        if v.dtype.is_floating_point:
            got = np.array([float(v.double().sum()), float(v.double().abs().sum())])
            np.testing.assert_allclose(got, fx["wsum_final_" + k], rtol=1e-5, atol=1e-4)


def test_compute_masks_match_reference():
    fx = np.load(os.path.join(GOLDEN, "ref_upit_masks.npz"))
    torch.manual_seed(int(fx["seed"]))
    model = O.OracleSepDNN()
    with torch.no_grad():
        model.bn.running_mean.copy_(torch.from_numpy(fx["running_mean"]))
        model.bn.running_var.copy_(torch.from_numpy(fx["running_var"]))
    model.eval()
    lens = fx["lens"].tolist()
    samples = fixture_samples(fx, "", len(lens), ["mix"])
    for i, d in enumerate(samples):
        d["name"] = "utt%02d.npz" % i
    batch = O.collate(samples)
    assert batch["name"] == [str(n) for n in fx["names"]]
    with torch.no_grad():
        out = O.compute_masks(model, batch, (torch.from_numpy(fx["h0"]), torch.from_numpy(fx["c0"])))
    for name, d in out.items():
        for k, v in d.items():
            ref = fx["mask_%s_%s" % (name, k)]
            assert v.shape == ref.shape and v.dtype == np.float32
            np.testing.assert_allclose(v, ref, rtol=1e-5, atol=1e-6)
