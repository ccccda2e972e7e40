"""Pins oracle/rsh.py to golden vectors produced by the reference's archs/RSH.py (CPU only)."""
import os

import numpy as np
import torch

from conftest import GOLDEN
from oracle import rsh as R


def rsh_fixture_samples(fx, test=False):
    out = []
    for i, (T, n) in enumerate(fx["spec"].tolist()):
        d = {"combo": fx["sample%d_combo" % i]}
        if test:
            d["name"] = "utt%02d.npz" % i
            d["num_spk"] = n
        else:
            for s in range(n):
                d["source%d" % (s + 1)] = fx["sample%d_source%d" % (i, s + 1)]
        out.append(d)
    return out


def fixture_hiddens(fx):
    out, j = [], 0
    while "h0_%d" % j in fx:
        out.append((torch.from_numpy(fx["h0_%d" % j]), torch.from_numpy(fx["c0_%d" % j])))
        j += 1
    return out


def test_rsh_loss_and_grads_match_reference():
    fx = np.load(os.path.join(GOLDEN, "ref_rsh_loss.npz"))
    torch.manual_seed(int(fx["seed"]))
    model = R.OracleRSH()
    model.train()
    assert list(model.state_dict().keys())[-7:] == ["lin.weight", "lin.bias", "bn.weight", "bn.bias", "bn.running_mean",
                                                    "bn.running_var", "bn.num_batches_tracked"]
    batch = R.collate(rsh_fixture_samples(fx))
    assert batch.sub_batch_lens == fx["sub_batch_lens"].tolist() == [0, 0, 3, 2]
    loss, norm, aux = R.compute_loss(model, batch, fixture_hiddens(fx))
    loss.backward()
    np.testing.assert_allclose(float(norm), float(fx["norm"]), rtol=0)
    np.testing.assert_allclose(float(loss.detach()), float(fx["loss"]), rtol=1e-6)
    assert int(model.bn.num_batches_tracked) == int(fx["num_batches_tracked"]) == 5     # one BN call per pass
    for k, v in model.state_dict().items():
        if v.dtype.is_floating_point:
            got = np.array([float(v.double().sum()), float(v.double().abs().sum())])
            np.testing.assert_allclose(got, fx["wsum_" + k], rtol=1e-6, atol=1e-6, err_msg=k)
    for k, p in model.named_parameters():
        np.testing.assert_allclose(float(p.grad.double().norm()), float(fx["gnorm_" + k]), rtol=1e-4, err_msg=k)
        flat = p.grad.flatten()
        np.testing.assert_allclose(flat[:: max(1, flat.numel() // 64)][:64].numpy(), fx["gslice_" + k], rtol=1e-3, atol=1e-8)


def test_rsh_masks_match_reference():
    fx = np.load(os.path.join(GOLDEN, "ref_rsh_masks.npz"))
    torch.manual_seed(int(fx["seed"]))
    model = R.OracleRSH()
    with torch.no_grad():
        model.bn.running_mean.copy_(torch.from_numpy(fx["running_mean"]))
        model.bn.running_var.copy_(torch.from_numpy(fx["running_var"]))
    model.eval()
    batch = R.collate(rsh_fixture_samples(fx, test=True))
    with torch.no_grad():
        out = R.compute_masks(model, batch, fixture_hiddens(fx))
    assert len(out) == 3
    for name, d in out.items():
        for k, v in d.items():
            ref = fx["mask_%s_%s" % (name, k)]
            assert v.shape == ref.shape
            np.testing.assert_allclose(v, ref, rtol=1e-5, atol=1e-6)
