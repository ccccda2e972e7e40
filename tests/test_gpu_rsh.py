"""The RSH arch (speech-separation_amd/archs/RSH.py) on the MI355X against golden vectors
produced by the reference's own archs/RSH.py and against the CPU oracle."""
import os
import sys

import numpy as np
import pytest
import torch

from conftest import GOLDEN, ROOT
from oracle import rsh as OR
from test_oracle_rsh import fixture_hiddens, rsh_fixture_samples

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(ROOT, "speech-separation_amd", "archs"))


@pytest.fixture(scope="module")
def arch():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need an MI355X")
    import RSH
    return RSH


def test_rsh_loss_and_attention_kernels():
    from sepkern import ops
    torch.manual_seed(0)
    T, B, F, S = 11, 5, 257, 3
    lens = torch.tensor([11, 9, 9, 4, 2])
    valid = (torch.arange(T)[:, None] < lens[None, :]).float().unsqueeze(2)
    mask = torch.rand(T, B, F).requires_grad_(True)
    x = torch.cat([torch.rand(T, B, F) * valid, torch.rand(T, B, F) * valid], 2)
    srcs = [torch.rand(T, B, F) * valid for _ in range(S)]
    used = torch.zeros(S, B, dtype=torch.int32)
    used[1, 0] = 1
    used[0, 3] = 1
    masked = mask * x[:, :, :F]
    sse = torch.stack([((masked - s) ** 2).sum((0, 2)) for s in srcs])              # (S,B)
    blocked = sse.detach().clone()
    blocked[used.bool()] = float("inf")
    mins, idx = blocked.min(0)
    term = sse.gather(0, idx[None])[0].sum() / S
    term.backward()
    u = used.clone().cuda()
    res = ops.rsh_loss_fwd(mask.detach().cuda(), x.cuda(), [s.cuda() for s in srcs], lens.int().cuda(), u)
    np.testing.assert_allclose(res["sse"].cpu().numpy(), sse.detach().numpy(), rtol=2e-6)
    assert res["sel"].cpu().tolist() == idx.tolist()
    np.testing.assert_allclose(res["out"].cpu().numpy(), [float(term), float(lens.sum() * F)], rtol=2e-6)
    exp_used = used.clone()
    exp_used[idx, torch.arange(B)] = 1
    assert torch.equal(u.cpu(), exp_used)
    dm = ops.rsh_loss_bwd(mask.detach().cuda(), x.cuda(), [s.cuda() for s in srcs], res["sel"], torch.ones(1).cuda())
    np.testing.assert_allclose(dm.cpu().numpy(), mask.grad.numpy(), rtol=1e-5, atol=1e-7)
    # attention update, both activations, with its backward
    for relu in (True, False):
        xr = x.clone().requires_grad_(True)
        mr = torch.rand(T, B, F, requires_grad=True)
        ref = xr - torch.cat((torch.zeros_like(mr), mr), 2)
        if relu:
            ref = torch.relu(ref)
        dout = torch.randn(T, B, 2 * F)
        ref.backward(dout)
        out = ops.att_update(xr.detach().cuda(), mr.detach().cuda(), relu)
        np.testing.assert_array_equal(out.cpu().numpy(), ref.detach().numpy())
        dx, dmk = ops.att_update_bwd(dout.cuda(), out, F, relu)
        np.testing.assert_array_equal(dx.cpu().numpy(), xr.grad.numpy())
        np.testing.assert_array_equal(dmk.cpu().numpy(), mr.grad.numpy())


def _cuda_hiddens(fx):
    return [(h.cuda(), c.cuda()) for h, c in fixture_hiddens(fx)]


def test_rsh_compute_loss_matches_reference_golden(arch):
    fx = np.load(os.path.join(GOLDEN, "ref_rsh_loss.npz"))
    torch.manual_seed(int(fx["seed"]))
    model = arch.SepDNN(0)
    model.cuda()
    model.train()
    batch = arch.Collator("combo")(rsh_fixture_samples(fx))
    assert batch.sub_batch_lens == fx["sub_batch_lens"].tolist()
    model.next_hidden = _cuda_hiddens(fx)
    loss, norm = arch.compute_loss(model, 0, batch)
    loss.backward()
    assert float(norm) == float(fx["norm"])
    np.testing.assert_allclose(float(loss), float(fx["loss"]), rtol=2e-5)
    assert int(model.bn.num_batches_tracked) == int(fx["num_batches_tracked"])
    for k, v in model.state_dict().items():
        if v.dtype.is_floating_point:
            got = np.array([float(v.double().sum()), float(v.double().abs().sum())])
            np.testing.assert_allclose(got, fx["wsum_" + k], rtol=2e-5, atol=2e-5, err_msg=k)
    for k, p in model.named_parameters():
        np.testing.assert_allclose(float(p.grad.double().norm()), float(fx["gnorm_" + k]), rtol=3e-4, err_msg=k)
        flat = p.grad.flatten()
        sl = flat[:: max(1, flat.numel() // 64)][:64].cpu().numpy()
        np.testing.assert_allclose(sl, fx["gslice_" + k], rtol=3e-3, atol=3e-7, err_msg=k)


def test_rsh_compute_masks_matches_reference_golden(arch, tmp_path):
    fx = np.load(os.path.join(GOLDEN, "ref_rsh_masks.npz"))
    torch.manual_seed(int(fx["seed"]))
    model = arch.SepDNN(0)
    sd = model.state_dict()
    sd["bn.running_mean"] = torch.from_numpy(fx["running_mean"])
    sd["bn.running_var"] = torch.from_numpy(fx["running_var"])
    model.cuda()
    model.load_state_dict(sd)
    model.eval()
    batch = arch.Collator("combo")(rsh_fixture_samples(fx, test=True))
    model.next_hidden = _cuda_hiddens(fx)
    arch.compute_masks(model, batch, str(tmp_path))
    for i, (T, n) in enumerate(fx["spec"].tolist()):
        z = np.load(os.path.join(str(tmp_path), "utt%02d.npz" % i))
        assert z.files == ["s%d" % (k + 1) for k in range(n)]
        for k in z.files:
            ref = fx["mask_utt%02d.npz_%s" % (i, k)]
            assert z[k].shape == ref.shape == (257, T)
            assert np.abs(z[k] - ref).max() <= 1e-4 * np.abs(ref).max()


def test_rsh_small_config_matches_oracle(arch):
    """4-speaker sub-batch (CHiME-5-shaped, BASELINE config 5) with a reduced model, against the oracle."""
    torch.manual_seed(8)
    rng = np.random.default_rng(8)
    H, L, F = 64, 2, 257
    model = arch.SepDNN(0, hidden_dim=str(H), num_layers=str(L))
    model.cuda()
    model.train()
    orc = OR.OracleRSH(F, H, L)
    orc.load_state_dict({k: v.cpu() for k, v in model.state_dict().items()})
    orc.train()
    spec = [(10, 4), (8, 4), (9, 2), (6, 4)]
    samples = []
    for T, n in spec:
        mix = np.abs(rng.standard_normal((T, F))).astype(np.float32)
        d = {"combo": np.concatenate((mix, np.ones(mix.shape)), axis=1).astype(np.float32)}
        for s in range(n):
            d["source%d" % (s + 1)] = (np.abs(rng.standard_normal((T, F))) * 0.5).astype(np.float32)
        samples.append(d)
    hid = [(torch.randn(2 * L, 1, H), torch.randn(2 * L, 1, H)), (torch.randn(2 * L, 3, H), torch.randn(2 * L, 3, H))]
    lo, no, _ = OR.compute_loss(orc, OR.collate(samples), hid)
    lo.backward()
    model.next_hidden = [(h.cuda(), c.cuda()) for h, c in hid]
    loss, norm = arch.compute_loss(model, 0, arch.Collator("combo")(samples))
    loss.backward()
    np.testing.assert_allclose(float(loss), float(lo), rtol=2e-5)
    assert float(norm) == float(no)
    og = dict(orc.named_parameters())
    for k, p in model.named_parameters():
        ref = og[k].grad
        err = float((p.grad.cpu().double() - ref.double()).norm() / (ref.double().norm() + 1e-30))
        assert err < 3e-4, (k, err)
