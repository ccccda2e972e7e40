"""The drop-in arch module (speech-separation_amd/archs/uPIT.py) on the MI355X against
(1) golden vectors produced by the reference's own archs/uPIT.py and (2) the CPU oracle.

Tolerances (fp32 end to end; differences are summation order and libm only):
  masks 1e-4 relative (BASELINE.json), loss 1e-5 relative, same arg-min permutation,
  SI-SDR of reconstructed waveforms within +-0.1 dB.
"""
import os
import sys

import numpy as np
import pytest
import torch

from conftest import GOLDEN, PKG, ROOT, fixture_samples
from oracle import stft as OS
from oracle import upit as OU

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.join(ROOT, "speech-separation_amd", "archs"))


@pytest.fixture(scope="module")
def arch():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need an MI355X")
    import uPIT
    return uPIT


def _hidden(fx, prefix=""):
    return (torch.from_numpy(fx[prefix + "h0"]).cuda(), torch.from_numpy(fx[prefix + "c0"]).cuda())


def _check_init(model, fx, prefix="wsum_"):
    for k, v in model.state_dict().items():
        if v.dtype.is_floating_point:
            got = np.array([float(v.double().sum()), float(v.double().abs().sum())])
            np.testing.assert_allclose(got, fx[prefix + k], rtol=1e-6, atol=1e-6, err_msg="init drift in " + k)


@pytest.mark.parametrize("lens", [[23, 17, 17, 9, 4], [12, 12, 12]])
def test_reference_style_loss_through_model_call_backprops(arch, lens):
    """The reference's own step is `mask_out = model(mix)` -> padded PIT-MSE in torch -> loss.backward()
    (archs/uPIT.py:175-197).  SepDNN.forward therefore has to be DIFFERENTIABLE on variable-length batches too (ADVICE r04:
    Packing.unpack alone is a raw kernel launch): the loss written with the reference's torch expressions on model(mix)
    gives the loss and the parameter gradients of compute_loss (fused PIT kernels on packed rows) on the same batch."""
    import itertools
    from torch.nn.utils.rnn import pad_packed_sequence
    S, F, H, L = 2, 257, 64, 2
    torch.manual_seed(3)
    model = arch.SepDNN(0, num_spk=str(S), hidden_dim=str(H), num_layers=str(L))
    model.cuda()
    model.train()
    rng = np.random.default_rng(5)
    samples = []
    for n in lens:
        d = {"mix": np.abs(rng.standard_normal((n, F))).astype(np.float32)}
        for s_ in range(S):
            d["source%d" % (s_ + 1)] = np.abs(rng.standard_normal((n, F))).astype(np.float32) * 0.7
        samples.append(d)
    batch = arch.Collator("mix")(samples)
    B = len(lens)
    hid = (torch.randn(2 * L, B, H).cuda(), torch.randn(2 * L, B, H).cuda())
    model.next_hidden = hid
    loss, norm = arch.compute_loss(model, 0, batch)
    loss.backward()
    want = model.flat_parameters()[1].clone()
    # ---- the reference's expressions
    mix = batch["mix"].cuda()
    sources = [pad_packed_sequence(batch["source%d" % (i + 1)].cuda(), batch_first=True)[0] for i in range(S)]
    model.zero_grad()
    model.next_hidden = hid
    model.hidden = model.init_hidden(B)
    mask_out = model(mix)
    assert mask_out.requires_grad and tuple(mask_out.shape) == (B, max(lens), F * S) and mask_out.is_contiguous()
    mixes, ln = pad_packed_sequence(mix, batch_first=True)
    masked = mask_out * torch.cat([mixes for _ in range(S)], dim=2)
    perms = list(itertools.permutations(range(S)))
    losses = torch.stack([torch.sum(((masked - torch.cat([sources[i] for i in perm], dim=2)) ** 2).view(B, -1), dim=1)
                          for perm in perms])
    min_losses, _ = torch.min(losses, 0)
    norm2 = torch.sum(ln.float().cuda()) * F
    loss2 = torch.sum(min_losses) / S / norm2
    loss2.backward()
    got = model.flat_parameters()[1]
    assert float(norm2) == float(norm)
    np.testing.assert_allclose(float(loss2), float(loss), rtol=1e-5)
    rel = float((got - want).norm() / want.norm())
    assert rel <= 2e-5, rel
    assert model._pending == 0


@pytest.mark.parametrize("tag", ["s2", "s3"])
def test_compute_loss_matches_reference_golden(arch, tag):
    fx = np.load(os.path.join(GOLDEN, "ref_upit_loss_%s.npz" % tag))
    S, lens = int(fx["num_spk"]), fx["lens"].tolist()
    torch.manual_seed(int(fx["seed"]))
    model = arch.SepDNN(0, num_spk=str(S))
    model.cuda()
    model.train()
    assert list(model.state_dict().keys())[:4] == ["blstm.weight_ih_l0", "blstm.weight_hh_l0", "blstm.bias_ih_l0",
                                                   "blstm.bias_hh_l0"]
    assert list(model.state_dict().keys())[-7:] == ["lin.weight", "lin.bias", "bn.weight", "bn.bias",
                                                    "bn.running_mean", "bn.running_var", "bn.num_batches_tracked"]
    samples = fixture_samples(fx, "", len(lens), ["mix"] + ["source%d" % (s + 1) for s in range(S)])
    batch = arch.Collator("mix")(samples)
    model.next_hidden = _hidden(fx)
    loss, norm = arch.compute_loss(model, 0, batch)
    loss.backward()
    _check_init(model, fx)                              # weights equal the reference's, running stats moved once
    assert float(norm) == float(fx["norm"])
    np.testing.assert_allclose(float(loss), float(fx["loss"]), rtol=1e-5)
    for k, p in model.named_parameters():
        g = p.grad
        np.testing.assert_allclose(float(g.double().norm()), float(fx["gnorm_" + k]), rtol=2e-4, err_msg=k)
        flat = g.flatten()
        sl = flat[:: max(1, flat.numel() // 64)][:64].cpu().numpy()
        np.testing.assert_allclose(sl, fx["gslice_" + k], rtol=2e-3, atol=2e-7, err_msg=k)
    # masks: same (h0,c0), train-mode BN (batch statistics), no grad
    model.next_hidden = _hidden(fx)
    model.hidden = model.init_hidden(len(lens))
    with torch.no_grad():
        mask = model(batch["mix"])
    ref = fx["mask_out"]
    assert tuple(mask.shape) == ref.shape
    got = mask.cpu().numpy()
    assert np.abs(got - ref).max() <= 1e-4 * np.abs(ref).max()
    np.testing.assert_allclose(((got - ref) ** 2).mean() / (ref ** 2).mean(), 0, atol=1e-8)


@pytest.mark.parametrize("fused", [False, True])
def test_three_train_steps_match_reference_golden(arch, fused):
    from sepkern.optim import ClipAdam
    fx = np.load(os.path.join(GOLDEN, "ref_upit_train3.npz"))
    torch.manual_seed(int(fx["seed"]))
    model = arch.SepDNN(0)
    model.cuda()
    model.train()
    opt = ClipAdam(model, lr=0.001, max_norm=0.25) if fused else torch.optim.Adam(model.parameters(), lr=0.001)
    for step in range(3):
        lens = fx["step%d_lens" % step].tolist()
        samples = fixture_samples(fx, "step%d_" % step, len(lens), ["mix", "source1", "source2"])
        batch = arch.Collator("mix")(samples)
        model.next_hidden = _hidden(fx, "step%d_" % step)
        loss, norm = arch.compute_loss(model, 0, batch)      # steps/train_qsub.py:117-122
        loss.backward()
        if fused:
            gnorm = float(opt.step()[0])
        else:
            gnorm = float(torch.nn.utils.clip_grad_norm_(model.parameters(), 0.25))
            opt.step()
        np.testing.assert_allclose(float(loss), float(fx["step%d_loss" % step]), rtol=5e-5)
        assert float(norm) == float(fx["step%d_norm" % step])
        np.testing.assert_allclose(gnorm, float(fx["step%d_gnorm" % step]), rtol=2e-4)
    for k, v in model.state_dict().items():
        if v.dtype.is_floating_point:
            got = np.array([float(v.double().sum()), float(v.double().abs().sum())])
            np.testing.assert_allclose(got, fx["wsum_final_" + k], rtol=2e-5, atol=2e-4, err_msg=k)
    assert int(model.bn.num_batches_tracked) == 3


def test_compute_masks_matches_reference_golden(arch, tmp_path):
    fx = np.load(os.path.join(GOLDEN, "ref_upit_masks.npz"))
    torch.manual_seed(int(fx["seed"]))
    model = arch.SepDNN(0)
    sd = model.state_dict()
    sd["bn.running_mean"] = torch.from_numpy(fx["running_mean"])
    sd["bn.running_var"] = torch.from_numpy(fx["running_var"])
    model.cuda()
    model.load_state_dict(sd)                               # exercises the state_dict round trip
    model.eval()
    lens = fx["lens"].tolist()
    samples = fixture_samples(fx, "", len(lens), ["mix"])
    for i, d in enumerate(samples):
        d["name"] = "utt%02d.npz" % i
    batch = arch.Collator("mix")(samples)
    model.next_hidden = _hidden(fx)
    with torch.no_grad():
        arch.compute_masks(model, batch, str(tmp_path))
    for i, n in enumerate(lens):
        z = np.load(os.path.join(str(tmp_path), "utt%02d.npz" % i))
        assert z.files == ["s1", "s2"]
        for k in z.files:
            ref = fx["mask_utt%02d.npz_%s" % (i, k)]
            assert z[k].shape == ref.shape == (257, n) and z[k].dtype == np.float32
            assert np.abs(z[k] - ref).max() <= 1e-4 * np.abs(ref).max()


def _oracle_like(model, H, L, S):
    """An OracleSepDNN carrying the same weights as the GPU model."""
    o = OU.OracleSepDNN(num_spk=S, hidden_dim=H, num_layers=L)
    o.load_state_dict({k: v.cpu() for k, v in model.state_dict().items()})
    return o


@pytest.mark.parametrize("H,L,S,B,T", [(300, 2, 2, 8, 40), (896, 3, 2, 32, 24), (600, 2, 3, 5, 30), (20, 1, 1, 3, 10),
                                       (256, 1, 4, 2, 6)])
def test_configs_match_oracle(arch, H, L, S, B, T):
    """BASELINE configs 1 (2x300), 2 (3x896, B=32) and 4 (3 speakers) against the CPU oracle, ragged lengths."""
    torch.manual_seed(H + L)
    rng = np.random.default_rng(H)
    model = arch.SepDNN(0, num_spk=str(S), hidden_dim=str(H), num_layers=str(L))
    model.cuda()
    model.train()
    orc = _oracle_like(model, H, L, S)
    orc.train()
    lens = sorted([int(v) for v in rng.integers(max(2, T // 2), T + 1, B)])
    lens[-1] = T
    samples = []
    for n in lens:
        d = {"mix": np.abs(rng.standard_normal((n, 257))).astype(np.float32)}
        for s in range(S):
            d["source%d" % (s + 1)] = np.abs(rng.standard_normal((n, 257))).astype(np.float32) * 0.6
        samples.append(d)
    batch = arch.Collator("mix")(samples)
    h0, c0 = torch.randn(2 * L, B, H), torch.randn(2 * L, B, H)
    lo, no, aux = OU.compute_loss(orc, OU.collate(samples), (h0, c0))
    lo.backward()
    model.next_hidden = (h0.cuda(), c0.cuda())
    loss, norm = arch.compute_loss(model, 0, batch)
    loss.backward()
    np.testing.assert_allclose(float(loss), float(lo), rtol=1e-5)
    assert float(norm) == float(no)
    og = dict(orc.named_parameters())
    for k, p in model.named_parameters():
        ref = og[k].grad
        err = float((p.grad.cpu().double() - ref.double()).norm() / (ref.double().norm() + 1e-30))
        assert err < 2e-4, (k, err)
    model.next_hidden = (h0.cuda(), c0.cuda())
    model.hidden = model.init_hidden(B)
    with torch.no_grad():
        mask = model(batch["mix"]).cpu().numpy()
    ref = aux["mask_out"].detach().numpy()
    assert np.abs(mask - ref).max() <= 1e-4 * np.abs(ref).max()


@pytest.mark.parametrize("H,L,S,B,T", [(600, 2, 3, 5, 30), (896, 3, 3, 4, 12)])
def test_bf16_config_matches_bf16_oracle_and_fp32_within_tolerance(arch, H, L, S, B, T):
    """BASELINE configs[3] (3 speakers, bf16 matrix-core inputs): the HIP path against a CPU computation of the
    SAME arithmetic (oracle/upit_bf16.py: operands of every non-recurrent product rounded to bf16, fp32
    accumulate) at a tight tolerance, and against the fp32 oracle at the stated bf16 tolerance (2e-2 abs on
    masks)."""
    from oracle import upit_bf16 as OB
    torch.manual_seed(H + L + 1)
    rng = np.random.default_rng(H + 1)
    model = arch.SepDNN(0, num_spk=str(S), hidden_dim=str(H), num_layers=str(L), dtype="bf16")
    model.cuda()
    model.train()
    orc = _oracle_like(model, H, L, S)
    orc.train()
    lens = sorted([int(v) for v in rng.integers(max(2, T // 2), T + 1, B)])
    lens[-1] = T
    samples = []
    for n in lens:
        d = {"mix": np.abs(rng.standard_normal((n, 257))).astype(np.float32)}
        for s in range(S):
            d["source%d" % (s + 1)] = np.abs(rng.standard_normal((n, 257))).astype(np.float32) * 0.6
        samples.append(d)
    batch = arch.Collator("mix")(samples)
    h0, c0 = torch.randn(2 * L, B, H), torch.randn(2 * L, B, H)
    lo, no, aux = OB.compute_loss(orc, OU.collate(samples), (h0, c0))
    lo.backward()
    model.next_hidden = (h0.cuda(), c0.cuda())
    loss, norm = arch.compute_loss(model, 0, batch)
    loss.backward()
    np.testing.assert_allclose(float(loss), float(lo), rtol=2e-4)
    assert float(norm) == float(no)
    og = dict(orc.named_parameters())
    for k, p in model.named_parameters():
        ref = og[k].grad
        err = float((p.grad.cpu().double() - ref.double()).norm() / (ref.double().norm() + 1e-30))
        # an fp32 rounding-order difference (or the kernels' v_exp/v_rcp cell math) can flip a bf16 rounding of an
        # operand (2^-9 relative on that term); through 3 layers and T steps the flips reach ~3e-3 of a gradient's norm
        assert err < 6e-3, (k, err)
    model.next_hidden = (h0.cuda(), c0.cuda())
    model.hidden = model.init_hidden(B)
    with torch.no_grad():
        mask = model(batch["mix"]).cpu().numpy()
    ref = aux["mask_out"].detach().numpy()
    assert np.abs(mask - ref).max() <= 2e-3
    # against the fp32 oracle: the bf16 tolerance
    orc32 = _oracle_like(model, H, L, S)
    orc32.train()
    l32, _, a32 = OU.compute_loss(orc32, OU.collate(samples), (h0, c0))
    assert np.abs(mask - a32["mask_out"].detach().numpy()).max() <= 2e-2
    assert abs(float(loss) - float(l32)) <= 1e-2 * abs(float(l32))


def test_end_to_end_si_sdr_parity(arch):
    """wav -> STFT -> masks -> mask-apply + iSTFT -> int16 wav, GPU path vs oracle path, SI-SDR +-0.1 dB."""
    from sepkern import ops, synth
    from sepkern.sisdr import si_sdr_best_perm
    torch.manual_seed(3)
    H, L, S = 300, 2, 2
    model = arch.SepDNN(0, hidden_dim=str(H), num_layers=str(L))
    model.cuda()
    model.eval()
    orc = _oracle_like(model, H, L, S)
    orc.eval()
    pcms = synth.pcm_batch(3, lengths=[24000, 20000, 16000])
    # GPU path
    specs = ops.stft_batch([torch.from_numpy(p[0]).cuda() for p in pcms], want_complex=True, layout="FT")
    samples = [{"mix": np.abs(s.cpu().numpy()).T, "name": "u%d.npz" % i} for i, s in enumerate(specs)]
    batch = arch.Collator("mix")(samples)
    order = OU.collate_order([len(d["mix"]) for d in samples])
    B = len(samples)
    h0, c0 = torch.randn(2 * L, B, H), torch.randn(2 * L, B, H)
    model.next_hidden = (h0.cuda(), c0.cuda())
    model.hidden = model.init_hidden(B)
    with torch.no_grad():
        mask = model(batch["mix"])                                   # (B, T, S*F) in collated order
    # oracle path
    ospecs = [OS.stft(OS.pcm16_to_float(p[0])) for p in pcms]
    osamples = [{"mix": np.abs(s).T, "name": "u%d.npz" % i} for i, s in enumerate(ospecs)]
    with torch.no_grad():
        omasks = OU.compute_masks(orc, OU.collate(osamples), (h0, c0))
    for pos, u in enumerate(order):
        T = specs[u].shape[1]
        gm = [mask[pos, :T, s * 257:(s + 1) * 257].t().contiguous() for s in range(S)]
        wav, pcm = ops.mask_istft([specs[u]], [gm])
        refs = [OS.pcm16_to_float(p)[:128 * (T - 1)] for p in pcms[u][1:]]
        est_g = [pcm[0][s].cpu().numpy().astype(np.float64) for s in range(S)]
        est_o = [OS.reconstruct(ospecs[u], omasks["u%d.npz" % u]["s%d" % (s + 1)])[1].astype(np.float64) for s in range(S)]
        sg, so = si_sdr_best_perm(est_g, refs), si_sdr_best_perm(est_o, refs)
        assert abs(sg - so) <= 0.1, (sg, so)
        for s in range(S):
            d = np.abs(est_g[s] - est_o[s])
            assert d.max() <= 2.0                                    # int16 LSBs


def test_coscheduled_backward_is_bitwise_the_serial_backward(arch):
    """The engine issues the weight-gradient GEMMs of layer l on a side stream so that they run co-resident with
    layer l-1's recurrence.  That is scheduling only: gradients must be bit-identical to the
    serial order, run after run (a missing dependency would show up as a difference here)."""
    H, L, S, B, T = 896, 3, 2, 32, 40
    torch.manual_seed(11)
    rng = np.random.default_rng(11)
    model = arch.SepDNN(0, num_spk=str(S), hidden_dim=str(H), num_layers=str(L))
    model.cuda()
    model.train()
    lens = sorted([int(v) for v in rng.integers(T // 2, T + 1, B)])
    lens[-1] = T
    samples = []
    for n in lens:
        d = {"mix": np.abs(rng.standard_normal((n, 257))).astype(np.float32)}
        for s in range(S):
            d["source%d" % (s + 1)] = np.abs(rng.standard_normal((n, 257))).astype(np.float32) * 0.6
        samples.append(d)
    batch = arch.Collator("mix")(samples)
    h0, c0 = torch.randn(2 * L, B, H).cuda(), torch.randn(2 * L, B, H).cuda()

    def grads(overlap):
        model._engine.overlap = overlap
        model.next_hidden = (h0, c0)
        loss, _ = arch.compute_loss(model, 0, batch)
        loss.backward()
        torch.cuda.synchronize()
        return model._engine.grad.clone(), float(loss)

    model.next_hidden = (h0, c0)
    arch.compute_loss(model, 0, batch)[0].backward()          # builds the engine
    assert model._engine.overlap, "co-scheduling is expected to be on by default"
    model._engine.var_side = model._engine.var_main            # same GEMM kernel on either stream (the default picks the
                                                               # register-staged one beside a recurrence: other K order)
    g_ser, l_ser = grads(False)
    for _ in range(3):
        g_co, l_co = grads(True)
        assert l_co == l_ser
        assert torch.equal(g_co, g_ser)


def test_data_parallel_sync_bn_equals_global_batch(tmp_path):
    """with the optional BatchNorm exchange (sync_bn=1: one all-gather of per-rank statistics forward, one
    all-reduce of the two per-channel sums backward) the data-parallel update equals the single-device update on the
    global batch.  2 ranks on cuda:0 over gloo (LSTM per-step mode: two processes cannot both keep a persistent grid
    resident on one GPU) against one process holding all 8 utterances."""
    import socket
    import subprocess
    import sys as _sys
    tests_dir = os.path.dirname(os.path.abspath(__file__))
    worker = os.path.join(tests_dir, "_dp_syncbn_worker.py")
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    env = dict(os.environ, SEPKERN_DIST_BACKEND="gloo", SEPKERN_LSTM_MODE="2")
    dp = str(tmp_path / "dp.npz")
    r = subprocess.run([_sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), worker, dp, "1"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    single = str(tmp_path / "single.npz")
    env1 = {k: v for k, v in env.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([_sys.executable, worker, single, "0"], env=env1, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    a, b = np.load(dp), np.load(single)
    assert float(a["norm"]) == float(b["norm"])
    np.testing.assert_allclose(float(a["loss"]), float(b["loss"]), rtol=2e-6)
    np.testing.assert_allclose(a["running_mean"], b["running_mean"], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(a["running_var"], b["running_var"], rtol=1e-5, atol=1e-7)
    err = np.linalg.norm(a["grad"].astype(np.float64) - b["grad"]) / np.linalg.norm(b["grad"])
    assert err < 2e-5, err
    # and without the exchange the two differ (per-rank statistics): the option is doing something
    nosync = str(tmp_path / "nosync.npz")
    r = subprocess.run([_sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), worker, nosync, "0"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    c = np.load(nosync)
    assert np.linalg.norm(c["grad"].astype(np.float64) - b["grad"]) / np.linalg.norm(b["grad"]) > 1e-3


def test_timed_out_recurrence_never_reaches_the_weights(arch):
    """A persistent launch whose bounded wait gave up leaves garbage gradients.  Its sticky status word travels
    behind the flat gradient to the fused clip+Adam, which skips the step on the device (no host sync per step);
    the host learns of it from ClipAdam.check() / model.check_status()."""
    from sepkern import ops
    from sepkern._lib import SepkernError
    from sepkern.optim import ClipAdam
    torch.manual_seed(5)
    H, L, B, T = 64, 2, 4, 9
    model = arch.SepDNN(0, hidden_dim=str(H), num_layers=str(L))
    model.cuda()
    model.train()
    opt = ClipAdam(model, lr=1e-3, max_norm=0.25)
    rng = np.random.default_rng(0)
    samples = [{"mix": np.abs(rng.standard_normal((T, 257))).astype(np.float32),
                "source1": np.abs(rng.standard_normal((T, 257))).astype(np.float32),
                "source2": np.abs(rng.standard_normal((T, 257))).astype(np.float32)} for _ in range(B)]
    batch = arch.Collator("mix")(samples)

    def one_step():
        loss, _ = arch.compute_loss(model, 0, batch)
        loss.backward()
        return opt.step()
    one_step()
    assert opt.skipped() == 0
    before = model.flat_parameters()[0].clone()
    m_before = opt.m.clone()
    rm_before, rv_before = model.bn.running_mean.clone(), model.bn.running_var.clone()
    assert float(rm_before.abs().sum()) > 0           # (the clean step did update them)
    ws = ops.lstm_ws(T, B, H)
    ops.lstm_sticky(ws).fill_(1)                      # what an aborting workgroup does (csrc/lstm.hip, wait_flags)
    scal = one_step()
    assert float(scal[2]) == 1.0 and opt.skipped() == 1
    assert torch.equal(model.flat_parameters()[0], before) and torch.equal(opt.m, m_before)
    # ... nor the BatchNorm running statistics a checkpoint would hold (the forward of such a step is garbage too)
    assert torch.equal(model.bn.running_mean, rm_before) and torch.equal(model.bn.running_var, rv_before)
    with pytest.raises(SepkernError):
        opt.check()
    with pytest.raises(SepkernError):
        model.check_status()                          # reports once and clears the word
    model.check_status()
    one_step()
    assert opt.skipped() == 1 and not torch.equal(model.flat_parameters()[0], before)
    assert opt.state_dict()["step"] == opt.step_count - 1     # updates actually applied (the bias corrections' count)


def test_training_driver_recovers_in_process_from_a_timed_out_launch(arch, capsys):
    """steps/train_qsub.py::train_epoch watches the fused optimizer's skipped-step counter (its asynchronous host copy
    after every step, a synchronous read every POLL_EVERY steps): after a timed-out persistent launch (simulated: the sticky
    word set) it clears the word, drops the loss terms of that window, switches the engine to one launch per step IN THIS
    PROCESS and goes on training -- instead of skipping every later step of the epoch and dying at its end."""
    import importlib
    from sepkern import ops
    from sepkern.optim import ClipAdam
    sys.path.insert(0, os.path.join(PKG, "steps"))
    tq = importlib.import_module("train_qsub")
    torch.manual_seed(6)
    H, L, B, T = 64, 2, 4, 9
    model = arch.SepDNN(0, hidden_dim=str(H), num_layers=str(L))
    model.cuda()
    model.train()
    opt = ClipAdam(model, lr=1e-3, max_norm=0.25)
    rng = np.random.default_rng(1)
    samples = [{"mix": np.abs(rng.standard_normal((T, 257))).astype(np.float32),
                "source1": np.abs(rng.standard_normal((T, 257))).astype(np.float32),
                "source2": np.abs(rng.standard_normal((T, 257))).astype(np.float32)} for _ in range(B)]
    batch = arch.Collator("mix")(samples)
    acc = tq.train_epoch(arch, model, opt, [batch] * 2, 0, 1, False)        # binds the engine, allocates the workspace
    assert opt.skipped() == 0 and model._engine.lstm_mode == 0
    clean = float(acc[0] / acc[1])
    ops.lstm_sticky(ops.lstm_ws(T, B, H)).fill_(1)
    before = model.flat_parameters()[0].clone()
    old = tq.POLL_EVERY
    tq.POLL_EVERY = 2
    try:
        acc = tq.train_epoch(arch, model, opt, [batch] * 6, 1, 1, False)
    finally:
        tq.POLL_EVERY = old
    assert 1 <= opt.skipped() <= 2                              # noticed a step or two later; none after the recovery
    assert model._engine.lstm_mode == 2
    assert "continuing with one launch per step" in capsys.readouterr().err
    assert not torch.equal(model.flat_parameters()[0], before)  # the later steps were applied
    assert bool(torch.isfinite(model.bn.running_var).all()) and bool(torch.isfinite(model.bn.running_mean).all())
    value = float(acc[0] / acc[1])
    assert np.isfinite(value) and 0 < value < 2 * clean
    tq.report_failures(model, opt, 1)                           # nothing left to report
