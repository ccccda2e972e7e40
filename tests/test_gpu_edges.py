"""Edge cases (minimal and ragged sizes, ties, segments) and size-independent properties at BASELINE's full
configuration (3x896, 2 speakers, batch 32 x 400 frames).  The HIP-vs-oracle comparison at that size is in
tests/test_gpu_fullsize.py."""
import itertools
import os
import subprocess
import sys

import numpy as np
import pytest
import scipy.io.wavfile
import torch

from conftest import PKG, ROOT
from oracle import stft as OS
from oracle import upit as OU

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(ROOT, "speech-separation_amd", "archs"))


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need an MI355X")
    from sepkern import ops as _ops
    return _ops


@pytest.mark.parametrize("T,B,H,lens", [(1, 1, 4, [1]), (3, 1, 12, [3]), (2, 33, 20, [2] * 20 + [1] * 13), (9, 2, 8, [9, 1])])
def test_lstm_minimal_and_ragged_shapes(ops, T, B, H, lens):
    g = torch.Generator().manual_seed(T + B + H)
    I = 5
    w = [[tuple((torch.rand(s, generator=g) - 0.5) for s in ((4 * H, I), (4 * H, H), (4 * H,), (4 * H,))) for _ in range(2)]]
    x = torch.randn(T, B, I, generator=g)
    for b, n in enumerate(lens):
        x[n:, b] = 0
    h0, c0 = torch.randn(2, B, H, generator=g), torch.randn(2, B, H, generator=g)
    y_ref, hn_ref, cn_ref = OU.blstm_padded(x, lens, w, h0, c0)
    wih = torch.stack([w[0][d][0] for d in range(2)]).cuda()
    whh = torch.stack([w[0][d][1] for d in range(2)]).cuda()
    bsum = torch.stack([w[0][d][2] + w[0][d][3] for d in range(2)]).reshape(-1).cuda()
    for mode in (1, 2):
        gx = torch.empty(T, B, 2, 4 * H).cuda()
        # gx in the recurrence's gate-interleaved order: reordered weight rows / bias, plain GEMM
        ops.gemm(x.cuda(), ops.gate_rows(wih.view(8 * H, I), H), gx, T * B, 8 * H, I, I, I, 8 * H, transB=True,
                 bias=ops.gate_rows(bsum, H))
        y = torch.full((T, B, 2 * H), float("nan")).cuda()
        hn, cn = torch.empty(2, B, H).cuda(), torch.empty(2, B, H).cuda()
        ws = ops.lstm_fwd(gx, whh, h0.cuda(), c0.cuda(), torch.tensor(lens, dtype=torch.int32).cuda(), y, None, None, hn, cn,
                          T, B, H, mode)
        ops.lstm_status(ws)
        np.testing.assert_allclose(y.cpu().numpy(), y_ref.numpy(), rtol=2e-5, atol=2e-6)
        np.testing.assert_allclose(hn.cpu().numpy(), hn_ref.numpy(), rtol=2e-5, atol=2e-6)
        np.testing.assert_allclose(cn.cpu().numpy(), cn_ref.numpy(), rtol=2e-5, atol=2e-6)


def test_pit_tie_takes_first_permutation_and_zero_input(ops):
    T, B, F, S = 4, 3, 257, 2
    mask = torch.full((T, B, S * F), 0.5).cuda()
    mix = torch.ones(T, B, F).cuda()
    src = [torch.full((T, B, F), 0.25).cuda() for _ in range(S)]        # identical sources: every permutation ties
    res = ops.pit_mse_fwd(mask, mix, src, torch.tensor([4, 3, 1], dtype=torch.int32).cuda())
    assert res["best_perm"].cpu().tolist() == [0, 0, 0]                 # torch.min returns the first minimum too
    z = torch.zeros(T, B, F).cuda()
    res = ops.pit_mse_fwd(mask, z, [z, z], torch.tensor([4, 3, 1], dtype=torch.int32).cuda())
    assert float(res["out"][0]) == 0.0 and float(res["out"][1]) == 8 * F


def test_single_source_npz_trains_on_the_mixture(tmp_path):
    """TrainSet with a mix-only npz maps source1 to the mixture (reference archs/uPIT.py:72-73): 1-speaker PIT == plain MSE."""
    import uPIT
    rng = np.random.default_rng(0)
    os.makedirs(str(tmp_path / "f"))
    with open(str(tmp_path / "feats_train.scp"), "w") as f:
        for i, T in enumerate((6, 4)):
            path = str(tmp_path / "f" / ("u%d.npz" % i))
            np.savez_compressed(path, mix=np.abs(rng.standard_normal((257, T))).astype(np.float32))
            f.write("u%d %s\n" % (i, path))
    ds = uPIT.TrainSet(str(tmp_path))
    batch = ds.collator([ds[0], ds[1]])
    assert sorted(batch.keys()) == ["mix", "source1"]
    torch.manual_seed(0)
    model = uPIT.SepDNN(0, num_spk="1", hidden_dim="32")
    model.cuda()
    model.train()
    loss, norm = uPIT.compute_loss(model, 0, batch)
    loss.backward()
    assert float(norm) == 10 * 257 and np.isfinite(float(loss)) and float(loss) > 0


def test_extract_feats_segments_mode(tmp_path):
    """data-dir/segments cuts utterances by `<seg> <reco> <t0> <t1>` (reference steps/extract_feats.py:51-58,70-81)."""
    from sepkern import synth
    root = str(tmp_path)
    wavroot, data = os.path.join(root, "wav8k"), os.path.join(root, "data")
    ids = synth.write_wav_tree(wavroot, 2, num_spk=2, fixed_samples=16000)
    synth.write_data_dir(data, wavroot, ids)
    with open(os.path.join(data, "segments"), "w") as f:
        f.write("%s-a %s 0.25 1.0\n%s-b %s 1.0 1.75\n%s-a %s 0.5 2.0\n" % (ids[0], ids[0], ids[0], ids[0], ids[1], ids[1]))
    env = dict(os.environ, SEPKERN_HOME=PKG)
    r = subprocess.run([sys.executable, os.path.join(PKG, "steps", "extract_feats.py"), data, "train", os.path.join(root, "feats")],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    segs = [l.split(' ')[0] for l in open(os.path.join(data, "feats_train.scp"))]
    assert segs == [ids[0] + "-a", ids[0] + "-b", ids[1] + "-a"]
    z = np.load(os.path.join(root, "feats", ids[0] + "-b.npz"))
    _, pcm = scipy.io.wavfile.read(os.path.join(wavroot, "s1", ids[0] + ".wav"))
    ref = OS.stft_mag(OS.pcm16_to_float(pcm[8000:8000 + 6000]))
    assert z["s1"].shape == ref.shape == (257, 1 + 6000 // 128)
    np.testing.assert_allclose(z["s1"], ref, atol=1e-5 * ref.max())


def test_full_size_properties():
    """BASELINE configs[1] (3x896, 2 speakers, 32 x 400): bitwise run-to-run determinism (no atomics anywhere),
    persistent == one-launch-per-step recurrence, PIT loss invariant under a source swap, loss decreases."""
    import uPIT
    from sepkern.optim import ClipAdam
    torch.manual_seed(0)
    model = uPIT.SepDNN(0, hidden_dim="896", num_layers="3")
    model.cuda()
    model.train()
    g = torch.Generator(device="cuda").manual_seed(5)
    T, B, F = 400, 32, 257
    lens = torch.full((B,), T, dtype=torch.int32, device="cuda")
    lens[1::3] = 333
    valid = (torch.arange(T, device="cuda")[:, None] < lens[None, :]).float().unsqueeze(2)
    mix = torch.rand(T, B, F, device="cuda", generator=g) * valid
    srcs = [torch.rand(T, B, F, device="cuda", generator=g) * 0.5 * valid for _ in range(2)]
    h = (torch.randn(6, B, 896, device="cuda", generator=g), torch.randn(6, B, 896, device="cuda", generator=g))

    def run(sources, mode=0):
        model._bind().lstm_mode = mode
        model.next_hidden = h
        loss, norm = uPIT.compute_loss_padded(model, mix, sources, lens)
        loss.backward()
        return float(loss), float(norm), model.flat_parameters()[1].clone()
    l1, n1, g1 = run(srcs)
    l2, n2, g2 = run(srcs)
    assert l1 == l2 and torch.equal(g1, g2)                                  # bitwise reproducible
    l3, _, g3 = run(srcs[::-1])
    np.testing.assert_allclose(l3, l1, rtol=1e-6)                             # permutation invariant
    np.testing.assert_allclose(float((g3 - g1).norm() / g1.norm()), 0.0, atol=1e-5)
    l4, _, g4 = run(srcs, mode=2)
    np.testing.assert_allclose(l4, l1, rtol=1e-6)                             # per-step launches == persistent launch
    np.testing.assert_allclose(float((g4 - g1).norm() / g1.norm()), 0.0, atol=1e-6)
    assert n1 == float(lens.sum()) * F and torch.isfinite(g1).all()
    model._bind().lstm_mode = 0
    opt = ClipAdam(model, lr=1e-3, max_norm=0.25)
    first = None
    for _ in range(4):
        model.next_hidden = h
        loss, _ = uPIT.compute_loss_padded(model, mix, srcs, lens)
        loss.backward()
        opt.step()
        first = first if first is not None else float(loss)
    assert float(loss) < first
