"""The recipe's data path end to end on the GPU, through the step programs that mirror the
reference's steps/*.py CLIs: synthetic wav tree -> extract_feats (train + test) -> train_qsub ->
eval_qsub on the frozen arch copy -> reconstruct_sources.  Every file format of SURVEY.md Appendix A
is checked, numerics against the CPU oracle (BASELINE config 1: 8 synthetic utterances)."""
import os
import shutil
import subprocess
import sys

import numpy as np
import pytest
import scipy.io.wavfile
import torch

from conftest import PKG
from oracle import stft as OS

pytestmark = pytest.mark.gpu
STEPS = os.path.join(PKG, "steps")


def run(*cmd, cwd=None):
    env = dict(os.environ, SEPKERN_HOME=PKG, PYTHONPATH=PKG + os.pathsep + os.environ.get("PYTHONPATH", ""))
    r = subprocess.run([sys.executable] + list(cmd), cwd=cwd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, "%s failed:\n%s\n%s" % (cmd[0], r.stdout[-2000:], r.stderr[-3000:])
    return r.stdout


def test_recipe_steps_end_to_end(tmp_path):
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need an MI355X")
    from sepkern import synth
    root = str(tmp_path)
    wavroot, data = os.path.join(root, "wav8k"), os.path.join(root, "data", "syn")
    ids = synth.write_wav_tree(wavroot, 8, num_spk=2, min_s=1.0, max_s=2.0, id_list=os.path.join(root, "id_lists", "syn.txt"))
    synth.write_data_dir(data, wavroot, ids)
    assert open(os.path.join(data, "wav.scp")).readline().split(' ')[1].startswith(wavroot + "/mix/")

    # ---- stage 1: features (train = magnitudes of mix+sources, test = complex mix)
    ftrain, ftest = os.path.join(root, "feats", "syn_train"), os.path.join(root, "feats", "syn_test")
    run(os.path.join(STEPS, "extract_feats.py"), data, "train", ftrain)
    lines = open(os.path.join(data, "feats_train.scp")).read().splitlines()
    assert [l.split(' ')[0] for l in lines] == ids and all(l.endswith(".npz") for l in lines)
    assert open(os.path.join(data, "utt2num_spk")).read().splitlines() == ["%s 2" % i for i in ids]
    for i in ids[:3]:
        z = np.load(os.path.join(ftrain, i + ".npz"))
        assert z.files == ["mix", "s1", "s2"]
        for key, sub in (("mix", "mix"), ("s1", "s1"), ("s2", "s2")):
            _, pcm = scipy.io.wavfile.read(os.path.join(wavroot, sub, i + ".wav"))
            ref = OS.stft_mag(OS.pcm16_to_float(pcm))
            assert z[key].dtype == np.float32 and z[key].shape == ref.shape == (257, 1 + len(pcm) // 128)
            np.testing.assert_allclose(z[key], ref, atol=1e-5 * ref.max())
    run(os.path.join(STEPS, "extract_feats.py"), data, "test", ftest)
    z = np.load(os.path.join(ftest, ids[0] + ".npz"))
    _, pcm = scipy.io.wavfile.read(os.path.join(wavroot, "mix", ids[0] + ".wav"))
    assert z.files == ["mix"] and z["mix"].dtype == np.complex64
    np.testing.assert_allclose(z["mix"], OS.stft(OS.pcm16_to_float(pcm)), atol=1e-5 * np.abs(z["mix"]).max())

    # ---- stage 2: training (2x64 via the conf-file mechanism, strings as in steps/train_qsub.py:87-91)
    exp = os.path.join(root, "exp", "uPIT_syn")
    os.makedirs(os.path.join(exp, "train_stats"), exist_ok=True)
    os.makedirs(os.path.join(root, "exp", "uPIT_wav", "train_stats"), exist_ok=True)
    shutil.copy(os.path.join(PKG, "archs", "uPIT.py"), os.path.join(exp, "arch.py"))     # run_train.sh:56
    with open(os.path.join(exp, "conf"), "w") as f:
        f.write("hidden_dim=64\nnum_layers=2\n")
    out = run(os.path.join(STEPS, "train_qsub.py"), "uPIT", "0", data, exp, "--model-config", os.path.join(exp, "conf"),
              "--batch-size", "4", "--num-epochs", "5", "--seed", "1", "--cv-data-dir", data)
    assert "For epoch: 005 loss is:" in out and "cv set loss is" in out
    for f in ("intermediate_models/init.mdl", "intermediate_models/005.mdl", "final.mdl"):
        assert os.path.isfile(os.path.join(exp, f))
    tl = [l.split() for l in open(os.path.join(exp, "train_stats", "train_loss.txt")).read().splitlines()]
    assert [l[0] for l in tl] == ["001", "002", "003", "004", "005"]
    losses = [float(l[1]) for l in tl]
    assert all(np.isfinite(losses)) and losses[-1] < losses[0]
    assert open(os.path.join(exp, "train_stats", "cv_loss.txt")).read().split()[0] == "005"
    sd = torch.load(os.path.join(exp, "final.mdl"), map_location="cpu")
    assert list(sd.keys())[0] == "blstm.weight_ih_l0" and sd["blstm.weight_hh_l1_reverse"].shape == (256, 64)

    out = run(os.path.join(STEPS, "train_qsub.py"), "uPIT", "0", data, os.path.join(root, "exp", "uPIT_wav"), "--model-config",
              os.path.join(exp, "conf"), "--batch-size", "4", "--num-epochs", "2", "--seed", "1", "--wav-input")
    assert "For epoch: 002 loss is:" in out

    # ---- stage 3: masks from the FROZEN arch copy, then reconstruction
    mdir = os.path.join(exp, "masks")
    run(os.path.join(STEPS, "eval_qsub.py"), os.path.join(exp, "arch.py"), "0", os.path.join(exp, "final.mdl"), data, mdir,
        "--model-config", os.path.join(exp, "conf"), "--batch-size", "3", "--seed", "2")
    run(os.path.join(STEPS, "reconstruct_sources.py"), data, exp)
    for i in ids:
        mz = np.load(os.path.join(mdir, i + ".npz"))
        spec = np.load(os.path.join(ftest, i + ".npz"))["mix"]
        assert mz.files == ["s1", "s2"]
        for k in mz.files:
            assert mz[k].shape == spec.shape and mz[k].dtype == np.float32 and 0 <= mz[k].min() and mz[k].max() <= 1
            fs, got = scipy.io.wavfile.read(os.path.join(exp, "wav", k, i + ".wav"))
            _, ref = OS.reconstruct(spec, mz[k])
            assert fs == 8000 and got.dtype == np.int16 and got.shape == ref.shape == (128 * (spec.shape[1] - 1),)
            d = np.abs(got.astype(np.int32) - ref.astype(np.int32))
            assert d.max() <= 1 and (d > 0).mean() < 5e-3

    # ---- stage 4: scoring (BSS Eval SDR/SIR/SAR in the reference's results files, run_eval.sh:88-93; SI-SDR beside them)
    run(os.path.join(STEPS, "evaluate_sources.py"), data, exp)
    st = open(os.path.join(exp, "results", "SDR_stats.txt")).read().splitlines()
    assert [l.split("\t")[0] for l in st] == ["Mean:", "Std:", "Max:", "Min:"] and np.isfinite(float(st[0].split("\t")[1]))
    sess = open(os.path.join(exp, "results", "session_SDRs.txt")).read().splitlines()
    assert [l.split(' ')[0] for l in sess] == ids
    src = open(os.path.join(exp, "results", "source_SDRs.txt")).readline().split(' ')
    assert src[0] == ids[0] and len(src) == 3
    for name in ("SIR_stats.txt", "SAR_stats.txt", "session_SIRs.txt", "source_SARs.txt", "SISDR_stats.txt", "SISDRi_stats.txt",
                 "session_SISDRs.txt", "source_SISDRis.txt"):
        assert os.path.isfile(os.path.join(exp, "results", name)), name
    sdr0 = [float(v) for v in src[1:]]
    si0 = [float(v) for v in open(os.path.join(exp, "results", "source_SISDRs.txt")).readline().split(' ')[1:]]
    assert all(a >= b - 1e-6 for a, b in zip(sorted(sdr0), sorted(si0)))   # a 512-tap filter explains at least what a gain does

    # ---- oracle-mask upper bound through the same STFT -> mask -> iSTFT kernels (steps/evaluate_oracle.py)
    run(os.path.join(STEPS, "evaluate_oracle.py"), data)
    soft = float(open(os.path.join(data, "oracle_soft_mask_eval", "SDR_stats.txt")).readline().split("\t")[1])
    run(os.path.join(STEPS, "evaluate_oracle.py"), data, "--hard-mask")
    hard = float(open(os.path.join(data, "oracle_hard_mask_eval", "SDR_stats.txt")).readline().split("\t")[1])
    trained = float(st[0].split("\t")[1])
    assert soft > trained + 3 and hard > trained + 3 and soft > 5      # ideal masks bound a 5-epoch toy model from above
    # ... and against the CPU oracle of the same computation (reference steps/evaluate_oracle.py:120-145): STFT of the
    # mixture and the sources, ideal ratio mask, mask-apply + iSTFT (oracle/stft.py), BSS Eval without permutation
    from sepkern.bsseval import bss_eval_sources
    lines = open(os.path.join(data, "oracle_soft_mask_eval", "source_SDRs.txt")).read().splitlines()
    assert [l.split(' ')[0] for l in lines] == ids
    for i in (0, len(ids) - 1):
        pcm = [OS.pcm16_to_float(scipy.io.wavfile.read(os.path.join(wavroot, d, ids[i] + ".wav"))[1]) for d in ("mix", "s1", "s2")]
        mix_spec = OS.stft(pcm[0])
        mags = [np.abs(OS.stft(p)) for p in pcm[1:]]
        ests = np.stack([OS.istft(mix_spec * (m / np.maximum(np.abs(mix_spec), 1e-20))) for m in mags])
        refs = np.stack([p[:ests.shape[1]] for p in pcm[1:]])
        sdr, _, _, _ = bss_eval_sources(refs.astype(np.float64), ests.astype(np.float64), compute_permutation=False)
        got = [float(v) for v in lines[i].split(' ')[1:]]
        np.testing.assert_allclose(got, sdr, atol=0.02)                # dB; fp32 kernels vs the fp64-accumulating oracle

    # ---- resume from epoch 5 with the optimizer state saved next to the checkpoint: with --seed the continued run
    # ends on EXACTLY the weights of an uninterrupted 6-epoch run (deterministic kernels, Adam moments restored,
    # shuffling and h0/c0 re-seeded per epoch) -- the reference restarts Adam on resume (steps/train_qsub.py:107)
    assert os.path.isfile(os.path.join(exp, "intermediate_models", "005.opt"))
    out = run(os.path.join(STEPS, "train_qsub.py"), "uPIT", "0", data, exp, "--model-config", os.path.join(exp, "conf"),
              "--batch-size", "4", "--num-epochs", "6", "--start-epoch", "5", "--seed", "1")
    assert "For epoch: 006 loss is:" in out
    exp2 = os.path.join(root, "exp", "uPIT_syn_straight")
    os.makedirs(exp2)
    run(os.path.join(STEPS, "train_qsub.py"), "uPIT", "0", data, exp2, "--model-config", os.path.join(exp, "conf"),
        "--batch-size", "4", "--num-epochs", "6", "--seed", "1")
    a = torch.load(os.path.join(exp, "final.mdl"), map_location="cpu")
    b = torch.load(os.path.join(exp2, "final.mdl"), map_location="cpu")
    for k in a:
        assert torch.equal(a[k], b[k]), k
    l5 = open(os.path.join(exp, "train_stats", "train_loss.txt")).read().splitlines()
    l6 = open(os.path.join(exp2, "train_stats", "train_loss.txt")).read().splitlines()
    assert l5 == l6 and len(l6) == 6


def test_wav_input_pipeline_equals_npz_pipeline(tmp_path):
    """WavTrainSet (PCM in, STFT on the GPU inside the step) gives the same loss and gradients as
    the npz feature path on the same utterances and the same (h0, c0)."""
    sys.path.insert(0, os.path.join(PKG, "archs"))
    import uPIT
    from sepkern import synth
    root = str(tmp_path)
    wavroot, data = os.path.join(root, "wav8k"), os.path.join(root, "data", "syn")
    ids = synth.write_wav_tree(wavroot, 5, num_spk=2, min_s=0.6, max_s=1.2)
    synth.write_data_dir(data, wavroot, ids)
    run(os.path.join(STEPS, "extract_feats.py"), data, "train", os.path.join(root, "feats"))
    torch.manual_seed(4)
    model = uPIT.SepDNN(0, hidden_dim="64", num_layers="2")
    model.cuda()
    model.train()
    h = (torch.randn(4, 5, 64).cuda(), torch.randn(4, 5, 64).cuda())
    out = []
    for ds in (uPIT.TrainSet(data), uPIT.WavTrainSet(data)):
        batch = ds.collator([ds[i] for i in range(len(ds))])
        model.next_hidden = h
        loss, norm = uPIT.compute_loss(model, 0, batch)
        loss.backward()
        out.append((float(loss), float(norm), model.flat_parameters()[1].clone()))
    assert out[0][1] == out[1][1]
    np.testing.assert_allclose(out[1][0], out[0][0], rtol=2e-5)
    np.testing.assert_allclose(out[1][2].cpu().numpy(), out[0][2].cpu().numpy(), rtol=2e-3, atol=2e-7)


def test_bench_two_rank_rehearsal_on_one_gpu(tmp_path):
    """bench.py's N>1 path (barriers, global-norm + gradient all-reduce, max-over-ranks timing, one JSON line
    from rank 0) with 2 ranks on cuda:0 over gloo; the per-step LSTM mode because two processes cannot both
    keep a persistent grid resident on one GPU.  The real N>1 runs use nccl (= RCCL), one rank per GPU."""
    import json
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, SEPKERN_BENCH_ONE_DEVICE="1", SEPKERN_DIST_BACKEND="gloo", SEPKERN_LSTM_MODE="2")
    root = os.path.dirname(PKG)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(root, "bench.py"),
                        "--gpus", "2", "--steps", "2", "--warmup", "1", "--hidden", "64", "--layers", "2", "--batch", "4",
                        "--frames", "40"], cwd=root, env=env, capture_output=True, text=True, timeout=400)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["frames_per_step"] == 2 * 4 * 40
    assert d["value"] > 0 and d["unit"] == "frames/s" and "cpu_baseline" not in d


def test_bench_launches_its_own_ranks_when_no_launcher_did():
    """`python bench.py --gpus 2` with no torchrun around it (the way the round driver calls bench.py): the parent starts
    the two ranks itself as fresh child processes BEFORE anything touches the GPU, relays rank 0's single JSON line and the
    worst return code.  Rehearsed here with both ranks on cuda:0 over gloo (one-device box); on a node it is nccl = RCCL,
    one rank per GPU."""
    import json
    env = dict(os.environ, SEPKERN_BENCH_ONE_DEVICE="1", SEPKERN_DIST_BACKEND="gloo", SEPKERN_LSTM_MODE="2")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    root = os.path.dirname(PKG)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--hidden", "64",
                        "--layers", "2", "--batch", "4", "--frames", "40"], cwd=root, env=env, capture_output=True, text=True,
                       timeout=400)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["frames_per_step"] == 2 * 4 * 40 and d["value"] > 0
    dd = d["distributed"]
    assert dd["backend"] == "gloo" and dd["world_size"] == 2 and dd["distinct_devices"] == 1      # both ranks on cuda:0 here
    assert dd["launcher"] == "self" and dd["grad_allreduce"] == "single"
    # the opt-in chunked exchange (SEPKERN_DP_OVERLAP=1: layer-ordered chunks on a communication stream while the backward
    # pass is still running) gives the same training trajectory: every gradient element is summed exactly once
    r2 = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--hidden", "64",
                         "--layers", "2", "--batch", "4", "--frames", "40"], cwd=root, env=dict(env, SEPKERN_DP_OVERLAP="1"),
                        capture_output=True, text=True, timeout=400)
    assert r2.returncode == 0, r2.stderr[-3000:]
    d2 = json.loads([l for l in r2.stdout.splitlines() if l.startswith("{")][0])
    assert d2["distributed"]["grad_allreduce"] == "chunked-overlapped"
    assert d2["config"]["mean_loss"] == d["config"]["mean_loss"]
    assert d2["distributed"]["lstm_per_step_launches_by_rank"] == [1, 1]
    # ... and beside a PERSISTENT recurrence: rank 0 keeps its one-launch grid (mode 0), rank 1 launches per step, the chunks
    # of the gradient go out on the communication stream while rank 0's persistent backward kernels run (GradReducer's
    # stream / event ordering next to a persistent grid; on a node every rank is persistent and the backend is RCCL).
    # `distinct_devices` is a CHECKED field: 1 here (both ranks on cuda:0), == world_size on a real node.
    r3 = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--hidden", "64",
                         "--layers", "2", "--batch", "4", "--frames", "40"], cwd=root,
                        env=dict({k: v for k, v in env.items() if k != "SEPKERN_LSTM_MODE"}, SEPKERN_DP_OVERLAP="1",
                                 SEPKERN_LSTM_MODE_BY_RANK="0,2"), capture_output=True, text=True, timeout=400)
    assert r3.returncode == 0, r3.stderr[-3000:]
    d3 = json.loads([l for l in r3.stdout.splitlines() if l.startswith("{")][0])
    dd3 = d3["distributed"]
    assert dd3["grad_allreduce"] == "chunked-overlapped" and dd3["world_size"] == 2 and dd3["distinct_devices"] == 1 < dd3["world_size"]
    assert dd3["lstm_per_step_launches_by_rank"] in ([0, 1], [1, 1])      # ([1, 1]: rank 0's grid timed out beside rank 1 and fell back)
    assert (dd3["lstm_per_step_launches_by_rank"] == [0, 1]) == ("lstm_fallback" not in d3)
    assert abs(d3["config"]["mean_loss"] - d["config"]["mean_loss"]) <= 1e-5 * d["config"]["mean_loss"]
    # a rank that fails takes the job down with a non-zero code instead of leaving the others in a collective
    bad = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--hidden", "63"],
                         cwd=root, env=env, capture_output=True, text=True, timeout=400)
    assert bad.returncode != 0 and not [l for l in bad.stdout.splitlines() if l.startswith("{")]


def test_bench_line_carries_the_contract_fields():
    """One small `python bench.py` run on the GPU: ONE JSON line with the driver's contract fields, the `roofline` object
    (dominant class + every MFMA-bound class under `by_kernel`, the recurrences with their MFMA floor and hand-off share)
    and -- on request -- the CPU baseline."""
    import json
    root = os.path.dirname(PKG)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "3", "--warmup", "1", "--hidden", "320", "--layers", "2",
                        "--batch", "32", "--frames", "60", "--no-cpu-baseline"], cwd=root, capture_output=True, text=True, timeout=400)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["unit"] == "frames/s" and d["dtype"] == "f32" and d["vs_baseline"] is None
    assert d["config"]["frames_per_step"] == 32 * 60 and "workload" in d["config"]
    assert abs(d["value"] - 32 * 60 / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-3
    rf = d["roofline"]
    assert rf["bound"] == "mfma" and rf["unit"] == "TFLOP/s" and 0 < rf["frac"] <= 1 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    assert "traffic" in rf and rf["top_kernel"] in rf["by_kernel"]
    by = rf["by_kernel"]
    assert {"gemm_f32_kernel", "lstm_fwd_kernel", "lstm_bwd_kernel"} <= set(by)
    ms = [v["ms_per_step"] for v in by.values()]
    assert ms == sorted(ms, reverse=True)                          # largest share of the step first
    for k in ("lstm_fwd_kernel", "lstm_bwd_kernel"):
        assert by[k]["us_per_time_step"] > by[k]["mfma_floor_us"] > 0 and by[k]["handoff_us"] > 0


def test_bench_default_run_carries_the_secondary_block():
    """The DEFAULT `python bench.py` (the command the round driver runs; here with fewer headline steps and without the CPU
    baseline) also times the other one-GPU BASELINE configurations in the same process and reports them under "secondary":
    the variable-length set (fp32 and bf16), bf16 3-speaker, RSH 4-speaker -- VERDICT r04 item 1 --, the headline workload on
    the fp32-MFMA kernels throughout (the reference's literal arithmetic) and the reference's own default model and batch size
    (2 x 600, 100 utterances) -- VERDICT r05 item 2; `aux` carries the streaming kernels `north_star` names (STFT, mask-iSTFT, PIT
    forward / backward) as us, GB/s of algorithmic bytes and fraction of the HBM peak, `library` the path and build flags of the
    library that ran, `roofline.power_note` the sustained rate / clock / watts of the dominant GEMM class.  The headline fields
    stay what they were; every secondary entry is self-consistent (value = frames per step / time per step) and names its
    workload."""
    import json
    root = os.path.dirname(PKG)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "3", "--warmup", "1", "--no-cpu-baseline"],
                       cwd=root, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["dtype"] == "f32" and d["config"]["frames_per_step"] == 32 * 400 and "uPIT 3x896 BLSTM, 2-spk" in d["config"]["workload"]
    sec = d["secondary"]
    assert set(sec) == {"ragged", "bf16_3spk", "bf16_ragged", "rsh_4spk", "fp32_mfma", "ref_default_2x600_b100"}
    for name, s in sec.items():
        assert "error" not in s, (name, s)
        for k in ("value", "ms_per_step", "frames_per_step", "dtype", "step_frac_of_mfma_peak", "by_kernel", "workload", "numerics"):
            assert k in s, (name, k)
        assert abs(s["value"] - s["frames_per_step"] / (s["ms_per_step"] * 1e-3)) / s["value"] < 1e-3
        assert 0 < s["step_frac_of_mfma_peak"] <= 1 and "lstm_fallback" not in s
        assert {"lstm_fwd_kernel", "lstm_bwd_kernel"} <= set(s["by_kernel"])
    assert sec["ragged"]["dtype"] == "f32" and "U(24000, 64000)" in sec["ragged"]["workload"] and "packed rows" in sec["ragged"]["workload"]
    assert 32 * 188 <= sec["ragged"]["frames_per_step"] <= 32 * 501
    assert sec["bf16_3spk"]["dtype"] == "bf16" and "3-spk" in sec["bf16_3spk"]["workload"] and sec["bf16_3spk"]["frames_per_step"] == 12800
    assert sec["bf16_ragged"]["dtype"] == "bf16" and "U(24000, 64000)" in sec["bf16_ragged"]["workload"]
    assert sec["bf16_ragged"]["frames_per_step"] == sec["ragged"]["frames_per_step"]      # the same batches
    assert "RSH 2x600" in sec["rsh_4spk"]["workload"] and "4-spk" in sec["rsh_4spk"]["workload"]
    # the bf16 configurations are faster than their fp32 counterparts, the ragged step is not slower than the uniform one
    assert sec["bf16_3spk"]["ms_per_step"] < d["ms_per_step"] and sec["bf16_ragged"]["ms_per_step"] < sec["ragged"]["ms_per_step"]
    # the literal fp32 arithmetic: same workload, GEMMs on the fp32-MFMA kernels and named so; slower than the shipped arithmetic
    fm = sec["fp32_mfma"]
    assert fm["frames_per_step"] == 12800 and fm["dtype"] == "f32" and "GEMMs on the fp32-MFMA kernels" in fm["numerics"]
    assert "recurrences: fp32-MFMA products" in fm["numerics"] and fm["ms_per_step"] > d["ms_per_step"]
    assert not any(k.startswith("gemm_f32_split") for k in fm["by_kernel"])
    rd = sec["ref_default_2x600_b100"]
    assert "uPIT 2x600 BLSTM, 2-spk" in rd["workload"] and rd["frames_per_step"] == 100 * 400
    # streaming kernels: all four, sane numbers
    aux = d["aux"]
    assert {"stft_kernel", "istft_kernel", "pit_fwd", "pit_bwd"} <= set(aux)
    for k in ("stft_kernel", "istft_kernel", "pit_fwd", "pit_bwd"):
        assert aux[k]["us_per_launch"] > 0 and 0 < aux[k]["frac_of_hbm_peak"] <= 1, (k, aux[k])
        assert abs(aux[k]["GBs_algorithmic"] - aux[k]["MB_algorithmic_per_launch"] / aux[k]["us_per_launch"] * 1e3) / aux[k]["GBs_algorithmic"] < 0.02
    assert d["library"]["build_flags"] == 0 and d["library"]["path"].endswith("libsepkern.so")
    pn = d["roofline"]["power_note"]
    assert "error" not in pn and pn["sustained"]["tflops_fp32_equivalent"] > 100


def test_staged_batches_equal_the_plain_loader_and_keep_up_with_the_step(tmp_path):
    """steps/train_qsub.py's loop through sepkern.data.Prefetcher (batches staged on the GPU ahead of their step, as packed
    rows) -- VERDICT r03 item 2:
      * numerics: a staged batch gives the same loss and gradients as the same batch handed over by the plain loader
        (npz path: bit for bit -- the same PackedSequence.data, copied by another route; wav path: the same STFT kernel);
      * rate: an epoch of train_epoch THROUGH the loader + staging runs at >= 0.6 of the rate of the same steps on batches
        that are already resident on the GPU (on this 20-step corpus the epoch's start-up -- one loader batch being built
        from scratch -- is a visible share; on 2 000 utterances the factor is 0.93, profiles/r04_stage_walls.txt)."""
    sys.path.insert(0, os.path.join(PKG, "archs"))
    sys.path.insert(0, STEPS)
    import importlib
    import uPIT
    from torch.utils.data import DataLoader
    from sepkern import synth
    from sepkern.data import Prefetcher, host_threads
    from sepkern.optim import ClipAdam
    tq = importlib.import_module("train_qsub")
    host_threads()
    root = str(tmp_path)
    wavroot, data = os.path.join(root, "wav8k"), os.path.join(root, "data", "syn")
    ids = synth.write_wav_tree(wavroot, 320, num_spk=2, min_s=1.0, max_s=3.0)
    synth.write_data_dir(data, wavroot, ids)
    run(os.path.join(STEPS, "extract_feats.py"), data, "train", os.path.join(root, "feats"))
    torch.manual_seed(9)
    model = uPIT.SepDNN(0, hidden_dim="600", num_layers="2")           # the reference's own size (archs/uPIT.py:115)
    model.cuda()
    model.train()
    dev = torch.device("cuda", 0)
    # ---- numerics
    for ds in (uPIT.TrainSet(data), uPIT.WavTrainSet(data)):
        batch = ds.collator([ds[i] for i in range(7)])
        staged = Prefetcher.stage(batch, dev)
        assert set(staged) >= {"packed"} and staged["packed"][2].B == 7
        h = (torch.randn(4, 7, 600).cuda(), torch.randn(4, 7, 600).cuda())
        got = []
        for b in (batch, staged):
            model.next_hidden = h
            loss, norm = uPIT.compute_loss(model, 0, b)
            loss.backward()
            got.append((float(loss), float(norm), model.flat_parameters()[1].clone()))
        assert got[0][:2] == got[1][:2] and torch.equal(got[0][2], got[1][2])
    # ---- rate
    opt = ClipAdam(model, lr=1e-3, max_norm=0.25)
    ds = uPIT.TrainSet(data)
    loader = DataLoader(ds, batch_size=16, shuffle=True, collate_fn=ds.collator, num_workers=6, persistent_workers=True,
                        prefetch_factor=2)
    pf = Prefetcher(loader, dev, depth=2)
    resident = list(pf)
    assert len(resident) == 20 and sum(b["packed"][2].B for b in resident) == 320

    def epoch(batches):
        import time
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        acc = tq.train_epoch(uPIT, model, opt, batches, 0, 1, False)
        value = float(acc[0] / acc[1])
        assert np.isfinite(value) and float(acc[1]) == sum(b["packed"][2].R for b in resident) * 257
        return time.perf_counter() - t0
    epoch(resident)                                                    # warm-up (allocator, workspaces)
    t_res = min(epoch(resident) for _ in range(2))
    epoch(pf)
    t_pf = min(epoch(pf) for _ in range(2))
    assert opt.skipped() == 0
    assert t_res / t_pf >= 0.6, "epoch through the loader %.3f s, on resident batches %.3f s" % (t_pf, t_res)


def test_bench_ragged_line_and_numerics_field():
    """`bench.py --ragged` (SURVEY.md 8d's variable-length set: U(24k, 64k) samples, a different batch every step, packed
    rows): frames_per_step counts the valid frames only, the workload string says so, and config.numerics names the forward
    recurrence's arithmetic (the exact bf16-split product by default, the r03 tagged hand-off or the plain fp32-MFMA product
    on request) -- VERDICT r03 items 1, 3 and 4."""
    import json
    root = os.path.dirname(PKG)
    out = {}
    for tag, env in (("split3", {}), ("tagged", {"SEPKERN_LSTM_FWD": "0,1,1,0,0,8,1,0"}), ("exact", {"SEPKERN_LSTM_FWD": "0,1,1,0,0,0,0,0"})):
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "4", "--warmup", "1", "--hidden", "320", "--layers", "2",
                            "--batch", "32", "--ragged", "--no-cpu-baseline"], cwd=root, env=dict(os.environ, **env),
                           capture_output=True, text=True, timeout=400)
        assert r.returncode == 0, r.stderr[-3000:]
        out[tag] = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    d = out["split3"]
    assert "three-way bf16 split" in d["config"]["numerics"]
    d = out["tagged"]
    assert "U(24000, 64000)" in d["config"]["workload"] and "packed rows" in d["config"]["workload"]
    assert 32 * 188 <= d["config"]["frames_per_step"] <= 32 * 501
    assert abs(d["value"] - d["config"]["frames_per_step"] / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-3
    assert "tagged" in d["config"]["numerics"] and "3 ulp" in d["config"]["numerics"]
    assert "exact hand-off" in out["exact"]["config"]["numerics"] and "tagged" not in out["exact"]["config"]["numerics"].split(";")[0]
    # same seeds, same batches: the two hand-offs agree on the loss to fp32 rounding
    assert abs(out["exact"]["config"]["mean_loss"] - d["config"]["mean_loss"]) <= 2e-5 * d["config"]["mean_loss"]
