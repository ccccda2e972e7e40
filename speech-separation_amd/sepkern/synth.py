"""Synthetic WSJ0-2mix-shaped data (no corpus is available offline).

Utterances are 8 kHz int16 mixtures of `num_spk` "speech-like" sources: seeded white Gaussian
noise through a one-pole low-pass, modulated by a 3-6 Hz syllabic envelope, peak-normalised to
0.5, with per-source gains of +-snr/2 dB, snr ~ U(0, 2.5) (the range seen in the reference's
id_lists/wsj_tr.txt, whose ids look like 011a0101_0.061105_401c020r_-0.061105).  The directory
layout is the reference's: <root>/{mix,s1,s2[,s3]}/<id>.wav plus an id list, so
local/prepare_data_dir.sh-style wav.scp files and steps/extract_feats.py work on it unchanged.
"""
import os

import numpy as np
import scipy.io.wavfile
import scipy.signal

SR = 8000


def speech_like(n, seed):
    rng = np.random.default_rng(seed)
    x = rng.standard_normal(n)
    x = scipy.signal.lfilter([1.0], [1.0, -0.9], x)                  # one-pole low-pass
    rate = rng.uniform(3.0, 6.0)
    phase = rng.uniform(0, 2 * np.pi)
    env = 0.55 + 0.45 * np.sin(2 * np.pi * rate * np.arange(n) / SR + phase)
    x = x * env ** 2
    return 0.5 * x / (np.abs(x).max() + 1e-12)


def utterance(utt, n_samples, num_spk=2):
    """-> (id, mix int16, [source int16...], float sources)."""
    rng = np.random.default_rng(10_000 + utt)
    snr = float(rng.uniform(0.0, 2.5))
    gains_db = [snr / 2, -snr / 2] + [0.0] * (num_spk - 2)
    srcs = [speech_like(n_samples, 1000 * utt + s) * 10 ** (gains_db[s] / 20.0) for s in range(num_spk)]
    scale = 0.9 / max(1.0, np.abs(np.sum(srcs, axis=0)).max() / 0.9)
    srcs = [s * min(1.0, scale) for s in srcs]
    mix = np.sum(srcs, axis=0)
    to16 = lambda v: np.clip(np.round(v * 32768.0), -32768, 32767).astype(np.int16)   # noqa: E731
    uid = "%03da%04d_%.6f_%03dc%04d_%.6f" % (utt % 1000, utt, snr / 2, (utt * 7) % 1000, utt, -snr / 2)
    return uid, to16(mix), [to16(s) for s in srcs]


def write_wav_tree(root, n_utts, num_spk=2, min_s=3.0, max_s=8.0, fixed_samples=None, seed=0, id_list=None):
    """Writes <root>/{mix,s1..}/<id>.wav and returns the list of ids (also to `id_list` if given)."""
    rng = np.random.default_rng(seed)
    for d in ["mix"] + ["s%d" % (s + 1) for s in range(num_spk)]:
        os.makedirs(os.path.join(root, d), exist_ok=True)
    ids = []
    for u in range(n_utts):
        n = int(fixed_samples) if fixed_samples else int(rng.uniform(min_s, max_s) * SR)
        uid, mix, srcs = utterance(u, n, num_spk)
        scipy.io.wavfile.write(os.path.join(root, "mix", uid + ".wav"), SR, mix)
        for s, w in enumerate(srcs):
            scipy.io.wavfile.write(os.path.join(root, "s%d" % (s + 1), uid + ".wav"), SR, w)
        ids.append(uid)
    if id_list:
        os.makedirs(os.path.dirname(os.path.abspath(id_list)), exist_ok=True)
        with open(id_list, "w") as f:
            f.write("".join(i + "\n" for i in ids))
    return ids


def write_data_dir(data_dir, wav_root, ids):
    """data/<set>/wav.scp as local/prepare_data_dir.sh:35 writes it: `<id> <wavroot>/mix/<id>.wav`."""
    os.makedirs(data_dir, exist_ok=True)
    with open(os.path.join(data_dir, "wav.scp"), "w") as f:
        for i in ids:
            f.write("%s %s/mix/%s.wav\n" % (i, os.path.abspath(wav_root), i))


def pcm_batch(batch, n_samples=51072, num_spk=2, first_utt=0, lengths=None):
    """Host int16 waveforms for a batch: list over utterances of [mix, s1, ..]; 51072 samples = 400 frames."""
    out = []
    for b in range(batch):
        n = int(lengths[b]) if lengths is not None else n_samples
        _, mix, srcs = utterance(first_utt + b, n, num_spk)
        out.append([mix] + srcs)
    return out
