#!/usr/bin/env python3
"""Mask-apply + iSTFT reconstruction on the MI355X: drop-in for the reference's
steps/reconstruct_sources.py (same arguments, inputs and outputs).

For every line of <data-dir>/feats_test.scp: complex mix spectrogram npz + <exp-dir>/masks/<ID>.npz
-> <exp-dir>/wav/<source>/<ID>.wav (int16, --sample-rate).  np.multiply + librosa.istft + *32767 +
astype(int16) of the reference (steps/reconstruct_sources.py:39-42) run fused in sk_mask_istft,
including the reference's no-clipping (wrapping) int16 conversion.
"""
import argparse
import os
import sys

import numpy as np
import scipy.io.wavfile

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))


def get_args():
  parser = argparse.ArgumentParser(
    description="""This script reconstructs wav files from mix spectrograms and estimated source masks""")
  parser.add_argument("data_dir", metavar="data-dir", type=str, help="Data directory")
  parser.add_argument("exp_dir", metavar="exp-dir", type=str, help="Experiment directory")
  parser.add_argument("--step-size", type=int, help="STFT step size", default=128)
  parser.add_argument("--sample-rate", type=int, help="Audio sample rate", default=8000)
  parser.add_argument("--batch-files", type=int, help="utterances per kernel launch", default=128)
  return parser.parse_args()


def main():
  args = get_args()
  if args.step_size != 128:
    raise ValueError("the HIP iSTFT kernel is built for --step-size 128")
  import torch
  from sepkern import ops

  entries = []
  with open(args.data_dir + '/feats_test.scp', 'r') as featsF:
    for line in featsF:
      ID, path = line.rstrip().split(' ')[:2]
      entries.append((ID, path))

  for i0 in range(0, len(entries), args.batch_files):
    chunk = entries[i0:i0 + args.batch_files]
    groups = {}                                   # utterances with the same source keys go in one launch
    for ID, path in chunk:
      masks = np.load(args.exp_dir + "/masks/" + ID + '.npz')
      groups.setdefault(tuple(masks.files), []).append((ID, np.load(path)['mix'], masks))
    for keys, items in groups.items():
      specs = [torch.from_numpy(np.ascontiguousarray(m.astype(np.complex64))).cuda() for _, m, _ in items]
      mk = [[torch.from_numpy(np.ascontiguousarray(z[k].astype(np.float32))).cuda() for k in keys] for _, _, z in items]
      _, pcm = ops.mask_istft(specs, mk, want_float=False)
      for u, (ID, _, _) in enumerate(items):
        for s, source in enumerate(keys):
          wav_out = args.exp_dir + "/wav/" + source + '/' + ID + ".wav"
          os.makedirs(os.path.dirname(wav_out), exist_ok=True)
          scipy.io.wavfile.write(wav_out, args.sample_rate, pcm[u][s].cpu().numpy())


if __name__ == '__main__':
  main()
