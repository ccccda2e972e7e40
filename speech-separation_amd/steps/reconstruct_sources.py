#!/usr/bin/env python3
"""Mask-apply + iSTFT reconstruction on the MI355X: drop-in for the reference's
steps/reconstruct_sources.py (same arguments, inputs and outputs).

For every line of <data-dir>/feats_test.scp: complex mix spectrogram npz + <exp-dir>/masks/<ID>.npz
-> <exp-dir>/wav/<source>/<ID>.wav (int16, --sample-rate).  np.multiply + librosa.istft + *32767 +
astype(int16) of the reference (steps/reconstruct_sources.py:39-42) run fused in sk_mask_istft,
including the reference's no-clipping (wrapping) int16 conversion.

Host side: the npz files of the NEXT chunk are inflated by a thread pool while this chunk is on the GPU, every chunk
crosses PCIe as one pinned copy per buffer (spectra, masks, samples), and the wav files are written by the pool.
"""
import argparse
import concurrent.futures
import os
import sys
import time

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
  parser.add_argument("--writers", type=int, default=8, help="threads that inflate the npz inputs and write the wav files")
  return parser.parse_args()


def main():
  args = get_args()
  if args.step_size != 128:
    raise ValueError("the HIP iSTFT kernel is built for --step-size 128")
  import torch
  from sepkern import ops
  from sepkern.data import host_threads
  host_threads()

  entries = []
  with open(args.data_dir + '/feats_test.scp', 'r') as featsF:
    for line in featsF:
      ID, path = line.rstrip().split(' ')[:2]
      entries.append((ID, path))

  F = 257
  t_start, n_frames = time.perf_counter(), 0
  chunks = [entries[i0:i0 + args.batch_files] for i0 in range(0, len(entries), args.batch_files)]

  def load(ID, path):
    masks = np.load(args.exp_dir + "/masks/" + ID + '.npz')
    keys = tuple(masks.files)
    return ID, np.ascontiguousarray(np.load(path)['mix'].astype(np.complex64)), keys, \
        [np.ascontiguousarray(masks[k].astype(np.float32)) for k in keys]

  def write_wav(path, samples):
    os.makedirs(os.path.dirname(path), exist_ok=True)
    scipy.io.wavfile.write(path, args.sample_rate, samples)

  with concurrent.futures.ThreadPoolExecutor(max_workers=max(1, args.writers)) as pool:
    loading = [pool.submit(load, ID, path) for ID, path in chunks[0]] if chunks else []
    writing = []
    for ci in range(len(chunks)):
      loaded = [f.result() for f in loading]
      loading = [pool.submit(load, ID, path) for ID, path in chunks[ci + 1]] if ci + 1 < len(chunks) else []
      groups = {}                                   # utterances with the same source keys go in one launch
      for item in loaded:
        groups.setdefault(item[2], []).append(item)
      for keys, items in groups.items():
        S, Ts = len(keys), [int(m.shape[1]) for _, m, _, _ in items]
        for _, m, _, mk in items:
          if m.shape[0] != F or any(z.shape != m.shape for z in mk):
            raise ValueError("reconstruct_sources: spectra and masks must be (257, T) with equal T")
        mix_h = torch.from_numpy(np.concatenate([m.reshape(-1) for _, m, _, _ in items])).pin_memory()
        mask_h = torch.from_numpy(np.concatenate([z.reshape(-1) for _, _, _, mk in items for z in mk])).pin_memory()
        _, pcm, offs = ops.mask_istft_flat(mix_h.to("cuda", non_blocking=True), mask_h.to("cuda", non_blocking=True), Ts, S,
                                           want_float=False)
        pcm_h = torch.empty(pcm.numel(), dtype=torch.int16).pin_memory()
        pcm_h.copy_(pcm, non_blocking=True)
        torch.cuda.synchronize()
        samples = pcm_h.numpy()
        for u, (ID, _, _, _) in enumerate(items):
          n_frames += Ts[u]
          for s_, source in enumerate(keys):
            o = offs[u * S + s_]
            writing.append(pool.submit(write_wav, args.exp_dir + "/wav/" + source + '/' + ID + ".wav", samples[o:o + 128 * (Ts[u] - 1)]))
      writing = [w for w in writing if not (w.done() and w.result() is None)]
    for w in writing:
      w.result()
  dt = time.perf_counter() - t_start
  print("reconstruct_sources: %d utterances, %d frames in %.2f s = %.0f frames/s" % (len(entries), n_frames, dt, n_frames / max(dt, 1e-9)),
        file=sys.stderr)


if __name__ == '__main__':
  main()
