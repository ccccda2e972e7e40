#!/usr/bin/env python3
"""STFT feature extraction on the MI355X: drop-in for the reference's steps/extract_feats.py
(same positional arguments, options, input wav.scp/segments and output files).

  data-dir/wav.scp[.N]      `<id> <root>/mix/<id>.wav`; sources are found by globbing /mix/ -> /*/
  train: <feat-dir>/<id>.npz with mix, s1..sS = |STFT| float32 (257, T)
  test : <feat-dir>/<id>.npz with mix = complex64 STFT (257, T)
  data-dir/feats_<type>.scp[.N], data-dir/utt2num_spk[.N]

The reference calls librosa.load + librosa.stft per file on the CPU (steps/extract_feats.py:85-89,
104-105); here wav files are read as int16 PCM, batched, and transformed by sk_stft (PCM scaling,
reflect padding, periodic Hann, 512-point FFT and magnitude fused in one kernel) writing directly in
the on-disk (257, T) layout.  Only 16-bit PCM at the requested sample rate is supported (the
reference's data is wav8k); other inputs raise instead of being silently resampled.
"""
import argparse
import glob
import os
import sys

import numpy as np
import scipy.io.wavfile

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))


def get_args():
  parser = argparse.ArgumentParser(description="""Extracts and saves STFT-based features for source separation""")
  parser.add_argument("data_dir", metavar="data-dir", type=str, help="Data directory with wav.scp")
  parser.add_argument("data_type", metavar="data-type", type=str, choices=['train', 'test'],
                      help="""Dataset type. Train stores magnitude spectra for mixture and all sources.
                      Test stores just mix spectrum.""")
  parser.add_argument("feat_dir", metavar="feat-dir", type=str, help="Output directory for features")
  parser.add_argument("--fft-dim", type=int, help="Dimension of FFT", default=512)
  parser.add_argument("--step-size", type=int, help="STFT step size", default=128)
  parser.add_argument("--sample-rate", type=int, help="Audio sample rate", default=8000)
  parser.add_argument("--batch-files", type=int, help="wav files per kernel launch", default=256)
  return parser.parse_args()


def read_pcm(path, sr, offset=None, duration=None):
  fs, x = scipy.io.wavfile.read(path)
  if fs != sr:
    raise ValueError("%s is sampled at %d Hz, expected %d (resampling is not built)" % (path, fs, sr))
  if x.dtype != np.int16:
    raise ValueError("%s: only 16-bit PCM wav is supported, got %s" % (path, x.dtype))
  if x.ndim > 1:
    raise ValueError("%s: only mono wav is supported" % path)
  if offset is not None:       # librosa.load(offset=, duration=): whole frames from int(offset*sr)
    start = int(offset * sr)
    x = x[start:start + int(duration * sr)]
  return np.ascontiguousarray(x)


def main():
  args = get_args()
  if args.fft_dim != 512 or args.step_size != 128:
    raise ValueError("the HIP STFT kernel is built for --fft-dim 512 --step-size 128")
  import torch
  from sepkern import ops

  job_suffix = ''
  if os.environ.get("SGE_TASK_ID", 'undefined') != 'undefined':
    job_suffix = '.' + os.environ["SGE_TASK_ID"]

  os.makedirs(args.feat_dir, exist_ok=True)
  featF = open(args.data_dir + "/feats_" + args.data_type + ".scp" + job_suffix, 'w')
  utt2num_spkF = open(args.data_dir + "/utt2num_spk" + job_suffix, 'w')

  seg_dict = None
  if os.path.isfile(args.data_dir + "/segments" + job_suffix):
    seg_dict = {}
    for line in open(args.data_dir + "/segments" + job_suffix):
      seg = line.rstrip().split()
      seg_dict.setdefault(seg[1], []).append((seg[0], float(seg[2]), float(seg[3])))

  # work list: (output id, [(npz key, wav path, offset, duration)], num_spk)
  work = []
  with open(args.data_dir + "/wav.scp" + job_suffix, 'r') as listF:
    for line in listF:
      reco_id, filename = line.rstrip().split(' ')
      wav_files = sorted(glob.glob(filename.replace("/mix/", "/*/")))
      num_spk = max(1, len(wav_files) - 1)
      if args.data_type == "train":
        keyed = [('mix' if i == 0 else 's' + str(i), f) for i, f in enumerate(wav_files)]
      else:
        keyed = [('mix', filename)]
      segs = seg_dict[reco_id] if seg_dict is not None else [(reco_id, None, None)]
      for seg_id, t0, t1 in segs:
        dur = None if t0 is None else t1 - t0
        work.append((seg_id, [(k, f, t0, dur) for k, f in keyed], num_spk))

  want_complex = args.data_type == "test"
  step = max(1, args.batch_files)
  for i0 in range(0, len(work), step):
    chunk = work[i0:i0 + step]
    wavs, owner = [], []
    for wi, (_, items, _) in enumerate(chunk):
      for key, f, t0, dur in items:
        wavs.append(torch.from_numpy(read_pcm(f, args.sample_rate, t0, dur)).cuda())
        owner.append((wi, key))
    specs = ops.stft_batch(wavs, want_complex=want_complex, layout="FT")
    out = [dict() for _ in chunk]
    for (wi, key), sp in zip(owner, specs):
      out[wi][key] = sp.cpu().numpy()
    for (seg_id, _, num_spk), file_dict in zip(chunk, out):
      np.savez_compressed(os.path.join(args.feat_dir, seg_id), **file_dict)
      featF.write(seg_id + ' ' + os.path.join(args.feat_dir, seg_id) + '.npz\n')
      utt2num_spkF.write(seg_id + ' ' + str(num_spk) + '\n')

  featF.close()
  utt2num_spkF.close()


if __name__ == '__main__':
  main()
