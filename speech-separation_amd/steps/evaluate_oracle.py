#!/usr/bin/env python3
"""Oracle-mask upper bound: counterpart of the reference's steps/evaluate_oracle.py (non-segment branch,
steps/evaluate_oracle.py:120-145; its segments branch does not run: `use_seg`/`rage`/`oracle_mask` NameErrors).

For every utterance of <data-dir>/wav.scp: STFT of the mixture and of each source on the GPU (sk_stft), the
ideal ratio mask |S_i| / |M| (or the binary mask with --hard-mask), mask-apply + iSTFT on the GPU
(sk_mask_istft), then the score: BSS Eval SDR / SIR / SAR without permutation search (the reference calls
mir_eval's bss_eval_sources with compute_permutation=False, steps/evaluate_oracle.py:118,143; here
sepkern/bsseval.py) into {session,source}_{SDR,SIR,SAR}s.txt + *_stats.txt under
<data-dir>/oracle_{soft,hard}_mask_eval/, and SI-SDR under its own SISDR names.
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
  parser = argparse.ArgumentParser(description="""Evaluates oracle (ideal) masks through the same STFT -> mask ->
  iSTFT path the separation models use""")
  parser.add_argument("data_dir", metavar="data-dir", type=str, help="Data directory with wav.scp")
  parser.add_argument("--hard-mask", action='store_true', help="Use hard mask", default=False)
  parser.add_argument("--fft-dim", type=int, help="Dimension of FFT", default=512)
  parser.add_argument("--step-size", type=int, help="STFT step size", default=128)
  parser.add_argument("--sample-rate", type=int, help="Audio sample rate", default=8000)
  return parser.parse_args()


def main():
  args = get_args()
  if args.fft_dim != 512 or args.step_size != 128:
    raise ValueError("the HIP STFT kernels are built for --fft-dim 512 --step-size 128")
  import torch
  from sepkern import ops
  from sepkern.bsseval import bss_eval_sources
  from sepkern.sisdr import si_sdr
  from evaluate_sources import MetricFiles
  dir_out = args.data_dir + ("/oracle_hard_mask_eval/" if args.hard_mask else "/oracle_soft_mask_eval/")
  os.makedirs(dir_out, exist_ok=True)
  out = {m: MetricFiles(dir_out, m) for m in ("SDR", "SIR", "SAR", "SISDR")}
  with open(args.data_dir + "/wav.scp", 'r') as listF:
    for line in listF:
      reco_id, filename = line.rstrip().split(' ')
      wav_files = sorted(glob.glob(filename.replace("/mix/", "/*/")))
      pcm = []
      for f in wav_files:
        fs, x = scipy.io.wavfile.read(f)
        if fs != args.sample_rate or x.dtype != np.int16:
          raise ValueError("%s: expected %d Hz 16-bit PCM" % (f, args.sample_rate))
        pcm.append(torch.from_numpy(np.ascontiguousarray(x)).cuda())
      num_src = len(pcm) - 1
      mix_spec = ops.stft_batch([pcm[0]], want_complex=True, layout="FT")[0]           # (257, T) complex64
      mags = torch.stack(ops.stft_batch(pcm[1:], want_complex=False, layout="FT"))      # (S, 257, T)
      if args.hard_mask:
        masks = torch.nn.functional.one_hot(mags.argmax(0), num_src).permute(2, 0, 1).float()
      else:
        masks = mags / mix_spec.abs().clamp_min(1e-20)
      wav, _ = ops.mask_istft([mix_spec], [[masks[i].contiguous() for i in range(num_src)]], want_pcm=False)
      ests = np.stack([wav[0][i].cpu().numpy().astype(np.float64) for i in range(num_src)])
      refs = np.stack([pcm[i + 1].cpu().numpy().astype(np.float64)[:ests.shape[1]] / 32768.0 for i in range(num_src)])
      sdr, sir, sar, _ = bss_eval_sources(refs, ests, compute_permutation=False)
      out["SDR"].add(reco_id, sdr)
      out["SIR"].add(reco_id, sir)
      out["SAR"].add(reco_id, sar)
      out["SISDR"].add(reco_id, [si_sdr(ests[i], refs[i]) for i in range(num_src)])
  for files in out.values():
    files.close()


if __name__ == '__main__':
  main()
