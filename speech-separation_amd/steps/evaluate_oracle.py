#!/usr/bin/env python3
"""Oracle-mask upper bound: counterpart of the reference's steps/evaluate_oracle.py (non-segment branch,
steps/evaluate_oracle.py:120-145; its segments branch does not run: `use_seg`/`rage`/`oracle_mask` NameErrors).

For every utterance of <data-dir>/wav.scp: STFT of the mixture and of each source on the GPU (sk_stft), the
ideal ratio mask |S_i| / |M| (or the binary mask with --hard-mask), mask-apply + iSTFT on the GPU
(sk_mask_istft), then the score.  The reference scores with mir_eval BSS-eval (absent); this writes SI-SDR
(no permutation search, like the reference's compute_permutation=False) in the same files under
<data-dir>/oracle_{soft,hard}_mask_eval/.
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
  from sepkern.sisdr import si_sdr
  dir_out = args.data_dir + ("/oracle_hard_mask_eval/" if args.hard_mask else "/oracle_soft_mask_eval/")
  os.makedirs(dir_out, exist_ok=True)
  sessF = open(dir_out + "session_SDRs.txt", 'w')
  srcF = open(dir_out + "source_SDRs.txt", 'w')
  allv = []
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
      vals = []
      for i in range(num_src):
        est = wav[0][i].cpu().numpy().astype(np.float64)
        ref = pcm[i + 1].cpu().numpy().astype(np.float64)[:len(est)] / 32768.0
        vals.append(si_sdr(est, ref))
      sessF.write(reco_id + ' ' + str(sum(vals) / num_src) + '\n')
      srcF.write(reco_id + ''.join(' ' + str(v) for v in vals) + '\n')
      allv += vals
  sessF.close()
  srcF.close()
  with open(dir_out + "SDR_stats.txt", 'w') as outF:
    v = np.array(allv)
    outF.write("Mean:\t%s\nStd:\t%s\nMax:\t%s\nMin:\t%s\n" % (np.mean(v), np.std(v), np.amax(v), np.amin(v)))


if __name__ == '__main__':
  main()
