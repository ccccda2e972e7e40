#!/usr/bin/env python3
"""Scoring of reconstructed sources: counterpart of the reference's steps/evaluate_sources.py (same arguments,
same wav inputs, same results/ file names and line formats, so run_eval.sh:88-93 keeps printing "mean SDR").

The reference scores BSS-eval SDR/SIR/SAR with mir_eval (steps/evaluate_sources.py:57), which is a third-party
CPU algorithm that is neither vendored nor available here.  This scorer computes the metric BASELINE.json's
parity gate names instead: scale-invariant SDR (Le Roux et al. 2019; zero-mean), under the best speaker
permutation (as bss_eval_sources searches), written into the *SDR* files; its improvement over the unprocessed
mixture goes to SDRi files.  SIR/SAR are not defined for SI-SDR and are not written.  Off the hot path: numpy.
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
    description="""This script computes SI-SDR (and its improvement over the mixture) for a set of estimated
    sources and ground truth sources.""")
  parser.add_argument("data_dir", metavar="data-dir", type=str, help="Test set data directory")
  parser.add_argument("exp_dir", metavar="exp-dir", type=str, help="Experiment directory")
  return parser.parse_args()


def load_num_src_dict(utt2num_spk_file):
  num_src = dict()
  with open(utt2num_spk_file, 'r') as num_srcF:
    for line in num_srcF:
      num_src[line.split(' ')[0]] = int(line.rstrip().split(' ')[1])
  return num_src


def load_wav(path):
  fs, x = scipy.io.wavfile.read(path)
  return x.astype(np.float64) / 32768.0


def stats_file(path, values):
  values = np.array(values)
  with open(path, 'w') as outF:
    outF.write("Mean:\t" + str(np.mean(values)) + '\n')
    outF.write("Std:\t" + str(np.std(values)) + '\n')
    outF.write("Max:\t" + str(np.amax(values)) + '\n')
    outF.write("Min:\t" + str(np.amin(values)) + '\n')


def main():
  args = get_args()
  from sepkern.sisdr import si_sdr
  import itertools
  num_src_dict = load_num_src_dict(args.data_dir + "/utt2num_spk")
  os.makedirs(args.exp_dir + "/results", exist_ok=True)
  sdrs, sdris = [], []
  files = {n: open(args.exp_dir + "/results/" + n + ".txt", 'w')
           for n in ("session_SDRs", "source_SDRs", "session_SDRis", "source_SDRis")}
  with open(args.data_dir + "/wav.scp", 'r') as wavF:
    for line in wavF:
      ID, oracle_mix_wav = line.rstrip().split(' ')[:2]
      num_src = num_src_dict[ID]
      ests = [load_wav(args.exp_dir + "/wav/s" + str(s + 1) + "/" + ID + ".wav") for s in range(num_src)]
      n = len(ests[0])                                   # the estimates set the length (steps/evaluate_sources.py:51-55)
      refs = [load_wav(oracle_mix_wav.replace("/mix/", "/s" + str(s + 1) + "/"))[:n] for s in range(num_src)]
      mix = load_wav(oracle_mix_wav)[:n]
      best, best_perm = None, None
      for perm in itertools.permutations(range(num_src)):
        v = [si_sdr(ests[perm[s]], refs[s]) for s in range(num_src)]
        if best is None or sum(v) > sum(best):
          best, best_perm = v, perm
      base = [si_sdr(mix, refs[s]) for s in range(num_src)]
      imp = [b - m for b, m in zip(best, base)]
      files["session_SDRs"].write(ID + ' ' + str(sum(best) / num_src) + '\n')
      files["source_SDRs"].write(ID + ''.join(' ' + str(v) for v in best) + '\n')
      files["session_SDRis"].write(ID + ' ' + str(sum(imp) / num_src) + '\n')
      files["source_SDRis"].write(ID + ''.join(' ' + str(v) for v in imp) + '\n')
      sdrs += best
      sdris += imp
  for f in files.values():
    f.close()
  stats_file(args.exp_dir + "/results/SDR_stats.txt", sdrs)
  stats_file(args.exp_dir + "/results/SDRi_stats.txt", sdris)


if __name__ == '__main__':
  main()
