#!/usr/bin/env python3
"""Scoring of reconstructed sources: the arguments, wav inputs and results/ files of the reference's
steps/evaluate_sources.py (:36-110), so that run_eval.sh:88-93 finds results/SDR_stats.txt and prints "mean SDR".

  results/{session,source}_{SDR,SIR,SAR}s.txt, results/{SDR,SIR,SAR}_stats.txt
      BSS Eval v3 SDR / SIR / SAR with the 512-tap allowed-distortion filter and the permutation search, as the
      reference gets them from mir_eval.separation.bss_eval_sources (steps/evaluate_sources.py:57); computed by
      sepkern/bsseval.py, a restatement of the published algorithm (mir_eval is not in this image).
  results/{session,source}_SISDRs.txt, {session,source}_SISDRis.txt, SISDR_stats.txt, SISDRi_stats.txt
      additionally: scale-invariant SDR (Le Roux et al. 2019) under its best permutation and its improvement over
      the unprocessed mixture -- the metric BASELINE.json's +-0.1 dB parity gate names.  Never mixed into the
      SDR files.
Off the hot path: numpy on the host.
"""
import argparse
import itertools
import os
import sys

import numpy as np
import scipy.io.wavfile

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))


def get_args(argv=None):
  parser = argparse.ArgumentParser(
    description="""This script computes BSS Eval SDR/SIR/SAR (and SI-SDR with its improvement over the mixture) for a
    set of estimated sources and ground truth sources.""")
  parser.add_argument("data_dir", metavar="data-dir", type=str, help="Test set data directory")
  parser.add_argument("exp_dir", metavar="exp-dir", type=str, help="Experiment directory")
  return parser.parse_args(argv)


def read_pairs(path):
  """`<key> <value>` lines (wav.scp, utt2num_spk) in file order."""
  with open(path) as f:
    return [tuple(line.rstrip('\n').split(' ')[:2]) for line in f if line.strip()]


def load_wav(path):
  fs, x = scipy.io.wavfile.read(path)
  return x.astype(np.float64) / 32768.0


class MetricFiles:
  """One metric's three files: session_<M>s.txt (`<id> <mean over sources>`), source_<M>s.txt (`<id> v1 v2 ...`)
  and <M>_stats.txt (Mean/Std/Max/Min over all sources of all utterances, tab-separated)."""

  def __init__(self, results_dir, metric):
    self.results_dir, self.metric, self.values = results_dir, metric, []
    self.session = open(os.path.join(results_dir, "session_%ss.txt" % metric), 'w')
    self.source = open(os.path.join(results_dir, "source_%ss.txt" % metric), 'w')

  def add(self, utt_id, per_source):
    per_source = [float(v) for v in per_source]
    self.session.write(utt_id + ' ' + str(sum(per_source) / len(per_source)) + '\n')
    self.source.write(utt_id + ''.join(' ' + str(v) for v in per_source) + '\n')
    self.values += per_source

  def close(self):
    self.session.close()
    self.source.close()
    v = np.array(self.values)
    with open(os.path.join(self.results_dir, "%s_stats.txt" % self.metric), 'w') as f:
      f.write("Mean:\t" + str(np.mean(v)) + '\n')
      f.write("Std:\t" + str(np.std(v)) + '\n')
      f.write("Max:\t" + str(np.amax(v)) + '\n')
      f.write("Min:\t" + str(np.amin(v)) + '\n')


def best_si_sdr(ests, refs):
  from sepkern.sisdr import si_sdr
  best = None
  for perm in itertools.permutations(range(len(refs))):
    v = [si_sdr(ests[perm[s]], refs[s]) for s in range(len(refs))]
    if best is None or sum(v) > sum(best):
      best = v
  return best


def main(argv=None):
  args = get_args(argv)
  from sepkern.bsseval import bss_eval_sources
  from sepkern.sisdr import si_sdr
  num_src = {k: int(v) for k, v in read_pairs(args.data_dir + "/utt2num_spk")}
  results = args.exp_dir + "/results"
  os.makedirs(results, exist_ok=True)
  out = {m: MetricFiles(results, m) for m in ("SDR", "SIR", "SAR", "SISDR", "SISDRi")}
  for utt_id, mix_wav in read_pairs(args.data_dir + "/wav.scp"):
    S = num_src[utt_id]
    ests = [load_wav(args.exp_dir + "/wav/s" + str(s + 1) + "/" + utt_id + ".wav") for s in range(S)]
    n = len(ests[0])                              # the first estimate sets the length (steps/evaluate_sources.py:51-55)
    ests = np.stack([e[:n] for e in ests])
    refs = np.stack([load_wav(mix_wav.replace("/mix/", "/s" + str(s + 1) + "/"))[:n] for s in range(S)])
    sdr, sir, sar, _ = bss_eval_sources(refs, ests)
    out["SDR"].add(utt_id, sdr)
    out["SIR"].add(utt_id, sir)
    out["SAR"].add(utt_id, sar)
    si = best_si_sdr(ests, refs)
    mix = load_wav(mix_wav)[:n]
    out["SISDR"].add(utt_id, si)
    out["SISDRi"].add(utt_id, [v - si_sdr(mix, refs[s]) for s, v in enumerate(si)])
  for files in out.values():
    files.close()


if __name__ == '__main__':
  main()
