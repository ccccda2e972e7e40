#!/usr/bin/env python3
"""Mask generation driver: drop-in for the reference's steps/eval_qsub.py (same arguments; the
arch file is imported BY PATH, i.e. the frozen copy exp/<...>/arch.py, steps/eval_qsub.py:43-44).
Writes <dirout>/<id>.npz with keys s1..sS, float32 (257, T), via the arch's compute_masks."""
import argparse
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.abspath(os.path.join(HERE, ".."))
os.environ.setdefault("SEPKERN_HOME", PKG)
for p in (PKG, os.path.join(PKG, "tools"), 'tools'):
  if p not in sys.path:
    sys.path.append(p)

import torch
from torch.utils.data import DataLoader


def get_args():
  parser = argparse.ArgumentParser(description="""This generates and saves output for a test set""")
  parser.add_argument("arch_file", metavar="arch-file", type=str, help="DNN architecture file")
  parser.add_argument("gpu_id", metavar="gpu-id", type=int, help="GPU ID")
  parser.add_argument("model", type=str, help="Trained model to use")
  parser.add_argument("data_dir", metavar="data-dir", type=str, help="Test data directory")
  parser.add_argument("dirout", type=str, help="Output directory")
  parser.add_argument("--model-config", type=str, help="Config file for DNN", default="")
  parser.add_argument("--batch-size", type=int, help="Batch size", default=100)
  parser.add_argument("--seed", type=int, default=None, help="seed for the random h0/c0")
  return parser.parse_args()


def main():
  args = get_args()
  print("Using " + args.arch_file + " DNN architecture")
  sys.path.append(os.path.dirname(os.path.abspath(args.arch_file)))
  m = __import__(os.path.splitext(os.path.basename(args.arch_file))[0])

  print("Using GPU", args.gpu_id)
  torch.cuda.set_device(args.gpu_id)

  print("loading dataset")
  dataset = m.TestSet(args.data_dir)
  dataloader = DataLoader(dataset, batch_size=min(args.batch_size, len(dataset)), shuffle=False,
                          collate_fn=dataset.collator)

  print("loading model")
  kwargs = dict()
  if args.model_config:
    for line in open(args.model_config):
      if '=' in line:
        kwargs[line.split('=')[0]] = line.rstrip().split('=')[1]
  model = m.SepDNN(args.gpu_id, **kwargs)
  model.cuda()
  model.load_state_dict(torch.load(args.model, map_location=lambda storage, loc: storage.cuda()))
  if args.seed is not None:
    model.hidden_generator = torch.Generator(device="cuda")
    model.hidden_generator.manual_seed(args.seed)

  os.makedirs(args.dirout, exist_ok=True)
  model.eval()
  with torch.no_grad():
    for i_batch, sample_batch in enumerate(dataloader):
      m.compute_masks(model, sample_batch, args.dirout)


if __name__ == '__main__':
  main()
