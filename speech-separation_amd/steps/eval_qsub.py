#!/usr/bin/env python3
"""Mask generation for a test set: <dirout>/<id>.npz with keys s1..sS, float32 (257, T).

Takes the arguments steps/qsub_eval.sh passes to the reference's steps/eval_qsub.py (:14-37; the reference's own
file also runs unchanged on this package's arch modules -- INTEGRATION.md).  The arch file is given BY PATH: it is
the copy frozen next to the model at training time, exp/<...>/arch.py (steps/eval_qsub.py:43-44).

This driver is organised around the two costs of the stage, which the reference runs strictly one after the other:
the GPU forward pass of a batch, and the zlib compression of its masks (np.savez_compressed, a few ms per
utterance on one core).  Here a small pool of writer threads compresses batch k while the GPU computes batch k+1
(zlib releases the GIL), and under torch.distributed.run the utterances are dealt to the ranks by index -- one GPU
each, no collective -- the way the reference shards feature extraction over SGE array tasks
(steps/extract_feats.sh:41-53).
"""
import argparse
import concurrent.futures
import importlib.util
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.abspath(os.path.join(HERE, ".."))
os.environ.setdefault("SEPKERN_HOME", PKG)      # a frozen arch.py finds the sepkern package through it
for p in (PKG, os.path.join(PKG, "tools"), 'tools'):
  if p not in sys.path:
    sys.path.append(p)

import numpy as np
import torch
from torch.utils.data import DataLoader, Subset


def get_args(argv=None):
  parser = argparse.ArgumentParser(description="""This generates and saves output for a test set""")
  parser.add_argument("arch_file", metavar="arch-file", type=str, help="DNN architecture file")
  parser.add_argument("gpu_id", metavar="gpu-id", type=int, help="GPU ID")
  parser.add_argument("model", type=str, help="Trained model to use")
  parser.add_argument("data_dir", metavar="data-dir", type=str, help="Test data directory")
  parser.add_argument("dirout", type=str, help="Output directory")
  parser.add_argument("--model-config", type=str, help="Config file for DNN", default="")
  parser.add_argument("--batch-size", type=int, help="Batch size", default=100)
  parser.add_argument("--seed", type=int, default=None, help="seed for the random h0/c0 (archs/uPIT.py:121-127)")
  parser.add_argument("--writers", type=int, default=8, help="threads compressing and writing the npz files (zlib: ~25 MB/s per thread on spectra at any level)")
  parser.add_argument("--num-workers", type=int, default=4, help="loader processes inflating the test features")
  return parser.parse_args(argv)


def load_arch(path):
  """Import an arch module from its file path under its own base name."""
  path = os.path.abspath(path)
  sys.path.append(os.path.dirname(path))
  name = os.path.splitext(os.path.basename(path))[0]
  spec = importlib.util.spec_from_file_location(name, path)
  module = importlib.util.module_from_spec(spec)
  sys.modules[name] = module
  spec.loader.exec_module(module)
  return module


def restore_model(m, args, gpu):
  conf = {}
  if args.model_config:
    with open(args.model_config) as f:
      conf = dict(line.rstrip().split('=', 1) for line in f if '=' in line)
  model = m.SepDNN(gpu, **conf)
  model.cuda()
  model.load_state_dict(torch.load(args.model, map_location=lambda storage, loc: storage.cuda()))
  if args.seed is not None:
    model.hidden_generator = torch.Generator(device="cuda")
    model.hidden_generator.manual_seed(args.seed)
  model.eval()
  return model


def write_masks(dirout, name, arrays):
  np.savez_compressed(os.path.join(dirout, name), **arrays)


def main(argv=None):
  args = get_args(argv)
  from sepkern import dist as skdist
  rank, world, local = skdist.init_from_env()
  gpu = local if world > 1 else args.gpu_id
  torch.cuda.set_device(gpu)
  from sepkern.data import host_threads
  host_threads()
  m = load_arch(args.arch_file)
  if rank == 0:
    print("mask generation with", args.arch_file, "on", world, "GPU(s)")

  dataset = m.TestSet(args.data_dir)
  mine = dataset if world == 1 else Subset(dataset, skdist.shard_indices(len(dataset), rank, world))
  os.makedirs(args.dirout, exist_ok=True)
  model = restore_model(m, args, gpu)
  pending = []
  import time
  t_start, n_frames = time.perf_counter(), 0
  if len(mine):
    batches = DataLoader(mine, batch_size=min(args.batch_size, len(mine)), shuffle=False, collate_fn=dataset.collator,
                         num_workers=max(0, args.num_workers))
    with concurrent.futures.ThreadPoolExecutor(max_workers=max(1, args.writers)) as pool, torch.no_grad():
      for batch in batches:
        if hasattr(m, "estimate_masks"):
          for name, arrays in m.estimate_masks(model, batch):
            n_frames += next(iter(arrays.values())).shape[1]
            pending.append(pool.submit(write_masks, args.dirout, name, arrays))
        else:                                   # an arch module that only implements the reference protocol
          m.compute_masks(model, batch, args.dirout)
      for job in pending:
        job.result()                            # re-raise a writer's exception
  if world > 1:
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()
  if rank == 0:
    print("wrote masks for", len(dataset), "utterances to", args.dirout)
    dt = time.perf_counter() - t_start
    print("eval_qsub: %d frames (this rank) in %.2f s = %.0f frames/s" % (n_frames, dt, n_frames / max(dt, 1e-9)), file=sys.stderr)


if __name__ == '__main__':
  main()
