#!/usr/bin/env python3
"""Training driver: drop-in for the reference's steps/train_qsub.py (same positional arguments
and options, same files written: intermediate_models/{init,NNN}.mdl, final.mdl,
train_stats/{train,cv}_loss.txt with lines "EEE <loss>", plots).

Loop semantics follow steps/train_qsub.py:113-155: per batch compute_loss -> backward ->
clip_grad_norm_(0.25) -> Adam(lr) step; epoch loss = sum(loss*norm)/sum(norm); CV pass and
checkpoint when epoch % 5 == 4.  Differences, all on the host side:
  * the per-batch loss/norm bookkeeping stays on the GPU (one sync per epoch instead of two per
    step, steps/train_qsub.py:118-119);
  * clip + Adam run fused over the flat parameter buffer (sepkern.optim.ClipAdam); --torch-optimizer
    restores the reference's torch calls on the same parameters;
  * launched under torch.distributed.run it trains data-parallel: utterances are sharded across
    ranks, gradients all-reduced over RCCL inside backward, rank 0 writes the files;
  * `np.float` (steps/train_qsub.py:60, gone from numpy) is float.
"""
import argparse
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.abspath(os.path.join(HERE, ".."))
for p in (PKG, os.path.join(PKG, "tools"), os.path.join(PKG, "archs"), 'tools', 'archs'):
  if p not in sys.path:
    sys.path.append(p)

import torch
from torch.utils.data import DataLoader, Subset


def get_args():
  parser = argparse.ArgumentParser(description="""This script trains a separation neural network""")
  parser.add_argument("arch_file", metavar="arch-file", type=str, help="DNN architecture file")
  parser.add_argument("gpu_id", metavar="gpu-id", type=int, help="GPU ID")
  parser.add_argument("data_dir", metavar="data-dir", type=str, help="Training data directory")
  parser.add_argument("dirout", type=str, help="Output directory")
  parser.add_argument("--cv-data-dir", type=str, help="Cross validation data directory", default="")
  parser.add_argument("--train-copy-location", type=str, help="Copy training data here for I/O purposes", default="")
  parser.add_argument("--model-config", type=str, help="Config file for DNN", default="")
  parser.add_argument("--batch-size", type=int, help="Batch size", default=100)
  parser.add_argument("--start-epoch", type=int, help="Epoch to start from", default=0)
  parser.add_argument("--num-epochs", type=int, help="Total number of training epochs", default=200)
  parser.add_argument("--learning-rate", type=float, help="Learning rate", default=0.001)
  parser.add_argument("--torch-optimizer", action="store_true",
                      help="use torch clip_grad_norm_ + optim.Adam instead of the fused kernel")
  parser.add_argument("--num-workers", type=int, default=1)
  parser.add_argument("--wav-input", action="store_true",
                      help="read <data-dir>/wav.scp and compute the STFT features on the GPU inside the step "
                           "(arch must provide WavTrainSet) instead of loading feats_train.scp npz files")
  parser.add_argument("--seed", type=int, default=None, help="seed for weights, shuffling and h0/c0")
  return parser.parse_args()


def load_losses(filename, loss_array):
  with open(filename, 'r') as lossF:
    for line in lossF:
      split = line.rstrip().split()
      loss_array[0].append(int(split[0]))
      loss_array[1].append(float(split[1]))


def main():
  args = get_args()
  from sepkern import dist as skdist
  from sepkern.optim import ClipAdam
  rank, world, local = skdist.init_from_env()
  gpu = local if world > 1 else args.gpu_id
  if rank == 0:
    print("Using " + args.arch_file + " DNN architecture")
  m = __import__(args.arch_file)

  if rank == 0:
    print("Using GPU", gpu)
  torch.cuda.set_device(gpu)
  if args.seed is not None:
    torch.manual_seed(args.seed)

  int_model_dir = args.dirout + '/intermediate_models/'
  plot_dir = args.dirout + '/train_stats/plots/'
  loss_file = args.dirout + '/train_stats/train_loss.txt'
  cv_loss_file = args.dirout + '/train_stats/cv_loss.txt'
  if rank == 0:
    os.makedirs(int_model_dir, exist_ok=True)
    os.makedirs(plot_dir, exist_ok=True)

  print("loading datset")
  if args.wav_input:
    dataset = m.WavTrainSet(args.data_dir)
  else:
    dataset = m.TrainSet(args.data_dir, args.train_copy_location)
  collate = dataset.collator
  train_data = dataset if world == 1 else Subset(dataset, skdist.shard_indices(len(dataset), rank, world))
  gen = torch.Generator()
  gen.manual_seed((args.seed or 0) + rank)
  dataloader = DataLoader(train_data, batch_size=args.batch_size, shuffle=True, collate_fn=collate,
                          num_workers=args.num_workers, generator=gen if args.seed is not None else None)
  if args.cv_data_dir:
    cv_dataset = m.WavTrainSet(args.cv_data_dir) if args.wav_input else m.TrainSet(args.cv_data_dir)
    cv_dataloader = DataLoader(cv_dataset, batch_size=args.batch_size, collate_fn=cv_dataset.collator)

  print("initializing model")
  kwargs = dict()
  if args.model_config:
    for line in open(args.model_config):
      if '=' in line:
        kwargs[line.split('=')[0]] = line.rstrip().split('=')[1]
  model = m.SepDNN(gpu, **kwargs)
  model.cuda()
  if args.seed is not None:
    model.hidden_generator = torch.Generator(device="cuda")
    model.hidden_generator.manual_seed(args.seed + 7919 * rank)
  if args.torch_optimizer:
    optimizer = torch.optim.Adam(model.parameters(), lr=args.learning_rate)
  else:
    optimizer = ClipAdam(model, lr=args.learning_rate, max_norm=0.25)
  print("using lr=" + str(args.learning_rate))

  epoch_losses = [[], []]
  epoch_cv_losses = [[], []]
  lossF = open(loss_file, 'a') if rank == 0 else None
  cv_lossF = open(cv_loss_file, 'a') if (rank == 0 and args.cv_data_dir) else None

  if args.start_epoch == 0:
    if rank == 0:
      torch.save(model.state_dict(), int_model_dir + 'init.mdl')
  else:
    model.load_state_dict(torch.load(int_model_dir + str(args.start_epoch).zfill(3) + '.mdl',
                                     map_location=lambda storage, loc: storage.cuda()))
    # beyond the reference (which restarts Adam's moments on resume, steps/train_qsub.py:107): optimizer state
    opt_file = int_model_dir + str(args.start_epoch).zfill(3) + '.opt'
    if os.path.isfile(opt_file):
      optimizer.load_state_dict(torch.load(opt_file, map_location=lambda storage, loc: storage.cuda()))
    load_losses(loss_file, epoch_losses)
    if args.cv_data_dir:
      load_losses(cv_loss_file, epoch_cv_losses)

  print("training")
  for epoch in range(args.start_epoch, args.num_epochs):
    acc = torch.zeros(2, device="cuda", dtype=torch.float64)      # [sum(loss*norm), sum(norm)] of my shard
    for i_batch, sample_batch in enumerate(dataloader):
      loss, norm = m.compute_loss(model, epoch, sample_batch)
      ld = loss.detach().double()
      acc[0] += ld * norm.double()
      acc[1] += norm.double() / world if world > 1 else norm.double()   # norm is already the global one under DP
      loss.backward()
      if args.torch_optimizer:
        torch.nn.utils.clip_grad_norm_(model.parameters(), 0.25)
      optimizer.step()
    if world > 1:
      torch.distributed.all_reduce(acc)
    epoch_loss, epoch_norm = float(acc[0]), float(acc[1])

    if args.cv_data_dir and epoch % 5 == 4:
      cv_acc = torch.zeros(2, device="cuda", dtype=torch.float64)
      model.eval()
      with torch.no_grad():
        for i_batch_cv, sample_batch_cv in enumerate(cv_dataloader):
          if i_batch_cv == 0 and rank == 0:
            cv_loss, cv_norm = m.compute_cv_loss(model, epoch, sample_batch_cv, plot_dir + 'epoch' + str(epoch + 1).zfill(3))
          else:
            cv_loss, cv_norm = m.compute_cv_loss(model, epoch, sample_batch_cv)
          cv_acc[0] += cv_loss.detach().double() * cv_norm.double()
          cv_acc[1] += cv_norm.double()
      model.train()
      cv_val = float(cv_acc[0] / cv_acc[1])
      if rank == 0:
        print("For epoch: " + str(epoch + 1).zfill(3) + " cv set loss is: " + str(cv_val))
        cv_lossF.write(str(epoch + 1).zfill(3) + ' ' + str(cv_val) + '\n')
        cv_lossF.flush()
      epoch_cv_losses[0].append(epoch + 1)
      epoch_cv_losses[1].append(cv_val)

    if rank == 0:
      print("For epoch: " + str(epoch + 1).zfill(3) + " loss is: " + str(epoch_loss / epoch_norm))
      lossF.write(str(epoch + 1).zfill(3) + ' ' + str(epoch_loss / epoch_norm) + '\n')
      lossF.flush()
    epoch_losses[0].append(epoch + 1)
    epoch_losses[1].append(epoch_loss / epoch_norm)
    if epoch % 5 == 4 and rank == 0:
      print("Saving model for epoch " + str(epoch + 1).zfill(3))
      torch.save(model.state_dict(), int_model_dir + str(epoch + 1).zfill(3) + '.mdl')
      torch.save(optimizer.state_dict(), int_model_dir + str(epoch + 1).zfill(3) + '.opt')
      try:
        import plot
        os.makedirs(plot_dir + 'epoch' + str(epoch + 1).zfill(3), exist_ok=True)
        plot.plot_loss(epoch_losses, epoch_cv_losses, plot_dir + 'epoch' + str(epoch + 1).zfill(3) + '/Loss_' +
                       str(epoch_losses[0][0]).zfill(3) + '-' + str(epoch + 1).zfill(3) + '.png')
      except ImportError:
        pass
    sys.stdout.flush()

  if rank == 0:
    torch.save(model.state_dict(), args.dirout + '/final.mdl')
    try:
      import plot
      plot.plot_loss(epoch_losses, epoch_cv_losses, plot_dir + 'Loss_' + str(epoch_losses[0][0]).zfill(3) + '-' +
                     str(args.num_epochs).zfill(3) + '.png')
    except (ImportError, IndexError):
      pass
  if world > 1:
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


if __name__ == '__main__':
  main()
