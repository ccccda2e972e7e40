#!/usr/bin/env python3
"""Training driver with the command line, file names and line formats of the reference's steps/train_qsub.py
(:17-53 arguments; :73-76,105,138,144,150,155 files: intermediate_models/{init,NNN}.mdl, final.mdl,
train_stats/{train,cv}_loss.txt with lines "EEE <loss>", train_stats/plots/...), so steps/qsub_train.sh can call
it unchanged.  The step itself is the arch module's: compute_loss -> backward -> clip 0.25 -> Adam
(steps/train_qsub.py:116-122); an epoch's loss is sum(loss * norm) / sum(norm) (:118-119,143); every 5th epoch
runs the cross-validation set and writes a checkpoint (:124-152).

What is organised differently here, on the host side only:
  * the per-batch loss bookkeeping stays on the GPU: one host sync per epoch instead of two per step;
  * clip + Adam run fused over the model's flat parameter buffer (sepkern.optim.ClipAdam; --torch-optimizer
    restores the reference's torch calls on the same parameters);
  * under torch.distributed.run (one process per GPU) it trains data-parallel: replicas are made identical by a
    broadcast from rank 0, every epoch's utterances are dealt to the ranks in length-balanced global batches
    (sepkern.dist.EpochShards: same step count on every rank), gradients are summed over RCCL inside backward,
    the ranks' BatchNorm running statistics are averaged after every epoch (sepkern.dist.average_bn_buffers: the CV
    loss printed is the loss of the model saved), the cross-validation set is sharded too, rank 0 writes the files;
  * checkpoints also carry the optimizer state (NNN.opt), which the reference loses on resume (:107);
  * a recurrence launch that timed out never reaches the weights (the fused optimizer skips that step on the
    device) and is reported when the epoch ends;
  * batches are staged on the GPU AHEAD of the step that consumes them (sepkern.data.Prefetcher: loader workers inflate
    and pack in parallel, a thread copies through pinned memory on its own stream, --prefetch batches deep): the
    reference inflates, packs and copies each batch synchronously in front of its step (:113-117), which at 36 ms per
    step is the bound.  --prefetch 0 --num-workers 1 is the reference's loop.  Every epoch's wall time and frames/s go
    to stderr.
"""
import argparse
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.abspath(os.path.join(HERE, ".."))
for p in (PKG, os.path.join(PKG, "tools"), os.path.join(PKG, "archs"), 'tools', 'archs'):
  if p not in sys.path:
    sys.path.append(p)

import torch
from torch.utils.data import DataLoader, Subset

CHECKPOINT_EVERY = 5          # steps/train_qsub.py:124,148: epoch % 5 == 4
CLIP_NORM = 0.25              # steps/train_qsub.py:121


def get_args(argv=None):
  parser = argparse.ArgumentParser(description="""This script trains a separation neural network""")
  parser.add_argument("arch_file", metavar="arch-file", type=str, help="DNN architecture file")
  parser.add_argument("gpu_id", metavar="gpu-id", type=int, help="GPU ID")
  parser.add_argument("data_dir", metavar="data-dir", type=str, help="Training data directory")
  parser.add_argument("dirout", type=str, help="Output directory")
  parser.add_argument("--cv-data-dir", type=str, help="Cross validation data directory", default="")
  parser.add_argument("--train-copy-location", type=str, help="Copy training data here for I/O purposes", default="")
  parser.add_argument("--model-config", type=str, help="Config file for DNN", default="")
  parser.add_argument("--batch-size", type=int, help="Batch size (per GPU)", default=100)
  parser.add_argument("--start-epoch", type=int, help="Epoch to start from", default=0)
  parser.add_argument("--num-epochs", type=int, help="Total number of training epochs", default=200)
  parser.add_argument("--learning-rate", type=float, help="Learning rate", default=0.001)
  parser.add_argument("--torch-optimizer", action="store_true",
                      help="use torch clip_grad_norm_ + optim.Adam instead of the fused kernel")
  parser.add_argument("--num-workers", type=int, default=None,
                      help="loader processes (default: up to 12 for npz features -- zlib inflate of ~1 MB per utterance is the "
                           "loader's cost: 8 feed a 35 ms step, the 13 ms bf16 step wants more --, 4 with --wav-input (reading 96 small wav files "
                           "per batch: ~15 ms per worker); capped by "
                           "the CPUs this process may use)")
  parser.add_argument("--prefetch", type=int, default=2,
                      help="batches staged on the GPU ahead of the step (pinned memory, own copy stream); 0: off")
  parser.add_argument("--wav-input", action="store_true",
                      help="read <data-dir>/wav.scp and compute the STFT features on the GPU inside the step "
                           "(arch must provide WavTrainSet) instead of loading feats_train.scp npz files")
  parser.add_argument("--seed", type=int, default=None, help="seed for weights, shuffling and h0/c0")
  return parser.parse_args(argv)


# ----------------------------------------------------------------------------------------------- files
class RunDir:
  """The files of one training run (layout of the reference, steps/train_qsub.py:73-76)."""

  def __init__(self, dirout, writer):
    self.models = dirout + '/intermediate_models/'
    self.plots = dirout + '/train_stats/plots/'
    self.final = dirout + '/final.mdl'
    self.writer = writer                      # only rank 0 touches the disk
    self.logs = {"train": LossLog(dirout + '/train_stats/train_loss.txt'),
                 "cv": LossLog(dirout + '/train_stats/cv_loss.txt')}
    if writer:
      os.makedirs(self.models, exist_ok=True)
      os.makedirs(self.plots, exist_ok=True)

  def epoch_tag(self, epoch_number):
    return str(epoch_number).zfill(3)

  def model_file(self, epoch_number):
    return self.models + self.epoch_tag(epoch_number) + '.mdl'

  def optimizer_file(self, epoch_number):
    return self.models + self.epoch_tag(epoch_number) + '.opt'


class LossLog:
  """train_loss.txt / cv_loss.txt: one line "EEE <loss>" per epoch (steps/train_qsub.py:138,144); the history is
  kept as the [[epochs], [losses]] pair tools/plot.py:plot_loss takes."""

  def __init__(self, path):
    self.path, self.history, self.handle = path, [[], []], None

  def reload(self):
    with open(self.path) as f:
      for line in f:
        fields = line.split()
        if len(fields) >= 2:
          self.history[0].append(int(fields[0]))
          self.history[1].append(float(fields[1]))      # the reference's np.float (:60) is gone from numpy

  def record(self, epoch_number, value, write):
    self.history[0].append(epoch_number)
    self.history[1].append(value)
    if write:
      if self.handle is None:
        self.handle = open(self.path, 'a')
      self.handle.write(str(epoch_number).zfill(3) + ' ' + str(value) + '\n')
      self.handle.flush()


def read_model_conf(path):
  """key=value per line; values stay strings, as SepDNN(**kwargs) expects (steps/train_qsub.py:87-91)."""
  conf = {}
  if path:
    with open(path) as f:
      for line in f:
        if '=' in line:
          key, value = line.rstrip().split('=', 1)
          conf[key] = value
  return conf


# ----------------------------------------------------------------------------------------------- data
def training_batches(m, args, rank, world):
  """DataLoader over this rank's share of the training set, and the sampler to re-seed per epoch (or None).
  One process: the reference's shuffled loader (steps/train_qsub.py:80-81).  Several: EpochShards."""
  from sepkern import dist as skdist
  dataset = m.WavTrainSet(args.data_dir) if args.wav_input else m.TrainSet(args.data_dir, args.train_copy_location)
  seed = args.seed if args.seed is not None else 0
  workers = loader_workers(args, world)
  extra = dict(persistent_workers=True, prefetch_factor=2) if workers > 0 else {}      # workers live across epochs
  if world == 1 and args.seed is None:           # the reference's shuffled loader, order from the global RNG
    loader = DataLoader(dataset, batch_size=args.batch_size, shuffle=True, collate_fn=dataset.collator,
                        num_workers=workers, **extra)
    return staged(loader, args), None
  # with --seed (or several ranks) the order of an epoch is a function of (seed, epoch) alone -- EpochShards, whatever the
  # loader draws from its generator for itself (a persistent-worker loader draws a base seed on its first epoch only), so
  # that `--start-epoch N` continues exactly where an uninterrupted run would be
  counts = dataset.frame_counts() if (world > 1 and hasattr(dataset, "frame_counts")) else None
  shards = skdist.EpochShards(len(dataset), args.batch_size, rank, world, lengths=counts, seed=seed)
  loader = DataLoader(dataset, batch_sampler=shards, collate_fn=dataset.collator, num_workers=workers, **extra)
  return staged(loader, args), shards


def loader_workers(args, world):
  if args.num_workers is not None:
    return max(0, args.num_workers)
  try:
    cpus = len(os.sched_getaffinity(0))
  except AttributeError:
    cpus = os.cpu_count() or 1
  return max(1, min(4 if args.wav_input else 12, cpus // max(1, world) - 3))


def staged(loader, args):
  """The loader's batches staged on the GPU ahead of their step (--prefetch N > 0), or the loader as it is."""
  if args.prefetch <= 0:
    return loader
  from sepkern.data import Prefetcher
  return Prefetcher(loader, torch.device("cuda", torch.cuda.current_device()), depth=args.prefetch)


def validation_batches(m, args, rank, world):
  from sepkern import dist as skdist
  if not args.cv_data_dir:
    return None
  dataset = m.WavTrainSet(args.cv_data_dir) if args.wav_input else m.TrainSet(args.cv_data_dir)
  part = dataset if world == 1 else Subset(dataset, skdist.shard_indices_contiguous(len(dataset), rank, world))
  if len(part) == 0:
    return []                                    # more ranks than utterances: this rank only joins the final sum
  return DataLoader(part, batch_size=args.batch_size, collate_fn=dataset.collator)


# ----------------------------------------------------------------------------------------------- model
def build_model(m, args, gpu, rank):
  """SepDNN from the conf file on the GPU, identical on every rank, with its optimizer."""
  from sepkern import dist as skdist
  from sepkern.optim import ClipAdam
  if args.seed is not None:
    torch.manual_seed(args.seed)
  model = m.SepDNN(gpu, **read_model_conf(args.model_config))
  model.cuda()
  if hasattr(model, "flat_parameters"):
    model.flat_parameters()                      # bind the flat buffers before anything is broadcast or optimised
  skdist.broadcast_model(model)                  # whatever RNG state each rank had
  if args.seed is not None:
    model.hidden_generator = torch.Generator(device="cuda")
    model.hidden_generator.manual_seed(args.seed + 7919 * rank)       # per-rank h0/c0 stream
  if args.torch_optimizer:
    optimizer = torch.optim.Adam(model.parameters(), lr=args.learning_rate)
  else:
    optimizer = ClipAdam(model, lr=args.learning_rate, max_norm=CLIP_NORM)
  return model, optimizer


def reseed_epoch(args, rank, epoch, order, model):
  """With --seed, everything random in an epoch (utterance order, the per-batch h0/c0 of archs/uPIT.py:121-127) is a
  function of (seed, rank, epoch) alone, so `--start-epoch N` continues exactly where an uninterrupted run would be."""
  if hasattr(order, "set_epoch"):
    order.set_epoch(epoch)                       # sepkern.dist.EpochShards
  elif isinstance(order, torch.Generator):
    order.manual_seed(args.seed * 1000003 + epoch)
  if args.seed is not None and getattr(model, "hidden_generator", None) is not None:
    model.hidden_generator.manual_seed(args.seed * 1000003 + 7919 * rank + 104729 * (epoch + 1))


def resume(model, optimizer, run, args):
  model.load_state_dict(torch.load(run.model_file(args.start_epoch), map_location=lambda storage, loc: storage.cuda()))
  if os.path.isfile(run.optimizer_file(args.start_epoch)):
    optimizer.load_state_dict(torch.load(run.optimizer_file(args.start_epoch),
                                         map_location=lambda storage, loc: storage.cuda()))
  run.logs["train"].reload()
  if args.cv_data_dir:
    run.logs["cv"].reload()


# ----------------------------------------------------------------------------------------------- passes
POLL_EVERY = 50      # steps between two reads of the optimizer's skipped-step counter (one small host sync each)


def rank_of():
  return torch.distributed.get_rank() if (torch.distributed.is_available() and torch.distributed.is_initialized()) else 0


def recover_from_timeout(model, optimizer, newly_skipped, epoch):
  """A persistent recurrence launch timed out (its bounded wait gave up: the grid was not co-resident).  The device
  already kept that step away from the weights on EVERY rank (the guard word is all-reduced with the gradients, so the
  skipped count is the same everywhere; train_epoch reads it at the same steps on every rank, so all ranks take this branch
  together).  Clear the sticky word, continue IN
  THIS PROCESS with one launch per step (never re-exec a process that holds the GPU), and say so."""
  try:
    model.check_status()                 # reads and clears the workspace's sticky word
  except Exception as e:                 # SepkernError(SK_ETIMEOUT): expected here
    print("train: %s" % e, file=sys.stderr)
  eng = getattr(model, "_engine", None)
  if eng is None or eng.lstm_mode == 2:
    raise RuntimeError("the recurrence failed in per-step launch mode too (%d steps skipped in epoch %d)"
                       % (newly_skipped, epoch + 1))
  eng.lstm_mode = 2
  print("train: epoch %d: %d step(s) skipped after a timed-out persistent recurrence launch; their loss terms are dropped, "
        "continuing with one launch per step (SEPKERN_LSTM_MODE=2)" % (epoch + 1, newly_skipped), file=sys.stderr)
  sys.stderr.flush()


def train_epoch(m, model, optimizer, batches, epoch, world, torch_clip):
  """One pass over this rank's batches.  Returns the device tensor [sum(loss * norm), sum(norm)] of the GLOBAL
  epoch (all-reduced), still un-synchronised.  Every POLL_EVERY steps (and at the end) the fused optimizer's
  skipped-step counter is read: the loss terms of a window that contains a skipped step (garbage or NaN forwards) are
  dropped instead of poisoning the epoch's sums, and the run continues in per-step launch mode."""
  dev = next(model.parameters()).device
  # [sum(loss * norm), sum(norm), frames] of the steps that reached the weights; `win` = the same for the open window
  acc = torch.zeros(3, device=dev, dtype=torch.float64)
  win = torch.zeros(3, device=dev, dtype=torch.float64)
  fused = hasattr(optimizer, "skipped")
  seen = optimizer.skipped() if fused else 0

  def close_window():
    nonlocal seen, win
    if fused:
      now = optimizer.skipped()
      if now > seen:
        recover_from_timeout(model, optimizer, now - seen, epoch)
        seen = now
        win = torch.zeros_like(win)
        return
    acc.add_(win)
    win = torch.zeros_like(win)

  t_epoch = time.perf_counter()
  steps = 0
  for batch in batches:
    steps += 1
    loss, norm = m.compute_loss(model, epoch, batch)
    win[0] += loss.detach().double() * norm.double()
    # under data parallelism `norm` is already the global frame count: every rank adds its 1/world share
    win[1] += norm.double() / world
    win[2] += norm.double() / world
    loss.backward()
    if torch_clip:
      # torch's optimizer knows nothing of the guard word: look at it here (one host sync per step on this
      # compatibility path; it is all-reduced with the gradients, so every rank decides alike)
      eng = getattr(model, "_engine", None)
      if eng is not None and float(eng.guard.item()) != 0.0:
        recover_from_timeout(model, optimizer, 1, epoch)
        win = torch.zeros_like(win)
        continue
      torch.nn.utils.clip_grad_norm_(model.parameters(), CLIP_NORM)
    optimizer.step()
    # One process: the counter's asynchronous host copy, one step behind -- after a timed-out launch the run switches to
    # per-step launches within a step or two instead of skipping every step up to the next poll.  Several ranks: WHEN that
    # copy lands differs from rank to rank, so only the poll at a fixed step -- whose count is the same everywhere, the
    # guard word being all-reduced with the gradients -- may close a window: all ranks then drop the same steps' loss
    # terms and switch the recurrence mode at the same step.
    if steps % POLL_EVERY == 0 or (fused and world == 1 and optimizer.skipped_nowait() > seen):
      close_window()
  close_window()
  if world > 1:
    torch.distributed.all_reduce(acc)
  # wall time and frames/s of the epoch as the user sees it (data loading included): stderr, the reference's stdout
  # lines stay as they are.  norm = frames x feat_dim (archs/uPIT.py:197); frames of windows that were dropped after a
  # timed-out launch are not counted.  (One host sync per epoch, here.)  An epoch without a batch (an empty shard, a
  # resume on a tiny set) reports zero steps.
  n_frames = float(acc[2].item()) / float(getattr(model, "feat_dim", 257))
  dt = time.perf_counter() - t_epoch
  if rank_of() == 0:
    print("train: epoch %d: %d steps, %.0f frames in %.2f s = %.0f frames/s"
          % (epoch + 1, steps, n_frames, dt, n_frames / dt if dt > 0 else 0.0), file=sys.stderr, flush=True)
  return acc[:2]


def validation_pass(m, model, batches, epoch, world, plot_dir):
  """Eval-mode pass over this rank's shard of the CV set; no collective inside (the arch keeps its local norm
  when not training).  Returns the global [sum(loss * norm), sum(norm)].  The recurrence's sticky status word rides
  with the sums, so that a timed-out launch on one rank is raised by ALL ranks together (a raise on one rank alone
  would leave the others waiting in this all-reduce)."""
  acc = torch.zeros(3, device=next(model.parameters()).device, dtype=torch.float64)
  model.eval()
  with torch.no_grad():
    for i, batch in enumerate(batches):
      where = plot_dir if (i == 0 and plot_dir) else ""
      loss, norm = m.compute_cv_loss(model, epoch, batch, where) if where else m.compute_cv_loss(model, epoch, batch)
      acc[0] += loss.detach().double() * norm.double()
      acc[1] += norm.double()
  model.train()
  eng = getattr(model, "_engine", None)
  if eng is not None:
    acc[2] = (eng.sticky() != 0).double().sum()
  if world > 1:
    torch.distributed.all_reduce(acc)
  if float(acc[2]) != 0.0:
    try:
      model.check_status()               # clears this rank's word
    except Exception:
      pass
    raise RuntimeError("validation pass of epoch %d: a persistent recurrence launch timed out on %d rank(s)"
                       % (epoch + 1, int(acc[2])))
  return acc[:2]


def report_failures(model, optimizer, world):
  """Epoch boundary: surface a timed-out recurrence launch that train_epoch's recovery did not absorb (host sync).  The
  verdict is all-reduced first: every rank raises, or none does."""
  bad = 0
  if hasattr(optimizer, "skipped") and getattr(model, "_engine", None) is not None:
    bad = int(model._engine.sticky().item() != 0)
  if world > 1:
    t = torch.tensor([float(bad)], device=next(model.parameters()).device)
    torch.distributed.all_reduce(t)
    bad = int(t.item() != 0)
  if bad:
    try:
      model.check_status()
    except Exception:
      pass
    raise RuntimeError("a persistent recurrence launch timed out after the fallback to per-step launches")


def write_checkpoint(model, optimizer, run, epoch_number):
  torch.save(model.state_dict(), run.model_file(epoch_number))
  torch.save(optimizer.state_dict(), run.optimizer_file(epoch_number))
  draw_losses(run, run.plots + 'epoch' + run.epoch_tag(epoch_number) + '/', epoch_number)


def draw_losses(run, where, last_epoch_number):
  try:
    import plot
  except ImportError:
    return
  train, cv = run.logs["train"].history, run.logs["cv"].history
  if not train[0]:
    return
  os.makedirs(where, exist_ok=True)
  plot.plot_loss(train, cv, where + 'Loss_' + run.epoch_tag(train[0][0]) + '-' + run.epoch_tag(last_epoch_number) + '.png')


# ----------------------------------------------------------------------------------------------- main
def main(argv=None):
  args = get_args(argv)
  from sepkern import dist as skdist
  rank, world, local = skdist.init_from_env()
  gpu = local if world > 1 else args.gpu_id
  chief = rank == 0
  if chief:
    print("Using " + args.arch_file + " DNN architecture")
    print("Using GPU", gpu, "of", world)
  m = __import__(args.arch_file)
  torch.cuda.set_device(gpu)
  from sepkern.data import host_threads
  host_threads()                                 # (the arithmetic is on the GPU; see sepkern.data.host_threads)

  run = RunDir(args.dirout, chief)
  train_batches, shards = training_batches(m, args, rank, world)
  cv_batches = validation_batches(m, args, rank, world)
  model, optimizer = build_model(m, args, gpu, rank)
  if chief:
    print("using lr=" + str(args.learning_rate))

  if args.start_epoch == 0:
    if chief:
      torch.save(model.state_dict(), run.models + 'init.mdl')
  else:
    resume(model, optimizer, run, args)

  for epoch in range(args.start_epoch, args.num_epochs):
    number = epoch + 1
    reseed_epoch(args, rank, epoch, shards, model)
    acc = train_epoch(m, model, optimizer, train_batches, epoch, world, args.torch_optimizer)
    # one set of BatchNorm running statistics on every rank before anything is scored or saved (the reference has one
    # model: what scores the CV set is what the checkpoint holds, steps/train_qsub.py:124-152)
    skdist.average_bn_buffers(model)
    checkpoint = epoch % CHECKPOINT_EVERY == CHECKPOINT_EVERY - 1
    if cv_batches is not None and checkpoint:
      cv = validation_pass(m, model, cv_batches, epoch, world,
                           run.plots + 'epoch' + run.epoch_tag(number) if chief else "")
      cv_value = float(cv[0] / cv[1])
      if chief:
        print("For epoch: " + run.epoch_tag(number) + " cv set loss is: " + str(cv_value))
      run.logs["cv"].record(number, cv_value, chief)
    value = float(acc[0] / acc[1])                  # the epoch's one host sync
    report_failures(model, optimizer, world)
    if chief:
      print("For epoch: " + run.epoch_tag(number) + " loss is: " + str(value))
    run.logs["train"].record(number, value, chief)
    if checkpoint and chief:
      print("Saving model for epoch " + run.epoch_tag(number))
      write_checkpoint(model, optimizer, run, number)
    sys.stdout.flush()

  if chief:
    torch.save(model.state_dict(), run.final)
    draw_losses(run, run.plots, args.num_epochs)
  if world > 1:
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


if __name__ == '__main__':
  main()
