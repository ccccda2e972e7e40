#!/usr/bin/env python3
"""uPIT arch plug-in, MI355X-native: drop-in for the reference's archs/uPIT.py.

Same module-level protocol (looked up by name from steps/train_qsub.py:66,80-95,117-135 and
steps/eval_qsub.py:43-72 of the reference):

    TrainSet(datadir, location="")   .collator      reads <datadir>/feats_train.scp
    TestSet(datadir)                 .collator      reads <datadir>/feats_test.scp
    SepDNN(gpuid, **kwargs)          nn.Module; kwargs are strings from the conf file
    compute_loss(model, epoch, batch_sample, plotdir="")     -> (loss/norm, norm) 0-dim tensors
    compute_cv_loss(model, epoch, batch_sample, plotdir="")  -> same
    compute_masks(model, batch_sample, out_dir)              -> writes <out_dir>/<id>.npz

All numerics run in libsepkern.so (hand-written gfx950 kernels, see include/sepkern.h) through
sepkern.engine; there is NO CPU path -- SepDNN(-1) or a missing library raises.

Beyond the reference (all optional, defaults reproduce it):
  * conf keys hidden_dim (600) and num_layers (2): the reference hard-codes 2x600
    (archs/uPIT.py:115-119); BASELINE's 2x300 / 3x896 configurations need them.
  * model.hidden_generator: a torch.Generator for the randn h0/c0 the reference draws per batch
    (archs/uPIT.py:121-127); tests inject (h0, c0) by assigning model.next_hidden.
  * under torch.distributed (one process per GPU) gradients are all-reduced over RCCL inside
    backward and the loss is normalised by the GLOBAL frame count.
"""
import itertools
import os
import sys

import numpy as np
import torch
import torch.nn as nn
from torch.utils.data import Dataset
from torch.utils.data.dataloader import default_collate

try:
    import sepkern  # noqa: F401
except ImportError:  # the frozen copy exp/<...>/arch.py is imported from another directory
    for cand in (os.environ.get("SEPKERN_HOME"), "speech-separation_amd",
                 os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")):
        if cand and os.path.isdir(os.path.join(cand, "sepkern")):
            sys.path.insert(0, os.path.abspath(cand))
            break
    import sepkern  # noqa: F401
from sepkern import dist as skdist
from sepkern import ops
from sepkern.data import features_from_pcm as _features_from_pcm
from sepkern.collate import collate_sorted, eval_magnitudes, read_scp, stage_copies, train_sample
from sepkern.model import SepDNNBase, UnpackFn, to_packed as _to_packed
from sepkern.packing import Packing
from sepkern._lib import SepkernError


class Collator():
  """Same contract as the reference Collator (archs/uPIT.py:23-48): dict samples are sorted by the length of `sort_key`
  (descending, via argsort()[::-1]) and every ndarray entry becomes a float32 PackedSequence; everything else goes
  through default_collate (sepkern.collate)."""

  def __init__(self, sort_key):
    self.key = sort_key
    if not sort_key:       # (the text a user of the reference sees)
      print("Warning: you have not provided a sort key.")
      print("  If you are using RNNs with variable-length input, you must")
      print("  provide the key for element in each sample that is the input")
      print("  of variable length.")

  def __call__(self, batch):
    return collate_sorted(batch, self.key) if self.key else default_collate(batch)


class TrainSet(Dataset):
  """feats_train.scp -> {'mix': (T,F), 'source1': (T,F), ...} (reference archs/uPIT.py:51-79)."""

  def __init__(self, datadir, location=""):
    self.list = read_scp(datadir + "/feats_train.scp")
    if location:
      self.list = stage_copies(self.list, location)
    self.collator = Collator('mix')

  def __len__(self):
    return len(self.list)

  def frame_counts(self):
    """Frames per utterance, from the npz headers only (length-balanced sharding across GPUs)."""
    from sepkern.data import npz_frames
    return [npz_frames(path) for path in self.list]

  def __getitem__(self, idx):
    return train_sample(self.list[idx])


class TestSet(Dataset):
  """feats_test.scp -> {'mix': |complex STFT| (T,F), 'name': '<id>.npz'} (reference archs/uPIT.py:81-94)."""

  def __init__(self, datadir):
    self.list = read_scp(datadir + "/feats_test.scp")
    self.collator = Collator('mix')

  def __len__(self):
    return len(self.list)

  def __getitem__(self, idx):
    mags, name = eval_magnitudes(self.list[idx])
    return {'mix': mags, 'name': name}


class WavTrainSet(Dataset):
  """On-GPU input pipeline: reads <datadir>/wav.scp and yields the int16 PCM of the mixture
  and its sources (found like steps/extract_feats.py:65 does, by globbing /mix/ -> /*/); the STFT that stage 1
  of the recipe would have stored as npz is computed on the GPU inside compute_loss (sk_stft straight into the
  padded (T,B,F) batch).  12x less host I/O than float32 spectrograms and no zlib in the loader."""

  def __init__(self, datadir, location=""):
    import glob
    self.items = []
    for line in open(datadir + "/wav.scp"):
      reco_id, filename = line.rstrip('\n').split(' ')
      self.items.append(sorted(glob.glob(filename.replace("/mix/", "/*/"))))
    self.collator = WavCollator()

  def __len__(self):
    return len(self.items)

  def frame_counts(self):
    from sepkern.data import wav_frames
    return [wav_frames(files[0]) for files in self.items]

  def __getitem__(self, idx):
    import scipy.io.wavfile
    out = {}
    for i, f in enumerate(self.items[idx]):
      fs, x = scipy.io.wavfile.read(f)
      if x.dtype != np.int16 or x.ndim != 1:
        raise ValueError("%s: only mono 16-bit PCM wav is supported" % f)
      out['mix' if i == 0 else 'source' + str(i)] = x
    if len(out) == 1:
      out['source1'] = out['mix']
    return out


class WavCollator():
  """Sorts by frame count (descending, as Collator does) and hands the batch over as ONE int16 tensor: the signals of all
  utterances, key-major ('mix', 'source1', ...), longest utterance first -- {'pcm': {'flat', 'keys', 'lens'}}.  One tensor
  crosses from the loader's worker process to the trainer instead of 32 x (S + 1) (each of which costs a shared-memory
  hand-over: measured 15-18 ms per batch whatever the worker count, more than a 14 ms bf16 step)."""

  def __call__(self, batch):
    order = np.argsort(np.array([1 + len(d['mix']) // 128 for d in batch]))[::-1]
    others = [k for k in batch[0] if k != 'mix']
    bad = [k for k in others if not (k.startswith('source') and k[6:].isdigit())]
    if bad:
      raise ValueError("WavCollator: signals besides 'mix' must be named 'source<N>' (got %r)" % bad)
    keys = ['mix'] + sorted(others, key=lambda k: int(k[6:]))
    # ONE list of sample counts ('lens', the mixture's) describes every key of the flat tensor: a source whose length differs
    # from its mixture's by a single sample would shift every later signal -- wrong targets, no error -- so it is one here
    for d in batch:
      for k in keys:
        if k not in d or len(d[k]) != len(d['mix']):
          raise ValueError("WavCollator: signal %r of an utterance has %s samples, its mixture %d -- the sources of a mixture "
                           "must have the mixture's length" % (k, len(d[k]) if k in d else "no", len(d['mix'])))
    flat = np.concatenate([batch[i][k] for k in keys for i in order])
    return {'pcm': {'flat': torch.from_numpy(flat), 'keys': keys, 'lens': [int(len(batch[i]['mix'])) for i in order]}}


class _PitFn(torch.autograd.Function):
  """out = [loss/norm, norm, sum_b min_p L/S] (reference archs/uPIT.py:181-197,206)."""

  @staticmethod
  def forward(ctx, mask, mix, pk, norm_dev, *srcs):
    # mask (R, S*F), mix / srcs (R, F): packed rows of the batch pk (PackedSequence.data, archs/uPIT.py:160-167)
    res = ops.pit_mse_fwd(mask, mix, list(srcs), None, norm_dev, packing=pk)
    ctx.save_for_backward(mask, mix, res["best_perm"], res["out"], *srcs)
    ctx.pk = pk
    ctx.mark_non_differentiable(res["best_perm"])
    return res["out"], res["best_perm"]

  @staticmethod
  def backward(ctx, gout, _gperm):
    mask, mix, best, out = ctx.saved_tensors[:4]
    srcs = list(ctx.saved_tensors[4:])
    dmask = ops.pit_mse_bwd(mask, mix, srcs, best, out, gout[0:1].contiguous(), packing=ctx.pk)
    return (dmask, None, None, None) + (None,) * len(srcs)


class SepDNN(SepDNNBase):
  def __init__(self, gpuid, **kwargs):
    super(SepDNN, self).__init__()
    self.feat_dim = int(kwargs.get('feat_dim', 257))
    self.num_spk = int(kwargs.get('num_spk', 2))
    for key in kwargs.keys():
      print('modelparam:', key, kwargs[key])
    # the reference hard-codes 2 x 600 (archs/uPIT.py:115-119); hidden_dim / num_layers widen it
    self._build(gpuid, self.feat_dim, self.feat_dim * self.num_spk, int(kwargs.get('hidden_dim', 600)),
                int(kwargs.get('num_layers', 2)), str(kwargs.get('dtype', 'fp32')), kwargs.get('sync_bn', '0'))

  def forward_packed(self, x2d, pk):
    """x2d (R,F) packed rows of the batch pk (PackedSequence.data on the GPU) -> mask (R,F*S) packed."""
    if self.hidden is None:
      self.hidden = self.init_hidden(pk.B)
    h0, c0 = self.hidden
    return self.run_net_packed(x2d, pk, h0, c0)

  def forward_padded(self, x, lens):
    """x (T,B,F) time-major zero-padded CUDA tensor, lens int32 CUDA (B) -> mask (T,B,F*S)."""
    if self.hidden is None:
      self.hidden = self.init_hidden(x.shape[1])
    h0, c0 = self.hidden
    return self.run_net(x, lens, h0, c0)

  def forward(self, x):
    # x: packed sequence of dim feat_dim  ->  tensor of shape (batch, seq_length, feat_dim*num_spk)
    # RESTRICTION (variable-length batches, gradients): the values at zero-padded frames are filled in as the constant the
    # reference's network shows there, sigmoid(lin(bn(0))), and carry NO gradient back into lin / bn (UnpackFn).  A loss that
    # zeroes padded frames -- the reference's PIT-MSE does: mix = 0 there, archs/uPIT.py:181-197 -- gets the reference's parameter
    # gradients exactly; a loss that reads the padded frames of model(x) would miss their contribution to d lin / d bn.
    x2d, pk = _to_packed(x, self.lin.weight.device)
    mask = self.forward_packed(x2d, pk)
    # (padded frames: the constant the reference's BatchNorm / Linear / sigmoid produce there, archs/uPIT.py:135-144)
    fill = None if pk.uniform else self._engine.pad_row()
    if mask.requires_grad and not pk.uniform:       # (uniform: unpack is a view, differentiable as it is)
      padded = UnpackFn.apply(mask, pk, fill)
    else:
      padded = pk.unpack(mask, fill=fill)
    # batch-first AND contiguous, as the reference's Linear + sigmoid output is: its own loss code takes .view(batch, -1) of an
    # elementwise result of this tensor (archs/uPIT.py:192), which a permuted view would refuse
    return padded.permute(1, 0, 2).contiguous()


def compute_cv_loss(model, epoch, batch_sample, plotdir=""):
  if plotdir:
    loss, norm = compute_loss(model, epoch, batch_sample, plotdir)
  else:
    loss, norm = compute_loss(model, epoch, batch_sample)
  return loss, norm


def compute_loss_packed(model, mix, sources, pk, plotdir=""):
  """compute_loss on inputs that are already resident on the GPU as PACKED rows: mix (R,F) and sources [(R,F)]*S
  float32 -- PackedSequence.data of the collator's batch -- and their Packing pk.  Same return as compute_loss."""
  batch = pk.B
  model.zero_grad()
  model.hidden = model.init_hidden(batch)

  # data-parallel: divide by the GLOBAL frame count so that the summed gradients equal the
  # single-device gradient of the global batch (None = single process, kernel uses sum(lens)*F)
  # -- only while training: the CV pass runs the whole (unsharded) set on every rank, local norm.
  training_step = model.training and torch.is_grad_enabled()
  norm_override = skdist.global_norm(pk.lens, model.feat_dim) if training_step else None

  mask_out = model.forward_packed(mix, pk)
  # mask_out: tensor of shape (sum of lengths, feat_dim*num_spk)
  out, best = _PitFn.apply(mask_out, mix, pk, norm_override, *sources)
  loss, norm = out[0], out[1].detach()
  model.last_best_perm = best.detach()      # arg-min permutation per utterance (index into itertools.permutations)

  if plotdir:
    sys.path.append('tools')
    import plot
    os.system("mkdir -p " + plotdir)
    F = model.feat_dim
    rows0 = pk.offs[:int(pk.lens_host[0])].long()              # the rows of the batch's first (longest) utterance
    m0, x0 = mask_out.detach()[rows0], mix[rows0]
    masked = (m0.view(-1, model.num_spk, F) * x0.unsqueeze(1)).reshape(-1, model.num_spk * F)
    plot.plot_spec(x0.cpu().numpy(), plotdir + '/Mixture.png')
    plot.plot_spec(masked.cpu().numpy(), plotdir + '/Masked_Mixture.png')
    permutation = list(itertools.permutations(range(model.num_spk)))[int(best[0])]
    plot.plot_spec(torch.cat([sources[i][rows0] for i in permutation], dim=1).cpu().numpy(),
                   plotdir + '/Chosen_Permutation.png')

  return loss, norm


def compute_loss_padded(model, mix, sources, lens, plotdir=""):
  """compute_loss on zero-padded time-major inputs resident on the GPU: mix (T,B,F), sources [(T,B,F)]*S float32,
  lens int32 (B) in any order.  They are packed (sk_pack_rows; a view when all lengths are equal) and go the packed way."""
  pk = model.packing_of(lens)
  if pk.perm is not None:
    # not length-sorted (the collator's batches are): the rows are packed in sorted order, so the initial state drawn for
    # this batch follows, and the chosen permutations are handed back in the caller's order
    h, c = model.init_hidden(pk.B)
    pair = (pk.sort_batch(h, 1), pk.sort_batch(c, 1))
    rest = model.next_hidden
    model.next_hidden = pair if rest is None else [pair] + (rest if isinstance(rest, list) else [rest])
  out = compute_loss_packed(model, pk.pack(mix), [pk.pack(s) for s in sources], pk, plotdir)
  if pk.perm is not None:
    model.last_best_perm = pk.unsort_batch(model.last_best_perm, 0)
  return out


def compute_loss(model, epoch, batch_sample, plotdir=""):
  dev = model.lin.weight.device
  if 'pcm' in batch_sample:        # WavTrainSet batches: features are computed on the GPU
    mix, sources, pk = _features_from_pcm(batch_sample['pcm'], dev)
    return compute_loss_packed(model, mix, sources[:model.num_spk], pk, plotdir)
  if 'packed' in batch_sample:     # batches staged on the GPU ahead of the step (sepkern.data.Prefetcher)
    mix, sources, pk = batch_sample['packed']
    return compute_loss_packed(model, mix, sources[:model.num_spk], pk, plotdir)
  mix, pk = _to_packed(batch_sample['mix'], dev)
  sources = [_to_packed(batch_sample['source' + str(i + 1)], dev)[0] for i in range(model.num_spk)]
  return compute_loss_packed(model, mix, sources, pk, plotdir)


def estimate_masks(model, batch_sample):
  """The arithmetic half of compute_masks: [(file name, {'s1': (257,T_i) float32, ...}), ...] for one batch, without
  touching the disk (steps/eval_qsub.py overlaps the zlib compression of one batch with the next batch's GPU work)."""
  dev = model.lin.weight.device
  mix, pk = _to_packed(batch_sample['mix'], dev)
  name = batch_sample['name']
  batch = pk.B

  model.zero_grad()
  model.hidden = model.init_hidden(batch)

  with torch.no_grad():
    mask_np = model.forward_packed(mix, pk).cpu().numpy()            # (sum of lengths, feat_dim*num_spk)
  lens, offs = pk.lens_host, pk.offs_host
  out = []
  for i in range(len(name)):
    mask = mask_np[offs[:lens[i]] + i].transpose()                  # rows (t, i), t < lens[i]  ->  (feat_dim*num_spk, T_i)
    file_dict = dict()
    for src in range(model.num_spk):
      file_dict['s' + str(src + 1)] = mask[src * model.feat_dim:(src + 1) * model.feat_dim]
    out.append((name[i], file_dict))
  return out


def compute_masks(model, batch_sample, out_dir):
  for name, file_dict in estimate_masks(model, batch_sample):
    np.savez_compressed(out_dir + '/' + name, **file_dict)
