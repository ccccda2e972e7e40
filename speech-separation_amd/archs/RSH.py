#!/usr/bin/env python3
"""Recurrent-selective-hearing arch plug-in, MI355X-native: drop-in for the reference's archs/RSH.py.

Same module-level protocol as archs/uPIT.py (TrainSet, TestSet, .collator, SepDNN, compute_loss,
compute_cv_loss, compute_masks).  One mask is estimated per pass of a BLSTM over [mixture | attention]
(2F inputs, F outputs); the attention is reduced by each estimated mask and the LSTM state is carried from
pass to pass (reference archs/RSH.py:172,209), so the backward pass chains through both -- which is why the
BLSTM kernels take dhn/dcn and return dx, dh0, dc0 (sepkern.model.NetFn).  Per-pass loss is the greedy
not-yet-used-source minimum (archs/RSH.py:225-244, sk_rsh_loss_fwd/bwd); the attention update is
relu(att - mask) in training and att - mask at test time (archs/RSH.py:254-257, 278-281, sk_att_update).
Conf keys beyond the reference: hidden_dim (600), num_layers (2).  There is no CPU path.
"""
import os
import sys

import numpy as np
import torch
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
from sepkern.collate import collate_sorted, eval_magnitudes, read_scp, stage_copies, train_sample
from sepkern._lib import SepkernError
from sepkern.model import SepDNNBase, to_padded as _to_padded


class MultiSpkBatch():
  """sub_batches[k]: the samples that have k sources, collated like a uPIT batch (archs/RSH.py:70-85)."""

  def __init__(self, max_spk, length):
    self.sub_batches = list()
    self.sub_batch_lens = list()
    self.max_spk = max_spk
    self.length = length

  def __len__(self):
    return self.length

  def __getitem__(self, idx):
    return self.sub_batches[idx]

  def append(self, elem):
    self.sub_batches.append(elem)


class Collator():
  """Groups a batch by speaker count ('num_spk' entry at test time, number of source keys otherwise) and
  collates every group as archs/uPIT.py does (reference archs/RSH.py:21-68)."""

  def __init__(self, sort_key):
    self.key = sort_key

  def __call__(self, batch):
    if not self.key:
      return default_collate(batch)
    if "num_spk" in batch[0].keys():
      counts = [int(d["num_spk"]) for d in batch]
    else:
      counts = [len(d.keys()) - 1 for d in batch]
    max_spk = max(counts)
    batch_out = MultiSpkBatch(max_spk + 1, len(batch))
    for num_spk in range(max_spk + 1):
      inds = [i for i in range(len(batch)) if counts[i] == num_spk]
      batch_out.sub_batch_lens.append(len(inds))
      batch_out.append(collate_sorted([batch[i] for i in inds], self.key) if inds else {})
    return batch_out


def _combo(mix_mag_spec):
  # [mixture | attention mask of ones] (archs/RSH.py:104-106)
  return np.concatenate((mix_mag_spec, np.ones(mix_mag_spec.shape)), axis=1)


class TrainSet(Dataset):
  """feats_train.scp -> {'combo': [mixture | attention of ones] (T,2F), 'source1': (T,F), ...} (archs/RSH.py:90-118)."""

  def __init__(self, datadir, location=""):
    self.list = read_scp(datadir + "/feats_train.scp")
    if location:
      self.list = stage_copies(self.list, location)
    self.collator = Collator('combo')

  def __len__(self):
    return len(self.list)

  def frame_counts(self):
    """Frames per utterance, from the npz headers only (length-balanced sharding across GPUs)."""
    from sepkern.data import npz_frames
    return [npz_frames(path) for path in self.list]

  def __getitem__(self, idx):
    return train_sample(self.list[idx], mix_key='combo', mix_map=_combo)


class TestSet(Dataset):
  """feats_test.scp + utt2num_spk -> {'combo', 'name', 'num_spk'} (archs/RSH.py:120-138)."""

  def __init__(self, datadir):
    self.list = read_scp(datadir + "/feats_test.scp")
    self.num_spks = [int(v) for v in read_scp(datadir + "/utt2num_spk")]
    self.collator = Collator('combo')

  def __len__(self):
    return len(self.list)

  def __getitem__(self, idx):
    mags, name = eval_magnitudes(self.list[idx])
    return {'combo': _combo(mags), 'name': name, 'num_spk': self.num_spks[idx]}


class _PassLossFn(torch.autograd.Function):
  """One pass of the greedy assignment loss: out = [sum_b min / num_spk, sum(lens) * F]."""

  @staticmethod
  def forward(ctx, mask, x, lens, used, *srcs):
    res = ops.rsh_loss_fwd(mask, x, list(srcs), lens, used)
    ctx.save_for_backward(mask, x, res["sel"], *srcs)
    ctx.mark_non_differentiable(res["sel"])
    return res["out"], res["sel"]

  @staticmethod
  def backward(ctx, gout, _gsel):
    mask, x, sel = ctx.saved_tensors[:3]
    srcs = list(ctx.saved_tensors[3:])
    dmask = ops.rsh_loss_bwd(mask, x, srcs, sel, gout[0:1].contiguous())
    return (dmask, None, None, None) + (None,) * len(srcs)


class _AttFn(torch.autograd.Function):
  """combos <- act(combos - [0 | mask]); act = relu in training (archs/RSH.py:254-257), identity at test."""

  @staticmethod
  def forward(ctx, x, mask, relu):
    out = ops.att_update(x, mask, relu)
    ctx.save_for_backward(out)
    ctx.relu, ctx.F = relu, mask.shape[-1]
    return out

  @staticmethod
  def backward(ctx, dout):
    (out,) = ctx.saved_tensors
    dx, dmask = ops.att_update_bwd(dout.contiguous(), out, ctx.F, ctx.relu)
    return dx, dmask, None


class SepDNN(SepDNNBase):
  def __init__(self, gpuid, **kwargs):
    super(SepDNN, self).__init__()
    self.feat_dim = int(kwargs.get('feat_dim', 257))
    for key in kwargs.keys():
      print('modelparam:', key, kwargs[key])
    # sync_bn is a uPIT option: here the network runs once per speaker and the ranks of a data-parallel job may hold
    # batches with different speaker counts, so a collective inside the forward pass could not be matched
    if str(kwargs.get('sync_bn', '0')).lower() in ('1', 'true', 'yes'):
      raise SepkernError("RSH does not support sync_bn (ranks may run different numbers of passes)")
    self._build(gpuid, self.feat_dim * 2, self.feat_dim, int(kwargs.get('hidden_dim', 600)),
                int(kwargs.get('num_layers', 2)), str(kwargs.get('dtype', 'fp32')))

  def forward_padded(self, x, lens):
    """x (T,B,2F) -> mask (T,B,F); self.hidden is replaced by the LSTM's final state (archs/RSH.py:172)."""
    h0, c0 = self.hidden
    mask, hn, cn = self.run_net(x, lens, h0, c0, want_state=True)
    self.hidden = (hn, cn)
    return mask

  def forward(self, x):
    xp, lens = _to_padded(x, self.lin.weight.device)
    # (batch-first and contiguous, as the reference's output is: its loss code takes .view(batch, -1), archs/RSH.py:229)
    return self.forward_padded(xp, lens).permute(1, 0, 2).contiguous()


def compute_cv_loss(model, epoch, batch_sample, plotdir=""):
  if plotdir:
    loss, norm = compute_loss(model, epoch, batch_sample, plotdir)
  else:
    loss, norm = compute_loss(model, epoch, batch_sample)
  return loss, norm


def compute_loss_padded(model, groups, plotdir=""):
  """compute_loss on sub-batches already resident on the GPU.  groups: list of (num_spk, combos (T,B,2F),
  sources [(T,B,F)] * num_spk, lens int32 (B)), zero-padded time-major.  Same return as compute_loss."""
  F = model.feat_dim
  model.zero_grad()
  loss = 0
  norm = 0
  for num_spk, combos, sources, lens in groups:
    batch = combos.shape[1]
    model.hidden = model.init_hidden(batch)
    used = torch.zeros(num_spk, batch, dtype=torch.int32, device=combos.device)   # source_usage of archs/RSH.py:218
    for dnn_pass in range(num_spk):
      mask_out = model.forward_padded(combos, lens)
      # mask_out: (seq_length, batch, feat_dim)
      out, sel = _PassLossFn.apply(mask_out, combos, lens, used, *sources)
      loss = loss + out[0]
      norm = norm + out[1].detach()
      if plotdir:
        sys.path.append('tools')
        import plot
        os.system("mkdir -p " + plotdir)
        c0 = combos[:, 0].detach().cpu().numpy()
        prefix = plotdir + '/' + str(num_spk) + '-Spk_Pass-' + str(dnn_pass + 1) + '_'
        if dnn_pass == 0:
          plot.plot_spec(c0[:, 0:F], plotdir + '/' + str(num_spk) + '-Spk_Mix.png')
        plot.plot_spec(c0, prefix + 'Input.png')
        plot.plot_spec(mask_out[:, 0].detach().cpu().numpy(), prefix + 'Mask_Out.png')
        plot.plot_spec((mask_out[:, 0] * combos[:, 0, :F]).detach().cpu().numpy(), prefix + 'Masked_Mix.png')
        plot.plot_spec(sources[int(sel[0])][:, 0].cpu().numpy(), prefix + 'Chosen_Source.png')
      combos = _AttFn.apply(combos, mask_out, True)

  # data-parallel: the global frame count normalises every rank's loss (sepkern.dist) -- training steps only:
  # the CV pass runs the whole, unsharded set on every rank and keeps its local norm
  if skdist.is_parallel() and model.training and torch.is_grad_enabled():
    gn = norm.reshape(1).clone()
    torch.distributed.all_reduce(gn)
    norm = gn[0]
  return loss / norm, norm


def compute_loss(model, epoch, batch_sample, plotdir=""):
  dev = model.lin.weight.device
  groups = []
  for num_spk in range(batch_sample.max_spk):
    if batch_sample.sub_batch_lens[num_spk] > 0:
      combos, lens = _to_padded(batch_sample[num_spk]['combo'], dev)
      # combos: (seq_length, batch, feat_dim*2)
      sources = [_to_padded(batch_sample[num_spk]['source' + str(i + 1)], dev)[0] for i in range(num_spk)]
      groups.append((num_spk, combos, sources, lens))
  return compute_loss_padded(model, groups, plotdir)


def estimate_masks(model, batch_sample):
  """The arithmetic half of compute_masks: [(file name, {'s1': (257,T_i) float32, ...}), ...] for one batch."""
  dev = model.lin.weight.device
  model.zero_grad()
  out = []

  for num_spk in range(batch_sample.max_spk):
    if batch_sample.sub_batch_lens[num_spk] > 0:
      batch = batch_sample.sub_batch_lens[num_spk]
      combos, lens = _to_padded(batch_sample[num_spk]['combo'], dev)
      name = batch_sample[num_spk]['name']
      model.hidden = model.init_hidden(batch)
      lens_h = lens.cpu().numpy()
      dicts = [dict() for _ in range(batch)]
      with torch.no_grad():
        for dnn_pass in range(num_spk):
          mask_out = model.forward_padded(combos, lens)
          combos = ops.att_update(combos, mask_out, False)
          mask_np = mask_out.permute(1, 0, 2).cpu().numpy()
          for i in range(batch):
            dicts[i]['s' + str(dnn_pass + 1)] = mask_np[i].transpose()[:, 0:lens_h[i]]
      out += [(name[i], dicts[i]) for i in range(batch)]
  return out


def compute_masks(model, batch_sample, out_dir):
  for name, file_dict in estimate_masks(model, batch_sample):
    np.savez_compressed(out_dir + '/' + name, **file_dict)
