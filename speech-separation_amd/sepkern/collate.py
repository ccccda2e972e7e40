"""Host-side data plumbing shared by the arch plug-ins (archs/uPIT.py, archs/RSH.py): the Kaldi-style scp lists, the
per-utterance npz samples and the length-sorted PackedSequence batches the reference's collators build
(reference archs/uPIT.py:23-94, archs/RSH.py:21-138).  The BEHAVIOUR is the reference's -- callers and the golden
fixtures depend on the exact batch order, dict keys and dtypes -- the code is this project's:

  batch order   np.argsort(lengths)[::-1]  (an ascending sort reversed: among equal lengths the LATER sample comes
                first; archs/uPIT.py:40.  Not the same as a stable descending sort, and the goldens pin it.)
  arrays        float32 PackedSequence per key, packed in that order (pack_sequence on an already sorted list)
  other values  torch's default_collate on the re-ordered list (names stay lists of str, counts become tensors)
"""
import collections.abc
import os
import shutil

import numpy as np
import torch
from torch.nn.utils.rnn import pack_sequence
from torch.utils.data.dataloader import default_collate


def batch_order(samples, key):
  """Positions of `samples` (dicts) from the longest `key` entry to the shortest, the reference's way."""
  lengths = np.array([len(sample[key]) for sample in samples])
  return np.argsort(lengths)[::-1]


def collate_values(values):
  """One key of an already ordered batch: ndarrays -> float32 PackedSequence, anything else -> default_collate."""
  head = values[0]
  if isinstance(head, np.ndarray):
    if head.dtype.kind in "SaUO":           # strings / objects cannot be packed (default_collate's own rule)
      raise TypeError("batch must contain tensors, numbers, dicts or lists; found {}".format(head.dtype))
    return pack_sequence([torch.from_numpy(v).float() for v in values])
  return default_collate(values)


def collate_sorted(samples, key):
  """A list of dict samples -> dict of collated values, every entry in batch_order(samples, key)."""
  if not isinstance(samples[0], collections.abc.Mapping):
    return collate_values(samples)
  order = batch_order(samples, key)
  return {name: collate_values([samples[i][name] for i in order]) for name in samples[0]}


def read_scp(path, column=1):
  """`<id> <value>` per line (single space, as local/prepare_data_dir.sh and steps/extract_feats.py write them)."""
  with open(path) as f:
    return [line.rstrip('\n').split(' ')[column] for line in f]


def stage_copies(paths, location):
  """--train-copy-location: the feature files copied under `location` (the reference shells out to
  tools/copy_scp_data_to_dir.sh / rsync, archs/uPIT.py:57-59; same effect, in-process).  Returns the new paths."""
  staged = []
  for path in paths:
    dst = location + '/' + path
    os.makedirs(os.path.dirname(dst), exist_ok=True)
    if not os.path.exists(dst):
      shutil.copy2(path, dst)
    staged.append(dst)
  return staged


def train_sample(path, mix_key='mix', mix_map=None):
  """One feats_train npz (`mix`, `s1`..`sS` float32 (F, T)) -> {mix_key: (T, F'), 'source1': (T, F), ...}.  A file that
  holds only the mixture trains on it as its own single source (archs/uPIT.py:71-72).  mix_map: what the network sees
  of the mixture (RSH: [mixture | attention of ones])."""
  feat = np.load(path)
  mix = feat['mix'].transpose()
  sample = {mix_key: mix if mix_map is None else mix_map(mix)}
  n_src = len(feat.files) - 1
  if n_src == 0:
    sample['source1'] = mix
  for s in range(1, n_src + 1):
    sample['source%d' % s] = feat['s%d' % s].transpose()
  return sample


def eval_magnitudes(path):
  """One feats_test npz (`mix` complex64 (F, T)) -> (|mix| (T, F) float32, '<id>.npz')."""
  return np.abs(np.load(path)['mix']).transpose(), os.path.basename(path)
