"""CPU restatement (torch-CPU) of the reference's recurrent-selective-hearing arch, archs/RSH.py.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Pinned against golden vectors produced by the reference's
own archs/RSH.py (tests/golden/make_fixtures.py -> ref_rsh_*.npz).  The reference draws (h0, c0) with randn
once per speaker-count sub-batch (archs/RSH.py:209,273); here they are passed in, one pair per non-empty
sub-batch in increasing speaker-count order.
"""
import numpy as np
import torch
import torch.nn as nn
from torch.nn.utils.rnn import pack_padded_sequence, pack_sequence, pad_packed_sequence


class MultiSpkBatch:
    """archs/RSH.py:70-85: sub_batches[k] holds the samples with k sources (possibly empty dict)."""

    def __init__(self, max_spk, length):
        self.sub_batches, self.sub_batch_lens = [], []
        self.max_spk, self.length = max_spk, length

    def __len__(self):
        return self.length

    def __getitem__(self, idx):
        return self.sub_batches[idx]


def _collate_sub(samples, key="combo"):
    order = np.argsort(np.array([len(d[key]) for d in samples]))[::-1]           # archs/RSH.py:35
    out = {}
    for k in samples[0]:
        vals = [samples[i][k] for i in order]
        out[k] = pack_sequence([torch.from_numpy(v).float() for v in vals]) if isinstance(vals[0], np.ndarray) else vals
    return out


def collate(samples):
    """Collator.__call__ (archs/RSH.py:45-68): group by speaker count ('num_spk' entry at test time, number of
    source keys at train time), each group collated like uPIT's."""
    if "num_spk" in samples[0]:
        counts = [int(d["num_spk"]) for d in samples]
    else:
        counts = [len(d.keys()) - 1 for d in samples]
    mx = max(counts)
    batch = MultiSpkBatch(mx + 1, len(samples))
    for n in range(mx + 1):
        inds = [i for i, c in enumerate(counts) if c == n]
        batch.sub_batch_lens.append(len(inds))
        batch.sub_batches.append(_collate_sub([samples[i] for i in inds]) if inds else {})
    return batch


class OracleRSH(nn.Module):
    """SepDNN of archs/RSH.py:141-187: BLSTM over [mixture | attention] (2F) -> BN -> Linear(F) -> sigmoid; the
    LSTM's final state replaces self.hidden on every call (archs/RSH.py:172), so it carries across passes."""

    def __init__(self, feat_dim=257, hidden_dim=600, num_layers=2):
        super().__init__()
        self.feat_dim, self.hidden_dim, self.num_layers = int(feat_dim), int(hidden_dim), int(num_layers)
        self.blstm = nn.LSTM(self.feat_dim * 2, self.hidden_dim, num_layers=self.num_layers, bidirectional=True)
        self.lin = nn.Linear(self.hidden_dim * 2, self.feat_dim)
        self.bn = nn.BatchNorm1d(self.hidden_dim * 2)
        self.hidden = None

    def init_hidden(self, batch_size, generator=None):
        shape = (2 * self.num_layers, batch_size, self.hidden_dim)
        return (torch.randn(shape, generator=generator), torch.randn(shape, generator=generator))

    def forward(self, x):
        x, self.hidden = self.blstm(x, self.hidden)
        x, _ = pad_packed_sequence(x, batch_first=True)
        x = self.bn(x.permute(0, 2, 1).contiguous()).permute(0, 2, 1)
        return torch.sigmoid(self.lin(x))


def compute_loss(model, batch, hiddens):
    """compute_loss (archs/RSH.py:197-259).  Returns (loss/norm, norm, aux) with per-pass masks and choices."""
    F = model.feat_dim
    model.zero_grad()
    loss, norm = 0, 0
    aux = {"masks": [], "choices": []}
    hid = iter(hiddens)
    for num_spk in range(batch.max_spk):
        if batch.sub_batch_lens[num_spk] == 0:
            continue
        nb = batch.sub_batch_lens[num_spk]
        combo = batch[num_spk]["combo"]
        model.hidden = next(hid)
        sources = [pad_packed_sequence(batch[num_spk]["source" + str(i + 1)], batch_first=True)[0] for i in range(num_spk)]
        usage = [[] for _ in range(num_spk)]
        for _ in range(num_spk):
            mask_out = model(combo)
            combos, lens = pad_packed_sequence(combo, batch_first=True)
            mixes = combos[:, :, :F]
            masked = mask_out * mixes
            losses = torch.stack([torch.sum(((masked - s) ** 2).view(nb, -1), dim=1) for s in sources])
            for si in range(num_spk):
                for index in usage[si]:
                    losses[si][index] = float("Inf")
            min_losses, indices = torch.min(losses, 0)
            for b in range(nb):
                usage[int(indices[b])].append(b)
            loss = loss + torch.sum(min_losses) / num_spk
            norm = norm + torch.sum(lens.float()) * F
            aux["masks"].append(mask_out)
            aux["choices"].append(indices.clone())
            combos = torch.relu(combos - torch.cat((torch.zeros_like(mask_out), mask_out), 2))
            combo = pack_padded_sequence(combos, lens, batch_first=True)
    return loss / norm, norm, aux


def compute_masks(model, batch, hiddens):
    """compute_masks (archs/RSH.py:262-287) -> {name: {'s1': (F,T_i), ...}}; no relu in the attention update."""
    out = {}
    hid = iter(hiddens)
    for num_spk in range(batch.max_spk):
        if batch.sub_batch_lens[num_spk] == 0:
            continue
        nb = batch.sub_batch_lens[num_spk]
        combo = batch[num_spk]["combo"]
        names = batch[num_spk]["name"]
        model.hidden = next(hid)
        dicts = [dict() for _ in range(nb)]
        for p in range(num_spk):
            mask_out = model(combo)
            combos, lens = pad_packed_sequence(combo, batch_first=True)
            combos = combos - torch.cat((torch.zeros_like(mask_out), mask_out), 2)
            combo = pack_padded_sequence(combos, lens, batch_first=True)
            for i in range(nb):
                dicts[i]["s" + str(p + 1)] = mask_out[i].detach().numpy().transpose()[:, 0:int(lens[i])]
        for i in range(nb):
            out[names[i]] = dicts[i]
    return out
