"""CPU restatement (torch-CPU / numpy) of the reference's uPIT model, loss and train step.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Pinned against golden vectors produced by
the reference's own archs/uPIT.py (tests/golden/make_fixtures.py, tests/test_oracle_golden.py).

The reference hard-codes 2 layers x 600 units (archs/uPIT.py:115-119); `hidden_dim` and
`num_layers` here default to those and widen the same arithmetic to BASELINE's 2x300 and
3x896 configurations.  The reference draws h0/c0 from randn for every batch
(archs/uPIT.py:121-127); parity is only definable with (h0, c0) passed in explicitly.
"""
import itertools

import numpy as np
import torch
import torch.nn as nn
from torch.nn.utils.rnn import pack_sequence, pad_packed_sequence, PackedSequence


# --------------------------------------------------------------------------- data side
def collate_order(lengths):
    """Batch order used by Collator.__call__ (archs/uPIT.py:40): argsort ascending, reversed."""
    return np.argsort(np.array(lengths))[::-1]


def collate(samples, key="mix"):
    """Collator.__call__ for a list of dict samples (archs/uPIT.py:33-48) -> dict of PackedSequence."""
    order = collate_order([len(d[key]) for d in samples])
    out = {}
    for k in samples[0]:
        vals = [samples[i][k] for i in order]
        if isinstance(vals[0], np.ndarray):
            out[k] = pack_sequence([torch.from_numpy(v).float() for v in vals])
        else:
            out[k] = vals
    return out


# --------------------------------------------------------------------------- model
class OracleSepDNN(nn.Module):
    """SepDNN (archs/uPIT.py:97-147): BLSTM -> pad -> BatchNorm1d over (B, 2H, T) -> Linear -> sigmoid.

    Sub-module construction order (blstm, lin, bn) follows archs/uPIT.py:115-119 so that the
    same torch.manual_seed gives the same initial weights and the same state_dict keys.
    """

    def __init__(self, feat_dim=257, num_spk=2, hidden_dim=600, num_layers=2):
        super().__init__()
        self.feat_dim, self.num_spk = int(feat_dim), int(num_spk)
        self.hidden_dim, self.num_layers = int(hidden_dim), int(num_layers)
        self.blstm = nn.LSTM(self.feat_dim, self.hidden_dim, num_layers=self.num_layers, bidirectional=True)
        self.lin = nn.Linear(self.hidden_dim * 2, self.feat_dim * self.num_spk)
        self.bn = nn.BatchNorm1d(self.hidden_dim * 2)

    def init_hidden(self, batch_size, generator=None):
        """archs/uPIT.py:121-127: two independent randn draws of (2L, B, H)."""
        shape = (2 * self.num_layers, batch_size, self.hidden_dim)
        return (torch.randn(shape, generator=generator), torch.randn(shape, generator=generator))

    def forward(self, packed, hidden):
        """archs/uPIT.py:129-147.  Returns mask (B, T_max, F*S) and the final (hn, cn)."""
        x, hidden_out = self.blstm(packed, hidden)
        x, _ = pad_packed_sequence(x, batch_first=True)
        x = self.bn(x.permute(0, 2, 1).contiguous()).permute(0, 2, 1)
        x = self.lin(x)
        return torch.sigmoid(x), hidden_out


def pit_mse(mask_out, mixes, sources, lens, num_spk, feat_dim):
    """The loss body of compute_loss (archs/uPIT.py:178-197, 206).

    mask_out (B,T,F*S), mixes (B,T,F), sources list of S (B,T,F), lens (B,) ->
    (loss/norm, norm, losses (S!,B), argmin indices (B,)).
    """
    batch = mask_out.shape[0]
    stacked_mix = torch.cat([mixes for _ in range(num_spk)], dim=2)
    masked = mask_out * stacked_mix
    perms = list(itertools.permutations(range(num_spk)))           # lexicographic (archs/uPIT.py:186)
    losses = torch.stack([
        torch.sum(((masked - torch.cat([sources[i] for i in perm], dim=2)) ** 2).view(batch, -1), dim=1)
        for perm in perms])
    min_losses, indices = torch.min(losses, 0)
    loss = torch.sum(min_losses) / num_spk
    norm = torch.sum(lens.float()) * feat_dim
    return loss / norm, norm, losses, indices


def compute_loss(model, batch_sample, hidden):
    """compute_loss (archs/uPIT.py:157-206) with (h0,c0) injected instead of drawn."""
    mix = batch_sample["mix"]
    sources = [pad_packed_sequence(batch_sample["source" + str(i + 1)], batch_first=True)[0]
               for i in range(model.num_spk)]
    model.zero_grad()
    mask_out, _ = model(mix, hidden)
    mixes, lens = pad_packed_sequence(mix, batch_first=True)
    loss, norm, losses, indices = pit_mse(mask_out, mixes, sources, lens, model.num_spk, model.feat_dim)
    return loss, norm, dict(mask_out=mask_out, losses=losses, indices=indices)


def compute_masks(model, batch_sample, hidden):
    """compute_masks (archs/uPIT.py:209-225) -> {name: {'s1': (F,T_i), ...}} instead of writing npz."""
    mix = batch_sample["mix"]
    mask_out, _ = model(mix, hidden)
    lens = pad_packed_sequence(mix, batch_first=True)[1]
    out = {}
    for i, name in enumerate(batch_sample["name"]):
        mask = mask_out[i].detach().numpy().transpose()[:, 0:int(lens[i])]
        out[name] = {"s" + str(s + 1): mask[s * model.feat_dim:(s + 1) * model.feat_dim]
                     for s in range(model.num_spk)}
    return out


def train_step(model, optimizer, batch_sample, hidden, max_norm=0.25):
    """One iteration of the loop at steps/train_qsub.py:116-122."""
    loss, norm, aux = compute_loss(model, batch_sample, hidden)
    loss.backward()
    gnorm = torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm)
    optimizer.step()
    return float(loss.detach()), float(norm.detach()), float(gnorm), aux


# ------------------------------------------------- padded / masked formulation of the BLSTM
def blstm_padded(x, lens, weights, h0, c0):
    """The BLSTM as the HIP kernels compute it: padded time-major input, per-row length masks.

    Restates nn.LSTM-on-PackedSequence semantics (archs/uPIT.py:132,135) without packing:
    for row b the state is frozen and the output is zero at t >= len_b; the reverse
    direction therefore starts from (h0,c0) at t = len_b-1.  Checked against nn.LSTM in
    tests/test_oracle_units.py.

    x (T,B,I); weights: list over layers of dict(dir -> (w_ih, w_hh, b_ih, b_hh));
    h0,c0 (2L,B,H).  Returns y (T,B,2H), hn, cn (2L,B,H).
    """
    T, B, _ = x.shape
    L = len(weights)
    H = h0.shape[2]
    lens = torch.as_tensor(lens)
    inp = x
    hn, cn = torch.zeros_like(h0), torch.zeros_like(c0)
    for l in range(L):
        outs = []
        for d in range(2):
            w_ih, w_hh, b_ih, b_hh = weights[l][d]
            gx = inp @ w_ih.t() + (b_ih + b_hh)                  # input projection for all t
            h, c = h0[2 * l + d].clone(), c0[2 * l + d].clone()
            y = torch.zeros(T, B, H, dtype=x.dtype)
            steps = range(T) if d == 0 else range(T - 1, -1, -1)
            for t in steps:
                g = gx[t] + h @ w_hh.t()
                i, f, gg, o = torch.sigmoid(g[:, :H]), torch.sigmoid(g[:, H:2 * H]), \
                    torch.tanh(g[:, 2 * H:3 * H]), torch.sigmoid(g[:, 3 * H:])
                c_new = f * c + i * gg
                h_new = o * torch.tanh(c_new)
                valid = (t < lens).unsqueeze(1)
                c = torch.where(valid, c_new, c)
                h = torch.where(valid, h_new, h)
                y[t] = torch.where(valid, h_new, torch.zeros_like(h_new))
            hn[2 * l + d], cn[2 * l + d] = h, c
            outs.append(y)
        inp = torch.cat(outs, dim=2)
    return inp, hn, cn


def lstm_weights(model):
    """Per-layer, per-direction (w_ih, w_hh, b_ih, b_hh) from an nn.LSTM-holding model."""
    out = []
    for l in range(model.num_layers):
        layer = []
        for sfx in ("", "_reverse"):
            layer.append(tuple(getattr(model.blstm, "%s_l%d%s" % (n, l, sfx)).detach()
                               for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")))
        out.append(layer)
    return out


# --------------------------------------------------------------------------- scoring
def si_sdr(est, ref):
    """Scale-invariant SDR in dB (Le Roux et al. 2019), zero-mean; not in the reference
    (it scores with mir_eval BSS-eval, steps/evaluate_sources.py:57) -- used for the
    +-0.1 dB parity gate on reconstructed waveforms."""
    est = np.asarray(est, dtype=np.float64) - np.mean(est)
    ref = np.asarray(ref, dtype=np.float64) - np.mean(ref)
    alpha = np.dot(est, ref) / (np.dot(ref, ref) + 1e-30)
    target = alpha * ref
    noise = est - target
    return 10.0 * np.log10((np.dot(target, target) + 1e-30) / (np.dot(noise, noise) + 1e-30))


def si_sdr_best_perm(ests, refs):
    """Mean SI-SDR under the best speaker permutation."""
    S = len(refs)
    best = None
    for perm in itertools.permutations(range(S)):
        v = float(np.mean([si_sdr(ests[s], refs[perm[s]]) for s in range(S)]))
        best = v if best is None or v > best else best
    return best
