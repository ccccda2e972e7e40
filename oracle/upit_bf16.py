"""CPU restatement of the uPIT step in BASELINE configs[3] arithmetic ("bf16").

TEST INFRASTRUCTURE (see oracle/__init__.py).  The reference itself has no reduced-precision mode
(archs/uPIT.py runs torch fp32 throughout); configs[3] asks for "bf16 MFMA inputs, fp32 accumulate"
(SURVEY.md 8 a-4).  This file states exactly which products that means in this build, so that the HIP path
can be checked against a CPU computation of the SAME arithmetic (tight tolerance) and, separately, against
the fp32 oracle (oracle/upit.py, loose "bf16" tolerance):

  * every matrix product -- the LSTM input projections x W_ih^T and the recurrence h_{t-1} W_hh^T
    (archs/uPIT.py:132), nn.Linear (archs/uPIT.py:141), and in their backward passes the data gradients
    dY W (incl. dG W_hh through time) and the weight gradients dY^T X (incl. dW_hh = dG^T h_prev) -- rounds
    BOTH operands to bf16 (round to nearest even) and accumulates the exact products in fp32;
  * the cell non-linearities and cell state, BatchNorm, the PIT-MSE loss, gradient clipping and Adam are
    fp32, as are all tensors in memory (h_t itself is kept and emitted in fp32; only the copy that enters
    the next step's product is rounded).
"""
import torch
from torch.nn.utils.rnn import pad_packed_sequence

from . import upit as OU


def rnd(x):
    """fp32 -> bf16 (RNE) -> fp32."""
    return x.bfloat16().float()


class _Bf16Linear(torch.autograd.Function):
    """y = rnd(a) rnd(w)^T ;  da = rnd(dy) rnd(w) ;  dw = rnd(dy)^T rnd(a)."""

    @staticmethod
    def forward(ctx, a, w):
        ar, wr = rnd(a), rnd(w)
        ctx.save_for_backward(ar, wr)
        return ar @ wr.t()

    @staticmethod
    def backward(ctx, dy):
        ar, wr = ctx.saved_tensors
        dyr = rnd(dy)
        return dyr @ wr, dyr.t() @ ar


_Recurrent = _Bf16Linear          # g = rnd(h) rnd(w)^T ;  dh = rnd(dg) rnd(w) ;  dw = rnd(dg)^T rnd(h)


def blstm_padded(x, lens, weights, h0, c0):
    """oracle/upit.py::blstm_padded (nn.LSTM on a PackedSequence, archs/uPIT.py:132,135) with the products
    split as described in the module docstring."""
    T, B, _ = x.shape
    H = h0.shape[2]
    lens = torch.as_tensor(lens)
    inp = x
    hn, cn = [], []
    for l in range(len(weights)):
        outs = []
        for d in range(2):
            w_ih, w_hh, b_ih, b_hh = weights[l][d]
            gx = _Bf16Linear.apply(inp.reshape(T * B, -1), w_ih).view(T, B, 4 * H) + (b_ih + b_hh)
            h, c = h0[2 * l + d], c0[2 * l + d]
            ys = [None] * T
            for t in (range(T) if d == 0 else range(T - 1, -1, -1)):
                g = gx[t] + _Recurrent.apply(h, w_hh)
                i, f, gg, o = torch.sigmoid(g[:, :H]), torch.sigmoid(g[:, H:2 * H]), \
                    torch.tanh(g[:, 2 * H:3 * H]), torch.sigmoid(g[:, 3 * H:])
                c_new = f * c + i * gg
                h_new = o * torch.tanh(c_new)
                valid = (t < lens).unsqueeze(1)
                c = torch.where(valid, c_new, c)
                h = torch.where(valid, h_new, h)
                ys[t] = torch.where(valid, h_new, torch.zeros_like(h_new))
            outs.append(torch.stack(ys))
            hn.append(h)
            cn.append(c)
        inp = torch.cat(outs, dim=2)
    return inp, torch.stack(hn), torch.stack(cn)


def _weights(model):
    out = []
    for l in range(model.num_layers):
        out.append([tuple(getattr(model.blstm, "%s_l%d%s" % (n, l, sfx))
                          for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")) for sfx in ("", "_reverse")])
    return out


def forward(model, packed_mix, hidden):
    """SepDNN.forward (archs/uPIT.py:129-147) -> mask (B, T_max, F*S)."""
    x, lens = pad_packed_sequence(packed_mix)                    # (T, B, F)
    T, B, _ = x.shape
    y, _, _ = blstm_padded(x, lens, _weights(model), hidden[0], hidden[1])     # (T, B, 2H), zeros past len
    y = model.bn(y.permute(1, 2, 0).contiguous()).permute(0, 2, 1)            # BatchNorm1d over (B, 2H, T)
    z = _Bf16Linear.apply(y.reshape(B * T, -1), model.lin.weight).view(B, T, -1) + model.lin.bias
    return torch.sigmoid(z)


def compute_loss(model, batch_sample, hidden):
    """oracle/upit.py::compute_loss (archs/uPIT.py:157-206) in configs[3] arithmetic."""
    mix = batch_sample["mix"]
    sources = [pad_packed_sequence(batch_sample["source" + str(i + 1)], batch_first=True)[0]
               for i in range(model.num_spk)]
    model.zero_grad()
    mask_out = forward(model, mix, hidden)
    mixes, lens = pad_packed_sequence(mix, batch_first=True)
    loss, norm, losses, indices = OU.pit_mse(mask_out, mixes, sources, lens, model.num_spk, model.feat_dim)
    return loss, norm, dict(mask_out=mask_out, losses=losses, indices=indices)
