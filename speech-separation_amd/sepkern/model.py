"""Shared machinery of the arch plug-ins (archs/uPIT.py, archs/RSH.py): an nn.Module whose parameters keep the
reference's names (state_dict-compatible) but live as views in the engine's flat parameter / gradient
buffers, plus the autograd bridge to the libsepkern forward/backward."""
import torch
import torch.nn as nn
from torch.nn.utils.rnn import PackedSequence, pad_packed_sequence

from . import dist as skdist
from .engine import Engine
from .packing import Packing
from ._lib import SepkernError


class _Params(nn.Module):
  """A bag of named parameters/buffers (keeps the reference's state_dict keys, e.g. blstm.weight_ih_l0)."""


class NetFn(torch.autograd.Function):
  """(mask, hn, cn) = net(x, h0, c0).  backward runs the libsepkern backward kernels: parameter gradients
  go to the flat gradient buffer every param.grad is a view of; gradients wrt x, h0, c0 are returned
  (the RSH arch chains passes through the attention input and the carried hidden state).
  x is either the packed rows (R, in_dim) of the batch `pk` (the native layout: PackedSequence.data) or, with
  padded=True, a zero-padded (T, B, in_dim) tensor in the caller's utterance order, which is packed on the way in and
  unpacked on the way out (h0 / c0 / hn / cn then follow the caller's order too)."""

  @staticmethod
  def forward(ctx, anchor, model, x, pk, h0, c0, want_state, padded):
    ctx.model, ctx.pk, ctx.padded = model, pk, padded
    if padded:
      ctx.T_pad = T_pad = x.shape[0]
      x, h0, c0 = pk.pack(x), pk.sort_batch(h0, 1).contiguous(), pk.sort_batch(c0, 1).contiguous()
    mask, hn, cn, ctx.fwd = model._engine.forward(x, pk, h0, c0, model.training, save=True, want_state=want_state)
    if padded:
      mask = pk.unpack(mask, fill=None if (pk.uniform and T_pad == pk.T) else model._engine.pad_row(), T=T_pad)
      if want_state:
        hn, cn = pk.unsort_batch(hn, 1), pk.unsort_batch(cn, 1)
    if not want_state:
      return mask
    return mask, hn, cn

  @staticmethod
  def backward(ctx, dmask, dhn=None, dcn=None):
    model, pk = ctx.model, ctx.pk
    want_dx = ctx.needs_input_grad[2]
    want_ds = ctx.needs_input_grad[4] or ctx.needs_input_grad[5]
    last = model._pending <= 1        # the last backward of the step: the gradients become final chunk by chunk
    reducer = model._reducer() if (last and skdist.is_parallel() and skdist.overlap_enabled()) else None
    dmask = dmask.contiguous()
    if ctx.padded:
      dmask = pk.pack(dmask)
      dhn = pk.sort_batch(dhn, 1) if dhn is not None else None
      dcn = pk.sort_batch(dcn, 1) if dcn is not None else None
    dx, dh0, dc0 = model._engine.backward(ctx.fwd, dmask,
                                          dhn.contiguous() if dhn is not None else None,
                                          dcn.contiguous() if dcn is not None else None,
                                          want_dx=want_dx, want_dstate=want_ds, reducer=reducer)
    if ctx.padded:
      dx = pk.unpack(dx, T=ctx.T_pad) if dx is not None else None
      dh0 = pk.unsort_batch(dh0, 1) if dh0 is not None else None
      dc0 = pk.unsort_batch(dc0, 1) if dc0 is not None else None
    ctx.fwd = None
    model._pending -= 1
    if model._pending <= 0:
      if reducer is not None:
        reducer.finish()              # the chunks went out during the pass (SEPKERN_DP_OVERLAP=1)
      else:
        model._allreduce_grads()      # once per step, after the last pass's backward
    return None, None, dx, None, dh0, dc0, None, None


class UnpackFn(torch.autograd.Function):
  """padded (T, B, C) = unpack(packed rows (R, C)) with `fill` at the padded positions; backward is the opposite row mover
  (sk_pack_rows of the incoming gradient: padded positions carry no gradient into the network -- their value is a constant
  of the batch as far as this graph goes).  Keeps `model(x)` (reference loop: archs/uPIT.py:175, mask_out = model(mix) ->
  loss -> backward) differentiable on variable-length batches, where Packing.unpack is a raw kernel launch."""

  @staticmethod
  def forward(ctx, packed, pk, fill):
    ctx.pk, ctx.R = pk, packed.shape[0]
    return pk.unpack(packed, fill=fill)

  @staticmethod
  def backward(ctx, dpadded):
    return ctx.pk.pack(dpadded.contiguous())[:ctx.R], None, None


class SepDNNBase(nn.Module):
  """BLSTM(in_dim -> H, L layers, bidirectional) -> BatchNorm1d(2H) -> Linear(2H -> out_dim) -> sigmoid."""

  def _build(self, gpuid, in_dim, out_dim, hidden_dim, num_layers, precision="fp32", sync_bn=False):
    self.gpuid = gpuid
    if int(gpuid) < 0:
      raise SepkernError("SepDNN(gpuid=%s): this build has no CPU path; it runs on an MI355X only" % gpuid)
    self.in_dim, self.out_dim = int(in_dim), int(out_dim)
    self.hidden_dim, self.num_layers = int(hidden_dim), int(num_layers)
    self.precision = str(precision)
    self.sync_bn = str(sync_bn).lower() in ("1", "true", "yes")
    H, L = self.hidden_dim, self.num_layers
    # Initial values exactly as the reference draws them (nn.LSTM, nn.Linear, nn.BatchNorm1d constructed in
    # this order consume the RNG identically, archs/uPIT.py:115-119 / archs/RSH.py:155-159); the modules are
    # only used as initialisers, the kernels never call them.
    init_lstm = nn.LSTM(self.in_dim, H, num_layers=L, bidirectional=True)
    init_lin = nn.Linear(H * 2, self.out_dim)
    init_bn = nn.BatchNorm1d(H * 2)
    self.blstm = _Params()
    for name, p in init_lstm.named_parameters():
      self.blstm.register_parameter(name, nn.Parameter(p.detach().clone()))
    self.lin = _Params()
    self.lin.register_parameter('weight', nn.Parameter(init_lin.weight.detach().clone()))
    self.lin.register_parameter('bias', nn.Parameter(init_lin.bias.detach().clone()))
    self.bn = _Params()
    self.bn.register_parameter('weight', nn.Parameter(init_bn.weight.detach().clone()))
    self.bn.register_parameter('bias', nn.Parameter(init_bn.bias.detach().clone()))
    self.bn.register_buffer('running_mean', init_bn.running_mean.clone())
    self.bn.register_buffer('running_var', init_bn.running_var.clone())
    self.bn.register_buffer('num_batches_tracked', init_bn.num_batches_tracked.clone())
    self.hidden = None
    self.next_hidden = None          # tests: (h0, c0) used by the next init_hidden call
    self.hidden_generator = None
    self._engine = None
    self._anchor = None
    self._pending = 0

  # ---- parameter storage: every nn.Parameter is a view into the engine's flat buffer
  def _named_views(self, flat_view):
    for l in range(self.num_layers):
      for d, sfx in enumerate(("", "_reverse")):
        for base in ("weight_ih", "weight_hh", "bias_ih", "bias_hh"):
          yield getattr(self.blstm, "%s_l%d%s" % (base, l, sfx)), flat_view("%s_l%d" % (base, l))[d]
    yield self.lin.weight, flat_view("lin.weight")
    yield self.lin.bias, flat_view("lin.bias")
    yield self.bn.weight, flat_view("bn.weight")
    yield self.bn.bias, flat_view("bn.bias")

  def _bind(self):
    """(Re)attach parameters to the flat buffers; cheap when already bound."""
    dev = self.lin.weight.device
    if dev.type != 'cuda':
      raise SepkernError("SepDNN must be moved to the GPU (model.cuda()) before use; there is no CPU path")
    eng = self._engine
    if eng is not None and eng.device == dev and self.lin.weight.data_ptr() == eng.p("lin.weight").data_ptr():
      eng.running_mean, eng.running_var = self.bn.running_mean, self.bn.running_var
      return eng
    with torch.cuda.device(dev):
      eng = Engine(self.in_dim, self.out_dim, self.hidden_dim, self.num_layers, dev, self.precision, self.sync_bn)
    with torch.no_grad():
      for p, v in self._named_views(eng.p):
        v.copy_(p.data)
        p.data = v
      for p, gv in self._named_views(eng.g):
        if p.grad is not None:
          gv.copy_(p.grad)
        p.grad = gv
    eng.running_mean, eng.running_var = self.bn.running_mean, self.bn.running_var
    self._engine = eng
    self._anchor = torch.zeros(1, device=dev, requires_grad=True)
    return eng

  def zero_grad(self, set_to_none=False):
    # gradients stay views of the flat buffer; the next backward overwrites them
    self._pending = 0
    if self._engine is not None:
      self._engine.zero_grad()
    else:
      super(SepDNNBase, self).zero_grad(set_to_none=False)

  def flat_parameters(self):
    """(params, grads): the two flat fp32 buffers (for sepkern.optim.ClipAdam and the DP all-reduce)."""
    eng = self._bind()
    return eng.flat, eng.grad

  def _reducer(self):
    if getattr(self, "_grad_reducer", None) is None:
      self._grad_reducer = skdist.GradReducer()
    return self._grad_reducer

  def _allreduce_grads(self):
    skdist.allreduce_grads(self._engine.grad_full)  # one RCCL collective: every gradient + the status word

  def check_status(self):
    """Raise if a persistent recurrence launch timed out since the last check (host sync; epoch / checkpoint time)."""
    if self._engine is not None:
      self._engine.check_status()

  def init_hidden(self, batch_size):
    """h0, c0 ~ N(0,1), shape (2L, B, H), fresh for every batch (reference archs/uPIT.py:121-127)."""
    if self.next_hidden is not None:
      if isinstance(self.next_hidden, list):         # a queue: one pair per init_hidden call (RSH sub-batches)
        h = self.next_hidden.pop(0)
        if not self.next_hidden:
          self.next_hidden = None
        return h
      h, self.next_hidden = self.next_hidden, None
      return h
    dev = self.lin.weight.device
    shape = (2 * self.num_layers, batch_size, self.hidden_dim)
    return (torch.randn(shape, device=dev, generator=self.hidden_generator),
            torch.randn(shape, device=dev, generator=self.hidden_generator))

  def packing_of(self, lens):
    """The Packing of a batch given per-utterance lengths (device int32 tensor, list or array; any order).  A device
    tensor costs a host read-back, so the last one is remembered (the RSH arch runs several passes per batch)."""
    if torch.is_tensor(lens):
      key = (lens.data_ptr(), lens._version, int(lens.numel()))
      hit = getattr(self, "_pk_last", None)
      if hit is not None and hit[0] == key:
        return hit[1]
      pk = Packing.from_lens(lens, self.lin.weight.device)
      self._pk_last = (key, pk, lens)         # (holds `lens`: its address cannot be reused while the entry is alive)
      return pk
    return Packing.from_lens(lens, self.lin.weight.device)

  def _run(self, x, pk, h0, c0, want_state, padded):
    eng = self._bind()
    h0 = h0.to(x.device, torch.float32).contiguous()
    c0 = c0.to(x.device, torch.float32).contiguous()
    if self.training:
      self.bn.num_batches_tracked += 1
    if torch.is_grad_enabled():
      self._pending += 1
      return NetFn.apply(self._anchor, self, x, pk, h0, c0, want_state, padded)
    if padded:
      T_pad = x.shape[0]
      x, h0, c0 = pk.pack(x), pk.sort_batch(h0, 1).contiguous(), pk.sort_batch(c0, 1).contiguous()
    mask, hn, cn, _ = eng.forward(x, pk, h0, c0, self.training, save=False, want_state=want_state)
    if padded:
      mask = pk.unpack(mask, fill=None if (pk.uniform and T_pad == pk.T) else eng.pad_row(), T=T_pad)
      if want_state:
        hn, cn = pk.unsort_batch(hn, 1), pk.unsort_batch(cn, 1)
    return (mask, hn, cn) if want_state else mask

  def run_net_packed(self, x2d, pk, h0, c0, want_state=False):
    """x2d (R, in_dim) packed rows of the batch `pk` (sepkern.packing.Packing; PackedSequence.data as the collator built
    it), h0 / c0 (2L,B,H) in the batch's sorted order -> mask (R, out_dim) packed [, hn, cn].  Differentiable when grad
    is enabled.  This is the engine's native layout: nothing is copied."""
    return self._run(x2d, pk, h0, c0, want_state, False)

  def run_net(self, x, lens, h0, c0, want_state=False):
    """x (T,B,in_dim) time-major zero-padded CUDA tensor, lens int32 CUDA (B) or host lengths, any order -> mask
    (T,B,out_dim), at padded positions the value the reference's network shows there [, hn, cn (2L,B,H)].  Differentiable when grad is enabled.  The batch is
    packed on the way in and unpacked on the way out (sk_pack_rows / sk_unpack_rows)."""
    return self._run(x, self.packing_of(lens), h0, c0, want_state, True)


def to_packed(packed, device):
  """PackedSequence (as the Collators build it) -> (its data on the device (R, C), Packing).  No padding is ever made:
  PackedSequence.data IS the engine's row layout."""
  if not isinstance(packed, PackedSequence):
    raise TypeError("expected a PackedSequence from the arch's collator")
  if packed.sorted_indices is not None:
    raise TypeError("expected a PackedSequence of an already length-sorted batch (the arch's collator sorts)")
  pk = Packing.from_batch_sizes(packed.batch_sizes, device)
  data = packed.data
  if data.device != torch.device(device):
    data = data.to(device, non_blocking=True)
  return data, pk


def to_padded(packed, device):
  """PackedSequence (as the Collators build it) -> zero-padded time-major CUDA tensor + int32 lengths."""
  if not isinstance(packed, PackedSequence):
    raise TypeError("expected a PackedSequence from the arch's collator")
  x, lens = pad_packed_sequence(packed.to(device))
  return x.contiguous(), lens.to(device=device, dtype=torch.int32)
