"""Row layout of one length-sorted batch: torch's PackedSequence layout, kept on the device end to end.

The reference's collator packs every batch (archs/uPIT.py:36-46: sort by length, pack_sequence) and nn.LSTM runs
on the packed rows (archs/uPIT.py:132): only the R = sum(lens) valid frames exist, time-major -- row of (t, j) is
offs[t] + j for j < n_t, n_t = number of utterances longer than t.  The engine keeps that layout through every
product, statistic and store (include/sepkern.h, "packed rows"); a Packing is the small table that describes it.
"""
import numpy as np
import torch

from . import ops
from ._lib import SepkernError


class Packing:
    """T (longest utterance), B, R = sum(lens), Rp = R rounded up to 64 (row buffers carry a zero tail so that products
    that contract over rows run whole K steps), lens / offs as host numpy int32 and as device int32 tensors, `perm`
    (device int32, or None): sorted position j holds the caller's utterance perm[j], `uniform`: all lengths equal
    (packed == padded)."""

    def __init__(self, lens_sorted, device, perm=None):
        lens_sorted = np.ascontiguousarray(lens_sorted, dtype=np.int32)
        if lens_sorted.ndim != 1 or lens_sorted.size == 0 or lens_sorted[-1] < 1:
            raise SepkernError("Packing: need at least one utterance and every length >= 1")
        if np.any(lens_sorted[1:] > lens_sorted[:-1]):
            raise SepkernError("Packing: lengths must be sorted in descending order")
        self.B, self.T = int(lens_sorted.size), int(lens_sorted[0])
        # n_t = #{j: len_j > t}; offs = exclusive prefix sum (T + 1 entries)
        n = np.searchsorted(-lens_sorted, -np.arange(1, self.T + 1, dtype=np.int32), side="right").astype(np.int32)
        offs = np.zeros(self.T + 1, dtype=np.int32)
        np.cumsum(n, out=offs[1:])
        self.R = int(offs[-1])
        self.Rp = ops.pad_to(self.R, 64)
        self.uniform = bool(lens_sorted[-1] == lens_sorted[0])
        self.lens_host, self.offs_host, self.perm_host = lens_sorted, offs, perm
        self.device = torch.device(device)
        # one staging buffer, one copy: [lens | offs | perm]
        parts = [lens_sorted, offs] + ([np.ascontiguousarray(perm, dtype=np.int32)] if perm is not None else [])
        host = torch.from_numpy(np.concatenate(parts))
        if self.device.type == "cuda":
            host = host.pin_memory()
        dev = host.to(self.device, non_blocking=True)
        self._host = host                                    # keeps the pinned staging alive until the copy has run
        self.lens = dev[:self.B]
        self.offs = dev[self.B:self.B + self.T + 1]
        self.perm = dev[self.B + self.T + 1:] if perm is not None else None

    # ------------------------------------------------------------------ constructors
    @classmethod
    def from_batch_sizes(cls, batch_sizes, device):
        """From PackedSequence.batch_sizes (a CPU int64 tensor: n_t), as the reference's collator produces it."""
        n = np.asarray(batch_sizes, dtype=np.int64)
        B = int(n[0])
        # len_j = #{t: n_t > j}
        lens = np.searchsorted(-n, -np.arange(1, B + 1, dtype=np.int64), side="right").astype(np.int32)
        return cls(lens, device)

    @classmethod
    def from_lens(cls, lens, device):
        """From per-utterance lengths in the caller's order (any order: a stable sort makes the layout, `perm` maps
        sorted positions back)."""
        if torch.is_tensor(lens):
            lens = lens.detach().cpu().numpy()              # (synchronises: callers on the hot path pass host lengths)
        lens = np.asarray(lens, dtype=np.int32)
        if lens.size > 1 and np.any(lens[1:] > lens[:-1]):
            perm = np.argsort(-lens, kind="stable").astype(np.int32)
            return cls(lens[perm], device, perm=perm)
        return cls(lens, device)

    # ------------------------------------------------------------------ padded <-> packed
    def rows(self, C, dtype=torch.float32):
        """An (Rp, C) row buffer whose tail rows R.. are zero (the first R rows are the caller's to fill)."""
        t = torch.empty(self.Rp, C, dtype=dtype, device=self.device)
        if self.Rp > self.R:
            t[self.R:].zero_()
        return t

    def pack(self, padded):
        """(T, B, C) zero-padded, caller's utterance order -> (Rp, C) packed rows (a view when nothing moves)."""
        T, B, C = padded.shape
        if T < self.T or B != self.B:          # (T > self.T: frames past the longest utterance are all padding)
            raise SepkernError("pack: tensor is (%d, %d, .), the batch is (%d, %d)" % (T, B, self.T, self.B))
        padded = padded.contiguous()
        if self.uniform and self.perm is None and self.Rp == self.R and T == self.T:
            return padded.view(self.R, C)
        out = self.rows(C)
        ops.pack_rows(padded, self, out)
        return out

    def unpack(self, packed, C=None, fill=None, T=None):
        """(>= R, ld) packed rows -> (T, B, C) in the caller's utterance order (T >= the longest utterance, default equal);
        padded positions hold zeros, or the row `fill` (C floats)."""
        C = packed.shape[1] if C is None else C
        T = self.T if T is None else T
        if self.uniform and self.perm is None and packed.shape[1] == C and packed.is_contiguous() and T == self.T:
            return packed[:self.R].view(self.T, self.B, C)
        out = torch.empty(T, self.B, C, device=packed.device)
        ops.unpack_rows(packed, self, out[:self.T], fill)
        if T > self.T:
            out[self.T:] = 0.0 if fill is None else fill
        return out

    def sort_batch(self, t, dim):
        """Reorder a per-utterance tensor (h0, c0, ...) from the caller's order into sorted order along `dim`."""
        return t if self.perm is None else t.index_select(dim, self.perm.long())

    def unsort_batch(self, t, dim):
        if self.perm is None:
            return t
        out = torch.empty_like(t)
        out.index_copy_(dim, self.perm.long(), t)
        return out
