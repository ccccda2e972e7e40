"""A CPU stand-in for an arch module (archs/uPIT.py protocol) used ONLY by the host-logic tests of the training
driver: the per-rank arithmetic is the CPU oracle, the data-parallel plumbing (sepkern.dist) is the product's.
BatchNorm runs in eval mode (its batch statistics are per-rank by design) and h0/c0 are zeros, so that a
data-parallel run must reproduce the single-process run on the same global batches."""
import numpy as np
import torch
from torch.utils.data import Dataset

from oracle import upit as OU
from sepkern import dist as skdist

F, H, L = 33, 12, 2
_N, _SEED, _BN_TRAIN = 8, 0, False


def configure(n, seed, bn_train=False):
    """bn_train=True: BatchNorm stays in train mode while training (per-rank batch statistics, per-rank running statistics
    -- what the product does without sync_bn); the default keeps it in eval mode so that data-parallel == single process."""
    global _N, _SEED, _BN_TRAIN
    _N, _SEED, _BN_TRAIN = n, seed, bool(bn_train)


class TrainSet(Dataset):
    def __init__(self, datadir, location=""):
        rng = np.random.default_rng(_SEED + (17 if "cv" in datadir else 0))
        self.items = []
        for _ in range(_N):
            T = int(rng.integers(4, 12))
            d = {"mix": np.abs(rng.standard_normal((T, F))).astype(np.float32)}
            for s in range(2):
                d["source%d" % (s + 1)] = np.abs(rng.standard_normal((T, F))).astype(np.float32)
            self.items.append(d)
        self.collator = OU.collate

    def __len__(self):
        return len(self.items)

    def frame_counts(self):
        return [len(d["mix"]) for d in self.items]

    def __getitem__(self, i):
        return self.items[i]


class SepDNN(OU.OracleSepDNN):
    def __init__(self, gpuid, **kwargs):
        super().__init__(feat_dim=F, num_spk=2, hidden_dim=H, num_layers=L)
        if not _BN_TRAIN:
            self.bn.eval()

    def cuda(self):
        return self

    def train(self, mode=True):
        super().train(mode)
        if not _BN_TRAIN:
            self.bn.eval()
        return self


def compute_loss(model, epoch, batch, plotdir=""):
    B = int(batch["mix"].batch_sizes[0])
    hidden = (torch.zeros(2 * L, B, H), torch.zeros(2 * L, B, H))
    loss, norm, _ = OU.compute_loss(model, batch, hidden)
    if model.training and torch.is_grad_enabled():
        lens = torch.nn.utils.rnn.pad_packed_sequence(batch["mix"], batch_first=True)[1]
        gn = skdist.global_norm(lens, F)
        if gn is not None:                       # same rule as the product arch: divide by the GLOBAL frame count
            loss = loss * norm / gn[0]
            norm = gn[0]
    return loss, norm.detach() if torch.is_tensor(norm) else norm


def compute_cv_loss(model, epoch, batch, plotdir=""):
    return compute_loss(model, epoch, batch)


class SummingAdam:
    """torch Adam behind the gradient all-reduce the product does inside backward (sepkern.model.NetFn)."""

    def __init__(self, model, lr):
        self.model, self.opt = model, torch.optim.Adam(model.parameters(), lr=lr)

    def step(self):
        flat = torch.cat([p.grad.reshape(-1) for p in self.model.parameters()])
        skdist.allreduce_grads(flat)
        off = 0
        for p in self.model.parameters():
            p.grad.copy_(flat[off:off + p.numel()].view_as(p))
            off += p.numel()
        torch.nn.utils.clip_grad_norm_(self.model.parameters(), 0.25)
        self.opt.step()
