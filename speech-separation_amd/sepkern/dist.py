"""Utterance-level data parallelism: one process per GPU, torch.distributed over RCCL ("nccl").

The reference is single-GPU (steps/qsub_train.sh:5 `-l gpu=1`); this is new functionality whose
contract is "same update as one device seeing the global batch" for everything except BatchNorm,
whose batch statistics stay per-rank (as torch's DistributedDataParallel does by default):
  * utterances are sharded by index across ranks (no data-path collective),
  * the PIT loss of every rank is divided by the GLOBAL norm sum(len)*F (one scalar all-reduce,
    known before the forward pass because it depends on lengths only),
  * the flat fp32 gradient buffer is summed with ONE all-reduce per step, after which every rank
    runs the identical clip + Adam update,
  * optionally (conf key sync_bn=1 / SEPKERN_SYNC_BN=1) BatchNorm uses the statistics of the GLOBAL batch: one
    all-gather of the per-rank (count, mean, variance) in the forward pass and one all-reduce of the two
    per-channel sums in the backward pass; the update then equals the single-device global-batch update.
These helpers are backend-agnostic so the N>1 path is covered by world_size-2 gloo tests on CPU.
"""
import os

import torch
import torch.distributed as dist


def is_parallel():
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def world():
    return dist.get_world_size() if is_parallel() else 1


def rank():
    return dist.get_rank() if (dist.is_available() and dist.is_initialized()) else 0


def init_from_env(backend=None):
    """Initialise the process group from torchrun's environment (RANK / WORLD_SIZE / MASTER_*).
    Returns (rank, world, local_rank).  No-op for a single process."""
    w = int(os.environ.get("WORLD_SIZE", "1"))
    r = int(os.environ.get("RANK", "0"))
    lr = int(os.environ.get("LOCAL_RANK", "0"))
    if w > 1 and not dist.is_initialized():
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = backend or os.environ.get("SEPKERN_DIST_BACKEND", "nccl")
        if backend == "nccl":
            torch.cuda.set_device(lr)
            dist.init_process_group("nccl", device_id=torch.device("cuda", lr))
        else:
            dist.init_process_group(backend)
    return r, w, lr


def shard_indices(n, rank_, world_):
    """Utterance indices of this rank: a strided shard, every index exactly once over all ranks."""
    return list(range(rank_, n, world_))


def global_norm(lens, feat_dim):
    """Device scalar sum over ALL ranks of sum(len_b) * F (reference archs/uPIT.py:197 on the global batch),
    or None when not parallel (the kernel then uses its local norm).  Stays on the device: the
    all-reduce is enqueued on the stream, no host round trip per step."""
    if not is_parallel():
        return None
    t = (lens.sum().to(torch.float32) * float(feat_dim)).reshape(1)
    dist.all_reduce(t)
    return t


def allreduce_grads(flat_grad):
    """Sum the flat gradient buffer over ranks in place (one collective over xGMI)."""
    if is_parallel():
        dist.all_reduce(flat_grad)
    return flat_grad


def combine_bn_stats(mean, var, count):
    """Per-rank batch statistics (mean, biased variance over `count` rows) -> statistics of the union of all ranks'
    rows, by the pairwise-combination formula M2 = sum_r n_r (var_r + (mean_r - mean)^2).  One all-gather.
    Returns (mean, var, total_count); the inputs unchanged when not parallel."""
    if not is_parallel():
        return mean, var, float(count)
    C = mean.numel()
    mine = torch.cat([mean.reshape(-1), var.reshape(-1), mean.new_full((1,), float(count))])
    allr = [torch.empty_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(allr, mine)
    st = torch.stack(allr)                                   # (world, 2C+1)
    n = st[:, 2 * C:2 * C + 1]                               # (world, 1)
    total = n.sum()
    gmean = (st[:, :C] * n).sum(0) / total
    gvar = (n * (st[:, C:2 * C] + (st[:, :C] - gmean) ** 2)).sum(0) / total
    return gmean.contiguous(), gvar.contiguous(), float(total.item())


def allreduce_bn_sums(dgamma, dbeta):
    """Global per-channel sums for the BatchNorm backward (copies; the local ones stay the parameter gradients that
    the flat all-reduce sums later)."""
    both = torch.stack([dgamma, dbeta])
    if is_parallel():
        dist.all_reduce(both)
    return both[0].contiguous(), both[1].contiguous()
