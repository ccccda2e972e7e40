"""Utterance-level data parallelism: one process per GPU, torch.distributed over RCCL ("nccl").

The reference is single-GPU (steps/qsub_train.sh:5 `-l gpu=1`); this is new functionality whose
contract is "same update as one device seeing the global batch" for everything except BatchNorm,
whose BATCH statistics stay per-rank (as under torch's DistributedDataParallel without SyncBatchNorm; DDP keeps the
RUNNING statistics consistent by re-broadcasting the buffers from rank 0 at every forward -- here the ranks' running
statistics are averaged once per epoch, before anything is scored or saved: average_bn_buffers):
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
        if overlap_enabled():
            # chunked exchange beside the backward pass: keep the collective's kernels within the CUs a persistent
            # recurrence grid leaves free (32 of 256 at the benchmark shape)
            os.environ.setdefault("NCCL_MAX_NCHANNELS", "16")
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


# bench.py: TIMING = [] makes allreduce_grads record a (start, end) HIP-event pair per call on the current stream
TIMING = None


def overlap_mode_name():
    """How the gradient exchange of a step is issued (bench.py prints it): one collective after the backward pass
    (default) or layer-ordered chunks on a communication stream while the backward pass is still running
    (SEPKERN_DP_OVERLAP=1, GradReducer)."""
    return "chunked-overlapped" if overlap_enabled() else "single"


def allreduce_grads(flat_grad):
    """Sum the flat gradient buffer over ranks in place (one collective over xGMI)."""
    if is_parallel():
        if TIMING is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            dist.all_reduce(flat_grad)
            e1.record()
            TIMING.append((e0, e1))
        else:
            dist.all_reduce(flat_grad)
    return flat_grad


class GradReducer:
    """SEPKERN_DP_OVERLAP=1 (default off): the gradient exchange of a step as layer-ordered chunks instead of one
    collective after the backward pass.  The engine hands over each chunk of its flat gradient buffer (Linear +
    BatchNorm, then the LSTM layers from the top down, the bottom layer with the guard words last:
    engine.ParamLayout.grad_chunks) as soon as the kernels that complete it are enqueued; the chunk is all-reduced on a
    communication stream behind an event of the producing stream, while the backward pass goes on with the next layer's
    recurrence.  finish() makes the current stream wait for all of them.  Same sums as the single collective (every
    element is reduced exactly once), every rank issues the same sequence.

    Why it is opt-in: the persistent recurrence needs its workgroups co-resident (224 of 256 CUs at 3x896 / batch 32);
    an RCCL kernel that holds more CUs than the grid leaves free would park some of them behind it (bounded spins: a
    skipped step, never a wrong one).  init_from_env caps NCCL_MAX_NCHANNELS for this mode.  What it hides of the exchange
    has not been measured: RCCL has not run on hardware yet (tools/scale_sweep.py times both modes on a node)."""

    def __init__(self):
        self.comm = None
        self.pending = []
        self.first = None

    def chunk(self, buf, lo, hi, producer_stream=None):
        if not is_parallel():
            return
        part = buf[lo:hi]
        if not part.is_cuda:                                  # CPU tensors (gloo tests): in line
            dist.all_reduce(part)
            return
        if self.comm is None:
            self.comm = torch.cuda.Stream(device=part.device)
        ev = torch.cuda.Event()
        ev.record(producer_stream if producer_stream is not None else torch.cuda.current_stream())
        self.comm.wait_event(ev)
        with torch.cuda.stream(self.comm):
            if TIMING is not None and self.first is None:
                self.first = torch.cuda.Event(enable_timing=True)
                self.first.record()
            self.pending.append(dist.all_reduce(part, async_op=True))

    def finish(self):
        """The current stream waits for every chunk issued since the last finish()."""
        for w in self.pending:
            w.wait()
        self.pending = []
        if self.comm is not None:
            cur = torch.cuda.current_stream()
            cur.wait_stream(self.comm)
            if TIMING is not None and self.first is not None:
                e1 = torch.cuda.Event(enable_timing=True)
                e1.record(cur)
                TIMING.append((self.first, e1))            # span from the first chunk's issue to the last one's end
            self.first = None


def overlap_enabled():
    return os.environ.get("SEPKERN_DP_OVERLAP", "0") == "1"


def combine_bn_stats(mean, var, B, T):
    """Per-rank batch statistics (mean, biased variance over this rank's zero-padded (B, T) grid of frames) -> statistics
    of the GLOBAL batch's grid, (sum of the ranks' B) x (longest utterance of any rank), by the pairwise-combination
    formula M2 = sum_r n_r (var_r + (mean_r - mean)^2): a rank whose longest utterance is shorter than the global one
    contributes B_r (T_max - T_r) further all-zero frames, as the reference's pad_packed_sequence of the global batch would
    hold them (archs/uPIT.py:135-138).  One all-gather.  Returns (mean, var, total_count); unchanged when not parallel."""
    if not is_parallel():
        return mean, var, float(B * T)
    C = mean.numel()
    mine = torch.cat([mean.reshape(-1), var.reshape(-1), mean.new_tensor([float(B), float(T)])])
    allr = [torch.empty_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(allr, mine)
    st = torch.stack(allr)                                   # (world, 2C+2)
    Br, Tr = st[:, 2 * C:2 * C + 1], st[:, 2 * C + 1:2 * C + 2]
    n = Br * Tr                                              # (world, 1) frames of every rank's own grid
    total = (Br * Tr.max()).sum()
    n_zero = total - n.sum()                                 # frames of the global grid that no rank's grid holds: zeros
    gmean = (st[:, :C] * n).sum(0) / total
    gvar = ((n * (st[:, C:2 * C] + (st[:, :C] - gmean) ** 2)).sum(0) + n_zero * gmean ** 2) / total
    return gmean.contiguous(), gvar.contiguous(), float(total.item())


def allreduce_bn_sums(dgamma, dbeta):
    """Global per-channel sums for the BatchNorm backward (copies; the local ones stay the parameter gradients that
    the flat all-reduce sums later)."""
    both = torch.stack([dgamma, dbeta])
    if is_parallel():
        dist.all_reduce(both)
    return both[0].contiguous(), both[1].contiguous()


# ----------------------------------------------------------------------------- replicas start identical
def broadcast_model(model, src=0):
    """Make every rank's replica bit-identical to rank `src`'s: parameters (the engine's flat buffer when the model
    has one, else every parameter) and all buffers (BatchNorm running statistics, num_batches_tracked).  The
    reference has one process and therefore no such step; without it ranks that drew their initial weights from
    unseeded RNGs would apply the summed gradients to different parameters for ever."""
    if not is_parallel():
        return
    flat = getattr(model, "flat_parameters", None)
    if flat is not None:
        dist.broadcast(flat()[0], src)
    else:
        for p in model.parameters():
            dist.broadcast(p.data, src)
    for b in model.buffers():
        dist.broadcast(b, src)


def average_bn_buffers(model):
    """One set of BatchNorm running statistics on every rank.  Without sync_bn each rank's running_mean / running_var are
    moving averages of ITS OWN batches' statistics, and nothing in a training step exchanges them: a cross-validation pass
    sharded over the ranks would score every shard with different statistics, and the checkpoint would hold rank 0's --
    the printed CV loss would not be the loss of the .mdl written beside it.  The reference has one model, hence one set
    of statistics that scores and is saved (steps/train_qsub.py:124-152).  Called by the training driver after every
    epoch's last step (before the CV pass, the checkpoint and final.mdl), by all ranks together.

    Rule: running_mean / running_var <- the mean over ranks, weighted by each rank's num_batches_tracked (equal under
    EpochShards, which gives every rank the same number of steps: then a plain mean); num_batches_tracked <- the maximum.
    The moving average is linear in the batch statistics, so the averaged running_mean IS the moving average of the
    mean-over-ranks of the batch means, i.e. of the global batch's mean when the ranks hold equally many frames
    (balanced_deal), and the averaged running_var is the moving average of the mean within-rank variance -- what each
    rank's train-mode forward actually normalises with.  Averaging rather than broadcasting rank 0's (DDP's rule) uses
    every rank's data and is symmetric in the ranks.  One all-reduce of 2 x 2H + 1 floats and one of an int64; no host
    sync.  A model without such buffers, or a single process: no-op."""
    if not is_parallel():
        return
    named = list(model.named_buffers())
    stats = [b for n, b in named if n.endswith(("running_mean", "running_var"))]
    counts = [b for n, b in named if n.endswith("num_batches_tracked")]
    if not stats:
        return
    w = counts[0].detach().to(torch.float32).clamp(min=1.0).reshape(1) if counts else stats[0].new_ones(1)
    flat = torch.cat([b.detach().reshape(-1).to(torch.float32) * w for b in stats] + [w])
    dist.all_reduce(flat)
    flat = flat[:-1] / flat[-1]
    off = 0
    with torch.no_grad():
        for b in stats:
            b.copy_(flat[off:off + b.numel()].view_as(b))
            off += b.numel()
        for c in counts:
            dist.all_reduce(c, op=dist.ReduceOp.MAX)


# ----------------------------------------------------------------------------- who trains on what, per epoch
def balanced_deal(indices, lengths, world_):
    """Deal the utterances of ONE global batch to `world_` ranks: equal counts (+-1) and near-equal total frames, so
    that no rank's recurrence runs much longer than the others' (the slowest rank sets the step).  Longest first,
    in boustrophedon order (0..W-1, W-1..0, ...).  Returns a list of `world_` index lists."""
    order = sorted(indices, key=lambda i: (-lengths[i], i)) if lengths is not None else list(indices)
    out = [[] for _ in range(world_)]
    for k, i in enumerate(order):
        lap, pos = divmod(k, world_)
        out[pos if lap % 2 == 0 else world_ - 1 - pos].append(i)
    return out


class EpochShards:
    """A batch sampler for torch's DataLoader: the batches of THIS rank for one epoch of data-parallel training.

    Every epoch draws one permutation of the whole set from (seed, epoch) -- identical on all ranks --, cuts it into
    global batches of world * batch_size utterances and deals each to the ranks with balanced_deal().  Every rank
    therefore runs the SAME number of steps (the collectives inside a step always match), every utterance is used
    once per epoch, and ranks exchange utterances from epoch to epoch.  A last global batch with fewer than `world`
    utterances is topped up from the start of the permutation so that no rank is left without data (as torch's
    DistributedSampler pads)."""

    def __init__(self, n, batch_size, rank_, world_, lengths=None, seed=0, shuffle=True):
        if n <= 0 or batch_size <= 0 or not 0 <= rank_ < world_:
            raise ValueError("EpochShards: bad arguments")
        self.n, self.bs, self.rank, self.world = int(n), int(batch_size), int(rank_), int(world_)
        self.lengths = None if lengths is None else [int(v) for v in lengths]
        if self.lengths is not None and len(self.lengths) != self.n:
            raise ValueError("EpochShards: one length per utterance expected")
        self.seed, self.shuffle, self.epoch = int(seed), bool(shuffle), 0

    def set_epoch(self, epoch):
        self.epoch = int(epoch)

    def __len__(self):
        g = self.bs * self.world
        return (self.n + g - 1) // g

    def global_batches(self):
        if self.shuffle:
            gen = torch.Generator()
            gen.manual_seed(self.seed * 1000003 + self.epoch)
            perm = torch.randperm(self.n, generator=gen).tolist()
        else:
            perm = list(range(self.n))
        g = self.bs * self.world
        for k in range(len(self)):
            chunk = perm[k * g:(k + 1) * g]
            j = 0
            while len(chunk) < self.world:          # fewer utterances than ranks: top up (wraps around)
                chunk.append(perm[j % self.n])
                j += 1
            yield chunk

    def __iter__(self):
        for chunk in self.global_batches():
            yield balanced_deal(chunk, self.lengths, self.world)[self.rank]


def shard_indices_contiguous(n, rank_, world_):
    """Evaluation sets (no collective inside the pass): a contiguous shard per rank, sizes differing by at most 1;
    empty when there are more ranks than utterances."""
    lo = (n * rank_) // world_
    hi = (n * (rank_ + 1)) // world_
    return list(range(lo, hi))
