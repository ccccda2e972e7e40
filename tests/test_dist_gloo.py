"""The N>1 path on CPU: world_size-2 gloo run of the data-parallel helpers (sepkern.dist) that the
arch module uses on the GPU (global-norm all-reduce before the forward, one all-reduce of the flat
gradient after the backward, strided utterance sharding).  The per-rank compute here is the CPU
oracle; BatchNorm is put in eval mode because its batch statistics are per-rank by design
(DESIGN.md), so the summed shard gradients must equal the single-process global-batch gradient."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import PKG, ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _samples(n):
    rng = np.random.default_rng(5)
    out = []
    for i in range(n):
        T = int(rng.integers(4, 9))
        d = {"mix": np.abs(rng.standard_normal((T, 33))).astype(np.float32)}
        for s in range(2):
            d["source%d" % (s + 1)] = np.abs(rng.standard_normal((T, 33))).astype(np.float32)
        out.append(d)
    return out


def _model():
    from oracle import upit as OU
    torch.manual_seed(11)
    m = OU.OracleSepDNN(feat_dim=33, num_spk=2, hidden_dim=12, num_layers=2)
    with torch.no_grad():
        m.bn.running_mean.normal_(0, 0.1)
        m.bn.running_var.uniform_(0.5, 1.5)
    m.train()
    m.bn.eval()
    return m


def _loss_sum(m, samples, hidden, norm):
    """sum_b min_p L / S / norm for the given samples (oracle)."""
    from oracle import upit as OU
    loss, local_norm, _ = OU.compute_loss(m, OU.collate(samples), hidden)
    return loss * local_norm / norm


def _worker(rank, world, port, q):
    for p in (ROOT, PKG):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), SEPKERN_DIST_BACKEND="gloo")
    torch.set_num_threads(1)
    from sepkern import dist as skdist
    from oracle import upit as OU
    r, w, _ = skdist.init_from_env()
    assert (r, w) == (rank, world) and skdist.is_parallel() and skdist.world() == world
    samples = _samples(6)
    mine = [samples[i] for i in skdist.shard_indices(len(samples), rank, world)]
    m = _model()
    torch.manual_seed(100)
    h_all = (torch.randn(4, 6, 12), torch.randn(4, 6, 12))
    order_all = OU.collate_order([len(d["mix"]) for d in samples])
    # hidden rows of my utterances, in my collated order
    my_ids = skdist.shard_indices(len(samples), rank, world)
    my_order = OU.collate_order([len(samples[i]["mix"]) for i in my_ids])
    pos = {int(u): k for k, u in enumerate(order_all)}
    rows = [pos[my_ids[j]] for j in my_order]
    hidden = (h_all[0][:, rows].contiguous(), h_all[1][:, rows].contiguous())
    gnorm = float(skdist.global_norm(torch.tensor([len(d["mix"]) for d in mine], dtype=torch.int32), 33))
    loss = _loss_sum(m, mine, hidden, gnorm)
    m.zero_grad()
    loss.backward()
    flat = torch.cat([p.grad.reshape(-1) for p in m.parameters()])
    skdist.allreduce_grads(flat)
    tot = torch.tensor([float(loss)], dtype=torch.float64)
    dist.all_reduce(tot)
    # the chunked exchange (SEPKERN_DP_OVERLAP=1: sepkern.dist.GradReducer over engine.ParamLayout.grad_chunks) reduces
    # every element of [guard words | gradients] exactly once: bit-identical to the single collective
    from sepkern.engine import ParamLayout
    lay = ParamLayout(33, 66, 12, 2)
    chunks = lay.grad_chunks()
    cover = sorted((lo, hi) for _, lo, hi in chunks)
    assert cover[0][0] == 0 and cover[-1][1] == lay.total + ParamLayout.GUARD
    assert all(a[1] == b[0] for a, b in zip(cover, cover[1:]))
    assert [n for n, _, _ in chunks] == ["lin+bn", "layer1", "guard+layer0"]          # completion order of the backward pass
    buf = torch.randn(lay.total + ParamLayout.GUARD, generator=torch.Generator().manual_seed(100 + rank))
    one, two = buf.clone(), buf.clone()
    skdist.allreduce_grads(one)
    red = skdist.GradReducer()
    for _, lo, hi in chunks:
        red.chunk(two, lo, hi)
    red.finish()
    assert torch.equal(one, two) and not torch.equal(one, buf)
    q.put((rank, gnorm, float(tot), flat.numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_dp_equals_single_process_global_batch():
    sys.path.insert(0, PKG)
    from oracle import upit as OU
    from sepkern import dist as skdist
    assert sorted(skdist.shard_indices(7, 0, 2) + skdist.shard_indices(7, 1, 2)) == list(range(7))
    assert skdist.global_norm(torch.tensor([10]), 33) is None and not skdist.is_parallel()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    # single process, global batch
    samples = _samples(6)
    m = _model()
    torch.manual_seed(100)
    h_all = (torch.randn(4, 6, 12), torch.randn(4, 6, 12))
    loss, norm, _ = OU.compute_loss(m, OU.collate(samples), h_all)
    m.zero_grad()
    loss.backward()
    ref = torch.cat([p.grad.reshape(-1) for p in m.parameters()]).numpy()
    for rank, gnorm, tot, flat in res:
        assert gnorm == float(norm)
        np.testing.assert_allclose(tot, float(loss), rtol=1e-5)
        np.testing.assert_allclose(flat, ref, rtol=2e-4, atol=1e-7)
    np.testing.assert_array_equal(res[0][3], res[1][3])          # every rank holds the identical summed gradient


def _bn_worker(rank, world, port, q):
    for p in (ROOT, PKG):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from sepkern import dist as skdist
    g = torch.Generator().manual_seed(3)
    # rank 0: a grid of B = 2 utterances x T = 5 frames, rank 1: B = 3 x T = 4 -- the global batch's grid is 5 x 5 frames, of
    # which the 3 that no rank's own grid holds are zero padding (rank 1's utterances end before the global longest one)
    grids = ((2, 5), (3, 4))
    xs = [torch.randn(b * t, 5, generator=g) * (1 + i) + i for i, (b, t) in enumerate(grids)]
    x = xs[rank]
    mean, var, n = skdist.combine_bn_stats(x.mean(0), x.var(0, unbiased=False), *grids[rank])
    dg, db = skdist.allreduce_bn_sums(x.sum(0), (x * x).sum(0))
    allx = torch.cat(xs + [torch.zeros(3, 5)])
    ok = (n == 25.0 and torch.allclose(mean, allx.mean(0), atol=1e-6) and
          torch.allclose(var, allx.var(0, unbiased=False), rtol=1e-5) and
          torch.allclose(dg, allx.sum(0), rtol=1e-6) and torch.allclose(db, (allx * allx).sum(0), rtol=1e-6))
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


def test_bn_statistics_of_the_global_batch_over_two_ranks():
    """The optional BatchNorm exchange of the data-parallel path (sepkern.dist.combine_bn_stats / allreduce_bn_sums):
    per-rank (grid, mean, biased variance) combine to the statistics of the global batch's zero-padded grid, per-channel sums add up."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bn_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
    assert res == [(0, True), (1, True)]


def _eight_worker(rank, world, port, q):
    for p in (ROOT, PKG):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from sepkern import dist as skdist
    # a ragged corpus (WSJ0-2mix-shaped frame counts), one epoch of global batches of world x 4 utterances
    n, bs, C = 100, 4, 6
    lengths = [int(v) for v in np.random.default_rng(11).integers(188, 502, n)]
    shards = skdist.EpochShards(n, bs, rank, world, lengths=lengths, seed=3)
    shards.set_epoch(1)
    mine = list(shards)
    ok = len(mine) == len(shards) == -(-n // (bs * world))
    report = []
    for step, utts in enumerate(mine):
        # this rank's zero-padded (B, T) grid of C-channel frames: utterance u's valid frames are a function of u alone
        B, T = len(utts), max(lengths[u] for u in utts)
        grid = torch.zeros(B, T, C, dtype=torch.float64)
        for j, u in enumerate(utts):
            g = torch.Generator().manual_seed(1000 + u)
            grid[j, :lengths[u]] = torch.randn(lengths[u], C, generator=g, dtype=torch.float64) * (1 + u % 3) + (u % 5)
        x = grid.view(B * T, C)
        mean, var, total = skdist.combine_bn_stats(x.mean(0), x.var(0, unbiased=False), B, T)
        dg, db = skdist.allreduce_bn_sums(x.sum(0), (x * x).sum(0))
        # who holds what this step: every rank tells every rank (ids padded with -1)
        ids = torch.full((bs,), -1, dtype=torch.int64)
        ids[:B] = torch.tensor(utts)
        every = [torch.zeros_like(ids) for _ in range(world)]
        dist.all_gather(every, ids)
        report.append((step, [[int(v) for v in e if v >= 0] for e in every], mean.numpy(), var.numpy(), total, dg.numpy(), db.numpy()))
    q.put((rank, bool(ok), report if rank in (0, world - 1) else None))
    dist.barrier()
    dist.destroy_process_group()


def test_eight_rank_shards_and_global_batchnorm_statistics_on_ragged_lengths():
    """world_size 8 on CPU (the node's rank count): EpochShards + balanced_deal give every rank the same
    number of steps, every utterance exactly once (the short last global batch dealt unequally: 100 = 3 x 32 + 4, so four ranks
    are topped up), near-equal frames per rank; combine_bn_stats / allreduce_bn_sums over the eight ranks' DIFFERENT grids
    (own batch size, own longest utterance) equal the statistics of the global batch's zero-padded
    (sum of B) x (longest utterance of any rank) grid computed in one process."""
    world, n, bs, C = 8, 100, 4, 6
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_eight_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=300) for _ in procs), key=lambda t: t[0])
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res)
    lengths = [int(v) for v in np.random.default_rng(11).integers(188, 502, n)]
    rep0, rep7 = res[0][2], res[-1][2]
    seen = []
    for (step, every, mean, var, total, dg, db), other in zip(rep0, rep7):
        sizes = [len(e) for e in every]
        assert max(sizes) - min(sizes) <= 1 and min(sizes) >= 1
        if min(sizes) == bs:
            frames = [sum(lengths[u] for u in e) for e in every]
            assert max(frames) <= 1.10 * min(frames), frames                  # the slowest rank sets the step
        seen += [u for e in every for u in e]
        # one process, the global batch's grid
        Tg = max(lengths[u] for e in every for u in e)
        rows = []
        for e in every:
            for u in e:
                g = torch.Generator().manual_seed(1000 + u)
                grid = torch.zeros(Tg, C, dtype=torch.float64)
                grid[:lengths[u]] = torch.randn(lengths[u], C, generator=g, dtype=torch.float64) * (1 + u % 3) + (u % 5)
                rows.append(grid)
        allx = torch.cat(rows)
        assert total == float(allx.shape[0]) == float(sum(sizes) * Tg)
        np.testing.assert_allclose(mean, allx.mean(0).numpy(), rtol=1e-10, atol=1e-12)
        np.testing.assert_allclose(var, allx.var(0, unbiased=False).numpy(), rtol=1e-9)
        np.testing.assert_allclose(dg, allx.sum(0).numpy(), rtol=1e-10)
        np.testing.assert_allclose(db, (allx * allx).sum(0).numpy(), rtol=1e-10)
        # every rank computed the same global statistics
        np.testing.assert_array_equal(mean, other[2])
        np.testing.assert_array_equal(var, other[3])
    assert set(seen) == set(range(n)) and len(seen) == n + 4          # 4 utterances in the last global batch, 8 ranks: 4 top-ups
