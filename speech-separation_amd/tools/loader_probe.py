#!/usr/bin/env python3
"""Where does an epoch of steps/train_qsub.py go?  Rates of the pieces, each alone, on an existing data dir
(tools/stage_walls.py --keep leaves one): the loader's workers, the loader + GPU staging (sepkern.data.Prefetcher), the
training step on batches that are already staged, and all of it together.

    python speech-separation_amd/tools/loader_probe.py <data-dir> [--wav-input] [--workers 8] [--hidden 896 --layers 3]
"""
import argparse
import contextlib
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.abspath(os.path.join(HERE, ".."))
for p in (PKG, os.path.join(PKG, "archs"), os.path.join(PKG, "steps")):
    sys.path.insert(0, p)

import torch  # noqa: E402
from torch.utils.data import DataLoader  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("data_dir")
    ap.add_argument("--wav-input", action="store_true")
    ap.add_argument("--workers", type=int, default=8)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--hidden", type=int, default=896)
    ap.add_argument("--layers", type=int, default=3)
    ap.add_argument("--depth", type=int, default=2)
    ap.add_argument("--threads", type=int, default=0, help="torch.set_num_threads in this process (0: leave)")
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--switch", type=float, default=0.0, help="sys.setswitchinterval (seconds; 0: leave at 5 ms)")
    a = ap.parse_args()
    import uPIT
    import train_qsub as tq
    from sepkern.data import Prefetcher
    from sepkern.optim import ClipAdam
    torch.cuda.set_device(0)
    if a.threads:
        torch.set_num_threads(a.threads)
    if a.switch:
        sys.setswitchinterval(a.switch)
    print("torch threads %d, pin %s" % (torch.get_num_threads(), os.environ.get("SEPKERN_PREFETCH_PIN", "1")), flush=True)
    ds = uPIT.WavTrainSet(a.data_dir) if a.wav_input else uPIT.TrainSet(a.data_dir)
    loader = DataLoader(ds, batch_size=a.batch, shuffle=True, collate_fn=ds.collator, num_workers=a.workers,
                        persistent_workers=a.workers > 0, prefetch_factor=2 if a.workers > 0 else None)

    def rate(tag, it, per_batch=None):
        t0, n = time.perf_counter(), 0
        for b in it:
            if per_batch:
                per_batch(b)
            n += 1
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print("%-52s %4d batches in %6.2f s = %6.1f ms per batch" % (tag, n, dt, 1e3 * dt / n), flush=True)
        return dt / n

    rate("loader alone (first pass: workers start)", loader)
    rate("loader alone", loader)
    pf = Prefetcher(loader, torch.device("cuda", 0), depth=a.depth)
    rate("loader + staging on the GPU (Prefetcher)", pf)
    staged = list(pf)
    torch.manual_seed(0)
    with contextlib.redirect_stdout(sys.stderr):
        model = uPIT.SepDNN(0, num_spk="2", hidden_dim=str(a.hidden), num_layers=str(a.layers))
    model.cuda()
    model.train()
    opt = ClipAdam(model, lr=1e-3, max_norm=0.25)
    # how long does the HOST need to enqueue a step?  (no synchronisation inside the loop: the host runs ahead of the GPU)
    for rep_ in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for b in staged:
            loss, norm = uPIT.compute_loss(model, 0, b)
            loss.backward()
            opt.step()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print("host enqueue %.1f ms per step, GPU done %.1f ms per step" % (1e3 * (t1 - t0) / len(staged), 1e3 * (t2 - t0) / len(staged)), flush=True)
    plan = (("train_epoch on resident staged batches (warm-up)", staged), ("train_epoch on resident staged batches", staged),
            ("train_epoch through the Prefetcher", pf), ("train_epoch through the Prefetcher (again)", pf),
            ("train_epoch on the plain loader (reference loop)", loader))
    if a.quick:
        plan = (plan[0], plan[2], plan[3])
    for tag, batches in plan:
        t0 = time.perf_counter()
        acc = tq.train_epoch(uPIT, model, opt, batches, 0, 1, False)
        float(acc[0])
        dt = time.perf_counter() - t0
        print("%-52s %4d batches in %6.2f s = %6.1f ms per batch" % (tag, len(batches), dt, 1e3 * dt / len(batches)), flush=True)


if __name__ == "__main__":
    main()
