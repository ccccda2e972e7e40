#!/usr/bin/env python3
"""Sanity run, not a benchmark: N optimisation steps of the 3x896 uPIT model on ONE fixed synthetic batch (the bench's
WSJ0-2mix-shaped 32 x 400 frames) with the bench's optimiser settings, printing the loss every few steps -- the loss must
fall monotonically-ish and stay finite, in fp32 and in bf16 arithmetic, with the engine's co-scheduling on."""
import contextlib
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "speech-separation_amd"), os.path.join(ROOT, "speech-separation_amd", "archs")):
    sys.path.insert(0, p)
import bench  # noqa: E402  (make_batch)
import uPIT  # noqa: E402
from sepkern import ops, synth  # noqa: E402
from sepkern.optim import ClipAdam  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    every = int(sys.argv[2]) if len(sys.argv) > 2 else 20          # print the loss of every `every`-th step
    hseed = int(sys.argv[3]) if len(sys.argv) > 3 else 1234        # seed of the per-step h0 / c0 draws
    only = sys.argv[4] if len(sys.argv) > 4 else ""               # 'fp32' / 'bf16': that arithmetic only
    for dtype in ("fp32", "bf16"):
        if only and dtype != only:
            continue
        torch.manual_seed(0)
        with contextlib.redirect_stdout(sys.stderr):
            model = uPIT.SepDNN(0, num_spk="2", hidden_dim="896", num_layers="3", dtype=dtype)
        model.cuda()
        model.train()
        model.hidden_generator = torch.Generator(device="cuda")
        model.hidden_generator.manual_seed(hseed)
        opt = ClipAdam(model, lr=1e-3, max_norm=0.25)
        mix, srcs, pk, _, _ = bench.make_batch(torch, ops, synth, 32, 400, 2, 0)
        out = []
        acc = torch.zeros(1, device="cuda")
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            loss, norm = uPIT.compute_loss_packed(model, mix, srcs, pk)
            loss.backward()
            opt.step()
            if i % every == 0 or i == steps - 1:
                out.append((i, loss.detach().clone()))          # no host sync inside the timed loop
        torch.cuda.synchronize()
        secs = time.perf_counter() - t0
        v = float(loss.detach())
        assert v == v and v > 0
        opt.check()
        frames = pk.R * steps
        print("%s  loss by step  %s" % (dtype, "  ".join("%d:%.5f" % (i, float(l)) for i, l in out)), flush=True)
        # the bench times 20 steps (0.8 s); this is the same step sustained for `steps` steps (clocks settle)
        print("%s  sustained: %d steps in %.2f s = %.3f ms/step = %.1f frames/s" % (dtype, steps, secs, 1e3 * secs / steps,
                                                                             frames / secs), flush=True)


if __name__ == "__main__":
    main()
