#!/usr/bin/env python3
"""Diagnostic: per-phase time of the persistent BLSTM kernels (needs `make -C csrc stamps` and
SEPKERN_LIB=.../libsepkern_stamps.so).  Prints 100 MHz wall-clock ticks per step for workgroup (0,0,0)."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from sepkern import ops  # noqa: E402

T, B, H = 400, 32, 896
NAMES_F = ["-", "hand-off wait (consumer waves: flags, DMA) + barrier", "MFMA", "LDS reduce barrier", "cell + h store", "drain vmcnt(0)",
           "barrier+flag+bulk stores", "gx prefetch issue"]
NAMES_B = ["wait flags+barrier", "-", "matmul (DMA ring+MFMA+reduce)", "-", "cell backward", "dG store + drain",
           "barrier+flag+dgx stores", "saved-activation loads issue"]


def stamps(ws):
    return ws[256:512].view(torch.int64)[4:12].cpu().tolist()   # the per-launch control block


def main():
    bf = "--bf16" in sys.argv
    # the geometry / protocol the engine ships (sepkern/engine.py: fwd_bits, bwd_bits); --legacy: the r01 one (all zero)
    if "--tagged" in sys.argv:                       # forward: tagged data instead of flags (mode bit 29), hold-back x 0.1 us
        d = int(sys.argv[sys.argv.index("--tagged") + 1])
        fbits = ops.lstm_variant_bits(False, 1, True, False, False, d, tagged=True)
        bbits = ops.lstm_variant_bits(False, 1, False, False, False, 31)
    elif "--split3" in sys.argv:                     # forward: exact three-way bf16 split of the product (mode bit 28), flags
        fbits = ops.lstm_variant_bits(False, 1, True, False, False, 0, split3=True)
        bbits = ops.lstm_variant_bits(False, 1, False, False, False, 31)
    elif "--legacy" in sys.argv:
        fbits = bbits = 0
    else:
        fbits = ops.lstm_variant_bits(False, 1, True, False, bf, 0)
        bbits = ops.lstm_variant_bits(False, 1, False, False, False, 31)
    torch.manual_seed(0)
    gx = torch.randn(T, B, 2, 4 * H, device="cuda") * 0.5
    whh = torch.randn(2, 4 * H, H, device="cuda") / 30
    h0, c0 = torch.randn(2, B, H, device="cuda"), torch.randn(2, B, H, device="cuda")
    lens = torch.full((B,), T, dtype=torch.int32, device="cuda")
    y, cs = torch.empty(T, B, 2 * H, device="cuda"), torch.empty(T, B, 2, H, device="cuda")
    for rep in range(2):
        g = gx.clone()
        ws = ops.lstm_fwd(g, whh, h0, c0, lens, y, g, cs, None, None, T, B, H, 1 | fbits, bf16=bf)
        ops.lstm_status(ws)
        sf = stamps(ws)
        dy = torch.randn(T, B, 2 * H, device="cuda")
        ws = ops.lstm_bwd(dy, whh, g, cs, c0, lens, g, None, None, T, B, H, 1 | bbits, bf16=bf)
        ops.lstm_status(ws)
        sb = stamps(ws)
    print("forward  (us per step, workgroup 0):")
    for n, v in zip(NAMES_F, sf):
        print("   %-34s %7.3f" % (n, v / 100.0 / T))
    print("   %-34s %7.3f" % ("total", sum(sf) / 100.0 / T))
    print("backward (us per step, workgroup 0):")
    for n, v in zip(NAMES_B, sb):
        print("   %-34s %7.3f" % (n, v / 100.0 / T))
    print("   %-34s %7.3f" % ("total", sum(sb) / 100.0 / T))


if __name__ == "__main__":
    main()
