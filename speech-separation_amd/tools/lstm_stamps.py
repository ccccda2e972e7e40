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


NAMES_2M = ["dir 0: wait for 'flags up'", "dir 0: pull 7 pieces (issue + landing)", "dir 0: wait for the other tile's pieces",
            "dir 0: MFMA + partial tile", "dir 1: wait for 'flags up'", "dir 1: pull 7 pieces (issue + landing)",
            "dir 1: wait for the other tile's pieces", "dir 1: MFMA + partial tile"]
NAMES_2C = ["wait for partial tiles", "cell update + gather + h store issue", "drain vmcnt(0)", "pair arrival + flag",
            "bulk stores + gx prefetch", "hold-back + flag poll (tile-0 wave)"]


def dual(delay, spread=False):
    """Two-stream forward kernel (mode bit 28): phases of workgroup 0's MFMA wave 0 and cell wave 8, and of every
    workgroup's tile-0 cell waves."""
    torch.manual_seed(0)
    gx = torch.randn(T, B, 2, 4 * H, device="cuda") * 0.5
    whh = torch.randn(2, 4 * H, H, device="cuda") / 30
    h0, c0 = torch.randn(2, B, H, device="cuda"), torch.randn(2, B, H, device="cuda")
    lens = torch.full((B,), T, dtype=torch.int32, device="cuda")
    y, cs = torch.empty(T, B, 2 * H, device="cuda"), torch.empty(T, B, 2, H, device="cuda")
    bits = ops.lstm_variant_bits(False, 1, True, False, spread, delay, dual=True)
    for rep in range(2):
        g = gx.clone()
        ws = ops.lstm_fwd(g, whh, h0, c0, lens, y, g, cs, None, None, T, B, H, 1 | bits)
        ops.lstm_status(ws)
        st = ws[256:512].view(torch.int64)[4:18].cpu().tolist()
    print("two-stream forward, hold-back %.1f us%s (us per step, workgroup 0):" % (delay / 10.0, ", one flag per line" if spread else ""))
    for n, v in zip(NAMES_2M, st[:8]):
        print("   MFMA wave 0  %-44s %7.3f" % (n, v / 100.0 / T))
    print("   %-57s %7.3f" % ("total (MFMA wave)", sum(st[:8]) / 100.0 / T))
    for n, v in zip(NAMES_2C, st[8:14]):
        print("   cell wave 8  %-44s %7.3f" % (n, v / 100.0 / T))
    print("   %-57s %7.3f" % ("total (cell wave)", sum(st[8:14]) / 100.0 / T))
    # every workgroup's tile-0 cell wave of both directions: the diagnostic build leaves them behind the exchange blocks
    nbg, hp = (B + 15) // 16, 896 if H == 896 else 1024
    nwg = hp // 8 * nbg
    xbuf_off = 512 + ((32 * 2 * nbg * 2 * (hp // 16) * 4 + 255) // 256) * 256          # csrc/lstm.hip::ws_layout
    dbg = ws[xbuf_off + 4 * nbg * 16 * hp * 4:][:nwg * 2 * 16 * 8].view(torch.int64).view(nwg, 2, 16).cpu().double()
    ph = dbg[:, :, :6] / 100.0 / T
    print("   over all %d workgroups x 2 directions: min / median / max us per step" % nwg)
    for i, n in enumerate(NAMES_2C):
        col = ph[:, :, i].reshape(-1)
        print("     %-50s %6.3f %6.3f %6.3f" % (n, col.min(), col.median(), col.max()))


def main():
    if "--dual" in sys.argv:
        i = sys.argv.index("--dual")
        for dl in ([int(x) for x in sys.argv[i + 1].split(",")] if len(sys.argv) > i + 1 else [16]):
            dual(dl, "--spread" in sys.argv)
        return
    bf = "--bf16" in sys.argv
    # the geometry / protocol the engine ships (sepkern/engine.py: fwd_bits, bwd_bits); --legacy: the r01 one (all zero)
    if "--tagged" in sys.argv:                       # forward: tagged data instead of flags (mode bit 29), hold-back x 0.1 us
        d = int(sys.argv[sys.argv.index("--tagged") + 1])
        fbits = ops.lstm_variant_bits(False, 1, True, False, False, d, tagged=True)
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
