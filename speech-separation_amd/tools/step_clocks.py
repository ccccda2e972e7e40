#!/usr/bin/env python3
"""Shader clock and package power (sysfs hwmon, sampled every 20 ms) while the phases of the fp32 training step run on their own:
the forward recurrence, the backward recurrence alone, the backward recurrence HOSTING a weight-gradient product on a side stream
(co-resident, and with mode bit 17 = exclusive), and the products alone.  Question: how much of what hosting costs the recurrence's
chain (6.6 -> 10.5 us per step) is the clock the power cap leaves it?

    python tools/step_clocks.py [--seconds 2.0]
"""
import argparse
import glob
import os
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from sepkern import ops  # noqa: E402
from sepkern.packing import Packing  # noqa: E402

T, B, H = 400, 32, 896
R = T * B


def read(path):
    try:
        with open(path) as f:
            return float(f.read().strip())
    except (OSError, ValueError):
        return None


class Sampler:
    def __init__(self):
        self.hw = sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"))
        self.rows, self.stop = [], threading.Event()

    def __enter__(self):
        self.rows, self.stop = [], threading.Event()

        def loop():
            while not self.stop.is_set():
                row = []
                for h in self.hw:
                    pw = read(h + "/power1_average")
                    pw = read(h + "/power1_input") if pw is None else pw
                    row.append((pw, read(h + "/freq1_input")))
                self.rows.append(row)
                time.sleep(0.02)
        self.th = threading.Thread(target=loop)
        self.th.start()
        return self

    def __exit__(self, *exc):
        self.stop.set()
        self.th.join()

    def summary(self):
        late = self.rows[len(self.rows) // 3:]
        best, best_w = None, -1.0
        for i in range(len(self.hw)):
            ws = [r[i][0] for r in late if r[i][0] is not None]
            if ws and sum(ws) / len(ws) > best_w:
                best, best_w = i, sum(ws) / len(ws)
        if best is None:
            return "no sensors"
        fr = sorted(r[best][1] / 1e6 for r in late if r[best][1] is not None)
        if not fr:
            return "%.0f W" % (best_w / 1e6)
        return "%4.0f W  sclk mean %4.0f  p10 %4.0f  p50 %4.0f  p90 %4.0f MHz  (%d samples)" % (
            best_w / 1e6, sum(fr) / len(fr), fr[len(fr) // 10], fr[len(fr) // 2], fr[(9 * len(fr)) // 10], len(fr))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=2.0)
    a = ap.parse_args()
    torch.manual_seed(0)
    gx = torch.randn(T, B, 2, 4 * H, device="cuda") * 0.5
    whh = torch.randn(2, 4 * H, H, device="cuda") / 30
    h0, c0 = torch.randn(2, B, H, device="cuda"), torch.randn(2, B, H, device="cuda")
    pk = Packing.from_lens([T] * B, "cuda")
    dy = torch.randn(T, B, 2 * H, device="cuda")
    fbits = ops.lstm_variant_bits(False, 1, True, False, False, 0, tagged=False, split3=True)
    bbits = ops.lstm_variant_bits(False, 1, False, False, False, 31)
    g = gx.clone()
    y, cs = torch.zeros(T, B, 2 * H, device="cuda"), torch.zeros(T, B, 2, H, device="cuda")
    hn, cn = torch.empty(2, B, H, device="cuda"), torch.empty(2, B, H, device="cuda")
    dgx = torch.empty_like(g)
    dh0, dc0 = torch.empty(2, B, H, device="cuda"), torch.empty(2, B, H, device="cuda")

    def fwd():
        g.copy_(gx)
        return ops.lstm_fwd(g, whh, h0, c0, pk.lens, y, g, cs, hn, cn, T, B, H, 1 | fbits, bf16=False)

    def bwd(excl=False):
        return ops.lstm_bwd(dy, whh, g, cs, c0, pk.lens, dgx, dh0, dc0, T, B, H, 1 | bbits | (0x20000 if excl else 0), bf16=False)
    ws = fwd()
    torch.cuda.synchronize()
    ops.lstm_status(ws)
    # the hosted product: dW_ih of a layer on planes (7168 x 1792 x 12800), as the engine launches it beside a recurrence
    Apl = ops.split_rows(torch.randn(R, 8 * H, device="cuda"))
    Bpl = ops.split_rows(torch.randn(R, 2 * H, device="cuda"))
    Cw = torch.empty(8 * H, 2 * H, device="cuda")
    An, Bn, Cn = torch.randn(R, 2 * H, device="cuda"), torch.randn(8 * H, 2 * H, device="cuda"), torch.empty(R, 8 * H, device="cuda")
    gflop_w, gflop_n = 2.0 * 8 * H * 2 * H * R / 1e9, 2.0 * R * 8 * H * 2 * H / 1e9

    def wgrad():
        ops.gemm_pl3_tn(Apl, Bpl, Cw, 8 * H, 2 * H, R, splitk=0, ws_tag="clk_side")

    def proj():
        ops.gemm(An, Bn, Cn, R, 8 * H, 2 * H, 2 * H, 2 * H, 8 * H, transB=True)
    side = torch.cuda.Stream()
    smp = Sampler()

    def phase(name, main_fn, side_fn, main_flop=None, side_flop=None):
        for _ in range(2):
            if main_fn:
                main_fn()
            if side_fn:
                with torch.cuda.stream(side):
                    side_fn()
        torch.cuda.synchronize()
        em, es, nm, ns = [], [], 0, 0
        t0 = time.time()
        with smp:
            while time.time() - t0 < a.seconds:
                for _ in range(8):
                    if main_fn:
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record()
                        main_fn()
                        e1.record()
                        em.append((e0, e1))
                    if side_fn:
                        with torch.cuda.stream(side):
                            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                            e0.record()
                            side_fn()
                            e1.record()
                            es.append((e0, e1))
                torch.cuda.synchronize()
        out = "%-58s" % name
        if em:
            ms = sorted(x.elapsed_time(y_) for x, y_ in em)[len(em) // 2]
            out += "  main %7.3f ms" % ms + ("  %5.2f us/step" % (1e3 * ms / T) if main_flop is None else "  %5.1f TFLOP/s" % (main_flop / ms))
        if es:
            ms = sorted(x.elapsed_time(y_) for x, y_ in es)[len(es) // 2]
            out += "  side %7.3f ms  %5.1f TFLOP/s" % (ms, side_flop / ms)
        print(out + "  | " + smp.summary(), flush=True)

    phase("idle-ish (nothing launched)", None, None)
    phase("forward recurrence (split product) alone", fwd, None)
    phase("backward recurrence alone", bwd, None)
    phase("backward recurrence + dW_ih on planes beside it (co-resident)", bwd, wgrad, None, gflop_w)
    phase("backward recurrence (exclusive) + dW_ih on planes", lambda: bwd(True), wgrad, None, gflop_w)
    phase("dW_ih on planes alone (side stream)", None, wgrad, None, gflop_w)
    phase("input projection (variant 9) alone", proj, None, gflop_n, None)
    ops.lstm_status(ops.lstm_ws(T, B, H))


if __name__ == "__main__":
    main()
