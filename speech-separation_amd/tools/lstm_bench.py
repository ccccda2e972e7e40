#!/usr/bin/env python3
"""Stand-alone timing of the persistent BLSTM recurrence kernels (one layer, both directions), variants interleaved
in ONE process (cdna_hip_programming.md 5.4 rule 24): us per time step, median and min over the rounds, plus the
largest difference of every variant's outputs from variant 0's (the variants are the same arithmetic: expected 0
or rounding-order level).

    python tools/lstm_bench.py [--bf16] [--T 400] [--B 32] [--H 896] [--rounds 7] [--fwd "0,0;1,0;1,2"] [--bwd "0;1"]

--fwd / --bwd: lists of "half,map,poll1,repflags" (sk_lstm_fwd / sk_lstm_bwd mode bits 17, 18..19, 20, 21; trailing
fields default to 0); 5th field: minimum number of batch groups per workgroup (mode bits 8..15); 6th: one flag per 128-byte line; 7th: hold-back of the first poll after the own flag store, x 0.1 us (0 = library's choice, 31 = none); 8th (forward): tagged data instead of flags (mode bit 29).
--ragged: lengths stepping down from T to 5T/6 over the batch, PACKED rows (the engine's layout).
"""
import argparse
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from sepkern import ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--bf16", action="store_true")
    ap.add_argument("--T", type=int, default=400)
    ap.add_argument("--B", type=int, default=32)
    ap.add_argument("--H", type=int, default=896)
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--fwd", default="0,0;0,1;0,1,1;0,1,0,1;0,1,1,1")
    ap.add_argument("--bwd", default="0,0;0,1;0,1,0,1")
    ap.add_argument("--ragged", action="store_true")
    a = ap.parse_args()
    T, B, H, bf = a.T, a.B, a.H, a.bf16
    torch.manual_seed(0)
    gx = torch.randn(T, B, 2, 4 * H, device="cuda") * 0.5
    whh = torch.randn(2, 4 * H, H, device="cuda") / 30
    h0, c0 = torch.randn(2, B, H, device="cuda"), torch.randn(2, B, H, device="cuda")
    from sepkern.packing import Packing
    lens_h = [T] * B
    if a.ragged:
        lens_h = sorted((T - (b * (T // 6)) // max(1, B - 1) for b in range(B)), reverse=True)
    pk = Packing.from_lens(lens_h, "cuda")
    lens, offs = pk.lens, (pk.offs if a.ragged else None)
    dy = torch.randn(T, B, 2 * H, device="cuda")
    def parse(spec):
        out = []
        for s in spec.split(";"):
            if s:
                v = [int(x) for x in s.split(",")]
                out.append(tuple(v + [0] * (10 - len(v))))
        return out
    fwd_vars, bwd_vars = parse(a.fwd), parse(a.bwd)

    def bits(var):
        return ops.lstm_variant_bits(bool(var[0]), var[1], bool(var[2]), bool(var[3]), bool(var[5]), var[6], tagged=bool(var[7]), split3=bool(var[8]),
                                     xl8=bool(var[9])) | (var[4] << 8)

    def run_fwd(var):
        g = gx.clone()
        y, cs = torch.zeros(T, B, 2 * H, device="cuda"), torch.zeros(T, B, 2, H, device="cuda")
        hn, cn = torch.empty(2, B, H, device="cuda"), torch.empty(2, B, H, device="cuda")
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ws = ops.lstm_fwd(g, whh, h0, c0, lens, y, g, cs, hn, cn, T, B, H, 1 | bits(var), bf16=bf, offs=offs)
        e1.record()
        torch.cuda.synchronize()
        ops.lstm_status(ws)
        return e0.elapsed_time(e1), (y, g, cs, hn, cn)

    def run_bwd(var, saved):
        y, g, cs, hn, cn = saved
        gg = g.clone()
        dh0, dc0 = torch.empty(2, B, H, device="cuda"), torch.empty(2, B, H, device="cuda")
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ws = ops.lstm_bwd(dy, whh, gg, cs, c0, lens, gg, dh0, dc0, T, B, H, 1 | bits(var), bf16=bf, offs=offs)
        e1.record()
        torch.cuda.synchronize()
        ops.lstm_status(ws)
        return e0.elapsed_time(e1), (gg, dh0, dc0)

    ref_f = ref_b = None
    tf = {v: [] for v in fwd_vars}
    tb = {v: [] for v in bwd_vars}
    df = {v: 0.0 for v in fwd_vars}
    db = {v: 0.0 for v in bwd_vars}
    for r in range(a.rounds + 1):
        for v in fwd_vars:
            ms, out = run_fwd(v)
            if ref_f is None:
                ref_f = out
            df[v] = max(df[v], max(float((x - y).abs().max()) for x, y in zip(out, ref_f)))
            if r:
                tf[v].append(ms)
        for v in bwd_vars:
            ms, out = run_bwd(v, ref_f)
            if ref_b is None:
                ref_b = out
            db[v] = max(db[v], max(float((x - y).abs().max()) for x, y in zip(out, ref_b)))
            if r:
                tb[v].append(ms)
    print("BLSTM recurrence, T=%d B=%d H=%d %s%s: us per step (median / min over %d rounds), max |diff| vs first variant"
          % (T, B, H, "bf16" if bf else "fp32", " ragged" if a.ragged else "", a.rounds))
    for v in fwd_vars:
        print("  fwd half=%d map=%d poll1=%d rep=%d gmin=%d spread=%d delay=%d tagged=%d split3=%d xl8=%d : %7.3f / %7.3f   diff %.3g"
              % (v + (1e3 * statistics.median(tf[v]) / T, 1e3 * min(tf[v]) / T, df[v])))
    for v in bwd_vars:
        print("  bwd half=%d map=%d poll1=%d rep=%d gmin=%d spread=%d delay=%d tagged=%d split3=%d xl8=%d : %7.3f / %7.3f   diff %.3g"
              % (v + (1e3 * statistics.median(tb[v]) / T, 1e3 * min(tb[v]) / T, db[v])))


if __name__ == "__main__":
    main()
