#!/usr/bin/env python3
"""First-contact kit for a multi-GPU node: the 1/2/4/8-GPU table `north_star` asks for, from bench.py's own lines.

    python3 speech-separation_amd/tools/scale_sweep.py                 # N = 1, 2, 4, 8; plain, then SEPKERN_DP_OVERLAP=1
    python3 speech-separation_amd/tools/scale_sweep.py --gpus 1,2 --steps 10 --out sweep.json

For every N (and, for N > 1, both gradient-exchange modes) it starts `python3 bench.py --gpus N --steps K --warmup W
--no-cpu-baseline --no-secondary` as a CHILD process -- this script never imports torch and never touches a GPU -- reads
the one JSON line and CHECKS what a first run on real hardware has to establish before its numbers mean anything:
  * distributed.backend == "nccl" (RCCL), distributed.world_size == N, distributed.distinct_devices == N
    (every rank computed on its own GPU), no rank fell back to per-step recurrence launches, no `lstm_fallback`;
  * the exchange-mode field says what was asked for.
Then it prints: frames/s (absolute), speed-up over N = 1, scaling efficiency, gradient all-reduce ms per step and bus GB/s
(2 (N-1)/N x bytes / time: the ring figure, per xGMI link the bound is ~153 GB/s one way), ms per step by rank (spread).
A failed check makes the exit code non-zero AFTER the table is printed, with the reason per line.

--rehearse: the same sweep on ONE GPU (every rank on cuda:0 over gloo, per-step recurrence launches) -- exercises this
script and bench.py's N > 1 path where no node exists; its numbers are not scaling figures and the checks expect
backend gloo / one device.  The N = 1 leg needs no rehearsal switch and runs in the GPU test suite.
"""
import argparse
import json
import os
import subprocess
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))


def run_bench(n, overlap, args):
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)                                  # bench.py is its own launcher (one fresh process per rank)
    env["SEPKERN_DP_OVERLAP"] = "1" if overlap else "0"
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if args.rehearse and n > 1:
        env.update(SEPKERN_BENCH_ONE_DEVICE="1", SEPKERN_DIST_BACKEND="gloo", SEPKERN_LSTM_MODE="2")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", str(args.steps), "--warmup", str(args.warmup),
           "--no-cpu-baseline", "--no-secondary"] + args.bench_args
    t0 = time.time()
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=args.timeout)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if r.returncode != 0 or len(lines) != 1:
        return None, "bench.py --gpus %d exited %d with %d JSON line(s): %s" % (n, r.returncode, len(lines), r.stderr[-1500:]), time.time() - t0
    return json.loads(lines[0]), None, time.time() - t0


def check(line, n, overlap, rehearse):
    """Reasons why this line is not a valid N-GPU data point (empty list: it is)."""
    bad = []
    if line.get("n_gpus") != n:
        bad.append("n_gpus %r != %d" % (line.get("n_gpus"), n))
    if "lstm_fallback" in line:
        bad.append("lstm_fallback: " + str(line["lstm_fallback"]))
    if n == 1:
        return bad
    d = line.get("distributed")
    if not d:
        return bad + ["no `distributed` object in the line"]
    want_backend = "gloo" if rehearse else "nccl"
    if d.get("backend") != want_backend:
        bad.append("backend %r, expected %r" % (d.get("backend"), want_backend))
    if d.get("world_size") != n:
        bad.append("world_size %r != %d" % (d.get("world_size"), n))
    want_dev = 1 if rehearse else n
    if d.get("distinct_devices") != want_dev:
        bad.append("distinct_devices %r, expected %d" % (d.get("distinct_devices"), want_dev))
    if not rehearse and any(d.get("lstm_per_step_launches_by_rank", [])):
        bad.append("ranks on per-step recurrence launches: %r" % (d.get("lstm_per_step_launches_by_rank"),))
    want_mode = "chunked-overlapped" if overlap else "single"
    if d.get("grad_allreduce") != want_mode:
        bad.append("grad_allreduce %r, expected %r" % (d.get("grad_allreduce"), want_mode))
    return bad


def table(rows):
    base = next((r["value"] for r in rows if r["n"] == 1 and r.get("value")), None)
    out = ["%-4s %-9s %14s %9s %7s %12s %10s %10s  %s" % ("N", "exchange", "frames/s", "ms/step", "x N=1", "efficiency", "allred ms", "bus GB/s", "checks")]
    for r in rows:
        if r.get("value") is None:
            out.append("%-4d %-9s %14s  %s" % (r["n"], r["mode"], "FAILED", r["problems"][0][:120]))
            continue
        sp = r["value"] / base if base else float("nan")
        out.append("%-4d %-9s %14.0f %9.3f %7.2f %11.1f%% %10s %10s  %s" % (
            r["n"], r["mode"], r["value"], r["ms_per_step"], sp, 100.0 * sp / r["n"],
            "-" if r.get("allreduce_ms") is None else "%.3f" % r["allreduce_ms"],
            "-" if r.get("busbw") is None else "%.1f" % r["busbw"],
            "ok" if not r["problems"] else "; ".join(r["problems"])))
    return "\n".join(out)


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--gpus", default="1,2,4,8", help="comma-separated GPU counts (default 1,2,4,8)")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--timeout", type=int, default=900, help="seconds per bench.py run")
    ap.add_argument("--no-overlap-leg", action="store_true", help="skip the SEPKERN_DP_OVERLAP=1 runs")
    ap.add_argument("--rehearse", action="store_true", help="N > 1 on ONE GPU over gloo (see the module docstring)")
    ap.add_argument("--out", default="", help="also write the rows (with every bench line) as JSON here")
    ap.add_argument("bench_args", nargs="*", help="extra arguments for bench.py (after --), e.g. -- --dtype bf16 --num-spk 3")
    args = ap.parse_args(argv)
    rows = []
    for n in [int(v) for v in args.gpus.split(",") if v]:
        for overlap in ([False] if (n == 1 or args.no_overlap_leg) else [False, True]):
            line, err, secs = run_bench(n, overlap, args)
            row = {"n": n, "mode": "overlap" if overlap else "single", "wall_s": round(secs, 1)}
            if line is None:
                row.update(value=None, problems=[err])
            else:
                d = line.get("distributed") or {}
                row.update(value=line["value"], ms_per_step=line["ms_per_step"], allreduce_ms=d.get("allreduce_ms_per_step"),
                           busbw=d.get("allreduce_busbw_GBs"), by_rank=d.get("ms_per_step_by_rank"),
                           problems=check(line, n, overlap, args.rehearse), line=line)
            rows.append(row)
            print("[scale_sweep] N=%d %s: %s (%.0f s)" % (n, row["mode"], "FAILED" if row["value"] is None else "%.0f frames/s" % row["value"], secs),
                  file=sys.stderr, flush=True)
    print(table(rows))
    if args.out:
        with open(args.out, "w") as f:
            json.dump(rows, f, indent=1)
    return 1 if any(r["problems"] for r in rows) else 0


if __name__ == "__main__":
    sys.exit(main())
