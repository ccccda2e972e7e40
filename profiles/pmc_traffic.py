#!/usr/bin/env python3
"""profiles/pmc_traffic.json from the PMC passes collected by profiles/collect.sh.

    python profiles/pmc_traffic.py <tag>     # reads gpurun_out/prof_<tag>_{f32,bf16}_{FETCH_SIZE,WRITE_SIZE}

Per-launch averages over every launch of a kernel in the run.  Counters are KB; FETCH_SIZE is doubled
(MI355X_MICROARCH.md, HBM: gfx950 tallies 128-B read requests at 64 B), WRITE_SIZE is used as read.  Both
count the L2s' memory-side requests, so Infinity-Cache hits are included.
"""
import collections
import csv
import glob
import json
import os
import sys

def is_split(name):
    """fp32 products on the bf16 pipe: gemm_f32_kernel_split3<..>, gemm_f32_kernel_planes<..>, gemm_f32_kernel_pl3 (r06; the S6
    stream-K form was retired)."""
    return "gemm_f32_kernel_split3" in name or "gemm_f32_kernel_planes" in name or "gemm_f32_kernel_pl3" in name


KEYS = ("gemm_f32_split_kernel", "gemm_f32_kernel", "gemm_bf16_kernel", "gemm_bf16_nt_kernel", "cast_kernel", "hprev_rows_kernel", "lstm_fwd_kernel",
        "lstm_bwd_kernel", "lstm_fwd_xl8_kernel", "lstm_bwd_xl8_kernel", "split_rows_kernel", "clip_adam", "pit_pair", "pit_bwd", "bn_apply", "bn_bwd", "splitk_reduce", "colred", "sumsq")
BF16_ONLY = ("gemm_bf16_kernel", "gemm_bf16_nt_kernel", "cast_kernel", "lstm_fwd_xl8_kernel", "lstm_bwd_xl8_kernel")


def kernel_source_id(root):
    """The same id bench.py computes: sha256 over the HIP sources, so a stale file is recognised at run time."""
    sys.path.insert(0, root)
    import bench
    return bench.kernel_source_id()


def git_sha(root):
    import subprocess
    try:
        return subprocess.check_output(["git", "-C", root, "rev-parse", "HEAD"], text=True).strip()
    except (OSError, subprocess.CalledProcessError):
        return None


def main():
    tag = sys.argv[1]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    acc = collections.defaultdict(lambda: {"FETCH_SIZE": [], "WRITE_SIZE": []})
    for dt in ("f32", "bf16"):
        for c in ("FETCH_SIZE", "WRITE_SIZE"):
            files = glob.glob(os.path.join(root, "gpurun_out", "prof_%s_%s_%s" % (tag, dt, c), "**", "*counter_collection.csv"),
                              recursive=True)
            for r in csv.DictReader(open(files[0])):
                for k in KEYS:
                    name = r["Kernel_Name"]
                    hit = is_split(name) if k == "gemm_f32_split_kernel" else (k in name and not (k == "gemm_f32_kernel" and is_split(name)))
                    if hit and (dt == "f32" or k in BF16_ONLY):
                        acc[k][c].append(float(r["Counter_Value"]))
    res = {}
    for k, e in acc.items():
        f = sum(e["FETCH_SIZE"]) / max(1, len(e["FETCH_SIZE"])) * 1024
        w = sum(e["WRITE_SIZE"]) / max(1, len(e["WRITE_SIZE"])) * 1024
        res[k] = {"launches": len(e["FETCH_SIZE"]), "fetch_bytes_raw": round(f), "fetch_bytes_x2": round(2 * f),
                  "write_bytes": round(w), "hbm_bytes": round(2 * f + w)}
    doc = {"_comment": "rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes, profiles/collect.sh) "
                       "over `python3 bench.py [--dtype bf16] --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-events` on one "
                       "MI355X; per-launch averages; FETCH_SIZE doubled per MI355X_MICROARCH.md; memory-side requests of "
                       "the L2s (Infinity-Cache hits included).", "tag": tag, "kernels": res,
           "kernel_source_id": kernel_source_id(root), "git_sha": git_sha(root)}
    json.dump(doc, open(os.path.join(root, "profiles", "pmc_traffic.json"), "w"), indent=1, sort_keys=True)
    for k, v in sorted(res.items()):
        print("%-18s n=%3d  fetch x2 %8.1f MB  write %8.1f MB" % (k, v["launches"], v["fetch_bytes_x2"] / 1e6, v["write_bytes"] / 1e6))


if __name__ == "__main__":
    main()
