#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (kernel stats + separate FETCH_SIZE / WRITE_SIZE PMC passes) into
the text summaries committed next to this script.

    python profiles/summarize.py <tag> <trace_dir> [<fetch_dir> <write_dir>]  > profiles/<tag>_summary.txt

HBM bytes follow MI355X_MICROARCH.md: counters are in KB; on gfx950 FETCH_SIZE reports half the bytes
of wide coalesced reads, so fetch bytes are shown both raw and x2-corrected; WRITE_SIZE is exact.
"""
import collections
import csv
import glob
import sys


def find(d, pat):
    f = glob.glob(d + "/**/*" + pat, recursive=True)
    return f[0] if f else None


def main():
    tag, trace = sys.argv[1], sys.argv[2]
    print("# rocprofv3 summary %s" % tag)
    extra = " --dtype bf16 --num-spk 3" if tag.endswith("_bf16") else " --no-secondary --no-power-probe"
    print("# command: rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-aux" + extra)
    rows = list(csv.DictReader(open(find(trace, "kernel_stats.csv"))))
    print("%-78s %6s %12s %12s %7s" % ("kernel", "calls", "total_ms", "avg_us", "pct"))
    for r in rows[:24]:
        print("%-78s %6s %12.3f %12.1f %7s" % (r["Name"][:78], r["Calls"], float(r["TotalDurationNs"]) / 1e6,
                                                 float(r["AverageNs"]) / 1e3, r["Percentage"]))
    if len(sys.argv) >= 5:
        print("\n# PMC passes (separate runs: --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE; bench.py --steps 2 --warmup 1)")
        for name, d in (("FETCH_SIZE", sys.argv[3]), ("WRITE_SIZE", sys.argv[4])):
            agg = collections.defaultdict(list)
            for r in csv.DictReader(open(find(d, "counter_collection.csv"))):
                agg[r["Kernel_Name"][:78]].append(float(r["Counter_Value"]))
            print("%s per launch (KB%s)" % (name, "; x2 = gfx950 wide-read correction" if name == "FETCH_SIZE" else ""))
            for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:10]:
                avg = sum(v) / len(v)
                extra = "  x2 -> %10.1f MB" % (avg * 2 / 1024) if name == "FETCH_SIZE" else "        %10.1f MB" % (avg / 1024)
                print("   %-78s n=%4d avg=%12.1f KB%s" % (k, len(v), avg, extra))


def sq_summary(d):
    """MFMA-pipe busy fraction and effective clock per kernel from an SQ/GRBM PMC pass."""
    f = find(d, "counter_collection.csv")
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(int)
    dur = collections.defaultdict(float)
    seen = set()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:78]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Dispatch_Id"] not in seen:
            seen.add(r["Dispatch_Id"])
            cnt[k] += 1
            dur[k] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    print("\n# SQ/GRBM pass: clock = GRBM_GUI_ACTIVE / 8 XCDs / duration; MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / cycles")
    for k in sorted(agg, key=lambda k: -dur[k])[:6]:
        c = agg[k]
        cyc = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
        if cyc <= 0:
            continue
        print("   %-78s n=%3d clock=%.2f GHz  MFMA busy=%.3f  MFMA flop=%.1f GF/launch" % (
            k, cnt[k], cyc / dur[k], c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / 1024.0 / cyc,
            c.get("SQ_INSTS_VALU_MFMA_MOPS_F32", 0.0) * 512 / 1e9 / cnt[k]))


if __name__ == "__main__":
    main()
    if len(sys.argv) >= 6:
        sq_summary(sys.argv[5])
