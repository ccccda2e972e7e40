#!/usr/bin/env python3
"""Summary of profiles/stream_pmc.sh's passes: per kernel, per-launch averages of every counter and the ratios that say
what binds a streaming kernel (VALU issue time against wall time, LDS activity, HBM bytes against algorithmic bytes).

    python profiles/stream_pmc.py <tag>  > profiles/<tag>_stream_pmc.txt

Units (MI355X_MICROARCH.md): SQ_ACTIVE_INST_* and SQ_WAVE_CYCLES count quad-cycles (x4 = shader cycles) summed over all
SIMDs / waves; GRBM_GUI_ACTIVE counts cycles on each of the 8 XCDs; FETCH_SIZE is KB and doubled on gfx950."""
import collections
import csv
import glob
import os
import sys

KERNELS = ("::stft_kernel", "::istft_kernel", "pit_pair", "pit_bwd")
SIMDS = 1024


def main():
    tag = sys.argv[1]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    for d in sorted(glob.glob(os.path.join(root, "gpurun_out", "prof_%s_stream_*" % tag))):
        if not os.path.isdir(d):
            continue
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                for k in KERNELS:
                    if k in r["Kernel_Name"]:
                        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                for k in KERNELS:
                    if k in r["Kernel_Name"]:
                        dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for k in KERNELS:
        if k not in acc:
            continue
        c = {n: sum(v) / len(v) for n, v in acc[k].items()}
        us = sorted(dur[k])[len(dur[k]) // 2]
        print("%s   (median launch under the profiler: %.1f us, %d launches per pass)" % (k, us, len(acc[k]["SQ_WAVES"])))
        for n in sorted(c):
            print("    %-24s %16.0f" % (n, c[n]))
        clk = c.get("GRBM_GUI_ACTIVE", 0) / 8.0            # shader cycles of the launch
        if clk:
            print("    effective clock                  %.2f GHz" % (clk / us / 1e3))
            if "SQ_ACTIVE_INST_VALU" in c:
                print("    VALU issue busy                  %.1f %% of SIMD cycles  (SQ_ACTIVE_INST_VALU x 4 / (%d SIMDs x cycles))"
                      % (100.0 * c["SQ_ACTIVE_INST_VALU"] * 4 / (SIMDS * clk), SIMDS))
            if "SQ_ACTIVE_INST_LDS" in c:
                print("    LDS instruction busy             %.1f %% of SIMD cycles" % (100.0 * c["SQ_ACTIVE_INST_LDS"] * 4 / (SIMDS * clk)))
            if "SQ_LDS_IDX_ACTIVE" in c:
                print("    LDS array active                 %.1f %% of CU cycles, bank-conflict cycles %.1f %% of those"
                      % (100.0 * c["SQ_LDS_IDX_ACTIVE"] / (256 * clk), 100.0 * c.get("SQ_LDS_BANK_CONFLICT", 0) / max(1.0, c["SQ_LDS_IDX_ACTIVE"])))
        if "SQ_INSTS_VALU" in c and "SQ_WAVES" in c:
            print("    VALU / LDS / SALU instructions per wave   %.0f / %.0f / %.0f"
                  % (c["SQ_INSTS_VALU"] / c["SQ_WAVES"], c.get("SQ_INSTS_LDS", 0) / c["SQ_WAVES"], c.get("SQ_INSTS_SALU", 0) / c["SQ_WAVES"]))
        if "FETCH_SIZE" in c:
            hbm = 2 * c["FETCH_SIZE"] * 1024 + c.get("WRITE_SIZE", 0) * 1024
            print("    HBM-side bytes per launch        %.1f MB (fetch x2 %.1f + write %.1f)  = %.2f TB/s at the launch time above"
                  % (hbm / 1e6, 2 * c["FETCH_SIZE"] * 1024 / 1e6, c.get("WRITE_SIZE", 0) * 1024 / 1e6, hbm / us / 1e6))
        print()


if __name__ == "__main__":
    main()
