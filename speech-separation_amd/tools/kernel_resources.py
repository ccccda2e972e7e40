#!/usr/bin/env python3
"""Registers, spills, LDS and occupancy of every kernel in a HIP source, as the compiler reports them
(-Rpass-analysis=kernel-resource-usage; no GPU needed).

    python tools/kernel_resources.py csrc/lstm.hip [-DSK_... ...] [--grep lstm_bwd]
"""
import os
import re
import subprocess
import sys
import tempfile


def main():
    args = sys.argv[1:]
    pat = None
    if "--grep" in args:
        i = args.index("--grep")
        pat = args[i + 1]
        del args[i:i + 2]
    src, defs = args[0], args[1:]
    with tempfile.TemporaryDirectory() as tmp:
        cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-fno-gpu-rdc",
               "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", os.path.join(tmp, "x.o")] + defs
        txt = subprocess.run(cmd, capture_output=True, text=True).stderr
    rows, cur = {}, None
    for line in txt.splitlines():
        m = re.search(r"remark: Function Name: (\S+)", line)
        if m:
            cur = m.group(1)
            rows[cur] = {}
            continue
        m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\S+) \[-Rpass", line)
        if m and cur:
            rows[cur][m.group(1).strip()] = m.group(2)
    names = subprocess.run(["c++filt"], input="\n".join(rows), capture_output=True, text=True).stdout.splitlines()
    print("%-100s %5s %5s %7s %6s %7s %4s" % ("kernel", "VGPR", "AGPR", "scratch", "vspill", "LDS", "occ"))
    for mangled, name in zip(rows, names):
        name = name.replace("(anonymous namespace)::", "")
        name = re.sub(r"\(.*\)$", "", name)
        if pat and pat not in name:
            continue
        r = rows[mangled]
        print("%-100s %5s %5s %7s %6s %7s %4s" % (name[:100], r.get("VGPRs"), r.get("AGPRs"), r.get("ScratchSize"),
                                                  r.get("VGPRs Spill"), r.get("LDS Size"), r.get("Occupancy")))


if __name__ == "__main__":
    main()
