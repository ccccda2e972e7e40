#!/usr/bin/env python3
"""Sustained run of one GEMM shape per kernel variant while sampling rocm-smi (power, sclk): is the fp32 MFMA GEMM
clock/power-limited?  (diagnostic)  usage: gemm_power.py [seconds per variant]"""
import os
import subprocess
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from sepkern import ops  # noqa: E402


def sample(stop, out):
    while not stop.is_set():
        try:
            r = subprocess.run(["rocm-smi", "-d", "0", "--showpower", "--showclocks", "--showtemp"], capture_output=True, text=True, timeout=5)
            keep = [ln.strip() for ln in r.stdout.splitlines() if any(k in ln for k in ("Power", "sclk", "mclk", "junction", "Sensor edge"))]
            out.append(" | ".join(k.split(":", 1)[-1].strip() if False else k for k in keep))
        except Exception as e:  # noqa: BLE001
            out.append("rocm-smi failed: %r" % (e,))
        time.sleep(0.7)


def main():
    secs = float(sys.argv[1]) if len(sys.argv) > 1 else 6.0
    M, N, K = 12800, 7168, 1792
    A = torch.randn(M, K, device="cuda")
    B = torch.randn(N, K, device="cuda")
    C = torch.empty(M, N, device="cuda")
    for variant, name in ((1, "register-staged fp32 MFMA"), (0, "LDS-DMA fp32 MFMA"), (2, "three-way bf16 split, six products")):
        for _ in range(3):
            ops.gemm(A, B, C, M, N, K, K, K, N, transB=True, variant=variant)
        torch.cuda.synchronize()
        stop, log = threading.Event(), []
        th = threading.Thread(target=sample, args=(stop, log))
        th.start()
        t0 = time.time()
        n = 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        while time.time() - t0 < secs:
            for _ in range(50):
                ops.gemm(A, B, C, M, N, K, K, K, N, transB=True, variant=variant)
            n += 50
            torch.cuda.synchronize()
        e1.record()
        torch.cuda.synchronize()
        stop.set()
        th.join()
        ms = e0.elapsed_time(e1) / n
        print("%-28s %d launches  %.3f ms  %.1f TFLOP/s (fp32-equivalent)" % (name, n, ms, 2.0 * M * N * K / ms / 1e9), flush=True)
        for ln in log:
            print("    ", ln)
        time.sleep(2.0)


if __name__ == "__main__":
    main()
