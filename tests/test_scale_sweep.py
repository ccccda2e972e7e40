"""tools/scale_sweep.py (the first-contact kit for a multi-GPU node): its checks and its table on synthetic bench lines
(CPU), its behaviour when no GPU is there (CPU), and its N = 1 leg plus a one-device rehearsal of N = 2 on the GPU."""
import json
import os
import subprocess
import sys

import pytest

from conftest import PKG, ROOT

TOOL = os.path.join(PKG, "tools", "scale_sweep.py")


def _tool():
    sys.path.insert(0, os.path.join(PKG, "tools"))
    import scale_sweep
    return scale_sweep


def _line(n, value, backend="nccl", devices=None, mode="single", fallback=None, per_step=None):
    d = {"n_gpus": n, "value": value, "ms_per_step": 12800.0 * n / value * 1e3}
    if n > 1:
        d["distributed"] = {"backend": backend, "world_size": n, "distinct_devices": n if devices is None else devices,
                            "grad_allreduce": mode, "lstm_per_step_launches_by_rank": per_step or [0] * n,
                            "allreduce_ms_per_step": 2.5, "allreduce_busbw_GBs": 134.0, "ms_per_step_by_rank": [35.0] * n}
    if fallback:
        d["lstm_fallback"] = fallback
    return d


def test_checks_catch_what_a_first_node_run_has_to_establish():
    ss = _tool()
    assert ss.check(_line(1, 366e3), 1, False, False) == []
    assert ss.check(_line(8, 2.6e6), 8, False, False) == []
    assert ss.check(_line(8, 2.6e6, mode="chunked-overlapped"), 8, True, False) == []
    assert any("backend" in p for p in ss.check(_line(2, 7e5, backend="gloo"), 2, False, False))
    assert any("distinct_devices" in p for p in ss.check(_line(4, 1e6, devices=1), 4, False, False))
    assert any("lstm_fallback" in p for p in ss.check(_line(1, 3e5, fallback="per-step launches"), 1, False, False))
    assert any("per-step" in p for p in ss.check(_line(2, 5e5, per_step=[0, 1]), 2, False, False))
    assert any("grad_allreduce" in p for p in ss.check(_line(2, 7e5), 2, True, False))
    assert any("no `distributed`" in p for p in ss.check({"n_gpus": 2, "value": 1.0}, 2, False, False))
    # the rehearsal expects gloo on one device
    assert ss.check(_line(2, 3e5, backend="gloo", devices=1, per_step=[1, 1]), 2, False, True) == []
    rows = [{"n": 1, "mode": "single", "value": 366e3, "ms_per_step": 35.0, "problems": []},
            {"n": 8, "mode": "single", "value": 2.6e6, "ms_per_step": 39.4, "allreduce_ms": 2.5, "busbw": 134.0, "problems": []},
            {"n": 4, "mode": "overlap", "value": None, "problems": ["bench.py --gpus 4 exited 1"]}]
    text = ss.table(rows)
    assert "7.10" in text and "88.8%" in text and "FAILED" in text and "134.0" in text


def test_without_a_gpu_the_sweep_reports_failure_and_a_table():
    """Here (no GPU) every bench.py child fails at its first device call: the sweep still prints its table, marks the row
    FAILED and exits non-zero -- it never hangs and never touches a GPU itself."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: covered by the gpu-marked test")
    r = subprocess.run([sys.executable, TOOL, "--gpus", "1", "--steps", "1", "--warmup", "0", "--timeout", "300"], cwd=ROOT,
                       capture_output=True, text=True, timeout=400)
    assert r.returncode == 1 and "FAILED" in r.stdout and "frames/s" in r.stdout.splitlines()[0]


@pytest.mark.gpu
def test_n1_leg_and_one_device_rehearsal(tmp_path):
    out = str(tmp_path / "sweep.json")
    r = subprocess.run([sys.executable, TOOL, "--gpus", "1,2", "--steps", "2", "--warmup", "1", "--rehearse", "--out", out, "--",
                        "--hidden", "64", "--layers", "2", "--frames", "40"], cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr[-3000:]
    rows = json.load(open(out))
    assert [(x["n"], x["mode"]) for x in rows] == [(1, "single"), (2, "single"), (2, "overlap")]
    assert all(x["problems"] == [] and x["value"] > 0 for x in rows)
    assert rows[1]["line"]["distributed"]["backend"] == "gloo" and rows[2]["line"]["distributed"]["grad_allreduce"] == "chunked-overlapped"
    assert len(r.stdout.strip().splitlines()) == 4
