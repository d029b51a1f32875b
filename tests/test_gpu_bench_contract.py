"""The bench.py contract the driver relies on: one JSON line on stdout with the agreed keys, for the plain single-GPU
launch and for the torch.distributed launch (one rank here; RCCL init, chunked solve + asynchronous all-gather path)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
        "dtype", "data", "config", "roofline"}
SMALL = ["--size", "256", "--frames", "10", "--steps", "2", "--warmup", "1", "--no-variants"]   # frames given: same workload at any N


def run(cmd):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines                      # exactly ONE line on stdout
    return json.loads(lines[0])


def check(d, n_gpus):
    assert KEYS <= set(d)
    assert d["metric"] == "frame-pairs/sec" and d["unit"] == "frame-pairs/s" and d["higher_is_better"] is True
    assert d["n_gpus"] == n_gpus and d["steps"] == 2 and d["warmup"] == 1 and d["scaling"] == "weak"
    assert d["dtype"] == "f64" and d["data"] == "synthetic" and d["vs_baseline"] is None
    assert d["value"] > 0 and d["ms_per_step"] > 0 and d["config"]["converged"] is True
    assert "model" not in d["config"] and "workload" in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and r["unit"] == "GB/s"
    assert r["achieved"] > 0 and r["frac"] == pytest.approx(r["achieved"] / r["peak"])
    # an HBM fraction: bytes the launch has to move over its time (the per-sweep work rate is reported separately)
    assert r["achieved"] == pytest.approx(r["bytes_per_launch"] / (r["avg_launch_us"] * 1e-6) / 1e9)
    assert r["frac"] < 1 and r["effective_per_sweep"]["rate"] >= r["achieved"] * 0.999


def test_single_gpu_line_with_cpu_baseline():
    d = run([sys.executable, "bench.py"] + SMALL)
    check(d, 1)
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 1 and c["value"] > 0 and c["unit"] == "frame-pairs/s" and c["sample"]
    assert c["host_cores"] >= 1 and c["all_cores"]["cores"] >= 1 and c["all_cores"]["value"] > 0
    e = d["end_to_end"]
    assert e["unit"] == "frame-pairs/s" and 0 < e["value"] and e["converged"] is True
    w = d["roofline"]["whole_solve"]
    assert 0 < w["frac"] < 1 and w["bytes_per_step"] > 0 and w["frac"] == pytest.approx(w["achieved"] / 8000.0)
    assert d["value"] / c["value"] > 10                # sanity only: the ratio says nothing about kernel quality


def test_distributed_launch_line():
    d = run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
             "--master-port", "29517", "bench.py", "--gpus", "1", "--force-dist", "--no-cpu-baseline", "--gather-chunks", "3"] + SMALL)
    check(d, 1)
    assert d["config"]["allgather"] is True and d["config"]["gather_chunks"] == 3 and d["config"]["chunk_sizes"] == [3, 3, 3]
