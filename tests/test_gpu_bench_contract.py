"""bench.py's output contract on a small workload: one JSON line with the keys the driver reads, at N = 1 and in a
two-rank rehearsal (both ranks on the one GPU of the test box, barrier / MAX over gloo -- the driver's own multi-GPU
runs use nccl, one rank per GPU; the code path around the timed region is the same)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
        "vs_baseline", "dtype", "data", "config", "roofline"}


def _gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")


def _line(out):
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out[-2000:]
    return json.loads(lines[0])


def test_single_gpu_line():
    _gpu()
    env = dict(os.environ, SPX_NO_BUILD="1")
    p = subprocess.run([sys.executable, "bench.py", "--steps", "5", "--warmup", "2", "--elements", "4000000", "--no-extra"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    d = _line(p.stdout)
    assert KEYS <= set(d) and "cpu_baseline" in d
    assert d["n_gpus"] == 1 and d["steps"] == 5 and d["warmup"] == 2 and d["scaling"] == "weak" and d["dtype"] == "f64"
    assert d["value"] > 0 and d["higher_is_better"] is True and d["vs_baseline"] is None
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 1 and c["value"] > 0
    assert "workload" in d["config"]


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def test_two_rank_rehearsal():
    _gpu()
    env = dict(os.environ, SPX_NO_BUILD="1", SPX_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), "bench.py", "--gpus", "2", "--steps", "5",
                        "--warmup", "2", "--elements", "4000000", "--no-extra", "--no-cpu"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=400)
    assert p.returncode == 0, (p.stdout[-1000:], p.stderr[-3000:])
    d = _line(p.stdout)
    assert KEYS <= set(d)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak"
    # whole-job value = elements of BOTH ranks / max-over-ranks time
    assert d["value"] > 0 and abs(d["value"] - 2 * 4000000 * 5 / (d["ms_per_step"] * 5 * 1e-3) / 1e9) <= 0.02 * d["value"]
