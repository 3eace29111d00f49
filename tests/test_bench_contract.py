"""bench.py's contract (the driver depends on it): ONE JSON line on stdout with the metric, the roofline of the dominant
kernel and -- unless switched off -- the CPU baseline; `--gpus N` without a launcher starts the ranks itself."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, timeout=600):
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))     # --sharded at world size 1 rendezvous
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=timeout, cwd=ROOT, env=env)
    return p.returncode, p.stdout.decode(errors="replace"), p.stderr.decode(errors="replace")


def test_bench_without_a_gpu_fails_loudly():
    """No CPU path: on a box without an MI355X the bench refuses to run (never a silent fallback)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    rc, out, err = _run(["--gpus", "1", "--steps", "4", "--warmup", "1", "--no-cpu-baseline"], timeout=300)
    assert rc != 0 and "{" not in out and "MI355X" in err


def test_bench_gpus_n_starts_ranks_and_relays_failure():
    """`--gpus 2` with no launcher around it: the parent starts two ranks under torch.distributed.run; without GPUs they
    fail, and so does the parent (exit code relayed, no JSON line)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    rc, out, err = _run(["--gpus", "2", "--steps", "4", "--warmup", "1", "--no-cpu-baseline"], timeout=300)
    assert rc != 0 and '"metric"' not in out


@pytest.mark.gpu
@pytest.mark.parametrize("extra", [[], ["--sharded"]])
def test_bench_json_line(extra):
    """The single-GPU line at a short K (and the row-sharded step at world size 1, which captures graphs after RCCL
    collectives have run): exit code 0, exactly one JSON line, the keys the driver and the judge read."""
    rc, out, err = _run(["--gpus", "1", "--steps", "8", "--warmup", "2", "--no-cpu-baseline", "--adam-steps", "0"] + extra)
    assert rc == 0, err[-2000:]
    lines = [ln for ln in out.splitlines() if ln.strip()]
    assert len(lines) == 1, out
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 8 and d["value"] > 0 and d["higher_is_better"] is True
    assert abs(d["value"] - 8192 * 1e3 / d["ms_per_step"]) <= 1e-3 * d["value"]
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) <= 1e-6 and 0.0 < r["frac"] < 1.0
    assert r["traffic"] is None or r["traffic"] > 0


@pytest.mark.gpu
def test_bench_two_ranks_rehearsal_on_one_gpu():
    """`--gpus 2` started by bench.py itself, the two ranks sharing the one GPU with gloo exchanges through host memory
    (--dist-backend gloo): the multi-rank control flow of the benchmark -- launcher, rendezvous, row-sharded step with
    HIP kernels, barriers, max-over-ranks timing, the replicas reference point, rank 0's JSON line relayed by the
    parent -- runs end to end before a multi-GPU node ever sees it."""
    rc, out, err = _run(["--gpus", "2", "--dist-backend", "gloo", "--steps", "8", "--warmup", "2", "--no-cpu-baseline",
                         "--adam-steps", "0"], timeout=900)
    assert rc == 0, err[-2000:]
    lines = [ln for ln in out.splitlines() if ln.strip()]
    assert len(lines) == 1, out
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 2 * 8192 and "rehearsal" in d["config"]
    assert d["value"] > 0 and "replicas_no_exchange" in d and d["scaling"] == "weak"
