"""CPU: bench.py starts its own rank processes for --gpus N > 1 (the driver calls `python bench.py --gpus N ...` as well
as `python -m torch.distributed.run ... bench.py --gpus N ...`).  RC_BENCH_LAUNCH_ONLY=1 stops each rank after the
rendezvous (gloo) -- the hot path itself has no CPU fallback and is covered by the GPU tests."""
import json
import os
import socket
import subprocess
import sys
import time

from conftest import ROOT

BENCH = os.path.join(ROOT, "bench.py")


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _clean_env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env.update(extra)
    return env


def _json_lines(text):
    return [json.loads(l) for l in text.splitlines() if l.startswith("{")]


def test_bench_launches_its_own_ranks():
    p = subprocess.run([sys.executable, BENCH, "--gpus", "3", "--steps", "2", "--warmup", "1", "--dist-backend", "gloo"],
                       cwd=ROOT, env=_clean_env(RC_BENCH_LAUNCH_ONLY="1"), capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = _json_lines(p.stdout)
    assert len(lines) == 1, p.stdout                      # ONE line, from rank 0
    assert lines[0]["launch_check"] and lines[0]["n_gpus"] == 3 and lines[0]["max_rank_plus_one"] == 3.0
    assert lines[0]["env"] == {"WORLD_SIZE": "3", "MASTER_ADDR": "127.0.0.1", "LOCAL_RANK": "0"}


def test_bench_still_runs_under_torch_distributed_run():
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), BENCH, "--gpus", "2", "--steps", "2", "--warmup", "1"]
    p = subprocess.run(cmd, cwd=ROOT, env=_clean_env(RC_BENCH_LAUNCH_ONLY="1"), capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = _json_lines(p.stdout)
    assert len(lines) == 1 and lines[0]["n_gpus"] == 2


def test_a_failing_rank_fails_the_launch_quickly():
    """no GPU here: every rank refuses to run the hot path on the CPU -- the launcher must return non-zero, not hang"""
    t0 = time.time()
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "2", "--warmup", "1"], cwd=ROOT, env=_clean_env(),
                       capture_output=True, text=True, timeout=300)
    import torch
    if torch.cuda.is_available():
        return  # (on a GPU box this command is a real 2-rank run; the GPU tests cover it)
    assert p.returncode != 0
    assert "MI355X" in p.stderr and not _json_lines(p.stdout)
    assert time.time() - t0 < 120


def test_world_size_mismatch_is_refused():
    p = subprocess.run([sys.executable, BENCH, "--gpus", "4"], cwd=ROOT, env=_clean_env(WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"),
                       capture_output=True, text=True, timeout=120)
    assert p.returncode != 0 and "WORLD_SIZE=2" in p.stderr
