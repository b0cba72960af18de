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


def test_secondary_leg_rules_and_the_launcher_that_captures_rank_zero(monkeypatch):
    """host logic of the `secondary` object of the contract line: only the default contract workload carries the other configs;
    a leg contributes a summary, never its whole line; the launcher can hand rank 0's output back (it runs a second rank group
    for the sharded NeuMF leg and merges the two lines into ONE)"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", BENCH)
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    a = bench.parse()
    assert bench.secondary_wanted(a)
    for argv in (["--workload", "neumf"], ["--no-secondary"], ["--parallel", "replicas"], ["--graph"]):
        monkeypatch.setattr(sys, "argv", ["bench.py"] + argv)
        assert not bench.secondary_wanted(bench.parse()), argv
    names = [n for n, _, _ in bench.SECONDARY_LEGS]
    assert names == ["neumf", "sasrec", "deepfm_b1024", "neumf_100M", "deepfm_b131072"]
    # every leg carries a cpu_baseline: the port itself where it finishes in seconds (the 131,072-row DeepFM batch: 2 iterations),
    # the 10 M-item leg's figure as an upper bound for the 100 M-item tables (226 GB of dense gradients do not fit the command)
    assert [c is not None for _, _, c in bench.SECONDARY_LEGS] == [True, True, True, False, True]
    legs = {"neumf": {"value": 1.0, "cpu_baseline": {"value": 5.0, "sample": "x", "kind": "port"}}, "neumf_100M": {"value": 2.0}}
    bench._borrow_cpu_baseline(legs)
    assert legs["neumf_100M"]["cpu_baseline"]["value"] == 5.0 and legs["neumf_100M"]["cpu_baseline"]["bound"] == "upper"
    failed = {"neumf": {"value": 1.0, "cpu_baseline": {"value": 5.0}}, "neumf_100M": {"failed": "x"}}
    bench._borrow_cpu_baseline(failed)
    assert "cpu_baseline" not in failed["neumf_100M"]
    monkeypatch.setattr(sys, "argv", ["bench.py", "--workload", "deepfm"])
    assert bench.parse().dropout == 0.2      # docs/demo_scripts_results/CTR_MIND.sh:8
    for _, extra, _ in bench.SECONDARY_LEGS:   # every leg parses, and none of them recurses
        monkeypatch.setattr(sys, "argv", ["bench.py"] + extra + ["--no-secondary"])
        assert not bench.secondary_wanted(bench.parse())
    monkeypatch.setattr(sys, "argv", ["bench.py"] + bench.SHARDED_LEG)
    leg = bench.parse()
    assert (leg.workload, leg.items, leg.users, leg.emb_size, leg.num_neg, leg.opt) == ("neumf", 100_000_001, 10_000_001, 128, 4, "SGD")
    j = {"metric": "m", "value": 2.0, "unit": "u", "ms_per_step": 0.5, "config": {"workload": "w", "n_items": 7}, "cpu_baseline": {"x": 1},
         "roofline": {"frac": 0.3}, "phases_ms": {"a": 1.0}}
    sm = bench._summary(j)
    assert sm["workload"] == "w" and sm["roofline"] == {"frac": 0.3} and sm["cpu_baseline"] == {"x": 1} and "config" not in sm
    assert bench._last_json("noise\n{\"a\": 1}\ntrailing\n") == {"a": 1} and bench._last_json("nothing here") is None
    # the capturing launcher on the rendezvous-only ranks (gloo, no GPU)
    monkeypatch.setenv("RC_BENCH_LAUNCH_ONLY", "1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
        monkeypatch.delenv(k, raising=False)
    rc, text = bench.launch_ranks(2, ["--gpus", "2", "--dist-backend", "gloo"], capture=True)
    assert rc == 0 and bench._last_json(text)["n_gpus"] == 2
