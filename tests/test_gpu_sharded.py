"""GPU: the sharded "owner computes" step with the REAL local kernels (HipOps through the C ABI),
two ranks sharing cuda:0 and talking over gloo (the test box has one GPU; RCCL refuses two ranks on
one device).  Must equal single-table training of the concatenated global batch (numpy oracle)."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from test_sharded_gloo import _free_port, _reference

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, opt, lr, l2, n_users, n_items, d, B, C, steps, out_q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from rechorus_amd.sharded import ShardedBprmf
        dev = torch.device("cuda:0")
        rng = np.random.default_rng(5)
        U = rng.normal(0, 0.1, (n_users, d)).astype(np.float32)
        I = rng.normal(0, 0.1, (n_items, d)).astype(np.float32)
        m = ShardedBprmf(n_users, n_items, d, opt=opt, lr=lr, l2=l2, device=dev)
        m.load_global(torch.from_numpy(U).to(dev), torch.from_numpy(I).to(dev))
        losses, batches = [], []
        for s in range(steps):
            uid = rng.integers(0, n_users, size=(world, B)).astype(np.int64)
            iid = rng.integers(0, n_items, size=(world, B, C)).astype(np.int64)
            iid[:, :, 0] = iid[:, :, 0] % 7
            uid[:, : B // 2] %= 3   # hot users (the same stream of batches as tests/test_sharded_gloo.py::_reference)
            batches.append((torch.from_numpy(uid[rank]).to(dev), torch.from_numpy(iid[rank]).to(dev)))
        for s in range(steps):
            # the first step also routes the second batch ahead, on the side stream (sharded._LookAhead)
            nxt = batches[s + 1] if (s == 0 and s + 1 < steps) else None
            loss = m.step(*batches[s], next_batch=nxt)
            losses.append(float(loss))
        Ug, Ig = m.gather_global()
        if rank == 0:
            out_q.put((losses, Ug.cpu().numpy(), Ig.cpu().numpy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("opt,lr,l2", [("SGD", 0.1, 1e-3), ("Adam", 1e-2, 0.0)])
def test_two_ranks_hip_ops_equal_single_table_training(opt, lr, l2, cuda):
    world = 2
    shape = dict(n_users=203, n_items=1001, d=64, B=96, C=20, steps=2)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, opt, lr, l2, *shape.values(), q)) for r in range(world)]
    for p in procs:
        p.start()
    losses, Ug, Ig = q.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want_losses, U, I = _reference(world, opt, lr, l2, **shape)
    np.testing.assert_allclose(losses, want_losses, rtol=1e-5)
    atol = 2e-5 if opt == "Adam" else 5e-7  # Adam: |g| ~ eps elements (see conftest.assert_update_close)
    np.testing.assert_allclose(Ug, U, rtol=1e-4, atol=atol)
    np.testing.assert_allclose(Ig, I, rtol=1e-4, atol=atol)


# ---- local kernels of the sharded step (csrc/owner_step.hip) -------------------------------------------------

@pytest.mark.parametrize("world", [1, 2, 3, 8, 64])
def test_route_by_owner_is_a_stable_counting_sort(world, cuda):
    from rechorus_amd import engine
    rng = np.random.default_rng(world)
    for n in (1, 63, 4095, 4096, 4097, 300_001):
        ids = rng.integers(0, 10_000_000, size=n).astype(np.int64)
        ids[: n // 3] = ids[: n // 3] % 5  # hot ids: long same-owner runs
        order, counts, local = engine.route_by_owner(torch.from_numpy(ids).to(cuda), world)
        want_order = np.argsort(ids % world, kind="stable")
        assert np.array_equal(order.cpu().numpy(), want_order)
        assert np.array_equal(counts.cpu().numpy(), np.bincount(ids % world, minlength=world))
        assert np.array_equal(local.cpu().numpy(), ids[want_order] // world)
        C, base = 7, 1000
        order2, counts2, packed = engine.route_by_owner(torch.from_numpy(ids).to(cuda), world, tuple_base=base, div=C)
        assert torch.equal(order2, order) and torch.equal(counts2, counts)
        assert np.array_equal(packed.cpu().numpy(), ((base + want_order // C) << 32) | (ids[want_order] // world))
        t_idx, rows, t32 = engine.owner_unpack(packed)
        assert np.array_equal(t_idx.cpu().numpy(), base + want_order // C) and np.array_equal(rows.cpu().numpy(), ids[want_order] // world)
        assert np.array_equal(t32.cpu().numpy(), (base + want_order // C).astype(np.int32))
    order, counts, local = engine.route_by_owner(torch.zeros(0, dtype=torch.int64, device=cuda), world)
    assert order.numel() == 0 and int(counts.sum()) == 0


@pytest.mark.parametrize("opt,d", [("SGD", 64), ("Adam", 32), ("Adagrad", 128)])
def test_owner_backward_equals_the_unfused_owner_ops(opt, d, cuda):
    """rc_owner_backward + skip-singleton segmented update == partial_user_grads + full update_rows"""
    from rechorus_amd import engine
    from rechorus_amd.sharded import HipOps
    from conftest import assert_close, assert_update_close
    rng = np.random.default_rng(d)
    n_rows, n_tuples, n = 5000, 300, 4000
    I0 = rng.normal(0, 0.1, (n_rows, d)).astype(np.float32)
    Uall = torch.from_numpy(rng.normal(0, 0.1, (n_tuples, d)).astype(np.float32)).to(cuda)
    t = np.sort(rng.integers(0, n_tuples, size=n)).astype(np.int64)   # grouped by tuple, some tuples absent
    t[t == 17] = 18
    rows = rng.integers(0, n_rows, size=n).astype(np.int64)
    rows[::9] = rows[::9] % 11                                          # rows shared by many occurrences
    g = rng.normal(0, 1, size=n).astype(np.float32)
    dev = lambda x: torch.from_numpy(x).to(cuda)
    ops = HipOps()
    res = []
    for fused in (False, True):
        I = dev(I0.copy())
        st = ops.new_state(I, opt)
        hyper = ops.make_hyper(opt=opt, lr=0.05, l2=1e-3, step=1)
        if fused:
            prep = ops.prepare_owner(dev(rows), n_rows)
            pug = ops.owner_backward(I, st, dev(rows), dev(g), dev(t), dev(t.astype(np.int32)), Uall, n_tuples, hyper, prep)
        else:
            pug = ops.partial_user_grads(I, dev(rows), dev(g), dev(t), n_tuples)
            ops.update_rows(I, st, dev(rows), Uall, hyper, coef=dev(g), src_index=dev(t))
        res.append((pug.cpu().numpy(), I.cpu().numpy(), {k: v.cpu().numpy() for k, v in st.items()}))
    (pug_a, I_a, st_a), (pug_b, I_b, st_b) = res
    assert np.abs(pug_b[17]).max() == 0 and np.abs(pug_b[18]).max() > 0
    assert_close(pug_b, pug_a, what="pug", atol_scale=1e-6)
    ex = 1e-3 * 0.05 if opt != "SGD" else 0.0
    assert_update_close(I_b, I0, I_a, what="item rows", extra_atol=ex)
    untouched = np.setdiff1d(np.arange(n_rows), rows)
    assert np.array_equal(I_b[untouched], I0[untouched])
    for k in st_a:
        assert_close(st_b[k], st_a[k], what="state " + k, atol_scale=1e-5)


def _loopback_worker(port, out_q, mode="owner", C=30, sorted_route=False):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        from oracle import bprmf_oracle as O
        from rechorus_amd.sharded import ShardedBprmf
        if sorted_route:   # as if the bucket plan had no geometry for these lists: HipOps falls back to sort + segmented sum
            from rechorus_amd import engine
            engine.plan_supported = lambda *a: False
        dev = torch.device("cuda:0")
        rng = np.random.default_rng(8)
        n_users, n_items, d, B = 301, 2003, 64, 256
        U = rng.normal(0, 0.1, (n_users, d)).astype(np.float32)
        I = rng.normal(0, 0.1, (n_items, d)).astype(np.float32)
        out = {}
        for opt, lr in (("SGD", 0.1), ("Adam", 1e-2)):
            m = ShardedBprmf(n_users, n_items, d, opt=opt, lr=lr, l2=1e-4, device=dev, force_exchange=True, timing=True,
                             mode=mode)
            m.load_global(torch.from_numpy(U).to(dev), torch.from_numpy(I).to(dev))
            Un, In = U.copy(), I.copy()
            sU, sI = O.new_state(Un, opt), O.new_state(In, opt)
            losses = []
            for step in (1, 2):
                uid = rng.integers(0, n_users, size=B).astype(np.int64)
                iid = rng.integers(0, n_items, size=(B, C)).astype(np.int64)
                iid[:, 0] %= 9
                loss = float(m.step(torch.from_numpy(uid).to(dev), torch.from_numpy(iid).to(dev)))
                want, _ = O.bprmf_train_step(Un, In, sU, sI, uid, iid, opt=opt, lr=lr, l2=1e-4, step=step, rowwise=True)
                losses.append((loss, float(want)))
            Ug, Ig = m.gather_global()
            out[opt] = (losses, Ug.cpu().numpy(), Un, Ig.cpu().numpy(), In, sorted(m.timing_ms()))
        out_q.put((U, I, out))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode,C,n_phases,sorted_route", [("owner", 30, 8, False), ("rows", 30, 5, False), ("rows", 2, 5, False),
                                                          ("owner", 30, 8, True), ("rows", 30, 5, True)])
def test_world_of_one_through_the_exchange_path(mode, C, n_phases, sorted_route, cuda):
    """force_exchange: the full routed step on one rank vs the oracle, both plans -- "owner" (HIP counting sort,
    unpack, owner backward) and "rows" (routes, row fetch, fused kernel on compact row blocks, gradient push); sorted_route: the
    same with the bucket plan reported unavailable (HipOps' sort + segmented-sum fallback, sharded._SortPlan)"""
    from conftest import assert_update_close
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_loopback_worker, args=(_free_port(), q, mode, C, sorted_route))
    p.start()
    U0, I0, out = q.get(timeout=300)
    p.join(timeout=60)
    assert p.exitcode == 0
    for opt, (losses, Ug, Un, Ig, In, phases) in out.items():
        for got, want in losses:
            assert abs(got - want) <= 1e-5 * abs(want)
        ex = 1e-3 * 1e-2 if opt == "Adam" else 0.0
        assert_update_close(Ug, U0, Un, what=opt + " dU", extra_atol=ex)
        assert_update_close(Ig, I0, In, what=opt + " dI", extra_atol=ex)
        assert len(phases) == n_phases


@pytest.mark.parametrize("launcher,workload,extra", [
    ("torchrun", "bprmf", ["--batch", "2048", "--num-neg", "19", "--items", "200001", "--users", "20001"]),
    ("self", "bprmf", ["--batch", "2048", "--num-neg", "19", "--items", "200001", "--users", "20001"]),
    ("self", "neumf", ["--batch", "2048", "--items", "200001", "--users", "20001"]),
    ("self", "deepfm", ["--batch", "512"]),
])
def test_bench_multi_rank_code_path_on_one_gpu(launcher, workload, extra, cuda):
    """the driver's N > 1 command lines -- `python -m torch.distributed.run ... bench.py --gpus 2` and plain
    `python bench.py --gpus 2` (bench.py starts its own ranks) -- with both ranks on cuda:0 and gloo collectives
    (RC_BENCH_ONE_DEVICE): not a measurement, but every line of bench.py's multi-GPU branches runs, for the sharded
    BPRMF / NeuMF steps and the data-parallel DeepFM leg"""
    import json
    import subprocess
    import sys
    from conftest import ROOT
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env["RC_BENCH_ONE_DEVICE"] = "1"
    env["RC_BENCH_SECONDARY_ROWS"] = "300001,30001"     # the config-4 leg of the contract workload's line, on small tables here
    tail = [os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--dist-backend", "gloo",
            "--workload", workload, "--no-cpu-baseline"] + extra
    if launcher == "torchrun":
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port())] + tail
    else:
        cmd = [sys.executable] + tail
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["value"] > 0 and np.isfinite(out["final_loss"])
    ph = out["sharded_phases_ms"]
    if workload == "bprmf":
        assert len(ph) == 8
        # the contract workload carries the sharded NeuMF leg of BASELINE configs[3] (its own rank group under the self-launcher,
        # the same ranks under torch.distributed.run), summarised -- still ONE line
        sec = out["secondary"]["neumf_100M"]
        assert sec.get("value", 0) > 0 and sec["n_gpus"] == 2 and "NeuMF" in sec["workload"], sec
        assert {"fetch_rows", "head_fwd_loss", "push_grads", "table_update"} <= set(sec["sharded_phases_ms"]) and sec["sharded_wire_bytes_rank0"]
    elif workload == "neumf":
        assert {"fetch_rows", "head_fwd_loss", "head_bwd", "push_grads", "table_update", "dense_update"} <= set(ph)
        assert out["sharded_wire_bytes_rank0"]
    else:
        assert set(ph) == {"forward_backward", "gradient_allreduce", "optimizer"}
        assert out["sharded_wire_bytes_rank0"]["dense_gradient_allreduce"] > 0
        assert "replicated" in out["config"]["parallelism"]


def _neumf_worker(rank, world, port, opt, lr, l2, n_users, n_items, d, l1, B, C, steps, out_q, micro_batches=1, item_half="owner"):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from rechorus_amd.sharded import ShardedNeumf
        from test_sharded_gloo import _neumf_problem
        dev = torch.device("cuda:0")
        rng, P = _neumf_problem(n_users, n_items, d, l1)
        m = ShardedNeumf(n_users, n_items, d, l1, opt=opt, lr=lr, l2=l2, device=dev, micro_batches=micro_batches, item_half=item_half)
        assert m.item_half == item_half
        m.load_global({k: torch.from_numpy(v).to(dev) for k, v in P.items()})
        losses = []
        for s in range(steps):
            uid = rng.integers(0, n_users, size=(world, B)).astype(np.int64)
            iid = rng.integers(0, n_items, size=(world, B, C)).astype(np.int64)
            iid[:, :, 0] %= 5
            losses.append(float(m.step(torch.from_numpy(uid[rank]).to(dev), torch.from_numpy(iid[rank]).to(dev))))
        G = m.gather_global()
        if rank == 0:
            out_q.put((losses, {k: v.cpu().numpy() for k, v in G.items()}))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,opt,lr,l2,micro_batches,item_half", [
    (2, "SGD", 0.1, 1e-3, 1, "owner"), (2, "Adam", 1e-2, 0.0, 1, "owner"), (1, "SGD", 0.1, 1e-3, 1, "rows"), (2, "SGD", 0.1, 1e-3, 3, "owner"),
    (2, "SGD", 0.1, 1e-3, 1, "rows"), (2, "Adam", 1e-2, 0.0, 1, "rows"), (2, "SGD", 0.1, 1e-3, 3, "rows")])
def test_sharded_neumf_hip_ops_equal_single_table_training(world, opt, lr, l2, micro_batches, item_half, cuda):
    """ShardedNeumf with the real kernels (routing, gathers, the head on per-batch row blocks, segmented updates), ranks sharing
    cuda:0 over gloo, vs single-table training of the global batch (numpy oracle).  item_half "owner" (the
    owners compute W1i mlp_i -- rc_linear_fwd / rc_linear_bwd on the served rows, rc_neumf_zhead_fwd_bwd at home), "rows": both item
    rows travel and the MFMA head runs at home"""
    from test_sharded_gloo import _neumf_reference
    shape = dict(n_users=203, n_items=1001, d=64, l1=32, B=96, C=5, steps=2)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_neumf_worker, args=(r, world, port, opt, lr, l2, *shape.values(), q, micro_batches, item_half))
             for r in range(world)]
    for p in procs:
        p.start()
    losses, G = q.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want_losses, P = _neumf_reference(world, opt, lr, l2, **shape)
    np.testing.assert_allclose(losses, want_losses, rtol=2e-5)
    atol = 3e-5 if opt == "Adam" else 2e-6  # Adam: |g| ~ eps elements (see conftest.assert_update_close)
    for k, v in P.items():
        np.testing.assert_allclose(G[k], v, rtol=1e-4, atol=atol, err_msg=k)


@pytest.mark.parametrize("B,C,d,l1", [(96, 5, 64, 32), (7, 2, 8, 6), (300, 100, 128, 64), (33, 9, 48, 20), (2000, 5, 128, 64)])
def test_neumf_zhead_kernel_vs_numpy(B, C, d, l1, cuda):
    """engine.neumf_zhead (rc_linear_fwd -> rc_neumf_zhead_fwd_bwd -> rc_linear_bwd) against the float64 restatement the gloo tests use
    as the home rank's head (tests/test_sharded_gloo.py::NeumfOracleOps.neumf_zhead): loss rows, both gradient blocks, the dense grads"""
    from rechorus_amd import engine
    from test_sharded_gloo import NeumfOracleOps
    rng = np.random.default_rng(B + C)
    mk = lambda *s: torch.from_numpy(rng.normal(0, 0.4, s).astype(np.float32))
    urows, irows = mk(B, 2 * d), mk(B * C, d + l1)
    P = {"W1": mk(l1, 2 * d), "b1": mk(l1), "w_out": mk(d + l1)}
    want = NeumfOracleOps().neumf_zhead(urows, irows, P, B, C, 1.0 / B)
    got = engine.neumf_zhead(urows.to(cuda), irows.to(cuda), P["W1"].to(cuda), P["b1"].to(cuda), P["w_out"].to(cuda), B, C, 1.0 / B, want_pred=True)
    from conftest import assert_close
    assert_close(got[0].cpu().numpy(), want[0].numpy(), what="loss rows")
    assert_close(got[1].cpu().numpy(), want[1].numpy(), what="gu", atol_scale=2e-5)
    assert_close(got[2].cpu().numpy(), want[2].numpy(), what="gi", atol_scale=2e-5)
    for k in ("W1u", "b1", "w_out"):
        assert_close(got[3][k].cpu().numpy(), want[3][k].numpy(), what="d" + k, rtol=2e-5, atol_scale=2e-5, abs_floor=3e-7 * (B * C) ** 0.5)
    assert not engine.neumf_zhead_supported(1, 64, 32) and not engine.neumf_zhead_supported(300, 64, 32) and engine.neumf_zhead_supported(5, 128, 64)
