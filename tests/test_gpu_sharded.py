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
        losses = []
        for s in range(steps):
            uid = rng.integers(0, n_users, size=(world, B)).astype(np.int64)
            iid = rng.integers(0, n_items, size=(world, B, C)).astype(np.int64)
            iid[:, :, 0] = iid[:, :, 0] % 7
            loss = m.step(torch.from_numpy(uid[rank]).to(dev), torch.from_numpy(iid[rank]).to(dev))
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
