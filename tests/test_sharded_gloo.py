"""CPU, world_size 2 (and 3), gloo: the row-sharded "owner computes" BPRMF step
(rechorus_amd/sharded.py) must equal single-table training on the concatenated global batch.
The routing / collectives are the code under test; the local arithmetic is injected from the
numpy oracle (the HIP kernels are covered by the -m gpu tests and by W=1 equivalence there)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import bprmf_oracle as O


class OracleOps:
    """local ops of ShardedBprmf restated with the numpy oracle (CPU tensors in / out)"""

    def gather_rows(self, W, rows):
        return W[rows]

    def dot_rows(self, Uall, t_idx, I_loc, rows):
        u, i = Uall.numpy()[t_idx.numpy()], I_loc.numpy()[rows.numpy()]
        return torch.from_numpy((u * i).sum(axis=-1, dtype=np.float32))

    def bpr_loss(self, pred, inv_b):
        p = pred.numpy()
        rows, _, _, _ = O.bpr_loss_rows(p)
        return torch.from_numpy(rows), torch.from_numpy(O.bpr_loss_grad(p, inv_b=inv_b))

    def partial_user_grads(self, I_loc, rows, g, t_idx, n_tuples):
        out = np.zeros((n_tuples, I_loc.shape[1]), dtype=np.float32)
        np.add.at(out, t_idx.numpy(), g.numpy()[:, None] * I_loc.numpy()[rows.numpy()])
        return torch.from_numpy(out)

    def update_rows(self, W, state, rows, src, hyper, coef=None, src_index=None):
        if rows.numel() == 0:
            return
        r = rows.numpy()
        s = src.numpy()[src_index.numpy()] if src_index is not None else src.numpy()
        if coef is not None:
            s = coef.numpy()[:, None] * s
        G = np.zeros(W.shape, dtype=np.float32)
        np.add.at(G, r, s.astype(np.float32))
        Wn = W.numpy()
        st = {k: v.numpy() for k, v in state.items()}
        O.opt_step_dense(Wn, G, st, hyper["opt"], hyper["lr"], hyper["l2"], step=hyper["step"], rows=np.unique(r))

    def make_hyper(self, **kw):
        return kw

    def new_state(self, W, opt):
        return {k: torch.from_numpy(v) for k, v in O.new_state(W.numpy(), opt).items()}


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, opt, lr, l2, n_users, n_items, d, B, C, steps, out_q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from rechorus_amd.sharded import ShardedBprmf
        rng = np.random.default_rng(5)  # same global problem on every rank
        U = rng.normal(0, 0.1, (n_users, d)).astype(np.float32)
        I = rng.normal(0, 0.1, (n_items, d)).astype(np.float32)
        m = ShardedBprmf(n_users, n_items, d, opt=opt, lr=lr, l2=l2, ops=OracleOps())
        m.load_global(torch.from_numpy(U), torch.from_numpy(I))
        losses = []
        for s in range(steps):
            uid = rng.integers(0, n_users, size=(world, B)).astype(np.int64)
            iid = rng.integers(0, n_items, size=(world, B, C)).astype(np.int64)
            iid[:, :, 0] = iid[:, :, 0] % 7  # hot positives: duplicates across ranks and owners
            loss = m.step(torch.from_numpy(uid[rank]), torch.from_numpy(iid[rank]))
            losses.append(float(loss))
        Ug, Ig = m.gather_global()
        if rank == 0:
            out_q.put((losses, Ug.numpy(), Ig.numpy()))
    finally:
        dist.destroy_process_group()


def _reference(world, opt, lr, l2, n_users, n_items, d, B, C, steps):
    rng = np.random.default_rng(5)
    U = rng.normal(0, 0.1, (n_users, d)).astype(np.float32)
    I = rng.normal(0, 0.1, (n_items, d)).astype(np.float32)
    sU, sI = O.new_state(U, opt), O.new_state(I, opt)
    losses = []
    for s in range(steps):
        uid = rng.integers(0, n_users, size=(world, B)).astype(np.int64)
        iid = rng.integers(0, n_items, size=(world, B, C)).astype(np.int64)
        iid[:, :, 0] = iid[:, :, 0] % 7
        loss, _ = O.bprmf_train_step(U, I, sU, sI, uid.reshape(-1), iid.reshape(-1, C), opt=opt, lr=lr, l2=l2,
                                     step=s + 1, rowwise=True)
        losses.append(float(loss))
    return losses, U, I


@pytest.mark.parametrize("world,opt,lr,l2", [(2, "SGD", 0.1, 1e-3), (2, "Adam", 1e-2, 0.0), (3, "Adagrad", 0.05, 1e-4)])
def test_sharded_step_equals_single_table_training(world, opt, lr, l2):
    shape = dict(n_users=23, n_items=41, d=16, B=9, C=6, steps=3)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, opt, lr, l2, *shape.values(), q)) for r in range(world)]
    for p in procs:
        p.start()
    losses, Ug, Ig = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want_losses, U, I = _reference(world, opt, lr, l2, **shape)
    np.testing.assert_allclose(losses, want_losses, rtol=2e-6)
    np.testing.assert_allclose(Ug, U, rtol=1e-5, atol=2e-7)
    np.testing.assert_allclose(Ig, I, rtol=1e-5, atol=2e-7)


def test_single_rank_path_matches_oracle():
    """W = 1 runs the same ops without any collective"""
    from rechorus_amd.sharded import ShardedBprmf
    rng = np.random.default_rng(5)
    U = rng.normal(0, 0.1, (23, 16)).astype(np.float32)
    I = rng.normal(0, 0.1, (41, 16)).astype(np.float32)
    m = ShardedBprmf(23, 41, 16, opt="SGD", lr=0.1, l2=1e-3, ops=OracleOps())
    m.load_global(torch.from_numpy(U), torch.from_numpy(I))
    uid = rng.integers(0, 23, size=9).astype(np.int64)
    iid = rng.integers(0, 41, size=(9, 6)).astype(np.int64)
    loss = float(m.step(torch.from_numpy(uid), torch.from_numpy(iid)))
    want, _ = O.bprmf_train_step(U, I, {}, {}, uid, iid, opt="SGD", lr=0.1, l2=1e-3, rowwise=True)
    assert abs(loss - float(want)) < 2e-6
    Ug, Ig = m.gather_global()
    np.testing.assert_allclose(Ug.numpy(), U, rtol=1e-5, atol=2e-7)
    np.testing.assert_allclose(Ig.numpy(), I, rtol=1e-5, atol=2e-7)
