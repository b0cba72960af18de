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

    def sum_rows_by_index(self, rows, index, n_out):
        out = np.zeros((n_out, rows.shape[1]), dtype=np.float32)
        np.add.at(out, index.numpy(), rows.numpy())
        return torch.from_numpy(out)

    def make_hyper(self, **kw):
        return kw

    def new_state(self, W, opt):
        return {k: torch.from_numpy(v) for k, v in O.new_state(W.numpy(), opt).items()}

    # the integer side of the routing, restated in numpy (HipOps: rc_route_by_owner, bucket plan + rc_plan_distinct)
    def route(self, ids, world, tuple_base=None, div=1):
        a = ids.numpy()
        owner = a % world
        order = np.argsort(owner, kind="stable")
        counts = torch.from_numpy(np.bincount(owner, minlength=world).astype(np.int64))
        local = a[order] // world
        if tuple_base is None:
            return torch.from_numpy(order), counts, torch.from_numpy(local)
        return torch.from_numpy(order), counts, torch.from_numpy(((tuple_base + order // div) << 32) | local)

    def unique(self, ids, n_rows):
        """distinct ids in ANY order + inverse index (the plan's order is not sorted; a scrambled order here proves that
        nothing downstream relies on sortedness)"""
        u, inv = np.unique(ids.numpy(), return_inverse=True)
        perm = np.random.default_rng(len(u)).permutation(len(u))       # new position k holds u[perm[k]]
        where = np.empty_like(perm)
        where[perm] = np.arange(len(u))
        return torch.from_numpy(u[perm]), torch.from_numpy(where[inv.reshape(-1)].astype(np.int64))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, opt, lr, l2, n_users, n_items, d, B, C, steps, out_q, mode="owner", dedup_users=True):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from rechorus_amd.sharded import ShardedBprmf
        rng = np.random.default_rng(5)  # same global problem on every rank
        U = rng.normal(0, 0.1, (n_users, d)).astype(np.float32)
        I = rng.normal(0, 0.1, (n_items, d)).astype(np.float32)
        m = ShardedBprmf(n_users, n_items, d, opt=opt, lr=lr, l2=l2, ops=OracleOps(), mode=mode, dedup_users=dedup_users)
        m.load_global(torch.from_numpy(U), torch.from_numpy(I))
        losses, batches = [], []
        for s in range(steps):
            uid = rng.integers(0, n_users, size=(world, B)).astype(np.int64)
            iid = rng.integers(0, n_items, size=(world, B, C)).astype(np.int64)
            iid[:, :, 0] = iid[:, :, 0] % 7  # hot positives: duplicates across ranks and owners
            uid[:, : B // 2] %= 3             # hot users: duplicates inside a rank's batch and across ranks
            batches.append((torch.from_numpy(uid[rank].copy()), torch.from_numpy(iid[rank].copy())))
        for s in range(steps):
            # odd steps route the following batch one step ahead (next_batch=), even ones on the spot: same result
            nxt = batches[s + 1] if (s % 2 == 1 and s + 1 < steps) else None
            loss = m.step(*batches[s], next_batch=nxt)
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
        uid[:, : B // 2] %= 3
        loss, _ = O.bprmf_train_step(U, I, sU, sI, uid.reshape(-1), iid.reshape(-1, C), opt=opt, lr=lr, l2=l2,
                                     step=s + 1, rowwise=True)
        losses.append(float(loss))
    return losses, U, I


@pytest.mark.parametrize("world,opt,lr,l2,mode", [(2, "SGD", 0.1, 1e-3, "owner"), (2, "Adam", 1e-2, 0.0, "owner"),
                                                  (3, "Adagrad", 0.05, 1e-4, "owner"), (2, "SGD", 0.1, 1e-3, "rows"),
                                                  (3, "Adam", 1e-2, 1e-4, "rows"), (3, "SGD", 0.1, 1e-3, "owner-nodedup")])
def test_sharded_step_equals_single_table_training(world, opt, lr, l2, mode):
    """both plans of the step -- user rows to the item owners ("owner": only the DISTINCT user rows of a rank's batch
    travel; "owner-nodedup": every tuple's row, the round-1 exchange), all rows to the tuples ("rows")"""
    dedup = mode != "owner-nodedup"
    mode = mode.split("-")[0]
    shape = dict(n_users=23, n_items=41, d=16, B=9, C=6, steps=3)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, opt, lr, l2, *shape.values(), q, mode, dedup)) for r in range(world)]
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


def test_plan_follows_the_traffic_model():
    """auto mode: few candidates per tuple -> the rows travel; many -> the user rows go to the item owners"""
    from rechorus_amd.sharded import ShardedBprmf
    m = ShardedBprmf(23, 41, 64, ops=OracleOps())
    for world, C, want in ((8, 2, "rows"), (8, 100, "owner"), (2, 3, "owner"), (4, 2, "rows"), (8, 5, "rows"), (4, 8, "owner")):
        m.world = world
        assert m._plan(C) == want, (world, C)
    m.mode = "owner"
    assert m._plan(2) == "owner"
    with pytest.raises(ValueError):
        ShardedBprmf(23, 41, 64, ops=OracleOps(), mode="sideways")


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


# ---- ShardedNeumf: the rows travel, the MFMA head runs on per-batch row blocks ------------------------------

from oracle import neumf_oracle as NO  # noqa: E402

_NAMES = {"mf_u": "mf_u_embeddings.weight", "mf_i": "mf_i_embeddings.weight", "mlp_u": "mlp_u_embeddings.weight",
          "mlp_i": "mlp_i_embeddings.weight", "W1": "mlp.0.weight", "b1": "mlp.0.bias"}


def _ref_params(P):
    out = {v: P[k].numpy() if hasattr(P[k], "numpy") else P[k] for k, v in _NAMES.items()}
    w = P["w_out"].numpy() if hasattr(P["w_out"], "numpy") else P["w_out"]
    out["prediction.weight"] = w.reshape(1, -1)
    return out


class NeumfOracleOps(OracleOps):
    """adds the NeuMF head (numpy oracle) and the dense optimizer step to the test double"""

    def neumf_fwd(self, P, uid, iid):
        pred, _ = NO.forward(_ref_params(P), uid.numpy(), iid.numpy())
        return torch.from_numpy(pred)

    def neumf_bwd(self, P, uid, iid, gpred):
        _, G = NO.backward(_ref_params(P), uid.numpy(), iid.numpy(), gpred.numpy())
        B, C = iid.shape
        d = P["mf_u"].shape[1]

        def per_occurrence(dense_u):  # the user "table" is the batch block [B, d]: put its gradient on candidate 0
            out = np.zeros((B, C, d), dtype=np.float32)
            out[:, 0, :] = dense_u
            return torch.from_numpy(out.reshape(B * C, d))
        rows = {"g_mf_u": per_occurrence(G[_NAMES["mf_u"]]), "g_mlp_u": per_occurrence(G[_NAMES["mlp_u"]]),
                "g_mf_i": torch.from_numpy(G[_NAMES["mf_i"]]), "g_mlp_i": torch.from_numpy(G[_NAMES["mlp_i"]])}
        dense = {"W1": torch.from_numpy(G["mlp.0.weight"]), "b1": torch.from_numpy(G["mlp.0.bias"]),
                 "w_out": torch.from_numpy(G["prediction.weight"][0].copy())}
        return rows, dense

    def dense_update(self, W, G, hyper, state):
        O.opt_step_dense(W.numpy(), G.numpy(), {k: v.numpy() for k, v in state.items()}, hyper["opt"], hyper["lr"],
                         hyper["l2"], step=hyper["step"])

    # ---- owner-computed item half (ShardedNeumf item_half="owner"): numpy restatements of rc_linear_fwd / rc_linear_bwd on the
    # served rows and of rc_neumf_zhead_fwd_bwd (models/general/NeuMF.py:61-75 with W1 [mlp_u ; mlp_i] = W1u mlp_u + W1i mlp_i)
    def item_half_fwd(self, mlp_rows, W1i):
        return torch.from_numpy((mlp_rows.numpy().astype(np.float64) @ W1i.numpy().T.astype(np.float64)).astype(np.float32))

    def item_half_bwd(self, mlp_rows, W1i, dz):
        dz64 = dz.numpy().astype(np.float64)
        return (torch.from_numpy((dz64 @ W1i.numpy()).astype(np.float32)),
                torch.from_numpy((dz64.T @ mlp_rows.numpy().astype(np.float64)).astype(np.float32)))

    def neumf_zhead(self, urows, irows, P, B, C, inv_b):
        d, l1 = urows.shape[1] // 2, P["W1"].shape[0]
        u, it = urows.numpy().astype(np.float64), irows.numpy().astype(np.float64)
        mf_u, mlp_u = u[:, :d], u[:, d:]
        mf_i, zi = it[:, :d].reshape(B, C, d), it[:, d:].reshape(B, C, l1)
        W1u, b1, w = P["W1"].numpy().astype(np.float64)[:, :d], P["b1"].numpy().astype(np.float64), P["w_out"].numpy().astype(np.float64)
        h = np.maximum((mlp_u @ W1u.T + b1)[:, None, :] + zi, 0.0)
        gmf = mf_u[:, None, :] * mf_i
        pred = ((gmf * w[:d]).sum(-1) + (h * w[d:]).sum(-1)).astype(np.float32)
        loss_vec, g = self.bpr_loss(torch.from_numpy(pred), inv_b)
        g = g.numpy().astype(np.float64)[..., None]
        dz = g * w[d:] * (h > 0)
        dzu = dz.sum(1)
        gu = np.concatenate([(g * mf_i).sum(1) * w[:d], dzu @ W1u], axis=1)
        gi = np.concatenate([(g * w[:d] * mf_u[:, None, :]).reshape(-1, d), dz.reshape(-1, l1)], axis=1)
        dense = {"W1u": dzu.T @ mlp_u, "b1": dzu.sum(0), "w_out": np.concatenate([(g * gmf).sum((0, 1)), (g * h).sum((0, 1))])}
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))
        return loss_vec, t(gu), t(gi), {k: t(v) for k, v in dense.items()}


def _neumf_problem(n_users, n_items, d, l1):
    rng = np.random.default_rng(9)
    P = {"mf_u": rng.normal(0, 0.3, (n_users, d)), "mlp_u": rng.normal(0, 0.3, (n_users, d)),
         "mf_i": rng.normal(0, 0.3, (n_items, d)), "mlp_i": rng.normal(0, 0.3, (n_items, d)),
         "W1": rng.normal(0, 0.3, (l1, 2 * d)), "b1": rng.normal(0, 0.3, l1), "w_out": rng.normal(0, 0.3, d + l1)}
    return rng, {k: v.astype(np.float32) for k, v in P.items()}


def _neumf_worker(rank, world, port, opt, lr, l2, n_users, n_items, d, l1, B, C, steps, out_q, micro_batches=1, item_half="owner"):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from rechorus_amd.sharded import ShardedNeumf
        rng, P = _neumf_problem(n_users, n_items, d, l1)
        m = ShardedNeumf(n_users, n_items, d, l1, opt=opt, lr=lr, l2=l2, ops=NeumfOracleOps(), micro_batches=micro_batches,
                         item_half=item_half)
        assert m.item_half == item_half
        m.load_global({k: torch.from_numpy(v) for k, v in P.items()})
        losses, batches = [], []
        for s in range(steps):
            uid = rng.integers(0, n_users, size=(world, B)).astype(np.int64)
            iid = rng.integers(0, n_items, size=(world, B, C)).astype(np.int64)
            iid[:, :, 0] %= 5
            batches.append((torch.from_numpy(uid[rank].copy()), torch.from_numpy(iid[rank].copy())))
        for s in range(steps):
            nxt = batches[s + 1] if (s % 2 == 0 and s + 1 < steps) else None   # look-ahead routing on every other step
            losses.append(float(m.step(*batches[s], next_batch=nxt)))
        # bytes over the links of the LAST step vs the model of tools/scale_model.py / DESIGN.md section 7: only the distinct
        # ids of a lookup that live on OTHER ranks travel, one row of both tables back, one gradient row out
        if micro_batches == 1 and world > 1:
            u_last, i_last = (t.numpy() for t in batches[-1])
            remote_u, remote_i = (int((np.unique(x) % world != rank).sum()) for x in (u_last, i_last.reshape(-1)))
            # a user id moves (mf_u | mlp_u) = 2 d floats each way; an item id (mf_i | mlp_i) = 2 d, or -- item half computed by the
            # owner -- (mf_i | W1i mlp_i) = d + l1 floats in and (d mf_i | dz) out
            item_floats = (d + l1) if m.item_half == "owner" else 2 * d
            rows = 4 * (2 * d * remote_u + item_floats * remote_i)
            want = {"ids_out": 8 * (remote_u + remote_i), "rows_in": rows, "grads_out": rows}
            got = {k: m.wire[k] for k in want}
            assert got == want, (got, want)
            assert m.wire["ids_sent"] == len(np.unique(u_last)) + len(np.unique(i_last)) < m.wire["lookups"]
        G = m.gather_global()
        if rank == 0:
            out_q.put((losses, {k: v.numpy() for k, v in G.items()}))
    finally:
        dist.destroy_process_group()


def _neumf_reference(world, opt, lr, l2, n_users, n_items, d, l1, B, C, steps):
    """single-table training on the concatenated global batch: row-wise update of the touched table rows,
    dense step of the MLP ('bias' parameter without weight decay)"""
    rng, P = _neumf_problem(n_users, n_items, d, l1)
    st = {k: O.new_state(v, opt) for k, v in P.items()}
    losses = []
    for s in range(steps):
        uid = rng.integers(0, n_users, size=(world, B)).astype(np.int64).reshape(-1)
        iid = rng.integers(0, n_items, size=(world, B, C)).astype(np.int64)
        iid[:, :, 0] %= 5
        iid = iid.reshape(-1, C)
        R = _ref_params(P)
        pred, _ = NO.forward(R, uid, iid)
        losses.append(float(O.bpr_loss(pred)))
        _, G = NO.backward(R, uid, iid, O.bpr_loss_grad(pred))
        for k in ("mf_u", "mlp_u"):
            O.opt_step_dense(P[k], G[_NAMES[k]], st[k], opt, lr, l2, step=s + 1, rows=np.unique(uid))
        for k in ("mf_i", "mlp_i"):
            O.opt_step_dense(P[k], G[_NAMES[k]], st[k], opt, lr, l2, step=s + 1, rows=np.unique(iid))
        O.opt_step_dense(P["W1"], G["mlp.0.weight"], st["W1"], opt, lr, l2, step=s + 1)
        O.opt_step_dense(P["b1"], G["mlp.0.bias"], st["b1"], opt, lr, 0.0, step=s + 1)
        O.opt_step_dense(P["w_out"], G["prediction.weight"][0], st["w_out"], opt, lr, l2, step=s + 1)
    return losses, P


@pytest.mark.parametrize("world,opt,lr,l2,micro_batches,item_half", [
    (2, "SGD", 0.1, 1e-3, 1, "owner"), (3, "Adam", 1e-2, 1e-4, 1, "owner"), (2, "Adam", 1e-2, 1e-4, 2, "owner"), (3, "SGD", 0.1, 1e-3, 3, "owner"),
    (2, "SGD", 0.1, 1e-3, 1, "rows"), (3, "Adam", 1e-2, 1e-4, 1, "rows"), (2, "Adam", 1e-2, 1e-4, 2, "rows"), (3, "SGD", 0.1, 1e-3, 3, "rows")])
def test_sharded_neumf_equals_single_table_training(world, opt, lr, l2, micro_batches, item_half):
    """micro_batches > 1: the pipelined step (chunks scored against the pre-step parameters, ONE update over the
    gradients of all chunks) must equal the same single-table training.  item_half "owner":
    the owners return (mf_i | W1i mlp_i) and receive (d mf_i | dz) -- d + l1 instead of 2 d floats per distinct item id each way,
    asserted on the byte counters -- and form d mlp_i and their share of dW1i themselves; "rows": both item rows travel"""
    shape = dict(n_users=19, n_items=37, d=8, l1=6, B=7, C=4, steps=3)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_neumf_worker, args=(r, world, port, opt, lr, l2, *shape.values(), q, micro_batches, item_half))
             for r in range(world)]
    for p in procs:
        p.start()
    losses, G = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want_losses, P = _neumf_reference(world, opt, lr, l2, **shape)
    np.testing.assert_allclose(losses, want_losses, rtol=5e-6)
    # chunked summation changes the last bits of a gradient; Adam turns that into a visible step only where the
    # gradient itself is rounding noise (|g| ~ eps: one element of b1 here) -- ill-conditioned in any implementation
    atol = 1e-6 if (opt == "SGD" or micro_batches == 1) else 0.05 * lr
    for k, v in P.items():
        # (that element of b1: Adam normalises its noise gradient to a step of up to lr per step in either direction)
        np.testing.assert_allclose(G[k], v, rtol=2e-5, atol=(0.3 * lr if (k == "b1" and atol > 1e-6) else atol), err_msg=k)


def test_sharded_neumf_single_rank():
    from rechorus_amd.sharded import ShardedNeumf
    shape = dict(n_users=19, n_items=37, d=8, l1=6, B=7, C=4, steps=2)
    rng, P = _neumf_problem(19, 37, 8, 6)
    m = ShardedNeumf(19, 37, 8, 6, opt="SGD", lr=0.1, l2=1e-3, ops=NeumfOracleOps())
    m.load_global({k: torch.from_numpy(v) for k, v in P.items()})
    losses = []
    for s in range(2):
        uid = rng.integers(0, 19, size=(1, 7)).astype(np.int64)
        iid = rng.integers(0, 37, size=(1, 7, 4)).astype(np.int64)
        iid[:, :, 0] %= 5
        losses.append(float(m.step(torch.from_numpy(uid[0]), torch.from_numpy(iid[0]))))
    want_losses, Pw = _neumf_reference(1, "SGD", 0.1, 1e-3, **shape)
    np.testing.assert_allclose(losses, want_losses, rtol=5e-6)
    G = m.gather_global()
    for k, v in Pw.items():
        np.testing.assert_allclose(G[k].numpy(), v, rtol=2e-5, atol=1e-6, err_msg=k)


# ---- data-parallel leg of the replicated-parameter models (config 5) ------------------------------------------------

class _TinyCtr(torch.nn.Module):
    """embedding + MLP + BCE, plain torch ops (the wrapper under test only touches gradients and the optimizer)"""

    def __init__(self):
        super().__init__()
        torch.manual_seed(3)
        self.emb = torch.nn.Embedding(23, 8)
        self.mlp = torch.nn.Sequential(torch.nn.Linear(16, 12), torch.nn.ReLU(), torch.nn.Linear(12, 1))
        self.optimizer = None

    def forward(self, b):
        return self.mlp(self.emb(b["ids"]).flatten(1)).view(-1).sigmoid()

    def loss(self, p):
        return torch.nn.BCELoss()(p, self._y)


def _dp_batches(world, steps, B):
    rng = np.random.default_rng(11)
    return [[(rng.integers(0, 23, size=(B, 2)), rng.integers(0, 2, size=B).astype(np.float32)) for _ in range(world)]
            for _ in range(steps)]


def _dp_worker(rank, world, port, opt, out_q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from rechorus_amd.sharded import DataParallelDense
        m = _TinyCtr()
        m.optimizer = getattr(torch.optim, opt)(m.parameters(), lr=0.05)

        def loss_of(model, b):
            model._y = b["y"]
            return model.loss(model(b))
        dp = DataParallelDense(m, loss_of=loss_of)
        losses = []
        for per_rank in _dp_batches(world, 3, 6):
            ids, y = per_rank[rank]
            losses.append(float(dp.step({"ids": torch.from_numpy(ids), "y": torch.from_numpy(y)})))
        if rank == 0:
            out_q.put((losses, {k: v.detach().numpy().copy() for k, v in m.state_dict().items()}, dp.bytes_per_step))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,opt", [(2, "SGD"), (3, "Adam")])
def test_data_parallel_dense_equals_training_on_the_global_batch(world, opt):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_dp_worker, args=(r, world, port, opt, q)) for r in range(world)]
    for p in procs:
        p.start()
    losses, sd, nbytes = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ref = _TinyCtr()
    optim = getattr(torch.optim, opt)(ref.parameters(), lr=0.05)
    want = []
    for per_rank in _dp_batches(world, 3, 6):
        ids = torch.from_numpy(np.concatenate([b[0] for b in per_rank]))
        ref._y = torch.from_numpy(np.concatenate([b[1] for b in per_rank]))
        optim.zero_grad()
        loss = ref.loss(ref({"ids": ids}))
        loss.backward()
        optim.step()
        want.append(float(loss))
    np.testing.assert_allclose(losses, want, rtol=1e-5)
    for k, v in ref.state_dict().items():
        np.testing.assert_allclose(sd[k], v.detach().numpy(), rtol=2e-5, atol=2e-6, err_msg=k)
    n_param = sum(p.numel() for p in ref.parameters())
    assert nbytes == 2 * (world - 1) * n_param * 4 // world
