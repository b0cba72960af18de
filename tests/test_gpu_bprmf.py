"""GPU parity: the HIP engine (through the C ABI) vs the reference's own outputs (golden
fixtures) and vs the numpy oracle on seeded inputs.  Tolerance: 1e-5 relative fp32
(BASELINE.json north_star), written out in conftest.assert_close."""
import numpy as np
import pytest
import torch

from conftest import assert_close, assert_update_close, golden_cases, load_golden
from oracle import bprmf_oracle as O

pytestmark = pytest.mark.gpu


def dev(a, cuda):
    return torch.from_numpy(np.ascontiguousarray(a)).to(cuda)


def host(t):
    return t.detach().cpu().numpy()


@pytest.fixture(scope="module")
def eng(cuda):
    from rechorus_amd import engine
    return engine


# ---- golden: forward / loss / backward ----------------------------------------------------

@pytest.mark.parametrize("case", golden_cases())
def test_gather_rows_and_dot(case, cuda, eng):
    g = load_golden(case)
    U, I, uid, iid = (dev(g[k], cuda) for k in ("U0", "I0", "uid", "iid"))
    rows = eng.gather_rows(I, iid)
    assert np.array_equal(host(rows), g["I0"][g["iid"]])  # pure copy: bit exact
    pred = eng.gather_dot(U, I, uid, iid)
    assert_close(host(pred), g["pred"], what="pred")


@pytest.mark.parametrize("case", golden_cases())
def test_bpr_loss(case, cuda, eng):
    g = load_golden(case)
    loss, loss_vec, gpred = eng.bpr_loss(dev(g["pred"], cuda))
    assert_close(host(loss)[0], g["loss"], what="loss")
    assert_close(host(gpred), g["gpred"], what="gpred")
    rows, _, _, _ = O.bpr_loss_rows(g["pred"])
    assert_close(host(loss_vec), rows, what="loss_vec")


@pytest.mark.parametrize("case", golden_cases())
def test_fused_fwd_bwd(case, cuda, eng):
    g = load_golden(case)
    U, I, uid, iid = (dev(g[k], cuda) for k in ("U0", "I0", "uid", "iid"))
    B, C = g["iid"].shape
    pred, loss_vec, gpred, ugrad = eng.bprmf_fwd_bwd(U, I, uid, iid)
    assert_close(host(pred), g["pred"], what="pred")
    assert_close(host(gpred), g["gpred"], what="gpred")
    assert_close(host(eng.reduce_sum(loss_vec, 1.0 / B))[0], g["loss"], what="loss")
    # dense table gradients, through the atomic-free sort + segmented sum
    GU = eng.embedding_dense_backward(ugrad, uid, g["U0"].shape[0])
    assert_close(host(GU), g["GU"], what="GU")
    keys, perm = eng.sort_ids(iid, g["I0"].shape[0])
    GI = torch.zeros_like(I)
    eng.segmented_update(keys, perm, U, coef=gpred.reshape(-1), src_index=uid, div=C, dense_grad=GI)
    assert_close(host(GI), g["GI"], what="GI")
    # without the optional pred output the other outputs are unchanged
    _, lv2, gp2, ug2 = eng.bprmf_fwd_bwd(U, I, uid, iid, want_pred=False)
    assert torch.equal(lv2, loss_vec) and torch.equal(gp2, gpred) and torch.equal(ug2, ugrad)


def test_sort_ids_is_a_stable_sort(cuda, eng):
    rng = np.random.default_rng(0)
    for n, n_rows in ((1, 5), (7, 3), (1000, 17), (50000, 1 << 20), (200000, 10_000_001)):
        ids = rng.integers(0, n_rows, size=n).astype(np.int64)
        keys, perm = eng.sort_ids(dev(ids, cuda), n_rows)
        order = np.argsort(ids, kind="stable")
        assert np.array_equal(host(perm).astype(np.int64) & 0xFFFFFFFF, order)
        assert np.array_equal(host(keys).astype(np.int64) & 0xFFFFFFFF, ids[order])


# ---- golden: optimizer steps ----------------------------------------------------------------

OPT_TAGS = [("SGD_l20", "SGD"), ("SGD_l20.001", "SGD"), ("Adam_l20", "Adam"),
            ("Adam_l20.0001", "Adam"), ("Adagrad_l20.0001", "Adagrad")]


@pytest.mark.parametrize("case", golden_cases())
@pytest.mark.parametrize("tag,opt", OPT_TAGS)
def test_dense_mode_matches_reference_fit(case, tag, opt, cuda, eng):
    """exact reference semantics: dense grads (segmented sum) + dense optimizer over all rows,
    two fit() iterations, vs the reference's own torch.optim run"""
    g = load_golden(case)
    lr, l2 = (float(x) for x in g[tag + "_hyper"])
    U, I = dev(g["U0"], cuda), dev(g["I0"], cuda)
    st = {k: torch.zeros_like(t) for k, t in (("mU", U), ("vU", U), ("mI", I), ("vI", I))}
    ex = 1e-3 * lr if opt in ("Adam", "Adagrad") else 0.0
    for step, (u, i) in enumerate(((g["uid"], g["iid"]), (g["uid2"], g["iid2"])), 1):
        uid, iid = dev(u, cuda), dev(i, cuda)
        B, C = i.shape
        _, loss_vec, gpred, ugrad = eng.bprmf_fwd_bwd(U, I, uid, iid, want_pred=False)
        assert_close(host(eng.reduce_sum(loss_vec, 1.0 / B))[0], g[tag + "_losses"][step - 1],
                     what=f"loss step {step}")
        GU = eng.embedding_dense_backward(ugrad, uid, U.shape[0])
        keys, perm = eng.sort_ids(iid, I.shape[0])
        GI = torch.zeros_like(I)
        eng.segmented_update(keys, perm, U, coef=gpred.reshape(-1), src_index=uid, div=C,
                             dense_grad=GI)
        h = eng.make_hyper(opt, lr=lr, l2=l2, step=step)
        eng.dense_update(I, GI, h, st["mI"], st["vI"])
        eng.dense_update(U, GU, h, st["mU"], st["vU"])
        assert_update_close(host(U), g["U0"], g[f"{tag}_U{step}"], what=f"dU step {step}", extra_atol=ex)
        assert_update_close(host(I), g["I0"], g[f"{tag}_I{step}"], what=f"dI step {step}", extra_atol=ex)


@pytest.mark.parametrize("case", golden_cases())
def test_train_step_sgd_l2_zero_matches_reference(case, cuda, eng):
    """row-wise SGD with l2 = 0 IS the reference's dense SGD (untouched rows have zero grad)"""
    g = load_golden(case)
    U, I = dev(g["U0"], cuda), dev(g["I0"], cuda)
    tr = eng.BprmfTrainer(U, I, opt="SGD", lr=0.05, l2=0.0)
    for step, (u, i) in enumerate(((g["uid"], g["iid"]), (g["uid2"], g["iid2"])), 1):
        loss = tr.step(dev(u, cuda), dev(i, cuda))
        assert_close(host(loss)[0], g["SGD_l20_losses"][step - 1], what=f"loss step {step}")
        assert_update_close(host(U), g["U0"], g[f"SGD_l20_U{step}"], what=f"dU step {step}")
        assert_update_close(host(I), g["I0"], g[f"SGD_l20_I{step}"], what=f"dI step {step}")


@pytest.mark.parametrize("case", ["bprmf_k1_d64", "bprmf_k99_d64_zipf", "bprmf_k5_d48", "bprmf_k3_d128"])
@pytest.mark.parametrize("opt,lr,l2", [("SGD", 0.05, 1e-3), ("Adam", 1e-3, 1e-4), ("Adagrad", 0.01, 1e-4)])
def test_train_step_rowwise_matches_oracle(case, opt, lr, l2, cuda, eng):
    """row-wise (lazy) mode: reference optimizer maths on the rows present in the batch only"""
    g = load_golden(case)
    Un, In = g["U0"].copy(), g["I0"].copy()
    sU, sI = O.new_state(Un, opt), O.new_state(In, opt)
    U, I = dev(g["U0"], cuda), dev(g["I0"], cuda)
    tr = eng.BprmfTrainer(U, I, opt=opt, lr=lr, l2=l2)
    ex = 1e-3 * lr if opt in ("Adam", "Adagrad") else 0.0
    for step, (u, i) in enumerate(((g["uid"], g["iid"]), (g["uid2"], g["iid2"]), (g["uid"], g["iid2"])), 1):
        want_loss, _ = O.bprmf_train_step(Un, In, sU, sI, u, i, opt=opt, lr=lr, l2=l2, step=step,
                                          rowwise=True)
        loss = tr.step(dev(u, cuda), dev(i, cuda))
        assert_close(host(loss)[0], want_loss, what=f"loss step {step}")
        assert_update_close(host(U), g["U0"], Un, what=f"dU step {step}", extra_atol=ex)
        assert_update_close(host(I), g["I0"], In, what=f"dI step {step}", extra_atol=ex)
    # rows never present in any batch are bit-identical to their initial values
    seen_i = np.unique(np.concatenate([g["iid"].ravel(), g["iid2"].ravel()]))
    mask = np.ones(In.shape[0], dtype=bool)
    mask[seen_i] = False
    assert np.array_equal(host(I)[mask], g["I0"][mask])


# ---- edge cases (seeded, vs the oracle) ------------------------------------------------------

def _random_problem(rng, n_users, n_items, d, B, C):
    U = rng.normal(0, 0.01, size=(n_users, d)).astype(np.float32)
    I = rng.normal(0, 0.01, size=(n_items, d)).astype(np.float32)
    uid = rng.integers(0, n_users, size=B).astype(np.int64)
    iid = rng.integers(0, n_items, size=(B, C)).astype(np.int64)
    return U, I, uid, iid


@pytest.mark.parametrize("d,B,C", [
    (64, 1, 2), (64, 3, 2), (64, 5, 3), (64, 9, 5), (64, 7, 9), (64, 5, 17), (64, 4, 33),
    (64, 3, 64), (64, 2, 65), (64, 6, 100), (64, 3, 128), (64, 2, 129), (64, 2, 300),
    (16, 5, 7), (16, 3, 40), (32, 37, 2), (32, 4, 100), (128, 5, 2), (128, 4, 5), (128, 3, 64),
    (128, 2, 100), (256, 3, 4), (48, 4, 6), (100, 3, 11), (1, 2, 3), (3, 4, 5),
])
def test_fused_shapes_vs_oracle(d, B, C, cuda, eng):
    """every (GS, CPL) register tiling, the generic fall-back, tuples-per-wave packing and
    ragged tails"""
    rng = np.random.default_rng(1000 * d + 10 * B + C)
    U, I, uid, iid = _random_problem(rng, 11, 50, d, B, C)
    U *= 30
    I *= 30  # scores O(1): exercises the softmax / sigmoid far from the linear regime
    pred, loss_vec, gpred, ugrad = eng.bprmf_fwd_bwd(*(dev(a, cuda) for a in (U, I, uid, iid)))
    want_pred = O.gather_dot(U, I, uid, iid)
    assert_close(host(pred), want_pred, what="pred")
    assert_close(host(eng.gather_dot(*(dev(a, cuda) for a in (U, I, uid, iid)))), want_pred,
                 what="pred(gather_dot)")
    rows, _, _, _ = O.bpr_loss_rows(want_pred)
    assert_close(host(loss_vec), rows, what="loss_vec", atol_scale=2e-5)
    g = O.bpr_loss_grad(want_pred)
    assert_close(host(gpred), g, what="gpred", atol_scale=2e-5)
    gu, _ = O.bprmf_row_grads(U, I, uid, iid, g)
    assert_close(host(ugrad), gu, what="ugrad", atol_scale=2e-5)


def test_loss_saturation_matches_clamp_semantics(cuda, eng):
    """P outside [1e-8, 1]: clamp(min=1e-8) gives loss = -log(1e-8) and ZERO gradient
    (models/BaseModel.py:185)"""
    pred = np.array([[-40.0, 30.0, 1.0, 2.0], [50.0, -3.0, -2.0, -1.0], [0.0, 0.0, 0.0, 0.0]],
                    dtype=np.float32)
    loss, loss_vec, gpred = eng.bpr_loss(dev(pred, cuda))
    rows, P, _, _ = O.bpr_loss_rows(pred)
    assert P[0] < 1e-8  # saturated low
    assert_close(host(loss_vec), rows, what="loss_vec")
    assert np.all(host(gpred)[0] == 0.0)
    assert_close(host(gpred), O.bpr_loss_grad(pred), what="gpred")
    assert np.isfinite(host(loss)[0])


@pytest.mark.parametrize("B,C", [(1, 2), (7, 5), (1000, 5), (259, 17), (130, 18), (65, 100)])
def test_bpr_loss_row_geometries(B, C, cuda, eng):
    """rc_bpr_loss_fwd_bwd packs four rows into a wave up to sixteen negatives (the reference's --num_neg 1, NeuMF's K = 4) and
    takes a wave per row beyond: both against the oracle, on either side of the switch and with row counts that leave a
    partly filled last wave / workgroup"""
    rng = np.random.default_rng(B * 131 + C)
    pred = (rng.normal(size=(B, C)) * 2.0).astype(np.float32)
    loss, loss_vec, gpred = eng.bpr_loss(dev(pred, cuda))
    rows, _, _, _ = O.bpr_loss_rows(pred)
    assert_close(host(loss_vec), rows, what=f"loss_vec B={B} C={C}", atol_scale=2e-5)
    assert_close(host(gpred), O.bpr_loss_grad(pred), what=f"gpred B={B} C={C}", atol_scale=2e-5)
    assert_close(host(loss)[0], rows.mean(), what="loss")
    loss2, loss_vec2, gpred2 = eng.bpr_loss(dev(pred, cuda))
    assert torch.equal(loss_vec, loss_vec2) and torch.equal(gpred, gpred2)


def test_long_segments_and_determinism(cuda, eng):
    """one hot row repeated thousands of times (deferred workgroup-per-row kernel), segments
    straddling the 32-occurrence threshold, and bit-reproducibility of the whole step"""
    rng = np.random.default_rng(7)
    n_users, n_items, d, B, C = 64, 500, 64, 700, 12
    U, I, uid, iid = _random_problem(rng, n_users, n_items, d, B, C)
    iid[:, 0] = 3                       # 700 occurrences of item 3
    iid[:31, 1] = 4                     # 31 (+ random hits): just below the threshold
    iid[:33, 2] = 5                     # 33: just above
    iid[:32, 3] = 6                     # exactly 32
    uid[:500] = 9                       # hot user: 500 occurrences
    Un, In = U.copy(), I.copy()
    O.bprmf_train_step(Un, In, {}, {}, uid, iid, opt="SGD", lr=0.1, l2=0.0, rowwise=True)
    outs = []
    for _ in range(2):
        Ud, Id = dev(U, cuda), dev(I, cuda)
        tr = eng.BprmfTrainer(Ud, Id, opt="SGD", lr=0.1, l2=0.0)
        loss = tr.step(dev(uid, cuda), dev(iid, cuda))
        outs.append((host(Ud), host(Id), host(loss)))
    assert_update_close(outs[0][0], U, Un, what="dU")
    assert_update_close(outs[0][1], I, In, what="dI")
    for a, b in zip(outs[0], outs[1]):
        assert np.array_equal(a, b), "training step is not bit-reproducible"


def test_all_tuples_identical_and_single_tuple(cuda, eng):
    rng = np.random.default_rng(8)
    U, I, uid, iid = _random_problem(rng, 5, 9, 64, 130, 4)
    uid[:] = 2
    iid[:] = iid[0]
    Un, In = U.copy(), I.copy()
    want, _ = O.bprmf_train_step(Un, In, {}, {}, uid, iid, opt="SGD", lr=0.1, rowwise=True)
    Ud, Id = dev(U, cuda), dev(I, cuda)
    loss = eng.BprmfTrainer(Ud, Id, opt="SGD", lr=0.1).step(dev(uid, cuda), dev(iid, cuda))
    assert_close(host(loss)[0], want, what="loss")
    assert_update_close(host(Ud), U, Un, what="dU")
    assert_update_close(host(Id), I, In, what="dI")


def test_empty_batch_and_bad_arguments(cuda, eng):
    from rechorus_amd import _lib
    U = torch.zeros(4, 64, device=cuda)
    I = torch.zeros(6, 64, device=cuda)
    uid = torch.zeros(0, dtype=torch.int64, device=cuda)
    iid = torch.zeros((0, 3), dtype=torch.int64, device=cuda)
    assert eng.gather_dot(U, I, uid, iid).shape == (0, 3)
    assert eng.gather_rows(I, uid).shape == (0, 64)
    with pytest.raises(_lib.RechorusHipError, match="C >= 2"):
        eng.bprmf_fwd_bwd(U, I, torch.zeros(2, dtype=torch.int64, device=cuda),
                          torch.zeros((2, 1), dtype=torch.int64, device=cuda))
    with pytest.raises(TypeError):
        eng.gather_dot(U, I, torch.zeros(2, dtype=torch.int32, device=cuda),
                       torch.zeros((2, 2), dtype=torch.int64, device=cuda))
    with pytest.raises(ValueError, match="not supported"):
        eng.make_hyper("RMSprop")
    # Adadelta exists for dense steps only: the row-wise step refuses it instead of guessing a semantics
    tr = eng.BprmfTrainer(torch.zeros(4, 64, device=cuda), torch.zeros(6, 64, device=cuda), opt="SGD")
    tr.hyper = eng.make_hyper("Adadelta")
    with pytest.raises(_lib.RechorusHipError, match="dense steps only"):
        tr.step(torch.zeros(2, dtype=torch.int64, device=cuda), torch.zeros((2, 3), dtype=torch.int64, device=cuda))


def test_embedding_dense_backward_vs_index_add(cuda, eng):
    """the generic nn.Embedding backward (any model file): segmented sum == index_add"""
    rng = np.random.default_rng(9)
    for d, n_rows, n in ((64, 40, 1000), (48, 10, 257), (8, 3, 100), (128, 1000, 64)):
        ids = rng.integers(0, n_rows, size=n).astype(np.int64)
        go = rng.normal(size=(n, d)).astype(np.float32)
        G = eng.embedding_dense_backward(dev(go, cuda), dev(ids, cuda), n_rows)
        want = np.zeros((n_rows, d), dtype=np.float64)
        np.add.at(want, ids, go.astype(np.float64))
        assert_close(host(G), want, what=f"dense grad d={d}")


@pytest.mark.parametrize("d,n_rows,n,hot", [(1, 300, 70000, 0.5), (1, 7, 131072, 0.0), (4, 100000, 40000, 0.2), (3, 50, 2048, 0.9),
                                            (2, 1, 5000, 0.0), (1, 5000, 513 * 17, 0.0)])
def test_narrow_embedding_dense_backward_in_tiles(d, n_rows, n, hot, cuda, eng):
    """[vocab, 1..4] tables (the first-order weights of the FM family, models/context/FM.py:49-57) under a large batch: the sorted
    positions are summed tile by tile (seg_narrow_tiles_kernel + seg_narrow_chains_kernel) -- rows whose occurrences span many
    tiles, rows inside one tile, a single row for everything, a ragged last tile; against a float64 index_add, twice (same bits)"""
    rng = np.random.default_rng(d * 1000 + n_rows)
    ids = rng.integers(0, n_rows, size=n).astype(np.int64)
    ids[rng.random(n) < hot] = n_rows // 2          # one very hot row
    go = rng.normal(size=(n, d)).astype(np.float32)
    G = eng.embedding_dense_backward(dev(go, cuda), dev(ids, cuda), n_rows, route="sort")
    G2 = eng.embedding_dense_backward(dev(go, cuda), dev(ids, cuda), n_rows, route="sort")
    assert torch.equal(G, G2)
    want = np.zeros((n_rows, d), dtype=np.float64)
    np.add.at(want, ids, go.astype(np.float64))
    cnt = np.bincount(ids, minlength=n_rows).max()
    assert_close(host(G), want, what=f"narrow dense grad d={d}", rtol=1e-5, abs_floor=2e-7 * np.sqrt(cnt) * 4)
    assert np.all(host(G)[np.bincount(ids, minlength=n_rows) == 0] == 0)


@pytest.mark.parametrize("opt,lr,l2", [("SGD", 0.05, 1e-3), ("Adam", 1e-3, 1e-4), ("Adagrad", 0.01, 1e-4)])
@pytest.mark.parametrize("d,B,C,n_items", [(64, 300, 100, 20000), (64, 513, 2, 700), (32, 100, 9, 300),
                                           (128, 64, 40, 2000), (16, 50, 100, 4000)])
def test_singleton_fast_path_is_bit_identical_to_segmented_path(opt, lr, l2, d, B, C, n_items, cuda, eng):
    """rows that occur once are updated inside the fused kernel; the result must equal the
    all-segmented pipeline bit for bit (same arithmetic, same operands)"""
    rng = np.random.default_rng(d + B + C)
    U, I, uid, iid = _random_problem(rng, 40, n_items, d, B, C)
    uid_d, iid_d = dev(uid, cuda), dev(iid, cuda)
    keys, perm = eng.sort_ids(iid_d, n_items)
    single = eng.mark_singletons(keys, perm)
    cnt = np.bincount(iid.ravel(), minlength=n_items)
    assert np.array_equal(host(single).reshape(B, C), (cnt[iid] == 1).astype(np.uint8))
    assert 0 < int(host(single).sum()) < B * C or C == 2
    h = eng.make_hyper(opt, lr=lr, l2=l2, step=3)
    res = []
    for fast in (False, True):
        Ud, Id = dev(U, cuda), dev(I, cuda)
        m = torch.full_like(Id, 0.25e-6) if opt != "SGD" else None
        v = torch.full_like(Id, 1e-9) if opt == "Adam" else None
        if fast:
            _, lv, gp, ug = eng.bprmf_fwd_bwd_update(Ud, Id, uid_d, iid_d, single, h, mI=m, vI=v)
        else:
            _, lv, gp, ug = eng.bprmf_fwd_bwd(Ud, Id, uid_d, iid_d, want_pred=False)
        eng.segmented_update(keys, perm, Ud, hyper=h, W=Id, m=m, v=v, coef=gp.reshape(-1),
                             src_index=uid_d, div=C, skip_singletons=fast)
        res.append((Id, m, v, lv, gp, ug))
    names = ("I", "m", "v", "loss_vec", "gpred", "ugrad")
    for nm, a, b in zip(names, res[0], res[1]):
        if a is None or torch.equal(a, b):
            continue
        diff = (a != b)
        msg = f"{nm}: {int(diff.sum())} elements differ, max |diff| {(a - b).abs().max().item():.3e}"
        if a.dim() == 2 and a.shape[0] == n_items:
            rows = torch.nonzero(diff.any(dim=1)).reshape(-1).cpu().numpy()
            msg += f"; rows differing: {len(rows)} of which singleton {int((cnt[rows] == 1).sum())}, multi {int((cnt[rows] > 1).sum())}"
        raise AssertionError(msg)
    assert not torch.equal(res[0][0], dev(I, cuda))


def test_dense_update_multi_equals_single_tensor_updates(cuda):
    """rc_dense_update_multi (one launch for a model's whole optimizer.step) == rc_dense_update per tensor,
    bit for bit, including unaligned views, tails and > 36 tensors"""
    from rechorus_amd import engine
    rng = np.random.default_rng(21)
    sizes = [1, 3, 4, 5, 64, 4096, 4097, 100_003, 7, 8192] * 5  # 50 tensors
    for opt in ("SGD", "Adam", "Adagrad"):
        big = torch.from_numpy(rng.normal(size=sum(sizes) + 64).astype(np.float32)).to(cuda)
        items_a, items_b = [], []
        off = 1  # views at odd offsets: not 16-byte aligned
        for k, n in enumerate(sizes):
            W = big[off:off + n] if k % 3 == 0 else torch.from_numpy(rng.normal(size=n).astype(np.float32)).to(cuda)
            off += n
            G = torch.from_numpy(rng.normal(size=n).astype(np.float32)).to(cuda)
            h = engine.make_hyper(opt, lr=0.01 * (1 + k % 3), l2=1e-3 if k % 2 else 0.0, step=3)
            m = torch.rand(n, device=cuda) if opt != "SGD" else None
            v = torch.rand(n, device=cuda) if opt == "Adam" else None
            items_a.append((W, G, h, m, v))
            items_b.append((W.clone(), G, h, None if m is None else m.clone(), None if v is None else v.clone()))
        engine.dense_update_multi(items_a, opt)
        for W, G, h, m, v in items_b:
            engine.dense_update(W, G, h, m, v)
        for (Wa, _, _, ma, va), (Wb, _, _, mb, vb) in zip(items_a, items_b):
            assert torch.equal(Wa, Wb)
            assert ma is None or torch.equal(ma, mb)
            assert va is None or torch.equal(va, vb)


def test_segment_end_search_over_segment_lengths(cuda, eng):
    """hot rows of many lengths -- around the 16-lane search's probe points (32 * 2^l), past the range of one
    gallop round (> 1 M occurrences), odd lengths, one at the very end of the key array -- summed through
    rc_segmented_update into a dense gradient; expected value from a float64 sum of the same rows"""
    rng = np.random.default_rng(31)
    d = 16
    lengths = [33, 64, 65, 1023, 1024, 1025, 4097, 32 * 2 ** 9, 32 * 2 ** 9 + 1, 70001, 1_200_003, 40, 777]
    keys = np.concatenate([np.full(n, 3 * k + 1, dtype=np.int64) for k, n in enumerate(lengths)])
    perm_in = rng.permutation(len(keys))
    ids = keys[perm_in]                                   # occurrence o has row id ids[o]
    src = rng.normal(size=(97, d)).astype(np.float32)      # gradient rows, indexed through src_index
    src_index = rng.integers(0, 97, size=len(ids)).astype(np.int64)
    coef = rng.normal(size=len(ids)).astype(np.float32)
    n_rows = 3 * len(lengths) + 2
    k_d, p_d = eng.sort_ids(dev(ids, cuda), n_rows)
    G = torch.zeros((n_rows, d), dtype=torch.float32, device=cuda)
    eng.segmented_update(k_d, p_d, dev(src, cuda), coef=dev(coef, cuda), src_index=dev(src_index, cuda), dense_grad=G)
    want = np.zeros((n_rows, d))
    np.add.at(want, ids, coef[:, None].astype(np.float64) * src[src_index])
    got = host(G)
    for k, n in enumerate(lengths):
        r = 3 * k + 1
        scale = np.abs(coef[ids == r][:, None] * src[src_index[ids == r]]).sum(0).max()
        assert np.abs(got[r] - want[r]).max() <= 2e-6 * scale, (n, np.abs(got[r] - want[r]).max(), scale)
    untouched = np.setdiff1d(np.arange(n_rows), 3 * np.arange(len(lengths)) + 1)
    assert np.all(got[untouched] == 0)
    G2 = torch.zeros_like(G)
    eng.segmented_update(k_d, p_d, dev(src, cuda), coef=dev(coef, cuda), src_index=dev(src_index, cuda), dense_grad=G2)
    assert torch.equal(G, G2)
