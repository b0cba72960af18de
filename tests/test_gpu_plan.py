"""GPU parity of the bucket plan (rc_bucket_plan, csrc/bucket_plan.hip) -- integer work, so bit-exact against the
numpy restatement oracle/plan_oracle.py -- and of the training step that consumes it: rc_bprmf_train_step gives
bit-identical tables and loss through the plan pipeline and through the radix-sort pipeline."""
import numpy as np
import pytest
import torch

from oracle import plan_oracle as PO

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng(cuda):
    from rechorus_amd import engine
    return engine


def _check_plan(eng, cuda, ids_a, range_a, ids_b, range_b, list_single_a):
    a = torch.from_numpy(np.ascontiguousarray(ids_a, dtype=np.int64)).to(cuda)
    b = None if ids_b is None else torch.from_numpy(np.ascontiguousarray(ids_b, dtype=np.int64)).to(cuda)
    out = eng.bucket_plan(a, range_a, b, range_b, list_single_a=list_single_a)
    want_a, want_b, want_single = PO.bucket_plan(ids_a, ids_b, list_single_a)
    occ = out["occ"].cpu().numpy().astype(np.int64) & 0xFFFFFFFF
    for rows, want, name in ((out["rows_a"], want_a, "a"), (out["rows_b"], want_b, "b")):
        r = rows.cpu().numpy().astype(np.int64) & 0xFFFFFFFF
        assert len(r) == len(want), f"list {name}: {len(r)} rows listed, {len(want)} expected"
        assert len(np.unique(r[:, 0])) == len(r), f"list {name}: a row is listed twice"
        assert np.all(r[:, 3] == 0)
        for row, start, n, _ in r:
            got = occ[start:start + n]
            exp = want[int(row)]
            assert np.array_equal(got, exp), f"list {name} row {row}: positions {got[:8]}.. != {exp[:8]}.."
    if list_single_a:
        assert out["single"] is None
    else:
        assert np.array_equal(out["single"].cpu().numpy(), want_single)


@pytest.mark.parametrize("list_single_a", [True, False])
@pytest.mark.parametrize("case", ["uniform_wide", "zipf_hot", "all_same", "tiny", "small_range", "edges", "no_b",
                                  "many_buckets", "tile_boundary", "huge_bucket"])
def test_bucket_plan_matches_the_oracle(case, list_single_a, cuda, eng):
    rng = np.random.default_rng(sum(map(ord, case)))
    if case == "uniform_wide":
        ra, rb = 10_000_001, 1_000_001
        a, b = rng.integers(1, ra, size=300_000), rng.integers(1, rb, size=3000)
    elif case == "zipf_hot":      # rows with thousands of occurrences, straddling the 32 / 256 thresholds
        ra, rb = 50_000, 2_000
        a = np.minimum(rng.zipf(1.2, size=120_000), ra - 1)
        b = np.minimum(rng.zipf(1.1, size=4_000), rb - 1)
    elif case == "all_same":
        ra, rb = 77, 5
        a, b = np.full(20_000, 42), np.full(600, 3)
    elif case == "tiny":
        ra, rb = 9, 4
        a, b = np.array([5, 5, 1]), np.array([2])
    elif case == "small_range":   # fewer ids than one bucket at the widest shift
        ra, rb = 50, 20
        a, b = rng.integers(0, ra, size=10_000), rng.integers(0, rb, size=100)
    elif case == "edges":         # first / last id of the tables, id 0
        ra, rb = 1_234_567, 4_097
        a = np.concatenate([np.zeros(3), np.full(2, ra - 1), rng.integers(0, ra, size=5000)])
        b = np.array([0, rb - 1, rb - 1, 0, 17])
    elif case == "no_b":
        ra, rb = 200_000, 0
        a, b = rng.integers(0, ra, size=70_000), None
    elif case == "many_buckets":  # the widest supported id space: 4,096 buckets of 8,192 ids
        ra, rb = 4000 * 8192, 96 * 8192
        a, b = rng.integers(0, ra, size=50_000), rng.integers(0, rb, size=1_000)
    elif case == "huge_bucket":   # one bucket past 32,768 keys (32-bit LDS cells) next to ordinary ones (16-bit cells)
        ra, rb = 5_000_000, 3_000
        a = np.concatenate([rng.integers(0, 100, size=40_000), rng.integers(0, ra, size=30_000)])
        rng.shuffle(a)
        b = rng.integers(0, rb, size=500)
    else:                         # tile_boundary: exactly one tile, one over, one under
        ra, rb = 100_000, 100
        a, b = rng.integers(0, ra, size=8192 * 2 + 1), rng.integers(0, rb, size=8191)
    _check_plan(eng, cuda, a, ra, b, rb, list_single_a)


def test_bucket_plan_rejects_too_wide_id_spaces(cuda, eng):
    from rechorus_amd import _lib
    assert _lib.load().rc_bucket_plan_supported(10, 10, 40_000_000, 10) == 0
    a = torch.zeros(10, dtype=torch.int64, device=cuda)
    with pytest.raises(_lib.RechorusHipError):
        eng.bucket_plan(a, 40_000_000, a, 10)


def _zipf(rng, n_rows, size):
    ranks = np.exp(rng.uniform(0, np.log(n_rows - 1), size=size)).astype(np.int64)
    return (ranks * 2654435761 % (n_rows - 1)) + 1


@pytest.mark.parametrize("opt,lr,l2", [("SGD", 0.05, 0.0), ("SGD", 0.05, 1e-3), ("Adam", 1e-3, 1e-4), ("Adagrad", 0.01, 1e-4)])
@pytest.mark.parametrize("d,B,C,n_users,n_items", [(64, 2048, 100, 3000, 40_000), (64, 700, 12, 64, 500),
                                                   (32, 513, 2, 40, 700), (128, 300, 40, 100, 5000),
                                                   (16, 1000, 100, 50, 100_000)])
def test_train_step_plan_pipeline_equals_sort_pipeline(opt, lr, l2, d, B, C, n_users, n_items, cuda, eng):
    """same arithmetic, same summation order: the two pipelines must agree bit for bit (tables, state, loss),
    hot rows (chunked path) and single-occurrence rows (fused fast path) included"""
    from rechorus_amd import _lib
    rng = np.random.default_rng(d + B + C)
    U0 = rng.normal(0, 0.01, size=(n_users, d)).astype(np.float32)
    I0 = rng.normal(0, 0.01, size=(n_items, d)).astype(np.float32)
    batches = []
    for _ in range(2):
        uid = _zipf(rng, n_users, B)
        iid = np.concatenate([_zipf(rng, n_items, (B, 1)), rng.integers(1, n_items, size=(B, C - 1))], axis=1)
        batches.append((torch.from_numpy(uid).to(cuda), torch.from_numpy(iid).to(cuda)))
    lib = _lib.load()
    res = []
    prev = lib.rc_bprmf_step_pipeline(-1)
    try:
        for mode in (1, 3, 2, 0):  # sort pipeline, bucket plan on two streams / on one stream, automatic choice
            lib.rc_bprmf_step_pipeline(mode)
            U, I = torch.from_numpy(U0).to(cuda), torch.from_numpy(I0).to(cuda)
            tr = eng.BprmfTrainer(U, I, opt=opt, lr=lr, l2=l2)
            losses = [tr.step(u, i).clone() for u, i in batches]
            res.append((U, I, tr.mU, tr.vU, tr.mI, tr.vI, torch.cat(losses)))
    finally:
        lib.rc_bprmf_step_pipeline(prev)
    assert lib.rc_bucket_plan_supported(B * C, B, n_items, n_users) == 1
    for other in (1, 2):
        for name, x, y in zip(("U", "I", "mU", "vU", "mI", "vI", "loss"), res[0], res[other]):
            if x is None:
                continue
            assert torch.equal(x, y), f"{name} (pipeline {other}): {int((x != y).sum())} elements differ, max |diff| {(x - y).abs().max().item():.3e}"
    assert not torch.equal(res[0][1], torch.from_numpy(I0).to(cuda))
    # automatic choice: the bucket plan (bit-identical again) or, up to 32,768 row ids, the two-launch small-batch step,
    # which sums rows of more than 32 occurrences in another (fixed) order: equal to fp32 summation order
    from conftest import assert_update_close
    small = B * C + B <= 32768
    assert torch.equal(res[0][6][0], res[3][6][0]), "loss of the first step"
    for name, x, y, x0 in (("U", res[0][0], res[3][0], U0), ("I", res[0][1], res[3][1], I0)):
        if not small:
            assert torch.equal(x, y), name
        else:
            assert_update_close(y.cpu().numpy(), x0, x.cpu().numpy(), what=f"{name} (small-batch step)",
                                extra_atol=0.0 if opt == "SGD" else 1e-3 * lr)


@pytest.mark.parametrize("opt,lr,l2", [("SGD", 0.05, 1e-3), ("Adam", 1e-3, 1e-4), ("Adagrad", 0.01, 1e-4)])
@pytest.mark.parametrize("d", [16, 64, 128])
def test_plan_update_matches_the_rowwise_oracle(opt, lr, l2, d, cuda, eng):
    """rc_plan_update / rc_plan_update_pair (generic plan-driven row updates, hot rows included) vs the numpy oracle:
    index_add of the per-occurrence gradient rows, then the optimizer on the touched rows"""
    from conftest import assert_update_close
    from oracle import bprmf_oracle as O
    rng = np.random.default_rng(d)
    n_rows, n_src, n_a, n_b, div = 3000, 50, 20_000, 700, 4
    ids_a = np.minimum(rng.zipf(1.3, size=n_a), n_rows - 1).astype(np.int64)   # hot rows: thousands of occurrences
    ids_b = rng.integers(0, 40, size=n_b).astype(np.int64)
    plan = eng.Plan(torch.from_numpy(ids_a).to(cuda), n_rows, torch.from_numpy(ids_b).to(cuda), 40, tag="test.plan")
    h = eng.make_hyper(opt, lr=lr, l2=l2, step=2)
    mk = lambda *s: rng.normal(0, 0.1, size=s).astype(np.float32)
    ex = 1e-3 * lr if opt != "SGD" else 0.0

    def state(W):
        st = O.new_state(W, opt)
        for k in st:
            st[k] += np.float32(1e-3)
        return st, {k: torch.from_numpy(v.copy()).to(cuda) for k, v in st.items()}

    # (1) list a, gradient rows rebuilt as coef[o] * src[src_index[o // div]]
    W0, src, coef = mk(n_rows, d), mk(n_src, d), mk(n_a)
    src_index = rng.integers(0, n_src, size=(n_a + div - 1) // div).astype(np.int64)
    st, st_d = state(W0)
    W = torch.from_numpy(W0.copy()).to(cuda)
    plan.update("a", W, h, m=st_d.get("m"), v=st_d.get("v"), coef=torch.from_numpy(coef).to(cuda), src=torch.from_numpy(src).to(cuda),
                src_index=torch.from_numpy(src_index).to(cuda), div=div)
    Wn = W0.copy()
    G = O.embedding_dense_backward(coef[:, None] * src[src_index[np.arange(n_a) // div]], ids_a, n_rows)
    O.opt_step_dense(Wn, G, st, opt, lr, l2, step=2, rows=np.unique(ids_a))
    assert_update_close(W.cpu().numpy(), W0, Wn, what="plan.update list a", extra_atol=ex)
    # (2) list b, plain per-occurrence rows (positions of list b start at n_a)
    Wb0, gb = mk(40, d), mk(n_b, d)
    st, st_d = state(Wb0)
    Wb = torch.from_numpy(Wb0.copy()).to(cuda)
    plan.update("b", Wb, h, m=st_d.get("m"), v=st_d.get("v"), src2=torch.from_numpy(gb).to(cuda), n_split=n_a)
    Wn = Wb0.copy()
    O.opt_step_dense(Wn, O.embedding_dense_backward(gb, ids_b, 40), st, opt, lr, l2, step=2, rows=np.unique(ids_b))
    assert_update_close(Wb.cpu().numpy(), Wb0, Wn, what="plan.update list b", extra_atol=ex)
    # (3) two tables that share list a's ids, one pass
    Wa0, Wc0, ga, gc = mk(n_rows, d), mk(n_rows, d), mk(n_a, d), mk(n_a, d)
    sa, sa_d = state(Wa0)
    sc, sc_d = state(Wc0)
    Wa, Wc = torch.from_numpy(Wa0.copy()).to(cuda), torch.from_numpy(Wc0.copy()).to(cuda)
    plan.update_pair("a", Wa, Wc, torch.from_numpy(ga).to(cuda), torch.from_numpy(gc).to(cuda), h, ma=sa_d.get("m"),
                     va=sa_d.get("v"), mb=sc_d.get("m"), vb=sc_d.get("v"))
    for W_, W0_, g_, st_ in ((Wa, Wa0, ga, sa), (Wc, Wc0, gc, sc)):
        Wn = W0_.copy()
        O.opt_step_dense(Wn, O.embedding_dense_backward(g_, ids_a, n_rows), st_, opt, lr, l2, step=2, rows=np.unique(ids_a))
        assert_update_close(W_.cpu().numpy(), W0_, Wn, what="plan.update_pair", extra_atol=ex)


@pytest.mark.parametrize("opt,lr,l2", [("SGD", 0.05, 1e-3), ("Adam", 1e-3, 1e-4), ("Adagrad", 0.01, 1e-4)])
@pytest.mark.parametrize("kind", ["default_b256_k99", "duplicates_overflow", "tiny", "d128_users_hot"])
def test_small_batch_step_vs_oracle(kind, opt, lr, l2, cuda, eng):
    """the two-launch small-batch step (automatic up to 32,768 row ids) vs the numpy oracle, three steps: the reference's
    default batch shape, a batch whose ids are thousands of duplicates of a few rows (one plan workgroup overflows its
    LDS budget and selects its rows by scans; hot rows summed by whole waves), a batch smaller than one workgroup"""
    from conftest import assert_close, assert_update_close
    from oracle import bprmf_oracle as O
    from rechorus_amd import _lib
    rng = np.random.default_rng(len(kind))
    if kind == "default_b256_k99":
        n_users, n_items, d, B, C = 5000, 60_000, 64, 256, 100
        mk = lambda: (_zipf(rng, n_users, B), np.concatenate([_zipf(rng, n_items, (B, 1)), rng.integers(1, n_items, size=(B, C - 1))], axis=1))
    elif kind == "duplicates_overflow":
        n_users, n_items, d, B, C = 40, 3000, 64, 300, 100   # 30,300 ids, almost all of them multiples of 256 (one owner workgroup)
        def mk():
            iid = 256 * rng.integers(1, 6, size=(B, C))
            m = rng.random((B, C)) < 0.05
            iid[m] = rng.integers(1, n_items, size=int(m.sum()))
            iid[:, 0] = rng.integers(1, n_items, size=B)
            return rng.integers(0, 3, size=B).astype(np.int64), iid.astype(np.int64)
    elif kind == "tiny":
        n_users, n_items, d, B, C = 9, 30, 32, 3, 5
        mk = lambda: (rng.integers(0, n_users, size=B).astype(np.int64), rng.integers(0, n_items, size=(B, C)).astype(np.int64))
    else:
        n_users, n_items, d, B, C = 6, 900, 128, 500, 4
        mk = lambda: (rng.integers(0, n_users, size=B).astype(np.int64), rng.integers(0, n_items, size=(B, C)).astype(np.int64))
    assert B * C + B <= 32768 and _lib.load().rc_bprmf_step_pipeline(-1) == 0
    U0 = rng.normal(0, 0.01, size=(n_users, d)).astype(np.float32)
    I0 = rng.normal(0, 0.01, size=(n_items, d)).astype(np.float32)
    Un, In = U0.copy(), I0.copy()
    sU, sI = O.new_state(Un, opt), O.new_state(In, opt)
    U, I = torch.from_numpy(U0).to(cuda), torch.from_numpy(I0).to(cuda)
    tr = eng.BprmfTrainer(U, I, opt=opt, lr=lr, l2=l2)
    ex = 1e-3 * lr if opt in ("Adam", "Adagrad") else 0.0
    seen = []
    for step in (1, 2, 3):
        u, i = mk()
        seen.append(i.ravel())
        want_loss, _ = O.bprmf_train_step(Un, In, sU, sI, u, i, opt=opt, lr=lr, l2=l2, step=step, rowwise=True)
        loss = tr.step(torch.from_numpy(u).to(cuda), torch.from_numpy(i).to(cuda))
        assert_close(loss.cpu().numpy()[0], want_loss, what=f"loss step {step}")
        assert_update_close(U.cpu().numpy(), U0, Un, what=f"dU step {step}", extra_atol=ex, outlier_atol=lr * step)
        assert_update_close(I.cpu().numpy(), I0, In, what=f"dI step {step}", extra_atol=ex, outlier_atol=lr * step)
    mask = np.ones(n_items, dtype=bool)
    mask[np.unique(np.concatenate(seen))] = False
    assert np.array_equal(I.cpu().numpy()[mask], I0[mask])   # untouched rows bit-identical
    # run-to-run reproducibility (deterministic layout, fixed summation orders)
    U2, I2 = torch.from_numpy(U0).to(cuda), torch.from_numpy(I0).to(cuda)
    tr2 = eng.BprmfTrainer(U2, I2, opt=opt, lr=lr, l2=l2)
    rng = np.random.default_rng(len(kind))
    rng.normal(0, 0.01, size=(n_users, d)); rng.normal(0, 0.01, size=(n_items, d))
    for step in (1, 2, 3):
        u, i = mk()
        tr2.step(torch.from_numpy(u).to(cuda), torch.from_numpy(i).to(cuda))
    assert torch.equal(U, U2) and torch.equal(I, I2)


@pytest.mark.parametrize("opt,lr,l2", [("SGD", 0.05, 1e-3), ("Adam", 1e-3, 1e-4)])
def test_train_step_with_look_ahead_is_bit_identical(opt, lr, l2, cuda, eng):
    """rc_bprmf_train_step_ahead: the grouping front of the FOLLOWING batch runs beside this step's row updates -- same
    tables, state and losses bit for bit, whether every step looks ahead, only some do, or the announced batch is not the
    one that follows (the front is then redone)"""
    rng = np.random.default_rng(17)
    n_users, n_items, d, B, C = 3000, 40_000, 64, 2048, 100
    U0 = rng.normal(0, 0.01, size=(n_users, d)).astype(np.float32)
    I0 = rng.normal(0, 0.01, size=(n_items, d)).astype(np.float32)
    batches = []
    for _ in range(5):
        uid = _zipf(rng, n_users, B)
        iid = np.concatenate([_zipf(rng, n_items, (B, 1)), rng.integers(1, n_items, size=(B, C - 1))], axis=1)
        batches.append((torch.from_numpy(uid).to(cuda), torch.from_numpy(iid).to(cuda)))
    res = []
    for mode in ("none", "all", "some", "wrong"):
        U, I = torch.from_numpy(U0).to(cuda), torch.from_numpy(I0).to(cuda)
        tr = eng.BprmfTrainer(U, I, opt=opt, lr=lr, l2=l2)
        losses = []
        for k, (u, i) in enumerate(batches):
            nxt = None
            if mode == "all" or (mode == "some" and k % 2 == 0):
                nxt = batches[(k + 1) % len(batches)]
            elif mode == "wrong":
                nxt = batches[(k + 2) % len(batches)]   # not the batch the next call brings
            losses.append(tr.step(u, i, next_batch=nxt).clone())
        torch.cuda.synchronize()
        res.append((U, I, tr.mI, tr.vI, torch.cat(losses)))
    for other in res[1:]:
        for name, x, y in zip(("U", "I", "mI", "vI", "loss"), res[0], other):
            if x is not None:
                assert torch.equal(x, y), name
