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
                                  "many_buckets", "tile_boundary", "huge_bucket", "hashed_100M", "hashed_hot", "hashed_sharded",
                                  "hashed_2M_keys", "skipped_ids", "skipped_ids_hashed", "narrow_dense", "narrow_one_id_per_bucket",
                                  "narrow_max"])
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
    elif case == "hashed_100M":   # too wide for id-range buckets (NeuMF config 4: 100 M items): hashed buckets + LDS hash tables
        ra, rb = 100_000_001, 10_000_001
        a, b = rng.integers(1, ra, size=327_680), _zipf(rng, rb, 65_536)
    elif case == "hashed_hot":    # hashed geometry with rows of thousands of occurrences and ids that collide modulo anything
        ra, rb = 3_000_000_000, 900_000_000
        a = np.concatenate([np.full(9_000, 2_999_999_999), np.full(300, 7), rng.integers(0, ra, size=40_000),
                            (np.arange(20_000) * 65_536) % ra])
        rng.shuffle(a)
        b = np.concatenate([np.full(33, 899_999_999), rng.integers(0, rb, size=2_000), np.zeros(34)])
    elif case == "hashed_sharded":  # what a rank of a sharded table plans: its local rows of a global batch, few keys, wide space
        ra, rb = 12_500_001, 0
        a, b = rng.integers(0, ra, size=41_000), None
    elif case == "hashed_2M_keys":  # 2,048 buckets per list, ~1,000 keys each
        ra, rb = 2_000_000_000, 50
        a, b = rng.integers(0, ra, size=2_000_000), rng.integers(0, rb, size=3)
    elif case in ("skipped_ids", "skipped_ids_hashed"):   # negative ids take no part (whole tiles of them, too)
        ra, rb = (60_000, 900) if case == "skipped_ids" else (90_000_000, 70_000_000)
        a = rng.integers(0, ra, size=40_000)
        a[rng.random(40_000) < 0.4] = -1
        a[8192:3 * 8192] = -1
        b = rng.integers(0, rb, size=9_000)
        b[::3] = -1
    elif case == "narrow_dense":  # dense batch over a small table (SASRec: candidates + histories over 8.7 K items): narrow buckets,
        ra, rb = 8_714, 1_500      # ballot ranks; one row with a third of all occurrences, padding ids skipped
        a = np.minimum(rng.zipf(1.15, size=300_000), ra - 1)
        a[rng.random(300_000) < 0.3] = 17
        a[rng.random(300_000) < 0.1] = -1
        b = np.minimum(rng.zipf(1.3, size=9_000), rb - 1)
    elif case == "narrow_one_id_per_bucket":
        ra, rb = 900, 40
        a, b = rng.integers(0, ra, size=20_000), rng.integers(0, rb, size=3_000)
    elif case == "narrow_max":    # 32 ids per bucket, 4,096 buckets
        ra, rb = 120_000, 11_000
        a, b = rng.integers(0, ra, size=600_000), rng.integers(0, rb, size=2_000)
    elif case == "huge_bucket":   # one bucket past 32,768 keys (32-bit LDS cells) next to ordinary ones (16-bit cells)
        ra, rb = 5_000_000, 3_000
        a = np.concatenate([rng.integers(0, 100, size=40_000), rng.integers(0, ra, size=30_000)])
        rng.shuffle(a)
        b = rng.integers(0, rb, size=500)
    else:                         # tile_boundary: exactly one tile, one over, one under
        ra, rb = 100_000, 100
        a, b = rng.integers(0, ra, size=8192 * 2 + 1), rng.integers(0, rb, size=8191)
    _check_plan(eng, cuda, a, ra, b, rb, list_single_a)


def test_bucket_plan_geometry_limits(cuda, eng):
    """id spaces too wide for id-range buckets take the hashed geometry; what neither geometry covers is refused"""
    from rechorus_amd import _lib
    lib = _lib.load()
    assert lib.rc_bucket_plan_supported(10, 10, 40_000_000, 10) == 1          # hashed
    assert lib.rc_bucket_plan_supported(6_553_600, 65_536, 10_000_001, 1_000_001) == 1   # direct (the bench shape)
    assert lib.rc_bucket_plan_supported(20_000_000, 10, 3_000_000_000, 10) == 0   # > 4,096 keys per hashed bucket
    assert lib.rc_bucket_plan_supported(10, 10, 5_000_000_000, 10) == 0           # ids do not fit 32 bits
    a = torch.zeros(10, dtype=torch.int64, device=cuda)
    with pytest.raises(_lib.RechorusHipError):
        eng.bucket_plan(a, 5_000_000_000, a, 10)
    # the multi-occurrence bitmap is indexed by id: direct geometry only
    with pytest.raises(_lib.RechorusHipError):
        eng.bucket_multi_bitmap(a, 400_000_000)


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
    hits = {}
    for mode in ("none", "all", "some", "wrong", "refill_in_place", "pingpong_generations", "stale_generation"):
        U, I = torch.from_numpy(U0).to(cuda), torch.from_numpy(I0).to(cuda)
        tr = eng.BprmfTrainer(U, I, opt=opt, lr=lr, l2=l2)
        losses = []
        # static buffers that are refilled in place, the way rechorus_amd/graph.py feeds its captured steps
        su = [torch.empty_like(batches[0][0]) for _ in range(2)]
        si = [torch.empty_like(batches[0][1]) for _ in range(2)]
        hit = 0
        for k, (u, i) in enumerate(batches):
            nxt, kw = None, {}
            if mode == "all" or (mode == "some" and k % 2 == 0):
                nxt = batches[(k + 1) % len(batches)]
            elif mode == "wrong":
                nxt = batches[(k + 2) % len(batches)]   # not the batch the next call brings
            elif mode == "refill_in_place":
                # ONE buffer pair, refilled with torch ops: the announced tensors ARE the next call's tensors (same
                # objects, same addresses) but their contents change in between -- the version counters say so, the
                # prepared plan (made from the OLD contents) must be discarded
                if k == 0:
                    su[0].copy_(u); si[0].copy_(i)
                u, i = su[0], si[0]
                nxt = (su[0], si[0])
            elif mode in ("pingpong_generations", "stale_generation"):
                # two buffer pairs written through raw pointers (torch's version counters do not move): the caller's
                # generation ids are the only thing that identifies a batch
                if k == 0:
                    su[0].untyped_storage().copy_(u.untyped_storage()); si[0].untyped_storage().copy_(i.untyped_storage())
                nu, ni = batches[(k + 1) % len(batches)]
                su[(k + 1) % 2].untyped_storage().copy_(nu.untyped_storage())
                si[(k + 1) % 2].untyped_storage().copy_(ni.untyped_storage())
                u, i = su[k % 2], si[k % 2]
                nxt = (su[(k + 1) % 2], si[(k + 1) % 2])
                kw = dict(generation=k + 1, next_generation=k + 2)
                if mode == "stale_generation" and k % 2 == 1:
                    # the announced buffers were refilled AFTER the announcement (here: with the batch after next, then
                    # -- below -- put right again): an honest caller bumps the generation, the stale plan is dropped
                    kw["generation"] = 1000 + k
            gen_used = kw["generation"] if "generation" in kw else tr._generation_of(u, i)
            hit += int(gen_used != 0 and tr._ticket.generation == gen_used)   # this call consumes the prepared plan
            losses.append(tr.step(u, i, next_batch=nxt, **kw).clone())
            if mode == "refill_in_place" and k + 1 < len(batches):
                nu, ni = batches[k + 1]
                su[0].copy_(nu); si[0].copy_(ni)      # in place, after it was announced
        torch.cuda.synchronize()
        hits[mode] = hit
        res.append((U, I, tr.mI, tr.vI, torch.cat(losses)))
    for other in res[1:]:
        for name, x, y in zip(("U", "I", "mI", "vI", "loss"), res[0], other):
            if x is not None:
                assert torch.equal(x, y), name
    n = len(batches)
    assert hits == {"none": 0, "all": n - 1, "some": (n - 1 + 1) // 2, "wrong": 0, "refill_in_place": 0,
                    "pingpong_generations": n - 1, "stale_generation": (n - 1) // 2}, hits


def test_ticket_is_matched_by_generation_not_by_pointer(cuda, eng):
    """C ABI level: a ticket prepared for generation g is consumed only by a call that brings g; the same id buffers
    with other contents under another generation re-plan (results equal the plain step's), and a cleared ticket is inert"""
    import ctypes as C
    from rechorus_amd import _lib
    rng = np.random.default_rng(5)
    n_users, n_items, d, B, Cn = 500, 30_000, 64, 1024, 100
    mk = lambda: (torch.from_numpy(_zipf(rng, n_users, B)).to(cuda),
                  torch.from_numpy(rng.integers(1, n_items, size=(B, Cn))).to(cuda))
    b1, b2, b3 = mk(), mk(), mk()
    U0 = torch.from_numpy(rng.normal(0, 0.01, size=(n_users, d)).astype(np.float32)).to(cuda)
    I0 = torch.from_numpy(rng.normal(0, 0.01, size=(n_items, d)).astype(np.float32)).to(cuda)

    def run(seq):
        U, I = U0.clone(), I0.clone()
        tr = eng.BprmfTrainer(U, I, opt="SGD", lr=0.05)
        out = []
        for (u, i, gen, nxt, ngen) in seq:
            tr.step(u, i, next_batch=nxt, generation=gen, next_generation=ngen)
            out.append(int(tr._ticket.generation))
        torch.cuda.synchronize()
        return U, I, out
    # reference: no look-ahead at all
    Ua, Ia, _ = run([(b1[0], b1[1], 0, None, 0), (b3[0], b3[1], 0, None, 0)])
    # b2 is announced under generation 7, but the buffers then hold b3 under generation 8 (copied in place)
    buf = (b2[0].clone(), b2[1].clone())
    U, I = U0.clone(), I0.clone()
    tr = eng.BprmfTrainer(U, I, opt="SGD", lr=0.05)
    tr.step(b1[0], b1[1], next_batch=buf, generation=1, next_generation=7)
    assert tr._ticket.generation == 7 and tr._ticket.B == B and tr._ticket.C == Cn and tr._ticket.flavour == 1
    buf[0].untyped_storage().copy_(b3[0].untyped_storage())
    buf[1].untyped_storage().copy_(b3[1].untyped_storage())
    tr.step(buf[0], buf[1], generation=8)
    assert tr._ticket.generation == 0
    torch.cuda.synchronize()
    assert torch.equal(U, Ua) and torch.equal(I, Ia)
    assert C.sizeof(_lib.StepTicket) == 72


@pytest.mark.parametrize("case", ["uniform_wide", "zipf_hot", "all_same", "small_range", "many_buckets", "huge_bucket"])
def test_multi_bitmap_matches_numpy(case, cuda, eng):
    """rc_bucket_multi_bitmap: bit (id) = 1 iff the row occurs at least twice -- for every id of the batch"""
    rng = np.random.default_rng(sum(map(ord, case)) + 1)
    if case == "uniform_wide":
        n_rows, ids = 10_000_001, rng.integers(1, 10_000_001, size=400_000)
    elif case == "zipf_hot":
        n_rows = 50_000
        ids = np.minimum(rng.zipf(1.2, size=120_000), n_rows - 1)
    elif case == "all_same":
        n_rows, ids = 77, np.full(20_000, 42)
    elif case == "small_range":
        n_rows, ids = 50, rng.integers(0, 50, size=10_000)
    elif case == "many_buckets":
        n_rows, ids = 33_000_000, rng.integers(0, 33_000_000, size=50_000)
    else:  # one id range far hotter than the 16-bit cells hold: the 32-bit instantiation takes the bucket
        n_rows = 100_000
        ids = np.concatenate([rng.integers(8192, 16384, size=70_000), rng.integers(0, n_rows, size=5_000)])
    ids = np.ascontiguousarray(ids, dtype=np.int64)
    bm = eng.bucket_multi_bitmap(torch.from_numpy(ids).to(cuda), n_rows).cpu().numpy().astype(np.int64) & 0xFFFFFFFF
    cnt = np.bincount(ids, minlength=n_rows)
    got = (bm[ids >> 5] >> (ids & 31)) & 1
    assert np.array_equal(got, (cnt[ids] >= 2).astype(np.int64))
    # ids of touched 32-id words that do not occur in the batch read 0 (their cells count 0)
    present = np.unique(ids)
    neigh = np.setdiff1d(np.unique(np.concatenate([present ^ 1, present ^ 7])), present)
    neigh = neigh[neigh < n_rows]
    assert not np.any((bm[neigh >> 5] >> (neigh & 31)) & 1)


@pytest.mark.parametrize("opt,lr,l2", [("SGD", 0.05, 1e-3), ("Adam", 1e-3, 1e-4)])
@pytest.mark.parametrize("d,B,C,n_items", [(64, 300, 100, 20000), (64, 513, 2, 700), (32, 100, 9, 300),
                                           (128, 64, 40, 2000), (16, 50, 100, 4000), (64, 70, 128, 5000), (32, 40, 200, 3000)])
def test_fused_update_bitmap_equals_flag_path(opt, lr, l2, d, B, C, n_items, cuda, eng):
    """the fused kernel with the singleton information as a bitmap over item ids (bucket plan) vs. as a flag byte per
    batch position (sort pipeline): bit-identical tables, state and gradients, for every lookup variant of the kernel
    (one tuple per wave with C <= 64 / <= 128, several tuples per wave, C > 128)"""
    rng = np.random.default_rng(d + B + C + 1)
    U = rng.normal(0, 0.1, size=(40, d)).astype(np.float32)
    I = rng.normal(0, 0.1, size=(n_items, d)).astype(np.float32)
    uid = torch.from_numpy(rng.integers(0, 40, size=B)).to(cuda)
    iid = torch.from_numpy(rng.integers(0, n_items, size=(B, C))).to(cuda)
    keys, perm = eng.sort_ids(iid, n_items)
    single = eng.mark_singletons(keys, perm)
    multi = eng.bucket_multi_bitmap(iid, n_items)
    h = eng.make_hyper(opt, lr=lr, l2=l2, step=3)
    res = []
    for use_bitmap in (False, True):
        Ud, Id = torch.from_numpy(U).to(cuda), torch.from_numpy(I).to(cuda)
        m = torch.full_like(Id, 0.25e-6) if opt != "SGD" else None
        v = torch.full_like(Id, 1e-9) if opt == "Adam" else None
        _, lv, gp, ug = eng.bprmf_fwd_bwd_update(Ud, Id, uid, iid, None if use_bitmap else single, h, mI=m, vI=v,
                                                 multi=multi if use_bitmap else None)
        res.append((Id, m, v, lv, gp, ug))
    for nm, a, b in zip(("I", "m", "v", "loss_vec", "gpred", "ugrad"), res[0], res[1]):
        assert a is None or torch.equal(a, b), nm
    assert not torch.equal(res[0][0], torch.from_numpy(I).to(cuda))


@pytest.mark.parametrize("case", ["sparse_hashed", "dense_direct", "hot"])
@pytest.mark.parametrize("d", [32, 64, 128])
def test_plan_row_sums_and_distinct(case, d, cuda, eng):
    """rc_plan_row_sums = index_add in ascending position order (vs numpy, float64 reference + fixed-order fp32 bound);
    rc_plan_distinct = torch.unique(return_inverse=True) up to the order of the distinct ids"""
    rng = np.random.default_rng(d + len(case))
    if case == "sparse_hashed":
        n_rows, ids = 50_000_000, rng.integers(0, 50_000_000, size=60_000)
        ids[::7] = ids[0]
    elif case == "dense_direct":
        n_rows, ids = 9_000, rng.integers(0, 9_000, size=60_000)
    else:
        n_rows = 3_000
        ids = np.minimum(rng.zipf(1.3, size=40_000), n_rows - 1)
    ids = np.ascontiguousarray(ids, dtype=np.int64)
    src = rng.normal(0, 1, size=(len(ids), d)).astype(np.float32)
    ids_d, src_d = torch.from_numpy(ids).to(cuda), torch.from_numpy(src).to(cuda)
    # distinct + inverse
    uniq, inverse = eng.unique_ids(ids_d, n_rows)
    u, inv = uniq.cpu().numpy(), inverse.cpu().numpy()
    assert len(u) == len(np.unique(ids)) and len(np.unique(u)) == len(u)
    assert np.array_equal(u[inv], ids)
    # row sums through the inverse index (what the sharded steps' de-duplicated exchanges do) and through the ids
    out = torch.zeros((len(u), d), device=cuda)
    eng.Plan(inverse, len(u), tag="t.rs").row_sums("a", out, src2=src_d)
    want = np.zeros((len(u), d), dtype=np.float64)
    np.add.at(want, inv, src.astype(np.float64))
    cnt = np.bincount(inv, minlength=len(u))
    tol = 1e-6 * np.sqrt(cnt)[:, None] * np.abs(src).max() * 4 + 1e-6
    assert np.all(np.abs(out.cpu().numpy() - want) <= tol + 1e-6 * np.abs(want))
    if n_rows <= 10_000:
        G = eng.embedding_dense_backward(src_d, ids_d, n_rows)
        want = np.zeros((n_rows, d), dtype=np.float64)
        np.add.at(want, ids, src.astype(np.float64))
        cnt = np.bincount(ids, minlength=n_rows)
        tol = 1e-6 * np.sqrt(np.maximum(cnt, 1))[:, None] * np.abs(src).max() * 4 + 1e-6
        assert np.all(np.abs(G.cpu().numpy() - want) <= tol + 1e-6 * np.abs(want))
        assert np.all(G.cpu().numpy()[cnt == 0] == 0)
        # bit-identical to the radix-sort route where both sum a row in one ascending chain (rows of at most 32 occurrences); hot
        # rows are cut into chunks whose size follows the row width on the plan route (plan_update.hip, side_chunk) and is 256 on
        # the sort route: the same numbers up to fp32 association
        keys, perm = eng.sort_ids(ids_d, n_rows)
        G2 = torch.zeros_like(G)
        eng.segmented_update(keys, perm, src_d, dense_grad=G2)
        short = torch.from_numpy(cnt <= 32).to(cuda)
        assert torch.equal(G[short], G2[short])
        assert np.all(np.abs(G.cpu().numpy() - G2.cpu().numpy()) <= 2 * tol)


@pytest.mark.parametrize("n,n_rows,d,hot", [(8192, 279_000, 64, 0), (8192, 279_000, 1, 0), (25_600, 8_714, 64, 0), (1, 5, 16, 0),
                                             (32_768, 1_000_000, 128, 0), (3000, 50, 32, 0), (8000, 300, 64, 6000), (8192, 40, 1, 7000),
                                             (777, 1000, 3, 0), (5632, 8_714, 64, 3300),
                                             (6000, 40, 16, 0), (4000, 30, 128, 2500), (8192, 9, 32, 0)])
def test_small_embedding_dense_backward(n, n_rows, d, hot, cuda, eng, monkeypatch):
    """embedding_dense_backward of a small id list (rc_small_row_sums: the small-batch plan workgroups + one lane-group per touched
    row) against a float64 index_add and the radix-sort route: CTR-sized and candidate-sized lists, the [vocab, 1] first-order
    tables, a table of 50 rows (every row hot), one row with thousands of occurrences (a plan workgroup past its LDS budget)"""
    rng = np.random.default_rng(n + d)
    ids = rng.integers(0, n_rows, size=n).astype(np.int64)
    if hot:
        ids[rng.permutation(n)[:hot]] = n_rows // 2
    src = rng.normal(size=(n, d)).astype(np.float32)
    want = np.zeros((n_rows, d), np.float64)
    np.add.at(want, ids, src.astype(np.float64))
    ids_d, src_d = torch.from_numpy(ids).to(cuda), torch.from_numpy(src).to(cuda)
    out = {}
    for small in (True, False, True):
        monkeypatch.setattr(eng, "_EDB_SMALL", small)
        monkeypatch.setattr(eng, "_EDB_SMALL_MAX", 32768)   # (the engine's default keeps lists past 8,192 ids on the plan / sort routes)
        G = eng.embedding_dense_backward(src_d, ids_d, n_rows, route="small" if small else ("sort" if hot else None))
        torch.cuda.synchronize()
        if small in out:
            assert torch.equal(G, out[small]), "not reproducible"
        out[small] = G
    got = out[True].cpu().numpy()
    cnt = np.bincount(ids, minlength=n_rows)
    tol = 1e-6 * np.sqrt(np.maximum(cnt, 1))[:, None] * np.abs(src).max() * 4 + 1e-6
    assert np.all(np.abs(got - want) <= tol + 1e-6 * np.abs(want))
    assert np.all(got[cnt == 0] == 0)
    if hot == 0 and d >= 16 and cnt.max() <= 32:
        assert torch.equal(out[True], out[False])   # both routes sum a row's occurrences in ascending position
    else:
        assert np.all(np.abs(got - out[False].cpu().numpy()) <= 2 * tol)


def test_neumf_trainer_plan_on_second_stream_equals_one_stream(cuda, eng, monkeypatch):
    """NeumfTrainer builds the batch's bucket plan on a second stream beside the head kernels (large batches); same kernels, same
    inputs: three row-wise Adam steps leave every table, dense parameter and optimizer state bit-identical to the one-stream order"""
    g = torch.Generator(device=cuda)
    n_users, n_items, d, l1, B, Cn = 20_000, 3_000_000, 128, 64, 3000, 5
    def run(overlap):
        monkeypatch.setattr(eng, "_NEUMF_OVERLAP", overlap)
        monkeypatch.setattr(eng, "_SAS_OVERLAP_MIN", 0)
        g.manual_seed(11)
        mk = lambda *sh: torch.empty(sh, device=cuda).normal_(0, 0.05, generator=g)
        P = {"mf_u": mk(n_users, d), "mlp_u": mk(n_users, d), "mf_i": mk(n_items, d), "mlp_i": mk(n_items, d),
             "W1": mk(l1, 2 * d), "b1": mk(l1), "w_out": mk(d + l1)}
        tr = eng.NeumfTrainer(P, opt="Adam", lr=1e-2, l2=1e-5, rowwise=True)
        r = np.random.default_rng(12)
        losses = []
        for _ in range(3):
            uid = torch.from_numpy(_zipf(r, n_users, B)).to(cuda)
            iid = torch.from_numpy(np.concatenate([_zipf(r, n_items, (B, 1)), r.integers(1, n_items, size=(B, Cn - 1))], axis=1)).to(cuda)
            losses.append(float(tr.step(uid, iid)[0]))
        torch.cuda.synchronize()
        assert (tr._side is not None) == overlap
        return P, tr.state, losses
    Pa, Sa, La = run(True)
    Pb, Sb, Lb = run(False)
    assert La == Lb
    for k in Pa:
        assert torch.equal(Pa[k], Pb[k]), k
        for st in ("m", "v"):
            assert torch.equal(Sa[k][st], Sb[k][st]), (k, st)


def test_neumf_and_sasrec_trainers_plan_equals_sort(cuda, eng, monkeypatch):
    """the trainers' table updates through the bucket plan and behind the radix sort: bit-identical item tables, dense
    parameters and state after two steps (both sum a row's gradient rows in ascending batch position)"""
    rng = np.random.default_rng(3)
    g = torch.Generator(device=cuda)
    # NeuMF, row-wise Adam, items sparse enough for the hashed geometry
    n_users, n_items, d, l1, B, Cn = 30_000, 6_000_000, 128, 64, 4096, 5
    def neumf(use_plan):
        monkeypatch.setattr(eng, "_USE_PLAN", use_plan)
        monkeypatch.setattr(eng, "_NEUMF_FUSED", False)   # the three-kernel step on both routes (the fused kernel sums in its own order: tests/test_gpu_neumf.py)
        g.manual_seed(1)
        mk = lambda *sh: torch.empty(sh, device=cuda).normal_(0, 0.05, generator=g)
        P = {"mf_u": mk(n_users, d), "mlp_u": mk(n_users, d), "mf_i": mk(n_items, d), "mlp_i": mk(n_items, d),
             "W1": mk(l1, 2 * d), "b1": mk(l1), "w_out": mk(d + l1)}
        tr = eng.NeumfTrainer(P, opt="Adam", lr=1e-2, l2=1e-5, rowwise=True)
        r = np.random.default_rng(4)
        for _ in range(1):   # (one step: after it the user tables differ by fp32 association, and with them every later gradient)
            uid = torch.from_numpy(_zipf(r, n_users, B)).to(cuda)
            iid = torch.from_numpy(np.concatenate([_zipf(r, n_items, (B, 1)), r.integers(1, n_items, size=(B, Cn - 1))], axis=1)).to(cuda)
            tr.step(uid, iid)
        torch.cuda.synchronize()
        return P, tr.state
    Pa, Sa = neumf(True)
    Pb, Sb = neumf(False)
    for k in Pa:
        if k.endswith("_u"):
            # the plan route sums a tuple's C per-candidate user gradients first, then the tuples of a user; the sort route
            # sums all occurrences of a user in one chain: same numbers up to fp32 association
            # (Adam normalises the step: an element whose gradient is rounding noise may move by lr either way -- allow a few)
            diff = (Pa[k] - Pb[k]).abs()
            assert float((diff > 1e-6).float().mean()) < 5e-3 and float(diff.max()) <= 2 * 1e-2 + 1e-6, k
            continue
        if k.endswith("_i"):
            # rows of at most 32 occurrences are summed in one ascending chain on both routes: bit-identical; the few hot rows are
            # cut into chunks of 64 (plan, 256-float pair rows) against 256 (sort): equal to rounding (and Adam's normalised step)
            same = (Pa[k] == Pb[k]).all(dim=1)
            assert float((~same).float().mean()) < 1e-4, k
            diff = (Pa[k] - Pb[k]).abs()
            assert float(diff.max()) <= 2 * 1e-2 + 1e-6, k
            continue
        assert torch.equal(Pa[k], Pb[k]), k
        for st in ("m", "v"):
            assert torch.equal(Sa[k][st], Sb[k][st]), (k, st)
    del Pa, Pb, Sa, Sb
    # SASRec, row-wise Adam, padded histories
    n_items, d, L, B, Cn = 9_000, 64, 50, 1024, 100
    def sasrec(use_plan):
        monkeypatch.setattr(eng, "_SASREC_PLAN", use_plan)
        # (the sorted route's head list sums in ascending batch position like the plan; its one-wave-per-row variant for small
        #  catalogues interleaves four lane-groups -- checked against both in tests/test_gpu_sasrec.py, within rounding)
        monkeypatch.setattr(eng, "_SEG_ROWS", False)
        g.manual_seed(2)
        mk = lambda *sh: torch.empty(sh, device=cuda).normal_(0, 0.05, generator=g)
        lay = {k: (mk(d, d) if k.startswith("W") else mk(d)) for k in eng.SAS_LAYER_KEYS}
        lay["ln1w"] += 1.0
        lay["ln2w"] += 1.0
        P = {"item_emb": mk(n_items, d), "pos_emb": mk(L + 1, d), "layers": [lay]}
        tr = eng.SasrecTrainer(P, 4, opt="Adam", lr=1e-2, l2=1e-5, rowwise=True)
        r = np.random.default_rng(6)
        for _ in range(2):
            lengths = torch.from_numpy(r.integers(1, L + 1, size=B)).to(cuda)
            hist = torch.from_numpy(_zipf(r, n_items, (B, L))).to(cuda)
            hist = (hist * (torch.arange(L, device=cuda)[None, :] < lengths[:, None])).contiguous()
            iid = torch.from_numpy(np.concatenate([_zipf(r, n_items, (B, 1)), r.integers(1, n_items, size=(B, Cn - 1))], axis=1)).to(cuda)
            tr.step(hist, lengths, iid)
        torch.cuda.synchronize()
        return P, tr
    Pa, ta = sasrec(True)
    Pb, tb = sasrec(False)
    assert torch.equal(Pa["item_emb"], Pb["item_emb"]) and torch.equal(Pa["pos_emb"], Pb["pos_emb"])
    for k in eng.SAS_LAYER_KEYS:
        assert torch.equal(Pa["layers"][0][k], Pb["layers"][0][k]), k
    sa, sb = ta._st(Pa["item_emb"]), tb._st(Pb["item_emb"])
    assert torch.equal(sa["m"], sb["m"]) and torch.equal(sa["v"], sb["v"])


@pytest.mark.parametrize("hashed", [False, True])
def test_plan_status_word_reports_ids_outside_their_table(hashed, cuda, eng):
    """rc_bucket_plan_status_ptr: 0 for a complete plan, 1 when an id lies outside its table (nn.Embedding would raise a device assert),
    in the id-range and in the hashed geometry; Plan.check() raises (RC_PLAN_CHECK=1 runs it on every plan)"""
    from rechorus_amd import _lib
    rng = np.random.default_rng(5)
    n_rows = 50_000_000 if hashed else 200_000
    ids = rng.integers(0, n_rows, size=20_000).astype(np.int64)
    good = eng.Plan(torch.from_numpy(ids).to(cuda), n_rows, tag="t.status")
    assert good.status() == 0
    good.check()
    ids[123] = n_rows + 5
    bad = eng.Plan(torch.from_numpy(ids).to(cuda), n_rows, tag="t.status")
    assert bad.status() == 1
    with pytest.raises(_lib.RechorusHipError, match="outside its table"):
        bad.check()


@pytest.mark.parametrize("n,n_rows", [(1, 1), (63, 2), (64, 200), (2048, 256), (2049, 257), (10_000, 65_536), (100_003, 65_537), (70_000, 7),
                                      (300_000, 1 << 24), (250_001, (1 << 24) + 1), (50_000, 1 << 32), (1_300_000, 280_000)])
def test_hand_written_radix_sort_is_the_stable_sort(n, n_rows, cuda, eng):
    """rc_sort_ids (csrc/sort_ids.hip: 8-bit LSD passes of count / offsets / ballot-ranked scatter) == numpy's stable argsort: sorted
    keys, and equal keys keep their batch order -- one to four passes, tiles with a ragged tail, a handful of very hot keys"""
    rng = np.random.default_rng(n)
    ids = rng.integers(0, n_rows, size=n).astype(np.int64)
    if n > 1000:
        ids[rng.integers(0, n, size=n // 3)] = n_rows - 1         # a hot row
    keys, perm = eng.sort_ids(torch.from_numpy(ids).to(cuda), n_rows)
    want = np.argsort(ids, kind="stable")
    assert np.array_equal(perm.cpu().numpy().astype(np.int64), want)
    assert np.array_equal(keys.cpu().numpy().view(np.uint32).astype(np.int64), ids[want])
    again = eng.sort_ids(torch.from_numpy(ids).to(cuda), n_rows)
    assert torch.equal(again[0], keys) and torch.equal(again[1], perm)


def test_two_lists_sorted_as_one(cuda, eng):
    """rc_sort_ids2: item ids and (offset) user ids of a step as one virtual list -- the user keys form the tail"""
    import ctypes as C
    from rechorus_amd import _lib
    rng = np.random.default_rng(7)
    n_a, n_b, n_items, n_users = 50_000, 3_000, 70_000, 900
    a, b = rng.integers(0, n_items, n_a).astype(np.int64), rng.integers(0, n_users, n_b).astype(np.int64)
    ad, bd = torch.from_numpy(a).to(cuda), torch.from_numpy(b).to(cuda)
    keys = torch.empty(n_a + n_b, dtype=torch.int32, device=cuda)
    perm = torch.empty(n_a + n_b, dtype=torch.int32, device=cuda)
    ws = torch.empty(_lib.load().rc_sort_workspace_bytes(n_a + n_b), dtype=torch.uint8, device=cuda)
    _lib.call("rc_sort_ids2", C.c_void_p(ad.data_ptr()), n_a, C.c_void_p(bd.data_ptr()), n_b, n_items, n_items + n_users,
              C.c_void_p(keys.data_ptr()), C.c_void_p(perm.data_ptr()), C.c_void_p(ws.data_ptr()), ws.numel(),
              C.c_void_p(torch.cuda.current_stream().cuda_stream))
    joint = np.concatenate([a, n_items + b])
    want = np.argsort(joint, kind="stable")
    assert np.array_equal(perm.cpu().numpy().astype(np.int64), want)
    assert np.array_equal(keys.cpu().numpy().view(np.uint32).astype(np.int64), joint[want])
