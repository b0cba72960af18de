"""GPU parity of the batch-assembly and evaluation kernels (csrc/sampler.hip, csrc/eval_rank.hip):
bit-exact against the oracle for the integer work, score ties aside for the fp32 MFMA ranking."""
import numpy as np
import pytest
import torch

from oracle import sampler_oracle as S
from test_sampler_cpu import make_clicked

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng(cuda):
    from rechorus_amd import engine
    return engine


def dev(x, cuda):
    return torch.from_numpy(np.ascontiguousarray(x)).to(cuda)


@pytest.mark.parametrize("K", [1, 4, 99])
def test_sample_negatives_bit_exact(K, cuda, eng):
    rng = np.random.default_rng(K)
    n_users, n_items = 40, 60
    sets, ptr, items = make_clicked(n_users, n_items, rng, heavy_user=5)
    users = rng.integers(0, n_users, size=97)
    users[:4] = 5
    for seed, base in ((0, 0), (2**63 + 12345, 10**12)):
        got = eng.sample_negatives(dev(users, cuda), K, n_items, dev(ptr, cuda), dev(items, cuda), seed=seed, base_index=base)
        want = S.sample_negatives(users, K, n_items, ptr, items, seed=seed, base_index=base)
        assert np.array_equal(got.cpu().numpy(), want)
    got = eng.sample_negatives(dev(users, cuda), K, n_items, seed=3)  # no clicked sets: plain uniform
    assert np.array_equal(got.cpu().numpy(), S.sample_negatives(users, K, n_items, seed=3))
    assert eng.sample_negatives(dev(users[:0], cuda), K, n_items, seed=3).shape == (0, K)


def test_sample_negatives_exhausted_attempts_bit_exact(cuda, eng):
    """users who clicked (nearly) the whole catalogue: the direct selection after 1,024 rejected draws, bit for bit"""
    n_items, free = 3001, {17, 2999}
    clicked = np.array([i for i in range(1, n_items) if i not in free], dtype=np.int64)
    everything = np.arange(1, n_items, dtype=np.int64)
    ptr = np.array([0, len(clicked), len(clicked) + len(everything), len(clicked) + len(everything) + 3], dtype=np.int64)
    items = np.concatenate([clicked, everything, np.array([5, 6, 7], dtype=np.int64)])
    users = np.array([0, 2, 0, 1, 2, 0], dtype=np.int64)
    got = eng.sample_negatives(dev(users, cuda), 16, n_items, dev(ptr, cuda), dev(items, cuda), seed=5, base_index=77).cpu().numpy()
    want = S.sample_negatives(users, 16, n_items, ptr, items, seed=5, base_index=77)
    assert np.array_equal(got, want)
    assert set(np.unique(got[users == 0])) == free and not np.isin(got[users == 2], [5, 6, 7]).any()


def test_sample_negatives_full_size_properties(cuda, eng):
    """config-2 scale: 65,536 rows x 99 negatives over 10 M items; uniformity, range, exclusion"""
    n_users, n_items, n, K = 100_000, 10_000_001, 65_536, 99
    g = torch.Generator(device=cuda).manual_seed(0)
    users = torch.randint(1, n_users, (n,), device=cuda, generator=g)
    # every user clicked the 64 items {u, u+1, ..., u+63} (sorted, CSR)
    ptr = torch.arange(0, (n_users + 1) * 64, 64, device=cuda)
    items = (torch.arange(n_users, device=cuda)[:, None] + torch.arange(64, device=cuda)[None, :]).reshape(-1).clamp_(min=1)
    items, _ = items.reshape(n_users, 64).sort(dim=1)
    neg = eng.sample_negatives(users, K, n_items, ptr, items.reshape(-1).contiguous(), seed=11)
    assert int(neg.min()) >= 1 and int(neg.max()) < n_items
    delta = neg - users[:, None]
    assert not bool(((delta >= 0) & (delta < 64)).any())  # never a clicked item
    counts = torch.bincount((neg.reshape(-1) * 100 // n_items), minlength=100).double()
    expect = counts.sum() / 100
    assert float(((counts - expect) ** 2 / expect).sum()) < 200  # chi2, 99 dof
    again = eng.sample_negatives(users, K, n_items, ptr, items.reshape(-1).contiguous(), seed=11)
    assert torch.equal(neg, again)


def test_assemble_and_history(cuda, eng):
    rng = np.random.default_rng(4)
    n, K, B = 500, 5, 64
    users, items = rng.integers(1, 50, size=n), rng.integers(1, 300, size=n)
    neg = rng.integers(1, 300, size=(n, K))
    idx = rng.permutation(n)[:B]
    u, cand = eng.assemble_candidates(dev(idx, cuda), dev(users, cuda), dev(items, cuda), dev(neg, cuda))
    assert np.array_equal(u.cpu().numpy(), users[idx])
    assert np.array_equal(cand.cpu().numpy(), np.concatenate([items[idx, None], neg[idx]], axis=1))
    u, cand = eng.assemble_candidates(dev(idx, cuda), dev(users, cuda), dev(items, cuda), None)  # eval of CTR-like rows
    assert cand.shape == (B, 1) and np.array_equal(cand.cpu().numpy()[:, 0], items[idx])
    # histories: CSR of time-ordered (item, time) per user
    his = {uu: [(int(rng.integers(1, 300)), 100 * t + uu) for t in range(rng.integers(1, 30))] for uu in range(50)}
    ptr = np.zeros(51, dtype=np.int64)
    for uu in range(50):
        ptr[uu + 1] = ptr[uu] + len(his[uu])
    flat_i = np.array([x[0] for uu in range(50) for x in his[uu]], dtype=np.int64)
    flat_t = np.array([x[1] for uu in range(50) for x in his[uu]], dtype=np.int64)
    pos = np.array([rng.integers(0, len(his[uu]) + 1) for uu in users], dtype=np.int64)
    for L in (1, 5, 20, 50):
        h, t, ln = eng.gather_history(dev(idx, cuda), dev(users, cuda), dev(pos, cuda), dev(ptr, cuda), dev(flat_i, cuda),
                                      L, his_times=dev(flat_t, cuda))
        for b, i in enumerate(idx):
            wi, wt, wn = S.history_window(users[i], pos[i], his, L)
            assert np.array_equal(h[b].cpu().numpy(), wi) and np.array_equal(t[b].cpu().numpy(), wt) and int(ln[b]) == wn
    h, t, ln = eng.gather_history(None, dev(users, cuda), dev(pos, cuda), dev(ptr, cuda), dev(flat_i, cuda), 7)
    assert t is None and h.shape == (n, 7) and int(ln[3]) == min(pos[3], 7)


def test_target_rank_and_metrics(cuda, eng):
    from oracle import bprmf_oracle as BO
    rng = np.random.default_rng(5)
    for n, C in ((1, 1), (7, 2), (300, 100), (33, 1000), (5, 64), (2, 65)):
        pred = rng.normal(size=(n, C)).astype(np.float32)
        if C > 3:
            pred[::2, 3] = pred[::2, 0]  # ties count against the target
        rank = eng.target_rank(dev(pred, cuda))
        assert np.array_equal(rank.cpu().numpy(), S.target_rank(pred))
        got = eng.rank_metrics(rank, [1, 5, 10], ["HR", "NDCG"])
        want = BO.evaluate_method(pred, [1, 5, 10], ["HR", "NDCG"])
        assert got.keys() == want.keys() and all(abs(got[k] - want[k]) < 1e-12 for k in got)
    with pytest.raises(ValueError):
        eng.rank_metrics(rank, [5], ["MAP"])


@pytest.mark.parametrize("case", ["testall_bprmf_d64", "testall_bprmf_d32"])
def test_full_catalogue_rank_matches_the_reference_exactly(case, cuda, eng):
    """rc_full_catalogue_rank vs the ranks the reference's own --test_all evaluation produced (BaseRunner.predict +
    evaluate_method on its BPRMF, tests/golden/make_golden_testall.py); the fixture has no near-ties, so the ranks are exact"""
    import os
    from conftest import GOLDEN_DIR
    g = np.load(os.path.join(GOLDEN_DIR, case + ".npz"))
    Uv = g["U"][g["users"]]
    rank, tscore = eng.full_catalogue_rank(dev(Uv, cuda), dev(g["I"], cuda), dev(g["users"], cuda), dev(g["targets"], cuda),
                                           dev(g["clicked_ptr"], cuda), dev(g["clicked_items"], cuda))
    assert np.array_equal(rank.cpu().numpy().astype(np.int64), g["gt_rank"])
    assert np.allclose(tscore.cpu().numpy(), g["target_score"], rtol=1e-5, atol=1e-6)
    got = eng.rank_metrics(rank, [5, 10, 50], ["HR", "NDCG"])
    for k, v in got.items():
        assert abs(v - float(g["res/" + k])) < 1e-9, k


@pytest.mark.parametrize("d", [32, 64, 128])
def test_full_catalogue_rank_vs_oracle(d, cuda, eng):
    rng = np.random.default_rng(d)
    for n_items, N, n_users in ((130, 5, 4), (1000, 70, 30), (5000, 129, 60)):
        I = rng.normal(0, 0.3, size=(n_items, d)).astype(np.float32)
        Uv = rng.normal(0, 0.3, size=(N, d)).astype(np.float32)
        users = rng.integers(0, n_users, size=N)
        targets = rng.integers(1, n_items, size=N)
        sets, ptr, items = make_clicked(n_users, n_items, rng)
        for r in range(N):
            sets[users[r]].add(int(targets[r]))  # dev/test targets are in the residual clicked set
        ptr = np.zeros(n_users + 1, dtype=np.int64)
        flat = []
        for u in range(n_users):
            flat.extend(sorted(sets[u]))
            ptr[u + 1] = len(flat)
        flat = np.array(flat, dtype=np.int64)
        rank, tscore = eng.full_catalogue_rank(dev(Uv, cuda), dev(I, cuda), dev(users, cuda), dev(targets, cuda),
                                               dev(ptr, cuda), dev(flat, cuda))
        want = S.full_catalogue_rank(Uv, I, users, targets, sets)
        s64 = Uv.astype(np.float64) @ I.astype(np.float64).T
        t64 = s64[np.arange(N), targets]
        assert np.allclose(tscore.cpu().numpy(), t64, rtol=1e-5, atol=1e-5)
        near = (np.abs(s64 - t64[:, None]) <= 1e-5 * (1 + np.abs(t64[:, None]))).sum(axis=1) - 1  # fp32 vs fp64 near-ties
        diff = np.abs(rank.cpu().numpy().astype(np.int64) - want)
        assert (diff <= near).all(), (diff.max(), near.max())
        assert (diff == 0).mean() > 0.95
        # no masking at all
        rank0, _ = eng.full_catalogue_rank(dev(Uv, cuda), dev(I, cuda), None, dev(targets, cuda))
        want0 = 1 + ((s64 >= t64[:, None])[:, 1:].sum(axis=1) - 1)
        assert (np.abs(rank0.cpu().numpy() - want0) <= near).all()


def test_full_catalogue_scores_are_consistent_between_mfma_and_scalar_chain(cuda, eng):
    """items that are exact copies of the target row must tie with it (score >= target) in the MFMA
    kernel: the scalar chain used for target / clicked scores restates the MFMA summation order"""
    rng = np.random.default_rng(9)
    d, n_items, N = 64, 4000, 256
    I = rng.normal(0, 0.5, size=(n_items, d)).astype(np.float32)
    Uv = rng.normal(0, 0.5, size=(N, d)).astype(np.float32)
    targets = rng.integers(1, n_items // 2, size=N)
    copies = n_items // 2 + np.arange(N) * 3
    I[copies] = I[targets]
    I[copies + 1] = I[targets]
    rank, _ = eng.full_catalogue_rank(dev(Uv, cuda), dev(I, cuda), None, dev(targets, cuda))
    s64 = Uv.astype(np.float64) @ I.astype(np.float64).T
    t64 = s64[np.arange(N), targets]
    strictly = (s64[:, 1:] > t64[:, None] + 1e-4 * (1 + np.abs(t64[:, None]))).sum(axis=1)
    tied = (np.abs(s64[:, 1:] - t64[:, None]) <= 1e-4 * (1 + np.abs(t64[:, None]))).sum(axis=1) - 1  # minus the target
    got = rank.cpu().numpy()
    # every exact copy ties: at least 1 + strictly + (copies among tied); copies of targets shared by rows count too
    n_copies = np.array([(I[1:] == I[t]).all(axis=1).sum() - 1 for t in targets])
    assert (got >= 1 + strictly + n_copies).all()
    assert (got <= 1 + strictly + tied).all()
