"""CPU: the sampler / batch-assembly / rank oracle (oracle/sampler_oracle.py): Philox known answers,
the device algorithm's restatement vs the reference's own sampling loop (distribution), history
windows and rank metrics vs the reference semantics."""
import numpy as np
import pytest

from oracle import bprmf_oracle as BO
from oracle import sampler_oracle as S


def test_philox_known_answer_vectors():
    """Random123 kat_vectors, philox4x32 with 10 rounds"""
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
            (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, want in kat:
        got = S.philox4x32_10(np.array(ctr, dtype=np.uint32), np.array(key, dtype=np.uint32))
        assert [int(x) for x in got] == list(want)


def make_clicked(n_users, n_items, rng, heavy_user=None):
    sets = {u: set(rng.integers(1, n_items, size=rng.integers(0, 12)).tolist()) for u in range(n_users)}
    if heavy_user is not None:  # a user who clicked all but three items
        sets[heavy_user] = set(range(1, n_items)) - {2, 7, n_items - 1}
    ptr = np.zeros(n_users + 1, dtype=np.int64)
    items = []
    for u in range(n_users):
        items.extend(sorted(sets[u]))
        ptr[u + 1] = len(items)
    return sets, ptr, np.array(items, dtype=np.int64)


def test_sampler_never_returns_clicked_and_is_counter_based():
    rng = np.random.default_rng(0)
    n_users, n_items, K = 30, 50, 7
    sets, ptr, items = make_clicked(n_users, n_items, rng, heavy_user=3)
    users = rng.integers(0, n_users, size=200)
    users[:5] = 3
    neg = S.sample_negatives(users, K, n_items, ptr, items, seed=1234)
    assert neg.shape == (200, K) and neg.min() >= 1 and neg.max() < n_items
    for u, row in zip(users, neg):
        assert not (set(row.tolist()) & sets[u])
    assert set(neg[:5].ravel().tolist()) <= {2, 7, n_items - 1}
    # counter-based: a slice of the call equals the call on the slice with the matching base index
    part = S.sample_negatives(users[50:60], K, n_items, ptr, items, seed=1234, base_index=50 * K)
    assert np.array_equal(part, neg[50:60])
    assert not np.array_equal(S.sample_negatives(users, K, n_items, ptr, items, seed=1235), neg)


def test_sampler_distribution_matches_reference_loop():
    """same distribution as the reference's rejection loop: uniform over the user's non-clicked items"""
    rng = np.random.default_rng(1)
    n_users, n_items, K = 4, 40, 2000
    sets, ptr, items = make_clicked(n_users, n_items, rng)
    users = np.arange(n_users)
    ours = S.sample_negatives(users, K, n_items, ptr, items, seed=7)
    ref = S.reference_sampler(users, K, n_items, sets, np.random.RandomState(0))
    for u in range(n_users):
        allowed = np.array(sorted(set(range(1, n_items)) - sets[u]))
        for name, draws in (("ours", ours[u]), ("reference", ref[u])):
            counts = np.bincount(draws, minlength=n_items)[allowed]
            assert counts.sum() == K, name
            chi2 = ((counts - K / len(allowed)) ** 2 / (K / len(allowed))).sum()
            assert chi2 < 2.0 * len(allowed), (name, u, chi2)  # mean of chi2 = dof, sd = sqrt(2 dof)


def test_history_window_and_rank():
    his = {1: [(5, 10), (6, 20), (7, 30), (8, 40)], 2: [(9, 11)]}
    it, tm, n = S.history_window(1, 3, his, 2)
    assert it.tolist() == [6, 7] and tm.tolist() == [20, 30] and n == 2
    it, tm, n = S.history_window(1, 1, his, 3)
    assert it.tolist() == [5, 0, 0] and n == 1
    pred = np.array([[0.5, 0.5, 0.1, 0.9], [1.0, 0.2, 0.3, 0.4]], dtype=np.float32)
    assert S.target_rank(pred).tolist() == [3, 1]  # ties count against the target
    r = BO.evaluate_method(pred, [1, 3], ["HR", "NDCG"])
    assert r["HR@1"] == 0.5 and r["HR@3"] == 1.0


def test_full_catalogue_rank_masks_clicked_columns():
    rng = np.random.default_rng(2)
    n_items, d = 30, 8
    I = rng.normal(size=(n_items, d)).astype(np.float32)
    Uv = rng.normal(size=(5, d)).astype(np.float32)
    users = np.array([0, 1, 2, 1, 0])
    targets = np.array([3, 4, 5, 6, 7])
    clicked = {0: {3, 7, 9}, 1: {4, 6, 10, 11}, 2: {5}}
    got = S.full_catalogue_rank(Uv, I, users, targets, clicked)
    for r in range(5):
        s = I.astype(np.float64) @ Uv[r].astype(np.float64)
        others = [j for j in range(1, n_items) if j not in clicked[users[r]]]
        assert got[r] == 1 + sum(s[j] >= s[targets[r]] for j in others)


def test_sampler_exhausted_attempts_select_a_non_clicked_item_directly():
    """a user who clicked all but two items of a 3,000-item catalogue: most elements use up their 1,024 rejection
    attempts; the r-th non-clicked id is then selected directly (the reference's loop would keep drawing until it hits
    one, models/BaseModel.py:209-210) -- never a clicked item, both free ids reachable; a user who clicked everything
    keeps a (clicked) draw instead of hanging"""
    from oracle import sampler_oracle as S
    n_items, free = 3001, {17, 2999}
    clicked = np.array([i for i in range(1, n_items) if i not in free], dtype=np.int64)
    everything = np.arange(1, n_items, dtype=np.int64)
    ptr = np.array([0, len(clicked), len(clicked) + len(everything)], dtype=np.int64)
    items = np.concatenate([clicked, everything])
    neg = S.sample_negatives(np.zeros(4, dtype=np.int64), 16, n_items, ptr, items, seed=5)
    assert set(np.unique(neg)) == free
    neg_all = S.sample_negatives(np.ones(1, dtype=np.int64), 2, n_items, ptr, items, seed=5)
    assert neg_all.min() >= 1 and neg_all.max() < n_items


@pytest.mark.parametrize("case", ["testall_bprmf_d64", "testall_bprmf_d32"])
def test_full_catalogue_rank_oracle_vs_the_reference(case):
    """--test_all: the reference's own BaseRunner.predict on its BPRMF (scores over ALL items, clicked items -inf) and the
    rank rule of evaluate_method, on a dataset without near-ties (tests/golden/make_golden_testall.py): exact ranks"""
    import os
    from conftest import GOLDEN_DIR
    from oracle import bprmf_oracle as BO
    from oracle import sampler_oracle as S
    g = np.load(os.path.join(GOLDEN_DIR, case + ".npz"))
    assert float(g["min_rel_gap"]) > 1e-4
    ptr, flat = g["clicked_ptr"], g["clicked_items"]
    sets = [set(flat[ptr[u]:ptr[u + 1]].tolist()) for u in range(len(ptr) - 1)]
    Uv = g["U"][g["users"]]
    rank = S.full_catalogue_rank(Uv, g["I"], g["users"], g["targets"], sets)
    assert np.array_equal(np.asarray(rank, dtype=np.int64), g["gt_rank"])
    for k in (5, 10, 50):
        hit = (g["gt_rank"] <= k)
        assert abs(hit.mean() - float(g[f"res/HR@{k}"])) < 1e-12
        assert abs((hit / np.log2(g["gt_rank"] + 1)).mean() - float(g[f"res/NDCG@{k}"])) < 1e-12
