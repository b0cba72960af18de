"""CPU: the plan restatement (oracle/plan_oracle.py) is the index_add of embedding_dense_backward with a fixed
order: summing gradient rows along its groups reproduces np.add.at over the id list (the semantics the reference
gets from loss.backward(), helpers/BaseRunner.py:205), and the singleton flags partition the occurrences."""
import numpy as np

from oracle import plan_oracle as PO


def test_groups_reproduce_index_add():
    rng = np.random.default_rng(0)
    ids_a = rng.integers(0, 40, size=500)
    ids_b = rng.integers(0, 7, size=60)
    grad = rng.normal(size=(560, 8))
    ga, gb, single = PO.bucket_plan(ids_a, ids_b, list_single_a=True)
    assert single is None
    for ids, groups, n_rows, base in ((ids_a, ga, 40, 0), (ids_b, gb, 7, 500)):
        want = np.zeros((n_rows, 8))
        np.add.at(want, ids, grad[base:base + len(ids)])
        got = np.zeros((n_rows, 8))
        for row, pos in groups.items():
            assert np.all(np.diff(pos) > 0), "positions of a row must ascend (fixed summation order)"
            for p in pos:
                got[row] += grad[p]
        assert np.allclose(got, want, rtol=0, atol=1e-12)


def test_singleton_flags_partition_the_occurrences():
    ids = np.array([3, 9, 3, 4, 4, 4, 8])
    ga, gb, single = PO.bucket_plan(ids, None, list_single_a=False)
    assert gb == {}
    assert sorted(ga) == [3, 4]
    assert np.array_equal(single, [0, 1, 0, 0, 0, 0, 1])
    assert sum(len(p) for p in ga.values()) + int(single.sum()) == len(ids)
