"""CPU: the tile / chain logic of the narrow-row dense gradient (csrc/seg_update.hip: seg_narrow_tiles_kernel, seg_narrow_chains_kernel)
restated in numpy: sorted positions cut into tiles of 512, a segment inside a tile written at once, the pieces of a segment that
crosses tile borders left in the tile's head / tail slot and added by the tile where it starts.  Every row must receive exactly the
sum of its occurrences, once.  (The kernels against a float64 index_add: tests/test_gpu_bprmf.py::test_narrow_embedding_dense_backward_in_tiles.)"""
import numpy as np
import pytest

TILE = 512


def narrow_tiles(keys, vals, n_rows):
    n = len(keys)
    n_tiles = -(-n // TILE)
    G = np.zeros(n_rows, np.float64)
    written = np.zeros(n_rows, np.int64)
    head = [None] * n_tiles   # (key, flag, value): flag 1 the segment ends in this tile, 2 it goes on
    tail = [None] * n_tiles
    for w in range(n_tiles):
        j0, j1 = w * TILE, min((w + 1) * TILE, n)
        first_key = keys[j0]
        first_cont = j0 > 0 and keys[j0 - 1] == first_key
        run_key, run = None, 0.0
        for j in range(j0, j1):
            if keys[j] != run_key:
                run_key, run = keys[j], 0.0
            run += vals[j]
            last_of_tile = j == j1 - 1
            goes_on = last_of_tile and j + 1 < n and keys[j + 1] == keys[j]
            ends = (not goes_on) and (j + 1 >= n or keys[j + 1] != keys[j])
            from_left = keys[j] == first_key and first_cont
            if ends and not from_left:
                G[keys[j]] = run
                written[keys[j]] += 1
            if (ends or goes_on) and from_left:
                assert head[w] is None
                head[w] = (keys[j], 2 if goes_on else 1, run)
            if goes_on and not from_left:
                assert tail[w] is None
                tail[w] = (keys[j], 2, run)
    for w in range(n_tiles):   # chains: the tile where a border-crossing segment starts adds its pieces
        if tail[w] is None:
            continue
        key, _, total = tail[w]
        x = w + 1
        while x < n_tiles and head[x] is not None and head[x][0] == key:
            total += head[x][2]
            if head[x][1] == 1:
                break
            x += 1
        G[key] = total
        written[key] += 1
    return G, written


@pytest.mark.parametrize("n,n_rows,hot,seed", [(5000, 40, 0.0, 0), (5000, 3, 0.0, 1), (20000, 300, 0.6, 2), (512 * 6, 1, 0.0, 3),
                                               (513, 7, 0.9, 4), (4097, 5000, 0.0, 5), (512 * 4 + 1, 2, 0.5, 6)])
def test_tiles_and_chains_sum_every_row_once(n, n_rows, hot, seed):
    rng = np.random.default_rng(seed)
    ids = rng.integers(0, n_rows, size=n)
    ids[rng.random(n) < hot] = n_rows // 2
    vals = rng.integers(-8, 9, size=n).astype(np.float64)      # integers: sums are exact whatever the order
    order = np.argsort(ids, kind="stable")
    G, written = narrow_tiles(ids[order], vals[order], n_rows)
    want = np.zeros(n_rows)
    np.add.at(want, ids, vals)
    cnt = np.bincount(ids, minlength=n_rows)
    assert np.array_equal(G, want)
    assert np.array_equal(written, (cnt > 0).astype(np.int64))
