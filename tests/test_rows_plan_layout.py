"""CPU: the counting sort behind rc_rows_plan_build (csrc/seg_update.hip, section 5) restated step by step in numpy -- tile histograms,
prefix over the tiles, per-wave shares of a row inside a tile, ranks among the lanes of a round that hold the same row -- must hand
every live occurrence the position the stable sort gives it.  Pins the algorithm where no GPU is present; the kernels themselves are
compared with the radix-sorted route bit for bit in tests/test_gpu_sasrec.py::test_rows_plan_equals_the_sorted_rows_route."""
import numpy as np
import pytest

TILE, WAVES, LANES = 4096, 4, 64
ROUNDS = TILE // (WAVES * LANES)


def rows_plan_positions(ids_a, ids_b, lengths, L, n_rows):
    """-> (perm [n_live], start [n_rows], end [n_rows]) exactly as the three kernels form them"""
    n_a, n_b = len(ids_a), len(ids_b)
    n_occ = n_a + n_b
    key = np.full(n_occ, -1, np.int64)
    pad = np.zeros(n_occ, bool)
    key[:n_a] = ids_a
    for j in range(n_b):
        b, l = divmod(j, L) if lengths is not None else (0, 0)
        if lengths is not None and l >= lengths[b]:
            pad[n_a + j] = True
        else:
            key[n_a + j] = ids_b[j]
    key[(key < 0) | (key >= n_rows)] = -1
    n_tiles = -(-n_occ // TILE)
    # count: one histogram per tile; the tile's first padding slot
    matrix = np.zeros((n_tiles, n_rows), np.int64)
    tile_pad = np.full(n_tiles, -1, np.int64)
    for t in range(n_tiles):
        sl = slice(t * TILE, min((t + 1) * TILE, n_occ))
        k = key[sl]
        np.add.at(matrix[t], k[k >= 0], 1)
        p = np.nonzero(pad[sl])[0]
        if len(p):
            tile_pad[t] = t * TILE + p[0]
    # prefix: exclusive over the tiles, per row
    totals = matrix.sum(axis=0)
    offs = np.cumsum(matrix, axis=0) - matrix
    first_pad = next((int(x) for x in tile_pad if x >= 0), -1)
    has_pad = 1 if first_pad >= 0 else 0
    cnt = totals.copy()
    cnt[0] += has_pad
    start = np.cumsum(cnt) - cnt
    end = start + cnt
    perm = np.full(int(cnt.sum()), -1, np.int64)
    if has_pad:
        perm[totals[0]] = first_pad          # row 0 starts at 0: its list ends with the padding slot
    # scatter: per tile, every wave owns a quarter; its share of a row starts after the shares of the waves before it
    for t in range(n_tiles):
        base = start + offs[t]
        wcnt = np.zeros((WAVES, n_rows), np.int64)
        for w in range(WAVES):
            o0 = t * TILE + w * (TILE // WAVES)
            k = key[o0:min(o0 + TILE // WAVES, n_occ)] if o0 < n_occ else key[:0]
            np.add.at(wcnt[w], k[k >= 0], 1)
        wrel = np.cumsum(wcnt, axis=0) - wcnt
        for w in range(WAVES):
            for r in range(ROUNDS):
                o0 = t * TILE + w * (TILE // WAVES) + r * LANES
                lanes = [o for o in range(o0, min(o0 + LANES, n_occ)) if key[o] >= 0]
                seen = {}
                for o in lanes:              # rank among the lanes of the round that hold the same row: lower lanes first
                    k = key[o]
                    rank = seen.get(k, 0)
                    seen[k] = rank + 1
                    perm[base[k] + wrel[w][k] + rank] = o
                for k, c in seen.items():    # the leader advances the wave's offset by the group's size
                    wrel[w][k] += c
    return perm, start, end


@pytest.mark.parametrize("B,C,L,n_rows,seed", [(70, 5, 12, 30, 0), (300, 20, 7, 97, 1), (1100, 9, 0, 500, 2), (64, 100, 50, 40, 3),
                                               (2, 3, 4, 8, 4)])
def test_counting_sort_hands_out_the_stable_sorts_positions(B, C, L, n_rows, seed):
    rng = np.random.default_rng(seed)
    ids_a = rng.integers(1, n_rows, size=B * C)
    ids_a[: B * C // 3] = min(7, n_rows - 1)                       # a hot row
    lengths = rng.integers(0, L + 1, size=B) if L else None
    hist = rng.integers(1, n_rows, size=(B, max(L, 1)))[:, :L]
    if L:
        hist = hist * (np.arange(L)[None, :] < lengths[:, None])
    ids_b = hist.reshape(-1)
    perm, start, end = rows_plan_positions(ids_a, ids_b, lengths, max(L, 1), n_rows)
    n_a = B * C
    pad = (np.arange(L)[None, :] >= lengths[:, None]).reshape(-1) if L else np.zeros(0, bool)
    occ_key = np.concatenate([ids_a, ids_b])
    live = np.concatenate([np.ones(n_a, bool), ~pad])
    order = np.argsort(occ_key[live], kind="stable")
    want = np.nonzero(live)[0][order]
    keys = occ_key[live][order]
    if pad.any():
        first = n_a + int(np.argmax(pad))
        n0 = int((keys == 0).sum())
        want = np.concatenate([want[:n0], [first], want[n0:]])
        keys = np.concatenate([keys[:n0], [0], keys[n0:]])
    assert np.array_equal(perm, want)
    cnt = np.bincount(keys, minlength=n_rows)
    assert np.array_equal(end - start, cnt) and np.array_equal(start, np.cumsum(cnt) - cnt)
