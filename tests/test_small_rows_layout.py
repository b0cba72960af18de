"""CPU: the index algebra of small_row_sums_kernel's long-row path (csrc/small_step.hip) restated in Python for every row width it
is instantiated for: a row's occurrences are taken in chunks of 192 (three occurrence numbers per lane, one coalesced round), and
inside a chunk in rounds of U * GPW = 32 occurrences handed out by shuffles from ONE of the three registers.  Every occurrence
must be summed exactly once, and a round must never need two registers (it would read the wrong one).  Reference semantics:
aten::embedding_dense_backward (the index_add of the nn.Embedding tables, models/context/FM.py:33-41)."""
from collections import Counter

import pytest


@pytest.mark.parametrize("d", [16, 32, 64, 128])
def test_long_rows_are_covered_exactly_once(d):
    lpr = d // 4
    gpw = 64 // lpr
    u_per = 32 // gpw
    assert u_per >= 1 and u_per * gpw == 32
    for hn in (33, 63, 64, 65, 127, 128, 129, 146, 191, 192, 193, 384, 385, 1000, 6000):
        seen = Counter()
        for c0 in range(0, hn, 192):
            cn = min(192, hn - c0)
            for base in range(0, cn, u_per * gpw):
                k64 = base >> 6
                for u in range(u_per):
                    for grp in range(gpw):
                        idx = base + u * gpw + grp
                        assert idx >> 6 == k64 or idx >= cn + 64, (d, hn, base, idx)   # the whole round reads register k64
                        if idx < cn:
                            assert idx >> 6 == k64
                            assert idx & 63 < 64 and 64 * k64 + (idx & 63) == idx      # lane (idx & 63) of register k64 holds occurrence idx
                            seen[c0 + idx] += 1
        assert sorted(seen) == list(range(hn)) and set(seen.values()) == {1}, (d, hn)


def test_short_rows_walk_in_batches_of_eight_in_ascending_order():
    for n in range(1, 33):
        order = []
        for k0 in range(0, n, 8):
            order += [k0 + j for j in range(8) if k0 + j < n]
        assert order == list(range(n))
