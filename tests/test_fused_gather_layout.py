"""CPU: index algebra of the round-6 DeepFM small-batch launches, restated in numpy / pure Python --
  * the plan workgroups that ride in the gather's launch enumerate the composite (field, id) keys field-major, a chunk of 64 rows of
    one field at a time (csrc/fm_bce.hip FieldPlanKey): every (row, field) occurrence exactly once, recorded position r F + f, the
    positions of one key ascending with the scan;
  * the divisions by a runtime constant in those kernels are one v_mul_hi_u32 (csrc/small_plan.hpp small_div_magic / small_div,
    FmTap.magic_F): exact over the ranges the kernels use them on;
  * the protocol of the device counters that ride in other launches (rechorus_amd/engine.py: defer_increment / pending_deferred /
    fold_deferred / take_deferred and the early-bump registry of step_increment), with the library calls recorded instead of made.
Reference being replaced: aten::embedding_dense_backward's sort behind loss.backward() (helpers/BaseRunner.py:205) for the tables of
models/context/FM.py:33-57, and the per-forward nn.Dropout of utils/layers.py:201-243."""
import numpy as np
import pytest
import torch


def magic(d):
    return 0 if d <= 1 else ((1 << 32) + d - 1) // d


def mulhi_div(n, m):
    return n if m == 0 else (n * m) >> 32


def test_mulhi_division_is_exact_where_the_kernels_use_it():
    # r / C with r < n <= 32,768 rows, C <= n; o / F with o < 32,768 occurrences, F <= 48 fields; s / n in the first cut (s < 32,768)
    for d in list(range(1, 130)) + [255, 256, 1000, 4096, 8191, 8192, 32767, 32768]:
        m = magic(d)
        assert m < (1 << 32)
        n = np.arange(0, 32768, dtype=np.uint64)
        got = n if m == 0 else (n * np.uint64(m)) >> np.uint64(32)
        assert np.array_equal(got, n // np.uint64(d)), d
    # the bound behind it: exact while n * d < 2^32 (e = m d - 2^32 < d, so n e < 2^32)
    for d in (3, 7, 48, 641, 65535):
        m = magic(d)
        for n in (0, 1, d - 1, d, d + 1, (1 << 32) // d - 1):
            assert mulhi_div(n, m) == n // d, (n, d)


@pytest.mark.parametrize("n_rows,F,C,numeric", [(1024, 8, 1, (2,)), (37 * 5, 6, 5, (1, 4)), (64, 1, 1, ()), (65, 3, 4, ()), (1, 5, 1, (0,))])
def test_field_major_chunks_enumerate_every_occurrence_once(n_rows, F, C, numeric):
    """FieldPlanKey: chunk c -> field f = c / cpf, rows (c - f cpf) * 64 + lane; key = row_offset[f] + ids[f][r or r / C]; pos = r F + f"""
    rng = np.random.default_rng(n_rows + F)
    vocab = [int(rng.integers(2, 50)) for _ in range(F)]
    per_row = [bool(rng.integers(0, 2)) and n_rows % C == 0 for _ in range(F)]
    B = n_rows // C if n_rows % C == 0 else n_rows
    if n_rows % C != 0:
        C, per_row = 1, [False] * F
    ids = [rng.integers(0, vocab[f], size=(B if per_row[f] else n_rows)) for f in range(F)]
    row_offset = np.concatenate([[0], np.cumsum([0 if f in numeric else vocab[f] for f in range(F)])])[:F]
    cpf = (n_rows + 63) // 64
    seen, last_pos = {}, {}
    mC = magic(C)
    for c in range(cpf * F):
        f = c // cpf
        for lane in range(64):
            r = (c - f * cpf) * 64 + lane
            if r >= n_rows or f in numeric:
                continue
            key = int(row_offset[f] + ids[f][mulhi_div(r, mC) if per_row[f] else r])
            pos = r * F + f
            assert pos not in seen
            seen[pos] = key
            assert last_pos.get(key, -1) < pos      # the overflow path emits a key's positions in scan order: they must ascend
            last_pos[key] = pos
    # what the gather writes into cid for the same occurrences (rc_gather_fields_mixed: cid[r, f] = row_offset[f] + id)
    want = {r * F + f: int(row_offset[f] + ids[f][r // C if per_row[f] else r]) for r in range(n_rows) for f in range(F) if f not in numeric}
    assert seen == want


class _Calls:
    def __init__(self):
        self.names = []

    def __call__(self, name, *a):
        self.names.append(name)


def test_counters_that_ride_in_other_launches(monkeypatch):
    from rechorus_amd import _lib, engine
    calls = _Calls()
    monkeypatch.setattr(_lib, "call", calls)
    monkeypatch.setattr(engine, "_stream", lambda: None)
    monkeypatch.setattr(engine, "_ptr", lambda t, *a, **k: t)
    engine.clear_bumped_early()
    adam, seed, other = (torch.zeros(1, dtype=torch.int64) for _ in range(3))
    # Adam's count is promised; nobody has taken it along yet
    engine.defer_increment(adam)
    assert engine.pending_deferred() is adam
    # a launch with a slot (the CTR head's forward) takes it along: the optimizer finds it done and makes no launch of its own
    engine.fold_deferred(adam)
    assert engine.pending_deferred() is None
    assert engine.take_deferred(adam) is True and engine.pending_deferred() is None
    assert calls.names == []
    # without a carrier the next step_increment of ANOTHER counter folds it in (one launch for both) ...
    engine.defer_increment(adam)
    engine.step_increment(seed)
    assert calls.names == ["rc_step_increment2"] and engine.take_deferred(adam) is True
    # ... and with nobody at all the owner increments itself
    engine.defer_increment(adam)
    assert engine.take_deferred(adam) is False
    # the gather's launch bumped the tower's seed early: the tower's own step_increment is a no-op, once
    del calls.names[:]
    engine._bumped_early.append(seed)
    engine.step_increment(seed)
    assert calls.names == []
    engine.step_increment(seed)
    assert calls.names == ["rc_step_increment"]
    # another counter is not confused with it, and a step that raised leaves nothing behind
    engine._bumped_early.append(seed)
    engine.step_increment(other)
    assert calls.names == ["rc_step_increment", "rc_step_increment"]
    engine.clear_bumped_early()
    engine.step_increment(seed)
    assert calls.names[-1] == "rc_step_increment" and len(calls.names) == 3
    # a promise is not folded into an early-bumped owner's no-op call: it waits for a real carrier
    engine.defer_increment(adam)
    engine._bumped_early.append(seed)
    engine.step_increment(seed)
    assert engine.pending_deferred() is adam
    assert engine.take_deferred(adam) is False
