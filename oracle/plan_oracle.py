"""TEST INFRASTRUCTURE (checker only): numpy restatement of what rc_bucket_plan (rechorus_amd/csrc/bucket_plan.hip)
must produce -- the grouping of a batch's row-id occurrences that stands in for the index_add of
aten::embedding_dense_backward (reference: loss.backward() at src/helpers/BaseRunner.py:205 on the nn.Embedding
tables of src/models/general/BPRMF.py:31-32).  Pure integer work: the HIP result has to match exactly.

Parity pin: there is no reference fixture for an intermediate grouping (the reference never materialises one);
the pin is semantic -- summing gradient rows over `groups[row]` in the listed order IS index_add with a fixed
order -- and is exercised end to end by the BPRMF goldens (tests/test_gpu_bprmf.py) that run through the plan."""
import numpy as np


def bucket_plan(ids_a, ids_b=None, list_single_a=True):
    """-> (groups_a, groups_b, single_a)
    groups_x: {row id: int64 array of positions, ascending}; positions are p in [0, n_a) for list a and
    n_a + j for list b.  With list_single_a False, rows of list a that occur once are left out of groups_a and
    flagged instead: single_a[p] = 1 at their position (uint8 [n_a]); single_a is None otherwise.
    Positions with a NEGATIVE id take no part (padding slots of a history window): they appear in no group."""
    a = np.asarray(ids_a, dtype=np.int64).reshape(-1)
    b = np.zeros(0, dtype=np.int64) if ids_b is None else np.asarray(ids_b, dtype=np.int64).reshape(-1)

    def groups(ids, base):
        order = np.argsort(ids, kind="stable")
        srt = ids[order]
        cuts = np.flatnonzero(np.concatenate([[True], srt[1:] != srt[:-1]])) if len(srt) else np.zeros(0, np.int64)
        ends = np.concatenate([cuts[1:], [len(srt)]]) if len(srt) else cuts
        return {int(srt[s]): order[s:e] + base for s, e in zip(cuts, ends) if srt[s] >= 0}

    ga, gb = groups(a, 0), groups(b, len(a))
    single = None
    if not list_single_a:
        single = np.zeros(len(a), dtype=np.uint8)
        for row in [r for r, pos in ga.items() if len(pos) == 1]:
            single[ga.pop(row)[0]] = 1
    return ga, gb, single
