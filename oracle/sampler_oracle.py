"""CPU ORACLE -- test infrastructure, not product code (see oracle/bprmf_oracle.py header).

Restates (a) the reference's negative sampler GeneralModel.Dataset.actions_before_epoch
(models/BaseModel.py:206-214: uniform over [1, n_items), redraw while the id is in the user's train
clicked set), (b) SequentialModel.Dataset._get_feed_dict's history window (:236-245) with
collate_batch's right padding (:135-152), and (c) the rank metric of helpers/BaseRunner.py:52-78 /
the --test_all masking of :243-250.

The reference samples with numpy's global MT19937 stream; the device sampler uses counter-based
Philox4x32-10 instead (csrc/sampler.hip), so (a) has two parts: `sample_negatives` restates the
DEVICE algorithm bit-exactly (same Philox, same mulhi mapping, same rejection order) and
`reference_sampler` is the reference's loop itself, used by the tests to compare DISTRIBUTIONS.
Philox is pinned by the known-answer vectors of the Random123 distribution (tests/test_sampler_cpu.py).
"""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = np.uint32(0x9E3779B9), np.uint32(0xBB67AE85)
MASK32 = np.uint64(0xFFFFFFFF)
MAX_ATTEMPTS = 1024


def philox4x32_10(counter, key):
    """counter [..., 4] uint32, key [..., 2] uint32 -> [..., 4] uint32 (Salmon et al., SC'11)"""
    c = [counter[..., i].astype(np.uint64) for i in range(4)]
    k0, k1 = key[..., 0].astype(np.uint32), key[..., 1].astype(np.uint32)
    with np.errstate(over="ignore"):
        for _ in range(10):
            p0, p1 = M0 * c[0], M1 * c[2]
            hi0, lo0 = p0 >> np.uint64(32), p0 & MASK32
            hi1, lo1 = p1 >> np.uint64(32), p1 & MASK32
            c = [hi1 ^ c[1] ^ k0.astype(np.uint64), lo1, hi0 ^ c[3] ^ k1.astype(np.uint64), lo0]
            k0, k1 = (k0 + W0).astype(np.uint32), (k1 + W1).astype(np.uint32)
    return np.stack([x.astype(np.uint32) for x in c], axis=-1)


def _draw(seed, index, block):
    """two 64-bit words per (index, block)"""
    index = np.asarray(index, dtype=np.uint64)
    ctr = np.stack([(index & MASK32).astype(np.uint32), (index >> np.uint64(32)).astype(np.uint32),
                    np.full(index.shape, block, dtype=np.uint32), np.zeros(index.shape, dtype=np.uint32)], axis=-1)
    key = np.broadcast_to(np.array([seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF], dtype=np.uint32), index.shape + (2,))
    r = philox4x32_10(ctr, key).astype(np.uint64)
    return (r[..., 1] << np.uint64(32)) | r[..., 0], (r[..., 3] << np.uint64(32)) | r[..., 2]


def _mulhi64(a, b):
    return np.array([(int(x) * int(b)) >> 64 for x in a.reshape(-1)], dtype=np.int64).reshape(a.shape)


def sample_negatives(users, K, n_items, clicked_ptr=None, clicked_items=None, seed=0, base_index=0):
    """bit-exact restatement of rc_sample_negatives: element e = i*K + k tries attempt 0, 1, 2, ...;
    attempt a uses 64-bit word (a & 1) of Philox block (a >> 1) of the stream (seed, base_index + e)"""
    users = np.asarray(users, dtype=np.int64)
    n = len(users)
    neg = np.zeros(n * K, dtype=np.int64)
    pending = np.arange(n * K)
    for attempt in range(MAX_ATTEMPTS):
        if len(pending) == 0:
            break
        w = _draw(seed, np.uint64(base_index) + pending.astype(np.uint64), attempt >> 1)[attempt & 1]
        cand = 1 + _mulhi64(w, n_items - 1)
        neg[pending] = cand  # (an element that exhausts its attempts keeps the last candidate)
        if clicked_ptr is None:
            break
        u = users[pending // K]
        bad = np.array([_contains(clicked_items, clicked_ptr[uu], clicked_ptr[uu + 1], c) for uu, c in zip(u, cand)],
                       dtype=bool)
        pending = pending[bad]
    else:
        # attempts exhausted (the user clicked nearly everything): the r-th non-clicked id, r from a fresh Philox block
        if clicked_ptr is not None and len(pending):
            w = _draw(seed, np.uint64(base_index) + pending.astype(np.uint64), MAX_ATTEMPTS >> 1)[0]
            for e, we in zip(pending, w):
                uu = users[e // K]
                cl = np.asarray(clicked_items[clicked_ptr[uu]:clicked_ptr[uu + 1]], dtype=np.int64)
                free = (n_items - 1) - len(cl)
                if free <= 0:
                    continue
                rank = int(_mulhi64(np.array([we], dtype=np.uint64), free)[0])
                j = int(np.searchsorted(cl - 1 - np.arange(len(cl)), rank, side="right"))  # smallest j with c_j - 1 - j > rank
                neg[e] = rank + 1 + j
    return neg.reshape(n, K)


def _contains(items, lo, hi, x):
    j = np.searchsorted(items[lo:hi], x)
    return j < hi - lo and items[lo + j] == x


def reference_sampler(users, K, n_items, clicked_sets, rng):
    """the reference's own loop (models/BaseModel.py:206-214) with an explicit numpy RandomState"""
    neg = rng.randint(1, n_items, size=(len(users), K))
    for i, u in enumerate(users):
        clicked = clicked_sets[u]
        for j in range(K):
            while neg[i][j] in clicked:
                neg[i][j] = rng.randint(1, n_items)
    return neg


def history_window(user, pos, user_his, L):
    """(items [L] right-padded with 0, times [L], length) -- models/BaseModel.py:236-245 + :135-152"""
    seq = user_his[user][:pos]
    if L > 0:
        seq = seq[-L:]
    items, times = np.zeros(L, dtype=np.int64), np.zeros(L, dtype=np.int64)
    items[: len(seq)] = [x[0] for x in seq]
    times[: len(seq)] = [x[1] for x in seq]
    return items, times, len(seq)


def target_rank(pred):
    """helpers/BaseRunner.py:62-63"""
    return (pred >= pred[:, 0:1]).sum(axis=-1).astype(np.int32)


def full_catalogue_rank(Uvec, I, users, targets, clicked_sets=None):
    """--test_all (models/BaseModel.py:194-195, helpers/BaseRunner.py:243-250): candidates = [target] +
    every item 1..n_items-1; columns of clicked items (train + residual) are set to -inf; rank of column 0"""
    n_items = I.shape[0]
    out = np.zeros(len(users), dtype=np.int32)
    for r, (u, t) in enumerate(zip(users, targets)):
        cols = np.concatenate([[t], np.arange(1, n_items)])
        pred = (Uvec[r][None, :].astype(np.float64) * I[cols].astype(np.float64)).sum(-1)
        if clicked_sets is not None:
            seen = np.array([c for c in clicked_sets[u] if 1 <= c < n_items], dtype=np.int64)
            pred[seen] = -np.inf  # column index == item id (column 0 is the target itself)
        out[r] = (pred >= pred[0]).sum()
    return out
