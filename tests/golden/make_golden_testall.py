"""Golden vectors for --test_all (full-catalogue) evaluation FROM THE REFERENCE: its BaseReader on a small synthetic
dataset, its BPRMF with test_all = 1, its BaseRunner.predict (scores over ALL items, clicked items set to -inf,
helpers/BaseRunner.py:225-252) and the rank rule of evaluate_method (:63).  Build container only:

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_testall.py

The seed is advanced until no competing score lies within 1e-4 (relative) of a target's score, so that a
re-implementation with a different fp32 summation order must reproduce the ranks EXACTLY (no near-tie allowance)."""
import os
import sys
import tempfile
from types import SimpleNamespace

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_golden import HERE, _import_reference, _runner_args  # noqa: E402
sys.path.insert(0, os.path.dirname(HERE))
import pandas as pd  # noqa: E402


def make_dataset(root, name, n_users, n_items, per_user, n_neg, seed):
    """train / dev / test csv files in the reference's format (data/README.md:9-60); negatives among the seen item ids"""
    rng = np.random.default_rng(seed)
    rows = {"train": [], "dev": [], "test": []}
    for u in range(1, n_users + 1):
        items = rng.choice(np.arange(1, n_items), size=per_user, replace=False)
        times = np.sort(rng.integers(1_000_000, 2_000_000, size=per_user))
        for k, (i, t) in enumerate(zip(items, times)):
            rows["test" if k == per_user - 1 else ("dev" if k == per_user - 2 else "train")].append((u, int(i), int(t)))
    top = max(i for r in rows.values() for _, i, _ in r)
    clicked = {}
    for r in rows.values():
        for u, i, _ in r:
            clicked.setdefault(u, set()).add(i)
    d = os.path.join(root, name)
    os.makedirs(d, exist_ok=True)
    for phase, r in rows.items():
        df = pd.DataFrame(r, columns=["user_id", "item_id", "time"])
        if phase != "train":
            df["neg_items"] = [rng.choice(np.setdiff1d(np.arange(1, top + 1), np.fromiter(clicked[u], dtype=int)), size=n_neg,
                                          replace=False).tolist() for u in df["user_id"]]
        df.to_csv(os.path.join(d, phase + ".csv"), sep="\t", index=False)


def make_case(name, n_users, n_items, d, seed):
    torch, BPRMF, BaseRunner = _import_reference()
    from helpers.BaseReader import BaseReader
    for attempt in range(50):
        torch.manual_seed(seed + attempt)
        with tempfile.TemporaryDirectory() as root:
            make_dataset(root, "ta", n_users=n_users, n_items=n_items, per_user=10, n_neg=5, seed=seed + attempt)
            corpus = BaseReader(SimpleNamespace(path=root + "/", dataset="ta", sep="\t"))
        args = SimpleNamespace(device=torch.device("cpu"), model_path="", buffer=0, num_neg=1, dropout=0, test_all=1, emb_size=d)
        model = BPRMF(args, corpus)
        with torch.no_grad():
            for p in model.parameters():
                p.mul_(30.0)   # std 0.3: scores of order 1
        ds = BPRMF.Dataset(model, corpus, "test")
        ds.prepare()
        rargs = _runner_args(BaseRunner, "Adam", 1e-3, 0.0)
        rargs.num_workers, rargs.eval_batch_size = 0, 64
        runner = BaseRunner(rargs)
        pred = runner.predict(ds)                     # [N, n_items]: column 0 = target, column j = item j, clicked = -inf
        gt_rank = (pred >= pred[:, 0].reshape(-1, 1)).sum(axis=-1)
        finite = np.isfinite(pred[:, 1:])
        gap = np.abs(pred[:, 1:] - pred[:, :1]) / np.maximum(np.abs(pred[:, :1]), 1e-3)
        if gap[finite].min() > 1e-4:
            break
    else:
        raise SystemExit("no tie-free seed found")
    users = np.asarray(ds.data["user_id"], dtype=np.int64)
    targets = np.asarray(ds.data["item_id"], dtype=np.int64)
    ptr, flat = [0], []
    for u in range(corpus.n_users):
        items = sorted(corpus.train_clicked_set.get(u, set()) | corpus.residual_clicked_set.get(u, set()))
        flat.extend(items)
        ptr.append(len(flat))
    res = runner.evaluate_method(pred, [5, 10, 50], ["HR", "NDCG"])
    out = {"U": model.u_embeddings.weight.detach().numpy().copy(), "I": model.i_embeddings.weight.detach().numpy().copy(),
           "users": users, "targets": targets, "clicked_ptr": np.array(ptr, dtype=np.int64), "clicked_items": np.array(flat, dtype=np.int64),
           "gt_rank": gt_rank.astype(np.int64), "target_score": pred[:, 0].astype(np.float32), "min_rel_gap": np.float64(gap[finite].min())}
    for k, v in res.items():
        out["res/" + k] = np.float64(v)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) >> 10, "KiB; N =", len(users), "min gap", gap[finite].min(), "seed +", attempt)


if __name__ == "__main__":
    make_case("testall_bprmf_d64", 50, 400, 64, 7)
    make_case("testall_bprmf_d32", 30, 1500, 32, 8)
