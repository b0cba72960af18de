"""Synthetic datasets in the reference's on-disk format (data/README.md of ReChorus):
tab-separated train/dev/test.csv with columns user_id, item_id, time (+ neg_items, a python list
literal, in dev/test), ids starting at 1, leave-one-out split, Zipf item popularity."""
import os

import numpy as np
import pandas as pd


def make_dataset(root, name="synth", n_users=60, n_items=200, per_user=12, n_neg=99, seed=0):
    rng = np.random.default_rng(seed)
    pop = 1.0 / np.arange(1, n_items + 1)
    pop /= pop.sum()
    rows = {"train": [], "dev": [], "test": []}
    for u in range(1, n_users + 1):
        # users have a taste cluster so that there is something to learn
        shift = (u % 7) * (n_items // 7)
        items = (rng.choice(n_items, size=per_user, replace=False, p=pop) + shift) % n_items + 1
        times = np.sort(rng.integers(1_000_000, 2_000_000, size=per_user))
        for k, (i, t) in enumerate(zip(items, times)):
            phase = "test" if k == per_user - 1 else ("dev" if k == per_user - 2 else "train")
            rows[phase].append((u, int(i), int(t)))
    clicked = {}
    for phase in rows:
        for u, i, _ in rows[phase]:
            clicked.setdefault(u, set()).add(i)
    d = os.path.join(root, name)
    os.makedirs(d, exist_ok=True)
    for phase, r in rows.items():
        df = pd.DataFrame(r, columns=["user_id", "item_id", "time"])
        if phase != "train":
            negs = []
            for u in df["user_id"]:
                cand = np.setdiff1d(np.arange(1, n_items + 1), np.fromiter(clicked[u], dtype=int))
                negs.append(rng.choice(cand, size=min(n_neg, len(cand)), replace=False).tolist())
            df["neg_items"] = negs
        df.to_csv(os.path.join(d, phase + ".csv"), sep="\t", index=False)
    return d
