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


def make_context_dataset(root, name="synth_ctx", n_users=80, n_items=120, per_user=16, ctr=True, n_neg=20, seed=0, numeric=False):
    """Context-aware data in the reference's layout (data/README.md, MIND_Large/MINDCTR):
    interactions carry situation columns c_*_c; item_meta.csv / user_meta.csv carry i_*_c / u_*_c.
    numeric=True adds the features FM.py:38-41 treats as numbers: c_day_f (whole days, an int64 column like MIND's,
    data/MIND_Large/MIND-large.ipynb cell 10) among the situation columns and i_age_f (a float64 column) in item_meta.csv.
    ctr=True: every row has a binary `label` (clicks depend on user-group x item-category affinity);
    ctr=False: top-k format (positives only, neg_items lists in dev/test)."""
    rng = np.random.default_rng(seed)
    n_cat, n_age = 6, 5
    item_cat = rng.integers(0, n_cat, size=n_items + 1)
    user_age = rng.integers(0, n_age, size=n_users + 1)
    user_gender = rng.integers(0, 2, size=n_users + 1)
    affinity = rng.normal(0, 1.5, size=(n_age, n_cat))
    rows = {"train": [], "dev": [], "test": []}
    clicked = {}
    for u in range(1, n_users + 1):
        times = np.sort(rng.integers(1_000_000, 2_000_000, size=per_user))
        for k, t in enumerate(times):
            i = int(rng.integers(1, n_items + 1))
            hour, weekday = int(t // 3600 % 24), int(t // 86400 % 7)
            logit = affinity[user_age[u], item_cat[i]] + 0.3 * (hour > 12)
            label = int(rng.random() < 1.0 / (1.0 + np.exp(-logit)))
            if not ctr:
                # positives only: resample the item towards the user's preferred categories
                best = np.argsort(-affinity[user_age[u]])[:2]
                cand = np.nonzero(np.isin(item_cat[1:], best))[0] + 1
                i = int(rng.choice(cand))
            phase = "test" if k >= per_user - 2 else ("dev" if k >= per_user - 4 else "train")
            rows[phase].append((u, i, int(t), label, hour, weekday))
            clicked.setdefault(u, set()).add(i)
    d = os.path.join(root, name)
    os.makedirs(d, exist_ok=True)
    for phase, r in rows.items():
        df = pd.DataFrame(r, columns=["user_id", "item_id", "time", "label", "c_hour_c", "c_weekday_c"])
        if numeric:
            df["c_day_f"] = (df["time"] - 1_000_000) // 86400
        if not ctr:
            df = df.drop(columns=["label"])
            if phase != "train":
                negs = []
                for u in df["user_id"]:
                    cand = np.setdiff1d(np.arange(1, n_items + 1), np.fromiter(clicked[u], dtype=int))
                    negs.append(rng.choice(cand, size=min(n_neg, len(cand)), replace=False).tolist())
                df["neg_items"] = negs
        df.to_csv(os.path.join(d, phase + ".csv"), sep="\t", index=False)
    item_meta = {"item_id": np.arange(1, n_items + 1), "i_category_c": item_cat[1:]}
    if numeric:
        item_meta["i_age_f"] = np.round(np.random.default_rng(seed + 1).random(n_items) * 3.0, 3)
    pd.DataFrame(item_meta).to_csv(os.path.join(d, "item_meta.csv"), sep="\t", index=False)
    pd.DataFrame({"user_id": np.arange(1, n_users + 1), "u_age_c": user_age[1:], "u_gender_c": user_gender[1:]}).to_csv(
        os.path.join(d, "user_meta.csv"), sep="\t", index=False)
    return d


def make_impression_dataset(root, name="synth_imp", n_users=40, n_items=90, n_imp=9, seed=0):
    """Impression data in the reference's layout (data/README.md "Impression-based"; ML_1MCTR / MINDCTR):
    every row is (user_id, item_id, time, label); rows of one request share (user_id, time).  Includes the
    corner cases the reader has to handle: impressions with only negatives or only positives (dropped),
    repeated items inside an impression, a single-row impression."""
    rng = np.random.default_rng(seed)
    taste = rng.normal(size=(n_users + 1, 4))
    feat = rng.normal(size=(n_items + 1, 4))
    rows = {"train": [], "dev": [], "test": []}
    for u in range(1, n_users + 1):
        times = np.sort(rng.choice(np.arange(1000, 100000), size=n_imp, replace=False))
        for k, t in enumerate(times):
            phase = "test" if k == n_imp - 1 else ("dev" if k == n_imp - 2 else "train")
            size = int(rng.integers(1, 9))
            items = rng.integers(1, n_items + 1, size=size)          # repeats possible
            logits = feat[items] @ taste[u]
            labels = (rng.random(size) < 1 / (1 + np.exp(-logits))).astype(int)
            kind = (u * 7 + k) % 11
            if kind == 0:
                labels[:] = 0                                         # only negatives
            elif kind == 1:
                labels[:] = 1                                         # only positives
            for i, l in zip(items, labels):
                rows[phase].append((u, int(i), int(t), int(l)))
    d = os.path.join(root, name)
    os.makedirs(d, exist_ok=True)
    for phase, r in rows.items():
        df = pd.DataFrame(r, columns=["user_id", "item_id", "time", "label"])
        df = df.sample(frac=1.0, random_state=seed).reset_index(drop=True)  # the reader must sort
        df.to_csv(os.path.join(d, phase + ".csv"), sep="\t", index=False)
    return d
