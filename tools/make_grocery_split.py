"""Real-data anchor (BASELINE configs[2] names Grocery_and_Gourmet_Food): rebuild a dev / test split for the bundled
train.csv of the reference with the recipe of its own preprocessing notebook
(data/Grocery_and_Gourmet_Food/Amazon.ipynb cells 12, 15-16: leave-one-out per user after keeping each user's first
interaction for training, 99 negatives outside the user's clicked set, np.random.seed(0)).  The upstream dev.csv /
test.csv are missing from the reference checkout (.MISSING_LARGE_BLOBS), so the recipe is applied to the bundled
training split itself: metrics are comparable in kind, not identical in value, to docs/demo_scripts_results/README.md.

Build container only (reads /root/reference, writes data_local/, which is git-ignored: the data are not ours to commit):

    python tools/make_grocery_split.py
"""
import os
import sys

import numpy as np
import pandas as pd

SRC = "/root/reference/data/Grocery_and_Gourmet_Food/train.csv"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "data_local", "Grocery_and_Gourmet_Food")
NEG_ITEMS = 99


def main():
    if not os.path.exists(SRC):
        raise SystemExit("reference data not found at " + SRC)
    out_df = pd.read_csv(SRC, sep="\t").reset_index(drop=True)  # already sorted by (time, user_id) upstream
    np.random.seed(0)
    clicked_item_set = {u: set(s["item_id"].values.tolist()) for u, s in out_df.groupby("user_id")}

    def generate_dev_test(data_df):
        result_dfs = []
        n_items = data_df["item_id"].value_counts().size
        for _ in range(2):
            result_df = data_df.groupby("user_id").tail(1).copy()
            data_df = data_df.drop(result_df.index)
            neg_items = np.random.randint(1, n_items + 1, (len(result_df), NEG_ITEMS))
            for i, uid in enumerate(result_df["user_id"].values):
                user_clicked = clicked_item_set[uid]
                for j in range(len(neg_items[i])):
                    while neg_items[i][j] in user_clicked:
                        neg_items[i][j] = np.random.randint(1, n_items + 1)
            result_df["neg_items"] = neg_items.tolist()
            result_dfs.append(result_df)
        return result_dfs, data_df

    leave_df = out_df.groupby("user_id").head(1)
    data_df = out_df.drop(leave_df.index)
    [test_df, dev_df], data_df = generate_dev_test(data_df)
    train_df = pd.concat([leave_df, data_df]).sort_index()
    os.makedirs(OUT, exist_ok=True)
    train_df.to_csv(os.path.join(OUT, "train.csv"), sep="\t", index=False)
    dev_df.to_csv(os.path.join(OUT, "dev.csv"), sep="\t", index=False)
    test_df.to_csv(os.path.join(OUT, "test.csv"), sep="\t", index=False)
    print(len(train_df), len(dev_df), len(test_df), "rows (train / dev / test) ->", OUT)


if __name__ == "__main__":
    sys.exit(main())
