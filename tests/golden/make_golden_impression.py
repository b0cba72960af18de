"""Golden vectors for the impression path FROM THE REFERENCE, build container only:
    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_impression.py
 * ImpressionModel.loss (models/BaseImpressionModel.py:44-129) for every loss name: loss value and autograd
   gradient on random lists with ragged positive / negative counts;
 * ImpressionRunner.evaluate_method (helpers/ImpressionRunner.py:73-135): per-row NDCG / MAP / HR;
 * ImpressionReader (helpers/ImpressionReader.py) run on tests/synth_data.make_impression_dataset: the
   impressions it keeps, with their positive / negative item sets."""
import json
import os
import sys
import tempfile
from types import SimpleNamespace

import numpy as np

HERE_DIR = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE_DIR)
sys.path.insert(0, os.path.dirname(HERE_DIR))
from make_golden import HERE, _import_reference  # noqa: E402
from synth_data import make_impression_dataset  # noqa: E402

LOSSES = ["BPR", "BPRhard", "BPRafter", "BPRbefore", "BPRhardafter", "listnet", "softmaxCE", "attention_rank"]


def lists(rng, B, max_pos, max_neg, need_neg):
    n = max_pos + max_neg
    pred = rng.normal(0, 1.5, size=(B, n)).astype(np.float32)
    target = np.full((B, n), -1, dtype=np.int64)
    for b in range(B):
        n_pos = rng.integers(1, max_pos + 1)
        n_neg = rng.integers(1, max_neg + 1) if (need_neg or b % 6) else 0
        target[b, :n_pos] = 1
        target[b, max_pos:max_pos + n_neg] = 0
    return pred, target


def main():
    torch, _, _ = _import_reference()
    from models.BaseImpressionModel import ImpressionModel
    from helpers.ImpressionReader import ImpressionReader
    from helpers.ImpressionRunner import ImpressionRunner
    out = {}
    rng = np.random.default_rng(77)
    for shape_id, (B, mp, mn) in enumerate(((48, 20, 20), (31, 3, 10), (9, 1, 70))):
        for name in LOSSES:
            pred, target = lists(rng, B, mp, mn, need_neg="BPR" in name)
            stub = SimpleNamespace(loss_n=name, train_max_pos_item=mp, device=torch.device("cpu"))
            p = torch.from_numpy(pred).requires_grad_(True)
            loss = ImpressionModel.loss(stub, {"prediction": p}, torch.from_numpy(target))
            loss.backward()
            key = "loss/{}/{}/".format(shape_id, name)
            out[key + "pred"], out[key + "target"], out[key + "max_pos"] = pred, target, np.int64(mp)
            out[key + "loss"], out[key + "gpred"] = np.float32(loss.item()), p.grad.numpy().copy()
    # metrics
    for case, (N, mp, mn) in enumerate(((60, 20, 20), (25, 5, 5))):
        pos_num = rng.integers(1, mp + 4, size=N)
        neg_num = rng.integers(1, mn + 4, size=N)
        pred = rng.normal(size=(N, mp + mn)).astype(np.float32)
        pred[::4, mp] = pred[::4, 0]  # ties between a positive and a negative
        col = np.arange(mp + mn)[None, :]
        keep = (col < np.minimum(pos_num, mp)[:, None]) | ((col >= mp) & (col < mp + np.minimum(neg_num, mn)[:, None]))
        masked = np.where(keep, pred, -np.inf)
        res = ImpressionRunner.evaluate_method(masked, [1, 2, 3, 5, 10], ["NDCG", "HR"], False, list(neg_num), mp,
                                               list(pos_num), ret_all=1)
        key = "metric/{}/".format(case)
        out[key + "pred"], out[key + "pos_num"], out[key + "neg_num"], out[key + "max_pos"] = masked, pos_num, neg_num, np.int64(mp)
        for k, v in res.items():
            out[key + "res/" + k] = np.asarray(v, dtype=np.float64)
    np.savez_compressed(os.path.join(HERE, "impression_losses_metrics.npz"), **out)
    # reader
    root = tempfile.mkdtemp(prefix="rc_imp_")
    make_impression_dataset(root, "synth_imp")
    args = SimpleNamespace(path=root + "/", dataset="synth_imp", sep="\t", impression_idkey="time")
    corpus = ImpressionReader(args)
    dump = {"n_users": int(corpus.n_users), "n_items": int(corpus.n_items)}
    for phase in ("train", "dev", "test"):
        df = corpus.data_df[phase]
        dump[phase] = [[int(u), int(t), sorted(int(x) for x in p), sorted(int(x) for x in n), int(pn), int(nn)]
                       for u, t, p, n, pn, nn in zip(df["user_id"], df["time"], df["pos_items"], df["neg_items"],
                                                     df["pos_num"], df["neg_num"])]
    # history-aware reader: positions and the (time, item)-sorted histories (within one impression the
    # reference's order is Python-set iteration order, which is not part of the contract)
    from helpers.ImpressionSeqReader import ImpressionSeqReader
    seq = ImpressionSeqReader(args)
    dump["seq"] = {phase: [[int(u), int(t), int(p), int(q)] for u, t, p, q in
                           zip(seq.data_df[phase]["user_id"], seq.data_df[phase]["time"], seq.data_df[phase]["position"],
                               seq.data_df[phase]["neg_position"])] for phase in ("train", "dev", "test")}
    dump["seq_his"] = {str(int(u)): {k: sorted([int(t), int(i)] for i, t in v[k]) for k in ("pos", "neg")}
                       for u, v in seq.user_his.items()}
    json.dump(dump, open(os.path.join(HERE, "impression_reader.json"), "w"))
    print("wrote impression goldens:", len(out), "arrays;", {p: len(dump[p]) for p in ("train", "dev", "test")})


if __name__ == "__main__":
    main()
