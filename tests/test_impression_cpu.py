"""CPU: the impression path -- list-BPR oracle, the mirror's reader / metrics / torch-op loss variants --
vs outputs of the reference itself (tests/golden/impression_*.{npz,json})."""
import argparse
import json
import os
import sys

import numpy as np
import pytest
import torch

from conftest import GOLDEN_DIR, ROOT, assert_close
from oracle import impression_oracle as IO
from oracle import listwise_oracle as LO
from synth_data import make_impression_dataset

PLUGIN = os.path.join(ROOT, "rechorus_amd", "rechorus")
if PLUGIN not in sys.path:
    sys.path.insert(0, PLUGIN)

G = dict(np.load(os.path.join(GOLDEN_DIR, "impression_losses_metrics.npz")))
LOSS_CASES = sorted({k.rsplit("/", 1)[0] for k in G if k.startswith("loss/")})
METRIC_CASES = sorted({"/".join(k.split("/")[:2]) for k in G if k.startswith("metric/")})


def case(prefix):
    return {k[len(prefix) + 1:]: v for k, v in G.items() if k.startswith(prefix + "/")}


@pytest.mark.parametrize("key", [k for k in LOSS_CASES if k.endswith(("/BPR", "/BPRhard", "/softmaxCE"))])
def test_oracle_list_losses_match_the_reference(key):
    c = case(key)
    name = key.split("/")[-1]
    if name == "softmaxCE":
        loss, _, _ = LO.softmax_ce(c["pred"], c["target"], int(c["max_pos"]))
        g = LO.softmax_ce_grad(c["pred"], c["target"], int(c["max_pos"]))
    else:
        loss, _, g = IO.list_bpr(c["pred"], c["target"], int(c["max_pos"]), hard=(name == "BPRhard"))
    assert_close(loss, c["loss"], what="loss " + key)
    assert_close(g, c["gpred"], what="grad " + key, atol_scale=2e-5)


# every --loss_n the reference's substring rules resolve (models/BaseImpressionModel.py:50-126) is a HIP kernel
# (rc_list_loss_fwd_bwd); parity vs these goldens runs on the GPU: tests/test_gpu_impression.py
KERNEL_LOSSES = sorted({k.split("/")[-1] for k in LOSS_CASES})


def test_kernel_backed_losses_refuse_cpu_tensors():
    from models.BaseImpressionModel import ImpressionModel
    from rechorus_amd import engine
    c = case(LOSS_CASES[0])
    assert len(KERNEL_LOSSES) == 8
    for name in KERNEL_LOSSES:
        assert engine.list_kind(name) is not None, name
        with pytest.raises(RuntimeError):
            ImpressionModel.loss(argparse.Namespace(loss_n=name, train_max_pos_item=int(c["max_pos"])),
                                 {"prediction": torch.from_numpy(c["pred"])}, torch.from_numpy(c["target"]))


def test_bpr_simple_golden_is_the_per_row_pair_sum():
    """'BPR...simple' (reference :82-83: every valid (positive, negative) pair, per-row sums left unreduced) is kernel kind 9;
    its reference-generated golden (tests/golden/impression_bpr_simple.npz) agrees with a direct numpy evaluation"""
    from models.BaseImpressionModel import ImpressionModel
    from rechorus_amd import engine
    assert engine.list_kind("BPRsimple") == engine.LIST_KINDS["BPRsimple"] == engine.list_kind("BPRhardsimple") == 9
    assert engine.list_kind("BPRsimpleafter") == engine.LIST_KINDS["BPRafter"]     # the reference's elif order: 'after' wins
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "impression_bpr_simple.npz"))
    for shape_id in range(3):
        key = "loss/{}/BPRsimple/".format(shape_id)
        pred, tgt, P = g[key + "pred"].astype(np.float64), g[key + "target"], int(g[key + "max_pos"])
        want = np.zeros(pred.shape[0])
        for b in range(pred.shape[0]):
            for i in range(P):
                for j in range(P, pred.shape[1]):
                    if tgt[b, i] != -1 and tgt[b, j] != -1:
                        want[b] += np.log1p(np.exp(-(pred[b, i] - pred[b, j])))
        assert_close(g[key + "rows"], want, what="BPRsimple rows", rtol=2e-5)
    c = case(LOSS_CASES[0])
    P = int(c["max_pos"])
    with pytest.raises(ValueError):
        ImpressionModel.loss(argparse.Namespace(loss_n="nope", train_max_pos_item=P),
                             {"prediction": torch.from_numpy(c["pred"])}, torch.from_numpy(c["target"]))


@pytest.mark.parametrize("key", METRIC_CASES)
def test_list_metrics_match_the_reference(key):
    from helpers.ImpressionRunner import ImpressionRunner
    c = case(key)
    topk = [1, 2, 3, 5, 10]
    want = {k[4:]: v for k, v in c.items() if k.startswith("res/")}
    got = ImpressionRunner.evaluate_method(c["pred"], topk, ["NDCG", "HR"], False, c["neg_num"], int(c["max_pos"]),
                                           c["pos_num"], ret_all=1)
    ora = IO.list_metrics(c["pred"], c["pos_num"], c["neg_num"], int(c["max_pos"]), topk)
    assert list(got) == ["%s@%d" % (m, k) for m in ("NDCG", "MAP", "HR") for k in topk]
    for k, v in want.items():
        assert np.allclose(got[k], v, atol=1e-12), k
        assert np.allclose(ora[k], v, atol=1e-12), k
    mean = ImpressionRunner.evaluate_method(c["pred"], topk, ["NDCG"], False, c["neg_num"], int(c["max_pos"]), c["pos_num"])
    assert abs(mean["NDCG@2"] - want["NDCG@2"].mean()) < 1e-12


def test_impression_reader_matches_the_reference(tmp_path):
    from helpers.ImpressionReader import ImpressionReader
    from models.BaseImpressionModel import ImpressionModel
    want = json.load(open(os.path.join(GOLDEN_DIR, "impression_reader.json")))
    make_impression_dataset(str(tmp_path), "synth_imp")
    corpus = ImpressionReader(argparse.Namespace(path=str(tmp_path) + "/", dataset="synth_imp", sep="\t", impression_idkey="time"))
    assert corpus.n_users == want["n_users"] and corpus.n_items == want["n_items"]
    for phase in ("train", "dev", "test"):
        df = corpus.data_df[phase]
        got = [[int(u), int(t), list(p), list(n), int(pn), int(nn)] for u, t, p, n, pn, nn in
               zip(df["user_id"], df["time"], df["pos_items"], df["neg_items"], df["pos_num"], df["neg_num"])]
        assert got == want[phase], phase
    # dataset: fixed-width lists, positives first
    model = argparse.Namespace(buffer=0, num_neg=1, test_all=0, train_max_pos_item=3, train_max_neg_item=4,
                               test_max_pos_item=3, test_max_neg_item=4)
    ds = ImpressionModel.Dataset(model, corpus, "train")
    batch = ds.collate_batch([ds[i] for i in range(16)])
    assert batch["item_id"].shape == (16, 7) and batch["item_id"].dtype == torch.long
    for r in range(16):
        pn, nn = int(batch["pos_num"][r]), int(batch["neg_num"][r])
        assert (batch["item_id"][r, :pn] > 0).all() and (batch["item_id"][r, pn:3] == 0).all()
        assert (batch["item_id"][r, 3:3 + nn] > 0).all() and (batch["item_id"][r, 3 + nn:] == 0).all()


def test_impression_seq_reader_matches_the_reference(tmp_path):
    from helpers.ImpressionSeqReader import ImpressionSeqReader
    from models.BaseImpressionModel import ImpressionSeqModel
    want = json.load(open(os.path.join(GOLDEN_DIR, "impression_reader.json")))
    make_impression_dataset(str(tmp_path), "synth_imp")
    corpus = ImpressionSeqReader(argparse.Namespace(path=str(tmp_path) + "/", dataset="synth_imp", sep="\t", impression_idkey="time"))
    for phase in ("train", "dev", "test"):
        df = corpus.data_df[phase]
        got = [[int(u), int(t), int(p), int(q)] for u, t, p, q in zip(df["user_id"], df["time"], df["position"], df["neg_position"])]
        assert got == want["seq"][phase], phase
    for u, his in corpus.user_his.items():
        for k in ("pos", "neg"):
            assert sorted([int(t), int(i)] for i, t in his[k]) == want["seq_his"][str(int(u))][k]
    model = argparse.Namespace(buffer=0, num_neg=1, test_all=0, history_max=4, train_max_pos_item=3, train_max_neg_item=4,
                               test_max_pos_item=3, test_max_neg_item=4)
    ds = ImpressionSeqModel.Dataset(model, corpus, "train")
    assert all(p > 0 for p in ds.data["position"])  # impressions without a click history are dropped
    batch = ds.collate_batch([ds[i] for i in range(12)])
    assert batch["item_id"].shape == (12, 7) and batch["history_items"].shape[1] <= 4
    assert (batch["lengths"] >= 1).all() and batch["history_items"].dtype == torch.long
    f0 = ds[0]
    u, pos = f0["user_id"], ds.data["position"][0]
    assert f0["history_items"].tolist() == [x[0] for x in corpus.user_his[u]["pos"][:pos]][-4:]
