"""CPU: host-side plugin surface (readers, datasets, sampler, metrics, arg surface) -- the parts
of the mirror that never touch the GPU."""
import argparse
import os
import sys

import numpy as np
import pytest

from conftest import ROOT
from synth_data import make_dataset

PLUGIN = os.path.join(ROOT, "rechorus_amd", "rechorus")
if PLUGIN not in sys.path:
    sys.path.insert(0, PLUGIN)


@pytest.fixture(scope="module")
def corpus(tmp_path_factory):
    from helpers.SeqReader import SeqReader
    root = str(tmp_path_factory.mktemp("data"))
    make_dataset(root, "synth")
    args = argparse.Namespace(path=root + "/", dataset="synth", sep="\t")
    return SeqReader(args)


def test_reader_statistics_and_history(corpus):
    assert corpus.n_users == 61 and corpus.n_items <= 201
    assert set(corpus.data_df) == {"train", "dev", "test"}
    assert len(corpus.data_df["dev"]) == 60 and len(corpus.data_df["test"]) == 60
    u = 7
    his = corpus.user_his[u]
    assert [t for _, t in his] == sorted(t for _, t in his)
    train_items = set(corpus.data_df["train"].query("user_id == @u")["item_id"])
    assert corpus.train_clicked_set[u] == train_items
    assert len(corpus.residual_clicked_set[u]) == 2
    # position = index of the interaction in the user's time-ordered history
    row = corpus.data_df["test"].query("user_id == @u").iloc[0]
    assert his[row["position"]][0] == row["item_id"] and row["position"] == len(his) - 1


def _model_stub(corpus, num_neg=4, history_max=5):
    return argparse.Namespace(buffer=1, num_neg=num_neg, test_all=0, history_max=history_max)


def test_negative_sampler_and_collate(corpus):
    from models.BaseModel import GeneralModel, SequentialModel
    ds = GeneralModel.Dataset(_model_stub(corpus), corpus, "train")
    np.random.seed(0)
    ds.actions_before_epoch()
    negs = np.asarray(ds.data["neg_items"])
    assert negs.shape == (len(ds), 4) and negs.min() >= 1 and negs.max() < corpus.n_items
    for u, row in zip(ds.data["user_id"], negs):
        assert not (set(row.tolist()) & corpus.train_clicked_set[u])
    batch = ds.collate_batch([ds[i] for i in range(10)])
    assert batch["item_id"].shape == (10, 5) and batch["user_id"].shape == (10,)
    assert batch["batch_size"] == 10 and batch["phase"] == "train"
    # negatives are uniform over non-clicked items: chi-square-ish sanity on a big sample
    big = GeneralModel.Dataset(_model_stub(corpus, num_neg=400), corpus, "train")
    big.actions_before_epoch()
    counts = np.bincount(np.asarray(big.data["neg_items"]).ravel(), minlength=corpus.n_items)[1:]
    assert counts.std() / counts.mean() < 0.1

    sds = SequentialModel.Dataset(_model_stub(corpus), corpus, "train")
    sds.actions_before_epoch()
    feeds = [sds[i] for i in range(12)]
    b = sds.collate_batch(feeds)
    L = max(f["lengths"] for f in feeds)
    assert b["history_items"].shape == (12, L) and L <= 5
    for r, f in enumerate(feeds):
        assert (b["history_items"][r, :f["lengths"]].numpy() == f["history_items"]).all()
        assert (b["history_items"][r, f["lengths"]:] == 0).all()  # right padding with 0
    dev = GeneralModel.Dataset(_model_stub(corpus), corpus, "dev")
    dev.prepare()
    assert dev[0]["item_id"].shape == (100,)


def test_metrics_and_formatting():
    from helpers.BaseRunner import BaseRunner
    from oracle import bprmf_oracle as O
    from utils import utils
    rng = np.random.default_rng(0)
    pred = rng.normal(size=(200, 100)).astype(np.float32)
    pred[::7, 3] = pred[::7, 0]  # ties count against the target
    got = BaseRunner.evaluate_method(pred, [1, 5, 10], ["HR", "NDCG"])
    want = O.evaluate_method(pred, [1, 5, 10], ["HR", "NDCG"])
    assert got.keys() == want.keys() and all(abs(got[k] - want[k]) < 1e-12 for k in got)
    with pytest.raises(ValueError):
        BaseRunner.evaluate_method(pred, [5], ["MAP"])
    s = utils.format_metric({"HR@5": 0.25, "NDCG@5": 0.125, "HR@10": 0.5, "NDCG@10": 0.25})
    assert s == "HR@5:0.2500,NDCG@5:0.1250,HR@10:0.5000,NDCG@10:0.2500"
    assert utils.non_increasing([3, 2, 3, 1]) and not utils.non_increasing([1, 2])


def test_cli_flag_surface_matches_reference():
    """same flag names/defaults as docs/Main_Arguments.md of the reference"""
    from helpers.BaseReader import BaseReader
    from helpers.BaseRunner import BaseRunner
    from models.general.BPRMF import BPRMF
    import main
    p = argparse.ArgumentParser()
    p = main.parse_global_args(p)
    p = BaseReader.parse_data_args(p)
    p = BaseRunner.parse_runner_args(p)
    p = BPRMF.parse_model_args(p)
    a = p.parse_args([])
    want = dict(gpu="0", random_seed=0, load=0, train=1, regenerate=0, path="data/", sep="\t",
                dataset="Grocery_and_Gourmet_Food", epoch=200, early_stop=10, lr=1e-3, l2=0, batch_size=256,
                eval_batch_size=256, optimizer="Adam", num_workers=5, topk="5,10,20,50", metric="NDCG,HR",
                emb_size=64, num_neg=1, dropout=0, test_all=0, buffer=1, model_path="")
    for k, v in want.items():
        assert getattr(a, k) == v, k
    assert BPRMF.reader == "BaseReader" and BPRMF.runner == "BaseRunner"
    assert BPRMF.extra_log_args == ["emb_size", "batch_size"]


# ---- context-aware / CTR surface ------------------------------------------------------------------

@pytest.fixture(scope="module")
def ctx_corpus(tmp_path_factory):
    from helpers.ContextReader import ContextReader
    from synth_data import make_context_dataset
    root = str(tmp_path_factory.mktemp("ctx"))
    make_context_dataset(root, "synth_ctx", ctr=True)
    args = argparse.Namespace(path=root + "/", dataset="synth_ctx", sep="\t", include_item_features=1,
                              include_user_features=1, include_situation_features=1)
    return ContextReader(args)


def test_context_reader_collects_features(ctx_corpus):
    c = ctx_corpus
    assert c.item_feature_names == ["i_category_c"] and c.user_feature_names == ["u_age_c", "u_gender_c"]
    assert c.situation_feature_names == ["c_hour_c", "c_weekday_c"]
    assert c.feature_max["user_id"] == c.n_users and c.feature_max["item_id"] == c.n_items
    assert c.feature_max["c_hour_c"] <= 24 and c.feature_max["c_weekday_c"] <= 7 and c.feature_max["u_gender_c"] == 2
    assert set(c.item_features[5]) == {"i_category_c"} and set(c.user_features[3]) == {"u_age_c", "u_gender_c"}
    assert "label" in c.all_df.columns


def test_context_ctr_dataset_feed_and_flags(ctx_corpus):
    from models.BaseContextModel import ContextCTRModel
    from models.context.DeepFM import DeepFMCTR, DeepFMTopK
    from models.context.FM import FMCTR
    ds = ContextCTRModel.Dataset(argparse.Namespace(buffer=0), ctx_corpus, "train")
    ds.actions_before_epoch()  # labelled data: nothing to sample
    feeds = [ds[i] for i in range(9)]
    b = ds.collate_batch(feeds)
    assert b["item_id"].shape == (9, 1) and b["label"].shape == (9, 1) and b["i_category_c"].shape == (9, 1)
    assert b["u_age_c"].shape == (9,) and b["c_hour_c"].shape == (9,)
    row = ctx_corpus.data_df["train"].iloc[0]
    assert feeds[0]["c_hour_c"] == ds.data["c_hour_c"][0]
    assert feeds[0]["i_category_c"][0] == ctx_corpus.item_features[feeds[0]["item_id"][0]]["i_category_c"]
    # flag surface / class wiring as in the reference
    assert (DeepFMCTR.reader, DeepFMCTR.runner) == ("ContextReader", "CTRRunner")
    assert (DeepFMTopK.reader, DeepFMTopK.runner) == ("ContextReader", "BaseRunner")
    a = DeepFMCTR.parse_model_args(argparse.ArgumentParser()).parse_args([])
    assert (a.emb_size, a.layers, a.loss_n) == (64, "[64]", "BPR")  # reference quirk: WideDeepCTR inherits 'BPR'
    assert FMCTR.parse_model_args(argparse.ArgumentParser()).parse_args([]).loss_n == "BCE"


def test_ctr_metrics():
    from helpers.CTRRunner import CTRRunner
    p = np.array([0.9, 0.2, 0.6, 0.4, 0.0, 1.0])
    y = np.array([1, 0, 0, 1, 0, 1])
    r = CTRRunner.evaluate_method(p, y, ["ACC", "AUC", "F1_SCORE", "LOG_LOSS"])
    assert abs(r["ACC"] - 4 / 6) < 1e-12
    assert abs(r["AUC"] - 8 / 9) < 1e-12  # 8 of the 9 (pos, neg) pairs ordered correctly
    assert abs(r["F1_SCORE"] - 2 * (2 / 3) * (2 / 3) / (4 / 3)) < 1e-12
    q = np.clip(p, 1e-7, 1 - 1e-7)
    assert abs(r["LOG_LOSS"] + (np.log(q) * y + np.log(1 - q) * (1 - y)).mean()) < 1e-12
    with pytest.raises(ValueError):
        CTRRunner.evaluate_method(p, y, ["HR"])


def test_mlp_block_layout_matches_reference_keys():
    from utils.layers import MLP_Block
    m = MLP_Block(12, [8, 4], hidden_activations="ReLU", dropout_rates=0.2, output_dim=1)
    # Linear, ReLU, Dropout, Linear, ReLU, Dropout, Linear  -> the reference's state_dict indices
    assert sorted(m.state_dict()) == ["mlp.0.bias", "mlp.0.weight", "mlp.3.bias", "mlp.3.weight", "mlp.6.bias", "mlp.6.weight"]
    m = MLP_Block(12, [8], hidden_activations="Dice", batch_norm=True, output_dim=None)
    assert "mlp.2.alpha" in m.state_dict() and "mlp.1.running_mean" in m.state_dict()


def test_pipeline_host_side_pieces(corpus, ctx_corpus, tmp_path):
    """rechorus_amd.pipeline: which datasets may be assembled on the device, and its host-built tables"""
    import torch
    from models.BaseContextModel import ContextCTRModel, ContextModel
    from models.BaseModel import GeneralModel, SequentialModel
    from rechorus_amd import pipeline
    stub = _model_stub(corpus)
    assert pipeline.dataset_kind(GeneralModel.Dataset(stub, corpus, "train")) == "general"
    assert pipeline.dataset_kind(SequentialModel.Dataset(stub, corpus, "dev")) == "sequential"
    assert pipeline.dataset_kind(ContextCTRModel.Dataset(argparse.Namespace(buffer=0), ctx_corpus, "train")) == "ctr"

    class Custom(GeneralModel.Dataset):  # a model file with its own feed dict keeps the DataLoader path
        def _get_feed_dict(self, index):
            return super()._get_feed_dict(index)
    assert pipeline.dataset_kind(Custom(stub, corpus, "train")) is None and not pipeline.eligible(Custom(stub, corpus, "train"))

    ptr, flat = pipeline._csr({1: {5, 3, 9}, 3: {2}}, 5, torch.device("cpu"))
    assert ptr.tolist() == [0, 0, 3, 3, 4, 4] and flat.tolist() == [3, 5, 9, 2]
    ptr, flat = pipeline.clicked_csr(corpus, torch.device("cpu"), "train")
    u = 7
    assert set(flat[ptr[u]:ptr[u + 1]].tolist()) == corpus.train_clicked_set[u]
    ptr_all, flat_all = pipeline.clicked_csr(corpus, torch.device("cpu"), "all")
    assert set(flat_all[ptr_all[u]:ptr_all[u + 1]].tolist()) == corpus.train_clicked_set[u] | corpus.residual_clicked_set[u]
    hp, hi, ht = pipeline.history_csr(corpus, torch.device("cpu"))
    assert hi[hp[u]:hp[u + 1]].tolist() == [x[0] for x in corpus.user_his[u]]
    assert ht[hp[u]:hp[u + 1]].tolist() == [x[1] for x in corpus.user_his[u]]
    assert pipeline._padded([[4, 5, 6], [], [7]], 2).tolist() == [[4, 5], [0, 0], [7, 0]]
    user, item = pipeline.context_tables(ctx_corpus, torch.device("cpu"))
    assert int(item["i_category_c"][5]) == ctx_corpus.item_features[5]["i_category_c"]
    assert int(user["u_age_c"][3]) == ctx_corpus.user_features[3]["u_age_c"]


def test_sasrec_trainer_graph_mode_host_rules():
    """SasrecTrainer(graph=True): host-side rules that need no GPU -- Adam keeps its step count in device memory only with
    row-wise updates, and only batch shapes whose item update takes the one-wave-per-row route use it (others run eagerly on
    the host's step count)"""
    import torch
    from rechorus_amd import engine
    d, L, n_items = 64, 50, 1000
    mk = lambda *s: torch.zeros(s)
    lay = {k: (mk(d, d) if k.startswith("W") else mk(d)) for k in engine.SAS_LAYER_KEYS}
    P = {"item_emb": mk(n_items, d), "pos_emb": mk(L + 1, d), "layers": [lay]}
    with pytest.raises(ValueError):
        engine.SasrecTrainer(P, 4, opt="Adam", rowwise=False, graph=True)
    tr = engine.SasrecTrainer(P, 4, opt="Adam", rowwise=True, graph=True)
    assert tr._step_dev is not None and int(tr._step_dev) == 0
    big = (torch.zeros((512, L), dtype=torch.int64), torch.zeros((512, 100), dtype=torch.int64))    # 76,800 occurrences over 1,000 rows
    small = (torch.zeros((4, L), dtype=torch.int64), torch.zeros((4, 2), dtype=torch.int64))         # 208 occurrences: head-list route
    assert tr._dev_step_route(*big) and not tr._dev_step_route(*small)
    sgd = engine.SasrecTrainer(P, 4, opt="SGD", rowwise=True, graph=True)
    assert sgd._step_dev is None and not sgd._dev_step_route(*big)


def test_sasrec_kernel_choice_by_batch_shape(monkeypatch):
    """engine._sasrec_impl: per-sequence kernels while the batch is one round of resident workgroups in the 32-row
    geometry, batch-level kernels beyond; explicit choice and the RC_SASREC_IMPL override win"""
    from rechorus_amd import engine
    monkeypatch.delenv("RC_SASREC_IMPL", raising=False)
    assert engine._sasrec_impl(256, 20, None) == "sequence"      # the reference's demo flags
    assert engine._sasrec_impl(16, 50, None) == "sequence"       # tiny batch
    assert engine._sasrec_impl(256, 50, None) == "batch"
    assert engine._sasrec_impl(4096, 20, None) == "batch"
    assert engine._sasrec_impl(4096, 50, "sequence") == "sequence"
    # one block without dropout: the batch encoder's one-row path is the whole encoder -> batch kernels at every size
    assert engine._sasrec_impl(256, 20, None, 64, 4, 1) == "batch"
    assert engine._sasrec_impl(256, 20, None, 64, 1, 1) == "batch"
    assert engine._sasrec_impl(256, 20, None, 64, 4, 2) == "sequence"     # two blocks: the old rule
    assert engine._sasrec_impl(256, 20, None, 64, 8, 1) == "sequence"     # eight heads: not covered by that path
    assert engine._sasrec_impl(256, 4, None, 64, 4, 1) == "sequence"      # history_max < heads + 1
    assert engine._sasrec_impl(256, 20, "sequence", 64, 4, 1) == "sequence"
    monkeypatch.setenv("RC_SASREC_IMPL", "batch")
    assert engine._sasrec_impl(16, 8, None) == "batch"
    with pytest.raises(ValueError):
        engine._sasrec_impl(16, 8, "fastest")


def test_bench_workload_defaults(monkeypatch):
    """bench.py: --workload sasrec switches to the Grocery-sized catalogue and B = 4096 unless told otherwise"""
    import importlib.util
    import sys
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--workload", "sasrec"])
    a = bench.parse()
    assert (a.items, a.batch, a.hist, a.heads) == (8714, 4096, 50, 4)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--workload", "sasrec", "--items", "50000", "--batch", "1024"])
    a = bench.parse()
    assert (a.items, a.batch) == (50000, 1024)
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    a = bench.parse()
    assert (a.workload, a.items, a.batch, a.num_neg, a.emb_size, a.gpus) == ("bprmf", 10_000_001, 65536, 99, 64, 1)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--workload", "neumf"])   # configs[3] shape
    a = bench.parse()
    assert (a.emb_size, a.num_neg, a.hidden, a.batch) == (128, 4, 64, 65536)


def test_engine_auto_respects_the_models_rowwise_predicate(caplog):
    """--engine auto: row-wise only when the tables are large AND the model's fused step supports the configuration
    (a default SASRec --dropout 0.2 run on a large catalogue must take the dense path, not raise); the choice is logged"""
    import logging
    import torch
    from helpers.BaseRunner import BaseRunner

    class Big(torch.nn.Module):
        def __init__(self, ok):
            super().__init__()
            self.t = torch.nn.Parameter(torch.empty((1 << 20) + 1, 1))
            self.ok = ok

        def hip_train_step(self, *a):
            raise RuntimeError("unsupported configuration")

        def hip_rowwise_supported(self):
            return self.ok

    def runner(engine):
        r = BaseRunner.__new__(BaseRunner)
        r.optimizer_name, r.engine = "Adam", engine
        return r

    with caplog.at_level(logging.INFO):
        assert runner("auto")._use_rowwise(Big(True)) is True
        assert runner("auto")._use_rowwise(Big(False)) is False
        assert runner("rowwise")._use_rowwise(Big(False)) is True     # explicit request: hip_train_step raises later
        assert runner("dense")._use_rowwise(Big(True)) is False
    text = caplog.text
    assert "Engine: row-wise updates" in text and "Engine: dense updates" in text
    assert "no fused row-wise step" in text
    small = torch.nn.Linear(4, 4)
    small.hip_train_step = lambda *a: None
    assert runner("auto")._use_rowwise(small) is False


def test_context_models_get_the_reference_data_appendix(monkeypatch, tmp_path):
    """reference main.py: readers named *Context* append _context<item><user><situation> to the corpus pickle / log /
    model names, so runs with different --include_*_features never share a cached corpus"""
    import main as plugin_main
    seen = {}

    class Stop(Exception):
        pass

    def fake_check_dir(path):
        seen["log"] = path
        raise Stop

    monkeypatch.setattr(plugin_main.utils, "check_dir", fake_check_dir)
    with pytest.raises(Stop):
        plugin_main.run(["--model_name", "DeepFM", "--model_mode", "CTR", "--dataset", "d", "--include_item_features", "1",
                         "--include_situation_features", "1", "--path", str(tmp_path) + "/"])
    assert "d_context101" in seen["log"]
    with pytest.raises(Stop):
        plugin_main.run(["--model_name", "BPRMF", "--dataset", "d", "--path", str(tmp_path) + "/"])
    assert "_context" not in seen["log"]
