"""GPU: the device-resident data pipeline (rechorus_amd/pipeline.py) against the mirror's own CPU
Dataset / collate path (which restates the reference's), and the on-device evaluation against
evaluate_method(predict())."""
import argparse
import os
import re
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT
from synth_data import make_dataset

pytestmark = pytest.mark.gpu

PLUGIN = os.path.join(ROOT, "rechorus_amd", "rechorus")
if PLUGIN not in sys.path:
    sys.path.insert(0, PLUGIN)


@pytest.fixture(scope="module")
def data_root(tmp_path_factory):
    root = str(tmp_path_factory.mktemp("pipe"))
    make_dataset(root, "synth", n_users=150, n_items=260, per_user=13, n_neg=99, seed=5)
    return root


def _setup(data_root, cuda, model_name, extra=(), test_all=0, device_pipeline=1):
    import importlib
    import main
    from helpers.BaseRunner import BaseRunner
    model_cls = main.find_class("model", (model_name, ""))
    reader_cls = main.find_class("helper", model_cls.reader)
    p = main.parse_global_args(argparse.ArgumentParser())
    p = reader_cls.parse_data_args(p)
    p = BaseRunner.parse_runner_args(p)
    p = model_cls.parse_model_args(p)
    args = p.parse_args(["--path", data_root + "/", "--dataset", "synth", "--num_neg", "6", "--emb_size", "32",
                         "--test_all", str(test_all), "--device_pipeline", str(device_pipeline), "--num_workers", "0",
                         "--topk", "5,10,50", "--eval_batch_size", "64"] + list(extra))
    args.device, args.model_path, args.log_file, args.train = cuda, "/tmp/rechorus_amd_test/m.pt", "/tmp/rechorus_amd_test/l.txt", 1
    corpus = reader_cls(args)
    torch.manual_seed(0)
    model = model_cls(args, corpus).to(cuda)
    with torch.no_grad():
        for prm in model.parameters():
            prm.mul_(15.0)
    data = {ph: model_cls.Dataset(model, corpus, ph) for ph in ("train", "dev", "test")}
    for d in data.values():
        d.prepare()
    return args, corpus, model, data, BaseRunner(args)


@pytest.mark.parametrize("model_name,extra", [("BPRMF", []), ("SASRec", ["--history_max", "8", "--num_heads", "2"])])
def test_device_batches_equal_the_collated_ones(model_name, extra, data_root, cuda):
    from rechorus_amd import pipeline
    args, corpus, model, data, runner = _setup(data_root, cuda, model_name, extra)
    for phase in ("dev", "test"):
        ds = data[phase]
        assert pipeline.eligible(ds)
        dd = pipeline.device_dataset(ds, cuda)
        idx = torch.arange(len(ds), device=cuda)
        feed = dd.feed(idx)
        want = ds.collate_batch([ds[i] for i in range(len(ds))])
        assert feed["batch_size"] == want["batch_size"] and feed["phase"] == phase
        assert torch.equal(feed["user_id"].cpu(), want["user_id"]) and torch.equal(feed["item_id"].cpu(), want["item_id"])
        if model_name == "SASRec":
            w = want["history_items"].shape[1]  # the reference pads to the batch maximum, the device to history_max
            assert torch.equal(feed["lengths"].cpu(), want["lengths"])
            assert torch.equal(feed["history_items"][:, :w].cpu(), want["history_items"]) and not feed["history_items"][:, w:].any()
            assert torch.equal(feed["history_times"][:, :w].cpu(), want["history_times"])
    # training epoch: every row once, fresh negatives never in the user's train clicked set
    tr = data["train"]
    np.random.seed(3)
    seen_rows, all_neg = [], []
    for batch in runner._batches(tr, 64, train=True):
        assert batch["item_id"].shape[1] == 7 and batch["user_id"].is_cuda
        seen_rows.append(torch.stack([batch["user_id"], batch["item_id"][:, 0]], dim=1).cpu())
        all_neg.append((batch["user_id"].cpu(), batch["item_id"][:, 1:].cpu()))
    got = torch.cat(seen_rows).numpy()
    want = np.stack([np.asarray(tr.data["user_id"], dtype=np.int64), np.asarray(tr.data["item_id"], dtype=np.int64)], axis=1)
    assert sorted(map(tuple, got)) == sorted(map(tuple, want))
    assert not np.array_equal(got, want)  # shuffled
    for users, negs in all_neg:
        assert negs.min() >= 1 and negs.max() < corpus.n_items
        for u, row in zip(users.tolist(), negs.tolist()):
            assert not (set(row) & corpus.train_clicked_set[u])
    first = torch.cat([n for _, n in all_neg])
    second = torch.cat([b["item_id"][:, 1:].cpu() for b in runner._batches(tr, 64, train=True)])
    assert not torch.equal(first, second)  # a new epoch draws new negatives


@pytest.mark.parametrize("model_name,extra", [("BPRMF", []), ("NeuMF", ["--layers", "[32]"]),
                                              ("SASRec", ["--history_max", "8", "--num_heads", "2"])])
def test_device_evaluate_equals_evaluate_method_of_predict(model_name, extra, data_root, cuda):
    args, corpus, model, data, runner = _setup(data_root, cuda, model_name, extra)
    for phase in ("dev", "test"):
        pred = runner.predict(data[phase])
        assert pred.shape == (len(data[phase]), 100)
        want = runner.evaluate_method(pred, runner.topk, runner.metrics)
        got = runner.evaluate(data[phase], runner.topk, runner.metrics)
        assert got.keys() == want.keys() and all(abs(got[k] - want[k]) < 1e-9 for k in got), (got, want)
    # the DataLoader path (what custom datasets use) produces the same predictions
    args0, corpus0, model0, data0, runner0 = _setup(data_root, cuda, model_name, extra, device_pipeline=0)
    model0.load_state_dict(model.state_dict())
    assert np.allclose(runner0.predict(data0["test"]), runner.predict(data["test"]), rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("model_name,extra", [("BPRMF", []), ("SASRec", ["--history_max", "8", "--num_heads", "2"])])
def test_test_all_full_catalogue_rank_matches_materialised_path(model_name, extra, data_root, cuda):
    """--test_all: rc_full_catalogue_rank (no [N, n_items] matrix) vs masking the full score matrix"""
    args, corpus, model, data, runner = _setup(data_root, cuda, model_name, extra, test_all=1)
    ds = data["test"]
    assert hasattr(model, "full_catalogue_vectors")
    fast = runner.evaluate(ds, runner.topk, runner.metrics)
    pred = runner.predict(ds)  # reference semantics: [N, n_items], clicked columns -inf
    assert pred.shape == (len(ds), corpus.n_items) and np.isinf(pred).any()
    slow = runner.evaluate_method(pred, runner.topk, runner.metrics)
    # fp32 summation order differs between the MFMA kernel and the gather-dot kernel: exact ties aside
    rank_fast = runner._full_catalogue_ranks(ds).cpu().numpy()
    rank_slow = (pred >= pred[:, :1]).sum(axis=-1)
    assert (rank_fast == rank_slow).mean() > 0.97 and np.abs(rank_fast - rank_slow).max() <= 2
    for k in fast:
        assert abs(fast[k] - slow[k]) < 0.02, (k, fast[k], slow[k])


def test_test_all_streamed_ranks_for_a_head_without_catalogue_vectors(data_root, cuda):
    """--test_all on NeuMF (no dot-product head: no full_catalogue_vectors): the runner masks and ranks one evaluation batch at a time
    -- [eval_batch_size, n_items] scores alive, not [N, n_items] -- and gets exactly the ranks of the reference's materialised matrix
    (helpers/BaseRunner.py:225-252 + evaluate_method :52-78)"""
    args, corpus, model, data, runner = _setup(data_root, cuda, "NeuMF", ["--layers", "[32]"], test_all=1)
    assert not hasattr(model, "full_catalogue_vectors")
    runner.eval_batch_size = 37          # several ragged batches
    ds = data["test"]
    got = runner.evaluate(ds, runner.topk, runner.metrics)
    pred = runner.predict(ds)            # [N, n_items], clicked columns -inf
    assert pred.shape == (len(ds), corpus.n_items) and np.isinf(pred).any()
    want = runner.evaluate_method(pred, runner.topk, runner.metrics)
    assert got.keys() == want.keys() and all(abs(got[k] - want[k]) < 1e-9 for k in got), (got, want)
    ranks = runner._streamed_test_all_ranks(ds).cpu().numpy()
    assert np.array_equal(ranks, (pred >= pred[:, :1]).sum(axis=-1))


def test_cli_dataloader_path_still_trains(data_root, tmp_path, cuda):
    import main
    log = str(tmp_path / "log" / "run.txt")
    res = main.run(["--model_name", "BPRMF", "--emb_size", "32", "--lr", "5e-3", "--l2", "0", "--dataset", "synth",
                    "--path", data_root + "/", "--epoch", "4", "--num_neg", "4", "--batch_size", "128", "--num_workers", "0",
                    "--device_pipeline", "0", "--regenerate", "1", "--log_file", log, "--topk", "5,10",
                    "--model_path", str(tmp_path / "model" / "m.pt"), "--save_final_results", "0"])
    losses = [float(x) for x in re.findall(r"Epoch \d+\s+loss=([0-9.]+)", open(log).read())]
    assert len(losses) >= 2 and losses[-1] < losses[0], losses


def test_cli_test_all_runs(data_root, tmp_path, cuda):
    import main
    log = str(tmp_path / "log" / "run.txt")
    res = main.run(["--model_name", "BPRMF", "--emb_size", "32", "--lr", "5e-3", "--l2", "0", "--dataset", "synth",
                    "--path", data_root + "/", "--epoch", "4", "--num_neg", "4", "--batch_size", "128", "--num_workers", "0",
                    "--test_all", "1", "--regenerate", "1", "--log_file", log, "--topk", "5,10",
                    "--model_path", str(tmp_path / "model" / "m.pt"), "--save_final_results", "0"])
    text = open(log).read()
    before = float(re.search(r"Test Before Training: \(HR@5:([0-9.]+)", text).group(1))
    after = float(re.search(r"HR@5:([0-9.]+)", res["test"]).group(1))
    assert after >= before


@pytest.mark.parametrize("model_name,extra", [("BPRMF", []), ("NeuMF", ["--layers", "[32]"]),
                                              ("SASRec", ["--history_max", "8", "--num_heads", "2"]),
                                              # B * history_max >= 4096, history_max > 32: the batch-level encoder kernels
                                              # (device-side row offsets and length classes, some classes empty)
                                              ("SASRec", ["--history_max", "40", "--num_heads", "2"])])
@pytest.mark.parametrize("opt", ["Adam", "SGD"])
def test_graph_replayed_training_equals_eager_training(model_name, extra, opt, data_root, cuda):
    """--graph 1 (hipGraph replay of model -> loss -> backward -> optimizer.step, Adam step count on the
    device) trains exactly like the eager loop: same batches, same parameters afterwards"""
    from rechorus_amd import graph as hgraph
    states, losses = [], []
    for use_graph in (0, 1):
        args, corpus, model, data, runner = _setup(data_root, cuda, model_name,
                                                   list(extra) + ["--graph", str(use_graph), "--optimizer", opt, "--lr", "0.01",
                                                                  "--l2", "1e-5", "--batch_size", "128"])
        np.random.seed(11)
        torch.manual_seed(11)
        torch.cuda.manual_seed(11)
        losses.append([runner.fit(data["train"], epoch=e) for e in (1, 2, 3)])
        states.append({k: v.detach().cpu().numpy().copy() for k, v in model.state_dict().items()})
        if use_graph:
            steps = list(runner._graphed.values())
            assert steps and any(s.graph is not None for s in steps)  # a graph was really captured and replayed
    assert np.allclose(losses[0], losses[1], rtol=1e-5, atol=1e-7), losses
    for k in states[0]:
        a, b = states[0][k], states[1][k]
        if opt == "Adam" and k.endswith("k_linear.bias"):
            # the key bias of softmax attention has an analytically zero gradient; Adam turns its round-off
            # noise into +-lr steps (tests/test_gpu_sasrec.py), so two correct runs only agree in magnitude
            assert np.abs(a - b).max() <= 3 * 0.01 * 3 * 10
            continue
        # Adam: the replayed step takes its bias correction from a device-side step counter (float arithmetic in the
        # kernel), the eager one from host doubles -- last-bit differences in the step size, accumulated over ~100
        # steps of size lr = 0.01; SGD has no such term and must agree bit for bit (below)
        assert np.allclose(a, b, rtol=1e-4, atol=2e-6 if opt == "SGD" else 3e-5), (k, np.abs(a - b).max())
        if opt == "SGD":
            assert np.array_equal(a, b), k  # same kernels, same order: bit-identical


@pytest.fixture(scope="module")
def ctx_root(tmp_path_factory):
    from synth_data import make_context_dataset
    root = str(tmp_path_factory.mktemp("pipe_ctx"))
    make_context_dataset(root, "ctr", n_users=90, n_items=70, per_user=12, ctr=True, seed=3)
    make_context_dataset(root, "topk", n_users=90, n_items=70, per_user=10, ctr=False, seed=4)
    make_context_dataset(root, "ctrf", n_users=90, n_items=70, per_user=12, ctr=True, seed=5, numeric=True)
    make_context_dataset(root, "topkf", n_users=90, n_items=70, per_user=10, ctr=False, seed=6, numeric=True)
    return root


# (ctrf / topkf: with the numeric features c_day_f -- int64 -- and i_age_f -- float64 --, models/context/FM.py:38-41)
@pytest.mark.parametrize("mode,dataset", [("CTR", "ctr"), ("TopK", "topk"), ("CTR", "ctrf"), ("TopK", "topkf")])
def test_context_device_batches_equal_the_collated_ones(mode, dataset, ctx_root, cuda):
    """context features (user / item / situation columns) gathered on the device == the reference-style
    Dataset -> collate_batch path, for the CTR and the top-k feed dicts"""
    import main
    from helpers.BaseRunner import BaseRunner
    from rechorus_amd import pipeline
    model_cls = main.find_class("model", ("DeepFM", mode))
    reader_cls = main.find_class("helper", model_cls.reader)
    runner_cls = main.find_class("helper", model_cls.runner)
    p = main.parse_global_args(argparse.ArgumentParser())
    p = reader_cls.parse_data_args(p)
    p = runner_cls.parse_runner_args(p)
    p = model_cls.parse_model_args(p)
    args = p.parse_args(["--path", ctx_root + "/", "--dataset", dataset, "--emb_size", "16", "--layers", "[16]", "--num_neg", "3",
                         "--loss_n", "BCE" if mode == "CTR" else "BPR", "--num_workers", "0", "--include_item_features", "1",
                         "--include_user_features", "1", "--include_situation_features", "1", "--metric", "AUC" if mode == "CTR" else "NDCG,HR"])
    args.device, args.model_path, args.log_file, args.train = cuda, "/tmp/rechorus_amd_test/m.pt", "/tmp/rechorus_amd_test/l.txt", 1
    corpus = reader_cls(args)
    model = model_cls(args, corpus).to(cuda)
    runner = runner_cls(args)
    for phase in ("dev", "test") + (("train",) if mode == "CTR" else ()):
        ds = model_cls.Dataset(model, corpus, phase)
        ds.prepare()
        assert pipeline.dataset_kind(ds) == ("ctr" if mode == "CTR" else "context")
        dd = pipeline.device_dataset(ds, cuda)
        feed = dd.feed(torch.arange(len(ds), device=cuda))
        want = ds.collate_batch([ds[i] for i in range(len(ds))])
        assert set(feed) == set(want), (sorted(feed), sorted(want))
        if dataset.endswith("f"):
            assert want["c_day_f"].dtype == torch.int64 and want["i_age_f"].dtype == torch.float64
        for k, v in want.items():
            if isinstance(v, torch.Tensor):
                assert feed[k].dtype == v.dtype, (k, feed[k].dtype, v.dtype)
                assert torch.equal(feed[k].cpu(), v), k
            else:
                assert feed[k] == v, k
    # and training runs on it
    tr = model_cls.Dataset(model, corpus, "train")
    l1 = runner.fit(tr, epoch=1)
    l2 = runner.fit(tr, epoch=2)
    assert np.isfinite(l1) and np.isfinite(l2)
    res = runner.evaluate(model_cls.Dataset(model, corpus, "dev"), [5], runner.metrics)
    assert all(np.isfinite(v) for v in res.values())


def test_graph_replay_with_dropout_draws_fresh_masks(data_root, cuda):
    """models with dropout are captured too: the NeuMF head kernels key their mask by a device-side seed that
    the captured step bumps itself (engine.step_increment), torch's nn.Dropout advances its generator offset
    per replay; either way every step sees a new mask (the loss of identical batches differs) and training
    still converges"""
    from rechorus_amd import graph as hgraph
    args, corpus, model, data, runner = _setup(data_root, cuda, "NeuMF", ["--layers", "[32]", "--dropout", "0.3", "--lr", "0.01",
                                                                         "--batch_size", "128"])
    model.optimizer = runner._build_optimizer(model)
    model.train()
    step = hgraph.GraphedStep(model)
    batch = next(iter(runner._batches(data["train"], 128, train=True)))
    losses = [float(step.run({k: (v.clone() if isinstance(v, torch.Tensor) else v) for k, v in batch.items()})) for _ in range(8)]
    assert step.graph is not None
    replayed = losses[3:]
    assert len(set(round(x, 7) for x in replayed)) > 1  # same batch, different dropout masks
    first = runner.fit(data["train"], epoch=1)
    last = [runner.fit(data["train"], epoch=e) for e in range(2, 6)][-1]
    assert last < first
