"""GPU: the impression path on the HIP engine -- list-level BPR and softmax-CE kernels vs the reference's own
loss / gradient (tests/golden/impression_losses_metrics.npz) and the numpy oracle; BPRMFImpression end to end."""
import argparse
import os
import re
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT, assert_close
from oracle import impression_oracle as IO
from synth_data import make_impression_dataset
from test_impression_cpu import LOSS_CASES, case

pytestmark = pytest.mark.gpu

PLUGIN = os.path.join(ROOT, "rechorus_amd", "rechorus")
if PLUGIN not in sys.path:
    sys.path.insert(0, PLUGIN)


@pytest.mark.parametrize("key", LOSS_CASES)
def test_impression_losses_on_the_device_match_the_reference(key, cuda):
    from models.BaseImpressionModel import ImpressionModel
    c = case(key)
    name = key.split("/")[-1]
    stub = argparse.Namespace(loss_n=name, train_max_pos_item=int(c["max_pos"]))
    p = torch.from_numpy(c["pred"]).to(cuda).requires_grad_(True)
    loss = ImpressionModel.loss(stub, {"prediction": p}, torch.from_numpy(c["target"]).to(cuda))
    loss.backward()
    assert_close(loss.item(), c["loss"], what="loss " + key, rtol=2e-5)
    assert_close(p.grad.cpu().numpy(), c["gpred"], what="grad " + key, rtol=2e-5, atol_scale=2e-5)


@pytest.mark.parametrize("name", ["BPRsimple", "BPRhardsimple"])
def test_bpr_simple_kernel_matches_the_reference(name, cuda):
    """rc_list_loss_fwd_bwd kind 9 vs the reference's own forward (unreduced rows) and autograd (of rows.sum()), through the
    mirror's ImpressionModel.loss: values, gradient with a unit and with a per-row incoming gradient"""
    import argparse
    from models.BaseImpressionModel import ImpressionModel
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "impression_bpr_simple.npz"))
    for shape_id in range(3):
        key = "loss/{}/{}/".format(shape_id, name)
        stub = argparse.Namespace(loss_n=name, train_max_pos_item=int(g[key + "max_pos"]))
        p = torch.from_numpy(g[key + "pred"]).to(cuda).requires_grad_(True)
        rows = ImpressionModel.loss(stub, {"prediction": p}, torch.from_numpy(g[key + "target"]).to(cuda))
        assert tuple(rows.shape) == (p.shape[0],)
        assert_close(rows.detach().cpu().numpy(), g[key + "rows"], what=name + " rows")
        w = torch.linspace(0.5, 2.0, p.shape[0], device=cuda)
        (rows * w).sum().backward()
        assert_close(p.grad.cpu().numpy(), g[key + "gpred"] * w.cpu().numpy()[:, None], what=name + " grad", atol_scale=2e-5)


def test_list_bpr_kernel_random_shapes_vs_oracle(cuda):
    from rechorus_amd import engine
    rng = np.random.default_rng(3)
    for B, mp, mn in ((1, 1, 1), (7, 2, 130), (300, 20, 20), (5, 70, 3), (64, 64, 64)):
        n = mp + mn
        pred = rng.normal(0, 2, size=(B, n)).astype(np.float32)
        target = np.full((B, n), -1, dtype=np.int64)
        for b in range(B):
            target[b, :rng.integers(1, mp + 1)] = 1
            target[b, mp:mp + rng.integers(1, mn + 1)] = 0
        for hard in (False, True):
            loss, g = engine.list_bpr(torch.from_numpy(pred).to(cuda), torch.from_numpy(target).to(cuda), mp, hard=hard)
            want_loss, _, want_g = IO.list_bpr(pred, target, mp, hard=hard)
            # -log Q with Q -> 1 is conditioned by the fp32 ulp of Q (6e-8), not of the loss
            assert_close(loss.cpu().numpy()[0], want_loss, what=f"loss {B}x{mp}+{mn}", rtol=2e-5, abs_floor=2e-7)
            assert_close(g.cpu().numpy(), want_g, what=f"grad {B}x{mp}+{mn}", rtol=2e-5, atol_scale=2e-5,
                         abs_floor=2e-7 / B)  # sigmoid'(x) = s (1 - s): the same cancellation
            assert (g.cpu().numpy()[target == -1] == 0).all()


def test_cli_bprmf_impression(tmp_path, cuda):
    import main
    make_impression_dataset(str(tmp_path), "imp", n_users=300, n_items=120, n_imp=12, seed=2)
    log = str(tmp_path / "log" / "run.txt")
    res = main.run(["--model_name", "BPRMF", "--model_mode", "Impression", "--emb_size", "32", "--lr", "5e-3", "--l2", "0",
                    "--loss_n", "BPR", "--dataset", "imp", "--path", str(tmp_path) + "/", "--epoch", "8", "--batch_size", "128",
                    "--num_workers", "0", "--regenerate", "1", "--metric", "NDCG,HR", "--topk", "1,2,3,5", "--main_metric", "NDCG@2",
                    "--log_file", log, "--model_path", str(tmp_path / "model" / "m.pt"), "--save_final_results", "1"])
    text = open(log).read()
    rec = (tmp_path / "log" / "run" / "rec-BPRMFImpression-dev.csv").read_text().splitlines()
    assert rec[0].split("\t") == ["user_id", "pos_items", "pos_predictions", "neg_items", "neg_predictions"] and len(rec) > 10
    losses = [float(x) for x in re.findall(r"Epoch \d+\s+loss=([0-9.]+)", text)]
    assert len(losses) >= 2 and losses[-1] < losses[0], losses
    before = float(re.search(r"Test Before Training: \(.*?NDCG@2:([0-9.]+)", text).group(1))
    after = float(re.search(r"NDCG@2:([0-9.]+)", res["test"]).group(1))
    assert after > before, (before, after)  # clicks follow a user x item affinity
    assert "MAP@2" in res["test"]


def test_cli_sasrec_impression(tmp_path, cuda):
    import main
    make_impression_dataset(str(tmp_path), "imp", n_users=300, n_items=120, n_imp=12, seed=2)
    log = str(tmp_path / "log" / "run.txt")
    res = main.run(["--model_name", "SASRec", "--model_mode", "Impression", "--emb_size", "32", "--num_layers", "1",
                    "--num_heads", "2", "--history_max", "10", "--lr", "3e-3", "--l2", "0", "--loss_n", "BPR", "--dataset", "imp",
                    "--path", str(tmp_path) + "/", "--epoch", "6", "--batch_size", "128", "--num_workers", "0", "--regenerate", "1",
                    "--metric", "NDCG,HR", "--topk", "1,2,3,5", "--main_metric", "NDCG@2", "--log_file", log,
                    "--model_path", str(tmp_path / "model" / "m.pt"), "--save_final_results", "0"])
    text = open(log).read()
    losses = [float(x) for x in re.findall(r"Epoch \d+\s+loss=([0-9.]+)", text)]
    assert len(losses) >= 2 and losses[-1] < losses[0], losses
    assert "NDCG@2" in res["test"] and "MAP@5" in res["test"]


@pytest.mark.parametrize("model_name,extra", [("BPRMF", []), ("SASRec", ["--history_max", "5", "--num_heads", "2"])])
def test_impression_device_batches_equal_the_collated_ones(model_name, extra, tmp_path, cuda):
    """impression lists (+ clicked / skipped histories) assembled on the device == Dataset -> collate_batch"""
    import main
    from rechorus_amd import pipeline
    make_impression_dataset(str(tmp_path), "imp", n_users=60, n_items=80, n_imp=9, seed=6)
    model_cls = main.find_class("model", (model_name, "Impression"))
    reader_cls = main.find_class("helper", model_cls.reader)
    runner_cls = main.find_class("helper", model_cls.runner)
    p = main.parse_global_args(argparse.ArgumentParser())
    p = reader_cls.parse_data_args(p)
    p = runner_cls.parse_runner_args(p)
    p = model_cls.parse_model_args(p)
    args = p.parse_args(["--path", str(tmp_path) + "/", "--dataset", "imp", "--emb_size", "32", "--num_workers", "0",
                         "--train_max_pos_item", "3", "--train_max_neg_item", "4", "--test_max_pos_item", "2",
                         "--test_max_neg_item", "5", "--metric", "NDCG,HR", "--topk", "1,2"] + extra)
    args.device, args.model_path, args.log_file, args.train = cuda, "/tmp/rechorus_amd_test/m.pt", "/tmp/rechorus_amd_test/l.txt", 1
    corpus = reader_cls(args)
    model = model_cls(args, corpus).to(cuda)
    runner = runner_cls(args)
    for phase in ("train", "dev"):
        ds = model_cls.Dataset(model, corpus, phase)
        ds.prepare()
        assert pipeline.dataset_kind(ds) == ("impression" if model_name == "BPRMF" else "impression_seq")
        feed = pipeline.device_dataset(ds, cuda).feed(torch.arange(len(ds), device=cuda))
        want = ds.collate_batch([ds[i] for i in range(len(ds))])
        assert set(feed) == set(want), (sorted(feed), sorted(want))
        for k, v in want.items():
            if not isinstance(v, torch.Tensor):
                assert feed[k] == v, k
            elif "history" in k:  # the reference pads to the batch maximum, the device to history_max
                w = v.shape[1]
                assert torch.equal(feed[k][:, :w].cpu(), v) and not feed[k][:, w:].any(), k
            else:
                assert torch.equal(feed[k].cpu(), v), k
    tr = model_cls.Dataset(model, corpus, "train")
    assert np.isfinite(runner.fit(tr, epoch=1))
    res = runner.evaluate(model_cls.Dataset(model, corpus, "dev"), [1, 2], ["NDCG", "HR"])
    assert set(res) == {"NDCG@1", "NDCG@2", "MAP@1", "MAP@2", "HR@1", "HR@2"}


def test_bprmf_impression_returns_u_v_and_i_v(cuda):
    """reference BPRMF.py:43-45,79-80: the impression variant's forward also returns the tiled user vectors and the
    gathered item vectors (what the rerankers read); BPRMF proper does not"""
    from models.general.BPRMF import BPRMF, BPRMFImpression
    args = argparse.Namespace(device=cuda, model_path="", buffer=1, num_neg=4, dropout=0, test_all=0, emb_size=32,
                              loss_n="BPR", train_max_pos_item=2, train_max_neg_item=3, test_max_pos_item=2, test_max_neg_item=3)
    corpus = argparse.Namespace(n_users=11, n_items=23)
    rng = np.random.default_rng(0)
    feed = {"user_id": torch.from_numpy(rng.integers(0, 11, size=6)).to(cuda),
            "item_id": torch.from_numpy(rng.integers(0, 23, size=(6, 5))).to(cuda), "batch_size": 6, "phase": "train"}
    m = BPRMFImpression(args, corpus).to(cuda)
    out = m(feed)
    U, I = m.u_embeddings.weight.detach().cpu().numpy(), m.i_embeddings.weight.detach().cpu().numpy()
    uid, iid = feed["user_id"].cpu().numpy(), feed["item_id"].cpu().numpy()
    assert out["u_v"].shape == (6, 5, 32) and out["i_v"].shape == (6, 5, 32)
    assert np.array_equal(out["u_v"].detach().cpu().numpy(), np.repeat(U[uid][:, None, :], 5, axis=1))
    assert np.array_equal(out["i_v"].detach().cpu().numpy(), I[iid])
    assert_close(out["prediction"].detach().cpu().numpy(), (U[uid][:, None, :] * I[iid]).sum(-1), what="prediction")
    assert set(BPRMF(args, corpus).to(cuda)(feed)) == {"prediction"}


# ---- list metrics on the device (rc_list_metrics) -----------------------------------------------------------------------

def _device_metrics(c_pred, pos, neg, mp, topk, cuda):
    from rechorus_amd import engine
    per_row, mean = engine.list_metrics(torch.from_numpy(np.ascontiguousarray(c_pred, dtype=np.float32)).to(cuda),
                                        None if pos is None else torch.from_numpy(np.asarray(pos, dtype=np.int64)).to(cuda),
                                        torch.from_numpy(np.asarray(neg, dtype=np.int64)).to(cuda), mp, topk)
    return per_row.cpu().numpy(), mean.cpu().numpy()


@pytest.mark.parametrize("key", ["metric/0", "metric/1"])
def test_list_metrics_on_the_device_match_the_reference(key, cuda):
    """rc_list_metrics vs the reference's own evaluate_method(ret_all=1) (tests/golden/impression_losses_metrics.npz): per-row
    NDCG / MAP / HR @k in float64 to 1e-12, the means to 1e-12"""
    c = case(key)
    topk = [1, 2, 3, 5, 10]
    per_row, mean = _device_metrics(c["pred"], c["pos_num"], c["neg_num"], int(c["max_pos"]), topk, cuda)
    for m, name in enumerate(("NDCG", "MAP", "HR")):
        for j, k in enumerate(topk):
            want = c["res/%s@%d" % (name, k)]
            assert np.allclose(per_row[:, m, j], want, atol=1e-12, rtol=0), (name, k, np.abs(per_row[:, m, j] - want).max())
            assert abs(mean[m, j] - want.mean()) < 1e-12, (name, k)


def test_list_metrics_on_the_device_ties_empty_groups_and_wide_lists(cuda):
    """against the mirror's numpy evaluate_method (itself pinned to the reference's, tests/test_impression_cpu.py) on what the
    golden lists do not hold: scores drawn from four values (ties between positives and negatives and inside a group), rows
    without positives or without negatives, counts beyond the slot widths, garbage (inf / huge) in the unused columns, one
    positive per row (pos_num None), a single column, 2,048 columns, no rows"""
    from helpers.ImpressionRunner import ImpressionRunner
    from rechorus_amd import engine
    rng = np.random.default_rng(5)
    for N, mp, mn, topk, with_pos in ((200, 20, 20, [1, 2, 3, 5, 10, 20, 50], True), (64, 3, 70, [1, 5, 100], True), (50, 1, 9, [1, 3], False),
                                      (17, 1, 0, [1, 2], True), (9, 40, 2008, [1, 10, 1000], True), (33, 0, 5, [2], True)):
        n = mp + mn
        pred = rng.choice(np.array([-1.5, 0.0, 0.25, 2.0], dtype=np.float32), size=(N, n))
        pos = rng.integers(0, mp + 3, size=N) if with_pos else None
        neg = rng.integers(0, mn + 3, size=N)
        # the reference sees -inf outside the valid slots (ImpressionRunner.evaluate :156-168); the kernel must not look there
        p_eff = np.minimum(pos if pos is not None else 1, mp)
        col = np.arange(n)[None, :]
        keep = (col < p_eff[:, None] if pos is not None else col < min(1, mp)) | ((col >= mp) & (col < mp + np.minimum(neg, mn)[:, None]))
        want = ImpressionRunner.evaluate_method(np.where(keep, pred, -np.inf), topk, [], False, neg, mp, pos, ret_all=1)
        dirty = np.where(keep, pred, rng.choice(np.array([np.inf, 1e30, -3.0], dtype=np.float32), size=(N, n)))
        per_row, mean = _device_metrics(dirty, pos, neg, mp, topk, cuda)
        for m, name in enumerate(("NDCG", "MAP", "HR")):
            for j, k in enumerate(topk):
                w = want["%s@%d" % (name, k)]
                assert np.allclose(per_row[:, m, j], w, atol=1e-12, rtol=0), (N, mp, mn, name, k, np.abs(per_row[:, m, j] - w).max())
                assert abs(mean[m, j] - w.mean()) < 1e-12
    per_row, mean = _device_metrics(np.zeros((0, 8), dtype=np.float32), np.zeros(0), np.zeros(0), 3, [1, 2], cuda)
    assert per_row.shape == (0, 3, 2) and not mean.any()
    assert not engine.list_metrics_supported(4096, 20, 3) and not engine.list_metrics_supported(40, 20, 17)


def test_impression_runner_evaluates_on_the_device(tmp_path, cuda, monkeypatch):
    """ImpressionRunner.evaluate: the predictions stay on the device (no BaseRunner.predict, i.e. no [N, n] D2H copy) and the
    result equals the numpy route on the same predictions, means and per-row values (all=1)"""
    import main
    from helpers.BaseRunner import BaseRunner
    make_impression_dataset(str(tmp_path), "imp")
    model_cls = main.find_class("model", ("BPRMF", "Impression"))
    reader_cls, runner_cls = main.find_class("helper", model_cls.reader), main.find_class("helper", model_cls.runner)
    p = main.parse_global_args(argparse.ArgumentParser())
    p = model_cls.parse_model_args(runner_cls.parse_runner_args(reader_cls.parse_data_args(p)))
    args = p.parse_args(["--path", str(tmp_path) + "/", "--dataset", "imp", "--emb_size", "32", "--num_workers", "0", "--train_max_pos_item", "3",
                         "--train_max_neg_item", "4", "--test_max_pos_item", "2", "--test_max_neg_item", "5", "--metric", "NDCG,HR", "--topk", "1,2,5"])
    args.device, args.model_path, args.log_file, args.train = cuda, "/tmp/rechorus_amd_test/m.pt", "/tmp/rechorus_amd_test/l.txt", 1
    corpus = reader_cls(args)
    model = model_cls(args, corpus).to(cuda)
    runner = runner_cls(args)
    runner.fit(model_cls.Dataset(model, corpus, "train"), epoch=1)
    ds = model_cls.Dataset(model, corpus, "dev")
    predictions = BaseRunner.predict(runner, ds)
    pos, neg = np.asarray(ds.data["pos_num"]), np.asarray(ds.data["neg_num"])
    col = np.arange(predictions.shape[1])[None, :]
    keep = (col < np.minimum(pos, 2)[:, None]) | ((col >= 2) & (col < 2 + np.minimum(neg, 5)[:, None]))
    masked = np.where(keep, predictions, -np.inf)

    def no_host_predictions(*a, **k):
        raise AssertionError("ImpressionRunner.evaluate copied the predictions to the host")
    monkeypatch.setattr(BaseRunner, "predict", no_host_predictions)
    for ret_all in (0, 1):
        got = runner.evaluate(ds, [1, 2, 5], ["NDCG", "HR"], all=ret_all)
        want = runner.evaluate_method(masked, [1, 2, 5], ["NDCG", "HR"], False, neg, 2, pos, ret_all=ret_all)
        assert list(got) == list(want)
        for k in want:
            assert np.allclose(got[k], want[k], atol=1e-12, rtol=0), (k, got[k], want[k])
