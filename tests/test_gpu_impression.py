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
                    "--log_file", log, "--model_path", str(tmp_path / "model" / "m.pt"), "--save_final_results", "0"])
    text = open(log).read()
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
