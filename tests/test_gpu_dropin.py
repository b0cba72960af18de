"""GPU: the drop-in claim of the plugin surface.  tests/dropin_models/DropinGRU.py is a model file written against
the REFERENCE's conventions only (plain nn.Embedding / nn.GRU / nn.Linear, `from models.BaseModel import
SequentialModel`, inherited loss, no rechorus_amd import).  (1) Adopted onto the engine it computes what plain torch
computes -- prediction, loss, every parameter gradient, two optimizer steps (torch's own ATen kernels as the
fp32 reference of the same ops, tolerance 1e-5 relative as everywhere).  (2) The plugin's main.py finds it by
name in a user directory, adopts its tables, trains it, evaluates it, checkpoints it."""
import argparse
import copy
import os
import re
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT, assert_close
from synth_data import make_dataset

pytestmark = pytest.mark.gpu

PLUGIN = os.path.join(ROOT, "rechorus_amd", "rechorus")
MODEL_DIR = os.path.join(ROOT, "tests", "dropin_models")
if PLUGIN not in sys.path:
    sys.path.insert(0, PLUGIN)


def _load(monkeypatch):
    monkeypatch.setenv("RECHORUS_MODEL_DIRS", MODEL_DIR)
    import main
    return main, main.find_class("model", ("DropinGRU", ""))


def test_adopted_model_file_matches_plain_torch(monkeypatch, cuda):
    from rechorus_amd import nn as hnn
    _, cls = _load(monkeypatch)
    src = open(os.path.join(MODEL_DIR, "DropinGRU.py")).read()
    code = src.split('"""', 2)[2]  # everything after the docstring
    assert "rechorus_amd" not in code and "HipEmbedding" not in code and "hnn" not in code
    torch.manual_seed(0)
    args = argparse.Namespace(device=cuda, model_path="", buffer=1, num_neg=4, dropout=0, test_all=0, emb_size=32,
                              hidden_size=48, history_max=10)
    corpus = argparse.Namespace(n_users=40, n_items=300)
    plain = cls(args, corpus).to(cuda)
    hip = copy.deepcopy(plain)
    assert hnn.adopt_embeddings(hip) == 2
    assert set(hip.state_dict()) == set(plain.state_dict())           # checkpoint keys unchanged
    assert type(hip.i_embeddings).__name__ == "HipEmbedding" and type(plain.i_embeddings) is torch.nn.Embedding
    rng = np.random.default_rng(3)
    B, L, C = 64, 10, 5
    lengths = rng.integers(1, L + 1, size=B)
    hist = rng.integers(1, 300, size=(B, L)) * (np.arange(L)[None, :] < lengths[:, None])
    feed = {"user_id": torch.from_numpy(rng.integers(1, 40, size=B)).to(cuda),
            "item_id": torch.from_numpy(rng.integers(1, 300, size=(B, C))).to(cuda),
            "history_items": torch.from_numpy(hist).to(cuda), "lengths": torch.from_numpy(lengths).to(cuda),
            "batch_size": B, "phase": "train"}
    opts = []
    for m in (plain, hip):
        opts.append(torch.optim.Adam(m.customize_parameters(), lr=1e-2, weight_decay=1e-4))
    for step in range(2):
        outs = []
        for m, opt in zip((plain, hip), opts):
            opt.zero_grad()
            pred = m(feed)["prediction"]
            # the plain twin takes the reference's loss formula in torch ops (models/BaseModel.py:182-185)
            if m is plain:
                pos, neg = pred[:, 0], pred[:, 1:]
                w = (neg - neg.max()).softmax(dim=1)
                loss = -(((pos[:, None] - neg).sigmoid() * w).sum(dim=1)).clamp(min=1e-8, max=1 - 1e-8).log().mean()
            else:
                loss = m.loss({"prediction": pred})
            loss.backward()
            outs.append((pred.detach().cpu().numpy(), float(loss), {k: p.grad.detach().cpu().numpy() for k, p in m.named_parameters()}))
            opt.step()
        assert_close(outs[1][0], outs[0][0], what=f"prediction step {step}")
        assert_close(outs[1][1], outs[0][1], what=f"loss step {step}")
        for k in outs[0][2]:
            assert_close(outs[1][2][k], outs[0][2][k], atol_scale=3e-5, what=f"grad {k} step {step}")
    for (k, a), (_, b) in zip(plain.state_dict().items(), hip.state_dict().items()):
        assert_close(b.cpu().numpy(), a.cpu().numpy(), atol_scale=3e-5, what=f"{k} after two steps",
                     loose="two Adam steps: the normalised update amplifies a 1e-7 gradient difference (observed 0.96 of the allowance)")


def test_cli_trains_a_user_model_file(monkeypatch, tmp_path, cuda):
    main, _ = _load(monkeypatch)
    root = str(tmp_path / "data")
    make_dataset(root, "synth")
    log = str(tmp_path / "log" / "run.txt")
    res = main.run(["--model_name", "DropinGRU", "--emb_size", "32", "--hidden_size", "48", "--history_max", "10",
                    "--lr", "5e-3", "--l2", "1e-6", "--dataset", "synth", "--path", root + "/", "--epoch", "6", "--num_neg", "4",
                    "--batch_size", "128", "--num_workers", "0", "--regenerate", "1", "--log_file", log,
                    "--model_path", str(tmp_path / "model" / "m.pt"), "--topk", "5,10", "--save_final_results", "0"])
    text = open(log).read()
    assert "Adopted 2 nn.Embedding table(s) onto the HIP engine" in text
    assert "[HIP]" in text                      # the printed module tree shows the adopted tables
    losses = [float(x) for x in re.findall(r"Epoch \d+\s+loss=([0-9.]+)", text)]
    assert len(losses) >= 2 and losses[-1] < losses[0], losses
    before = float(re.search(r"Test Before Training: \(HR@5:([0-9.]+)", text).group(1))
    after = float(re.search(r"HR@5:([0-9.]+)", res["test"]).group(1))
    assert after > before, (before, after)
    assert os.path.exists(str(tmp_path / "model" / "m.pt"))
