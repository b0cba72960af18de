"""CPU: the numpy SASRec oracle vs the reference's own outputs (tests/golden/sasrec_*.npz)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN_DIR, assert_close, load_golden
from oracle import bprmf_oracle as BO
from oracle import sasrec_oracle as SO

CASES = sorted(f[:-4] for f in os.listdir(GOLDEN_DIR) if f.startswith("sasrec_") and f.endswith(".npz"))


def params(g, prefix="P0/"):
    return {k[len(prefix):]: v for k, v in g.items() if k.startswith(prefix)}


@pytest.mark.parametrize("case", CASES)
def test_sasrec_forward_loss_grads(case):
    g = load_golden(case)
    P = params(g)
    n_heads = int(g["meta"][3])
    pred = SO.forward(P, g["hist"], g["len"], g["iid"], n_heads)
    assert_close(pred, g["pred"], what="pred", atol_scale=2e-5)
    assert_close(BO.bpr_loss(g["pred"]), g["loss"], what="loss")
    _, G = SO.backward(P, g["hist"], g["len"], g["iid"], n_heads, g["gpred"])
    want = params(g, "G/")
    floor = 1e-6 * max(float(np.abs(v).max()) for v in want.values())
    for k, v in want.items():
        assert_close(G[k], v, what="grad " + k, rtol=2e-5, atol_scale=5e-5, abs_floor=floor)


DROP_CASES = sorted(f[:-4] for f in os.listdir(GOLDEN_DIR) if f.startswith("sasrecdrop_") and f.endswith(".npz"))


@pytest.mark.parametrize("case", DROP_CASES)
def test_sasrec_training_mode_dropout(case):
    """the reference in training mode with its nn.Dropout modules applying the counter-based mask: the oracle
    regenerates the mask from the seed (same scheme as rc_sasrec_batch_fwd_dropout)"""
    g = load_golden(case)
    P = params(g)
    n_layers, n_heads = int(g["meta"][2]), int(g["meta"][3])
    B, L = g["hist"].shape
    p = float(g["p"])
    drop = SO.dropout_keep(int(g["mask_seed"]), g["len"], L, P["i_embeddings.weight"].shape[1], n_layers, p)
    assert len(drop) == 2 * n_layers
    valid = np.arange(L)[None, :] < g["len"][:, None]
    for m in drop:  # the drop rate on valid positions is p within sampling noise, kept values carry 1 / (1 - p)
        vals = m[valid]
        assert set(np.unique(vals)) <= {np.float32(0), np.float32(1) / (np.float32(1) - np.float32(p))}
        assert abs((vals == 0).mean() - p) < 4 * np.sqrt(p * (1 - p) / vals.size) + 1e-3
    assert not np.array_equal(drop[0], drop[1])
    pred = SO.forward(P, g["hist"], g["len"], g["iid"], n_heads, drop=drop)
    assert_close(pred, g["pred"], what="pred", atol_scale=2e-5)
    assert np.abs(pred - SO.forward(P, g["hist"], g["len"], g["iid"], n_heads)).max() > 1e-3  # the mask matters
    _, G = SO.backward(P, g["hist"], g["len"], g["iid"], n_heads, g["gpred"], drop=drop)
    want = params(g, "G/")
    floor = 1e-6 * max(float(np.abs(v).max()) for v in want.values())
    for k, v in want.items():
        assert_close(G[k], v, what="grad " + k, rtol=2e-5, atol_scale=5e-5, abs_floor=floor)
