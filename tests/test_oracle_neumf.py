"""CPU: the numpy NeuMF oracle vs the reference's own outputs (tests/golden/neumf_*.npz)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN_DIR, assert_close, load_golden
from oracle import bprmf_oracle as BO
from oracle import neumf_oracle as NO

CASES = sorted(f[:-4] for f in os.listdir(GOLDEN_DIR) if f.startswith("neumf_") and f.endswith(".npz"))
ML_CASES = sorted(f[:-4] for f in os.listdir(GOLDEN_DIR) if f.startswith("neumfml_") and f.endswith(".npz"))  # several hidden layers


def params(g, prefix="P0/"):
    return {k[len(prefix):]: v for k, v in g.items() if k.startswith(prefix)}


@pytest.mark.parametrize("case", CASES + ML_CASES)
def test_neumf_forward_loss_grads(case):
    g = load_golden(case)
    P = params(g)
    pred, _ = NO.forward(P, g["uid"], g["iid"])
    assert_close(pred, g["pred"], what="pred")
    assert_close(BO.bpr_loss(pred), g["loss"], what="loss")
    gp = BO.bpr_loss_grad(g["pred"])
    assert_close(gp, g["gpred"], what="gpred")
    _, G = NO.backward(P, g["uid"], g["iid"], g["gpred"])
    for k, v in params(g, "G/").items():
        assert_close(G[k], v, what="grad " + k, atol_scale=2e-5)


DROP_CASES = sorted(f[:-4] for f in os.listdir(GOLDEN_DIR) if f.startswith("neumfdrop_") and f.endswith(".npz"))


@pytest.mark.parametrize("case", DROP_CASES)
def test_neumf_dropout_vs_reference_with_given_mask(case):
    """training-mode head with hidden-layer dropout: the reference model run with the counter-based mask
    (tests/golden/make_golden_neumf.py: make_dropout_case) vs the oracle"""
    g = load_golden(case)
    P = params(g)
    B, C = g["iid"].shape
    keep = NO.dropout_keep(int(g["mask_seed"]), B * C, P["mlp.0.weight"].shape[0], float(g["p"]))
    assert np.float32((keep != 0).mean()) == g["keep_rate"]
    pred, _ = NO.forward(P, g["uid"], g["iid"], keep)
    assert_close(pred, g["pred"], what="pred")
    assert_close(BO.bpr_loss(pred), g["loss"], what="loss")
    _, G = NO.backward(P, g["uid"], g["iid"], g["gpred"], keep)
    for k, v in params(g, "G/").items():
        assert_close(G[k], v, what="grad " + k, atol_scale=2e-5)
    # and the mask matters: eval-mode prediction differs
    assert np.abs(NO.forward(P, g["uid"], g["iid"])[0] - g["pred"]).max() > 1e-3


def test_dropout_keep_statistics_and_determinism():
    for p in (0.1, 0.2, 0.5, 0.9):
        keep = NO.dropout_keep(77, 4096, 64, p)
        rate = (keep == 0).mean()
        assert abs(rate - p) < 4 * np.sqrt(p * (1 - p) / keep.size) + 1e-4
        kept = keep[keep != 0]
        assert np.all(kept == np.float32(1) / (np.float32(1) - np.float32(p)))
        # per-feature and per-candidate rates are uniform too (no stripe pattern from the 4-word blocks)
        assert np.abs((keep == 0).mean(0) - p).max() < 0.05 and np.abs((keep == 0).mean(1) - p).max() < 0.3
    a, b = NO.dropout_keep(5, 100, 32, 0.3), NO.dropout_keep(6, 100, 32, 0.3)
    assert np.array_equal(a, NO.dropout_keep(5, 100, 32, 0.3)) and (a != b).mean() > 0.2
    # a prefix of the candidates sees the same mask whatever the batch size (counter = candidate index)
    assert np.array_equal(NO.dropout_keep(5, 40, 32, 0.3), a[:40])
