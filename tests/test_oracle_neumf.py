"""CPU: the numpy NeuMF oracle vs the reference's own outputs (tests/golden/neumf_*.npz)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN_DIR, assert_close, load_golden
from oracle import bprmf_oracle as BO
from oracle import neumf_oracle as NO

CASES = sorted(f[:-4] for f in os.listdir(GOLDEN_DIR) if f.startswith("neumf_") and f.endswith(".npz"))


def params(g, prefix="P0/"):
    return {k[len(prefix):]: v for k, v in g.items() if k.startswith(prefix)}


@pytest.mark.parametrize("case", CASES)
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
