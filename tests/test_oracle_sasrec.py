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
