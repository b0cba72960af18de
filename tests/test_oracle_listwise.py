"""CPU: list-wise softmax CE oracle vs the reference's own loss / autograd (tests/golden/listwise_*.npz)."""
import pytest

from conftest import assert_close, golden_cases, load_golden
from oracle import listwise_oracle as LO


@pytest.mark.parametrize("case", golden_cases("listwise_"))
def test_softmax_ce_oracle(case):
    g = load_golden(case)
    P = int(g["max_pos"])
    loss, _, _ = LO.softmax_ce(g["pred"], g["target"], P)
    assert_close(loss, g["loss"], what="loss")
    assert_close(LO.softmax_ce_grad(g["pred"], g["target"], P), g["gpred"], what="gpred", atol_scale=2e-5)
