"""GPU: list-wise softmax CE kernel vs the reference's own loss / autograd and the oracle."""
import numpy as np
import pytest
import torch

from conftest import assert_close, golden_cases, load_golden
from oracle import listwise_oracle as LO

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case", golden_cases("listwise_"))
def test_softmax_ce_kernel(case, cuda):
    from rechorus_amd import engine
    g = load_golden(case)
    P = int(g["max_pos"])
    loss, gpred = engine.softmax_ce(torch.from_numpy(g["pred"]).to(cuda), torch.from_numpy(g["target"]).to(cuda), P)
    assert_close(loss.cpu().numpy()[0], g["loss"], what="loss")
    assert_close(gpred.cpu().numpy(), g["gpred"], what="gpred", atol_scale=2e-5)


def test_softmax_ce_wide_lists_and_saturation(cuda):
    from rechorus_amd import engine
    rng = np.random.default_rng(0)
    B, P, N = 37, 5, 300  # lists longer than a wave
    pred = rng.normal(0, 8.0, size=(B, P + N)).astype(np.float32)
    target = np.full((B, P + N), -1, dtype=np.int64)
    for b in range(B):
        target[b, :rng.integers(1, P + 1)] = 1
        target[b, P:P + rng.integers(1, N + 1)] = 0
    loss, gpred = engine.softmax_ce(torch.from_numpy(pred).to(cuda), torch.from_numpy(target).to(cuda), P)
    want, _, _ = LO.softmax_ce(pred, target, P)
    assert_close(loss.cpu().numpy()[0], want, what="loss")
    assert_close(gpred.cpu().numpy(), LO.softmax_ce_grad(pred, target, P), what="gpred", atol_scale=2e-5)
    assert np.all(gpred.cpu().numpy()[target == -1] == 0.0)
