"""GPU parity of the tower-tail kernels (csrc/tower_tail.hip: the last hidden layer + the width -> 1 output layer of an MLP tower as one
forward and one backward kernel) against oracle/mlp_oracle.py (float64 restatement of utils/layers.py:201-243 and nn.Linear's
autograd), and of the autograd node that uses them (rechorus_amd.nn._MlpFn) against the layer-by-layer GEMM route it replaces at
small batches -- same dropout masks, same values to rounding."""
import numpy as np
import pytest
import torch

from conftest import assert_close
from oracle import mlp_oracle as MO

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng(cuda):
    from rechorus_amd import engine
    return engine


def _seed(cuda, v):
    return torch.tensor([v], dtype=torch.int64, device=cuda)


@pytest.mark.parametrize("M,K,N2,p2,x_act,p_below,bias", [
    (1024, 512, 64, 0.2, True, 0.2, True),      # the DeepFM tower of docs/demo_scripts_results/CTR_MIND.sh:8 at its batch size
    (1000, 512, 64, 0.0, True, 0.0, True),      # ragged last block
    (37, 128, 32, 0.5, False, 0.0, True),       # the tower's input is not an activation (--layers [64]: no mask on dX)
    (1, 64, 16, 0.0, True, 0.3, False),
    (4100, 256, 64, 0.3, True, 0.1, False),     # more row blocks than workgroups: several blocks per partial
    (513, 512, 16, 0.1, False, 0.0, True),
])
def test_tower_tail_kernels_vs_oracle(M, K, N2, p2, x_act, p_below, bias, cuda, eng):
    rng = np.random.default_rng(M + K + N2)
    X = rng.normal(0, 0.5, (M, K)).astype(np.float32)
    if x_act:     # a drop(relu(.)) activation: zeros where the layer below was inactive / dropped
        X = np.where(rng.random((M, K)) < 0.45, 0.0, np.abs(X)).astype(np.float32)
    W2 = rng.normal(0, 0.1, (N2, K)).astype(np.float32)
    b2 = rng.normal(0, 0.1, N2).astype(np.float32) if bias else None
    W3 = rng.normal(0, 0.3, (1, N2)).astype(np.float32)
    b3 = rng.normal(0, 0.1, 1).astype(np.float32) if bias else None
    dz = rng.normal(0, 1.0, (M, 1)).astype(np.float32)
    sv, site = 123456789 + (M << 33), 1
    keep = MO.dropout_keep(sv, site, M, N2, p2) if p2 > 0 else None
    H2o, c2 = MO.linear_fwd(X, W2, b2, True, keep)
    zo, _ = MO.linear_fwd(H2o, W3, b3, False)
    dZ2o, dW3o, db3o = MO.linear_bwd_chain(H2o, W3, dz, x_mask=H2o > 0, x_scale=1.0 / (1.0 - p2))
    dXo, dW2o, db2o = MO.linear_bwd_chain(X, W2, dZ2o, x_mask=(X > 0) if x_act else None, x_scale=1.0 / (1.0 - p_below))
    t = lambda a: None if a is None else torch.from_numpy(a).to(cuda)
    assert eng.tower_tail_supported(M, K, N2)
    seed = _seed(cuda, sv) if p2 > 0 else None
    H2, z = eng.tower_tail_fwd(t(X), t(W2), t(b2), t(W3), t(b3), p2, seed, site)
    if keep is not None:
        assert np.array_equal(H2.cpu().numpy() == 0, (H2o == 0)), "dropout mask / ReLU pattern differs from the oracle's"
    assert_close(H2.cpu().numpy(), H2o, what="H2", rtol=2e-5, atol_scale=2e-5)
    assert_close(z.cpu().numpy(), zo, what="z", rtol=2e-5, atol_scale=2e-5)
    dX, dW2, db2, dW3, db3 = eng.tower_tail_bwd(t(X), t(W2), t(W3), t(H2o), t(dz), p2, need_dx=True, x_act=x_act, x_drop_p=p_below,
                                                need_db2=bias, need_db3=bias)
    assert_close(dX.cpu().numpy(), dXo, what="dX", rtol=2e-5, atol_scale=3e-5)
    assert_close(dW2.cpu().numpy(), dW2o, what="dW2", rtol=2e-5, atol_scale=3e-5)
    assert_close(dW3.cpu().numpy(), dW3o, what="dW3", rtol=2e-5, atol_scale=3e-5)
    if bias:
        assert_close(db2.cpu().numpy(), db2o, what="db2", rtol=2e-5, atol_scale=3e-5)
        assert_close(db3.cpu().numpy(), db3o, what="db3", rtol=2e-5, atol_scale=3e-5)
    else:
        assert db2 is None and db3 is None
    # no dX wanted: the weight gradients are the same bits
    _, dW2b, _, dW3b, _ = eng.tower_tail_bwd(t(X), t(W2), t(W3), t(H2o), t(dz), p2, need_dx=False, need_db2=False, need_db3=False)
    assert torch.equal(dW2b, dW2) and torch.equal(dW3b, dW3)
    # deterministic
    dX2, dW2c, *_ = eng.tower_tail_bwd(t(X), t(W2), t(W3), t(H2o), t(dz), p2, need_dx=True, x_act=x_act, x_drop_p=p_below)
    assert torch.equal(dX2, dX) and torch.equal(dW2c, dW2)


def test_tower_tail_shape_envelope(cuda, eng):
    assert eng.tower_tail_supported(1024, 512, 64) and eng.tower_tail_supported(5, 64, 16)
    assert not eng.tower_tail_supported(1024, 512, 48) and not eng.tower_tail_supported(1024, 500, 64)
    assert not eng.tower_tail_supported(1024, 1024, 64) and not eng.tower_tail_supported(10 ** 6, 512, 64)
    X, W2, w3 = (torch.zeros(s, device=cuda) for s in ((8, 96), (64, 96), (1, 64)))
    with pytest.raises(Exception, match="not supported"):
        eng.tower_tail_fwd(X, W2, None, w3, None)


@pytest.mark.parametrize("widths,p,M", [([512, 512, 64], 0.2, 1024), ([512, 64], 0.0, 333), ([128, 256, 32], 0.5, 70), ([256, 16], 0.1, 2000)])
def test_mlp_autograd_node_with_the_tail_kernels_equals_the_layer_by_layer_route(widths, p, M, cuda, eng, monkeypatch):
    """rechorus_amd.nn.mlp_forward on Linear -> ReLU -> Dropout groups + Linear(width, 1): with the tail kernels and with the GEMM
    route (RC_TOWER_TAIL=0) -- same masks (same seed, same layer indices), outputs and every gradient equal to rounding"""
    from rechorus_amd import nn as hnn
    torch.manual_seed(3)
    mods = []
    for a, b in zip(widths[:-1], widths[1:]):
        mods += [torch.nn.Linear(a, b), torch.nn.ReLU()] + ([torch.nn.Dropout(p)] if p > 0 else [])
    mods.append(torch.nn.Linear(widths[-1], 1))
    seq = torch.nn.Sequential(*mods).to(cuda)
    plan = hnn.mlp_plan(list(seq))
    assert plan is not None and len(plan) == len(widths)
    x0 = torch.randn(M, widths[0], device=cuda)
    gy = torch.randn(M, 1, device=cuda)
    res = []
    for tail in (True, False):
        monkeypatch.setattr(eng, "_TOWER_TAIL", tail)
        assert hnn._MlpFn._tail(tuple((True, p, True) for _ in widths[1:]) + ((False, 0.0, True),), [q for l, _, _ in plan for q in (l.weight, l.bias)], M) == tail
        x = x0.clone().requires_grad_(True)
        seq.zero_grad()
        y = hnn.mlp_forward(x, plan, True, _seed(cuda, 991))
        (y * gy).sum().backward()
        res.append((y.detach().clone(), x.grad.clone(), [q.grad.clone() for q in seq.parameters()]))
    (ya, xa, ga), (yb, xb, gb) = res
    assert_close(ya.cpu().numpy(), yb.cpu().numpy(), what="output", rtol=2e-5, atol_scale=2e-5)
    assert_close(xa.cpu().numpy(), xb.cpu().numpy(), what="d input", rtol=2e-5, atol_scale=3e-5)
    if p > 0:
        assert torch.equal(xa == 0, xb == 0) or float(((xa == 0) != (xb == 0)).float().mean()) < 1e-4
    for (name, _), a, b in zip(seq.named_parameters(), ga, gb):
        assert_close(a.cpu().numpy(), b.cpu().numpy(), what="grad " + name, rtol=2e-5, atol_scale=3e-5)
