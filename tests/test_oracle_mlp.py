"""CPU: the dense-layer oracle (oracle/mlp_oracle.py) vs torch's own nn.Linear -> ReLU -> mask arithmetic and autograd
(what utils/layers.py:201-243 MLP_Block and models/general/NeuMF.py:69-72 execute in the reference), the counter-based
dropout mask, and the host-side detection of the module chains the kernels cover (rechorus_amd.nn.mlp_plan)."""
import numpy as np
import pytest
import torch

from conftest import assert_close
from oracle import mlp_oracle as MO


@pytest.mark.parametrize("M,N,K", [(5, 3, 7), (64, 32, 16), (33, 65, 40)])
@pytest.mark.parametrize("relu,p", [(False, 0.0), (True, 0.0), (True, 0.4)])
def test_oracle_layer_matches_torch_autograd(M, N, K, relu, p):
    rng = np.random.default_rng(M * N + K)
    X = rng.normal(0, 1, (M, K)).astype(np.float32)
    W = rng.normal(0, 0.3, (N, K)).astype(np.float32)
    b = rng.normal(0, 0.3, N).astype(np.float32)
    dY = rng.normal(0, 1, (M, N)).astype(np.float32)
    keep = MO.dropout_keep(4242, 1, M, N, p) if p > 0 else None
    y, cache = MO.linear_fwd(X, W, b, relu, keep)
    dX, dW, db = MO.linear_bwd(X, W, cache, dY)
    xt = torch.from_numpy(X).double().requires_grad_(True)
    lin = torch.nn.Linear(K, N).double()
    with torch.no_grad():
        lin.weight.copy_(torch.from_numpy(W))
        lin.bias.copy_(torch.from_numpy(b))
    yt = lin(xt)
    if relu:
        yt = yt.relu()
    if keep is not None:
        yt = yt * torch.from_numpy(keep).double()   # nn.Dropout with this mask: zero or scale by 1 / (1 - p)
    yt.backward(torch.from_numpy(dY).double())
    assert_close(y, yt.detach().numpy(), what="y", rtol=1e-6, abs_floor=1e-6)
    assert_close(dX, xt.grad.numpy(), what="dX", rtol=1e-6, abs_floor=1e-6)
    assert_close(dW, lin.weight.grad.numpy(), what="dW", rtol=1e-6, abs_floor=1e-5)
    assert_close(db, lin.bias.grad.numpy(), what="db", rtol=1e-6, abs_floor=1e-5)


@pytest.mark.parametrize("p", [0.0, 0.3])
def test_chain_backward_with_the_mask_in_the_dx_product_equals_layer_by_layer(p):
    """the algebra behind rc_linear_bwd_chain (one autograd node per MLP): masking dX of layer i + 1 with layer i's saved
    output (X > 0, scale 1 / (1 - p)) and handing it down as that layer's dZ gives the gradients of the layer-by-layer backward,
    where every layer masks its own dY -- the saved output of drop(relu(.)) is its own mask (dropped or clipped elements are 0)"""
    rng = np.random.default_rng(7)
    M, dims = 37, (20, 48, 16, 3)
    Ws = [rng.normal(0, 0.4, (b, a)).astype(np.float32) for a, b in zip(dims[:-1], dims[1:])]
    bs = [rng.normal(0, 0.2, b).astype(np.float32) for b in dims[1:]]
    X0 = rng.normal(0, 1, (M, dims[0])).astype(np.float32)
    n = len(Ws)
    relu = [True] * (n - 1) + [False]
    xs, caches, h = [X0], [], X0
    for i in range(n):
        keep = MO.dropout_keep(99, i, M, dims[i + 1], p) if (p > 0 and relu[i]) else None
        h, c = MO.linear_fwd(h, Ws[i], bs[i], relu[i], keep)
        xs.append(h)
        caches.append(c)
    dY = rng.normal(0, 1, (M, dims[-1])).astype(np.float32)
    # layer by layer: every layer masks its own dY
    want, d = [], dY
    for i in range(n - 1, -1, -1):
        d, dW, db = MO.linear_bwd(xs[i], Ws[i], caches[i], d)
        want.append((dW, db))
    want_dx0 = d
    # chain: the top layer has no activation (dZ = dY); each dX product applies the mask of the layer below
    got, dz = [], dY
    for i in range(n - 1, -1, -1):
        below_act = i > 0 and relu[i - 1]
        dz, dW, db = MO.linear_bwd_chain(xs[i], Ws[i], dz, x_mask=(xs[i] > 0) if below_act else None,
                                         x_scale=1.0 / (1.0 - p) if below_act else 1.0)
        got.append((dW, db))
    for (a, b), (c, e) in zip(got, want):
        assert_close(a, c, what="dW", rtol=1e-6, abs_floor=1e-6)
        assert_close(b, e, what="db", rtol=1e-6, abs_floor=1e-6)
    assert_close(dz, want_dx0, what="dX of the first layer", rtol=1e-6, abs_floor=1e-6)


def test_dropout_mask_is_counter_based():
    a = MO.dropout_keep(7, 2, 37, 20, 0.25)
    assert a.shape == (37, 20) and set(np.unique(a)) <= {np.float32(0), np.float32(1) / (np.float32(1) - np.float32(0.25))}
    assert np.array_equal(a, MO.dropout_keep(7, 2, 37, 20, 0.25))                 # a function of (seed, site, m, n) only
    assert np.array_equal(a[:16], MO.dropout_keep(7, 2, 16, 20, 0.25))            # independent of the batch size
    assert not np.array_equal(a, MO.dropout_keep(8, 2, 37, 20, 0.25)) and not np.array_equal(a, MO.dropout_keep(7, 3, 37, 20, 0.25))
    big = MO.dropout_keep(11, 0, 400, 256, 0.3)
    assert abs((big == 0).mean() - 0.3) < 0.01


def test_mlp_plan_detects_the_chains_the_kernels_cover():
    import torch.nn as nn
    from rechorus_amd import nn as hnn
    l1, l2, l3 = nn.Linear(8, 16), nn.Linear(16, 4), nn.Linear(4, 1)
    plan = hnn.mlp_plan([l1, nn.ReLU(), nn.Dropout(0.2), l2, nn.ReLU(), l3])
    assert [(m is l, r, p) for (m, r, p), l in zip(plan, (l1, l2, l3))] == [(True, True, 0.2), (True, True, 0.0), (True, False, 0.0)]
    assert hnn.mlp_plan([l1]) is not None and hnn.mlp_plan([]) is None
    for bad in ([l1, nn.BatchNorm1d(16), nn.ReLU()], [l1, nn.LayerNorm(16)], [l1, nn.Sigmoid()], [nn.ReLU(), l1],
                [l1, nn.Dropout(0.5)]):   # dropout without ReLU: the saved output would not be its own mask
        assert hnn.mlp_plan(bad) is None
    with pytest.raises(RuntimeError):
        hnn.linear(torch.zeros(2, 8), l1.weight, l1.bias)   # CPU tensors: no fallback
