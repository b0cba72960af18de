"""GPU: the fp32 MFMA dense-layer kernels (rc_linear_fwd / rc_linear_bwd, csrc/mlp.hip) vs a float64 numpy evaluation of
nn.Linear -> ReLU -> Dropout and its autograd (utils/layers.py:201-243, models/general/NeuMF.py:69-72), the dropout mask
regenerated from the counter-based stream (oracle/mlp_oracle.py); torch's own fp32 modules as a second reference."""
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


SHAPES = [(1, 1, 1), (3, 5, 7), (64, 64, 16), (65, 63, 17), (100, 1, 96), (257, 130, 66), (1024, 512, 512), (777, 64, 512),
          (500, 33, 128), (2048, 32, 64),
          # 128 x 128 tiles (mlp_gemm_big_kernel: from 2048 workgroups) in the forward / the dX product, ragged edges, odd K
          (66000, 512, 96), (66000, 64, 512), (140000, 130, 127),
          # the weight-gradient product on 128 x 128 tiles as well (batch >= 16,384 rows, a multiple of 32; N, K multiples of 4)
          (16480, 132, 128), (32768, 256, 192),
          # one to four outputs over a large batch: the weight gradient as a weighted column sum (mlp_dw_narrow_kernel)
          (20000, 1, 64), (9000, 3, 128), (70000, 4, 512), (5000, 2, 20)]


@pytest.mark.parametrize("M,N,K", SHAPES)
@pytest.mark.parametrize("relu,p", [(False, 0.0), (True, 0.0), (True, 0.3)])
def test_linear_forward_backward(M, N, K, relu, p, cuda, eng):
    rng = np.random.default_rng(M + 3 * N + 7 * K)
    X = rng.normal(0, 1, (M, K)).astype(np.float32)
    W = (rng.normal(0, 1, (N, K)) / np.sqrt(K)).astype(np.float32)
    b = rng.normal(0, 0.5, N).astype(np.float32)
    dY = rng.normal(0, 1, (M, N)).astype(np.float32)
    seed_val, site = 123456789 + M, 3
    seed = torch.tensor([seed_val], dtype=torch.int64, device=cuda)
    t = lambda a: torch.from_numpy(a).to(cuda)
    Y = eng.linear_fwd(t(X), t(W), t(b), relu=relu, drop_p=p, seed=seed if p > 0 else None, site=site)
    keep = MO.dropout_keep(seed_val, site, M, N, p) if p > 0 else None
    want, cache = MO.linear_fwd(X, W, b, relu, keep)
    scale = float(np.abs(want).max()) + 1e-6
    assert_close(Y.cpu().numpy(), want, what="Y", rtol=2e-5, abs_floor=2e-6 * scale)
    if p > 0:
        kept = (keep > 0).mean()
        assert abs(kept - (1 - p)) < 4 * np.sqrt(p * (1 - p) / keep.size) + 1e-3
    dX, dW, db = eng.linear_bwd(t(X), t(W), Y if (relu or p > 0) else None, t(dY), drop_p=p)
    if relu:
        # a pre-activation within fp32 rounding of zero has no sign the two evaluations must agree on (about one element in 10^7,
        # i.e. one or two at the largest shapes, and each flips a whole row of dX): the backward is checked against the mask the
        # kernel's own forward produced, which may differ from the float64 one in a handful of places only
        mask_gpu = Y.cpu().numpy() > 0
        if keep is not None:
            mask_gpu = np.where(keep > 0, mask_gpu, cache["mask"])
        assert int((mask_gpu != cache["mask"]).sum()) <= 2 + M * N // 2_000_000
        cache = dict(cache, mask=mask_gpu)
    wX, wW, wb = MO.linear_bwd(X, W, cache, dY)
    for name, got, ref in (("dX", dX, wX), ("dW", dW, wW), ("db", db, wb)):
        sc = float(np.abs(ref).max()) + 1e-6
        assert_close(got.cpu().numpy(), ref, what=name, rtol=2e-5, abs_floor=4e-6 * sc)
    # optional outputs
    dX2, dW2, db2 = eng.linear_bwd(t(X), t(W), Y if (relu or p > 0) else None, t(dY), drop_p=p, need_dx=False, need_db=False)
    assert dX2 is None and db2 is None and torch.equal(dW2, dW)


def test_linear_is_deterministic_and_matches_torch_modules(cuda, eng):
    from rechorus_amd import nn as hnn
    torch.manual_seed(0)
    lin1, lin2 = torch.nn.Linear(96, 512).to(cuda), torch.nn.Linear(512, 64).to(cuda)
    x = torch.randn(300, 96, device=cuda, requires_grad=True)
    plan = hnn.mlp_plan([lin1, torch.nn.ReLU(), lin2, torch.nn.ReLU()])
    assert plan is not None and len(plan) == 2
    y = hnn.mlp_forward(x, plan, training=False, seed=None)
    y.square().sum().backward()
    g1 = [p.grad.clone() for p in (x, lin1.weight, lin1.bias, lin2.weight, lin2.bias)]
    for p_ in (x, lin1.weight, lin1.bias, lin2.weight, lin2.bias):
        p_.grad = None
    y2 = hnn.mlp_forward(x, plan, training=False, seed=None)
    y2.square().sum().backward()
    g2 = [p.grad.clone() for p in (x, lin1.weight, lin1.bias, lin2.weight, lin2.bias)]
    assert torch.equal(y, y2) and all(torch.equal(a, b) for a, b in zip(g1, g2))   # fixed summation orders
    for p_ in (x, lin1.weight, lin1.bias, lin2.weight, lin2.bias):
        p_.grad = None
    yt = lin2(lin1(x).relu()).relu()
    yt.square().sum().backward()
    g3 = [p.grad for p in (x, lin1.weight, lin1.bias, lin2.weight, lin2.bias)]
    assert_close(y.detach().cpu().numpy(), yt.detach().cpu().numpy(), what="y vs torch", rtol=1e-4, abs_floor=1e-5)
    for a, b in zip(g1, g3):
        assert_close(a.cpu().numpy(), b.cpu().numpy(), what="grad vs torch", rtol=2e-4, abs_floor=1e-4 * float(b.abs().max()))
    # chains the kernels do not cover stay on torch
    assert hnn.mlp_plan([lin1, torch.nn.BatchNorm1d(512), torch.nn.ReLU()]) is None
    assert hnn.mlp_plan([lin1, torch.nn.Dropout(0.2)]) is None


@pytest.mark.parametrize("M,widths,p", [(300, (96, 512, 64, 1), 0.0), (1024, (512, 512, 64), 0.3), (66000, (64, 512, 130), 0.0),
                                        (20000, (128, 256, 128, 8), 0.25)])
def test_mlp_chain_with_fused_masks_equals_layer_by_layer(M, widths, p, cuda, eng, monkeypatch):
    """the whole-MLP autograd node (rc_linear_bwd_chain: the dX product of layer i + 1 applies layer i's ReLU / dropout mask in its
    epilogue) against one autograd node per layer with the separate masking pass (rc_linear_bwd): same products, same mask
    arithmetic -- every gradient bit-identical; small and 128 x 128-tile shapes, split-K shapes, with and without dropout"""
    from rechorus_amd import nn as hnn
    # (the products compared bit for bit are the GEMM route's: a tower that ends hidden <= 64 -> 1 at a small batch takes the
    #  tower-tail kernels instead, held to the same reference in tests/test_gpu_tower_tail.py)
    monkeypatch.setattr(eng, "_TOWER_TAIL", False)
    torch.manual_seed(M)
    lins = [torch.nn.Linear(a, b).to(cuda) for a, b in zip(widths[:-1], widths[1:])]
    mods = []
    for k, lin in enumerate(lins):
        mods.append(lin)
        if k + 1 < len(lins):
            mods.append(torch.nn.ReLU())
            if p > 0:
                mods.append(torch.nn.Dropout(p))
    plan = hnn.mlp_plan(mods)
    assert plan is not None and len(plan) == len(lins)
    x = torch.randn(M, widths[0], device=cuda, requires_grad=True)
    seed = torch.tensor([77], dtype=torch.int64, device=cuda) if p > 0 else None
    params = [x] + [q for lin in lins for q in (lin.weight, lin.bias)]

    def grads_of(y):
        for q in params:
            q.grad = None
        (y * torch.linspace(-1, 1, y.shape[-1], device=cuda)).sum().backward()
        return [q.grad.clone() for q in params]
    y1 = hnn.mlp_forward(x, plan, training=p > 0, seed=seed)
    g1 = grads_of(y1)
    h = x
    for site, (lin, relu, pp) in enumerate(plan):
        h = hnn.linear(h, lin.weight, lin.bias, relu, pp, seed, site)
    g2 = grads_of(h)
    assert torch.equal(y1, h)
    for a, b, name in zip(g1, g2, ["x"] + [f"{n}{k}" for k in range(len(lins)) for n in ("W", "b")]):
        assert torch.equal(a, b), name


def test_mlp_block_runs_on_the_engine_and_draws_fresh_masks(cuda):
    import os
    import sys
    from conftest import ROOT
    plugin = os.path.join(ROOT, "rechorus_amd", "rechorus")
    if plugin not in sys.path:
        sys.path.insert(0, plugin)
    from utils import layers
    from rechorus_amd import nn as hnn
    blk = layers.MLP_Block(40, hidden_units=[64, 32], hidden_activations="ReLU", dropout_rates=0.5, output_dim=1).to(cuda)
    assert blk._hip_plan is not None and "drop_seed" not in blk.state_dict()
    assert list(blk.state_dict()) == ["mlp.0.weight", "mlp.0.bias", "mlp.3.weight", "mlp.3.bias", "mlp.6.weight", "mlp.6.bias"]
    x = torch.randn(200, 40, device=cuda)
    calls = []
    real, real_tail = hnn.engine.linear_fwd, hnn.engine.tower_tail_fwd
    hnn.engine.linear_fwd = lambda *a, **k: calls.append(1) or real(*a, **k)
    hnn.engine.tower_tail_fwd = lambda *a, **k: calls.append(2) or real_tail(*a, **k)   # (the last hidden layer + the output layer)
    try:
        blk.train()
        a, b = blk(x), blk(x)
        blk.eval()
        e1, e2 = blk(x), blk(x)
    finally:
        hnn.engine.linear_fwd, hnn.engine.tower_tail_fwd = real, real_tail
    assert sum(calls) == 12 and calls.count(2) == 4 and not torch.equal(a, b) and torch.equal(e1, e2)
    want = blk.mlp(x)  # torch's modules in eval mode: same parameters
    assert_close(e1.detach().cpu().numpy(), want.detach().cpu().numpy(), what="eval vs torch modules", rtol=1e-4, abs_floor=1e-5)
