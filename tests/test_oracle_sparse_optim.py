"""CPU: external pin of the ROW-WISE ("lazy") optimizer semantics (hard part 1b).

The reference only ever builds dense torch.optim optimizers (helpers/BaseRunner.py:110-114), so its own runs cannot
pin the engine's large-table mode, where only rows present in the batch are stepped.  PyTorch's own sparse path can:
nn.Embedding(sparse=True) + torch.optim.SparseAdam / torch.optim.Adagrad step exactly the touched rows, coalescing
duplicate ids, with the global step count in the bias corrections.  The numpy oracle's row-wise step
(oracle/bprmf_oracle.opt_step_dense(rows=...), which the HIP kernels are held to in tests/test_gpu_bprmf.py) must agree.

One documented difference: SparseAdam adds eps to sqrt(v) BEFORE the bias correction (denominator
sqrt(v) + eps, step lr*sqrt(bc2)/bc1), dense Adam -- whose formula the engine applies to the touched rows -- after
(sqrt(v)/sqrt(bc2) + eps).  With gradients of order 1 the two differ by ~eps/|g| ~ 1e-7 relative; the test uses
such gradients and a 2e-6 tolerance."""
import numpy as np
import pytest
import torch

from oracle import bprmf_oracle as O


def _run_torch(opt_name, W0, steps, lr):
    emb = torch.nn.Embedding(W0.shape[0], W0.shape[1], sparse=True)
    with torch.no_grad():
        emb.weight.copy_(torch.from_numpy(W0))
    if opt_name == "Adam":
        opt = torch.optim.SparseAdam(emb.parameters(), lr=lr)
    else:
        opt = torch.optim.Adagrad(emb.parameters(), lr=lr)
    out = []
    for ids, coef in steps:
        opt.zero_grad()
        (emb(torch.from_numpy(ids)) * torch.from_numpy(coef)).sum().backward()
        assert emb.weight.grad.is_sparse
        opt.step()
        out.append(emb.weight.detach().numpy().copy())
    return out


@pytest.mark.parametrize("opt_name,lr", [("Adam", 1e-2), ("Adagrad", 5e-2)])
def test_rowwise_oracle_matches_torch_sparse_optimizers(opt_name, lr):
    rng = np.random.default_rng(5)
    n_rows, d = 50, 8
    W0 = rng.normal(0, 0.5, size=(n_rows, d)).astype(np.float32)
    steps = []
    for s in range(4):
        # duplicates inside a step; rows 0-9 are touched in steps 0 and 2 only (stale moments in between)
        lo, hi = (0, 30) if s % 2 == 0 else (10, 50)
        ids = rng.integers(lo, hi, size=80).astype(np.int64)
        coef = rng.normal(0, 1.0, size=(80, d)).astype(np.float32)
        steps.append((ids, coef))
    want = _run_torch(opt_name, W0, steps, lr)
    W = W0.copy()
    state = O.new_state(W, opt_name)
    for s, (ids, coef) in enumerate(steps, 1):
        G = O.embedding_dense_backward(coef, ids, n_rows)
        before = W.copy()
        O.opt_step_dense(W, G, state, opt_name, lr, 0.0, step=s, rows=np.unique(ids))
        untouched = np.setdiff1d(np.arange(n_rows), ids)
        assert np.array_equal(W[untouched], before[untouched])
        np.testing.assert_allclose(W, want[s - 1], rtol=2e-6, atol=2e-7, err_msg=f"{opt_name} step {s}")
