"""CPU: the host side of HipOptimizer's rows mode (rechorus_amd/nn.py) with the two engine calls it makes replaced by plain torch
arithmetic: which tensors, slices, flags, hyper-parameters and step numbers reach rc_dense_update_rows_dev, and that the protocol
(rows_begin -> the gather's stamps -> the backward's row sums in an uncleared scratch -> step) reproduces torch.optim.Adam on
dense gradients (helpers/BaseRunner.py:110-114,206 over the tables of models/context/FM.py:33-41).  The kernels themselves are
held to the dense step bit for bit on the GPU (tests/test_gpu_rows_adam.py)."""
import numpy as np
import pytest
import torch


def _fake_engine(monkeypatch, calls):
    from rechorus_amd import engine

    def step_increment(counter):
        counter += 1

    def dense_update_rows(items, step_dev, touched=2, max_blocks=0):
        step = int(step_dev.item())
        calls.append(("rows", touched, step, len(items)))
        assert touched == 2
        for W, G, h, m, v, flags, row_w in items:
            if flags is None:
                g = G.clone()
            else:
                assert flags.dtype == torch.int32 and flags.numel() == W.shape[0] and row_w == W.shape[1]
                hit = flags == step
                g = torch.where(hit[:, None], G, torch.zeros_like(G))     # (G holds garbage outside the stamped rows)
            g = g + float(h.l2) * W
            m.mul_(h.beta1).add_(g, alpha=1 - h.beta1)
            v.mul_(h.beta2).addcmul_(g, g, value=1 - h.beta2)
            bc1, bc2 = 1 - h.beta1 ** step, 1 - h.beta2 ** step
            W.sub_((h.lr / bc1) * m / (v.sqrt() / np.sqrt(bc2) + h.eps))

    monkeypatch.setattr(engine, "step_increment", step_increment)
    monkeypatch.setattr(engine, "dense_update_rows", dense_update_rows)
    monkeypatch.setattr(torch.cuda, "is_current_stream_capturing", lambda: False)


def test_rows_mode_protocol_reproduces_dense_adam(monkeypatch):
    from rechorus_amd import nn as hnn
    calls = []
    _fake_engine(monkeypatch, calls)
    monkeypatch.setenv("RC_ROWS_ADAM", "1")
    g = torch.Generator().manual_seed(0)
    vocab, d = [7, 5, 11], 8
    mk = lambda *s: torch.nn.Parameter(torch.randn(*s, generator=g) * 0.1)
    tables, tables1 = [mk(v, d) for v in vocab], [mk(v, 1) for v in vocab]
    W, b = mk(4, 6), mk(4)
    params = tables + tables1 + [W, b]
    ref = [torch.nn.Parameter(p.detach().clone()) for p in params]
    groups = lambda ps: [{"params": ps[:-1], "weight_decay": 1e-3}, {"params": ps[-1:], "weight_decay": 0.0}]
    opt = hnn.HipOptimizer(groups(params), name="Adam", lr=1e-2, weight_decay=1e-3, capturable=True)
    want = torch.optim.Adam(groups(ref), lr=1e-2)
    assert opt.rows_ok()
    offs = np.concatenate([[0], np.cumsum(vocab)])
    n_rows = int(offs[-1])
    for step in range(1, 5):
        ids = [torch.randint(0, v, (6,), generator=g) for v in vocab]
        mark = opt.rows_begin(tables, tables1)
        assert mark is not None
        flags, step_dev = mark
        assert int(step_dev.item()) == step - 1 and flags.numel() == n_rows
        for f, x in enumerate(ids):                       # what rc_gather_fields_pair_mark does
            flags[offs[f] + x] = int(step_dev.item()) + 1
        # the backward pass: row sums of the touched rows into the scratch, everything else left as it was (here: NaN)
        scratch = opt.rows_scratch()
        scratch.fill_(float("nan"))
        Gv, Gl = scratch[:n_rows * d].view(n_rows, d), scratch[n_rows * d:].view(n_rows, 1)
        dense = [torch.zeros_like(p) for p in ref]
        for f, x in enumerate(ids):
            gv, gl = torch.randn(6, d, generator=g), torch.randn(6, 1, generator=g)
            dense[f].index_add_(0, x, gv)
            dense[len(vocab) + f].index_add_(0, x, gl)
            rows = torch.unique(x)
            Gv[offs[f] + rows] = dense[f][rows]
            Gl[offs[f] + rows] = dense[len(vocab) + f][rows]
        opt.rows_grads(Gv, Gl)
        for p, q, k in ((W, ref[-2], -2), (b, ref[-1], -1)):
            p.grad = torch.randn(p.shape, generator=g)
            dense[k] = p.grad.clone()
        for q, gr in zip(ref, dense):
            q.grad = gr
        opt.step()
        want.step()
        assert opt._rows is None and int(step_dev.item()) == step
        assert calls[-1] == ("rows", 2, step, len(params))
        assert all(t.grad is None for t in tables + tables1)
        for p, q in zip(params, ref):
            np.testing.assert_allclose(p.detach().numpy(), q.detach().numpy(), rtol=2e-5, atol=2e-7)
    # the no-decay group kept its own hyper-parameters (the bias moved like torch's)
    assert not torch.equal(b.detach(), ref[-1].detach() * 0)


def test_rows_mode_refusals_and_abort(monkeypatch):
    from rechorus_amd import nn as hnn
    calls = []
    _fake_engine(monkeypatch, calls)
    t = [torch.nn.Parameter(torch.zeros(5, 4))]
    t1 = [torch.nn.Parameter(torch.zeros(5, 1))]
    other = torch.nn.Parameter(torch.zeros(3))
    assert hnn.HipOptimizer([{"params": t + t1}], name="SGD", capturable=True).rows_begin(t, t1) is None          # Adam only
    assert hnn.HipOptimizer([{"params": t + t1}], name="Adam", capturable=False).rows_begin(t, t1) is None       # device step count only
    assert hnn.HipOptimizer([{"params": t}], name="Adam", capturable=True).rows_begin(t, t1) is None             # a table it does not own
    monkeypatch.setenv("RC_ROWS_ADAM", "0")
    assert hnn.HipOptimizer([{"params": t + t1}], name="Adam", capturable=True).rows_begin(t, t1) is None
    monkeypatch.setenv("RC_ROWS_ADAM", "1")
    opt = hnn.HipOptimizer([{"params": t + t1 + [other]}], name="Adam", capturable=True)
    flags, step_dev = opt.rows_begin(t, t1)
    with pytest.raises(RuntimeError, match="not followed by backward"):
        opt.rows_begin(t, t1)
    count = opt.step_count
    with pytest.raises(RuntimeError, match="without the backward pass"):
        opt.step()
    assert opt.step_count == count       # a refused step is not counted
    flags[2] = 1
    opt.rows_abort()
    assert opt._rows is None and int(flags.max().item()) == 0
    # a table of the rows-mode gather that ALSO received a dense gradient is a protocol error, not a silent double update
    flags, step_dev = opt.rows_begin(t, t1)
    opt.rows_grads(opt.rows_scratch()[:20].view(5, 4), opt.rows_scratch()[20:].view(5, 1))
    t[0].grad = torch.zeros(5, 4)
    count = opt.step_count
    with pytest.raises(RuntimeError, match="also received a dense gradient"):
        opt.step()
    assert opt.step_count == count


def test_a_towers_seed_increment_takes_adams_step_count_along(monkeypatch):
    """engine.defer_increment / step_increment / take_deferred: between the gather of a rows-mode step (which reads count + 1) and the
    update launch nothing reads Adam's device step count, so the next increment of ANOTHER counter (a tower's dropout seed) bumps both
    in one launch (rc_step_increment2); without such an increment the optimizer bumps its own; an aborted step gives the count back"""
    from rechorus_amd import _lib, engine, nn as hnn
    launches = []

    def call(name, *a):
        launches.append(name)
    monkeypatch.setattr(_lib, "call", call)
    monkeypatch.setattr(engine, "_ptr", lambda t, *a, **k: t)
    monkeypatch.setattr(engine, "_stream", lambda: None)
    seed, count, other = torch.zeros(1, dtype=torch.int64), torch.zeros(1, dtype=torch.int64), torch.zeros(1, dtype=torch.int64)
    engine.defer_increment(count)
    engine.step_increment(seed)                     # folds the promised increment in
    assert launches == ["rc_step_increment2"]
    engine.step_increment(other)                    # a later increment is on its own again
    assert launches[-1] == "rc_step_increment" and engine.take_deferred(count) and not engine.take_deferred(count)
    engine.defer_increment(count)
    assert not engine.take_deferred(count)          # nobody came by: the owner increments itself
    engine.defer_increment(count)
    engine.step_increment(count)                    # the owner's own counter is never "another counter"
    assert launches[-1] == "rc_step_increment" and not engine.take_deferred(count)
