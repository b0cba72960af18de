"""HipOptimizer's rows mode (rc_gather_fields_pair_mark + rc_dense_update_rows_dev): torch.optim.Adam over the dense gradients of
embedding tables (helpers/BaseRunner.py:110-114,206; the tables of models/context/FM.py:33-41) without the dense gradient: the
rows the batch touched carry the step's stamp and a row sum, the rest take g = 0.  It must be invisible: bit-identical parameters
and optimizer state to the dense step."""
import argparse
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def cuda():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


@pytest.mark.parametrize("l2", [0.0, 1e-4])
def test_rows_passes_equal_the_dense_update_bitwise(cuda, l2):
    """three 'tables' ([rows, 64], [rows, 1], [rows, 20] -- float4, scalar and unaligned-width paths) and one plain tensor: the
    pass over every row that reads a gradient only where the row carries the step's stamp -- and its two halves run one after
    the other -- == rc_dense_update_multi_dev with a zero-filled dense gradient"""
    from rechorus_amd import engine
    g = torch.Generator(device=cuda)
    g.manual_seed(3)
    rows = 5000
    flags = torch.zeros(rows, dtype=torch.int32, device=cuda)
    step_dev = torch.full((1,), 6, dtype=torch.int64, device=cuda)      # six steps done
    touched = torch.randperm(rows, device=cuda, generator=g)[:137]
    stale = torch.randperm(rows, device=cuda, generator=g)[:500]
    flags[stale] = 6       # stamped by the previous step: not this one
    shapes = [(rows, 64), (rows, 1), (rows, 20), (777,)]
    h = engine.make_hyper("Adam", lr=1e-3, l2=l2, step=1)
    W = [torch.randn(s, device=cuda, generator=g) for s in shapes]
    M = [torch.randn(s, device=cuda, generator=g) * 0.1 for s in shapes]
    V = [torch.rand(s, device=cuda, generator=g) * 0.01 for s in shapes]
    G = [torch.zeros(s, device=cuda) for s in shapes]
    for Gt in G[:3]:
        Gt[touched] = torch.randn((touched.numel(),) + tuple(Gt.shape[1:]), device=cuda, generator=g)
    G[3] = torch.randn(shapes[3], device=cuda, generator=g)
    # dense reference
    Wd, Md, Vd = [w.clone() for w in W], [m.clone() for m in M], [v.clone() for v in V]
    sd = step_dev.clone()
    engine.dense_update_multi([(w, gr, h, m, v) for w, gr, m, v in zip(Wd, G, Md, Vd)], "Adam", step_dev=sd)
    assert int(sd.item()) == 7
    # rows mode: stamp, increment; the gradient scratch is garbage outside the touched rows
    flags[touched] = 7
    engine.step_increment(step_dev)
    Gs = [torch.full(s, float("nan"), device=cuda) for s in shapes[:3]]
    for Gt, Gd in zip(Gs, G):
        Gt[touched] = Gd[touched]
    start = [(w.clone(), m.clone(), v.clone()) for w, m, v in zip(W, M, V)]

    def check(got_w, got_m, got_v):
        for name, got, want in (("W", got_w, Wd), ("m", got_m, Md), ("v", got_v, Vd)):
            for i in range(4):
                assert torch.equal(got[i], want[i]), (name, i, float((got[i] - want[i]).abs().max()))

    # (a) one pass over every row (what HipOptimizer.step() launches)
    every = [(W[3], G[3], h, M[3], V[3], None, 0)] + [(W[i], Gs[i], h, M[i], V[i], flags, shapes[i][1]) for i in range(3)]
    engine.dense_update_rows(every, step_dev, touched=2)
    torch.cuda.synchronize()
    check(W, M, V)
    # (b) the two halves: the unstamped rows without any gradient buffer, then the stamped rows (and the plain tensor)
    W2, M2, V2 = [x[0] for x in start], [x[1] for x in start], [x[2] for x in start]
    rest = [(W2[i], None, h, M2[i], V2[i], flags, shapes[i][1]) for i in range(3)]
    engine.dense_update_rows(rest, step_dev, touched=0, max_blocks=7)
    for i in range(3):      # the first half left the stamped rows alone
        assert torch.equal(W2[i][touched], start[i][0][touched])
    every2 = [(W2[3], G[3], h, M2[3], V2[3], None, 0)] + [(W2[i], Gs[i], h, M2[i], V2[i], flags, shapes[i][1]) for i in range(3)]
    engine.dense_update_rows(every2, step_dev, touched=1)
    torch.cuda.synchronize()
    check(W2, M2, V2)


def test_rows_entry_refuses_other_optimizers_and_bad_shapes(cuda):
    from rechorus_amd import engine
    w = torch.zeros(8, 4, device=cuda)
    flags = torch.zeros(8, dtype=torch.int32, device=cuda)
    sd = torch.ones(1, dtype=torch.int64, device=cuda)
    with pytest.raises(RuntimeError, match="Adam only"):
        engine.dense_update_rows([(w, w.clone(), engine.make_hyper("SGD", lr=0.1), w.clone(), w.clone(), flags, 4)], sd, touched=1)
    with pytest.raises(RuntimeError, match="rows x row_w"):
        engine.dense_update_rows([(w, None, engine.make_hyper("Adam"), w.clone(), w.clone(), flags, 5)], sd, touched=0)
    with pytest.raises(RuntimeError, match="null gradient"):
        engine.dense_update_rows([(w, None, engine.make_hyper("Adam"), w.clone(), w.clone(), flags, 4)], sd, touched=2)


def _deepfm(cuda, batch, dropout, rows):
    import bench
    from rechorus_amd import nn as hnn
    hnn._DROP_SEED_GEN = None
    os.environ["RC_ROWS_ADAM"] = "1" if rows else "0"
    args = argparse.Namespace(emb_size=64, mlp="[512,64]", lr=5e-4, l2=1e-6, opt="Adam", batch=batch, pool=4, dropout=dropout)
    w = bench.DeepfmBench(args, cuda)
    return w, [f for (f,) in w.batches(args, cuda, seed=11)]


@pytest.mark.parametrize("dropout", [0.0, 0.2])
def test_deepfm_steps_in_rows_mode_equal_the_dense_steps_bitwise(cuda, dropout):
    """DeepFMCTR as bench.py's `secondary.deepfm_b1024` builds it, seven steps through graph.GraphedStep (two eager, the capture,
    four replays): with HipOptimizer's rows mode and with the dense step (RC_ROWS_ADAM=0) every parameter and both Adam moments
    end bit-identical, and the rows mode really ran (no table ever held a gradient; the row flags carry the last step's stamp)"""
    from rechorus_amd import graph as hgraph
    if not hgraph.usable():
        pytest.skip("hipGraph capture not usable in this process")
    try:
        runs = []
        for rows in (True, False):
            w, batches = _deepfm(cuda, 1024, dropout, rows)
            w.model.train()
            assert w.graphed is not None
            losses = [float(w.graphed.run(dict(batches[i % 4]))) for i in range(7)]
            torch.cuda.synchronize()
            opt = w.model.optimizer
            if rows:
                assert opt._rows_state is not None and opt._rows is None
                assert int(opt._step_dev.item()) == 7
                assert int(opt._rows_state["flags"].max().item()) == 7      # stamped with the number of the step in progress
                # (no TABLE ever held a gradient; the Linear weights of the numeric c_day_f are ordinary dense parameters)
                assert all(p.grad is None for n, p in w.model.named_parameters() if "embedding" in n and not n.split(".")[1].endswith("_f"))
                assert all(p.grad is not None for n, p in w.model.named_parameters() if "embedding" in n and n.split(".")[1].endswith("_f"))
                assert '_step_optimizer' not in w.model.__dict__
            else:
                assert opt._rows_state is None
            runs.append((losses, {k: v.detach().clone() for k, v in w.model.state_dict().items()},
                         {n: (opt.state[p]["m"].clone(), opt.state[p]["v"].clone()) for n, p in w.model.named_parameters() if p in opt.state}))
        (la, pa, sa), (lb, pb, sb) = runs
        assert la == lb, (la, lb)
        for k in pa:
            assert torch.equal(pa[k], pb[k]), (k, float((pa[k].float() - pb[k].float()).abs().max()))
        assert set(sa) == set(sb)
        for k in sa:
            assert torch.equal(sa[k][0], sb[k][0]) and torch.equal(sa[k][1], sb[k][1]), k
    finally:
        os.environ.pop("RC_ROWS_ADAM", None)


def test_rows_mode_is_not_taken_outside_a_whole_step_or_at_large_batches(cuda):
    """a plain forward / backward (the reference's own loop, tests that read p.grad) keeps dense gradients; so does a batch
    beyond the small route"""
    from rechorus_amd import graph as hgraph
    try:
        w, batches = _deepfm(cuda, 1024, 0.0, True)
        m = w.model
        m.train()
        m.optimizer.zero_grad()
        m.loss(m(batches[0])).backward()
        tables = [p for n, p in m.named_parameters() if n.startswith("context_embedding")]
        assert all(p.grad is not None for p in tables)
        assert m.optimizer._rows_state is None
        m.optimizer.step()
        if hgraph.usable():
            w2, batches2 = _deepfm(cuda, 16384, 0.0, True)
            w2.model.train()
            for i in range(4):
                w2.graphed.run(dict(batches2[i % 4]))
            torch.cuda.synchronize()
            assert w2.model.optimizer._rows_state is None
    finally:
        os.environ.pop("RC_ROWS_ADAM", None)


def test_a_step_that_raises_between_forward_and_step_does_not_poison_the_next_one(cuda):
    """a loss function that raises after the forward (the gather has stamped its rows, the optimizer holds a pending rows-mode step):
    GraphedStep clears the pending step and the stamps; the following steps train exactly like a run that never saw the failed batch"""
    from rechorus_amd import graph as hgraph
    if not hgraph.usable():
        pytest.skip("hipGraph capture not usable in this process")
    try:
        res = []
        for fail in (True, False):
            w, batches = _deepfm(cuda, 1024, 0.0, True)
            w.model.train()
            g = w.graphed
            losses = [float(g.run(dict(batches[0])))]
            if fail:
                real = g.loss_of

                def boom(m, b):
                    m(b)
                    raise RuntimeError("boom")
                g.loss_of = boom
                with pytest.raises(RuntimeError, match="boom"):
                    g.run(dict(batches[1]))
                g.loss_of = real
                g.seen -= 1        # the failed call was not a training step
                opt = w.model.optimizer
                assert opt._rows is None and int(opt._rows_state["flags"].max().item()) == 0
            losses += [float(g.run(dict(batches[i % 4]))) for i in range(2, 7)]
            torch.cuda.synchronize()
            res.append((losses, {k: v.detach().clone() for k, v in w.model.state_dict().items()}))
        (la, pa), (lb, pb) = res
        assert la == lb
        for k in pa:
            assert torch.equal(pa[k], pb[k]), k
    finally:
        os.environ.pop("RC_ROWS_ADAM", None)
