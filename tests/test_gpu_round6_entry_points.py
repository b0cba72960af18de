"""GPU: the entry points round 6 added, each against the call sequence it replaces, bit for bit --
rc_gather_rows_pair (two rc_gather_rows + a concatenation), rc_plan_update_pair_zeroed (rc_plan_update_pair without its memset),
rc_plan_update_pair_block (the same update reading both gradients from one strided block),
rc_sasrec_batch_bwd_part (the backward pass of rc_sasrec_batch_bwd_dropout in two calls), rc_ctr_head_fwd_full (rc_ctr_head_fwd_bwd_sums
+ rc_ctr_head_bwd for a seed gradient of one, and the counter that rides along).  The fused field gather and the planned row sums
have their own tests in test_gpu_deepfm.py.
Reference semantics: models/general/NeuMF.py:39-42,61-66 (the two table families of a side), helpers/BaseRunner.py:193-206 (the
optimizer step of the touched rows), models/sequential/SASRec.py:58-76 + utils/layers.py:92-118 and their autograd, models/context/
FM.py:59-60 + BaseModel.py:259-267 (the CTR head and nn.BCELoss)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_gather_rows_pair_is_two_gathers_side_by_side(cuda):
    from rechorus_amd import engine
    g = torch.Generator(device=cuda).manual_seed(1)
    for n_rows, d, n in ((1000, 128, 5000), (37, 16, 9), (50000, 64, 100000), (8, 4, 1)):
        A = torch.randn(n_rows, d, device=cuda, generator=g)
        B = torch.randn(n_rows, d, device=cuda, generator=g)
        ids = torch.randint(0, n_rows, (n,), device=cuda, generator=g)
        out = engine.gather_rows_pair(A, B, ids)
        assert out.shape == (n, 2 * d)
        assert torch.equal(out, torch.cat([engine.gather_rows(A, ids), engine.gather_rows(B, ids)], dim=1))
        assert torch.equal(out, torch.cat([A[ids], B[ids]], dim=1))


@pytest.mark.parametrize("opt", ["SGD", "Adam", "Adagrad"])
def test_pair_update_with_prezeroed_counters_equals_the_plain_call(opt, cuda):
    """a batch with hot rows (one id a thousand times: the chunk path and its ticket counters are exercised) through
    Plan.update_pair twice from the same start: with the counters zero-filled at plan time and with the call's own memset"""
    from rechorus_amd import engine
    g = torch.Generator(device=cuda).manual_seed(2)
    n_rows, d, n = 5000, 64, 40000
    ids = torch.randint(0, n_rows, (n,), device=cuda, generator=g)
    ids[:3000] = 7                                       # a hot row: more than one chunk
    ga, gb = torch.randn(n, d, device=cuda, generator=g), torch.randn(n, d, device=cuda, generator=g)
    h = engine.make_hyper(opt, lr=0.05, l2=1e-4, step=3)
    results = []
    for zeroed in (True, False):
        Wa, Wb = (torch.randn(n_rows, d, device=cuda, generator=torch.Generator(device=cuda).manual_seed(10 + k)) for k in range(2))
        gs = torch.Generator(device=cuda).manual_seed(77)      # the same optimizer state for both runs
        st = [{k: torch.rand(W.shape, device=cuda, generator=gs) * 0.1 for k in (("m", "v") if opt == "Adam" else (("m",) if opt == "Adagrad" else ()))}
              for W in (Wa, Wb)]
        plan = engine.Plan(ids, n_rows, tag="t_r6_pair%d" % zeroed)
        if zeroed:
            plan.prezero_update_counters()
            assert set(plan.upd_counters) == {"a", "b"}
        plan.update_pair("a", Wa, Wb, ga, gb, h, ma=st[0].get("m"), va=st[0].get("v"), mb=st[1].get("m"), vb=st[1].get("v"))
        if zeroed:
            assert set(plan.upd_counters) == {"b"}     # one use per side
        torch.cuda.synchronize()
        results.append((Wa, Wb, st))
    (Wa1, Wb1, st1), (Wa0, Wb0, st0) = results
    assert torch.equal(Wa1, Wa0) and torch.equal(Wb1, Wb0)
    for a, b in zip(st1, st0):
        for k in a:
            assert torch.equal(a[k], b[k]), k


@pytest.mark.parametrize("opt", ["SGD", "Adam"])
def test_pair_update_from_one_block_equals_two_contiguous_sources(opt, cuda):
    """rc_plan_update_pair_block: the (gradient of table a | gradient of table b) rows of one [n, 2 d] block (also with a wider row
    stride) against rc_plan_update_pair on contiguous copies of the halves"""
    from rechorus_amd import engine
    g = torch.Generator(device=cuda).manual_seed(5)
    n_rows, d, n = 3000, 128, 30000
    ids = torch.randint(0, n_rows, (n,), device=cuda, generator=g)
    ids[:2500] = 11
    for ld in (2 * d, 2 * d + 64):
        block = torch.randn(n, ld, device=cuda, generator=g)
        h = engine.make_hyper(opt, lr=0.05, l2=1e-4, step=2)
        results = []
        for as_block in (True, False):
            Wa, Wb = (torch.randn(n_rows, d, device=cuda, generator=torch.Generator(device=cuda).manual_seed(20 + k)) for k in range(2))
            gs = torch.Generator(device=cuda).manual_seed(78)
            st = [{k: torch.rand(W.shape, device=cuda, generator=gs) * 0.1 for k in (("m", "v") if opt == "Adam" else ())} for W in (Wa, Wb)]
            plan = engine.Plan(ids, n_rows, tag="t_r6_block%d" % as_block)
            kw = dict(ma=st[0].get("m"), va=st[0].get("v"), mb=st[1].get("m"), vb=st[1].get("v"))
            if as_block:
                plan.update_pair("a", Wa, Wb, block, None, h, **kw)
            else:
                plan.update_pair("a", Wa, Wb, block[:, :d].contiguous(), block[:, d:2 * d].contiguous(), h, **kw)
            torch.cuda.synchronize()
            results.append((Wa, Wb, st))
        (Wa1, Wb1, st1), (Wa0, Wb0, st0) = results
        assert torch.equal(Wa1, Wa0) and torch.equal(Wb1, Wb0), ld
        for a, b in zip(st1, st0):
            for k in a:
                assert torch.equal(a[k], b[k]), (ld, k)


@pytest.mark.parametrize("B,L,heads,drop", [(300, 50, 4, 0.0), (64, 20, 2, 0.2), (257, 64, 1, 0.0), (40, 33, 4, 0.0)])
def test_sasrec_backward_in_two_parts_is_the_one_call(B, L, heads, drop, cuda):
    from rechorus_amd import _lib, engine
    d, n_items = 64, 500
    g = torch.Generator(device=cuda).manual_seed(B + L)
    mk = lambda *s: torch.randn(*s, device=cuda, generator=g) * 0.1
    I, Pe = mk(n_items, d), mk(L + 1, d)
    layers = [{k: (mk(d, d) if k.startswith("W") else (mk(d) if not k.startswith("ln") or k.endswith("b") else 1 + mk(d))) for k in engine.SAS_LAYER_KEYS}]
    lengths = torch.randint(1, L + 1, (B,), device=cuda, generator=g)
    hist = torch.randint(1, n_items, (B, L), device=cuda, generator=g) * (torch.arange(L, device=cuda)[None, :] < lengths[:, None])
    seed = torch.tensor([12345], dtype=torch.int64, device=cuda) if drop > 0 else None
    hv, saved = engine.sasrec_fwd(I, Pe, layers, heads, hist, lengths, save=True, drop_p=drop, seed=seed)
    dhv = mk(B, d)
    g1, d1 = engine.sasrec_bwd(layers, heads, lengths, saved, dhv, drop_p=drop, seed=seed)
    g1, d1 = g1.clone(), [{k: v.clone() for k, v in x.items()} for x in d1]
    g2, d2, finish = engine.sasrec_bwd(layers, heads, lengths, saved, dhv, drop_p=drop, seed=seed, split=True)
    splits = bool(_lib.load().rc_sasrec_batch_bwd_splits(d, 1, heads, B, L, float(drop))) and saved.impl == "batch"
    torch.cuda.synchronize()
    assert torch.equal(g2, g1), "g_hist is complete after part 1"
    finish()
    torch.cuda.synchronize()
    for a, b in zip(d1, d2):
        for k in a:
            assert torch.equal(a[k], b[k]), k
    assert isinstance(splits, bool)


def test_ctr_head_full_leaves_the_backward_of_a_unit_seed_and_bumps_its_counter(cuda):
    from rechorus_amd import engine
    g = torch.Generator(device=cuda).manual_seed(4)
    for n, F, terms in ((1024, 8, 2), (37, 3, 1), (5000, 5, 0)):
        bias = torch.randn(1, device=cuda, generator=g)
        lin = torch.randn(n, F, device=cuda, generator=g)
        t = [torch.randn(n, device=cuda, generator=g) for _ in range(terms)] + [None, None]
        label = torch.randint(0, 2, (n,), device=cuda, generator=g)
        counter = torch.tensor([41], dtype=torch.int64, device=cuda)
        engine.defer_increment(counter)
        p1, s1, gz1, g_lin, g_bias = engine.ctr_head_sums(bias, lin, t[0], t[1], label, full=True)
        assert engine.take_deferred(counter) is True and int(counter.item()) == 42      # the launch took the promise along
        p0, s0, gz0 = engine.ctr_head_sums(bias, lin, t[0], t[1], label)
        assert torch.equal(p1, p0) and torch.equal(s1, s0) and torch.equal(gz1, gz0)
        one = torch.ones(1, device=cuda)
        gq, gl, gb = engine.ctr_head_bwd(gz0, s0, one, F)
        assert torch.equal(g_lin, gl) and torch.equal(g_bias, gb) and torch.equal(gz1, gq)
        # without a pending promise nothing is bumped
        p2, s2, gz2, _, _ = engine.ctr_head_sums(bias, lin, t[0], t[1], label, full=True)
        assert int(counter.item()) == 42 and torch.equal(s2, s0)
