"""GPU parity of the NeuMF head kernels (fp32 MFMA) vs the reference's own outputs
(tests/golden/neumf_*.npz) and the numpy oracle."""
import os

import numpy as np
import pytest
import torch

from conftest import ROOT, assert_close, assert_update_close, load_golden
from oracle import neumf_oracle as NO
from test_oracle_neumf import CASES, params

pytestmark = pytest.mark.gpu

NAMES = {"mf_u": "mf_u_embeddings.weight", "mf_i": "mf_i_embeddings.weight", "mlp_u": "mlp_u_embeddings.weight",
         "mlp_i": "mlp_i_embeddings.weight", "W1": "mlp.0.weight", "b1": "mlp.0.bias", "w_out": "prediction.weight"}


def to_dev(P, cuda):
    return {k: torch.from_numpy(np.ascontiguousarray(P[v].reshape(-1) if k == "w_out" else P[v])).to(cuda)
            for k, v in NAMES.items()}


@pytest.fixture(scope="module")
def eng(cuda):
    from rechorus_amd import engine
    return engine


@pytest.mark.parametrize("case", CASES)
def test_neumf_forward_backward(case, cuda, eng):
    g = load_golden(case)
    P = to_dev(params(g), cuda)
    uid, iid = torch.from_numpy(g["uid"]).to(cuda), torch.from_numpy(g["iid"]).to(cuda)
    B, C = g["iid"].shape
    assert eng.neumf_supported(P["mf_u"].shape[1], P["W1"].shape[0])
    pred = eng.neumf_fwd(P, uid, iid)
    assert_close(pred.cpu().numpy(), g["pred"], what="pred")
    rows, dense = eng.neumf_bwd(P, uid, iid, torch.from_numpy(g["gpred"]).to(cuda))
    G = params(g, "G/")
    assert_close(dense["W1"].cpu().numpy(), G["mlp.0.weight"], what="dW1", atol_scale=2e-5)
    assert_close(dense["b1"].cpu().numpy(), G["mlp.0.bias"], what="db1", atol_scale=2e-5)
    assert_close(dense["w_out"].cpu().numpy(), G["prediction.weight"][0], what="dw_out", atol_scale=2e-5)
    uid_occ = uid.repeat_interleave(C)
    for tab, key, ids in (("mf_u", "g_mf_u", uid_occ), ("mlp_u", "g_mlp_u", uid_occ),
                          ("mf_i", "g_mf_i", iid), ("mlp_i", "g_mlp_i", iid)):
        dense_tab = eng.embedding_dense_backward(rows[key], ids, P[tab].shape[0])
        assert_close(dense_tab.cpu().numpy(), G[NAMES[tab]], what="grad " + tab, atol_scale=2e-5)


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("tag,opt", [("SGD_l20.001", "SGD"), ("Adam_l20.0001", "Adam")])
def test_neumf_two_fit_iterations_match_reference(case, tag, opt, cuda, eng):
    g = load_golden(case)
    lr, l2 = (float(x) for x in g[tag + "_hyper"])
    P0 = params(g)
    P = to_dev(P0, cuda)
    tr = eng.NeumfTrainer(P, opt=opt, lr=lr, l2=l2, rowwise=False)  # dense = exact reference semantics
    for step, (u, i) in enumerate(((g["uid"], g["iid"]), (g["uid2"], g["iid2"])), 1):
        loss = tr.step(torch.from_numpy(u).to(cuda), torch.from_numpy(i).to(cuda))
        assert_close(loss.cpu().numpy()[0], g[tag + "_losses"][step - 1], what=f"loss {step}")
    want = params(g, tag + "/")
    ex = 1e-3 * lr if opt == "Adam" else 0.0
    for k, name in NAMES.items():
        w0 = P0[name].reshape(-1) if k == "w_out" else P0[name]
        w1 = want[name].reshape(-1) if k == "w_out" else want[name]
        assert_update_close(P[k].cpu().numpy(), w0, w1, what=k, extra_atol=ex)


def test_neumf_random_shapes_vs_oracle(cuda, eng):
    rng = np.random.default_rng(3)
    for d, l1, B, C in ((32, 64, 7, 3), (64, 32, 33, 5), (64, 128, 20, 13), (128, 32, 65, 2), (32, 128, 3, 70)):
        P = {"mf_u_embeddings.weight": rng.normal(0, 0.3, (17, d)), "mf_i_embeddings.weight": rng.normal(0, 0.3, (40, d)),
             "mlp_u_embeddings.weight": rng.normal(0, 0.3, (17, d)), "mlp_i_embeddings.weight": rng.normal(0, 0.3, (40, d)),
             "mlp.0.weight": rng.normal(0, 0.2, (l1, 2 * d)), "mlp.0.bias": rng.normal(0, 0.2, l1),
             "prediction.weight": rng.normal(0, 0.2, (1, d + l1))}
        P = {k: v.astype(np.float32) for k, v in P.items()}
        uid = rng.integers(0, 17, size=B).astype(np.int64)
        iid = rng.integers(0, 40, size=(B, C)).astype(np.int64)
        gp = rng.normal(size=(B, C)).astype(np.float32)
        Pd = to_dev(P, cuda)
        u, i = torch.from_numpy(uid).to(cuda), torch.from_numpy(iid).to(cuda)
        pred = eng.neumf_fwd(Pd, u, i)
        want_pred, G = NO.backward(P, uid, iid, gp)
        assert_close(pred.cpu().numpy(), want_pred, what=f"pred d={d} l1={l1}", atol_scale=2e-5)
        rows, dense = eng.neumf_bwd(Pd, u, i, torch.from_numpy(gp).to(cuda))
        assert_close(dense["W1"].cpu().numpy(), G["mlp.0.weight"], what="dW1", atol_scale=3e-5)
        assert_close(dense["b1"].cpu().numpy(), G["mlp.0.bias"], what="db1", atol_scale=3e-5)
        assert_close(dense["w_out"].cpu().numpy(), G["prediction.weight"][0], what="dw_out", atol_scale=3e-5)
        t = eng.embedding_dense_backward(rows["g_mlp_i"], i, 40)
        assert_close(t.cpu().numpy(), G["mlp_i_embeddings.weight"], what="grad mlp_i", atol_scale=3e-5)
        t = eng.embedding_dense_backward(rows["g_mf_u"], u.repeat_interleave(C), 17)
        assert_close(t.cpu().numpy(), G["mf_u_embeddings.weight"], what="grad mf_u", atol_scale=3e-5)
    assert not eng.neumf_supported(48, 64) and not eng.neumf_supported(128, 128)


# ---- hidden-layer dropout inside the kernels (rc_neumf_fwd_dropout / rc_neumf_bwd_dropout) ---------------------

def _seed(cuda, value):
    return torch.tensor([value], dtype=torch.int64, device=cuda)


def _check_against(eng, P, Pd, uid, iid, gp, keep, p, seed, tol, cuda, what):
    u, i = torch.from_numpy(uid).to(cuda), torch.from_numpy(iid).to(cuda)
    C = iid.shape[1]
    pred = eng.neumf_fwd(Pd, u, i, p, seed)
    want_pred, G = NO.backward(P, uid, iid, gp, keep)
    assert_close(pred.cpu().numpy(), want_pred, what=what + " pred", atol_scale=tol)
    rows, dense = eng.neumf_bwd(Pd, u, i, torch.from_numpy(gp).to(cuda), p, seed)
    assert_close(dense["W1"].cpu().numpy(), G["mlp.0.weight"], what=what + " dW1", atol_scale=tol)
    assert_close(dense["b1"].cpu().numpy(), G["mlp.0.bias"], what=what + " db1", atol_scale=tol)
    assert_close(dense["w_out"].cpu().numpy(), G["prediction.weight"][0], what=what + " dw_out", atol_scale=tol)
    uid_occ = u.repeat_interleave(C)
    for tab, key, ids in (("mf_u", "g_mf_u", uid_occ), ("mlp_u", "g_mlp_u", uid_occ),
                          ("mf_i", "g_mf_i", i), ("mlp_i", "g_mlp_i", i)):
        t = eng.embedding_dense_backward(rows[key], ids, Pd[tab].shape[0])
        assert_close(t.cpu().numpy(), G[NAMES[tab]], what=what + " grad " + tab, atol_scale=tol)
    return pred


from test_oracle_neumf import DROP_CASES  # noqa: E402


@pytest.mark.parametrize("case", DROP_CASES)
def test_neumf_dropout_matches_reference_run_with_the_same_mask(case, cuda, eng):
    g = load_golden(case)
    P = params(g)
    B, C = g["iid"].shape
    p, seed = float(g["p"]), _seed(cuda, int(g["mask_seed"]))
    pred = eng.neumf_fwd(to_dev(P, cuda), torch.from_numpy(g["uid"]).to(cuda), torch.from_numpy(g["iid"]).to(cuda), p, seed)
    assert_close(pred.cpu().numpy(), g["pred"], what="pred")  # the reference's own output
    keep = NO.dropout_keep(int(g["mask_seed"]), B * C, P["mlp.0.weight"].shape[0], p)
    _check_against(eng, P, to_dev(P, cuda), g["uid"], g["iid"], g["gpred"], keep, p, seed, 2e-5, cuda, case)
    G = params(g, "G/")
    _, dense = eng.neumf_bwd(to_dev(P, cuda), torch.from_numpy(g["uid"]).to(cuda), torch.from_numpy(g["iid"]).to(cuda),
                             torch.from_numpy(g["gpred"]).to(cuda), p, seed)
    assert_close(dense["W1"].cpu().numpy(), G["mlp.0.weight"], what="dW1 vs reference", atol_scale=2e-5)


def test_neumf_dropout_all_shapes_vs_oracle_and_p0_is_the_plain_head(cuda, eng):
    rng = np.random.default_rng(11)
    shapes = [(d, l1) for d in (32, 64, 128) for l1 in (32, 64, 128) if (d, l1) != (128, 128)]
    for k, (d, l1) in enumerate(shapes):
        B, C = 5 + 9 * k, 1 + (k * 5) % 13   # tiles with ragged tails, several tiles per workgroup
        P = {"mf_u_embeddings.weight": rng.normal(0, 0.3, (17, d)), "mf_i_embeddings.weight": rng.normal(0, 0.3, (40, d)),
             "mlp_u_embeddings.weight": rng.normal(0, 0.3, (17, d)), "mlp_i_embeddings.weight": rng.normal(0, 0.3, (40, d)),
             "mlp.0.weight": rng.normal(0, 0.2, (l1, 2 * d)), "mlp.0.bias": rng.normal(0, 0.2, l1),
             "prediction.weight": rng.normal(0, 0.2, (1, d + l1))}
        P = {kk: v.astype(np.float32) for kk, v in P.items()}
        uid = rng.integers(0, 17, size=B).astype(np.int64)
        iid = rng.integers(0, 40, size=(B, C)).astype(np.int64)
        gp = rng.normal(size=(B, C)).astype(np.float32)
        Pd = to_dev(P, cuda)
        p, sv = (0.2, 0.5, 0.75)[k % 3], 1000003 * (k + 1) + (k << 40)
        keep = NO.dropout_keep(sv, B * C, l1, p)
        _check_against(eng, P, Pd, uid, iid, gp, keep, p, _seed(cuda, sv), 3e-5, cuda, f"d={d} l1={l1} p={p}")
        # p = 0 through the dropout entry points == the plain entry points, bit for bit
        u, i = torch.from_numpy(uid).to(cuda), torch.from_numpy(iid).to(cuda)
        a, b = eng.neumf_fwd(Pd, u, i), eng.neumf_fwd(Pd, u, i, 0.0, _seed(cuda, 5))
        assert torch.equal(a, b)
        # same seed -> same mask, other seed -> other prediction
        s1 = eng.neumf_fwd(Pd, u, i, p, _seed(cuda, sv))
        assert torch.equal(s1, eng.neumf_fwd(Pd, u, i, p, _seed(cuda, sv)))
        assert not torch.equal(s1, eng.neumf_fwd(Pd, u, i, p, _seed(cuda, sv + 1)))


def test_neumf_dropout_rejects_bad_arguments(cuda, eng):
    g = load_golden(CASES[0])
    P = to_dev(params(g), cuda)
    uid, iid = torch.from_numpy(g["uid"]).to(cuda), torch.from_numpy(g["iid"]).to(cuda)
    with pytest.raises(ValueError):
        eng.neumf_fwd(P, uid, iid, 0.2, None)
    with pytest.raises(RuntimeError, match="outside"):
        eng.neumf_fwd(P, uid, iid, 1.0, _seed(cuda, 1))


def test_neumf_trainer_with_dropout_draws_a_new_mask_every_step(cuda, eng):
    """NeumfTrainer(dropout=p, seed=s): step k uses the mask of seed s + k; parameters after two SGD steps
    equal the oracle's with those two masks"""
    from oracle import bprmf_oracle as BO
    g = load_golden(CASES[0])
    P0 = params(g)
    P = to_dev(P0, cuda)
    p, s, lr = 0.3, 4242, 0.05
    tr = eng.NeumfTrainer(P, opt="SGD", lr=lr, l2=0.0, rowwise=False, dropout=p, seed=s)
    W = {k: v.copy() for k, v in P0.items()}
    for step, (u, i) in enumerate(((g["uid"], g["iid"]), (g["uid2"], g["iid2"])), 1):
        loss = tr.step(torch.from_numpy(u).to(cuda), torch.from_numpy(i).to(cuda))
        keep = NO.dropout_keep(s + step, i.size, W["mlp.0.weight"].shape[0], p)
        pred, _ = NO.forward(W, u, i, keep)
        assert_close(loss.cpu().numpy()[0], BO.bpr_loss(pred), what=f"loss {step}")
        _, G = NO.backward(W, u, i, BO.bpr_loss_grad(pred), keep)
        W = {k: (v - np.float32(lr) * G[k]).astype(np.float32) for k, v in W.items()}
    assert int(tr.seed.item()) == s + 2
    for k, name in NAMES.items():
        w0 = P0[name].reshape(-1) if k == "w_out" else P0[name]
        w1 = W[name].reshape(-1) if k == "w_out" else W[name]
        assert_update_close(P[k].cpu().numpy(), w0, w1, what=k)


@pytest.mark.parametrize("d,opt", [(32, "SGD"), (64, "Adam"), (128, "Adagrad"), (64, "dense")])
def test_segmented_update_pair_equals_two_single_updates(d, opt, cuda, eng):
    """rc_segmented_update_pair (two tables with the same ids in one pass) vs two rc_segmented_update calls:
    bit-identical rows, optimizer state and dense gradients for ordinary rows (same sequential sum); hot rows take
    the chunked path, whose partition depends on the lanes per row, so they agree to rounding"""
    rng = np.random.default_rng(40 + d)
    n_rows, n_occ = 300, 20000
    ids = rng.integers(0, n_rows, size=n_occ).astype(np.int64)
    ids[:5000] = 7          # one hot row -> chunked path
    ids[5000:5040] = 9      # just above the 32-occurrence threshold
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(cuda)
    src_a, src_b = t(rng.normal(size=(n_occ, d)).astype(np.float32)), t(rng.normal(size=(n_occ, d)).astype(np.float32))
    keys, perm = eng.sort_ids(t(ids), n_rows)
    count = torch.from_numpy(np.bincount(ids, minlength=n_rows)).to(cuda)
    if opt == "dense":
        Ga, Gb, Ga2, Gb2 = (torch.zeros((n_rows, d), device=cuda) for _ in range(4))
        eng.segmented_update_pair(keys, perm, src_a, src_b, dense_grad=(Ga, Gb))
        eng.segmented_update(keys, perm, src_a, dense_grad=Ga2)
        eng.segmented_update(keys, perm, src_b, dense_grad=Gb2)
        hot = count > 32
        assert torch.equal(Ga[~hot], Ga2[~hot]) and torch.equal(Gb[~hot], Gb2[~hot])
        assert_close(Ga.cpu().numpy(), Ga2.cpu().numpy(), what="Ga", atol_scale=2e-6)
        assert_close(Gb.cpu().numpy(), Gb2.cpu().numpy(), what="Gb", atol_scale=2e-6)
        want = np.zeros((n_rows, d))
        np.add.at(want, ids, src_a.cpu().numpy().astype(np.float64))
        assert_close(Ga.cpu().numpy(), want.astype(np.float32), what="dense grad", atol_scale=2e-5)
        return
    W0 = [rng.normal(size=(n_rows, d)).astype(np.float32) for _ in range(2)]
    out = []
    for pair in (True, False):
        W = [t(w) for w in W0]
        m = [torch.zeros_like(w) if opt != "SGD" else None for w in W]
        v = [torch.zeros_like(w) if opt == "Adam" else None for w in W]
        for step in (1, 2):
            h = eng.make_hyper(opt, lr=0.05, l2=1e-3, step=step)
            if pair:
                eng.segmented_update_pair(keys, perm, src_a, src_b, hyper=h, W=tuple(W), m=tuple(m), v=tuple(v))
            else:
                eng.segmented_update(keys, perm, src_a, hyper=h, W=W[0], m=m[0], v=v[0])
                eng.segmented_update(keys, perm, src_b, hyper=h, W=W[1], m=m[1], v=v[1])
        out.append((W, m, v))
    hot = count > 32
    for k in range(2):
        assert torch.equal(out[0][0][k][~hot], out[1][0][k][~hot])
        assert_update_close(out[0][0][k].cpu().numpy(), W0[k], out[1][0][k].cpu().numpy(), what=f"W{k}",
                            extra_atol=1e-3 * 0.05 if opt != "SGD" else 0.0)
        assert not torch.equal(out[0][0][k], t(W0[k]))
        for st in (1, 2):
            if out[0][st][k] is not None:
                assert torch.equal(out[0][st][k][~hot], out[1][st][k][~hot])
                assert_close(out[0][st][k].cpu().numpy(), out[1][st][k].cpu().numpy(), what=f"state {st}", atol_scale=1e-5)
    untouched = np.setdiff1d(np.arange(n_rows), ids)
    if len(untouched):
        assert np.array_equal(out[0][0][0].cpu().numpy()[untouched], W0[0][untouched])


# ---- the fused fit step (rc_neumf_train_step, csrc/neumf_step.hip) --------------------------------------------------------

def _fused_buffers(eng, Pd, B, C, cuda):
    d = Pd["mf_u"].shape[1]
    e = lambda *shape: torch.full(shape, float("nan"), dtype=torch.float32, device=cuda)
    out = {"loss_vec": e(B), "g_mf_i": e(B * C, d), "g_mlp_i": e(B * C, d), "gu_mf": e(B, d), "gu_mlp": e(B, d),
           "W1": torch.empty_like(Pd["W1"]), "b1": torch.empty_like(Pd["b1"]), "w_out": torch.empty_like(Pd["w_out"])}
    from rechorus_amd import _lib
    marks = torch.zeros(int(_lib.load().rc_neumf_train_step_marks_bytes(Pd["mf_i"].shape[0])), dtype=torch.uint8, device=cuda)
    return out, marks


def _check_fused_against(eng, P, Pd, state, uid, iid, opt, lr, l2, step, want_pred, want_loss_rows, G, tol, cuda, what, drop=None):
    """one rc_neumf_train_step call against reference / oracle quantities: predictions, per-tuple loss, dense gradients, per-tuple
    user gradient rows, single-occurrence item rows after the row-wise optimizer step (and their state), multi-occurrence rows
    untouched with their gradient rows written, the flag bytes back to zero"""
    from oracle import bprmf_oracle as BO
    B, C = iid.shape
    out, marks = _fused_buffers(eng, Pd, B, C, cuda)
    pred = torch.empty((B, C), dtype=torch.float32, device=cuda)
    W0 = {k: Pd[k].cpu().numpy().copy() for k in ("mf_i", "mlp_i", "mf_u", "mlp_u")}
    S0 = {k: {s: t.cpu().numpy().copy() for s, t in state[k].items()} for k in ("mf_i", "mlp_i")}
    u, i = torch.from_numpy(uid).to(cuda), torch.from_numpy(iid).to(cuda)
    h = eng.make_hyper(opt, lr=lr, l2=l2, step=step)
    if drop is None:
        eng.neumf_train_step(Pd, state, u, i, h, marks, out, pred=pred)
    else:      # (p, device seed): rc_neumf_train_step_dropout
        eng.neumf_train_step(Pd, state, u, i, h, marks, out, pred=pred, drop_p=drop[0], seed=drop[1])
    torch.cuda.synchronize()
    n_items = Pd["mf_i"].shape[0]
    assert not marks[(4 * n_items + 255) // 256 * 256:].any(), what + ": the multi-occurrence flags must be zero again after the step"
    assert_close(pred.cpu().numpy(), want_pred, what=what + " pred", atol_scale=tol)
    assert_close(out["loss_vec"].cpu().numpy(), want_loss_rows, what=what + " loss rows", atol_scale=tol)
    assert_close(out["W1"].cpu().numpy(), G["mlp.0.weight"], what=what + " dW1", atol_scale=tol)
    assert_close(out["b1"].cpu().numpy(), G["mlp.0.bias"], what=what + " db1", atol_scale=tol)
    assert_close(out["w_out"].cpu().numpy(), G["prediction.weight"][0], what=what + " dw_out", atol_scale=tol)
    for key, tab in (("gu_mf", "mf_u"), ("gu_mlp", "mlp_u")):
        T = np.zeros(W0[tab].shape, dtype=np.float64)
        np.add.at(T, uid, out[key].cpu().numpy().astype(np.float64))
        assert_close(T, G[NAMES[tab]], what=what + " grad " + tab, atol_scale=tol)
        assert np.array_equal(Pd[tab].cpu().numpy(), W0[tab]), what + ": user tables are only read"
    ids, cnt = np.unique(iid, return_counts=True)
    single, multi = ids[cnt == 1], ids[cnt >= 2]
    flat = iid.reshape(-1)
    for key, tab in (("g_mf_i", "mf_i"), ("g_mlp_i", "mlp_i")):
        Wn = Pd[tab].cpu().numpy()
        Wref = W0[tab].copy()
        sref = {s: a.copy() for s, a in S0[tab].items()}
        BO.opt_step_dense(Wref, G[NAMES[tab]], sref, opt, lr, l2, step=step, rows=single)
        assert_update_close(Wn, W0[tab], Wref, what=what + " " + tab, extra_atol=1e-3 * lr if opt != "SGD" else 0.0)
        others = np.setdiff1d(np.arange(Wn.shape[0]), single)
        assert np.array_equal(Wn[others], W0[tab][others]), what + ": rows that are not single-occurrence rows must not move"
        for s in sref:
            assert_close(state[tab][s].cpu().numpy(), sref[s], what=what + f" state {s} of {tab}", atol_scale=2e-5)
        rows = out[key].cpu().numpy()
        pos_multi = np.isin(flat, multi)
        assert np.isnan(rows[~pos_multi]).all(), what + ": gradient rows are written for multi-occurrence positions only"
        T = np.zeros(Wn.shape, dtype=np.float64)
        np.add.at(T, flat[pos_multi], rows[pos_multi].astype(np.float64))
        assert_close(T[multi], G[NAMES[tab]][multi], what=what + " multi rows of " + tab, atol_scale=tol)
    return out


def _state_for(eng, Pd, opt, rng, cuda):
    st = {}
    for k, t in Pd.items():
        s = {}
        if opt in ("Adam", "Adagrad"):
            s["m"] = torch.from_numpy(rng.normal(0, 0.01, tuple(t.shape)).astype(np.float32) ** (2 if opt == "Adagrad" else 1)).to(cuda)
        if opt == "Adam":
            s["v"] = torch.from_numpy((rng.normal(0, 0.01, tuple(t.shape)).astype(np.float32)) ** 2).to(cuda)
        st[k] = s
    return st


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("opt", ["SGD", "Adam", "Adagrad"])
def test_neumf_fused_step_matches_the_reference_run(case, opt, cuda, eng):
    """the fused kernel on the golden batch: predictions, loss, every gradient the reference's autograd produced"""
    g = load_golden(case)
    P0 = params(g)
    Pd = to_dev(P0, cuda)
    B, C = g["iid"].shape
    assert eng.neumf_train_step_supported(C, Pd["mf_u"].shape[1], Pd["W1"].shape[0])
    from oracle import bprmf_oracle as BO
    state = _state_for(eng, Pd, opt, np.random.default_rng(5), cuda)
    out = _check_fused_against(eng, P0, Pd, state, g["uid"], g["iid"], opt, 0.05, 1e-3, 3, g["pred"], BO.bpr_loss_rows(g["pred"])[0],
                               params(g, "G/"), 2e-5, cuda, case + " " + opt)
    assert_close(out["loss_vec"].mean().cpu().numpy(), g["loss"], what="loss")


def test_neumf_fused_step_random_shapes_vs_oracle(cuda, eng):
    """shapes that span several workgroup rounds, ragged tails (B not a multiple of 16 / 64), hot rows and C from 2 to 100"""
    from oracle import bprmf_oracle as BO
    rng = np.random.default_rng(11)
    for d, l1, B, C, n_items, opt in ((128, 64, 1000, 5, 3000, "SGD"), (64, 64, 777, 2, 500, "Adam"), (32, 32, 130, 17, 4000, "Adagrad"),
                                      (64, 32, 65, 100, 900, "SGD"), (32, 64, 1, 3, 50, "SGD"), (128, 32, 200, 5, 100000, "Adam"),
                                      (64, 16, 300, 5, 2000, "Adam"), (128, 16, 150, 9, 700, "SGD")):
        n_users = 37
        P = {"mf_u_embeddings.weight": rng.normal(0, 0.3, (n_users, d)), "mf_i_embeddings.weight": rng.normal(0, 0.3, (n_items, d)),
             "mlp_u_embeddings.weight": rng.normal(0, 0.3, (n_users, d)), "mlp_i_embeddings.weight": rng.normal(0, 0.3, (n_items, d)),
             "mlp.0.weight": rng.normal(0, 0.2, (l1, 2 * d)), "mlp.0.bias": rng.normal(0, 0.2, l1),
             "prediction.weight": rng.normal(0, 0.2, (1, d + l1))}
        P = {k: v.astype(np.float32) for k, v in P.items()}
        uid = rng.integers(0, n_users, size=B).astype(np.int64)
        iid = rng.integers(0, n_items, size=(B, C)).astype(np.int64)
        iid[:, 0] = iid[:, 0] % 7      # hot positives
        Pd = to_dev(P, cuda)
        assert eng.neumf_train_step_supported(C, d, l1)
        pred, _ = NO.forward(P, uid, iid)
        gp = BO.bpr_loss_grad(pred)
        _, G = NO.backward(P, uid, iid, gp)
        state = _state_for(eng, Pd, opt, rng, cuda)
        _check_fused_against(eng, P, Pd, state, uid, iid, opt, 0.03, 1e-4, 2, pred, BO.bpr_loss_rows(pred)[0], G, 3e-5, cuda,
                             f"d={d} l1={l1} B={B} C={C} {opt}")
    assert not eng.neumf_train_step_supported(1, 64, 64) and not eng.neumf_train_step_supported(5, 48, 64)
    assert not eng.neumf_train_step_supported(5, 64, 128) and not eng.neumf_train_step_supported(200, 128, 64)
    assert eng.neumf_train_step_supported(5, 64, 16) and not eng.neumf_train_step_supported(5, 32, 16)


@pytest.mark.parametrize("opt,lookahead", [("SGD", True), ("Adam", False), ("Adagrad", True)])
def test_neumf_trainer_fused_equals_three_kernel_step(opt, lookahead, cuda, eng, monkeypatch):
    """NeumfTrainer (row-wise) on the fused kernel vs the forward / loss / backward / update chain it replaces: two steps, same
    tables, optimizer state and dense parameters to rounding (the kernels sum in different orders)"""
    rng = np.random.default_rng(17)
    d, l1, B, C, n_users, n_items = 64, 64, 3000, 5, 400, 20000
    P0 = {"mf_u": rng.normal(0, 0.2, (n_users, d)), "mf_i": rng.normal(0, 0.2, (n_items, d)), "mlp_u": rng.normal(0, 0.2, (n_users, d)),
          "mlp_i": rng.normal(0, 0.2, (n_items, d)), "W1": rng.normal(0, 0.2, (l1, 2 * d)), "b1": rng.normal(0, 0.2, l1),
          "w_out": rng.normal(0, 0.2, d + l1)}
    P0 = {k: v.astype(np.float32) for k, v in P0.items()}
    batches = []
    for _ in range(2):
        uid = rng.integers(0, n_users, size=B).astype(np.int64)
        iid = rng.integers(0, n_items, size=(B, C)).astype(np.int64)
        iid[:, 0] = iid[:, 0] % 50
        batches.append((torch.from_numpy(uid).to(cuda), torch.from_numpy(iid).to(cuda)))
    lr = 0.05 if opt == "SGD" else 1e-2
    res = []
    monkeypatch.setattr(eng, "_SAS_OVERLAP_MIN", 0)     # the two-stream paths (plan ahead, the two update sides side by side) at test size
    for fused in (True, False):
        monkeypatch.setattr(eng, "_NEUMF_FUSED", fused)
        P = {k: torch.from_numpy(v).to(cuda) for k, v in P0.items()}
        tr = eng.NeumfTrainer(P, opt=opt, lr=lr, l2=1e-4, rowwise=True)
        if opt == "Adagrad":
            # torch's Adagrad starts from state_sum = 0 with eps = 1e-10: the first step of an element is lr * g / |g|, so an
            # element whose gradient is summation-order noise (|g| ~ 1e-9) moves by a full lr in either implementation; a
            # non-zero initial accumulator (initial_accumulator_value) makes the comparison well-conditioned
            for st in tr.state.values():
                st["m"].fill_(1e-6)
        tr.timing = {}
        losses = [float(tr.step(u, i, next_batch=batches[(k + 1) % 2] if lookahead else None).item()) for k, (u, i) in enumerate(batches)]
        assert ("fused_step" in tr.timing) == fused
        res.append((P, tr.state, losses))
    (Pa, Sa, La), (Pb, Sb, Lb) = res
    assert_close(np.array(La), np.array(Lb), what="losses")
    ex = 1e-3 * lr if opt != "SGD" else 0.0
    for k in P0:
        assert_update_close(Pa[k].cpu().numpy(), P0[k], Pb[k].cpu().numpy(), what=k, extra_atol=ex)
        assert not np.array_equal(Pa[k].cpu().numpy(), P0[k])
        for s in Sa[k]:
            assert_close(Sa[k][s].cpu().numpy(), Sb[k][s].cpu().numpy(), what=f"state {s} of {k}", atol_scale=2e-5)


def test_neumf_head_on_strided_row_blocks_vs_oracle(cuda, eng):
    """rc_neumf_head_fwd_bwd (the fused kernel without table updates, rows read through a stride -- what the row-sharded step runs
    on the blocks it fetched): loss, every gradient row, dense gradients against the numpy oracle; the blocks are not modified"""
    from oracle import bprmf_oracle as BO
    rng = np.random.default_rng(29)
    for d, l1, B, C in ((128, 64, 300, 5), (64, 32, 70, 2), (32, 64, 33, 11)):
        urows = rng.normal(0, 0.3, (B, 2 * d)).astype(np.float32)
        irows = rng.normal(0, 0.3, (B * C, 2 * d)).astype(np.float32)
        W1, b1, wo = (rng.normal(0, 0.2, s).astype(np.float32) for s in ((l1, 2 * d), (l1,), (d + l1,)))
        # the oracle's view: positional ids into per-batch tables
        P = {"mf_u_embeddings.weight": urows[:, :d].copy(), "mlp_u_embeddings.weight": urows[:, d:].copy(),
             "mf_i_embeddings.weight": irows[:, :d].copy(), "mlp_i_embeddings.weight": irows[:, d:].copy(),
             "mlp.0.weight": W1, "mlp.0.bias": b1, "prediction.weight": wo[None]}
        uid, iid = np.arange(B), np.arange(B * C).reshape(B, C)
        pred, _ = NO.forward(P, uid, iid)
        inv_b = 1.0 / (3 * B)      # (a rank's share of a global batch)
        gp = BO.bpr_loss_grad(pred, inv_b)
        _, G = NO.backward(P, uid, iid, gp)
        t = lambda a: torch.from_numpy(a).to(cuda)
        ur, ir = t(urows), t(irows)
        loss_vec, gu, gi, dense, got_pred = eng.neumf_head_fwd_bwd(ur, ir, t(W1), t(b1), t(wo), B, C, inv_b, want_pred=True)
        torch.cuda.synchronize()
        assert torch.equal(ur, t(urows)) and torch.equal(ir, t(irows))
        tol = 3e-5
        assert_close(got_pred.cpu().numpy(), pred, what="pred", atol_scale=tol)
        assert_close(loss_vec.cpu().numpy(), BO.bpr_loss_rows(pred)[0], what="loss rows", atol_scale=tol)
        gu, gi = gu.cpu().numpy(), gi.cpu().numpy()
        assert_close(gu[:, :d], G["mf_u_embeddings.weight"], what="gu mf", atol_scale=tol)
        assert_close(gu[:, d:], G["mlp_u_embeddings.weight"], what="gu mlp", atol_scale=tol)
        assert_close(gi[:, :d], G["mf_i_embeddings.weight"], what="gi mf", atol_scale=tol)
        assert_close(gi[:, d:], G["mlp_i_embeddings.weight"], what="gi mlp", atol_scale=tol)
        assert_close(dense["W1"].cpu().numpy(), G["mlp.0.weight"], what="dW1", atol_scale=tol)
        assert_close(dense["b1"].cpu().numpy(), G["mlp.0.bias"], what="db1", atol_scale=tol)
        assert_close(dense["w_out"].cpu().numpy(), G["prediction.weight"][0], what="dw_out", atol_scale=tol)


def test_neumf_trainer_announced_batches_prepared_ahead_equal_unannounced(cuda, eng, monkeypatch):
    """NeumfTrainer.step(next_batch=...) prepares the next batch's single / multi-occurrence flags and its bucket plan beside this step's
    table updates (two alternating buffers): every step announced, none, and a WRONGLY announced batch (the prepared flags are cleared, the
    step marks its own batch) leave bit-identical tables -- the flags do not depend on who computed them, nor does any sum's order"""
    rng = np.random.default_rng(31)
    d, l1, B, C, n_users, n_items = 64, 64, 2500, 5, 300, 30000
    P0 = {"mf_u": rng.normal(0, 0.2, (n_users, d)), "mf_i": rng.normal(0, 0.2, (n_items, d)), "mlp_u": rng.normal(0, 0.2, (n_users, d)),
          "mlp_i": rng.normal(0, 0.2, (n_items, d)), "W1": rng.normal(0, 0.2, (l1, 2 * d)), "b1": rng.normal(0, 0.2, l1),
          "w_out": rng.normal(0, 0.2, d + l1)}
    P0 = {k: v.astype(np.float32) for k, v in P0.items()}
    batches = []
    for _ in range(5):
        uid = rng.integers(0, n_users, size=B).astype(np.int64)
        iid = rng.integers(0, n_items, size=(B, C)).astype(np.int64)
        iid[:, 0] = iid[:, 0] % 40
        batches.append((torch.from_numpy(uid).to(cuda), torch.from_numpy(iid).to(cuda)))
    monkeypatch.setattr(eng, "_SAS_OVERLAP_MIN", 0)
    res = []
    for mode in ("ahead", "none", "wrong"):
        P = {k: torch.from_numpy(v).to(cuda) for k, v in P0.items()}
        tr = eng.NeumfTrainer(P, opt="SGD", lr=0.05, l2=1e-4, rowwise=True)
        losses = []
        for k, (u, i) in enumerate(batches[:4]):
            nxt = None
            if mode == "ahead":
                nxt = batches[k + 1]
            elif mode == "wrong" and k % 2 == 0:
                nxt = batches[4]                 # never the batch that comes
            losses.append(float(tr.step(u, i, next_batch=nxt).item()))
        torch.cuda.synchronize()
        for m in tr._marks:
            assert not m[(4 * n_items + 255) // 256 * 256:].any() or mode == "ahead"    # (only a prepared, not yet consumed batch may be marked)
        res.append((losses, {k: v.clone() for k, v in P.items()}))
    for losses, P in res[1:]:
        assert losses == res[0][0]
        for k in P0:
            assert torch.equal(P[k], res[0][1][k]), k


def test_neumf_trainer_short_step_consumes_prepared_flags_and_clears_them(cuda, eng, monkeypatch):
    """A large step (two streams) announces the ragged LAST batch of an epoch, which is below the two-stream threshold: the short
    step consumes the prepared flags and must clear them itself (on its one stream) -- a flag left behind would turn a later
    single occurrence of that row into a 'multiple' one that no plan lists, and its update would be lost.  Checked: flags all zero
    after the short step, and the following epoch's first batch (every hot row of the short batch exactly once) leaves tables
    bit-identical to a run without any announcement."""
    rng = np.random.default_rng(41)
    d, l1, B, C, n_users, n_items = 64, 64, 2048, 5, 200, 20000
    P0 = {"mf_u": rng.normal(0, 0.2, (n_users, d)), "mf_i": rng.normal(0, 0.2, (n_items, d)), "mlp_u": rng.normal(0, 0.2, (n_users, d)),
          "mlp_i": rng.normal(0, 0.2, (n_items, d)), "W1": rng.normal(0, 0.2, (l1, 2 * d)), "b1": rng.normal(0, 0.2, l1),
          "w_out": rng.normal(0, 0.2, d + l1)}
    P0 = {k: v.astype(np.float32) for k, v in P0.items()}
    t = lambda a: torch.from_numpy(a).to(cuda)
    big = (rng.integers(0, n_users, size=B).astype(np.int64), rng.integers(0, n_items, size=(B, C)).astype(np.int64))
    Bs = 100
    short_i = rng.integers(1000, n_items, size=(Bs, C)).astype(np.int64)
    short_i[:, 0] = short_i[:, 0] % 10 + 500            # rows 500..509 occur many times in the short batch
    short = (rng.integers(0, n_users, size=Bs).astype(np.int64), short_i)
    after_i = rng.integers(1000, n_items, size=(B, C)).astype(np.int64)
    after_i[:10, 1] = np.arange(500, 510)                # ... and exactly once in the batch after it
    after = (rng.integers(0, n_users, size=B).astype(np.int64), after_i)
    assert all((after_i == r).sum() == 1 for r in range(500, 510))
    batches = [tuple(map(t, b)) for b in (big, short, after)]
    monkeypatch.setattr(eng, "_SAS_OVERLAP_MIN", B * C)   # big and `after` run on two streams, the short step on one
    res = []
    for announce in (True, False):
        P = {k: t(v) for k, v in P0.items()}
        tr = eng.NeumfTrainer(P, opt="SGD", lr=0.05, l2=0.0, rowwise=True)
        for k, (u, i) in enumerate(batches):
            tr.step(u, i, next_batch=batches[k + 1] if announce and k + 1 < len(batches) else None)
            torch.cuda.synchronize()
            if announce and k == 1:
                assert getattr(tr, "_ahead", None) is None, "a one-stream step prepares nothing"
                for m in tr._marks:
                    assert not m[(4 * n_items + 255) // 256 * 256:].any(), "the short step left prepared flags behind"
        res.append({k: v.clone() for k, v in P.items()})
    for k in P0:
        assert torch.equal(res[0][k], res[1][k]), k
    assert not torch.equal(res[0]["mf_i"][500:510], t(P0["mf_i"])[500:510])


# ---- dropout inside the fused step (rc_neumf_train_step_dropout) -------------------------------------------------------------

@pytest.mark.parametrize("opt", ["SGD", "Adam", "Adagrad"])
def test_neumf_fused_step_with_dropout_matches_the_reference_run_with_the_same_mask(opt, cuda, eng):
    """the reference's own command line trains NeuMF with --dropout 0.2 (docs/demo_scripts_results/Topk_Amazon.sh:8): the golden
    batch of tests/golden/neumfdrop_d64_l64_k4_p0.2.npz (the reference's forward under this mask) through the ONE-kernel step --
    predictions equal the reference's, loss rows / every gradient / the single-occurrence rows after the optimizer step equal the
    oracle's backward under the same mask (pass 2 of the kernel regenerates pass 1's mask)"""
    from oracle import bprmf_oracle as BO
    case = "neumfdrop_d64_l64_k4_p0.2"
    g = load_golden(case)
    P0 = params(g)
    Pd = to_dev(P0, cuda)
    B, C = g["iid"].shape
    p, sv = float(g["p"]), int(g["mask_seed"])
    assert eng.neumf_train_step_supported(C, Pd["mf_u"].shape[1], Pd["W1"].shape[0])
    keep = NO.dropout_keep(sv, B * C, P0["mlp.0.weight"].shape[0], p)
    pred_o, _ = NO.forward(P0, g["uid"], g["iid"], keep)
    assert_close(pred_o, g["pred"], what="oracle under the mask vs the reference's own output")
    _, G = NO.backward(P0, g["uid"], g["iid"], BO.bpr_loss_grad(pred_o), keep)
    state = _state_for(eng, Pd, opt, np.random.default_rng(5), cuda)
    _check_fused_against(eng, P0, Pd, state, g["uid"], g["iid"], opt, 0.05, 1e-3, 3, g["pred"], BO.bpr_loss_rows(g["pred"])[0], G, 2e-5, cuda,
                         case + " " + opt, drop=(p, _seed(cuda, sv)))


def test_neumf_fused_step_with_dropout_random_shapes_vs_oracle(cuda, eng):
    """every (d, hidden) instantiation, ragged tails, hot rows, C from 2 to 100, p from 0.1 to 0.75; p = 0 through the dropout entry
    point is the plain step bit for bit; the same seed repeats the mask, another seed draws another"""
    from oracle import bprmf_oracle as BO
    rng = np.random.default_rng(23)
    for k, (d, l1, B, C, n_items, opt, p) in enumerate(((128, 64, 700, 5, 3000, "SGD", 0.2), (64, 64, 333, 2, 500, "Adam", 0.5), (32, 32, 130, 17, 4000, "Adagrad", 0.1),
                                                        (64, 32, 65, 100, 900, "SGD", 0.75), (32, 64, 1, 3, 50, "SGD", 0.2), (128, 32, 200, 5, 100000, "Adam", 0.3))):
        n_users = 37
        P = {"mf_u_embeddings.weight": rng.normal(0, 0.3, (n_users, d)), "mf_i_embeddings.weight": rng.normal(0, 0.3, (n_items, d)),
             "mlp_u_embeddings.weight": rng.normal(0, 0.3, (n_users, d)), "mlp_i_embeddings.weight": rng.normal(0, 0.3, (n_items, d)),
             "mlp.0.weight": rng.normal(0, 0.2, (l1, 2 * d)), "mlp.0.bias": rng.normal(0, 0.2, l1),
             "prediction.weight": rng.normal(0, 0.2, (1, d + l1))}
        P = {kk: v.astype(np.float32) for kk, v in P.items()}
        uid = rng.integers(0, n_users, size=B).astype(np.int64)
        iid = rng.integers(0, n_items, size=(B, C)).astype(np.int64)
        iid[:, 0] = iid[:, 0] % 7
        sv = 1000003 * (k + 1) + (k << 41)
        keep = NO.dropout_keep(sv, B * C, l1, p)
        pred, _ = NO.forward(P, uid, iid, keep)
        _, G = NO.backward(P, uid, iid, BO.bpr_loss_grad(pred), keep)
        Pd = to_dev(P, cuda)
        state = _state_for(eng, Pd, opt, rng, cuda)
        _check_fused_against(eng, P, Pd, state, uid, iid, opt, 0.03, 1e-4, 2, pred, BO.bpr_loss_rows(pred)[0], G, 3e-5, cuda,
                             f"d={d} l1={l1} B={B} C={C} {opt} p={p}", drop=(p, _seed(cuda, sv)))
    # p = 0 == the plain entry point; seeds
    u, i = torch.from_numpy(uid).to(cuda), torch.from_numpy(iid).to(cuda)
    h = eng.make_hyper("SGD", lr=0.03, l2=0.0, step=1)
    res = []
    for drop in (None, (0.0, _seed(cuda, 9)), (0.3, _seed(cuda, 9)), (0.3, _seed(cuda, 9)), (0.3, _seed(cuda, 10))):
        Pd = to_dev(P, cuda)
        out, marks = _fused_buffers(eng, Pd, B, C, cuda)
        kw = {} if drop is None else dict(drop_p=drop[0], seed=drop[1])
        eng.neumf_train_step(Pd, {kk: {} for kk in Pd}, u, i, h, marks, out, **kw)
        res.append((Pd["mlp_i"].clone(), out["loss_vec"].clone(), out["W1"].clone()))
    same = lambda a, b: all(torch.equal(x, y) for x, y in zip(a, b))
    assert same(res[0], res[1]) and same(res[2], res[3]) and not same(res[2], res[4]) and not same(res[0], res[2])


@pytest.mark.parametrize("opt", ["SGD", "Adam"])
def test_neumf_trainer_with_dropout_fused_equals_three_kernel_step(opt, cuda, eng, monkeypatch):
    """NeumfTrainer(dropout=0.2): the fused kernel and the forward / loss / backward chain draw the same masks from the same seed
    counter -- three steps, same losses, tables and dense parameters to rounding"""
    rng = np.random.default_rng(19)
    d, l1, B, C, n_users, n_items = 64, 64, 2000, 5, 300, 15000
    P0 = {"mf_u": rng.normal(0, 0.2, (n_users, d)), "mf_i": rng.normal(0, 0.2, (n_items, d)), "mlp_u": rng.normal(0, 0.2, (n_users, d)),
          "mlp_i": rng.normal(0, 0.2, (n_items, d)), "W1": rng.normal(0, 0.2, (l1, 2 * d)), "b1": rng.normal(0, 0.2, l1),
          "w_out": rng.normal(0, 0.2, d + l1)}
    P0 = {k: v.astype(np.float32) for k, v in P0.items()}
    batches = []
    for _ in range(3):
        uid = rng.integers(0, n_users, size=B).astype(np.int64)
        iid = rng.integers(0, n_items, size=(B, C)).astype(np.int64)
        iid[:, 0] = iid[:, 0] % 50
        batches.append((torch.from_numpy(uid).to(cuda), torch.from_numpy(iid).to(cuda)))
    lr = 0.05 if opt == "SGD" else 1e-2
    monkeypatch.setattr(eng, "_SAS_OVERLAP_MIN", 0)
    res = []
    for fused in (True, False):
        monkeypatch.setattr(eng, "_NEUMF_FUSED", fused)
        P = {k: torch.from_numpy(v).to(cuda) for k, v in P0.items()}
        tr = eng.NeumfTrainer(P, opt=opt, lr=lr, l2=1e-4, rowwise=True, dropout=0.2, seed=77)
        tr.timing = {}
        losses = [float(tr.step(u, i, next_batch=batches[(k + 1) % 3]).item()) for k, (u, i) in enumerate(batches)]
        assert ("fused_step" in tr.timing) == fused
        assert int(tr.seed.item()) == 77 + 3
        res.append((P, losses))
    (Pa, La), (Pb, Lb) = res
    assert_close(np.array(La), np.array(Lb), what="losses")
    assert len(set(La)) == 3
    ex = 1e-3 * lr if opt != "SGD" else 0.0
    for k in P0:
        assert_update_close(Pa[k].cpu().numpy(), P0[k], Pb[k].cpu().numpy(), what=k, extra_atol=ex)


def test_hidden_16_goes_rowwise_only_where_the_fused_step_is_really_selected(cuda, eng, monkeypatch):
    """hidden 16 exists only inside the one-kernel step (rc_neumf_supported refuses it): `--engine auto` may pick the row-wise route only
    when NeumfTrainer.step would select that step for the model's 1 + num_neg candidates and the process's switches; a trainer that
    cannot take it says why instead of failing with RC_ERR_UNSUPPORTED inside rc_neumf_fwd"""
    import argparse
    import sys
    plugin = os.path.join(ROOT, "rechorus_amd", "rechorus")
    if plugin not in sys.path:
        sys.path.insert(0, plugin)
    from models.general.NeuMF import NeuMF
    from rechorus_amd import engine

    def model(num_neg):
        args = argparse.Namespace(device=cuda, model_path="", buffer=1, num_neg=num_neg, dropout=0, test_all=0, emb_size=64, layers="[16]")
        return NeuMF(args, argparse.Namespace(n_users=50, n_items=90)).to(cuda)
    assert not eng.neumf_supported(64, 16) and eng.neumf_train_step_supported(5, 64, 16)
    assert model(4).hip_rowwise_supported()
    assert not model(0).hip_rowwise_supported()          # one candidate per tuple: the fused step needs two
    assert not model(5000).hip_rowwise_supported()       # the candidates' LDS strips do not fit
    monkeypatch.setattr(engine, "_NEUMF_FUSED", False)
    assert not model(4).hip_rowwise_supported()          # RC_NEUMF_FUSED=0
    m = model(4)
    rng = np.random.default_rng(0)
    feed = {"user_id": torch.from_numpy(rng.integers(1, 50, 32)).to(cuda), "item_id": torch.from_numpy(rng.integers(1, 90, (32, 5))).to(cuda),
            "batch_size": 32, "phase": "train"}
    with pytest.raises(RuntimeError, match="one-kernel step"):     # explicit --engine rowwise on a configuration it cannot train
        m._step_ok = lambda: True
        m.hip_train_step(feed, "SGD", 0.01, 0.0)
    monkeypatch.undo()
    m2 = model(4)
    loss = m2.hip_train_step(feed, "SGD", 0.01, 0.0)
    assert np.isfinite(float(loss))
