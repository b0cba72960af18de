"""GPU parity of the NeuMF head kernels (fp32 MFMA) vs the reference's own outputs
(tests/golden/neumf_*.npz) and the numpy oracle."""
import numpy as np
import pytest
import torch

from conftest import assert_close, assert_update_close, load_golden
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
