"""CPU: the lane algebra of csrc/neumf_step.hip (the fused NeuMF fit step) emulated wave by wave in numpy and checked
against oracle/neumf_oracle.py + oracle/bprmf_oracle.py.

No kernel runs here.  The emulation follows the kernel's register / LDS layout statement by statement -- which lane holds
which element of which v_mfma_f32_16x16x4_f32 operand, the transposition strips, the wave-order sums -- so an indexing
mistake in the design shows up on the CPU box; tests/test_gpu_neumf.py holds the compiled kernel to the same oracle.
MFMA semantics (ISA): lane l supplies A[l % 16][l // 16] and B[l // 16][l % 16]; register r of lane l holds
D[4 (l // 16) + r][l % 16]."""
import numpy as np

from oracle import bprmf_oracle as BO
from oracle import neumf_oracle as NO

F32 = np.float32
LANE = np.arange(64)
I_, G_ = LANE % 16, LANE // 16


def mma16(a, b, acc):
    """acc [64, 4] += the 16x16x4 product of per-lane operands a, b [64]"""
    A = a.reshape(4, 16).T          # A[m][k] = a[m + 16 k]
    Bm = b.reshape(4, 16)           # B[k][n] = b[n + 16 k]
    Dm = A @ Bm
    out = acc.copy()
    for r in range(4):
        out[:, r] += Dm[4 * G_ + r, I_]
    return out


def emulate_round(P, uid, iid, lr, l2, multi_ids, D, L1):
    """one workgroup, rounds of 64 tuples; returns what the kernel writes"""
    B, C = iid.shape
    NCU, NT, K0 = D // 16, L1 // 16, 2 * D
    NTU = NT * NCU // 4
    W1, b1, wo = P["mlp.0.weight"], P["mlp.0.bias"], P["prediction.weight"][0]
    mf_u, mlp_u = P["mf_u_embeddings.weight"], P["mlp_u_embeddings.weight"]
    mf_i, mlp_i = P["mf_i_embeddings.weight"].copy(), P["mlp_i_embeddings.weight"].copy()
    out = dict(pred=np.zeros((B, C), F32), loss=np.zeros(B, F32), g_mf_i=np.zeros((B * C, D), F32), g_mlp_i=np.zeros((B * C, D), F32),
               gu_mf=np.zeros((B, D), F32), gu_mlp=np.zeros((B, D), F32))
    accW = np.zeros((4, NT, NCU, 64, 4))
    accU = np.zeros((4, NTU, 64, 4))
    wred = np.zeros((4, 2 * L1 + D))
    n_rounds = (((B + 15) // 16) + 3) // 4

    def slices(T, rows):     # lane (i, g) loads T[row_i][16 c + 4 g + e] -> [NCU][64][4]
        return np.stack([np.stack([T[rows, 16 * c + 4 * G_ + e] for e in range(4)], -1) for c in range(NCU)])

    for rnd in range(n_rounds):
        Tz = np.zeros((4, 16, L1))
        Th = np.zeros((4, 16, D))
        for wave in range(4):
            tup = (rnd * 4 + wave) * 16 + I_
            valid = tup < B
            tt = np.where(valid, tup, B - 1)
            u = uid[tt]
            hu, mu = slices(mlp_u, u), slices(mf_u, u)
            Zu = np.zeros((NT, 64, 4))
            for nt in range(NT):
                for cc in range(NCU):
                    for e in range(4):   # A: W1[16 nt + i][16 cc + 4 g + e]
                        Zu[nt] = mma16(W1[16 * nt + I_, 16 * cc + 4 * G_ + e], hu[cc][:, e], Zu[nt])
            muw = np.stack([mu[cc] * np.stack([wo[16 * cc + 4 * G_ + e] for e in range(4)], -1) for cc in range(NCU)])

            def hidden(hx):
                z = Zu.copy()
                for nt in range(NT):
                    for cc in range(NCU):
                        for e in range(4):
                            z[nt] = mma16(W1[16 * nt + I_, D + 16 * cc + 4 * G_ + e], hx[cc][:, e], z[nt])
                return z

            feat = lambda nt, r: 16 * nt + 4 * G_ + r
            sp = np.zeros((C, 16))
            for c in range(C):
                item = iid[tt, c]
                hx, mx = slices(mlp_i, item), slices(mf_i, item)
                z = hidden(hx)
                pp = np.zeros(64)
                for nt in range(NT):
                    for r in range(4):
                        pp += wo[D + feat(nt, r)] * np.maximum(z[nt][:, r] + b1[feat(nt, r)], 0)
                pp += (muw * mx).sum(axis=(0, 2))
                tot = pp.reshape(4, 16).sum(0)       # xor 16, xor 32: every g of row i holds the total
                sp[c] = tot
                out["pred"][tup[(G_ == 0) & valid], c] = tot[I_[(G_ == 0) & valid]]
            # loss per tuple (row i)
            pred_rows = sp.T.astype(F32)            # [16, C]
            rows_loss, _, _, _ = (-np.log(np.clip(_P(pred_rows), 1e-8, 1.0)), None, None, None)
            gp = _bpr_grad_rowmax(pred_rows, 1.0 / B)
            v16 = valid[:16]
            out["loss"][tup[:16][v16]] = rows_loss[v16]
            gp[~v16] = 0.0
            # pass 2
            dzs = np.zeros((NT, 64, 4))
            dwh = np.zeros((NT, 64, 4))
            S = np.zeros((NCU, 64, 4))
            for c in range(C):
                item = iid[tt, c]
                hx, mx = slices(mlp_i, item), slices(mf_i, item)
                single = ~np.isin(item, multi_ids)
                gc = gp[I_, c]
                z = hidden(hx)
                dz = np.zeros((NT, 64, 4))
                for nt in range(NT):
                    for r in range(4):
                        zz = z[nt][:, r] + b1[feat(nt, r)]
                        dz[nt][:, r] = np.where(zz > 0, gc * wo[D + feat(nt, r)], 0.0)
                        dwh[nt][:, r] += gc * np.maximum(zz, 0)
                    dzs[nt] += dz[nt]
                    for r in range(4):               # strip: Tz[i][16 nt + 4 g + r]
                        Tz[wave, I_, 16 * nt + 4 * G_ + r] = dz[nt][:, r]
                for cc in range(NCU):
                    for e in range(4):
                        Th[wave, I_, 16 * cc + 4 * G_ + e] = hx[cc][:, e]
                n = tup * C + c
                for kt in range(NCU):
                    acc = np.zeros((64, 4))
                    for nt in range(NT):
                        for r in range(4):           # A: W1[16 nt + 4 g + r][D + 16 kt + i]
                            acc = mma16(W1[16 * nt + 4 * G_ + r, D + 16 * kt + I_], dz[nt][:, r], acc)
                    for r in range(4):
                        col = 16 * kt + 4 * G_ + r
                        w_old = hx[kt][:, r]
                        upd = valid & single
                        gfull = acc[:, r] + l2 * w_old
                        mlp_i[item[upd], col[upd]] = (w_old - lr * gfull)[upd]
                        wr_ = valid & ~single
                        out["g_mlp_i"][n[wr_], col[wr_]] = acc[:, r][wr_]
                for cc in range(NCU):
                    for e in range(4):
                        col = 16 * cc + 4 * G_ + e
                        gr = gc * muw[cc][:, e]
                        S[cc][:, e] += gc * mx[cc][:, e]
                        upd = valid & single
                        mf_i[item[upd], col[upd]] = (mx[cc][:, e] - lr * (gr + l2 * mx[cc][:, e]))[upd]
                        wr_ = valid & ~single
                        out["g_mf_i"][n[wr_], col[wr_]] = gr[wr_]
                for s in range(4):                   # dW1i: K step s contracts candidates 4 s + g
                    for ft in range(NT):
                        for kt in range(NCU):
                            accW[wave, ft, kt] = mma16(Tz[wave, 4 * s + G_, 16 * ft + I_], Th[wave, 4 * s + G_, 16 * kt + I_], accW[wave, ft, kt])
            # tuple level
            for kt in range(NCU):
                acc = np.zeros((64, 4))
                for nt in range(NT):
                    for r in range(4):
                        acc = mma16(W1[16 * nt + 4 * G_ + r, 16 * kt + I_], dzs[nt][:, r], acc)
                for r in range(4):
                    out["gu_mlp"][tup[valid], (16 * kt + 4 * G_ + r)[valid]] = acc[:, r][valid]
            for cc in range(NCU):
                for e in range(4):
                    col = 16 * cc + 4 * G_ + e
                    out["gu_mf"][tup[valid], col[valid]] = (wo[col] * S[cc][:, e])[valid]
                    t = (mu[cc][:, e] * S[cc][:, e]).reshape(4, 16).sum(1)    # row_allreduce over i, lane i == 0 of group g adds
                    wred[wave, 2 * L1 + 16 * cc + 4 * np.arange(4) + e] += t
            for nt in range(NT):
                for r in range(4):
                    wred[wave, 16 * nt + 4 * np.arange(4) + r] += dzs[nt][:, r].reshape(4, 16).sum(1)
                    wred[wave, L1 + 16 * nt + 4 * np.arange(4) + r] += dwh[nt][:, r].reshape(4, 16).sum(1)
            for nt in range(NT):
                for r in range(4):
                    Tz[wave, I_, 16 * nt + 4 * G_ + r] = dzs[nt][:, r]
            for cc in range(NCU):
                for e in range(4):
                    Th[wave, I_, 16 * cc + 4 * G_ + e] = hu[cc][:, e]
        # after the barrier: every wave a quarter of the dW1u tiles over the four strips
        for wave in range(4):
            for w2 in range(4):
                for s in range(4):
                    for q in range(NTU):
                        tile = wave + 4 * q
                        ft, kt = tile % NT, tile // NT
                        accU[wave, q] = mma16(Tz[w2, 4 * s + G_, 16 * ft + I_], Th[w2, 4 * s + G_, 16 * kt + I_], accU[wave, q])
    dW1 = np.zeros((L1, K0))
    for wave in range(4):
        for q in range(NTU):
            tile = wave + 4 * q
            ft, kt = tile % NT, tile // NT
            for r in range(4):
                dW1[16 * ft + 4 * G_ + r, 16 * kt + I_] = accU[wave, q][:, r]
        for ft in range(NT):
            for kt in range(NCU):
                for r in range(4):
                    dW1[16 * ft + 4 * G_ + r, D + 16 * kt + I_] += accW[wave, ft, kt][:, r]
    t = wred.sum(0)
    out.update(dW1=dW1, db1=t[:L1], dw_out=np.concatenate([t[2 * L1:], t[L1:2 * L1]]), mf_i=mf_i, mlp_i=mlp_i)
    return out


def _P(pred):
    pos, neg = pred[:, 0], pred[:, 1:]
    e = np.exp(neg - neg.max(axis=1, keepdims=True))
    w = e / e.sum(axis=1, keepdims=True)
    s = 1.0 / (1.0 + np.exp(-(pos[:, None] - neg)))
    return (w * s).sum(axis=1)


def _bpr_grad_rowmax(pred, inv_b):
    pos, neg = pred[:, 0], pred[:, 1:]
    e = np.exp(neg - neg.max(axis=1, keepdims=True))
    w = e / e.sum(axis=1, keepdims=True)
    s = 1.0 / (1.0 + np.exp(-(pos[:, None] - neg)))
    P = (w * s).sum(axis=1)
    dl = np.where((P >= 1e-8) & (P <= 1.0), -inv_b / np.clip(P, 1e-8, 1.0), 0.0)
    g = np.empty_like(pred, dtype=np.float64)
    g[:, 0] = dl * (w * s * (1 - s)).sum(axis=1)
    g[:, 1:] = dl[:, None] * (w * ((s - P[:, None]) - s * (1 - s)))
    return g


def _case(D, L1, B, C, n_users, n_items, seed):
    rng = np.random.default_rng(seed)
    P = {"mf_u_embeddings.weight": rng.normal(0, 0.3, (n_users, D)), "mf_i_embeddings.weight": rng.normal(0, 0.3, (n_items, D)),
         "mlp_u_embeddings.weight": rng.normal(0, 0.3, (n_users, D)), "mlp_i_embeddings.weight": rng.normal(0, 0.3, (n_items, D)),
         "mlp.0.weight": rng.normal(0, 0.2, (L1, 2 * D)), "mlp.0.bias": rng.normal(0, 0.2, L1),
         "prediction.weight": rng.normal(0, 0.2, (1, D + L1))}
    P = {k: v.astype(F32) for k, v in P.items()}
    uid = rng.integers(0, n_users, size=B).astype(np.int64)
    iid = rng.integers(0, n_items, size=(B, C)).astype(np.int64)
    return P, uid, iid


def test_lane_algebra_of_the_fused_step_matches_the_oracle():
    for D, L1, B, C, n_items in ((32, 32, 40, 3, 90), (64, 32, 70, 2, 400)):
        P, uid, iid = _case(D, L1, B, C, 23, n_items, seed=D + B)
        lr, l2 = 0.05, 0.01
        ids, cnt = np.unique(iid, return_counts=True)
        multi = ids[cnt >= 2]
        assert 0 < len(multi) < len(ids)
        got = emulate_round(P, uid, iid, lr, l2, multi, D, L1)
        pred, _ = NO.forward(P, uid, iid)
        np.testing.assert_allclose(got["pred"], pred, rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(got["loss"], BO.bpr_loss_rows(pred)[0], rtol=1e-4, atol=1e-6)
        gp = BO.bpr_loss_grad(pred)
        _, G = NO.backward(P, uid, iid, gp)
        tol = dict(rtol=1e-4, atol=2e-6)
        np.testing.assert_allclose(got["dW1"], G["mlp.0.weight"], **tol)
        np.testing.assert_allclose(got["db1"], G["mlp.0.bias"], **tol)
        np.testing.assert_allclose(got["dw_out"], G["prediction.weight"][0], **tol)
        # user tables: per-tuple rows summed by user
        for key, name in (("gu_mf", "mf_u_embeddings.weight"), ("gu_mlp", "mlp_u_embeddings.weight")):
            T = np.zeros_like(P[name], dtype=np.float64)
            np.add.at(T, uid, got[key])
            np.testing.assert_allclose(T, G[name], **tol)
        # item tables: singleton rows moved in place by the row-wise SGD step, multi rows left alone with their gradient rows written
        for key, tab, name in (("g_mf_i", "mf_i", "mf_i_embeddings.weight"), ("g_mlp_i", "mlp_i", "mlp_i_embeddings.weight")):
            W0 = P[name].astype(np.float64)
            single_rows = ids[cnt == 1]
            want = W0[single_rows] - lr * (G[name][single_rows] + l2 * W0[single_rows])
            np.testing.assert_allclose(got[tab][single_rows], want, rtol=1e-4, atol=2e-6)
            untouched = np.setdiff1d(np.arange(W0.shape[0]), single_rows)
            np.testing.assert_array_equal(got[tab][untouched], P[name][untouched])
            T = np.zeros_like(W0)
            np.add.at(T, iid.reshape(-1), got[key])
            np.testing.assert_allclose(T[multi], G[name][multi], **tol)
            assert not got[key][np.isin(iid.reshape(-1), single_rows)].any()
