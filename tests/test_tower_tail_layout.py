"""CPU: the lane algebra of csrc/tower_tail.hip (the last hidden layer + the width -> 1 output layer of an MLP tower as one forward and
one backward kernel) and of mlp_gemm_small_kernel (csrc/mlp.hip: 32 x 64 tiles for the forward / dX products of a small batch),
emulated wave by wave in numpy and checked against oracle/mlp_oracle.py.

No kernel runs here.  The emulation follows the kernels' register / LDS layout statement by statement -- which lane supplies which
element of which v_mfma_f32_16x16x4_f32 operand, which reduction index an MFMA of a K step contracts, which (row, column) an
accumulator register stands for, the column permutation of the backward kernel's product 1, the dropout block of a lane -- so an
indexing mistake in the design shows up on the CPU box; tests/test_gpu_tower_tail.py and tests/test_gpu_mlp.py hold the compiled
kernels to the same oracle.
MFMA semantics (ISA): lane l supplies A[l % 16][l // 16] and B[l // 16][l % 16]; register r of lane l holds D[4 (l // 16) + r][l % 16]."""
import numpy as np
import pytest

from oracle import mlp_oracle as MO

LANE = np.arange(64)
I_, G_ = LANE % 16, LANE // 16


def mma16(a, b, acc):
    """acc [64, 4] += the 16 x 16 x 4 product of per-lane operands a, b [64]"""
    A = a.reshape(4, 16).T          # A[m][k] = a[m + 16 k]
    Bm = b.reshape(4, 16)           # B[k][n] = b[n + 16 k]
    Dm = A @ Bm
    out = acc.copy()
    for r in range(4):
        out[:, r] += Dm[4 * G_ + r, I_]
    return out


def emulate_tail_fwd(X, W2, b2, w3, b3, keep):
    """tower_tail_fwd_kernel: a workgroup per 16 rows, wave (w, kh): column slab w, half kh of the reduction"""
    M, K = X.shape
    N2 = W2.shape[0]
    H2 = np.zeros((M, N2))
    z = np.zeros(M)
    for blk in range((M + 15) // 16):
        r0 = 16 * blk
        zs = np.zeros((4, 16))
        half = {}
        for wave in range(8):
            w, kh = wave & 3, wave >> 2
            if 16 * w >= N2:
                continue
            row = np.minimum(r0 + I_, M - 1)
            steps = K // 16
            per = (steps + 1) // 2
            acc = np.zeros((64, 4))
            for s in range(kh * per, min(steps, kh * per + per)):
                xa = np.stack([X[row, 16 * s + 4 * G_ + e] for e in range(4)], -1)               # the lane's float4 of X
                wb = np.stack([W2[16 * w + I_, 16 * s + 4 * G_ + e] for e in range(4)], -1)      # ... and of W2
                for e in range(4):
                    acc = mma16(xa[:, e], wb[:, e], acc)
            half[(w, kh)] = acc
        for w in range(4):
            if 16 * w >= N2:
                continue
            acc = half[(w, 0)] + half[(w, 1)]
            n = 16 * w + I_
            zp = np.zeros((64, 4))
            for r in range(4):
                m = r0 + 4 * G_ + r
                v = np.maximum(acc[:, r] + (b2[n] if b2 is not None else 0.0), 0.0)
                if keep is not None:                        # block (m >> 2, n), word m & 3 = r: the lane's four rows share it
                    v = v * keep[np.minimum(m, M - 1), n]
                ok = m < M
                H2[m[ok], n[ok]] = v[ok]
                zp[:, r] = v * w3[n]
            for r in range(4):
                for g in range(4):
                    zs[w, 4 * g + r] = zp[(G_ == g), r].sum()        # the 16-lane DPP sum
        for t in range(16):
            if r0 + t < M:
                z[r0 + t] = sum(zs[w, t] for w in range(4) if 16 * w < N2) + (b3[0] if b3 is not None else 0.0)
    return H2, z


def emulate_tail_bwd(X, W2, w3, H2, dz, scale2, x_act, x_scale, n_wgs):
    """tower_tail_bwd_kernel: the row block's dZ2 and X in LDS; product 1 on 64-column groups with MFMA column i standing for column
    64 J + 4 i + c; product 2 accumulated across the workgroup's row blocks; per-workgroup partials summed in workgroup order"""
    M, K = X.shape
    N2 = W2.shape[0]
    NA, KS = N2 // 16, K // 64
    dX = np.zeros((M, K))
    pW2 = np.zeros((n_wgs, N2, K))
    pvec = np.zeros((n_wgs, 2 * N2 + 1))
    n_blocks = (M + 15) // 16
    for wg in range(n_wgs):
        for blk in range(wg, n_blocks, n_wgs):
            r0 = 16 * blk
            Zs, Xs = np.zeros((16, N2)), np.zeros((16, K))
            for m in range(16):
                if r0 + m < M:
                    Xs[m] = X[r0 + m]
                    h = H2[r0 + m]
                    Zs[m] = np.where(h > 0, dz[r0 + m] * w3 * scale2, 0.0)
                    pvec[wg, N2:2 * N2] += dz[r0 + m] * h
                    pvec[wg, :N2] += Zs[m]
                    pvec[wg, 2 * N2] += dz[r0 + m]
            for wave in range(4):
                za = [[Zs[I_, 16 * q + 4 * G_ + e] for e in range(4)] for q in range(NA)]           # A(row i, red 16 q + 4 g + e)
                for J in range(wave, KS, 4):
                    c = [np.zeros((64, 4)) for _ in range(4)]
                    for q in range(NA):
                        for e in range(4):
                            wv = np.stack([W2[16 * q + 4 * G_ + e, 64 * J + 4 * I_ + cc] for cc in range(4)], -1)   # the lane's float4
                            for cc in range(4):
                                c[cc] = mma16(za[q][e], wv[:, cc], c[cc])
                    for r in range(4):
                        for cc in range(4):
                            m, col = r0 + 4 * G_ + r, 64 * J + 4 * I_ + cc
                            v = c[cc][:, r]
                            if x_act:
                                v = np.where(Xs[4 * G_ + r, col] > 0, v * x_scale, 0.0)
                            ok = m < M
                            dX[m[ok], col[ok]] = v[ok]
                acc2 = np.zeros((KS, NA, 64, 4))
                for s in range(4):
                    av = [Zs[4 * s + G_, 16 * q + I_] for q in range(NA)]                           # A(row n, red m = 4 s + g)
                    bv = [Xs[4 * s + G_, 16 * (wave + 4 * j) + I_] for j in range(KS)]              # B(red m, col k)
                    for j in range(KS):
                        for q in range(NA):
                            acc2[j, q] = mma16(av[q], bv[j], acc2[j, q])
                for j in range(KS):
                    for q in range(NA):
                        for r in range(4):
                            pW2[wg, 16 * q + 4 * G_ + r, 16 * (wave + 4 * j) + I_] += acc2[j, q][:, r]
    return dX, pW2.sum(0), pvec[:, :N2].sum(0), pvec[:, N2:2 * N2].sum(0), pvec[:, 2 * N2].sum()


@pytest.mark.parametrize("M,K,N2,p2,x_act", [(37, 128, 32, 0.5, True), (16, 64, 16, 0.0, False), (70, 256, 64, 0.25, True)])
def test_tower_tail_lane_algebra_matches_the_oracle(M, K, N2, p2, x_act):
    rng = np.random.default_rng(M + K)
    X = rng.normal(0, 0.5, (M, K))
    if x_act:
        X = np.where(rng.random((M, K)) < 0.4, 0.0, np.abs(X))
    W2, b2 = rng.normal(0, 0.1, (N2, K)), rng.normal(0, 0.1, N2)
    W3, b3 = rng.normal(0, 0.3, (1, N2)), rng.normal(0, 0.1, 1)
    dz = rng.normal(0, 1.0, (M, 1))
    keep = MO.dropout_keep(12345 + (M << 35), 1, M, N2, p2).astype(np.float64) if p2 > 0 else None
    H2o, _ = MO.linear_fwd(X.astype(np.float32), W2.astype(np.float32), b2.astype(np.float32), True, keep)
    zo, _ = MO.linear_fwd(H2o, W3.astype(np.float32), b3.astype(np.float32), False)
    H2, z = emulate_tail_fwd(X, W2, b2, W3[0], b3, keep)
    assert np.allclose(H2, H2o, rtol=1e-5, atol=1e-6) and np.allclose(z, zo[:, 0], rtol=1e-5, atol=1e-6)
    x_scale = 1.0 / (1.0 - 0.2)
    dZ2o, dW3o, db3o = MO.linear_bwd_chain(H2o, W3.astype(np.float32), dz.astype(np.float32), x_mask=H2o > 0, x_scale=1.0 / (1.0 - p2))
    dXo, dW2o, db2o = MO.linear_bwd_chain(X.astype(np.float32), W2.astype(np.float32), dZ2o, x_mask=(X > 0) if x_act else None, x_scale=x_scale)
    dX, dW2, db2, dw3, db3 = emulate_tail_bwd(X, W2, W3[0], H2o.astype(np.float64), dz[:, 0], 1.0 / (1.0 - p2), x_act, x_scale, n_wgs=3)
    for got, want in ((dX, dXo), (dW2, dW2o), (db2, db2o), (dw3, dW3o[0]), (db3, db3o[0])):
        assert np.allclose(got, want, rtol=2e-5, atol=2e-5 * float(np.abs(want).max())), float(np.abs(got - want).max())


def emulate_gemm_small(A, B, bkm):
    """mlp_gemm_small_kernel: workgroup tile 32 x 64, wave (wr, wc) 16 x 32 as two MFMA column blocks; K step 32 = two groups of 16,
    MFMA e of group u contracts k = 16 u + 4 g + e.  A [M, K] reduction-contiguous; B [N, K] (bkm) or [K, N]."""
    M, K = A.shape
    N = B.shape[0] if bkm else B.shape[1]
    C = np.zeros((M, N))
    for bx in range((M + 31) // 32):
        for by in range((N + 63) // 64):
            m0, n0 = 32 * bx, 64 * by
            acc = np.zeros((4, 2, 64, 4))
            for step in range(K // 32):
                k0 = 32 * step
                As = np.zeros((32, 32))
                for t in range(256):                              # thread t stages row t / 8, k 4 (t % 8) ..
                    As[t >> 3, 4 * (t & 7):4 * (t & 7) + 4] = A[min(m0 + (t >> 3), M - 1), k0 + 4 * (t & 7):k0 + 4 * (t & 7) + 4]
                if bkm:
                    Bs = np.zeros((64, 32))
                    for t in range(256):
                        for q in range(2):
                            Bs[(t >> 3) + 32 * q, 4 * (t & 7):4 * (t & 7) + 4] = B[min(n0 + (t >> 3) + 32 * q, N - 1), k0 + 4 * (t & 7):k0 + 4 * (t & 7) + 4]
                else:
                    Bs = np.zeros((32, 64))
                    for t in range(256):
                        col = n0 + 4 * (t & 15)
                        for q in range(2):
                            if col < N:
                                Bs[(t >> 4) + 16 * q, 4 * (t & 15):4 * (t & 15) + 4] = B[k0 + (t >> 4) + 16 * q, col:col + 4]
                for wave in range(4):
                    wr, wc = wave >> 1, wave & 1
                    for u in range(2):
                        av = np.stack([As[16 * wr + I_, 16 * u + 4 * G_ + e] for e in range(4)], -1)
                        for cb in range(2):
                            if bkm:
                                bv = np.stack([Bs[32 * wc + 16 * cb + I_, 16 * u + 4 * G_ + e] for e in range(4)], -1)
                            else:
                                bv = np.stack([Bs[16 * u + 4 * G_ + e, 32 * wc + 16 * cb + I_] for e in range(4)], -1)
                            for e in range(4):
                                acc[wave, cb] = mma16(av[:, e], bv[:, e], acc[wave, cb])
            for wave in range(4):
                wr, wc = wave >> 1, wave & 1
                for cb in range(2):
                    for r in range(4):
                        m, j = m0 + 16 * wr + 4 * G_ + r, n0 + 32 * wc + 16 * cb + I_
                        ok = (m < M) & (j < N)
                        C[m[ok], j[ok]] = acc[wave, cb][ok, r]
    return C


@pytest.mark.parametrize("M,N,K", [(40, 72, 64), (33, 64, 96), (7, 20, 128)])
def test_small_gemm_lane_algebra(M, N, K):
    rng = np.random.default_rng(M * N)
    A = rng.normal(size=(M, K))
    W = rng.normal(size=(N, K))
    assert np.allclose(emulate_gemm_small(A, W, True), A @ W.T)               # forward: Y = X W^T
    Wt = rng.normal(size=(K, N))                                              # dX = dZ W: B stored [reduction][column]
    assert np.allclose(emulate_gemm_small(A, Wt, False), A @ Wt)
