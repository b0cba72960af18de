"""CPU ORACLE -- test infrastructure, not product code (see oracle/bprmf_oracle.py header).

Numpy restatement of the reference's NeuMF head (models/general/NeuMF.py:56-76) for a single
hidden MLP layer stack, forward and the gradients autograd derives, fp32.  Pinned against the
reference itself through tests/golden/neumf_*.npz (tests/golden/make_golden_neumf.py).
"""
import numpy as np

from . import sampler_oracle

F32 = np.float32


def dropout_keep(seed, n_cand, width, p):
    """keep-and-scale factors [n_cand, width] of rc_neumf_fwd_dropout (include/rechorus_hip.h): feature f of
    candidate n is dropped iff word (f & 3) of Philox4x32-10(key = seed, counter = (n, f >> 2)) < p * 2^32.
    (nn.Dropout semantics, NeuMF.py:58/70: zero with probability p, scale the rest by 1/(1-p); torch's own
    random stream is not reproduced.)"""
    n = np.arange(n_cand, dtype=np.uint64)[:, None] + np.zeros((1, width // 4), dtype=np.uint64)
    blk = np.zeros((n_cand, 1), dtype=np.uint32) + np.arange(width // 4, dtype=np.uint32)[None]
    ctr = np.stack([(n & sampler_oracle.MASK32).astype(np.uint32), (n >> np.uint64(32)).astype(np.uint32), blk,
                    np.zeros_like(blk)], axis=-1)
    key = np.broadcast_to(np.array([seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF], dtype=np.uint32), n.shape + (2,))
    words = sampler_oracle.philox4x32_10(ctr, key).reshape(n_cand, width)
    thresh = np.uint32(int(float(F32(p)) * 4294967296.0))
    return np.where(words < thresh, F32(0), F32(1) / (F32(1) - F32(p))).astype(F32)


def forward(P, uid, iid, keep=None):
    """P: dict of numpy params named like the reference's state_dict:
    mf_u_embeddings.weight, mf_i_embeddings.weight, mlp_u_embeddings.weight, mlp_i_embeddings.weight,
    mlp.<k>.weight [out,in], mlp.<k>.bias, prediction.weight [1, L_last + d].
    NeuMF.py:61-75: u ids tiled over candidates, 4 gathers, GMF product, MLP with ReLU (+ dropout: `keep`
    [B*C, width] from dropout_keep, single hidden layer; None = eval mode / p = 0) after every Linear,
    prediction = Linear([mf ; mlp], 1, bias=False)."""
    B, C = iid.shape
    u = np.repeat(uid[:, None], C, axis=1)
    mf_u, mf_i = P["mf_u_embeddings.weight"][u], P["mf_i_embeddings.weight"][iid]
    h = np.concatenate([P["mlp_u_embeddings.weight"][u], P["mlp_i_embeddings.weight"][iid]], axis=-1).astype(F32)
    mf = (mf_u * mf_i).astype(F32)
    acts = [h]
    k = 0
    while "mlp.%d.weight" % k in P:
        z = (h @ P["mlp.%d.weight" % k].T + P["mlp.%d.bias" % k]).astype(F32)
        h = np.maximum(z, 0).astype(F32)
        if keep is not None:
            h = (h * keep.reshape(B, C, -1)).astype(F32)
        acts.append(h)
        k += 1
    out = np.concatenate([mf, h], axis=-1)
    pred = (out @ P["prediction.weight"].T)[..., 0].astype(F32)
    return pred, dict(mf_u=mf_u, mf_i=mf_i, mf=mf, acts=acts, out=out, u=u)


def backward(P, uid, iid, gpred, keep=None):
    """gradients of sum(gpred * pred) w.r.t. every parameter; embedding grads are DENSE tables
    (aten::embedding_dense_backward semantics).  Returns dict with the reference's param names."""
    pred, c = forward(P, uid, iid, keep)
    B, C = iid.shape
    d = P["mf_u_embeddings.weight"].shape[1]
    g = gpred.astype(F32)[..., None]
    wout = P["prediction.weight"][0]
    G = {"prediction.weight": (g * c["out"]).reshape(-1, c["out"].shape[-1]).sum(0, dtype=F32)[None]}
    d_mf = g * wout[:d]
    dh = g * wout[d:]
    n_layers = len(c["acts"]) - 1
    for k in range(n_layers - 1, -1, -1):
        dz = dh * (c["acts"][k + 1] > 0)  # a dropped unit has acts == 0 too
        if keep is not None:
            dz = (dz * keep.reshape(B, C, -1)).astype(F32)
        a = c["acts"][k].reshape(-1, c["acts"][k].shape[-1])
        G["mlp.%d.weight" % k] = (dz.reshape(-1, dz.shape[-1]).T @ a).astype(F32)
        G["mlp.%d.bias" % k] = dz.reshape(-1, dz.shape[-1]).sum(0, dtype=F32)
        dh = dz @ P["mlp.%d.weight" % k]
    def scatter(n_rows, ids, rows):
        T = np.zeros((n_rows, rows.shape[-1]), dtype=F32)
        np.add.at(T, ids.reshape(-1), rows.reshape(-1, rows.shape[-1]).astype(F32))
        return T
    nu, ni = P["mf_u_embeddings.weight"].shape[0], P["mf_i_embeddings.weight"].shape[0]
    G["mf_u_embeddings.weight"] = scatter(nu, c["u"], d_mf * c["mf_i"])
    G["mf_i_embeddings.weight"] = scatter(ni, iid, d_mf * c["mf_u"])
    G["mlp_u_embeddings.weight"] = scatter(nu, c["u"], dh[..., :d])
    G["mlp_i_embeddings.weight"] = scatter(ni, iid, dh[..., d:])
    return pred, G
