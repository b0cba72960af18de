"""CPU ORACLE -- test infrastructure, not product code (see oracle/bprmf_oracle.py header).

Numpy restatement of the reference's NeuMF head (models/general/NeuMF.py:56-76) for a single
hidden MLP layer stack, forward and the gradients autograd derives, fp32.  Pinned against the
reference itself through tests/golden/neumf_*.npz (tests/golden/make_golden_neumf.py).
"""
import numpy as np

F32 = np.float32


def forward(P, uid, iid):
    """P: dict of numpy params named like the reference's state_dict:
    mf_u_embeddings.weight, mf_i_embeddings.weight, mlp_u_embeddings.weight, mlp_i_embeddings.weight,
    mlp.<k>.weight [out,in], mlp.<k>.bias, prediction.weight [1, L_last + d].
    NeuMF.py:61-75 (dropout p = 0): u ids tiled over candidates, 4 gathers, GMF product, MLP with
    ReLU after every Linear, prediction = Linear([mf ; mlp], 1, bias=False)."""
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
        acts.append(h)
        k += 1
    out = np.concatenate([mf, h], axis=-1)
    pred = (out @ P["prediction.weight"].T)[..., 0].astype(F32)
    return pred, dict(mf_u=mf_u, mf_i=mf_i, mf=mf, acts=acts, out=out, u=u)


def backward(P, uid, iid, gpred):
    """gradients of sum(gpred * pred) w.r.t. every parameter; embedding grads are DENSE tables
    (aten::embedding_dense_backward semantics).  Returns dict with the reference's param names."""
    pred, c = forward(P, uid, iid)
    d = P["mf_u_embeddings.weight"].shape[1]
    g = gpred.astype(F32)[..., None]
    wout = P["prediction.weight"][0]
    G = {"prediction.weight": (g * c["out"]).reshape(-1, c["out"].shape[-1]).sum(0, dtype=F32)[None]}
    d_mf = g * wout[:d]
    dh = g * wout[d:]
    n_layers = len(c["acts"]) - 1
    for k in range(n_layers - 1, -1, -1):
        dz = dh * (c["acts"][k + 1] > 0)
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
