"""CPU ORACLE -- test infrastructure, not product code (see oracle/bprmf_oracle.py header).

Numpy restatement of the reference's FM / WideDeep / DeepFM heads and the CTR loss, forward and
the gradients autograd derives, fp32:
    models/context/FM.py:44-63        field lookups, first-order term, pairwise term
    models/context/WideDeep.py:42-47  wide + MLP_Block(ReLU, output_dim=1)
    models/context/DeepFM.py:19-28    first-order + pairwise + MLP on the same field vectors
    models/BaseModel.py:259-274       nn.BCELoss on sigmoid(prediction)
Categorical fields ('*_c', '*_id': nn.Embedding) and numeric fields (any other name, e.g. MIND's
c_day_f: nn.Linear(1, d, bias=False) / nn.Linear(1, 1, bias=False) on feed_dict[f].float(),
FM.py:38-41,47-48,51-52) are both restated.  Pinned against the reference itself through
tests/golden/deepfm_*.npz (tests/golden/make_golden_deepfm.py; the deepfm_mind_* cases carry
MIND's field set with its numeric feature).
"""
import numpy as np

F32 = np.float32


def fm_second_order(V):
    """V [..., F, d] -> 0.5 * sum_k ((sum_f v)^2 - sum_f v^2)   (FM.py:61, DeepFM.py:22-23)"""
    V = V.astype(F32)
    s = V.sum(axis=-2, dtype=F32)
    return (F32(0.5) * (s * s - (V * V).sum(axis=-2, dtype=F32))).sum(axis=-1, dtype=F32)


def fm_second_order_bwd(V, gout):
    """d/dV of sum(gout * fm_second_order(V)):  gout * (sum_f v - v_f)"""
    V = V.astype(F32)
    s = V.sum(axis=-2, keepdims=True, dtype=F32)
    return (gout.astype(F32)[..., None, None] * (s - V)).astype(F32)


def bce(p, y):
    """nn.BCELoss(): -mean(y log p + (1-y) log(1-p)), logs clamped at -100 like torch"""
    p, y = p.astype(F32), y.astype(F32)
    lp = np.maximum(np.log(p), F32(-100))
    lq = np.maximum(np.log1p(-p), F32(-100))
    return F32(-(y * lp + (1 - y) * lq).mean(dtype=F32))


def bce_grad(p, y):
    """d loss / d p = (p - y) / (p (1-p)) / n, denominators clamped at 1e-12 (torch's EPSILON)"""
    p, y = p.astype(F32), y.astype(F32)
    return ((p - y) / np.maximum(p * (1 - p), F32(1e-12)) / F32(p.size)).astype(F32)


def _fields(P):
    pre = "context_embedding."
    return [k[len(pre):-len(".weight")] for k in P if k.startswith(pre)]


def _broadcast(v, n_cand):
    return v if v.ndim == 3 else np.repeat(v[:, None, :], n_cand, axis=1)


def is_categorical(f):
    """FM.py:38-41: a feature named '*_c' / '*_id' owns embedding tables, any other a Linear on its value"""
    return f.endswith("_c") or f.endswith("_id")


def _field(P, family, f, x):
    """one field of one family (FM.py:47-48 / 51-52): table rows, or Linear(1, w, bias=False) on x.float().unsqueeze(-1)"""
    W = P["%s.%s.weight" % (family, f)]
    if is_categorical(f):
        return W[x]
    return (x.astype(F32)[..., None] * W[:, 0].astype(F32)).astype(F32)      # [..., 1] @ W.T, W [w, 1]


def forward(P, feats, kind, field_order):
    """P: numpy params named like the reference's state_dict; feats: {field: int ids (or numeric values) [B] or [B, C]}
    (must include item_id [B, C]); kind in {'FM', 'WideDeep', 'DeepFM'}.  -> raw prediction [B, C]"""
    n_cand = feats["item_id"].shape[1]
    vec = [_broadcast(_field(P, "context_embedding", f, feats[f]), n_cand) for f in field_order]
    lin = [_broadcast(_field(P, "linear_embedding", f, feats[f]), n_cand) for f in field_order]
    V = np.stack(vec, axis=-2).astype(F32)                                  # [B, C, F, d]
    first = (P["overall_bias"] + np.concatenate(lin, axis=-1).sum(axis=-1, dtype=F32)).astype(F32)
    cache = dict(V=V)
    pred = first.copy()
    if kind in ("FM", "DeepFM"):
        pred = pred + fm_second_order(V)
    if kind in ("WideDeep", "DeepFM"):
        h = V.reshape(V.shape[0], V.shape[1], -1)
        acts, k = [h], 0
        lin_ids = sorted(int(n.split(".")[2]) for n in P if n.startswith("deep_layers.mlp.") and n.endswith(".weight"))
        for idx in lin_ids[:-1]:
            h = np.maximum(h @ P["deep_layers.mlp.%d.weight" % idx].T + P["deep_layers.mlp.%d.bias" % idx], 0).astype(F32)
            acts.append(h)
        last = lin_ids[-1]
        deep = (h @ P["deep_layers.mlp.%d.weight" % last].T + P["deep_layers.mlp.%d.bias" % last])[..., 0]
        pred = pred + deep.astype(F32)
        cache.update(acts=acts, lin_ids=lin_ids)
    return pred.astype(F32), cache


def backward(P, feats, kind, field_order, gpred):
    """gradients of sum(gpred * prediction) for every parameter (dense embedding-table grads)"""
    _, c = forward(P, feats, kind, field_order)
    V = c["V"]
    B, C, Fn, d = V.shape
    g = gpred.astype(F32)
    G = {"overall_bias": np.array([g.sum(dtype=F32)], dtype=F32)}
    dV = np.zeros_like(V)
    if kind in ("FM", "DeepFM"):
        dV += fm_second_order_bwd(V, g)
    if kind in ("WideDeep", "DeepFM"):
        acts, lin_ids = c["acts"], c["lin_ids"]
        last = lin_ids[-1]
        a = acts[-1].reshape(-1, acts[-1].shape[-1])
        G["deep_layers.mlp.%d.weight" % last] = (g.reshape(-1, 1).T @ a).astype(F32)
        G["deep_layers.mlp.%d.bias" % last] = np.array([g.sum(dtype=F32)], dtype=F32)
        dh = g[..., None] * P["deep_layers.mlp.%d.weight" % last][0]
        for pos in range(len(lin_ids) - 2, -1, -1):
            idx = lin_ids[pos]
            dz = (dh * (acts[pos + 1] > 0)).astype(F32)
            a = acts[pos].reshape(-1, acts[pos].shape[-1])
            G["deep_layers.mlp.%d.weight" % idx] = (dz.reshape(-1, dz.shape[-1]).T @ a).astype(F32)
            G["deep_layers.mlp.%d.bias" % idx] = dz.reshape(-1, dz.shape[-1]).sum(0, dtype=F32)
            dh = dz @ P["deep_layers.mlp.%d.weight" % idx]
        dV += dh.reshape(V.shape).astype(F32)
    for k, f in enumerate(field_order):
        ids = feats[f]
        T = np.zeros_like(P["context_embedding.%s.weight" % f], dtype=F32)
        L = np.zeros_like(P["linear_embedding.%s.weight" % f], dtype=F32)
        if not is_categorical(f):   # Linear backward: dW[:, 0] = sum_n x[n] * dV[n, k, :],  dw1 = sum_n x[n] * g[n]
            x = _broadcast(ids.astype(F32)[..., None], C)[..., 0] if ids.ndim == 1 else ids.astype(F32)
            T[:, 0] = (x[..., None] * dV[:, :, k, :]).reshape(-1, d).sum(axis=0, dtype=F32)
            L[0, 0] = (x * g).sum(dtype=F32)
        elif ids.ndim == 2:
            np.add.at(T, ids.reshape(-1), dV[:, :, k, :].reshape(-1, d))
            np.add.at(L, ids.reshape(-1), g.reshape(-1, 1))
        else:  # per-row field broadcast over candidates: its gradient sums over them
            np.add.at(T, ids, dV[:, :, k, :].sum(axis=1, dtype=F32))
            np.add.at(L, ids, g.sum(axis=1, dtype=F32)[:, None])
        G["context_embedding.%s.weight" % f] = T
        G["linear_embedding.%s.weight" % f] = L
    return G


def ctr_loss_and_grads(P, feats, labels, kind, field_order):
    """the CTR variants (FMCTR / WideDeepCTR / DeepFMCTR): p = sigmoid(prediction), BCE loss"""
    z, _ = forward(P, feats, kind, field_order)
    p = (1.0 / (1.0 + np.exp(-z.astype(np.float64)))).astype(F32).reshape(-1)
    y = labels.reshape(-1)
    loss = bce(p, y)
    gz = (bce_grad(p, y) * p * (1 - p)).astype(F32).reshape(z.shape)
    return p, loss, backward(P, feats, kind, field_order, gz)
