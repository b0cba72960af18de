"""CPU ORACLE -- test infrastructure, not product code.

A plain-numpy restatement of the reference's BPRMF training arithmetic.  Only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this module; the
product path (rechorus_amd/) never does.

Pinning status: the reference (THUwangcy/ReChorus) ships NO tests or golden vectors for this
path, so this oracle is pinned against outputs of the reference ITSELF,
generated in the build container by importing /root/reference/src (tests/golden/make_golden.py
-> tests/golden/*.npz) and checked in tests/test_oracle_golden.py.

Every function cites the reference lines it follows (paths relative to the reference's src/).
All arithmetic is fp32 unless a comment says otherwise (the reference is fp32 end to end).
"""
import numpy as np

F32 = np.float32


# ---- forward ----------------------------------------------------------------------------

def gather_dot(U, I, uid, iid):
    """models/general/BPRMF.py:39-42
        cf_u_vectors = self.u_embeddings(u_ids); cf_i_vectors = self.i_embeddings(i_ids)
        prediction = (cf_u_vectors[:, None, :] * cf_i_vectors).sum(dim=-1)
    """
    u = U[uid].astype(F32)            # [B, d]
    i = I[iid].astype(F32)            # [B, C, d]
    return (u[:, None, :] * i).sum(axis=-1, dtype=F32)


def _softmax_rows(x):
    x = x - x.max(axis=1, keepdims=True)
    e = np.exp(x, dtype=F32)
    return e / e.sum(axis=1, keepdims=True, dtype=F32)


def _sigmoid(x):
    return (F32(1.0) / (F32(1.0) + np.exp(-x, dtype=F32))).astype(F32)


def bpr_loss_rows(pred):
    """models/BaseModel.py:182-185, per-row value before `.mean()`:
        pos_pred, neg_pred = predictions[:, 0], predictions[:, 1:]
        neg_softmax = (neg_pred - neg_pred.max()).softmax(dim=1)
        loss = -(((pos_pred[:, None] - neg_pred).sigmoid() * neg_softmax).sum(dim=1))
                 .clamp(min=1e-8, max=1-1e-8).log().mean()
    `neg_pred.max()` is the GLOBAL max (Appendix B-1 of SURVEY.md); softmax then subtracts
    the row max again, exactly as torch's softmax kernel does.
    """
    pred = pred.astype(F32)
    pos, neg = pred[:, 0], pred[:, 1:]
    w = _softmax_rows(neg - neg.max())
    s = _sigmoid(pos[:, None] - neg)
    P = (s * w).sum(axis=1, dtype=F32)
    lo, hi = F32(1e-8), F32(1 - 1e-8)     # hi rounds to 1.0f, as in torch
    return -np.log(np.clip(P, lo, hi), dtype=F32), P, w, s


def bpr_loss(pred):
    """models/BaseModel.py:185 -> scalar mean loss."""
    rows, _, _, _ = bpr_loss_rows(pred)
    return F32(rows.mean(dtype=F32))


def bpr_loss_grad(pred, inv_b=None):
    """d(mean loss)/d pred -- what autograd derives from models/BaseModel.py:182-185
    (softmax, sigmoid, mul, sum, clamp, log, mean backward).  Closed form:
        dL/dP = -1/(B*P) inside the clamp range (clamp backward: lo <= P <= hi), else 0
        dP/dpos   = sum_k w_k s_k (1-s_k)
        dP/dneg_j = -w_j s_j (1-s_j) + w_j (s_j - P)
    """
    B = pred.shape[0]
    if inv_b is None:
        inv_b = 1.0 / B
    _, P, w, s = bpr_loss_rows(pred)
    lo, hi = F32(1e-8), F32(1 - 1e-8)
    Pc = np.clip(P, lo, hi)
    dLdP = np.where((P >= lo) & (P <= hi), -F32(inv_b) / Pc, F32(0)).astype(F32)
    g = np.empty_like(pred, dtype=F32)
    ss = s * (F32(1) - s)
    g[:, 0] = dLdP * (w * ss).sum(axis=1, dtype=F32)
    g[:, 1:] = dLdP[:, None] * (w * ((s - P[:, None]) - ss))
    return g


# ---- backward to the tables (autograd of BPRMF.py:39-42) -----------------------------------

def bprmf_row_grads(U, I, uid, iid, gpred):
    """MulBackward0 / SumBackward1 of models/general/BPRMF.py:42:
        d/d i_vectors[b,c,:] = g[b,c] * u_vectors[b,:]
        d/d u_vectors[b,:]   = sum_c g[b,c] * i_vectors[b,c,:]
    """
    u = U[uid].astype(F32)
    i = I[iid].astype(F32)
    gi = gpred[:, :, None].astype(F32) * u[:, None, :]        # [B, C, d]
    gu = (gpred[:, :, None].astype(F32) * i).sum(axis=1, dtype=F32)   # [B, d]
    return gu, gi


def embedding_dense_backward(grad_rows, ids, n_rows):
    """EmbeddingBackward0 -> aten::embedding_dense_backward: zero-filled [n_rows, d] grad,
    every occurrence index_add-ed (duplicates accumulate); reached from loss.backward(),
    helpers/BaseRunner.py:205.  Accumulated in occurrence order (np.add.at)."""
    d = grad_rows.shape[-1]
    G = np.zeros((n_rows, d), dtype=F32)
    np.add.at(G, ids.reshape(-1), grad_rows.reshape(-1, d).astype(F32))
    return G


def bprmf_dense_grads(U, I, uid, iid, inv_b=None):
    """loss + dense grads of both tables for one batch (forward + loss.backward())."""
    pred = gather_dot(U, I, uid, iid)
    g = bpr_loss_grad(pred, inv_b)
    gu, gi = bprmf_row_grads(U, I, uid, iid, g)
    GU = embedding_dense_backward(gu, uid, U.shape[0])
    GI = embedding_dense_backward(gi, iid, I.shape[0])
    return bpr_loss(pred), pred, g, GU, GI


# ---- optimizers (torch.optim, built at helpers/BaseRunner.py:110-114, stepped at :206) ---------

def opt_step_dense(W, G, state, opt, lr, l2=0.0, beta1=0.9, beta2=0.999, eps=None, step=1,
                   rows=None):
    """One torch.optim step on W (in place) with dense grad G.  `rows` restricts the update
    to those rows (the engine's row-wise / "lazy" mode: untouched rows keep w, m, v).

    SGD     (torch/optim/sgd.py, momentum 0):   g += l2*w;  w += -lr*g
    Adam    (torch/optim/adam.py, amsgrad off): g += l2*w;  m.lerp_(g, 1-b1);
            v = b2*v + (1-b2)*g*g;  denom = sqrt(v)/sqrt(1-b2^t) + eps;  w += -(lr/(1-b1^t)) * m/denom
    Adagrad (torch/optim/adagrad.py, lr_decay 0): g += l2*w;  s += g*g;  w += -lr * g/(sqrt(s)+eps)
    Scalars are python doubles narrowed to fp32 at the tensor op, as torch does.
    """
    sl = slice(None) if rows is None else rows
    w = W[sl]
    g = G[sl].astype(F32) + F32(l2) * w
    if opt == "SGD":
        W[sl] = w + F32(-lr) * g
    elif opt == "Adam":
        if eps is None:
            eps = 1e-8
        m, v = state["m"][sl], state["v"][sl]
        m = m + F32(1 - beta1) * (g - m)
        v = v * F32(beta2) + F32(1 - beta2) * g * g
        bc1 = 1.0 - beta1 ** step
        bc2 = 1.0 - beta2 ** step
        denom = np.sqrt(v, dtype=F32) / F32(bc2 ** 0.5) + F32(eps)
        W[sl] = w + F32(-(lr / bc1)) * (m / denom)
        state["m"][sl], state["v"][sl] = m, v
    elif opt == "Adagrad":
        if eps is None:
            eps = 1e-10
        s = state["m"][sl] + g * g
        W[sl] = w + F32(-lr) * (g / (np.sqrt(s, dtype=F32) + F32(eps)))
        state["m"][sl] = s
    elif opt == "Adadelta":  # torch/optim/adadelta.py (rho 0.9, eps 1e-6): m = square_avg, v = acc_delta
        rho = beta1  # Adadelta's rho travels in beta1 (default 0.9)
        if eps is None:
            eps = 1e-6
        sq = state["m"][sl] * F32(rho) + F32(1 - rho) * g * g
        std = np.sqrt(sq + F32(eps), dtype=F32)
        delta = np.sqrt(state["v"][sl] + F32(eps), dtype=F32) / std * g
        state["v"][sl] = state["v"][sl] * F32(rho) + F32(1 - rho) * delta * delta
        state["m"][sl] = sq
        W[sl] = w + F32(-lr) * delta
    else:
        raise ValueError("Undefined optimizer: {}".format(opt))


def new_state(W, opt):
    if opt in ("Adam", "Adadelta"):
        return {"m": np.zeros_like(W), "v": np.zeros_like(W)}
    if opt == "Adagrad":
        return {"m": np.zeros_like(W)}
    return {}


def bprmf_train_step(U, I, sU, sI, uid, iid, opt="SGD", lr=1e-3, l2=0.0, step=1, rowwise=True,
                     inv_b=None, **kw):
    """One helpers/BaseRunner.py:193-206 iteration (zero_grad, forward, loss, backward,
    optimizer.step) on numpy tables, in place.  rowwise=False is the reference's dense
    semantics (every row of every table is stepped); rowwise=True updates
    only rows present in the batch (the engine's large-table mode)."""
    loss, pred, g, GU, GI = bprmf_dense_grads(U, I, uid, iid, inv_b)
    ru = np.unique(uid) if rowwise else None
    ri = np.unique(iid) if rowwise else None
    opt_step_dense(I, GI, sI, opt, lr, l2, step=step, rows=ri, **kw)
    opt_step_dense(U, GU, sU, opt, lr, l2, step=step, rows=ru, **kw)
    return loss, pred


# ---- ranking metrics (helpers/BaseRunner.py:52-78) -------------------------------------------

def evaluate_method(predictions, topk, metrics):
    """gt_rank = (pred >= pred[:,0]).sum(-1); HR@k = mean(rank<=k); NDCG@k = mean(hit/log2(rank+1))."""
    out = {}
    gt_rank = (predictions >= predictions[:, 0].reshape(-1, 1)).sum(axis=-1)
    for k in topk:
        hit = gt_rank <= k
        for metric in metrics:
            key = "{}@{}".format(metric, k)
            if metric == "HR":
                out[key] = hit.mean()
            elif metric == "NDCG":
                out[key] = (hit / np.log2(gt_rank + 1)).mean()
            else:
                raise ValueError("Undefined evaluation metric: {}.".format(metric))
    return out
