"""CPU ORACLE -- test infrastructure, not product code (see oracle/bprmf_oracle.py header).

Numpy restatement of the reference's SASRec forward (models/sequential/SASRec.py:51-86 with
utils/layers.py TransformerLayer :92-118 and MultiHeadAttention :9-63) and of the gradients
autograd derives from it, fp32 storage with float64 accumulation inside matmuls (the checker
should be at least as accurate as the thing checked).  Training-mode dropout (utils/layers.py:104-117)
enters as explicit keep-and-scale masks (`drop`, from dropout_keep: the counter scheme of
rc_sasrec_batch_fwd_dropout); None = eval mode / p = 0.
Pinned against the reference itself: tests/golden/sasrec_*.npz, sasrecdrop_*.npz (make_golden_sasrec.py).

Parameter names are the reference's state_dict keys:
  i_embeddings.weight, p_embeddings.weight,
  transformer_block.<l>.masked_attn_head.{q,k,v}_linear.{weight,bias},
  transformer_block.<l>.layer_norm{1,2}.{weight,bias}, transformer_block.<l>.linear{1,2}.{weight,bias}
"""
import numpy as np

from . import sampler_oracle

F32 = np.float32
LN_EPS = 1e-5  # nn.LayerNorm default


def dropout_keep(seed, lengths, L, d, n_layers_, p):
    """keep-and-scale factors of rc_sasrec_batch_fwd_dropout (include/rechorus_hip.h) as a list of 2 * n_layers arrays
    [B, L, d] (site 2l = dropout1 of layer l, 2l + 1 = dropout2): element (b, i, f) with i < min(len_b, L) lives in
    compact row r = sum_{b' < b} min(len_b', L) + i and is dropped iff word (f & 3) of
    Philox4x32-10(key = seed, counter = (r, site * d/4 + (f >> 2))) < p * 2^32.  Padded positions get 1 (their
    outputs never reach a valid row: SASRec.py:69-74)."""
    lens = np.minimum(np.asarray(lengths, dtype=np.int64), L)
    off = np.concatenate([[0], np.cumsum(lens)])
    R = int(off[-1])
    thresh = np.uint32(int(float(F32(p)) * 4294967296.0))
    scale = F32(1) / (F32(1) - F32(p))
    out = []
    for site in range(2 * n_layers_):
        r = np.arange(R, dtype=np.uint64)[:, None] + np.zeros((1, d // 4), dtype=np.uint64)
        blk = np.zeros((R, 1), dtype=np.uint32) + (np.uint32(site * (d // 4)) + np.arange(d // 4, dtype=np.uint32))[None]
        ctr = np.stack([(r & sampler_oracle.MASK32).astype(np.uint32), (r >> np.uint64(32)).astype(np.uint32), blk,
                        np.zeros_like(blk)], axis=-1)
        key = np.broadcast_to(np.array([seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF], dtype=np.uint32), r.shape + (2,))
        words = sampler_oracle.philox4x32_10(ctr, key).reshape(R, d)
        rows = np.where(words < thresh, F32(0), scale).astype(F32)
        keep = np.ones((len(lens), L, d), dtype=F32)
        for b in range(len(lens)):
            keep[b, :lens[b]] = rows[off[b]:off[b + 1]]
        out.append(keep)
    return out


def _lin(x, W, b):
    return (x.astype(np.float64) @ W.T.astype(np.float64) + b).astype(F32)


def _ln_fwd(z, g, b):
    z64 = z.astype(np.float64)
    mu = z64.mean(-1, keepdims=True)
    var = ((z64 - mu) ** 2).mean(-1, keepdims=True)
    rstd = 1.0 / np.sqrt(var + LN_EPS)
    xhat = (z64 - mu) * rstd
    return (xhat * g + b).astype(F32), xhat, rstd


def _ln_bwd(dy, xhat, rstd, g):
    dxhat = dy.astype(np.float64) * g
    d = xhat.shape[-1]
    dz = rstd * (dxhat - dxhat.mean(-1, keepdims=True) - xhat * (dxhat * xhat).mean(-1, keepdims=True))
    return dz, (dy * xhat).reshape(-1, d).sum(0), dy.reshape(-1, d).sum(0)


def n_layers(P):
    k = 0
    while "transformer_block.%d.linear1.weight" % k in P:
        k += 1
    return k


def forward(P, history, lengths, iid, n_heads, keep=False, drop=None):
    """history [B, L] (right padded with 0), lengths [B], iid [B, C] -> prediction [B, C];
    drop: None or the 2 * n_layers keep-and-scale masks of dropout_keep (training mode)"""
    B, L = history.shape
    d = P["i_embeddings.weight"].shape[1]
    dk = d // n_heads
    valid = (history > 0)
    position = (lengths[:, None] - np.arange(L)[None, :]) * valid          # SASRec.py:64
    x = (P["i_embeddings.weight"][history] + P["p_embeddings.weight"][position]).astype(F32)
    mask = np.tril(np.ones((L, L), dtype=bool))                            # causal only, :69-70
    cache = {"valid": valid, "position": position, "layers": []}
    for l in range(n_layers(P)):
        pre = "transformer_block.%d." % l
        a = pre + "masked_attn_head."
        q = _lin(x, P[a + "q_linear.weight"], P[a + "q_linear.bias"])
        k = _lin(x, P[a + "k_linear.weight"], P[a + "k_linear.bias"])
        v = _lin(x, P[a + "v_linear.weight"], P[a + "v_linear.bias"])
        split = lambda t: t.reshape(B, L, n_heads, dk).transpose(0, 2, 1, 3)   # layers.py:30-32
        qh, kh, vh = split(q), split(k), split(v)
        s = (qh.astype(np.float64) @ kh.transpose(0, 1, 3, 2)) / dk ** 0.5      # :57
        s = np.where(mask, s, -np.inf)
        s = s - s.max()                                                       # global max, :60
        e = np.exp(s - s.max(-1, keepdims=True))
        att = e / e.sum(-1, keepdims=True)
        att = np.where(np.isnan(att), 0.0, att)                               # :61
        ctx = (att @ vh).transpose(0, 2, 1, 3).reshape(B, L, d).astype(F32)   # no output projection
        if drop is not None:
            ctx = (ctx * drop[2 * l]).astype(F32)                              # dropout1, layers.py:110
        y1, xh1, rs1 = _ln_fwd(ctx + x, P[pre + "layer_norm1.weight"], P[pre + "layer_norm1.bias"])
        hpre = _lin(y1, P[pre + "linear1.weight"], P[pre + "linear1.bias"])
        h = np.maximum(hpre, 0)
        f = _lin(h, P[pre + "linear2.weight"], P[pre + "linear2.bias"])
        if drop is not None:
            f = (f * drop[2 * l + 1]).astype(F32)                              # dropout2, layers.py:117
        x2, xh2, rs2 = _ln_fwd(f + y1, P[pre + "layer_norm2.weight"], P[pre + "layer_norm2.bias"])
        cache["layers"].append(dict(x=x, qh=qh, kh=kh, vh=vh, att=att, y1=y1, xh1=xh1, rs1=rs1, h=h, xh2=xh2, rs2=rs2))
        x = x2
    x = x * valid[:, :, None]                                                 # :74
    hv = x[np.arange(B), lengths - 1]                                          # :76
    iv = P["i_embeddings.weight"][iid]
    pred = (hv[:, None, :].astype(np.float64) * iv).sum(-1).astype(F32)        # :80-81
    if keep:
        cache.update(hv=hv, iv=iv, xlast=x)
        return pred, cache
    return pred


def backward(P, history, lengths, iid, n_heads, gpred, drop=None):
    """gradients of sum(gpred * prediction) w.r.t. every parameter (dense embedding grads)"""
    B, L = history.shape
    d = P["i_embeddings.weight"].shape[1]
    dk = d // n_heads
    pred, c = forward(P, history, lengths, iid, n_heads, keep=True, drop=drop)
    G = {k: np.zeros(v.shape, dtype=np.float64) for k, v in P.items()}
    g = gpred.astype(np.float64)
    np.add.at(G["i_embeddings.weight"], iid.reshape(-1), (g[:, :, None] * c["hv"][:, None, :]).reshape(-1, d))
    dhv = (g[:, :, None] * c["iv"]).sum(1)
    dx = np.zeros((B, L, d))
    dx[np.arange(B), lengths - 1] = dhv
    dx = dx * c["valid"][:, :, None]
    for l in range(n_layers(P) - 1, -1, -1):
        pre = "transformer_block.%d." % l
        a = pre + "masked_attn_head."
        lc = c["layers"][l]
        dz2, dg, db = _ln_bwd(dx, lc["xh2"], lc["rs2"], P[pre + "layer_norm2.weight"])
        G[pre + "layer_norm2.weight"] += dg
        G[pre + "layer_norm2.bias"] += db
        df = dz2 if drop is None else dz2 * drop[2 * l + 1]                   # the FFN branch sees the mask, the residual not
        G[pre + "linear2.weight"] += df.reshape(-1, d).T @ lc["h"].reshape(-1, lc["h"].shape[-1])
        G[pre + "linear2.bias"] += df.reshape(-1, d).sum(0)
        dh = (df @ P[pre + "linear2.weight"]) * (lc["h"] > 0)
        G[pre + "linear1.weight"] += dh.reshape(-1, dh.shape[-1]).T @ lc["y1"].reshape(-1, d)
        G[pre + "linear1.bias"] += dh.reshape(-1, dh.shape[-1]).sum(0)
        dy1 = dz2 + dh @ P[pre + "linear1.weight"]
        dz1, dg, db = _ln_bwd(dy1, lc["xh1"], lc["rs1"], P[pre + "layer_norm1.weight"])
        G[pre + "layer_norm1.weight"] += dg
        G[pre + "layer_norm1.bias"] += db
        dc = dz1 if drop is None else dz1 * drop[2 * l]
        dctx = dc.reshape(B, L, n_heads, dk).transpose(0, 2, 1, 3)
        att, qh, kh, vh = lc["att"], lc["qh"].astype(np.float64), lc["kh"].astype(np.float64), lc["vh"].astype(np.float64)
        datt = dctx @ vh.transpose(0, 1, 3, 2)
        dvh = att.transpose(0, 1, 3, 2) @ dctx
        ds = att * (datt - (datt * att).sum(-1, keepdims=True)) / dk ** 0.5
        dqh, dkh = ds @ kh, ds.transpose(0, 1, 3, 2) @ qh
        merge = lambda t: t.transpose(0, 2, 1, 3).reshape(B, L, d)
        dq, dkk, dv = merge(dqh), merge(dkh), merge(dvh)
        x = lc["x"].astype(np.float64)
        dxn = dz1.copy()
        for nm, dt in (("q", dq), ("k", dkk), ("v", dv)):
            G[a + nm + "_linear.weight"] += dt.reshape(-1, d).T @ x.reshape(-1, d)
            G[a + nm + "_linear.bias"] += dt.reshape(-1, d).sum(0)
            dxn += dt @ P[a + nm + "_linear.weight"]
        dx = dxn
    np.add.at(G["i_embeddings.weight"], history.reshape(-1), dx.reshape(-1, d))
    np.add.at(G["p_embeddings.weight"], c["position"].reshape(-1), dx.reshape(-1, d))
    return pred, {k: v.astype(F32) for k, v in G.items()}
