"""CPU ORACLE -- test infrastructure, not product code (see oracle/bprmf_oracle.py header).

Numpy restatement of the list-level BPR loss of the reference's ImpressionModel.loss
(models/BaseImpressionModel.py:50-89, loss_n 'BPR' = re-weighting "between", and 'BPRhard') with the
gradient autograd derives, and of ImpressionRunner's list metrics (helpers/ImpressionRunner.py:18-135).
Pinned against the reference itself: tests/golden/impression_losses_metrics.npz
(tests/golden/make_golden_impression.py calls the reference's own loss / evaluate_method).
"""
import numpy as np

F32 = np.float32


def _softmax(x, mask):
    x = np.where(mask, x, -np.inf)
    e = np.exp(x - x.max(axis=1, keepdims=True))
    return e / e.sum(axis=1, keepdims=True)


def list_bpr(pred, target, max_pos, hard=False):
    """-> (loss, per-row loss, d loss / d pred).  Per row: positives i = valid columns < max_pos, negatives j =
    valid columns >= max_pos; a = softmax_i(+-pred), b = softmax_j(pred)  (:66-74)
    Q = sum_i a_i sum_j b_j sigmoid(pred_i - pred_j);  loss = mean_b -log Q_b                     (:87-89)"""
    x = pred.astype(np.float64)
    B, n = x.shape
    valid = target != -1
    col = np.arange(n)[None, :]
    is_pos, is_neg = valid & (col < max_pos), valid & (col >= max_pos)
    sgn = -1.0 if hard else 1.0
    a, b = _softmax(sgn * x, is_pos), _softmax(x, is_neg)
    sig = 1.0 / (1.0 + np.exp(-(x[:, :, None] - x[:, None, :])))
    sig = sig * (is_pos[:, :, None] & is_neg[:, None, :])
    dsig = sig * (1.0 - sig)
    Qi = (sig * b[:, None, :]).sum(axis=2)        # [B, i]
    Rj = (sig * a[:, :, None]).sum(axis=1)        # [B, j]
    Q = (a * Qi).sum(axis=1)
    rows = -np.log(Q)
    dQ_pos = a * (dsig * b[:, None, :]).sum(axis=2) + sgn * a * (Qi - Q[:, None])
    dQ_neg = -b * (dsig * a[:, :, None]).sum(axis=1) + b * (Rj - Q[:, None])
    dQ = np.where(is_pos, dQ_pos, 0.0) + np.where(is_neg, dQ_neg, 0.0)
    g = -(1.0 / B) / Q[:, None] * dQ
    return F32(rows.mean()), rows.astype(F32), g.astype(F32)


def list_metrics(predictions, pos_num, neg_num, pos_num_max, topk):
    """per-row NDCG / MAP / HR @k (helpers/ImpressionRunner.py:18-135): predictions [N, n] with -inf on
    unused slots; a positive tying with a negative ranks below it (:94-96)"""
    N, n = predictions.shape
    pred = predictions.astype(np.float64).copy()
    pred[:, :pos_num_max] -= 1e-6
    order = np.argsort(-pred, axis=1, kind="mergesort")
    pos = np.minimum(np.asarray(pos_num), pos_num_max)
    neg = np.minimum(np.asarray(neg_num), n - pos_num_max)
    out = {"NDCG@%d" % k: np.zeros(N) for k in topk}
    out.update({"MAP@%d" % k: np.zeros(N) for k in topk})
    out.update({"HR@%d" % k: np.zeros(N) for k in topk})
    for r in range(N):
        length = pos[r] + neg[r]
        ranked = [1 if c < pos[r] else 0 for c in order[r][:length]]
        npos = sum(ranked)
        for k in topk:
            top = ranked[:k]
            dcg = sum(l / np.log2(p + 2) for p, l in enumerate(top))
            idcg = sum(1 / np.log2(p + 2) for p in range(min(npos, k)))
            out["NDCG@%d" % k][r] = dcg / idcg if idcg else 0.0
            hits, ap = 0, 0.0
            for p, l in enumerate(top):
                hits += l
                ap += l * hits / (p + 1)
            out["MAP@%d" % k][r] = ap / min(max(npos, 1), k)
            out["HR@%d" % k][r] = 1.0 if sum(top) > 0 else 0.0
    return out
