"""CPU ORACLE -- test infrastructure, not product code (see oracle/bprmf_oracle.py header).

Numpy restatement of the reference's list-wise softmax cross-entropy
(`ImpressionModel.loss`, loss_n == 'softmaxCE', models/BaseImpressionModel.py:96-107) and of its
gradient.  Pinned against the reference itself: tests/golden/listwise_*.npz
(tests/golden/make_golden_listwise.py calls the reference's own loss function).
"""
import numpy as np

F32 = np.float32


def softmax_ce(pred, target, max_pos):
    """pred [B, n] fp32; target [B, n] in {1, 0, -1} (-1 = padding); the first `max_pos` columns
    are the positive slots.  Returns (loss, per-row loss, have_neg).
        mask = target != -1;  have_neg = mask[:, max_pos]                       (:47-48)
        p = softmax(pred masked to -inf, row max subtracted)                    (:99-100)
        row = -(sum_{i < max_pos, mask} log p_i) / #(target == 1)               (:101-103)
        loss = mean(row * have_neg / sum(have_neg) * B) = sum(row*have_neg)/sum(have_neg)   (:105-106)
    """
    pred = pred.astype(np.float64)
    mask = target != -1
    have_neg = mask[:, max_pos].astype(np.float64)
    x = np.where(mask, pred, -np.inf)
    x = x - x.max(axis=1, keepdims=True)
    e = np.exp(x)
    p = e / e.sum(axis=1, keepdims=True)
    pos_len = (target == 1).sum(axis=1).astype(np.float64)
    tp = np.where(mask[:, :max_pos], p[:, :max_pos], 1.0)
    row = -np.log(tp).sum(axis=1) / pos_len
    loss = (row * have_neg).sum() / have_neg.sum()
    return F32(loss), row.astype(F32), have_neg


def softmax_ce_grad(pred, target, max_pos):
    """d loss / d pred: for a valid column j of row b,
    -(h_b / H) / n_b * (1[j in S_b] - |S_b| * p_j),  S_b = valid positive slots; 0 on padding."""
    pred64 = pred.astype(np.float64)
    mask = target != -1
    have_neg = mask[:, max_pos].astype(np.float64)
    x = np.where(mask, pred64, -np.inf)
    x = x - x.max(axis=1, keepdims=True)
    e = np.exp(x)
    p = e / e.sum(axis=1, keepdims=True)
    pos_len = (target == 1).sum(axis=1).astype(np.float64)
    in_s = np.zeros_like(mask)
    in_s[:, :max_pos] = mask[:, :max_pos]
    s_cnt = in_s.sum(axis=1, keepdims=True)
    g = -(have_neg / have_neg.sum() / pos_len)[:, None] * (in_s.astype(np.float64) - s_cnt * p)
    return np.where(mask, g, 0.0).astype(F32)
