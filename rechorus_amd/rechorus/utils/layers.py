"""Shared torch layers of the plugin surface (mirror of the reference's utils/layers.py for the
classes the in-scope models use: MultiHeadAttention :9-63, TransformerLayer :92-118).

These modules own the PARAMETERS (same names / state_dict keys as the reference, so checkpoints
interchange) and provide a torch forward for configurations the HIP encoder does not cover
(dropout > 0, d not in {32, 64}, history longer than 64).  SASRec's default path does not call
them: it hands their parameters to rc_sasrec_fwd / rc_sasrec_bwd (rechorus_amd.nn.sasrec_encode).
"""
import numpy as np
import torch
import torch.nn as nn


class MultiHeadAttention(nn.Module):
    def __init__(self, d_model, n_heads, kq_same=False, bias=True, attention_d=-1):
        super().__init__()
        self.d_model, self.h, self.kq_same = d_model, n_heads, kq_same
        self.attention_d = d_model if attention_d < 0 else attention_d
        self.d_k = self.attention_d // n_heads
        if not kq_same:
            self.q_linear = nn.Linear(d_model, self.attention_d, bias=bias)
        self.k_linear = nn.Linear(d_model, self.attention_d, bias=bias)
        self.v_linear = nn.Linear(d_model, self.attention_d, bias=bias)

    def head_split(self, x):  # [..., L, h*d_k] -> [..., h, L, d_k]
        return x.view(*x.shape[:-1], self.h, self.d_k).transpose(-2, -3)

    def forward(self, q, k, v, mask=None):
        shape = q.shape
        q = self.head_split((self.k_linear if self.kq_same else self.q_linear)(q))
        k, v = self.head_split(self.k_linear(k)), self.head_split(self.v_linear(v))
        out = self.scaled_dot_product_attention(q, k, v, self.d_k, mask)
        return out.transpose(-2, -3).reshape(list(shape[:-1]) + [self.attention_d])  # no output projection

    @staticmethod
    def scaled_dot_product_attention(q, k, v, d_k, mask=None):
        scores = torch.matmul(q, k.transpose(-2, -1)) / d_k ** 0.5
        if mask is not None:
            scores = scores.masked_fill(mask == 0, -np.inf)
        scores = (scores - scores.max()).softmax(dim=-1)   # global max shift (a no-op for softmax)
        scores = scores.masked_fill(torch.isnan(scores), 0)  # fully masked rows -> 0
        return torch.matmul(scores, v)


class TransformerLayer(nn.Module):
    def __init__(self, d_model, d_ff, n_heads, dropout=0, kq_same=False):
        super().__init__()
        self.masked_attn_head = MultiHeadAttention(d_model, n_heads, kq_same=kq_same)
        self.layer_norm1 = nn.LayerNorm(d_model)
        self.dropout1 = nn.Dropout(dropout)
        self.linear1 = nn.Linear(d_model, d_ff)
        self.linear2 = nn.Linear(d_ff, d_model)
        self.layer_norm2 = nn.LayerNorm(d_model)
        self.dropout2 = nn.Dropout(dropout)

    def forward(self, seq, mask=None):
        ctx = self.masked_attn_head(seq, seq, seq, mask)
        ctx = self.layer_norm1(self.dropout1(ctx) + seq)
        out = self.linear2(self.linear1(ctx).relu())
        return self.layer_norm2(self.dropout2(out) + ctx)
