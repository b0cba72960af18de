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

from rechorus_amd import engine, nn as hnn


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


class Dice(nn.Module):
    """DIN's data-adaptive activation: p = sigmoid(BN(x)), out = p*x + alpha*(1-p)*x (reference :246-288)"""

    def __init__(self, emb_size, dim=2, epsilon=1e-8, device='cpu'):
        super().__init__()
        assert dim in (2, 3)
        self.dim = dim
        self.bn = nn.BatchNorm1d(emb_size, eps=epsilon)
        self.alpha = nn.Parameter(torch.zeros((emb_size,) if dim == 2 else (emb_size, 1), device=device))

    def forward(self, x):
        assert x.dim() == self.dim
        if self.dim == 3:
            x = x.transpose(1, 2)
        p = torch.sigmoid(self.bn(x))
        out = p * x + self.alpha * (1 - p) * x
        return out.transpose(1, 2) if self.dim == 3 else out


class MLP_Block(nn.Module):
    """Linear -> [norm] -> activation -> [norm] -> [dropout] per hidden layer, optional output
    layer / activation (reference :201-243; parameters live in `self.mlp`, an nn.Sequential with
    the same module order so state_dict keys `mlp.<k>.weight` interchange).  Dense GEMMs: rocBLAS."""

    def __init__(self, input_dim, hidden_units=[], hidden_activations="ReLU", output_dim=None,
                 output_activation=None, dropout_rates=0.0, batch_norm=False, layer_norm=False,
                 norm_before_activation=True, use_bias=True):
        super().__init__()
        n = len(hidden_units)
        rates = dropout_rates if isinstance(dropout_rates, list) else [dropout_rates] * n
        acts = hidden_activations if isinstance(hidden_activations, list) else [hidden_activations] * n
        widths = [input_dim] + list(hidden_units)

        def norm(width):
            if batch_norm:
                return [nn.BatchNorm1d(width)]
            return [nn.LayerNorm(width)] if layer_norm else []

        mods = []
        for k in range(n):
            act = Dice(widths[k + 1]) if acts[k] == "Dice" else (getattr(nn, acts[k])() if acts[k] else None)
            mods.append(nn.Linear(widths[k], widths[k + 1], bias=use_bias))
            if norm_before_activation:
                mods += norm(widths[k + 1])
            if act is not None:
                mods.append(act)
            if not norm_before_activation:
                mods += norm(widths[k + 1])
            if rates[k] > 0:
                mods.append(nn.Dropout(p=rates[k]))
        if output_dim is not None:
            mods.append(nn.Linear(widths[-1], output_dim, bias=use_bias))
        if output_activation is not None:
            mods.append(getattr(nn, output_activation)())
        self.mlp = nn.Sequential(*mods)
        # Linear [-> ReLU] [-> Dropout] chains (every MLP_Block the context models build) run on the engine's fp32 MFMA
        # GEMMs with bias / ReLU / dropout fused into the epilogue (rc_linear_fwd / rc_linear_bwd); the parameters stay
        # in `self.mlp` (state_dict keys `mlp.<k>.weight` unchanged).  Norm layers / Dice keep torch's modules.
        self._hip_plan = hnn.mlp_plan(mods)
        if self._hip_plan is not None and any(p > 0 for _, _, p in self._hip_plan):
            # key of the dropout mask stream; not a parameter and not in the state_dict
            self.register_buffer('drop_seed', hnn.fresh_drop_seed(), persistent=False)

    def forward(self, inputs):
        if self._hip_plan is not None and inputs.is_cuda and inputs.dtype == torch.float32:
            seed = getattr(self, 'drop_seed', None)
            if self.training and seed is not None:
                engine.step_increment(seed)  # new mask for this forward; its backward reads the saved outputs
            return hnn.mlp_forward(inputs, self._hip_plan, self.training, seed)
        return self.mlp(inputs)
