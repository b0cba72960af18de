"""CPU ORACLE (torch flavour) -- test infrastructure, not product code.

A restatement of the reference's BPRMF training iteration with the SAME ATen operator sequence
the reference dispatches on CPU (nn.Embedding gather, broadcast mul + sum, the `u_v` repeat,
softmax-weighted BPR loss, autograd's dense embedding backward, torch.optim over ALL rows).
`bench.py` times it on the host cores as `cpu_baseline` (kind "port"): /root/reference does not
exist on the GPU box, so the reference itself cannot be timed there.  Pinned against the
reference's own outputs in tests/test_oracle_golden.py::test_torch_port_*.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file.
"""
import time

import torch
import torch.nn as nn


def _embedding(n_rows, emb_size, block_rows=1 << 18):
    """nn.Embedding whose table is drawn normal(0, 0.01) ONCE (nn.Embedding's own normal(0, 1) draw followed by init_weights'
    second one costs tens of seconds of serial host RNG on a 10 M-row table, outside anything that is timed); tables beyond
    `block_rows` rows repeat a block of that many independently drawn rows (a parallel copy) -- the timed fit() iterations move
    the same bytes whatever the values are, and the parity tests load the reference's own parameters over these."""
    w = torch.empty((n_rows, emb_size))
    blk = torch.empty((min(n_rows, block_rows), emb_size)).normal_(mean=0.0, std=0.01)
    for r0 in range(0, n_rows, blk.shape[0]):
        n = min(blk.shape[0], n_rows - r0)
        w[r0:r0 + n].copy_(blk[:n])
    return nn.Embedding(n_rows, emb_size, _weight=w)


class BprmfTorchPort(nn.Module):
    def __init__(self, n_users, n_items, emb_size):
        super().__init__()
        # models/general/BPRMF.py:31-32; init normal(0, 0.01): models/BaseModel.py:29-35
        self.u_embeddings = _embedding(n_users, emb_size)
        self.i_embeddings = _embedding(n_items, emb_size)

    def forward(self, user_id, item_id):
        # models/general/BPRMF.py:39-45 (u_v is materialised by the reference even though
        # BPRMF.forward drops it, :43 and :61-63 -- kept so the port costs what the reference costs)
        uv = self.u_embeddings(user_id)
        iv = self.i_embeddings(item_id)
        prediction = (uv[:, None, :] * iv).sum(dim=-1)
        u_v = uv.repeat(1, item_id.shape[1]).view(item_id.shape[0], item_id.shape[1], -1)
        del u_v
        return prediction

    @staticmethod
    def loss(prediction):
        # models/BaseModel.py:182-185
        pos, neg = prediction[:, 0], prediction[:, 1:]
        w = (neg - neg.max()).softmax(dim=1)
        p = ((pos[:, None] - neg).sigmoid() * w).sum(dim=1)
        return -p.clamp(min=1e-8, max=1 - 1e-8).log().mean()

    def make_optimizer(self, name, lr, l2):
        # helpers/BaseRunner.py:110-114 with models/BaseModel.py:64-73 (no 'bias' params here)
        return getattr(torch.optim, name)([{"params": list(self.parameters())}], lr=lr, weight_decay=l2)

    def fit_step(self, optimizer, user_id, item_id):
        # helpers/BaseRunner.py:193-206 (candidate shuffle :187-202 is a no-op for a dot head)
        optimizer.zero_grad()
        pred = self(user_id, item_id)
        loss = self.loss(pred)
        loss.backward()
        optimizer.step()
        return loss.detach()


def time_port(n_users, n_items, emb_size, batches, opt="SGD", lr=1e-3, l2=0.0, seed=0):
    """Run the port over `batches` [(uid, iid) int64 CPU tensors]; returns (seconds, tuples)."""
    torch.manual_seed(seed)
    model = BprmfTorchPort(n_users, n_items, emb_size)
    optim = model.make_optimizer(opt, lr, l2)
    t0 = time.perf_counter()
    n = 0
    for uid, iid in batches:
        model.fit_step(optim, uid, iid)
        n += uid.shape[0]
    return time.perf_counter() - t0, n


class NeumfTorchPort(nn.Module):
    """NeuMF with the reference's module names (state_dict compatible) and operator sequence:
    models/general/NeuMF.py:42-54 (parameters), :56-76 (forward); loss / optimizer as BprmfTorchPort."""

    def __init__(self, n_users, n_items, emb_size, layers=(64,), dropout=0.0):
        super().__init__()
        self.mf_u_embeddings = _embedding(n_users, emb_size)
        self.mf_i_embeddings = _embedding(n_items, emb_size)
        self.mlp_u_embeddings = _embedding(n_users, emb_size)
        self.mlp_i_embeddings = _embedding(n_items, emb_size)
        self.mlp = nn.ModuleList()
        pre = 2 * emb_size
        for size in layers:
            self.mlp.append(nn.Linear(pre, size))
            pre = size
        self.dropout_layer = nn.Dropout(p=dropout)
        self.prediction = nn.Linear(pre + emb_size, 1, bias=False)
        for m in self.modules():  # init_weights: normal(0, 0.01) for weights AND biases, models/BaseModel.py:29-35
            if isinstance(m, nn.Linear):     # (the tables are drawn by _embedding)
                nn.init.normal_(m.weight, mean=0.0, std=0.01)
                if getattr(m, "bias", None) is not None:
                    nn.init.normal_(m.bias, mean=0.0, std=0.01)

    def forward(self, user_id, item_id):
        u_ids = user_id.unsqueeze(-1).repeat((1, item_id.shape[1]))  # NeuMF.py:61
        mf_u, mf_i = self.mf_u_embeddings(u_ids), self.mf_i_embeddings(item_id)
        mlp_u, mlp_i = self.mlp_u_embeddings(u_ids), self.mlp_i_embeddings(item_id)
        mf = mf_u * mf_i
        h = torch.cat([mlp_u, mlp_i], dim=-1)
        for layer in self.mlp:
            h = self.dropout_layer(layer(h).relu())
        return self.prediction(torch.cat([mf, h], dim=-1)).view(item_id.shape[0], -1)

    loss = staticmethod(BprmfTorchPort.loss)

    def make_optimizer(self, name, lr, l2):
        # models/BaseModel.py:64-73: parameters whose NAME contains 'bias' get weight_decay 0
        w = [p for n, p in self.named_parameters() if "bias" not in n]
        b = [p for n, p in self.named_parameters() if "bias" in n]
        return getattr(torch.optim, name)([{"params": w}, {"params": b, "weight_decay": 0}], lr=lr, weight_decay=l2)

    def fit_step(self, optimizer, user_id, item_id):
        optimizer.zero_grad()
        loss = self.loss(self(user_id, item_id))
        loss.backward()
        optimizer.step()
        return loss.detach()


class _MultiHeadAttention(nn.Module):
    """utils/layers.py:9-63 (kq_same False, no output projection, global-max shift before the row softmax, NaN -> 0)"""

    def __init__(self, d_model, n_heads):
        super().__init__()
        self.d_model, self.h, self.d_k = d_model, n_heads, d_model // n_heads
        self.q_linear = nn.Linear(d_model, d_model)
        self.k_linear = nn.Linear(d_model, d_model)
        self.v_linear = nn.Linear(d_model, d_model)

    def _split(self, x):
        return x.view(*x.size()[:-1], self.h, self.d_k).transpose(-2, -3)

    def forward(self, q, k, v, mask):
        shape = q.size()
        q, k, v = self._split(self.q_linear(q)), self._split(self.k_linear(k)), self._split(self.v_linear(v))
        scores = torch.matmul(q, k.transpose(-2, -1)) / self.d_k ** 0.5
        scores = scores.masked_fill(mask == 0, -float("inf"))
        scores = (scores - scores.max()).softmax(dim=-1)
        scores = scores.masked_fill(torch.isnan(scores), 0)
        return torch.matmul(scores, v).transpose(-2, -3).reshape(shape)


class _TransformerLayer(nn.Module):
    """utils/layers.py:92-118 (d_ff = d_model in SASRec, models/sequential/SASRec.py:46)"""

    def __init__(self, d_model, d_ff, n_heads, dropout):
        super().__init__()
        self.masked_attn_head = _MultiHeadAttention(d_model, n_heads)
        self.layer_norm1 = nn.LayerNorm(d_model)
        self.dropout1 = nn.Dropout(dropout)
        self.linear1 = nn.Linear(d_model, d_ff)
        self.linear2 = nn.Linear(d_ff, d_model)
        self.layer_norm2 = nn.LayerNorm(d_model)
        self.dropout2 = nn.Dropout(dropout)

    def forward(self, seq, mask):
        context = self.masked_attn_head(seq, seq, seq, mask)
        context = self.layer_norm1(self.dropout1(context) + seq)
        out = self.linear2(self.linear1(context).relu())
        return self.layer_norm2(self.dropout2(out) + context)


class SasrecTorchPort(nn.Module):
    """SASRec with the reference's module names and operator sequence: models/sequential/SASRec.py:36-49
    (parameters), :51-86 (forward)."""

    def __init__(self, n_items, emb_size, history_max, n_layers=1, n_heads=4, dropout=0.0):
        super().__init__()
        self.i_embeddings = nn.Embedding(n_items, emb_size)
        self.p_embeddings = nn.Embedding(history_max + 1, emb_size)
        self.transformer_block = nn.ModuleList(
            [_TransformerLayer(emb_size, emb_size, n_heads, dropout) for _ in range(n_layers)])
        for m in self.modules():
            if isinstance(m, (nn.Linear, nn.Embedding)):
                nn.init.normal_(m.weight, mean=0.0, std=0.01)
                if getattr(m, "bias", None) is not None:
                    nn.init.normal_(m.bias, mean=0.0, std=0.01)

    def forward(self, history, lengths, item_id):
        B, L = history.shape
        valid = (history > 0).long()
        his = self.i_embeddings(history)
        position = (lengths[:, None] - torch.arange(L)[None, :]) * valid   # most recent = 1, padding = 0
        his = his + self.p_embeddings(position)
        mask = torch.tril(torch.ones((1, 1, L, L), dtype=torch.int64))
        for block in self.transformer_block:
            his = block(his, mask)
        his = his * valid[:, :, None].float()
        his_vector = his[torch.arange(B), lengths - 1]
        return (his_vector[:, None, :] * self.i_embeddings(item_id)).sum(-1)

    loss = staticmethod(BprmfTorchPort.loss)
    make_optimizer = NeumfTorchPort.make_optimizer

    def fit_step(self, optimizer, history, lengths, item_id):
        optimizer.zero_grad()
        loss = self.loss(self(history, lengths, item_id))
        loss.backward()
        optimizer.step()
        return loss.detach()


class _MlpBlock(nn.Module):
    """utils/layers.py:201-243 with the options the context models use (ReLU, no norm): Linear, ReLU, [Dropout] per
    hidden layer, then Linear(., 1); the Sequential is called `mlp` so the state_dict keys match (`deep_layers.mlp.N.*`)"""

    def __init__(self, input_dim, hidden_units, dropout):
        super().__init__()
        layers, pre = [], input_dim
        for h in hidden_units:
            layers += [nn.Linear(pre, h), nn.ReLU()]
            if dropout > 0:
                layers.append(nn.Dropout(p=dropout))
            pre = h
        layers.append(nn.Linear(pre, 1))
        self.mlp = nn.Sequential(*layers)

    def forward(self, x):
        return self.mlp(x)


class DeepfmCtrTorchPort(nn.Module):
    """DeepFMCTR with the reference's module names and operator sequence:
    models/context/FM.py:34-57 (per field nn.Embedding(vocab, d) + nn.Embedding(vocab, 1) -- or, for a feature named neither '*_c'
    nor '*_id', nn.Linear(1, d, bias=False) + nn.Linear(1, 1, bias=False) on feed[f].float().unsqueeze(-1) --, overall_bias,
    stacking), DeepFM.py:19-28 (FM second order + linear + deep), :37-41 (sigmoid), models/BaseModel.py:259-267 (nn.BCELoss)."""

    def __init__(self, fields, feature_max, emb_size, layers=(512, 64), dropout=0.0):
        super().__init__()
        self.fields = list(fields)
        cat = self.categorical
        self.context_embedding = nn.ModuleDict({f: nn.Embedding(feature_max[f], emb_size) if cat(f) else nn.Linear(1, emb_size, bias=False)
                                                for f in self.fields})
        self.linear_embedding = nn.ModuleDict({f: nn.Embedding(feature_max[f], 1) if cat(f) else nn.Linear(1, 1, bias=False)
                                               for f in self.fields})
        self.overall_bias = nn.Parameter(torch.tensor([0.01]))
        self.deep_layers = _MlpBlock(len(self.fields) * emb_size, list(layers), dropout)
        for m in self.modules():
            if isinstance(m, (nn.Linear, nn.Embedding)):
                nn.init.normal_(m.weight, mean=0.0, std=0.01)
                if getattr(m, "bias", None) is not None:
                    nn.init.normal_(m.bias, mean=0.0, std=0.01)

    @staticmethod
    def categorical(f):
        return f.endswith("_c") or f.endswith("_id")

    def forward(self, feed):
        item_num = feed["item_id"].shape[1]
        cat = self.categorical
        vec = [self.context_embedding[f](feed[f]) if cat(f) else self.context_embedding[f](feed[f].float().unsqueeze(-1)) for f in self.fields]
        vec = torch.stack([v if v.dim() == 3 else v.unsqueeze(-2).repeat(1, item_num, 1) for v in vec], dim=-2)
        lin = [self.linear_embedding[f](feed[f]) if cat(f) else self.linear_embedding[f](feed[f].float().unsqueeze(-1)) for f in self.fields]
        lin = torch.cat([v if v.dim() == 3 else v.unsqueeze(-2).repeat(1, item_num, 1) for v in lin], dim=-1)
        lin = self.overall_bias + lin.sum(dim=-1)
        fm = 0.5 * (vec.sum(dim=-2).pow(2) - vec.pow(2).sum(dim=-2))
        deep = self.deep_layers(vec.flatten(start_dim=-2)).squeeze(dim=-1)
        return (fm.sum(dim=-1) + lin + deep).view(-1).sigmoid()

    @staticmethod
    def loss(prediction, label):
        return nn.BCELoss()(prediction, label.view(-1).float())

    make_optimizer = NeumfTorchPort.make_optimizer

    def fit_step(self, optimizer, feed):
        optimizer.zero_grad()
        loss = self.loss(self(feed), feed["label"])
        loss.backward()
        optimizer.step()
        return loss.detach()
