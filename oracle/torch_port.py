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


class BprmfTorchPort(nn.Module):
    def __init__(self, n_users, n_items, emb_size):
        super().__init__()
        # models/general/BPRMF.py:31-32; init normal(0, 0.01): models/BaseModel.py:29-35
        self.u_embeddings = nn.Embedding(n_users, emb_size)
        self.i_embeddings = nn.Embedding(n_items, emb_size)
        nn.init.normal_(self.u_embeddings.weight, mean=0.0, std=0.01)
        nn.init.normal_(self.i_embeddings.weight, mean=0.0, std=0.01)

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
