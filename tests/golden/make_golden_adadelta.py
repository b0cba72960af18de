"""Golden vectors for --optimizer Adadelta (the fourth optimizer the reference documents, helpers/BaseRunner.py:37-38)
FROM THE REFERENCE ITSELF: BPRMF, two fit() iterations through the reference's own _build_optimizer
(helpers/BaseRunner.py:110-114 -> torch.optim.Adadelta(lr, weight_decay=l2)) in the call order of :193-206.

Build container only:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_adadelta.py"""
import os
from types import SimpleNamespace

import numpy as np

import make_golden as MG

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    torch, BPRMF, BaseRunner = MG._import_reference()
    n_users, n_items, d, B, K, seed = 30, 200, 64, 40, 9, 21
    torch.manual_seed(seed)
    rng = np.random.default_rng(seed)
    args = SimpleNamespace(device=torch.device("cpu"), model_path="", buffer=1, num_neg=K, dropout=0, test_all=0, emb_size=d)
    corpus = SimpleNamespace(n_users=n_users, n_items=n_items)
    model = BPRMF(args, corpus)
    out = {"U0": model.u_embeddings.weight.detach().numpy().copy(), "I0": model.i_embeddings.weight.detach().numpy().copy()}
    batches = []
    for _ in range(3):
        batches.append((rng.integers(1, n_users, size=B).astype(np.int64), rng.integers(1, n_items, size=(B, 1 + K)).astype(np.int64)))
    for k, (u, i) in enumerate(batches, 1):
        out[f"uid{k}"], out[f"iid{k}"] = u, i
    for lr, l2 in ((1.0, 1e-4), (1e-3, 0.0)):   # torch's default lr, and the reference's default --lr
        m = BPRMF(args, corpus)
        with torch.no_grad():
            m.u_embeddings.weight.copy_(torch.from_numpy(out["U0"]))
            m.i_embeddings.weight.copy_(torch.from_numpy(out["I0"]))
        runner = BaseRunner(MG._runner_args(BaseRunner, "Adadelta", lr, l2))
        m.optimizer = runner._build_optimizer(m)
        assert type(m.optimizer).__name__ == "Adadelta"
        tag = "lr{:g}_l2{:g}".format(lr, l2)
        losses = []
        for step, (u, i) in enumerate(batches, 1):
            m.optimizer.zero_grad()
            od = m({"user_id": torch.from_numpy(u), "item_id": torch.from_numpy(i), "batch_size": len(u), "phase": "train"})
            ls = m.loss(od)
            ls.backward()
            m.optimizer.step()
            losses.append(ls.item())
            out[f"{tag}_U{step}"] = m.u_embeddings.weight.detach().numpy().copy()
            out[f"{tag}_I{step}"] = m.i_embeddings.weight.detach().numpy().copy()
        out[tag + "_losses"] = np.array(losses, dtype=np.float32)
    path = os.path.join(HERE, "adadelta_bprmf_d64.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) >> 10, "KiB")


if __name__ == "__main__":
    main()
