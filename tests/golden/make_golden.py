"""Generate golden vectors for the BPRMF hot path FROM THE REFERENCE ITSELF.

Run in the build container only (needs /root/reference, which does not exist on the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

It imports the unmodified reference (models.general.BPRMF.BPRMF, helpers.BaseRunner.BaseRunner)
from /root/reference/src, runs forward / loss / backward / optimizer.step on seeded inputs on
CPU, and stores inputs + outputs as small .npz fixtures next to this script.  The fixtures are
what pins oracle/bprmf_oracle.py (tests/test_oracle_golden.py) and, through it or directly, the
HIP engine (tests/test_gpu_*.py).  Nothing is written under /root/reference
(PYTHONDONTWRITEBYTECODE must be set: the tree is read-only by policy).
"""
import argparse
import os
import sys
from types import SimpleNamespace

import numpy as np

# numpy >= 1.24 removed the aliases the reference still uses (BaseModel.py:141,146 ...)
for _n, _t in (("object", object), ("int", int), ("float", float), ("bool", bool)):
    if not hasattr(np, _n):
        setattr(np, _n, _t)

REF_SRC = "/root/reference/src"
HERE = os.path.dirname(os.path.abspath(__file__))


def _import_reference():
    if not os.path.isdir(REF_SRC):
        raise SystemExit("reference not mounted at /root/reference: golden vectors can only be "
                         "regenerated in the build container")
    if not sys.dont_write_bytecode:
        raise SystemExit("set PYTHONDONTWRITEBYTECODE=1 (do not drop __pycache__ into the reference)")
    sys.path.insert(0, REF_SRC)
    import torch  # noqa
    from models.general.BPRMF import BPRMF  # noqa
    from helpers.BaseRunner import BaseRunner  # noqa
    return torch, BPRMF, BaseRunner


def _runner_args(BaseRunner, optimizer, lr, l2):
    p = argparse.ArgumentParser()
    p = BaseRunner.parse_runner_args(p)
    a, _ = p.parse_known_args([])
    a.train, a.log_file = 1, "/tmp/rechorus_golden/log.txt"
    a.optimizer, a.lr, a.l2 = optimizer, lr, l2
    return a


def make_case(name, n_users, n_items, d, B, K, seed, zipf=False):
    torch, BPRMF, BaseRunner = _import_reference()
    torch.manual_seed(seed)
    rng = np.random.default_rng(seed)
    args = SimpleNamespace(device=torch.device("cpu"), model_path="", buffer=1, num_neg=K,
                           dropout=0, test_all=0, emb_size=d)
    corpus = SimpleNamespace(n_users=n_users, n_items=n_items)

    def batch():
        if zipf:  # heavy duplication: hot users / items repeat many times in a batch
            pu = 1.0 / np.arange(1, n_users); pu /= pu.sum()
            pi = 1.0 / np.arange(1, n_items); pi /= pi.sum()
            uid = rng.choice(np.arange(1, n_users), size=B, p=pu)
            iid = rng.choice(np.arange(1, n_items), size=(B, 1 + K), p=pi)
        else:
            uid = rng.integers(1, n_users, size=B)
            iid = rng.integers(1, n_items, size=(B, 1 + K))
        return uid.astype(np.int64), iid.astype(np.int64)

    out = {"meta": np.array([n_users, n_items, d, B, K, seed], dtype=np.int64)}
    model = BPRMF(args, corpus)          # init_weights: normal(0, 0.01), BaseModel.py:29-35
    U0 = model.u_embeddings.weight.detach().numpy().copy()
    I0 = model.i_embeddings.weight.detach().numpy().copy()
    out["U0"], out["I0"] = U0, I0

    uid, iid = batch()
    uid2, iid2 = batch()
    out.update(uid=uid, iid=iid, uid2=uid2, iid2=iid2)

    def feed(u, i):
        return {"user_id": torch.from_numpy(u), "item_id": torch.from_numpy(i),
                "batch_size": len(u), "phase": "train"}

    # forward / loss / backward on the first batch (reference code, autograd)
    model.zero_grad()
    o = model(feed(uid, iid))
    pred = o["prediction"]
    pred.retain_grad()
    loss = model.loss(o)
    loss.backward()
    out["pred"] = pred.detach().numpy().copy()
    out["loss"] = np.array(loss.item(), dtype=np.float32)
    out["gpred"] = pred.grad.numpy().copy()
    out["GU"] = model.u_embeddings.weight.grad.numpy().copy()
    out["GI"] = model.i_embeddings.weight.grad.numpy().copy()

    # two optimizer steps (batch 1 then batch 2) for each optimizer, through the reference's
    # own _build_optimizer (helpers/BaseRunner.py:110-114) and the fit() call order (:193-206)
    for opt_name, lr, l2 in (("SGD", 0.05, 0.0), ("SGD", 0.05, 1e-3), ("Adam", 1e-3, 0.0),
                             ("Adam", 1e-3, 1e-4), ("Adagrad", 0.01, 1e-4)):
        torch.manual_seed(seed)
        m = BPRMF(args, corpus)
        with torch.no_grad():
            m.u_embeddings.weight.copy_(torch.from_numpy(U0))
            m.i_embeddings.weight.copy_(torch.from_numpy(I0))
        runner = BaseRunner(_runner_args(BaseRunner, opt_name, lr, l2))
        m.optimizer = runner._build_optimizer(m)
        losses = []
        tag = "{}_l2{:g}".format(opt_name, l2)
        for step, (u, i) in enumerate(((uid, iid), (uid2, iid2)), 1):
            m.optimizer.zero_grad()
            od = m(feed(u, i))
            ls = m.loss(od)
            ls.backward()
            m.optimizer.step()
            losses.append(ls.item())
            out["{}_U{}".format(tag, step)] = m.u_embeddings.weight.detach().numpy().copy()
            out["{}_I{}".format(tag, step)] = m.i_embeddings.weight.detach().numpy().copy()
        out[tag + "_losses"] = np.array(losses, dtype=np.float32)
        out[tag + "_hyper"] = np.array([lr, l2], dtype=np.float64)

    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) >> 10, "KiB")


CASES = [
    # name,            n_users, n_items, d,   B,  K,  seed, zipf
    ("bprmf_k1_d64",       40,   120,   64,  32,  1,  11, False),
    ("bprmf_k4_d64",       40,   120,   64,  24,  4,  12, False),
    ("bprmf_k99_d64",      30,   300,   64,  12, 99,  13, False),
    ("bprmf_k99_d64_zipf", 30,   300,   64,  48, 99,  14, True),   # long segments (> 32 dups)
    ("bprmf_k7_d32",       20,    60,   32,  16,  7,  15, False),
    ("bprmf_k3_d128",      20,    60,  128,  16,  3,  16, False),
    ("bprmf_k5_d48",       20,    60,   48,  10,  5,  17, False),  # emb_size not a power of two
]

if __name__ == "__main__":
    for c in CASES:
        make_case(*c)
