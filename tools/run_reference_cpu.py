"""BASELINE configs[0]: the UNMODIFIED reference (BaseRunner.fit, helpers/BaseRunner.py:174-208) timed on CPU.

Runs only in the build container (needs /root/reference; nothing is written there):

    PYTHONDONTWRITEBYTECODE=1 python tools/run_reference_cpu.py [--out profiles/r02_config1_reference_cpu.json]

Legs:
  ml1m_e2e_w0 / ml1m_e2e_w5   one epoch of BaseRunner.fit on the ML-1M-like synthetic set (6,034 users, 3,125 items,
                              574,197 training rows, Zipf(1.0) item popularity; ML-1M itself is not in the container),
                              reference flags of docs/demo_scripts_results (emb_size 64, num_neg 1, batch 256, Adam
                              lr 1e-3 l2 1e-6), DataLoader num_workers 0 and 5: negative sampler + collate + compute
  ml1m_compute                the same epoch on pre-collated batches (the loop body of fit only)
  config2_compute             the loop body of fit on the bench.py batch stream of BASELINE configs[1] (B = 65,536, K = 99,
                              10 M items, 1 M users): the reference's own model / loss / autograd / torch.optim, dense
                              gradients and dense optimizer over every row, a bounded number of steps
Reports tuples/s, nproc, torch threads, torch version."""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np
import pandas as pd

for _n, _t in (("object", object), ("int", int), ("float", float), ("bool", bool)):  # numpy >= 1.24 (BaseModel.py:141,146)
    if not hasattr(np, _n):
        setattr(np, _n, _t)

REF_SRC = "/root/reference/src"


def make_ml1m_like(root, n_users=6034, n_items=3125, n_train=574_197, seed=0):
    """train/dev/test.csv in the reference's format (data/README.md): ids from 1, tab separated, dev/test carry 99
    sampled negatives.  Item popularity Zipf(1.0) over a seeded rank->id permutation, users uniform."""
    rng = np.random.default_rng(seed)
    p = 1.0 / np.arange(1, n_items + 1)
    p /= p.sum()
    perm = rng.permutation(n_items) + 1
    d = os.path.join(root, "ml1m_like")
    os.makedirs(d, exist_ok=True)
    n_eval = n_users
    users = np.concatenate([np.arange(1, n_users + 1), rng.integers(1, n_users + 1, size=n_train - n_users)])
    items = perm[rng.choice(n_items, size=n_train, p=p)]
    times = rng.integers(1_000_000, 2_000_000, size=n_train)
    pd.DataFrame({"user_id": users, "item_id": items, "time": times}).to_csv(os.path.join(d, "train.csv"), sep="\t", index=False)
    for phase in ("dev", "test"):
        eu = np.arange(1, n_eval + 1)
        ei = perm[rng.choice(n_items, size=n_eval, p=p)]
        negs = [rng.integers(1, n_items + 1, size=99).tolist() for _ in range(n_eval)]
        pd.DataFrame({"user_id": eu, "item_id": ei, "time": 2_000_001, "neg_items": negs}).to_csv(
            os.path.join(d, phase + ".csv"), sep="\t", index=False)
    return d


def zipf_ids(torch, n_rows, size, gen):
    u = torch.rand(size, generator=gen, dtype=torch.float64)
    ranks = torch.exp(u * np.log(n_rows - 1)).to(torch.int64).clamp_(1, n_rows - 1)
    return (ranks * 2654435761) % (n_rows - 1) + 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="")
    ap.add_argument("--config2-steps", type=int, default=3)
    ap.add_argument("--skip-config2", action="store_true")
    ap.add_argument("--rows", type=int, default=574_197)
    a = ap.parse_args()
    if not os.path.isdir(REF_SRC):
        raise SystemExit("reference not mounted at /root/reference")
    if not sys.dont_write_bytecode:
        raise SystemExit("set PYTHONDONTWRITEBYTECODE=1 (do not drop __pycache__ into the reference)")
    sys.path.insert(0, REF_SRC)
    import torch
    from helpers.BaseReader import BaseReader
    from helpers.BaseRunner import BaseRunner
    from models.general.BPRMF import BPRMF
    from utils import utils

    work = tempfile.mkdtemp(prefix="rechorus_ref_")
    make_ml1m_like(work, n_train=a.rows)
    res = {"host": {"nproc": os.cpu_count(), "torch_threads": torch.get_num_threads(), "torch": torch.__version__,
                    "cpu": next((l.split(":")[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")), "?")}}

    def build(num_workers, batch_size=256, num_neg=1, optimizer="Adam", lr=1e-3, l2=1e-6):
        p = argparse.ArgumentParser()
        p = BaseReader.parse_data_args(p)
        p = BaseRunner.parse_runner_args(p)
        p = BPRMF.parse_model_args(p)
        args, _ = p.parse_known_args(["--path", work + "/", "--dataset", "ml1m_like", "--emb_size", "64", "--num_neg", str(num_neg),
                                      "--lr", str(lr), "--l2", str(l2), "--batch_size", str(batch_size),
                                      "--optimizer", optimizer, "--num_workers", str(num_workers)])
        args.device = torch.device("cpu")
        args.train, args.log_file, args.model_path = 1, os.path.join(work, "log.txt"), os.path.join(work, "m.pt")
        return args

    utils.init_seed(0)
    args = build(0)
    corpus = BaseReader(args)
    res["dataset"] = {"n_users": int(corpus.n_users), "n_items": int(corpus.n_items),
                      "train_rows": int(len(corpus.data_df["train"]))}

    # ---- end to end: one epoch of the reference's fit, num_workers 0 and 5
    for nw in (0, 5):
        args = build(nw)
        model = BPRMF(args, corpus)
        model.apply(model.init_weights)
        ds = BPRMF.Dataset(model, corpus, "train")
        ds.prepare()
        runner = BaseRunner(args)
        t0 = time.perf_counter()
        loss = runner.fit(ds, epoch=1)
        dt = time.perf_counter() - t0
        res[f"ml1m_e2e_w{nw}"] = {"seconds": dt, "tuples_per_s": len(ds) / dt, "loss": float(loss)}
        print(f"e2e num_workers={nw}: {dt:.1f} s, {len(ds) / dt:.0f} tuples/s", flush=True)

    # ---- compute only: the loop body of fit on pre-collated batches
    def fit_body(model, runner, batches):
        if model.optimizer is None:
            model.optimizer = runner._build_optimizer(model)
        model.train()
        t0 = time.perf_counter()
        for batch in batches:
            item_ids = batch["item_id"]
            indices = torch.argsort(torch.rand(*item_ids.shape), dim=-1)
            batch = dict(batch)
            batch["item_id"] = item_ids[torch.arange(item_ids.shape[0]).unsqueeze(-1), indices]
            model.optimizer.zero_grad()
            out = model(batch)
            pred = out["prediction"]
            restored = torch.zeros(*pred.shape)
            restored[torch.arange(item_ids.shape[0]).unsqueeze(-1), indices] = pred
            out["prediction"] = restored
            loss = model.loss(out)
            loss.backward()
            model.optimizer.step()
            loss.detach().cpu().data.numpy()
        return time.perf_counter() - t0

    args = build(0)
    model = BPRMF(args, corpus)
    model.apply(model.init_weights)
    ds = BPRMF.Dataset(model, corpus, "train")
    ds.prepare()
    ds.actions_before_epoch()
    order = np.random.permutation(len(ds))
    batches = [ds.collate_batch([ds[i] for i in order[s:s + 256]]) for s in range(0, len(ds), 256)]
    dt = fit_body(model, BaseRunner(args), batches)
    res["ml1m_compute"] = {"seconds": dt, "tuples_per_s": len(ds) / dt}
    print(f"compute only: {dt:.1f} s, {len(ds) / dt:.0f} tuples/s", flush=True)

    # ---- config-2 batch stream through the reference's own model / loss / autograd / optimizer
    if not a.skip_config2:
        from types import SimpleNamespace
        B, K, n_items, n_users = 65536, 99, 10_000_001, 1_000_001
        margs = SimpleNamespace(device=torch.device("cpu"), model_path="", buffer=1, num_neg=K, dropout=0, test_all=0, emb_size=64)
        big = BPRMF(margs, SimpleNamespace(n_users=n_users, n_items=n_items))
        big.apply(big.init_weights)
        rargs = build(0, batch_size=B, num_neg=K, optimizer="SGD", lr=1e-3, l2=0.0)
        gen = torch.Generator()
        gen.manual_seed(1)
        bs = []
        for _ in range(a.config2_steps + 1):
            uid = zipf_ids(torch, n_users, (B,), gen)
            pos = zipf_ids(torch, n_items, (B, 1), gen)
            neg = torch.randint(1, n_items, (B, K), generator=gen)
            bs.append({"user_id": uid, "item_id": torch.cat([pos, neg], dim=1), "batch_size": B, "phase": "train"})
        runner = BaseRunner(rargs)
        fit_body(big, runner, bs[:1])  # warm-up (page faults, thread pool)
        dt = fit_body(big, runner, bs[1:])
        res["config2_compute"] = {"steps": a.config2_steps, "seconds": dt, "tuples_per_s": a.config2_steps * B / dt,
                                  "optimizer": "SGD l2=0 (dense, all rows)", "B": B, "K": K, "n_items": n_items, "n_users": n_users}
        print(f"config 2 stream: {dt:.1f} s for {a.config2_steps} steps, {a.config2_steps * B / dt:.0f} tuples/s", flush=True)

    print(json.dumps(res))
    if a.out:
        with open(a.out, "w") as f:
            json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
