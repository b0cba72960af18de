"""End-to-end epoch time THROUGH THE PLUGIN SURFACE (Reader -> Dataset -> BaseRunner.fit / evaluate), on a
synthetic dataset with the row counts of the reference's demo dataset (Grocery_and_Gourmet_Food: 14,681
users, 8,713 items, ~120 K training rows; docs/demo_scripts_results/README.md publishes 2.5 / 3.4 / 5.5 s
per epoch for BPRMF / NeuMF / SASRec on an unnamed GPU).  Everything is included: negative sampling,
batch assembly, forward, loss, backward, optimizer, and the per-epoch dev evaluation."""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "rechorus_amd", "rechorus"))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def setup(root, model_name, extra, mode="", dataset="grocery_like"):
    import main
    model_cls = main.find_class("model", (model_name, mode))
    reader_cls = main.find_class("helper", model_cls.reader)
    BaseRunner = main.find_class("helper", model_cls.runner)
    p = main.parse_global_args(argparse.ArgumentParser())
    p = reader_cls.parse_data_args(p)
    p = BaseRunner.parse_runner_args(p)
    p = model_cls.parse_model_args(p)
    args = p.parse_args(["--path", root + "/", "--dataset", dataset, "--num_workers", "0"] + extra)
    args.device, args.model_path, args.log_file, args.train = torch.device("cuda"), "/tmp/rc_bench/m.pt", "/tmp/rc_bench/l.txt", 1
    corpus = reader_cls(args)
    model = model_cls(args, corpus).to(args.device)
    data = {ph: model_cls.Dataset(model, corpus, ph) for ph in ("train", "dev", "test")}
    for d in data.values():
        d.prepare()
    return args, model, data, BaseRunner(args)


def main():
    from synth_data import make_context_dataset, make_dataset
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="", help="run only the configurations whose label contains this text")
    only = ap.parse_args().only
    root = tempfile.mkdtemp(prefix="rc_bench_")
    t0 = time.perf_counter()
    make_dataset(root, "grocery_like", n_users=14681, n_items=8713, per_user=10, n_neg=99, seed=0)
    gen_s = time.perf_counter() - t0
    configs = [
        ("BPRMF K=1 B=256 Adam (reference demo flags)", "BPRMF", ["--emb_size", "64", "--lr", "1e-3", "--l2", "1e-6"]),
        ("BPRMF K=1 B=256 Adam, DataLoader path", "BPRMF", ["--emb_size", "64", "--lr", "1e-3", "--l2", "1e-6", "--device_pipeline", "0"]),
        ("BPRMF K=99 B=4096 Adam rowwise", "BPRMF", ["--emb_size", "64", "--num_neg", "99", "--batch_size", "4096", "--engine", "rowwise"]),
        ("NeuMF K=1 B=256 Adam (demo flags, dropout 0)", "NeuMF", ["--emb_size", "64", "--layers", "[64]", "--lr", "5e-4", "--l2", "1e-7"]),
        ("NeuMF K=4 B=4096 Adam rowwise", "NeuMF", ["--emb_size", "64", "--layers", "[64]", "--num_neg", "4", "--batch_size", "4096", "--engine", "rowwise"]),
        ("SASRec L=20 H=1 K=1 B=256 Adam (demo flags)", "SASRec", ["--emb_size", "64", "--num_layers", "1", "--num_heads", "1", "--lr", "1e-4", "--l2", "1e-6", "--history_max", "20"]),
        ("SASRec L=50 H=4 K=99 B=4096 Adam rowwise", "SASRec", ["--emb_size", "64", "--num_layers", "1", "--num_heads", "4", "--history_max", "50", "--num_neg", "99", "--batch_size", "4096", "--engine", "rowwise"]),
    ]
    make_context_dataset(root, "ctr_like", n_users=3000, n_items=2000, per_user=44, ctr=True, seed=1)  # 120 K train rows
    ctr = ["--emb_size", "64", "--layers", "[512,64]", "--loss_n", "BCE", "--lr", "5e-4", "--batch_size", "1024", "--metric", "AUC",
           "--include_item_features", "1", "--include_user_features", "1", "--include_situation_features", "1"]
    configs += [("DeepFM-CTR F=7 B=1024 Adam (CTR_MIND.sh flags, dropout 0)", "DeepFM", ctr, "CTR", "ctr_like"),
                ("DeepFM-CTR, DataLoader path", "DeepFM", ctr + ["--device_pipeline", "0"], "CTR", "ctr_like")]
    out = {"dataset": "synthetic, Grocery-sized: 14,681 users, 8,713 items", "generate_s": round(gen_s, 1), "runs": []}
    for cfg in configs:
        label, model_name, extra = cfg[:3]
        if only and only not in label:
            continue
        args, model, data, runner = setup(root, model_name, extra, *cfg[3:])
        n = len(data["train"])
        np.random.seed(0)
        times = []
        for ep in range(4):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            loss = runner.fit(data["train"], epoch=ep + 1)
            torch.cuda.synchronize()
            times.append(time.perf_counter() - t0)
        t0 = time.perf_counter()
        res = runner.evaluate(data["dev"], [5, 10], runner.metrics)
        torch.cuda.synchronize()
        ev = time.perf_counter() - t0
        best = min(times[1:])
        out["runs"].append({"config": label, "train_rows": n, "epoch_s": [round(t, 3) for t in times],
                            "tuples_per_s": round(n / best), "eval_dev_s": round(ev, 3), "loss": round(loss, 4),
                            "dev": {k: round(float(v), 4) for k, v in list(res.items())[:2]}})
        print(json.dumps(out["runs"][-1]), flush=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
