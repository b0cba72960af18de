"""DeepFM-CTR training step at the sizes of config 5 (MIND-like cardinalities, F = 8 fields,
d = 64, MLP [512, 64], BCE, Adam) through the plugin's model file: gathers / FM term / BCE on the HIP engine,
MLP on rocBLAS, dense HipOptimizer (exact reference semantics).  Reports rows/s and a phase split."""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "rechorus_amd", "rechorus"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--steps", type=int, default=50)
    a = ap.parse_args()
    from helpers.BaseRunner import BaseRunner
    from models.context.DeepFM import DeepFMCTR
    dev = torch.device("cuda")
    vocab = {"user_id": 269312, "item_id": 9373, "c_hour_c": 24, "c_weekday_c": 7, "c_period_c": 9,
             "i_category_c": 18, "i_subcategory_c": 300, "u_group_c": 50}
    args = argparse.Namespace(device=dev, model_path="", buffer=0, num_neg=0, dropout=0.0, test_all=0, emb_size=64,
                              layers="[512,64]", loss_n="BCE")
    corpus = argparse.Namespace(n_users=vocab["user_id"], n_items=vocab["item_id"], user_feature_names=["u_group_c"],
                                item_feature_names=["i_category_c", "i_subcategory_c"],
                                situation_feature_names=["c_hour_c", "c_period_c", "c_weekday_c"], feature_max=vocab)
    model = DeepFMCTR(args, corpus).to(dev)
    ra = BaseRunner.parse_runner_args(argparse.ArgumentParser()).parse_args([])
    ra.train, ra.log_file, ra.lr, ra.l2 = 1, "/tmp/rc_bench/l.txt", 5e-4, 0.0
    model.optimizer = BaseRunner(ra)._build_optimizer(model)
    g = torch.Generator(device=dev).manual_seed(0)
    B = a.batch

    def batch():
        f = {k: torch.randint(0, v, (B, 1) if k.startswith("i") else (B,), device=dev, generator=g) for k, v in vocab.items()}
        f["label"] = torch.randint(0, 2, (B, 1), device=dev, generator=g)
        f["batch_size"], f["phase"] = B, "train"
        return f
    batches = [batch() for _ in range(4)]

    def step(f, timers=None):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)] if timers is not None else None
        def mark(i):
            if ev:
                ev[i].record()
        mark(0)
        model.optimizer.zero_grad()
        out = model(f)
        mark(1)
        loss = model.loss(out)
        mark(2)
        loss.backward()
        mark(3)
        model.optimizer.step()
        mark(4)
        if ev:
            torch.cuda.synchronize()
            for i, name in enumerate(("forward", "loss", "backward", "optimizer")):
                timers[name] = timers.get(name, 0.0) + ev[i].elapsed_time(ev[i + 1])
        return loss

    for w in range(5):
        step(batches[w % 4])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(a.steps):
        loss = step(batches[s % 4])
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    timers = {}
    for s in range(10):
        step(batches[s % 4], timers)
    flops = 2 * B * (8 * 64 * 512 + 512 * 64 + 64) * 3  # MLP fwd + 2x for bwd
    print(json.dumps({"workload": "DeepFMCTR F=8 d=64 layers [512,64] Adam dense, B=%d" % B, "ms_per_step": dt * 1e3,
                      "rows_per_s": B / dt, "mlp_tflops": flops / dt / 1e12, "loss": float(loss),
                      "phases_ms": {k: round(v / 10, 3) for k, v in timers.items()}}))


if __name__ == "__main__":
    main()
