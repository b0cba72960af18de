"""Step time of the plugin's SASRec on shapes outside the register-resident encoders' envelope (the shape-generic layers of
csrc/seq_layers.hip + the GEMMs of csrc/mlp.hip; dense HipOptimizer step, hipGraph replay like BaseRunner.fit), next to an
in-envelope shape through the same model route for scale.   python tools/bench_sasrec_layers.py > gpurun_out/<tag>/sasrec_layers.txt"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "rechorus_amd", "rechorus"))
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")


def run(d, layers, heads, L, B, K, p, n_items=8714, steps=30):
    from helpers.BaseRunner import BaseRunner
    from models.sequential.SASRec import SASRec
    from rechorus_amd import engine, graph as hgraph
    dev = torch.device("cuda:0")
    args = argparse.Namespace(device=dev, model_path="", buffer=0, num_neg=K, dropout=p, test_all=0, emb_size=d, num_layers=layers,
                              num_heads=heads, history_max=L)
    torch.manual_seed(0)
    model = SASRec(args, argparse.Namespace(n_users=10, n_items=n_items)).to(dev)
    ra = BaseRunner.parse_runner_args(argparse.ArgumentParser()).parse_args([])
    ra.train, ra.log_file, ra.lr, ra.l2, ra.optimizer, ra.engine = 1, "/tmp/rc_bench/l.txt", 1e-4, 1e-6, "Adam", "dense"
    model.optimizer = BaseRunner(ra)._build_optimizer(model)
    model.train()
    rng = np.random.default_rng(1)
    batches = []
    for _ in range(4):
        lengths = rng.integers(1, L + 1, B).astype(np.int64)
        lengths[0] = L
        hist = rng.integers(1, n_items, (B, L)).astype(np.int64) * (np.arange(L)[None, :] < lengths[:, None])
        batches.append({"history_items": torch.from_numpy(hist).to(dev), "lengths": torch.from_numpy(lengths).to(dev),
                        "item_id": torch.from_numpy(rng.integers(1, n_items, (B, 1 + K)).astype(np.int64)).to(dev),
                        "user_id": torch.zeros(B, dtype=torch.long, device=dev), "batch_size": B, "phase": "train"})
    step = hgraph.GraphedStep(model)
    for i in range(6):
        step.run(dict(batches[i % 4]))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        loss = step.run(dict(batches[i % 4]))
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / steps
    route = "register-resident encoder" if engine.sasrec_supported(d, layers, heads, L, p) else "generic layers"
    print(f"d={d:4d} blocks={layers} heads={heads} history={L:4d} B={B:5d} K={K} dropout={p}: {ms:8.3f} ms/step  {B / ms / 1e3:8.3f} M seq/s  "
          f"[{route}; graph {'replayed' if step.graph is not None else 'eager'}]  loss {float(loss):.4f}", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1:       # one configuration: d blocks heads history B K dropout (e.g. under rocprofv3)
        a = sys.argv[1:]
        run(int(a[0]), int(a[1]), int(a[2]), int(a[3]), int(a[4]), int(a[5]), float(a[6]))
        sys.exit(0)
    for cfg in ((64, 1, 4, 50, 4096, 99, 0.0), (128, 1, 4, 50, 4096, 99, 0.0), (128, 2, 4, 50, 256, 99, 0.2), (128, 2, 4, 50, 4096, 99, 0.2),
                (64, 2, 4, 100, 4096, 99, 0.2), (64, 1, 2, 200, 1024, 99, 0.0), (256, 2, 8, 50, 1024, 99, 0.0)):
        run(*cfg)
