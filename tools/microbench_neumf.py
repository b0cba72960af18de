"""NeuMF head at BASELINE configs[3] per-GPU shape (d=128, K=4, hidden 64): kernel times, TFLOP/s
of the fp32-MFMA GEMMs, tuples/s of a whole training step.  Prints one JSON object."""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rechorus_amd import engine  # noqa: E402
from bench import zipf_ids  # noqa: E402
from microbench import timeit  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=65536)
    ap.add_argument("--emb-size", type=int, default=128)
    ap.add_argument("--hidden", type=int, default=64)
    ap.add_argument("--num-neg", type=int, default=4)
    ap.add_argument("--items", type=int, default=10_000_001)
    ap.add_argument("--users", type=int, default=1_000_001)
    ap.add_argument("--opt", default="SGD")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    B, C, d, l1 = a.batch, a.num_neg + 1, a.emb_size, a.hidden
    gen = torch.Generator(device=dev)
    gen.manual_seed(0)
    mk = lambda *s: torch.empty(s, device=dev).normal_(0, 0.01, generator=gen)
    P = {"mf_u": mk(a.users, d), "mlp_u": mk(a.users, d), "mf_i": mk(a.items, d), "mlp_i": mk(a.items, d),
         "W1": mk(l1, 2 * d), "b1": mk(l1), "w_out": mk(d + l1)}
    uid = zipf_ids(a.users, (B,), gen, dev).contiguous()
    iid = torch.cat([zipf_ids(a.items, (B, 1), gen, dev),
                     torch.randint(1, a.items, (B, a.num_neg), generator=gen, device=dev)], dim=1).contiguous()
    n = B * C
    out = {"B": B, "C": C, "d": d, "hidden": l1}
    ms = timeit(lambda: engine.neumf_fwd(P, uid, iid))
    out["fwd_ms"] = ms
    out["fwd_TFLOPs"] = 2.0 * n * (2 * d) * l1 / ms / 1e9
    out["fwd_gather_GBps"] = n * 4 * d * 4 / ms / 1e6  # 4 rows of d floats per candidate (user rows mostly cached)
    pred = engine.neumf_fwd(P, uid, iid)
    _, _, gpred = engine.bpr_loss(pred)
    ms = timeit(lambda: engine.neumf_bwd(P, uid, iid, gpred))
    out["bwd_ms"] = ms
    out["bwd_TFLOPs"] = 4 * 2.0 * n * (2 * d) * l1 / ms / 1e9  # fwd recompute + dh0 + dW1 (+ small)
    tr = engine.NeumfTrainer(P, opt=a.opt, lr=1e-3, l2=0.0)
    ms = timeit(lambda: tr.step(uid, iid), iters=10)
    out["step_ms"] = ms
    out["tuples_per_s"] = B / ms * 1e3
    print(json.dumps(out))


if __name__ == "__main__":
    main()
