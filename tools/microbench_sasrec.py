"""SASRec at BASELINE configs[2] shape (d=64, history_max=50, K=99, Grocery-sized catalogue):
encoder kernel times and sequences/s of a whole training step.  Prints one JSON object."""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rechorus_amd import engine  # noqa: E402
from microbench import timeit  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--emb-size", type=int, default=64)
    ap.add_argument("--heads", type=int, default=4)
    ap.add_argument("--layers", type=int, default=1)
    ap.add_argument("--hist", type=int, default=50)
    ap.add_argument("--num-neg", type=int, default=99)
    ap.add_argument("--items", type=int, default=8714)
    ap.add_argument("--impl", default="auto", choices=["auto", "batch", "sequence"],
                    help="encoder kernels: one workgroup per sequence, batch-level row-space kernels, or by batch size")
    a = ap.parse_args()
    os.environ["RC_SASREC_IMPL"] = a.impl
    dev = torch.device("cuda:0")
    B, L, d, C = a.batch, a.hist, a.emb_size, a.num_neg + 1
    gen = torch.Generator(device=dev)
    gen.manual_seed(0)
    mk = lambda *s: torch.empty(s, device=dev).normal_(0, 0.05, generator=gen)
    layers = []
    for _ in range(a.layers):
        lay = {k: (mk(d, d) if k.startswith("W") else mk(d)) for k in engine.SAS_LAYER_KEYS}
        lay["ln1w"] += 1.0
        lay["ln2w"] += 1.0
        layers.append(lay)
    P = {"item_emb": mk(a.items, d), "pos_emb": mk(L + 1, d), "layers": layers}
    lengths = torch.randint(1, L + 1, (B,), generator=gen, device=dev)
    lengths[0] = L
    hist = torch.randint(1, a.items, (B, L), generator=gen, device=dev)
    hist = hist * (torch.arange(L, device=dev)[None, :] < lengths[:, None])
    iid = torch.randint(1, a.items, (B, C), generator=gen, device=dev)
    out = {"impl": engine._sasrec_impl(B, L, None), "B": B, "L": L, "d": d, "heads": a.heads, "layers": a.layers, "mean_len": float(lengths.float().mean())}
    ms = timeit(lambda: engine.sasrec_fwd(P["item_emb"], P["pos_emb"], layers, a.heads, hist, lengths), iters=10)
    out["fwd_ms"] = ms
    flops = float((lengths.double() * (10 * d * d) + lengths.double() ** 2 * 2 * d).sum()) * a.layers  # MAC*2 approx
    out["fwd_GFLOPs"] = flops / ms / 1e6
    hv, xs = engine.sasrec_fwd(P["item_emb"], P["pos_emb"], layers, a.heads, hist, lengths, save=True)
    dhv = torch.randn_like(hv)
    ms = timeit(lambda: engine.sasrec_bwd(layers, a.heads, lengths, xs, dhv), iters=10)
    out["bwd_ms"] = ms
    tr = engine.SasrecTrainer(P, a.heads, opt="Adam", lr=1e-4, l2=1e-6, rowwise=False)
    ms = timeit(lambda: tr.step(hist, lengths, iid), iters=10)
    out["step_ms"] = ms
    out["sequences_per_s"] = B / ms * 1e3
    print(json.dumps(out))


if __name__ == "__main__":
    main()
