"""Micro-benchmarks of the building blocks at BASELINE configs[1] sizes (run on the GPU box):
streaming copy (the practical HBM ceiling), random 256-B row gathers, the fused kernel with
and without the singleton update, the segmented update.  Prints one JSON object.

    python tools/microbench.py [--batch 65536] [--emb-size 64]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rechorus_amd import engine  # noqa: E402
from bench import zipf_ids  # noqa: E402


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters  # ms


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=65536)
    ap.add_argument("--emb-size", type=int, default=64)
    ap.add_argument("--num-neg", type=int, default=99)
    ap.add_argument("--items", type=int, default=10_000_001)
    ap.add_argument("--users", type=int, default=1_000_001)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    B, C, d = a.batch, a.num_neg + 1, a.emb_size
    gen = torch.Generator(device=dev)
    gen.manual_seed(0)
    U = torch.empty((a.users, d), device=dev).normal_(0, 0.01, generator=gen)
    I = torch.empty((a.items, d), device=dev).normal_(0, 0.01, generator=gen)
    uid = zipf_ids(a.users, (B,), gen, dev).contiguous()
    pos = zipf_ids(a.items, (B, 1), gen, dev)
    neg = torch.randint(1, a.items, (B, a.num_neg), generator=gen, device=dev)
    iid = torch.cat([pos, neg], dim=1).contiguous()
    n_occ = B * C
    row = 4 * d
    out = {"B": B, "C": C, "d": d}

    # streaming copy: the practical HBM ceiling (read + write)
    I2 = torch.empty_like(I)
    ms = timeit(lambda: I2.copy_(I))
    out["stream_copy_GBps"] = 2 * I.numel() * 4 / ms / 1e6
    # streaming read-only (sum)
    ms = timeit(lambda: I.sum())
    out["stream_read_torch_sum_GBps"] = I.numel() * 4 / ms / 1e6
    del I2

    # random row gather -> streaming write
    ms = timeit(lambda: engine.gather_rows(I, iid))
    out["gather_rows_ms"] = ms
    out["gather_rows_read_GBps"] = n_occ * row / ms / 1e6
    out["gather_rows_total_GBps"] = 2 * n_occ * row / ms / 1e6
    # random row gather, read only (dot with the user row)
    ms = timeit(lambda: engine.gather_dot(U, I, uid, iid))
    out["gather_dot_ms"] = ms
    out["gather_dot_read_GBps"] = n_occ * row / ms / 1e6
    # same rows, sorted ids (perfect locality): separates "random" from "kernel" cost
    iid_sorted = torch.sort(iid.reshape(-1)).values.reshape(B, C).contiguous()
    ms = timeit(lambda: engine.gather_dot(U, I, uid, iid_sorted))
    out["gather_dot_sorted_ids_ms"] = ms
    out["gather_dot_sorted_ids_read_GBps"] = n_occ * row / ms / 1e6

    # fused kernel without / with the singleton update
    ms = timeit(lambda: engine.bprmf_fwd_bwd(U, I, uid, iid, want_pred=False))
    out["fused_ms"] = ms
    out["fused_read_GBps"] = n_occ * row / ms / 1e6
    keys, perm = engine.sort_ids(iid, a.items)
    single = engine.mark_singletons(keys, perm)
    n_single = int(single.sum())
    h = engine.make_hyper("SGD", lr=1e-3, l2=0.0, step=1)
    ms = timeit(lambda: engine.bprmf_fwd_bwd_update(U, I, uid, iid, single, h))
    out["fused_update_ms"] = ms
    out["fused_update_rw_GBps"] = (n_occ + n_single) * row / ms / 1e6
    out["n_single"] = n_single
    _, _, gpred, ugrad = engine.bprmf_fwd_bwd(U, I, uid, iid, want_pred=False)
    _, heads, n_heads = engine.segment_heads(keys, perm, only_multi=True)
    ms = timeit(lambda: engine.segmented_update(keys, perm, U, hyper=h, W=I, coef=gpred.reshape(-1),
                                                src_index=uid, div=C, skip_singletons=True,
                                                heads=heads, n_heads=n_heads))
    out["seg_items_skip_ms"] = ms
    ms = timeit(lambda: engine.segmented_update(keys, perm, U, hyper=h, W=I, coef=gpred.reshape(-1),
                                                src_index=uid, div=C, skip_singletons=False))
    out["seg_items_all_ms"] = ms
    ku, pu = engine.sort_ids(uid, a.users)
    ms = timeit(lambda: engine.segmented_update(ku, pu, ugrad, hyper=h, W=U))
    out["seg_users_ms"] = ms
    ms = timeit(lambda: engine.sort_ids(iid, a.items))
    out["sort_items_ms"] = ms
    ms = timeit(lambda: engine.segment_heads(keys, perm, only_multi=True))
    out["segment_heads_ms"] = ms
    print(json.dumps(out))


if __name__ == "__main__":
    main()
