"""Byte / time model of the sharded NeuMF step (BASELINE configs[3]: d = 128, K = 4, 100 M items, 10 M users) for W GPUs,
from SAMPLED ids of the bench's own generator rather than hand arithmetic: per-destination de-duplication
(rechorus_amd/sharded.py::_Route: only distinct ids leave a rank, one row back, one pre-summed gradient row out),
the remote fraction (W - 1) / W, and the pipeline of `micro_batches` chunks whose exchanges overlap the head kernels.

    python tools/scale_model.py [--batch 65536] > profiles/r02_scale_model.json

compute_ms is what a RANK's step costs locally (round 2 put the single-GPU step there; since round 6 the measured cost of one rank
alone through the exchange path, profiles/r09_sharded_loopback.txt), single_gpu_ms the single-GPU step of the same per-GPU batch
the speed-ups are taken against; the link rates are parameters (xGMI: 7 links x ~153 GB/s peak per GPU).  No GPU needed."""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=65536)
    ap.add_argument("--emb-size", type=int, default=128)
    ap.add_argument("--num-neg", type=int, default=4)
    ap.add_argument("--items", type=int, default=100_000_001)
    ap.add_argument("--users", type=int, default=10_000_001)
    ap.add_argument("--compute-ms", type=float, default=1.50, help="local cost of a rank's sharded step")
    ap.add_argument("--single-gpu-ms", type=float, default=None, help="the single-GPU step the speed-ups are relative to (default: compute-ms)")
    ap.add_argument("--micro-batches", type=int, default=4)
    ap.add_argument("--hidden", type=int, default=64)
    ap.add_argument("--item-half", default="rows", choices=["rows", "owner"],
                    help="owner: an item id moves (mf_i | W1i mlp_i) = emb_size + hidden floats each way instead of 2 emb_size "
                         "(ShardedNeumf item_half='owner', profiles/r09_sharded_neumf_item_half.txt)")
    a = ap.parse_args()
    single = a.single_gpu_ms if a.single_gpu_ms is not None else a.compute_ms
    row_bytes = 2 * a.emb_size * 4   # the mf and mlp rows of a USER id travel together; an item id: see --item-half
    item_row_bytes = (a.emb_size + a.hidden) * 4 if a.item_half == "owner" else row_bytes
    gen = torch.Generator().manual_seed(99)
    dev = torch.device("cpu")
    out = {"config": vars(a), "row_bytes": row_bytes, "item_row_bytes": item_row_bytes, "worlds": {}}
    for W in (2, 4, 8):
        # one rank's batch (every rank draws from the same distributions)
        uid = bench.zipf_ids(a.users, (a.batch,), gen, dev)
        pos = bench.zipf_ids(a.items, (a.batch, 1), gen, dev)
        neg = torch.randint(1, a.items, (a.batch, a.num_neg), generator=gen)
        iid = torch.cat([pos, neg], dim=1).reshape(-1)
        res = {}
        for dedup in (False, True):
            ids_u = uid.unique() if dedup else uid
            ids_i = iid.unique() if dedup else iid
            remote = lambda ids: int((ids % W != 0).sum())   # rank 0's view: ids owned by other ranks
            n_remote = remote(ids_u) + remote(ids_i)
            res["dedup" if dedup else "every_occurrence"] = {
                "lookups": int(uid.numel() + iid.numel()), "ids_sent": int(ids_u.numel() + ids_i.numel()), "ids_remote": n_remote,
                "bytes_each_way": remote(ids_u) * row_bytes + remote(ids_i) * item_row_bytes, "bytes_ids": n_remote * 8}
        b = res["dedup"]["bytes_each_way"]
        # the owners of the OTHER ranks' ids serve as many rows as this rank fetches (symmetric load): in + out per direction
        t = {}
        for bw in (150, 300, 600, 1000):
            exch_ms = 2 * b / (bw * 1e9) * 1e3            # fetch + gradient push, each limited by the GPU's all-to-all rate
            M = a.micro_batches
            step_ms = max(a.compute_ms, exch_ms) + min(a.compute_ms, exch_ms) / M   # pipeline fill / drain of one chunk
            t[f"{bw}GBps"] = {"exchange_ms": round(exch_ms, 3), "step_ms": round(step_ms, 3),
                              "speedup_vs_1gpu": round(W * single / step_ms, 2),
                              "unpipelined_speedup": round(W * single / (a.compute_ms + exch_ms), 2)}
        res["time_model"] = t
        # what the topology offers: W - 1 direct xGMI links of ~153 GB/s per GPU, all used by an all-to-all
        links = {}
        for eff in (0.5, 0.75, 1.0):
            bw = (W - 1) * 153 * eff
            exch_ms = 2 * b / (bw * 1e9) * 1e3
            step_ms = max(a.compute_ms, exch_ms) + min(a.compute_ms, exch_ms) / a.micro_batches
            links[f"{W - 1}_links_at_{int(eff * 100)}pct"] = {"GBps": round(bw), "exchange_ms": round(exch_ms, 3), "step_ms": round(step_ms, 3),
                                                            "speedup_vs_1gpu": round(W * single / step_ms, 2)}
        res["xgmi_links"] = links
        out["worlds"][str(W)] = res
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
