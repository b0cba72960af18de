"""Loop-back profile of ShardedNeumf's step for ONE RANK ALONE at the config-4 per-GPU shape (B = 65,536 tuples, K = 4, d = 128,
hidden 64, four micro-batches): a world of size 1 (RCCL) forced through the general exchange path (force_exchange), so every local
cost -- grouping the ids by owner, de-duplication, serving rows, the head on the fetched blocks, the owner-side pair updates, the
self-copies through RCCL -- is on the clock; the xGMI transfer time is NOT (nothing leaves the GPU).
    python tools/bench_sharded_neumf_loopback.py [rows|owner] [n_items_per_rank] [micro_batches]"""
import json
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rechorus_amd.sharded import ShardedNeumf  # noqa: E402


def main():
    item_half = sys.argv[1] if len(sys.argv) > 1 else "rows"
    n_items = int(sys.argv[2]) if len(sys.argv) > 2 else 12_500_001
    M = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    B, K, d, hidden, n_users = 65536, 4, 128, 64, 1_250_001
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29541")
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    g = torch.Generator(device=dev).manual_seed(0)
    batches = [(torch.randint(1, n_users, (B,), device=dev, generator=g), torch.randint(1, n_items, (B, 1 + K), device=dev, generator=g))
               for _ in range(4)]
    m = ShardedNeumf(n_users, n_items, d, hidden, opt="SGD", lr=0.01, device=dev, micro_batches=M, item_half=item_half, force_exchange=True)
    for w in range(4):
        m.step(*batches[w % 4], next_batch=batches[(w + 1) % 4])
    torch.cuda.synchronize()
    steps = 20
    t0 = time.perf_counter()
    for k in range(steps):
        m.step(*batches[k % 4], next_batch=batches[(k + 1) % 4])
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    acc = {}
    m.timing = []
    for k in range(8):
        m.step(*batches[k % 4], next_batch=batches[(k + 1) % 4])
        for name, v in m.timing_ms().items():
            acc[name] = acc.get(name, 0.0) + v / 8
    m.timing = None
    print(json.dumps({"item_half": m.item_half, "micro_batches": M, "B": B, "K": K, "d": d, "hidden": hidden, "items_per_rank": n_items,
                      "loopback_ms_per_step": round(ms, 4), "phases_ms": {k: round(v, 4) for k, v in acc.items()}, "wire": m.wire}))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
