"""Loop-back profile of the sharded step on ONE GPU: a world of size 1 (RCCL) forced through the general
exchange path, so every local cost (owner grouping, packing, owner-side kernels, self-copies through
RCCL) is visible; the xGMI transfer time is NOT (nothing leaves the GPU)."""
import json
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, ".")
from rechorus_amd.sharded import ShardedBprmf  # noqa: E402


def main(B=65536, K=99, d=64, n_items=10_000_001, n_users=1_000_001, steps=10, mode="owner"):
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    g = torch.Generator(device=dev).manual_seed(0)
    batches = [(torch.randint(1, n_users, (B,), device=dev, generator=g),
                torch.randint(1, n_items, (B, 1 + K), device=dev, generator=g)) for _ in range(4)]
    m = ShardedBprmf(n_users, n_items, d, opt="SGD", lr=0.01, device=dev, force_exchange=True, timing=True, mode=mode)
    for w in range(3):
        m.step(*batches[w % 4], next_batch=batches[(w + 1) % 4])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        m.step(*batches[(k + 3) % 4], next_batch=batches[(k + 4) % 4])     # (every step announces the following batch: look-ahead routing)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    print(json.dumps({"mode": mode, "K": K, "loopback_ms_per_step": ms, "phases_ms": {k: round(v, 3) for k, v in m.timing_ms().items()}}))
    dist.destroy_process_group()


if __name__ == "__main__":
    # usage: bench_sharded_loopback.py [K [mode]]
    K = int(sys.argv[1]) if len(sys.argv) > 1 else 99
    main(K=K, mode=sys.argv[2] if len(sys.argv) > 2 else "owner")
