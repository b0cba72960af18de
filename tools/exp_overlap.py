"""Experiment: does the id sort (rocPRIM, latency-bound) hide behind the HBM-bound fused kernel when
it runs on a second stream?  Times K training steps alone, then K steps with an independent
sort+heads of the NEXT batch's ids enqueued on a side stream every step."""
import json
import sys
import time

import torch

sys.path.insert(0, ".")
from rechorus_amd import engine  # noqa: E402


def main(B=65536, K=99, d=64, n_items=10_000_001, n_users=1_000_001, steps=30):
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(0)
    U = torch.randn(n_users, d, device=dev, generator=g) * 0.01
    I = torch.randn(n_items, d, device=dev, generator=g) * 0.01
    batches = [(torch.randint(1, n_users, (B,), device=dev, generator=g),
                torch.randint(1, n_items, (B, 1 + K), device=dev, generator=g)) for _ in range(4)]
    tr = engine.BprmfTrainer(U, I, opt="SGD", lr=0.01)
    side = torch.cuda.Stream()

    def run(overlap):
        for w in range(3):
            tr.step(*batches[w % 4])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(steps):
            uid, iid = batches[k % 4]
            if overlap:
                side.wait_stream(torch.cuda.current_stream())  # (conservative) start no earlier than this step
                with torch.cuda.stream(side):
                    nu, ni = batches[(k + 1) % 4]
                    keys, perm = engine.sort_ids2(ni, nu, n_items, n_items + n_users) if hasattr(engine, "sort_ids2") else engine.sort_ids(ni, n_items)
                    engine.segment_heads(keys[: ni.numel()], perm[: ni.numel()], only_multi=True)
            tr.step(uid, iid)
            if overlap:
                torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps * 1e3

    res = {"alone_ms": run(False), "with_side_sort_ms": run(True), "alone_again_ms": run(False)}
    ph = tr.profile_step(*batches[0])
    res["phases"] = ph
    print(json.dumps(res))


if __name__ == "__main__":
    main()
