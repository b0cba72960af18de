"""Experiment: does the bucket plan of the NEXT batch (latency-bound index work) hide behind the HBM-bound kernels of
the current step when it runs on a second stream?  Times K training steps alone, K steps with an independent
rc_bucket_plan enqueued on a side stream every step, and the plan alone."""
import ctypes as C
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from rechorus_amd import _lib, engine  # noqa: E402


def main(steps=30):
    args = bench.parse()
    dev = torch.device("cuda:0")
    gen = torch.Generator(device=dev)
    gen.manual_seed(1234)
    U = torch.empty((args.users, args.emb_size), device=dev).normal_(0, 0.01, generator=gen)
    I = torch.empty((args.items, args.emb_size), device=dev).normal_(0, 0.01, generator=gen)
    batches = bench.make_batches(args, dev, seed=99)
    tr = engine.BprmfTrainer(U, I, opt="SGD", lr=1e-3)
    lib = _lib.load()
    n_a, n_b = args.batch * (args.num_neg + 1), args.batch
    rows_a = torch.zeros((n_a, 4), dtype=torch.int32, device=dev)
    rows_b = torch.zeros((n_b, 4), dtype=torch.int32, device=dev)
    cnt = torch.zeros(2, dtype=torch.int32, device=dev)
    occ = torch.zeros(n_a + n_b, dtype=torch.int32, device=dev)
    single = torch.empty(lib.rc_bucket_plan_flags_bytes(n_a), dtype=torch.uint8, device=dev)
    ws = torch.empty(lib.rc_bucket_plan_workspace_bytes(n_a, n_b), dtype=torch.uint8, device=dev)
    p = lambda t: C.c_void_p(t.data_ptr())
    side = torch.cuda.Stream()

    def plan(k):
        uid, iid = batches[(k + 1) % len(batches)]
        _lib.call("rc_bucket_plan", p(iid), n_a, args.items, p(uid), n_b, args.users, 0, p(single), p(rows_a), p(cnt),
                  p(rows_b), p(cnt[1:]), p(occ), p(ws), ws.numel(), engine._stream())

    def run(mode):
        for w in range(3):
            tr.step(*batches[w % len(batches)])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(steps):
            if mode == "plan_only":
                plan(k)
                continue
            if mode == "overlap":
                side.wait_stream(torch.cuda.current_stream())  # the plan of batch k+1 starts with step k
                with torch.cuda.stream(side):
                    plan(k)
            tr.step(*batches[k % len(batches)])
            if mode == "serial":
                plan(k)
            if mode == "overlap":
                torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps * 1e3

    res = {m: run(m) for m in ("alone", "plan_only", "serial", "overlap", "alone")}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
