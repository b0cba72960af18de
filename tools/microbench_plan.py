"""rc_bucket_plan alone at the bench shape (run under rocprofv3 --kernel-trace --stats for per-kernel times).
RC_PLAN_DEBUG=1|2|4 (timing experiments: drop the occ stores / the flag stores / pass 2 of the bucket kernel)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from rechorus_amd import _lib, engine  # noqa: E402


def main():
    args = bench.parse()
    dev = torch.device("cuda:0")
    batches = bench.make_batches(args, dev, seed=99)
    lib = _lib.load()
    n_a, n_b = args.batch * (args.num_neg + 1), args.batch
    rows_a = torch.zeros((n_a, 4), dtype=torch.int32, device=dev)
    rows_b = torch.zeros((n_b, 4), dtype=torch.int32, device=dev)
    cnt = torch.zeros(2, dtype=torch.int32, device=dev)
    occ = torch.zeros(n_a + n_b, dtype=torch.int32, device=dev)
    single = torch.empty(lib.rc_bucket_plan_flags_bytes(n_a), dtype=torch.uint8, device=dev)
    ws = torch.empty(lib.rc_bucket_plan_workspace_bytes(n_a, n_b), dtype=torch.uint8, device=dev)
    p = lambda t: C.c_void_p(t.data_ptr())
    for s in range(30):
        uid, iid = batches[s % len(batches)]
        _lib.call("rc_bucket_plan", p(iid), n_a, args.items, p(uid), n_b, args.users, 0, p(single), p(rows_a), p(cnt),
                  p(rows_b), p(cnt[1:]), p(occ), p(ws), ws.numel(), engine._stream())
    torch.cuda.synchronize()
    print("rows", cnt.tolist())


if __name__ == "__main__":
    main()
