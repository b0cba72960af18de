"""time rc_rows_plan_build's three kernels at SASRec config 3 (B 4096, C 100, L 50, 8714 rows) under RC_X_RP debug switches"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rechorus_amd import engine as eng
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(1)
B, C, L, n = 4096, 100, 50, 8714
import numpy as np
def zipf(size):
    u = torch.rand(size, generator=g, device=dev, dtype=torch.float64)
    return torch.exp(u * np.log(n - 1)).to(torch.int64).clamp_(1, n - 1)
lengths = torch.randint(1, L + 1, (B,), generator=g, device=dev)
hist = (zipf((B, L)) * (torch.arange(L, device=dev)[None, :] < lengths[:, None])).contiguous()
iid = torch.cat([zipf((B, 1)), torch.randint(1, n, (B, C - 1), generator=g, device=dev)], 1).contiguous()
for dbg in [int(x) for x in (sys.argv[1:] or ["0"])]:
    os.environ["RC_X_RP"] = str(dbg)
    for _ in range(5):
        eng.RowsPlan(iid, hist, lengths, n, 64, tag="probe")
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(50):
        eng.RowsPlan(iid, hist, lengths, n, 64, tag="probe")
    b.record(); torch.cuda.synchronize()
    print("dbg", dbg, "build", round(a.elapsed_time(b) / 50 * 1000, 1), "us", flush=True)
