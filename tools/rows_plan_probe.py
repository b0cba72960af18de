"""time rc_rows_plan_build / rc_rows_plan_update alone at SASRec config 3 (B 4096, C 100, L 50, 8714 rows, Zipf items)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from rechorus_amd import engine as eng
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(1)
B, C, L, n, d = 4096, 100, 50, 8714, 64
def zipf(size):
    u = torch.rand(size, generator=g, device=dev, dtype=torch.float64)
    return torch.exp(u * np.log(n - 1)).to(torch.int64).clamp_(1, n - 1)
lengths = torch.randint(1, L + 1, (B,), generator=g, device=dev)
hist = (zipf((B, L)) * (torch.arange(L, device=dev)[None, :] < lengths[:, None])).contiguous()
iid = torch.cat([zipf((B, 1)), torch.randint(1, n, (B, C - 1), generator=g, device=dev)], 1).contiguous()
hv = torch.randn(B, d, device=dev); gpred = torch.randn(B * C, device=dev); g_hist = torch.randn(B * L, d, device=dev)
W = torch.randn(n, d, device=dev)
h = eng.make_hyper("SGD", lr=1e-3, l2=0.0, step=1)
def timed(fn, reps=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    return round(a.elapsed_time(b) / reps * 1000, 1)
print("build", timed(lambda: eng.RowsPlan(iid, hist, lengths, n, d, tag="probe")), "us (three launches, eager)", flush=True)
rp = eng.RowsPlan(iid, hist, lengths, n, d, tag="probe")
_, _, st, en, _ = rp.views()
cnt = (en - st).cpu().numpy()
print("rows", n, "occurrences", int(cnt.sum()), "hot rows", int((cnt > 192).sum()), "their occurrences", int(cnt[cnt > 192].sum()),
      "max", int(cnt.max()), "median", int(np.median(cnt)), flush=True)
print("update", timed(lambda: rp.update(hv, hyper=h, W=W, coef=gpred, div=C, src2=g_hist)), "us (eager)", flush=True)
