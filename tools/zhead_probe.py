"""step pieces of the owner-computed item half at the config-4 chunk shape (B = 16,384 tuples, C = 5, d = 128, hidden 64): the home
head (engine.neumf_zhead) and the owner's GEMMs on ~41 K served rows; run under rocprofv3 --kernel-trace --stats."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rechorus_amd import engine  # noqa: E402

dev = torch.device("cuda:0")
B, C, d, l1, n_req = 16384, 5, 128, 64, 41000
g = torch.Generator(device=dev).manual_seed(0)
mk = lambda *s: torch.empty(s, device=dev).normal_(0, 0.1, generator=g)
urows, irows, W1, b1, w_out = mk(B, 2 * d), mk(B * C, d + l1), mk(l1, 2 * d), mk(l1), mk(d + l1)
mlp_req, dz = mk(n_req, d), mk(n_req, l1)
W1i = W1[:, d:].contiguous()
for name, fn in (("zhead", lambda: engine.neumf_zhead(urows, irows, W1, b1, w_out, B, C, 1.0 / B)),
                 ("fused MFMA head on (mf | mlp) rows", lambda: engine.neumf_head_fwd_bwd(urows, mk(B * C, 2 * d), W1, b1, w_out, B, C, 1.0 / B)),
                 ("owner fwd GEMM", lambda: engine.linear_fwd(mlp_req, W1i)),
                 ("owner bwd GEMMs", lambda: engine.linear_bwd(mlp_req, W1i, None, dz, need_db=False, ws_tag="probe"))):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    print("%-40s %.3f ms" % (name, (time.perf_counter() - t0) / 20 * 1e3), flush=True)
