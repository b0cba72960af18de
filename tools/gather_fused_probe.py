"""Durations of the field gather's launch variants at the DeepFM bench shape (B = 1,024, MIND's field set): the plain pair gather,
with the FM term, with the plan workgroups riding, with both; and the row sums with / without the FM term's backward.
    python tools/gather_fused_probe.py [B]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from rechorus_amd import engine  # noqa: E402


def timed(fn, reps=10, replays=40):
    """us per call, replayed from a hipGraph of `reps` calls (the host side of a call -- ctypes arrays, allocations -- is not in it)"""
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            fn()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        keep = [fn() for _ in range(reps)]
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(replays):
        g.replay()
    b.record()
    torch.cuda.synchronize()
    del keep
    return a.elapsed_time(b) / (replays * reps) * 1e3


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(1)
    names = list(bench.DEEPFM_VOCAB)
    d = 64
    tables, tables1, ids, kinds = [], [], [], []
    for f in names:
        v = bench.DEEPFM_VOCAB[f]
        if f.endswith("_f"):
            tables.append(torch.randn(d, 1, device=dev) * 0.01)
            tables1.append(torch.randn(1, 1, device=dev) * 0.01)
            ids.append(torch.randint(0, 7, (B,), device=dev, generator=g))
            kinds.append(engine.FIELD_I64)
        else:
            tables.append(torch.randn(v, d, device=dev) * 0.01)
            tables1.append(torch.randn(v, 1, device=dev) * 0.01)
            ids.append(torch.randint(0, v, (B, 1) if f.startswith("i_") or f == "item_id" else (B,), device=dev, generator=g))
            kinds.append(engine.FIELD_IDS)
    F = len(names)
    res = {}
    res["pair gather (mixed)"] = timed(lambda: engine.gather_fields(tables, ids, 1, tables1=tables1, kinds=kinds))
    res["+ FM term"] = timed(lambda: engine.gather_fields(tables, ids, 1, tables1=tables1, kinds=kinds, fm=True))
    res["+ plan"] = timed(lambda: engine.gather_fields(tables, ids, 1, tables1=tables1, kinds=kinds, plan=True))
    res["+ FM term + plan"] = timed(lambda: engine.gather_fields(tables, ids, 1, tables1=tables1, kinds=kinds, fm=True, plan=True))
    V, L, cid, offs, fm, S, ws = engine.gather_fields(tables, ids, 1, tables1=tables1, kinds=kinds, fm=True, plan=True)
    n, n_rows = cid.numel(), offs[-1]
    gv, gl, gf = torch.randn(n, d, device=dev), torch.randn(n, 1, device=dev), torch.randn(B, device=dev)
    into = torch.empty(n_rows * (d + 1), device=dev)
    num = [f for f in range(F) if kinds[f] != engine.FIELD_IDS]
    riding = ([ids[f] for f in num], num, F, 1)
    res["row sums, planned"] = timed(lambda: engine.small_row_sums_planned(ws, n, n_rows, gv, gl, d, (F, B, 1), into=into, numeric=riding))
    res["row sums, planned + FM backward"] = timed(lambda: engine.small_row_sums_planned(ws, n, n_rows, gv, gl, d, (F, B, 1), into=into, numeric=riding,
                                                                                            fm=(V.view(B, F, d), S.view(B, d), gf)))
    res["plan launch + row sums (round-5 route)"] = timed(lambda: engine.small_row_sums_pair(cid, n_rows, gv, gl, into=into, numeric=riding))
    res["FM backward alone"] = timed(lambda: engine.fm_second_order_bwd(V, gf.view(B, 1), add=gv.view(V.shape)))
    res["FM forward alone"] = timed(lambda: engine.fm_second_order(V))
    for k, v in res.items():
        print("%-44s %7.2f us" % (k, v))


if __name__ == "__main__":
    main()
