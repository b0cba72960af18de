"""How much of the SASRec step is host launch cost?  Times (a) the Python loop alone (enqueue, no sync inside),
(b) the loop + final sync, (c) the same steps replayed from hipGraphs.  GPU box only."""
import os, sys, time, types
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from rechorus_amd import engine

args = types.SimpleNamespace(emb_size=64, hist=50, layers=1, heads=4, items=8714, pool=8, batch=int(os.environ.get("B", 4096)),
                             num_neg=99, opt="SGD", lr=1e-3, l2=1e-6)
dev = torch.device("cuda:0")
trainer, batches = bench.make_sasrec(args, dev, engine, 99)
K = 200
for s in range(20):
    trainer.step(*batches[s % 8])
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    for s in range(K):
        trainer.step(*batches[s % 8])
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"eager: enqueue {1e3*(t1-t0)/K:.4f} ms/step, total {1e3*(t2-t0)/K:.4f} ms/step", flush=True)
try:
    graphs = []
    side = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(side):
        for b in range(8):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=side):
                trainer.step(*batches[b])
            graphs.append(g)
    torch.cuda.synchronize()
    for rep in range(3):
        t0 = time.perf_counter()
        for s in range(K):
            graphs[s % 8].replay()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"graph: total {1e3*(t2-t0)/K:.4f} ms/step loss {float(trainer.loss):.5f}", flush=True)
except Exception as e:
    print("graph capture failed:", repr(e)[:400])
