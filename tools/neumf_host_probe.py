"""host time of NeumfTrainer.step (enqueue only) against the GPU time of the step at the config-4 shape"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from rechorus_amd import engine

def main():
    sys.argv = [sys.argv[0], "--workload", "neumf"] + sys.argv[1:]
    args = bench.parse()
    dev = torch.device("cuda:0")
    tr = bench.make_neumf_trainer(args, 1, dev, engine)
    batches = bench.make_batches(args, dev, seed=7)
    n = len(batches)
    for s in range(20):
        tr.step(*batches[s % n], next_batch=batches[(s + 1) % n])
    torch.cuda.synchronize()
    host = []
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    t00 = time.perf_counter()
    for s in range(100):
        t0 = time.perf_counter()
        tr.step(*batches[s % n], next_batch=batches[(s + 1) % n])
        host.append(time.perf_counter() - t0)
    t_enq = time.perf_counter() - t00
    b.record()
    torch.cuda.synchronize()
    print("GPU ms/step %.4f   host enqueue ms/step mean %.4f  median %.4f  (all 100 enqueued in %.1f ms)" % (a.elapsed_time(b) / 100, 1e3 * sum(host) / len(host), 1e3 * sorted(host)[50], 1e3 * t_enq))

main()
