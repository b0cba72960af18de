"""Does recording phase events change the NeuMF step?  Wall time per step of the same loop with trainer.timing off / on
(bench.py's model_roofline records hipEvents around every phase): gpurun -- 'python tools/neumf_phase_probe.py'"""
import sys
import time

import torch

sys.path.insert(0, ".")
import bench  # noqa: E402
from rechorus_amd import engine  # noqa: E402

sys.argv = ["bench.py", "--workload", "neumf"]
args = bench.parse()
dev = torch.device("cuda:0")
batches = bench.make_batches(args, dev, seed=99)
tr = bench.make_neumf_trainer(args, 1, dev, engine)
n = len(batches)


def loop(steps, s0=0):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(s0, s0 + steps):
        tr.step(*batches[s % n], next_batch=batches[(s + 1) % n])
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


loop(10)
print("timing off: %.4f ms/step" % loop(40, 10))
tr.timing = {}
print("timing on : %.4f ms/step" % loop(40, 50), {k: round(v, 4) for k, v in engine.phases_ms(tr).items()})
tr.timing = None
print("timing off: %.4f ms/step" % loop(40, 90))
args.steps, args.warmup = 40, 130     # (the loops above ran 130 steps; bench.model_roofline continues the sequence from warmup + steps)
r = bench.model_roofline(args, tr, batches, engine)
print("bench.model_roofline phases:", r["phases_ms"])
tr.timing = {}
for s in range(190, 200):
    tr.step(*batches[s % n], next_batch=batches[(s + 1) % n])
torch.cuda.synchronize()
print("per-step sort ms (10 steps, no sync before):", [round(a.elapsed_time(b), 4) for a, b in tr.timing["sort"]])
print("per-step fused ms:", [round(a.elapsed_time(b), 4) for a, b in tr.timing["fused_step"]])
