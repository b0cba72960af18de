"""Workload for the rocprofv3 PMC passes: a calibration copy of known size (torch copy of the
item table: 2.56 GB read + 2.56 GB written, streaming) followed by a few BPRMF training steps
at the bench shape.  Run under `rocprofv3 --pmc <COUNTER> --kernel-trace`."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from rechorus_amd import engine  # noqa: E402


def main():
    args = bench.parse()
    dev = torch.device("cuda:0")
    gen = torch.Generator(device=dev)
    gen.manual_seed(1234)
    cal_d = 64 if args.workload in ("neumf", "sasrec", "deepfm") else args.emb_size      # the calibration copy is always a 10,000,001 x 64 table (2.56 GB)
    U = torch.empty((args.users if args.workload == "bprmf" else 8, cal_d), device=dev).normal_(0, 0.01, generator=gen)
    I = torch.empty((10_000_001 if args.workload in ("neumf", "sasrec", "deepfm") else args.items, cal_d), device=dev).normal_(0, 0.01, generator=gen)
    I2 = torch.empty_like(I)
    for _ in range(3):
        I2.copy_(I)  # calibration: known bytes
    del I2
    if args.workload in ("sasrec", "deepfm"):
        # whole-step traffic of the configs[2] / configs[4] legs: warm-up steps (optimizer state, workspaces), the calibration copy
        # AGAIN as a marker, then PMC_STEPS eager steps -- tools/pmc_summarize.py sums every kernel behind the last big copy
        n_steps = int(os.environ.get("PMC_STEPS", "5"))
        os.environ["RC_SAS_GRAPH"] = "0"            # eager launches: every kernel of a step is a dispatch of its own
        if args.workload == "sasrec":
            tr, bt = bench.make_sasrec(args, dev, engine, seed=99)
        else:
            tr = bench.DeepfmBench(args, dev)
            tr.model.train()
            bt = tr.batches(args, dev, seed=99)
        # DeepFM: the eager step GraphedStep runs before its capture -- the same kernels as the replayed step, HipOptimizer's rows
        # mode included (a plain forward / backward / step() outside GraphedStep keeps dense table gradients)
        step = (lambda f: tr.graphed._eager(f)) if args.workload == "deepfm" else tr.step
        for s in range(3):
            step(*bt[s % len(bt)])
        torch.cuda.synchronize()
        I2 = torch.empty_like(I)
        I2.copy_(I)                                  # marker
        torch.cuda.synchronize()
        for s in range(n_steps):
            step(*bt[s % len(bt)])
        torch.cuda.synchronize()
        print("pmc workload done; table bytes", I.numel() * 4, "steps", n_steps)
        return
    batches = bench.make_batches(args, dev, seed=99)
    if args.workload == "neumf":     # `--workload neumf`: the fused NeuMF step behind the same calibration copy
        cal_bytes = I.numel() * 4
        del U, I
        tr = bench.make_neumf_trainer(args, 1, dev, engine)
        for s in range(6):
            tr.step(*batches[s % len(batches)], next_batch=batches[(s + 1) % len(batches)])
        torch.cuda.synchronize()
        print("pmc workload done; table bytes", cal_bytes)
        return
    tr = engine.BprmfTrainer(U, I, opt=args.opt, lr=args.lr, l2=args.l2)
    for s in range(6):
        tr.step(*batches[s % len(batches)])
    torch.cuda.synchronize()
    print("pmc workload done; table bytes", I.numel() * 4)


if __name__ == "__main__":
    main()
