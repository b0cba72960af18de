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
    cal_d = 64 if args.workload == "neumf" else args.emb_size      # the calibration copy is always a 10,000,001 x 64 table (2.56 GB)
    U = torch.empty((args.users, cal_d), device=dev).normal_(0, 0.01, generator=gen)
    I = torch.empty((10_000_001 if args.workload == "neumf" else args.items, cal_d), device=dev).normal_(0, 0.01, generator=gen)
    I2 = torch.empty_like(I)
    for _ in range(3):
        I2.copy_(I)  # calibration: known bytes
    del I2
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
