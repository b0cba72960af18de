#!/usr/bin/env python
"""bench.py -- ranked (1+K)-tuples/sec of the BPRMF training hot path on MI355X.

Workload = BASELINE.json configs[1]: BPRMF emb_size=64, num_neg=99, synthetic 10M-item Zipf
catalogue, one GPU.  A "step" is one BaseRunner.fit iteration (forward, BPR loss, backward,
optimizer row update) over one batch of B synthetic (user, pos, neg[K]) tuples whose ids are
already resident in HBM.  Prints ONE JSON line (rank 0).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--opt SGD|Adam|Adagrad]

N > 1: one rank per GPU over RCCL.  `python bench.py --gpus N` starts its own N rank processes (rendezvous on
127.0.0.1); launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` it uses the ranks it
is given.  The tables are row-sharded over the ranks and every step trains ONE global batch of N*B tuples
(rechorus_amd/sharded.py); --workload deepfm replicates the model and all-reduces its dense
gradients (sharded.DataParallelDense).  --parallel replicas runs N independent single-GPU jobs instead (no exchange).
"""
import os as _os

# hipGraph launches must use the runtime's regular path (rechorus_amd/graph.py); set before HIP initialises
_os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=65536, help="tuples per step per GPU")
    ap.add_argument("--num-neg", type=int, default=99)
    ap.add_argument("--emb-size", type=int, default=64)
    ap.add_argument("--items", type=int, default=10_000_001)
    ap.add_argument("--users", type=int, default=1_000_001)
    ap.add_argument("--opt", default="SGD", choices=["SGD", "Adam", "Adagrad"])
    ap.add_argument("--lr", type=float, default=1e-3)
    ap.add_argument("--l2", type=float, default=0.0)
    ap.add_argument("--pool", type=int, default=8, help="distinct pre-generated batches (cycled)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=3,
                    help="timed fit() iterations of the CPU baseline (after one untimed warm-up iteration)")
    ap.add_argument("--dropout", type=float, default=None,
                    help="deepfm: dropout of the deep tower (default 0.2 = docs/demo_scripts_results/CTR_MIND.sh:8); neumf: dropout of the "
                         "hidden layer (default 0)")
    ap.add_argument("--graph", action="store_true", help="replay each step from a captured hipGraph (small batches)")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl = RCCL (the only measurement mode); gloo only to smoke-test N>1 on one GPU")
    ap.add_argument("--workload", default="bprmf", choices=["bprmf", "neumf", "sasrec", "deepfm"],
                    help="bprmf = BASELINE configs[1] (the contract workload); neumf = configs[3]: NeuMF emb_size 128, "
                         "num_neg 4, hidden 64 by default (the full config-4 tables: --items 100000001 --users 10000001)")
    ap.add_argument("--hidden", type=int, default=64, help="neumf: size of the hidden layer")
    ap.add_argument("--micro-batches", type=int, default=0,
                    help="neumf, N > 1: chunks of the local batch whose row exchanges overlap the head kernels "
                         "(ShardedNeumf._step_pipelined); 1 = unpipelined; 0 = automatic: 4 when N > 1 (the byte model of "
                         "DESIGN.md section 7 needs the exchange hidden behind the head kernels to reach 6x at 8 GPUs)")
    ap.add_argument("--hist", type=int, default=50, help="sasrec: history_max (BASELINE configs[2])")
    ap.add_argument("--heads", type=int, default=4, help="sasrec: attention heads")
    ap.add_argument("--layers", type=int, default=1, help="sasrec: transformer blocks")
    ap.add_argument("--parallel", default="sharded", choices=["sharded", "replicas"],
                    help="N>1: row-sharded tables + owner-computes exchange (default), or independent replicas")
    ap.add_argument("--mlp", default="[512,64]", help="deepfm: hidden layers of the deep tower (CTR_MIND.sh:8)")
    ap.add_argument("--no-secondary", action="store_true",
                    help="the contract workload only: skip the short legs of the other BASELINE configs that the default run attaches "
                         "as `secondary` (NeuMF at 10 M and 100 M items, SASRec, DeepFM at two batch sizes; N > 1: sharded NeuMF on the "
                         "100 M-item table)")
    args = ap.parse_args()
    if args.dropout is None:
        args.dropout = 0.2 if args.workload == "deepfm" else 0.0
    if args.workload == "deepfm":  # configs[4]: the reference's CTR script shape unless overridden
        if args.batch == 65536:
            args.batch = 1024
        if args.opt == "SGD":
            args.opt = "Adam"
        if args.lr == 1e-3:
            args.lr = 5e-4
    if args.workload == "neumf":  # configs[3]: NeuMF emb_size 128, num_neg 4 unless overridden (tables: --items / --users)
        if args.emb_size == 64:
            args.emb_size = 128
        if args.num_neg == 99:
            args.num_neg = 4
    if args.workload == "sasrec":  # configs[2] is quoted on a Grocery-sized catalogue; keep explicit overrides
        if args.items == 10_000_001:
            args.items = 8714
        if args.batch == 65536:
            args.batch = 4096
    return args


def zipf_ids(n_rows, size, gen, device):
    """Zipf(alpha=1) ranks (density ~ 1/r on [1, n_rows)) mapped to ids by a fixed bijection so
    hot rows are spread over the id space.  n_rows-1 must be coprime
    with the multiplier (true for 10^k)."""
    u = torch.rand(size, generator=gen, device=device, dtype=torch.float64)
    ranks = torch.exp(u * np.log(n_rows - 1)).to(torch.int64).clamp_(1, n_rows - 1)
    return (ranks * 2654435761) % (n_rows - 1) + 1


def make_batches(args, device, seed):
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    out = []
    for _ in range(args.pool):
        uid = zipf_ids(args.users, (args.batch,), gen, device)
        pos = zipf_ids(args.items, (args.batch, 1), gen, device)
        # negatives: uniform randint(1, n_items), models/BaseModel.py:207
        neg = torch.randint(1, args.items, (args.batch, args.num_neg), generator=gen, device=device)
        out.append((uid.contiguous(), torch.cat([pos, neg], dim=1).contiguous()))
    return out


def make_sasrec(args, device, engine, seed):
    """BASELINE configs[2]: SASRec emb_size 64, history_max 50, K = 99 on a Grocery-sized catalogue; histories of
    uniform length 1..history_max (real ones are shorter), Zipf items.  -> (trainer, batches of (hist, lengths, iid))"""
    d, L = args.emb_size, args.hist
    gen = torch.Generator(device=device)
    gen.manual_seed(1234)
    mk = lambda *shape: torch.empty(shape, device=device).normal_(0, 0.05, generator=gen)
    layers = []
    for _ in range(args.layers):
        lay = {k: (mk(d, d) if k.startswith("W") else mk(d)) for k in engine.SAS_LAYER_KEYS}
        lay["ln1w"] += 1.0
        lay["ln2w"] += 1.0
        layers.append(lay)
    P = {"item_emb": mk(args.items, d), "pos_emb": mk(L + 1, d), "layers": layers}
    # the step replays from a hipGraph after two eager steps (RC_SAS_GRAPH=0: eager, bound by the host's launch rate)
    args.sas_graph = os.environ.get("RC_SAS_GRAPH", "1") != "0"   # (Adam: step count in device memory)
    trainer = engine.SasrecTrainer(P, args.heads, opt=args.opt, lr=args.lr, l2=args.l2, rowwise=True, graph=args.sas_graph)
    gen.manual_seed(seed)
    batches = []
    for _ in range(args.pool):
        lengths = torch.randint(1, L + 1, (args.batch,), generator=gen, device=device)
        hist = zipf_ids(args.items, (args.batch, L), gen, device)
        hist = (hist * (torch.arange(L, device=device)[None, :] < lengths[:, None])).contiguous()
        pos = zipf_ids(args.items, (args.batch, 1), gen, device)
        neg = torch.randint(1, args.items, (args.batch, args.num_neg), generator=gen, device=device)
        batches.append((hist, lengths, torch.cat([pos, neg], dim=1).contiguous()))
    return trainer, batches


def make_neumf_trainer(args, world, device, engine):
    """BASELINE configs[3]: NeuMF with its four tables on one GPU (engine.NeumfTrainer) or row-sharded over
    the ranks with the rows travelling over RCCL (rechorus_amd.sharded.ShardedNeumf)"""
    d, l1 = args.emb_size, args.hidden
    if world == 1:
        gen = torch.Generator(device=device)
        gen.manual_seed(1234)
        mk = lambda *shape: torch.empty(shape, device=device).normal_(0, 0.01, generator=gen)
        P = {"mf_u": mk(args.users, d), "mlp_u": mk(args.users, d), "mf_i": mk(args.items, d), "mlp_i": mk(args.items, d),
             "W1": mk(l1, 2 * d), "b1": mk(l1), "w_out": mk(d + l1)}
        return engine.NeumfTrainer(P, opt=args.opt, lr=args.lr, l2=args.l2, rowwise=True, dropout=args.dropout)
    from rechorus_amd.sharded import ShardedNeumf
    # RC_SHARDED_ITEM_HALF=rows: both item rows travel and the MFMA head runs at home (round 5); default "auto": the owners of the
    # item rows compute the item half of the hidden layer (d + hidden floats per distinct id each way instead of 2 d)
    trainer = ShardedNeumf(args.users, args.items, d, l1, opt=args.opt, lr=args.lr, l2=args.l2, device=device, seed=1234,
                           micro_batches=args.micro_batches or (4 if world > 1 else 1),
                           item_half=os.environ.get("RC_SHARDED_ITEM_HALF", "auto"))
    trainer.loss = None
    _step = trainer.step

    def step_and_keep(uid, iid, _step=_step, **kw):
        trainer.loss = _step(uid, iid, **kw)
        return trainer.loss
    trainer.step = step_and_keep
    trainer.lookahead = True   # sharded step: route the NEXT batch before this step's kernels (no host stall per step)
    return trainer


# MIND's own field set (data/MIND_Large/MIND-large.ipynb cells 10-17; CTR_MIND.sh:8 passes --include_item_features 1
# --include_situation_features 1): item_meta's category / subcategory, the situation features hour / weekday / period and the
# NUMERIC c_day_f (days since the first impression: an integer csv column, i.e. an int64 tensor that models/context/FM.py:47-48
# turns into nn.Linear(1, d)(x.float())), user_id, item_id -- in the order ContextCTRModel lists them (BaseContextModel.py:82-83).
# Cardinalities: SURVEY.md 8(d) row 5 (the notebook prints users / items / hour / weekday / period / 7 days; category and
# subcategory counts are not printed: 18 / 300).  Values = vocabulary size, or the exclusive upper bound of a numeric feature.
DEEPFM_VOCAB = {"i_category_c": 18, "i_subcategory_c": 300, "c_day_f": 7, "c_hour_c": 24, "c_period_c": 9, "c_weekday_c": 7,
                "user_id": 269312, "item_id": 9373}
if os.environ.get("RC_BENCH_DEEPFM_DAY") == "categorical":   # A/B only: the day as an eighth categorical field (c_day_c) -- NOT MIND's set
    DEEPFM_VOCAB = {("c_day_c" if k == "c_day_f" else k): v for k, v in DEEPFM_VOCAB.items()}


class DeepfmBench:
    """BASELINE configs[4] on one GPU: DeepFMCTR (F = 8 single-valued fields, emb_size 64, MLP [512, 64], BCE, dense Adam =
    the reference's exact optimizer semantics) through the plugin's model file; the step is what BaseRunner.fit runs
    (model(batch) -> loss -> backward -> optimizer.step), replayed from a hipGraph like the runner does by default."""

    def __init__(self, args, device, data_parallel=False):
        import argparse as ap
        sys.path.insert(0, os.path.join(ROOT, "rechorus_amd", "rechorus"))
        from helpers.BaseRunner import BaseRunner
        from models.context.DeepFM import DeepFMCTR
        from rechorus_amd import graph as hgraph
        self.vocab = dict(DEEPFM_VOCAB)
        margs = ap.Namespace(device=device, model_path="", buffer=0, num_neg=0, dropout=float(getattr(args, "dropout", 0.0) or 0.0), test_all=0,
                             emb_size=args.emb_size,
                             layers=args.mlp, loss_n="BCE")
        corpus = ap.Namespace(n_users=self.vocab["user_id"], n_items=self.vocab["item_id"], user_feature_names=[],
                              item_feature_names=["i_category_c", "i_subcategory_c"],
                              situation_feature_names=[k for k in self.vocab if k.startswith("c_")], feature_max=self.vocab)
        torch.manual_seed(1234)   # replicated parameters: the same initial values on every rank
        self.model = DeepFMCTR(margs, corpus).to(device)
        assert self.model.context_features == list(DEEPFM_VOCAB)
        ra = BaseRunner.parse_runner_args(ap.ArgumentParser()).parse_args([])
        ra.train, ra.log_file, ra.lr, ra.l2, ra.optimizer = 1, "/tmp/rc_bench/l.txt", args.lr, args.l2, args.opt
        self.model.optimizer = BaseRunner(ra)._build_optimizer(self.model)
        self.dp = None
        if data_parallel:
            from rechorus_amd.sharded import DataParallelDense
            self.dp = DataParallelDense(self.model)
        # one GPU: the step is replayed from a hipGraph like the runner does; data-parallel: eager (the collective sits
        # between backward and the optimizer step)
        self.graphed = hgraph.GraphedStep(self.model) if (hgraph.usable() and self.dp is None) else None
        self.loss = None
        self.timing = None

    @property
    def wire(self):
        return {"dense_gradient_allreduce": self.dp.bytes_per_step} if self.dp is not None and self.dp.bytes_per_step else None

    def batches(self, args, device, seed):
        g = torch.Generator(device=device)
        g.manual_seed(seed)
        out = []
        for _ in range(args.pool):
            f = {k: torch.randint(0, v, (args.batch, 1) if k.startswith("i") else (args.batch,), device=device, generator=g)
                 for k, v in self.vocab.items()}
            f["label"] = torch.randint(0, 2, (args.batch, 1), device=device, generator=g)
            f["batch_size"], f["phase"] = args.batch, "train"
            out.append((f,))
        return out

    def step(self, f):
        if self.dp is not None:
            self.loss = self.dp.step(f)
            return self.loss
        if self.timing is None and self.graphed is not None:
            self.loss = self.graphed.run(f)
            return self.loss
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)] if self.timing is not None else None
        mark = (lambda i: ev[i].record()) if ev else (lambda i: None)
        m = self.model
        mark(0)
        m.optimizer.zero_grad()
        out = m(f)
        mark(1)
        loss = m.loss(out)
        mark(2)
        loss.backward()
        mark(3)
        m.optimizer.step()
        mark(4)
        if ev:
            for i, name in enumerate(("forward", "loss", "backward", "optimizer")):
                self.timing.setdefault(name, []).append((ev[i], ev[i + 1]))
        self.loss = loss.detach().reshape(1)
        return self.loss


def deepfm_roofline(args, trainer, batches, engine, ms_per_step=None):
    """MLP_Block GEMMs (csrc/mlp.hip + tower_tail.hip, fp32 MFMA) against the fp32 MFMA peak.  Time = the eager forward + backward
    phases (hipEvents on the launch stream) where the GPU is what they wait for (large batches); at the reference's batch size the eager
    phases are bound by the host's launch rate (0.8 ms of Python for 0.26 ms of GPU work), so the WHOLE replayed step -- gathers, FM
    term, field gradients and the dense Adam included -- is the denominator there: a lower bound on the layers' own rate."""
    trainer.timing = {}
    for s in range(10):
        trainer.step(*batches[s % len(batches)])
    ph = engine.phases_ms(trainer)
    trainer.timing = None
    F, d = len(DEEPFM_VOCAB), args.emb_size
    dims = [F * d] + list(eval(args.mlp)) + [1]
    fwd = 2.0 * args.batch * sum(a * b for a, b in zip(dims[:-1], dims[1:]))
    t_eager = ph["forward"] + ph["backward"]
    host_bound = ms_per_step is not None and t_eager > ms_per_step
    t = ms_per_step if host_bound else t_eager
    ach = 3.0 * fwd / (t * 1e-3) / 1e12
    what = ("whole hipGraph-replayed fit step (the eager phases are host-bound at this batch size): MLP_Block forward + backward (rc_linear_* / "
            "rc_tower_tail_*, hand-written fp32 MFMA) + field gathers, FM term, field gradients, dense Adam") if host_bound else \
           ("MLP_Block forward + backward (rc_linear_fwd / rc_linear_bwd, hand-written fp32 MFMA GEMMs; eager phases incl. the field "
            "gathers, FM term and their backward)")
    rl = {"bound": "mfma", "kernel": what, "achieved": ach, "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
          "frac": ach / F32_MFMA_PEAK_TFLOPS, "traffic": None, "algorithmic_flops_per_launch": 3.0 * fwd, "avg_ms": t}
    default = (args.emb_size, args.mlp.replace(" ", ""), args.opt, abs(float(args.dropout) - 0.2) < 1e-9) == (64, "[512,64]", "Adam", True) and \
        "c_day_f" in DEEPFM_VOCAB
    if default and args.batch in (1024, 131072):
        rl.update(pmc_step_traffic("deepfm_b%d" % args.batch, True))
    return {"phases_ms": {k: round(v, 4) for k, v in ph.items()}, "roofline": rl}


def cpu_baseline_deepfm(args, batches_cpu):
    from oracle.torch_port import DeepfmCtrTorchPort
    torch.manual_seed(0)
    cores = torch.get_num_threads()
    model = DeepfmCtrTorchPort(list(DEEPFM_VOCAB), DEEPFM_VOCAB, args.emb_size, layers=tuple(eval(args.mlp)), dropout=args.dropout)
    optim = model.make_optimizer(args.opt, args.lr, args.l2)
    steps, t_init = [], time.perf_counter()
    for s, (f,) in enumerate(batches_cpu):
        t0 = time.perf_counter()
        model.fit_step(optim, f)
        if s > 0:
            steps.append(time.perf_counter() - t0)
        if time.perf_counter() - t_init > 60 and steps:
            break
    return {"value": args.batch * len(steps) / sum(steps), "unit": "tuples/s", "cores": cores, "kind": "port",
            "step_s": {"min": min(steps), "median": float(np.median(steps)), "max": max(steps)},
            "sample": f"{len(steps)} fit() iterations of oracle/torch_port.py (DeepfmCtrTorchPort, torch {torch.__version__} CPU, {cores} "
                      f"threads, dense torch.optim.{args.opt} like the reference) at B={args.batch}, F={len(DEEPFM_VOCAB)}, d={args.emb_size}, "
                      f"MLP {args.mlp}, dropout {args.dropout:g}; {sum(steps):.1f} s"}


def algorithmic_bytes(args, batches):
    """ALGORITHMIC HBM bytes per launch of each phase: every byte the
    phase must move at least once, table rows counted once per distinct row per step."""
    B, C, d = args.batch, args.num_neg + 1, args.emb_size
    n_occ = B * C
    ui = us = um = mo = 0.0
    for _, i in batches:
        cnt = torch.unique(i, return_counts=True)[1]
        ui += cnt.numel() / len(batches)                      # distinct item rows
        us += int((cnt == 1).sum()) / len(batches)            # ... occurring once
        um += int((cnt > 1).sum()) / len(batches)             # ... occurring several times
        mo += int(cnt[cnt > 1].sum()) / len(batches)          # occurrences of the latter
    uu = float(np.mean([torch.unique(u).numel() for u, _ in batches]))
    row = 4 * d
    # SGD: single-occurrence rows are updated inside the fused kernel; with optimizer state (Adam: m, v;
    # Adagrad: state_sum) every touched row goes through the segmented update (csrc/train_step.hip)
    fast_path = args.opt == "SGD"
    n_state = {"SGD": 0, "Adagrad": 1, "Adam": 2}[args.opt]
    bitmap = args.items / 8.0    # multi-occurrence bitmap over the item ids (what the fused kernel looks its candidates up in)
    fused = (8 * B + 8 * n_occ + (bitmap if fast_path else 0)  # uid, iid, bitmap
             + uu * row + ui * row      # distinct user / item rows, read once
             + (us * row if fast_path else 0)   # single-occurrence item rows written back updated
             + 4 * n_occ + B * row + 4 * B)   # gpred, ugrad, loss_vec written
    upd_rows, upd_occ = (um, mo) if fast_path else (ui, float(n_occ))
    item_update = (16 * upd_rows        # row records
                   + 8 * upd_occ + 8 * B    # grouped positions, gpred + uid lookups of the rows updated here
                   + uu * row           # U rows rebuilt into g*U: distinct rows once
                   + 2 * (1 + n_state) * upd_rows * row)  # row (+ state rows): read + written once
    user_update = 16 * uu + 4 * B + B * row + 2 * (1 + n_state) * uu * row
    # bucket plan front: ids read by the count and by the scatter kernel, bucketed keys written as a 2-byte id stream + a
    # 4-byte position stream, the bitmap kernel reads the id stream and writes the bitmap; per-bucket pass: id stream
    # twice + positions once, grouped positions and row records written
    n = n_occ + B
    front = 2 * 8 * n + 6 * n + (2 * n + bitmap if fast_path else 0)
    back = 2 * 2 * n + 4 * n + 4 * (upd_occ + B) + 16 * (upd_rows + uu)
    return {"fused_fwd_bwd": fused, "item_update": item_update, "user_update": user_update,
            "sort_items": front, "segment_heads": back, "uniq_items": ui, "uniq_users": uu,
            "single_items": us, "multi_items": um, "multi_item_occurrences": mo}


def default_workload(args):
    return (args.opt == "SGD" and args.batch == 65536 and args.num_neg == 99 and args.emb_size == 64
            and args.items == 10_000_001 and args.users == 1_000_001)


def pmc_step_traffic(name, default_shape):
    """{"traffic": HBM bytes of the WHOLE training step, "traffic_scope": ...} from a committed whole-step PMC pass
    (profiles/pmc_<name>_latest.json: tools/pmc_collect.sh / pmc_workload.py run eager steps behind a marker copy, FETCH_SIZE and
    WRITE_SIZE passes calibrated on a table copy) -- only for the shape the pass was taken on"""
    if not default_shape:
        return {"traffic": None}
    try:
        st = json.load(open(os.path.join(ROOT, "profiles", "pmc_%s_latest.json" % name)))["_step"]
        top = sorted(st["kernels"].items(), key=lambda kv: -(kv[1]["hbm_read_bytes_per_step"] + kv[1]["hbm_write_bytes_per_step"]))[:4]
        return {"traffic": st["hbm_bytes_per_step"],
                "traffic_scope": "every kernel of one training step (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE over %d eager steps, "
                                 "profiles/pmc_%s_latest.json); largest: %s" % (st["steps"], name, "; ".join(
                                     "%s %.0f MB" % (k.replace("rc::", "")[:40], (v["hbm_read_bytes_per_step"] + v["hbm_write_bytes_per_step"]) / 1e6)
                                     for k, v in top))}
    except Exception:
        return {"traffic": None}


def load_pmc_traffic(kernel):
    """HBM bytes per launch from committed rocprofv3 PMC passes (profiles/pmc_latest.json), if any."""
    p = os.path.join(ROOT, "profiles", "pmc_latest.json")
    if not os.path.exists(p):
        return None
    prefix = {"fused_fwd_bwd": "bprmf_fwd_bwd_kernel", "item_update": "plan_rows_kernel",
              "user_update": "plan_final_kernel"}.get(kernel, kernel)
    try:
        rows = [v["hbm_bytes_per_launch"] for k, v in json.load(open(p)).items()
                if k.startswith(prefix) and isinstance(v, dict) and v.get("hbm_bytes_per_launch")]
        if not rows:
            return None
        return min(rows) if kernel == "user_update" else max(rows)  # item phase = the big launches
    except Exception:
        return None


def cpu_baseline(args):
    """The torch-CPU port of the reference's fit() iteration (oracle/torch_port.py), same table
    sizes and batch shape, a bounded number of steps on the host cores."""
    from oracle.torch_port import BprmfTorchPort
    torch.manual_seed(0)
    cores = torch.get_num_threads()
    t_init = time.perf_counter()
    model = BprmfTorchPort(args.users, args.items, args.emb_size)
    optim = model.make_optimizer(args.opt, args.lr, args.l2)
    gen = torch.Generator()
    gen.manual_seed(1)
    cpu = torch.device("cpu")
    steps = []
    n = 0
    for s in range(args.cpu_steps + 1):  # first step is warm-up (page faults, thread pool)
        uid = zipf_ids(args.users, (args.batch,), gen, cpu)
        pos = zipf_ids(args.items, (args.batch, 1), gen, cpu)
        neg = torch.randint(1, args.items, (args.batch, args.num_neg), generator=gen)
        iid = torch.cat([pos, neg], dim=1)
        t0 = time.perf_counter()
        model.fit_step(optim, uid, iid)
        dt = time.perf_counter() - t0
        if s > 0:
            steps.append(dt)
            n += args.batch
        if time.perf_counter() - t_init > 90 and steps:
            break
    v = n / sum(steps)
    return {"value": v, "unit": "tuples/s", "cores": cores, "kind": "port",
            "step_s": {"min": min(steps), "median": float(np.median(steps)), "max": max(steps)},
            "sample": f"{len(steps)} fit() iterations of oracle/torch_port.py (torch {torch.__version__} CPU, "
                      f"{cores} threads, dense grads + dense torch.optim.{args.opt} over all rows like the "
                      f"reference) at B={args.batch}, K={args.num_neg}, d={args.emb_size}, "
                      f"{args.items} items, {args.users} users; {sum(steps):.1f} s"}


F32_MFMA_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: f32-input MFMA = the fp32 vector rate


def cpu_baseline_model(args, batches_cpu):
    """NeuMF / SASRec: the torch-CPU ports of the reference's modules (oracle/torch_port.py, pinned against the
    reference-generated goldens in tests/test_torch_ports.py), dense gradients and dense torch.optim like the
    reference, a bounded number of fit() iterations on the host cores."""
    from oracle import torch_port as TP
    torch.manual_seed(0)
    cores = torch.get_num_threads()
    d = args.emb_size
    if args.workload == "neumf":
        table_bytes = 2 * (args.items + args.users) * d * 4
        if table_bytes * 3 > 96e9:  # weights + dense gradients + optimizer traffic would not fit / finish on the host
            return {"value": None, "unit": "tuples/s", "cores": cores, "kind": "port",
                    "sample": f"skipped: {table_bytes / 1e9:.0f} GB of tables (dense gradient + dense optimizer on the host)"}
        model = TP.NeumfTorchPort(args.users, args.items, d, layers=(args.hidden,), dropout=args.dropout)
    else:
        model = TP.SasrecTorchPort(args.items, d, args.hist, n_layers=args.layers, n_heads=args.heads)
    optim = model.make_optimizer(args.opt, args.lr, args.l2)
    steps, t_init = [], time.perf_counter()
    for s, b in enumerate(batches_cpu):
        t0 = time.perf_counter()
        model.fit_step(optim, *b)
        if s > 0:  # first step is warm-up
            steps.append(time.perf_counter() - t0)
        if time.perf_counter() - t_init > 60 and steps:
            break
    v = args.batch * len(steps) / sum(steps)
    return {"value": v, "unit": "tuples/s", "cores": cores, "kind": "port",
            "step_s": {"min": min(steps), "median": float(np.median(steps)), "max": max(steps)},
            "sample": f"{len(steps)} fit() iterations of oracle/torch_port.py ({type(model).__name__}, torch {torch.__version__} CPU, "
                      f"{cores} threads, dense grads + dense torch.optim.{args.opt} like the reference) at B={args.batch}, "
                      f"K={args.num_neg}, d={d}, {args.items} items; {sum(steps):.1f} s"}


def model_roofline(args, trainer, batches, engine):
    """NeuMF / SASRec: live per-phase times (events on the launch stream) -> the dominant phase against its bound:
    fp32 MFMA for the head / encoder kernels, HBM for the table update."""
    trainer.timing = {}
    s0 = args.warmup + args.steps      # the batch sequence of the timed loop continues: its last step announced batch s0
    for s in range(s0, s0 + 20):
        if args.workload == "neumf":   # steady state of the timed loop: every step announces the following batch
            trainer.step(*batches[s % len(batches)], next_batch=batches[(s + 1) % len(batches)])
        else:
            trainer.step(*batches[s % len(batches)])
    ph = engine.phases_ms(trainer)
    trainer.timing = None
    B, C, d = args.batch, args.num_neg + 1, args.emb_size
    out = {"phases_ms": {k: round(v, 4) for k, v in ph.items()}}
    if args.workload == "neumf":
        l1 = args.hidden
        fwd = B * C * (4.0 * d * l1 + 3 * d + 2 * l1)          # GEMM1 + GMF product + output dot
        bwd = B * C * (12.0 * d * l1 + 6 * d + 4 * l1)         # recomputed GEMM1, dh0 = W1^T dz1, dW1 = dz1 h0^T
        flops = {"head_fwd": fwd, "head_bwd": bwd}
        # table update: per-occurrence gradient rows of the four tables read once, distinct rows read + written once
        ui = float(np.mean([torch.unique(i).numel() for _, i in batches]))
        uu = float(np.mean([torch.unique(u).numel() for u, _ in batches]))
        n_state = {"SGD": 0, "Adagrad": 1, "Adam": 2}[args.opt]
        upd_bytes = 2 * (2 * B * C) * d * 4 + 2 * (1 + n_state) * 2 * (ui + uu) * d * 4 + 2 * 8 * B * C
        gather_bytes = {"head_fwd": (2 * B * C + 2 * B * C) * d * 4}
        if "fused_step" in ph:
            # rc_neumf_train_step (csrc/neumf_step.hip): ALGORITHMIC work of one fit iteration's head -- the hidden layer, dh0 and
            # dW1 once each (the kernel's recomputation of the hidden layer in its second pass is not counted), the user half per
            # tuple; bytes: every gathered row once, single-occurrence item rows written back once, one gradient row per tuple
            # and user table, gradient rows of the multi-occurrence positions, ids, loss
            cnt = [torch.unique(i, return_counts=True)[1] for _, i in batches]
            single = float(np.mean([int((c == 1).sum()) for c in cnt]))
            multi_occ = float(np.mean([int(c[c > 1].sum()) for c in cnt]))
            multi_rows = float(np.mean([int((c > 1).sum()) for c in cnt]))
            flops = {"fused_step": (B * C + B) * 6.0 * d * l1 + B * C * (6.0 * d + 4 * l1)}
            fused_bytes = (2 * B * C + 2 * B) * d * 4 + 2 * (1 + 2 * n_state) * single * d * 4 + 2 * B * d * 4 + 2 * multi_occ * d * 4 + \
                8 * (B * C + B) + 4 * B + 4 * B * C
            upd_bytes = 2 * (multi_occ + B) * d * 4 + 2 * (1 + n_state) * 2 * (multi_rows + uu) * d * 4 + 2 * 4 * (B * C + B)
            gather_bytes = {}
            out["fused_step_algorithmic"] = {"flops": flops["fused_step"], "hbm_bytes": fused_bytes, "single_item_rows": single,
                                             "multi_item_rows": multi_rows, "multi_item_occurrences": multi_occ}
    else:
        L, nl = args.hist, args.layers
        R = float(np.mean([int(l.sum()) for _, l, _ in batches]))        # valid history rows of a batch
        sq = float(np.mean([int((l.to(torch.float64) ** 2).sum()) for _, l, _ in batches]))
        full = R * (6.0 * d * d + 4.0 * d * d) + 2.0 * sq * d            # a block on all rows: QKV + FFN projections, causal QK^T + AV
        # the LAST block is needed at one position per sequence (SASRec.py:76) -- what the engine computes (the
        # all-rows count would flatter the rate).  Version 1: K / V projections on all rows, Q and the FFN on B rows, one
        # attention row per sequence.  Version 2 (default, csrc/sas_last_row.hpp): no K / V at all -- per row H dots of length d
        # and H weighted sums; q, Wk^T q, Wv xbar and the FFN on B rows.  With one block version 2 is a streaming pass over the
        # rows: it is rated against HBM below
        H = args.heads
        want = int(os.environ.get("RC_SAS_LAST_ROW", "2"))
        ok = want > 0 and d % H == 0
        v2 = ok and want >= 2 and H in (1, 2, 4) and max(3, H + 1) <= L <= 128 and (d // H) % (d * d // 256) == 0
        v1 = ok and 2 <= L <= 64 and (d // H) in (16, 32, 64) and B * L >= int(os.environ.get("RC_SAS_LAST_ROW_MIN", "32768"))
        mode = 2 if v2 else (1 if v1 else 0)
        out["encoder_path"] = {2: "last block on ONE query row per sequence without keys / values (csrc/sas_last_row.hpp: wave reductions, "
                                  "HBM-bound); lower blocks, if any, on all rows with fp32 MFMA attention",
                               1: "last block: K / V projections on all rows (fp32 MFMA), attention for one query row per sequence",
                               0: "every block on all rows: fp32 MFMA projections + register-resident MFMA attention "
                                  "(csrc/sasrec_batch.hip, sas_attn_reg.hpp)"}[mode]
        last = {0: full, 1: R * 4.0 * d * d + B * 6.0 * d * d + 4.0 * R * d, 2: B * 10.0 * d * d + 4.0 * R * H * d}[mode]
        fwd = (nl - 1) * full + last
        flops = {"encoder_fwd": fwd, "encoder_bwd": 2.0 * fwd}
        enc_bytes = {}
        if mode == 2 and nl == 1:
            per_seq = (2 * d + 2 * H * d + H * L) * 4.0          # x_last, q, Wk^T q, xbar, p
            enc_bytes = {"encoder_fwd": R * (8 + 2 * d * 4) + B * (per_seq + 8 * d * 4),      # ids + table rows read, x rows written
                         "encoder_bwd": R * d * 4 + B * L * d * 4 + B * (per_seq + 2 * H * d * 4 + 12 * d * 4)}   # x read, padded dX written
        ids = [torch.cat([i.reshape(-1), h.reshape(-1)]) for h, _, i in batches]
        ui = float(np.mean([torch.unique(x).numel() for x in ids]))
        n_state = {"SGD": 0, "Adagrad": 1, "Adam": 2}[args.opt]
        upd_bytes = 8 * (B * C + B * L) + R * d * 4 + 4 * B * C + 2 * (1 + n_state) * ui * d * 4
        gather_bytes = {}
    compute = {k: ph[k] for k in flops if k in ph}
    dom = max(list(compute) + ["table_update"], key=lambda k: ph.get(k, 0.0))
    hbm_bytes = dict(enc_bytes if args.workload == "sasrec" else {}, table_update=upd_bytes)
    if dom == "fused_step":
        # the one kernel that does both: rated against the roof it is nearer to, the other fraction beside it
        t = ph[dom] * 1e-3
        traffic = None     # FETCH_SIZE + WRITE_SIZE of the kernel from the committed PMC passes (tools/gpu_run.sh pmc:neumf), default shape only
        if (args.items, args.users, args.emb_size, args.num_neg, args.batch, args.opt) == (10_000_001, 1_000_001, 128, 4, 65536, "SGD"):
            try:
                pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_neumf_latest.json")))
                traffic = max(v["hbm_bytes_per_launch"] for k, v in pmc.items() if k.startswith("neumf_step_kernel"))
            except Exception:
                traffic = None
        f_mfma, f_hbm = flops[dom] / t / 1e12 / F32_MFMA_PEAK_TFLOPS, fused_bytes / t / 1e9 / HBM_PEAK_GBPS
        if f_hbm >= f_mfma:
            out["roofline"] = {"bound": "hbm", "kernel": dom, "achieved": fused_bytes / t / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                               "frac": f_hbm, "traffic": traffic, "algorithmic_bytes_per_launch": fused_bytes, "avg_ms": ph[dom],
                               "frac_of_mfma_peak": f_mfma}
        else:
            out["roofline"] = {"bound": "mfma", "kernel": dom, "achieved": flops[dom] / t / 1e12, "peak": F32_MFMA_PEAK_TFLOPS,
                               "unit": "TFLOP/s", "frac": f_mfma, "traffic": traffic, "algorithmic_flops_per_launch": flops[dom],
                               "avg_ms": ph[dom], "frac_of_hbm_peak": f_hbm, "algorithmic_bytes_per_launch": fused_bytes}
    elif dom in hbm_bytes:
        ach = hbm_bytes[dom] / (ph[dom] * 1e-3) / 1e9
        out["roofline"] = {"bound": "hbm", "kernel": dom, "achieved": ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                           "frac": ach / HBM_PEAK_GBPS, "traffic": None, "algorithmic_bytes_per_launch": hbm_bytes[dom],
                           "avg_ms": ph[dom]}
    else:
        ach = flops[dom] / (ph[dom] * 1e-3) / 1e12
        out["roofline"] = {"bound": "mfma", "kernel": dom, "achieved": ach, "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                           "frac": ach / F32_MFMA_PEAK_TFLOPS, "traffic": None, "algorithmic_flops_per_launch": flops[dom],
                           "avg_ms": ph[dom]}
    if args.workload == "sasrec" and isinstance(out.get("roofline"), dict):
        default = (args.items, args.emb_size, args.hist, args.num_neg, args.batch, args.layers, args.heads, args.opt) == (8714, 64, 50, 99, 4096, 1, 4, "SGD")
        out["roofline"].update(pmc_step_traffic("sasrec", default))
    out["phases_tflops"] = {k: round(flops[k] / (ph[k] * 1e-3) / 1e12, 2) for k in flops if ph.get(k)}
    out["table_update_gbps"] = round(upd_bytes / (ph["table_update"] * 1e-3) / 1e9, 1) if ph.get("table_update") else None
    if args.workload == "sasrec" and enc_bytes:
        out["phases_gbps"] = {k: round(v / (ph[k] * 1e-3) / 1e9, 1) for k, v in enc_bytes.items() if ph.get(k)}
    if gather_bytes.get("head_fwd") and ph.get("head_fwd"):
        out["head_fwd_gather_gbps"] = round(gather_bytes["head_fwd"] / (ph["head_fwd"] * 1e-3) / 1e9, 1)
    return out


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def launch_ranks(n, argv=None, capture=False):
    """`python bench.py --gpus N` without a launcher: start N rank processes of this command line (or of `argv`), one per GPU,
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their environment, rendezvous on 127.0.0.1; fail if any rank fails (the
    others are then stopped by their exact PIDs).  capture: return rank 0's stdout instead of passing it through.
    -> (exit code, rank 0's stdout or None)"""
    import subprocess
    import tempfile
    argv = sys.argv[1:] if argv is None else argv
    port = os.environ.get("MASTER_PORT") or str(_free_port())
    if capture and os.environ.get("MASTER_PORT"):
        port = str(_free_port())     # several rank groups in one run: a fresh rendezvous port each
    procs = []
    cap = tempfile.TemporaryFile(mode="w+") if capture else None
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RC_BENCH_CHILD="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # the host driver only supports dmabuf IPC (RCCL needs it)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=(cap if capture else None) if r == 0 else subprocess.DEVNULL))
    rc = 0
    try:
        pending = list(procs)
        while pending:
            for p in list(pending):
                code = p.poll()
                if code is None:
                    continue
                pending.remove(p)
                if code != 0 and rc == 0:
                    rc = code
                    for q in pending:      # one rank failed: the others would wait in a collective forever
                        q.terminate()
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    text = None
    if cap is not None:
        cap.seek(0)
        text = cap.read()
        cap.close()
    return rc, text


# ---- secondary legs: the other BASELINE configs, short, attached to the contract line -------------------------------------

def secondary_wanted(args):
    """the default invocation of the contract workload carries the other configs; any explicit workload / shape does not"""
    return (not args.no_secondary and args.workload == "bprmf" and os.environ.get("RC_BENCH_LAUNCH_ONLY") != "1"
            and not args.graph and args.parallel == "sharded")


def _last_json(text):
    for line in reversed((text or "").strip().splitlines()):
        line = line.strip()
        if line.startswith("{"):
            try:
                return json.loads(line)
            except ValueError:
                continue
    return None


def _summary(j):
    """what a secondary leg contributes to the contract line"""
    keep = ("metric", "value", "unit", "ms_per_step", "steps", "warmup", "n_gpus", "dtype", "roofline", "phases_ms", "sharded_phases_ms",
            "sharded_wire_bytes_rank0", "final_loss", "phases_tflops", "table_update_gbps", "encoder_path", "cpu_baseline", "launches_per_step")
    out = {k: j[k] for k in keep if k in j}
    out["workload"] = (j.get("config") or {}).get("workload")
    return out


SECONDARY_LEGS = (   # name, bench.py arguments, seconds of CPU baseline (None: no CPU leg -- 113 GB of tables / minutes per iteration)
    # BASELINE.json configs[3] at 10 M items, configs[2], configs[4] at the reference's batch size; then the large shapes
    # (steps / warm-up per leg: a hipGraph-replayed step of 0.2 ms is timed over 200 replays after 20 -- the first replays of a freshly
    #  captured graph run slower, and 20 steps of 0.2 ms are 4 ms of measurement)
    ("neumf", ["--workload", "neumf", "--steps", "40", "--warmup", "8"], 22.0),
    ("sasrec", ["--workload", "sasrec", "--steps", "200", "--warmup", "20"], 12.0),
    ("deepfm_b1024", ["--workload", "deepfm", "--steps", "200", "--warmup", "20"], 2.0),
    ("neumf_100M", ["--workload", "neumf", "--items", "100000001", "--users", "10000001", "--steps", "40", "--warmup", "8"], None),    # CPU: see _borrow_cpu_baseline
    ("deepfm_b131072", ["--workload", "deepfm", "--batch", "131072", "--steps", "20", "--warmup", "6"], 9.0),
)


def _borrow_cpu_baseline(out):
    """`neumf_100M` gets the CPU figure of the `neumf` leg as an UPPER BOUND: the port's dense step (zero-filled [n_rows, d]
    gradients, index_add, torch.optim over every row -- the reference's semantics) costs at least as much on tables ten times
    the size, and the four 100 M / 10 M-row tables with their dense gradients (226 GB) do not fit the time this command may take"""
    src, dst = out.get("neumf") or {}, out.get("neumf_100M")
    if isinstance(dst, dict) and "cpu_baseline" not in dst and isinstance(src.get("cpu_baseline"), dict) and "value" in dst:
        cb = dict(src["cpu_baseline"])
        cb["bound"] = "upper"
        cb["sample"] = ("UPPER BOUND, taken from the `neumf` leg of this run (same batch, same head, 10,000,001-item / 1,000,001-user tables): "
                        + str(cb.get("sample", "")) + " -- the dense optimizer step the port restates grows with the table rows")
        dst["cpu_baseline"] = cb


def secondary_single_gpu(args):
    """N = 1: every leg is its own process (a fault in one of them cannot take the contract line with it), one after the other
    on the same GPU, the steps / warm-up of SECONDARY_LEGS (20 after 5 unless the leg says otherwise), live roofline phases; the three legs whose CPU port finishes in seconds
    (NeuMF on the 10 M-item tables, SASRec, DeepFM at the reference's B = 1,024) carry their own `cpu_baseline` (2 timed fit()
    iterations of oracle/torch_port.py after one warm-up iteration) while the wall-clock budget (RC_BENCH_SECONDARY_BUDGET_S,
    default 85 s) has room for it; the budget keeps the driver's one command within minutes."""
    import subprocess
    budget = float(os.environ.get("RC_BENCH_SECONDARY_BUDGET_S", "85"))
    t0 = time.perf_counter()
    out = {}
    for name, extra, cpu_s in SECONDARY_LEGS:
        t_leg = time.perf_counter()
        left = budget - (t_leg - t0)
        if left < 8:
            out[name] = {"skipped": f"wall-clock budget of {budget:.0f} s for the secondary legs used up"}
            continue
        # (cpu_s = what the leg's CPU iterations cost on the pool's boxes; the legs after this one still need ~2 s each)
        with_cpu = cpu_s is not None and left > cpu_s + 2.5 * (len(SECONDARY_LEGS) - len(out))
        cmd = [sys.executable, os.path.abspath(__file__), "--steps", "20", "--warmup", "5"] + extra + \
              (["--cpu-steps", "2"] if with_cpu else ["--no-cpu-baseline"]) + ["--no-secondary"]
        try:
            p = subprocess.run(cmd, capture_output=True, text=True, timeout=max(left, 10.0))
            j = _last_json(p.stdout)
            out[name] = _summary(j) if (p.returncode == 0 and j) else {"failed": (p.stderr or p.stdout or "")[-300:], "rc": p.returncode}
        except subprocess.TimeoutExpired:
            out[name] = {"failed": f"did not finish within {left:.0f} s"}
        out[name]["wall_s"] = round(time.perf_counter() - t_leg, 1)
    _borrow_cpu_baseline(out)
    return out


# (RC_BENCH_SECONDARY_ROWS="items,users": smaller tables for this leg -- the tests of the N > 1 code path on one GPU set it)
_SEC_ITEMS, _SEC_USERS = (os.environ.get("RC_BENCH_SECONDARY_ROWS") or "100000001,10000001").split(",")
SHARDED_LEG = ["--workload", "neumf", "--items", _SEC_ITEMS, "--users", _SEC_USERS, "--steps", "20", "--warmup", "5",
               "--no-cpu-baseline", "--no-secondary"]


def secondary_sharded_in_process(args, rank, world, device, dist):
    """N > 1 under torch.distributed.run: BASELINE configs[3] -- NeuMF emb_size 128, num_neg 4 on the 100,000,001-item /
    10,000,001-user tables row-sharded over the ranks, SGD -- measured on the ranks of this job after the contract workload"""
    import copy
    import gc
    a = copy.copy(args)
    a.workload, a.items, a.users, a.emb_size, a.num_neg = "neumf", int(_SEC_ITEMS), int(_SEC_USERS), 128, 4
    a.steps, a.warmup, a.no_cpu_baseline = 20, 5, True
    gc.collect()
    torch.cuda.empty_cache()
    try:
        return {"neumf_100M": _summary(measure(a, rank, world, device, dist))}
    except Exception as e:   # the contract line survives a failure of this leg
        return {"neumf_100M": {"failed": repr(e)[-300:]}}


def launch_check(rank, world, backend):
    """RC_BENCH_LAUNCH_ONLY=1: the rank processes only rendezvous, reduce one number and print the line -- what the CPU
    test of the self-launcher runs (no GPU, no hot path; not a measurement)."""
    import torch.distributed as dist
    dist.init_process_group(backend, rank=rank, world_size=world)
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"launch_check": True, "n_gpus": world, "max_rank_plus_one": float(t.item()),
                          "env": {k: os.environ.get(k) for k in ("WORLD_SIZE", "MASTER_ADDR", "LOCAL_RANK")}}))


def measure(args, rank, world, device, dist):
    """one workload on the ranks of this job: warm-up, the timed steps between barriers, then (outside the timed region) the
    live roofline phases and the CPU baseline -> the JSON object of the line"""
    from rechorus_amd import engine

    batches = make_batches(args, device, seed=99 + rank) if args.workload not in ("sasrec", "deepfm") else None
    if args.workload == "deepfm":
        # N > 1 (BASELINE configs[4] is an 8-GPU config): replicated model, every rank trains its own B rows, ONE flat
        # all-reduce of the dense gradients per step, the same dense optimizer step everywhere (DataParallelDense)
        trainer = DeepfmBench(args, device, data_parallel=(world > 1 and args.parallel != "replicas"))
        batches = trainer.batches(args, device, seed=99 + rank)
    elif args.workload == "sasrec":
        if world > 1 and args.parallel != "replicas":
            raise SystemExit("--workload sasrec: the 8.7 K-row item table does not shard; use --parallel replicas for N > 1")
        trainer, batches = make_sasrec(args, device, engine, seed=99 + rank)
    elif args.workload == "neumf":
        trainer = make_neumf_trainer(args, world, device, engine)
    elif world == 1 or args.parallel == "replicas":
        gen = torch.Generator(device=device)
        gen.manual_seed(1234 + rank)
        U = torch.empty((args.users, args.emb_size), device=device).normal_(0, 0.01, generator=gen)
        I = torch.empty((args.items, args.emb_size), device=device).normal_(0, 0.01, generator=gen)
        trainer = engine.BprmfTrainer(U, I, opt=args.opt, lr=args.lr, l2=args.l2)
    else:
        # tables row-sharded over the ranks (id mod W), one GLOBAL batch of W*B tuples per step
        from rechorus_amd.sharded import ShardedBprmf
        trainer = ShardedBprmf(args.users, args.items, args.emb_size, opt=args.opt, lr=args.lr, l2=args.l2,
                               device=device, seed=1234)
        trainer.loss = None
        _step = trainer.step

        def step_and_keep(uid, iid, _step=_step, **kw):
            trainer.loss = _step(uid, iid, **kw)
            return trainer.loss
        trainer.step = step_and_keep
        trainer.lookahead = True

    def sync():
        torch.cuda.synchronize(device)
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize(device)

    run_step = lambda s: trainer.step(*batches[s % len(batches)])
    if world == 1 and args.workload == "bprmf" and not args.graph:
        # the step is given the following batch's ids (the reference's DataLoader runs ahead of the loop too): their grouping
        # front runs beside this step's row updates (rc_bprmf_train_step_ahead); same results bit for bit
        run_step = lambda s: trainer.step(*batches[s % len(batches)], next_batch=batches[(s + 1) % len(batches)])
    if world == 1 and args.workload == "neumf":
        # same contract for the NeuMF step: the following batch's bucket plan is built beside this step's table updates
        run_step = lambda s: trainer.step(*batches[s % len(batches)], next_batch=batches[(s + 1) % len(batches)])
    if getattr(trainer, "lookahead", False):
        # the sharded steps exchange per-destination split sizes; given the following batch they do that one step ahead
        run_step = lambda s: trainer.step(*batches[s % len(batches)], next_batch=batches[(s + 1) % len(batches)])
    if args.graph and world == 1 and args.workload != "sasrec":   # (SasrecTrainer(graph=True) captures its own step)
        # the whole step (about 20 launches, no host sync, shape-only grids) captured once per pooled
        # batch in a hipGraph and replayed: removes per-launch host cost, which dominates at B=256
        trainer.hyper.step = 1  # SGD/Adagrad ignore the step count; Adam's bias correction would freeze
        if args.opt == "Adam":
            raise SystemExit("--graph: Adam's bias-correction scalars are kernel arguments; use SGD/Adagrad")
        run_step(0)
        torch.cuda.synchronize(device)
        graphs = []
        side = torch.cuda.Stream(device=device)
        with torch.cuda.stream(side):
            for b in range(len(batches)):
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=side):
                    trainer.step(*batches[b])
                graphs.append(g)
        torch.cuda.synchronize(device)
        run_step = lambda s: graphs[s % len(graphs)].replay()

    for s in range(args.warmup):
        run_step(s)
    sync()
    t0 = time.perf_counter()
    for s in range(args.steps):
        run_step(args.warmup + s)   # the batch sequence continues: the last warm-up step announced this step's batch
    sync()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64,
                         device=device if args.dist_backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    loss = float(trainer.loss.item())
    if not np.isfinite(loss):
        raise SystemExit(f"non-finite loss {loss}")

    if args.workload == "deepfm":
        args.items, args.users, args.num_neg = DEEPFM_VOCAB["item_id"], DEEPFM_VOCAB["user_id"], 0
        workload_text = (f"DeepFMCTR fit step: MIND's F={len(DEEPFM_VOCAB)} single-valued fields {'/'.join(DEEPFM_VOCAB)} (c_day_f numeric: "
                         f"int64 days -> nn.Linear(1, d); MIND-like cardinalities, {DEEPFM_VOCAB['user_id']} users / "
                         f"{DEEPFM_VOCAB['item_id']} items), emb_size={args.emb_size}, MLP {args.mlp}, dropout={args.dropout:g}, BCE, B={args.batch} rows/GPU/step, "
                         f"optimizer={args.opt} (dense = torch.optim semantics, l2={args.l2:g}), hipGraph replay of the step, int64 ids, fp32")
    elif args.workload == "sasrec":
        workload_text = (f"SASRec fit step: emb_size={args.emb_size}, history_max={args.hist} (lengths uniform on 1..{args.hist}), "
                         f"{args.heads} heads, {args.layers} layer(s), num_neg={args.num_neg}, {args.items}-item table, Zipf(1.0) "
                         f"histories+positives, uniform negatives, B={args.batch} sequences/GPU/step, optimizer={args.opt} "
                         f"(row-wise, l2={args.l2:g}), {'hipGraph replay of the step (batch copied into static buffers), ' if getattr(args, 'sas_graph', False) else ''}"
                         f"int64 ids, fp32")
    else:
        workload_text = (f"{'NeuMF (hidden ' + str(args.hidden) + (', dropout ' + format(args.dropout, 'g') if args.dropout else '') + ')' if args.workload == 'neumf' else 'BPRMF'} fit step: emb_size={args.emb_size}, "
                         f"num_neg={args.num_neg}, {args.items}-item / {args.users}-user tables, Zipf(1.0) users+positives, "
                         f"uniform negatives, B={args.batch} tuples/GPU/step, optimizer={args.opt} "
                         f"(row-wise, l2={args.l2:g}), int64 ids, fp32")
    tuples = args.batch * args.steps * world
    out = {
        "metric": "ranked (1+K)-tuples/sec",
        "value": tuples / elapsed,
        "unit": "tuples/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": workload_text,
            "batch_per_gpu": args.batch, "num_neg": args.num_neg, "emb_size": args.emb_size,
            "n_items": args.items, "n_users": args.users, "optimizer": args.opt,
            "parallelism": "single GPU" if world == 1 else (
                f"{world} independent replicas" if args.parallel == "replicas" else
                (f"model replicated on {world} GPUs, dense gradients summed in one flat all-reduce over RCCL per step "
                 f"(sharded.DataParallelDense), global batch {world * args.batch}") if args.workload == "deepfm" else
                f"tables row-sharded over {world} GPUs (id mod W), {'rows travel' if args.workload == 'neumf' else 'owner-computes exchange'} over RCCL, "
                f"global batch {world * args.batch}"),
        },
        "final_loss": loss,
    }

    if rank == 0 and not args.no_roofline and hasattr(trainer, "profile_step"):
        # per-phase hipEvent timing on the launch stream, measured live (profiling steps are outside the timed region
        # above).  A profiled step is a step of the timed loop's steady state: the previous step announced its batch (its
        # plan is prepared, it starts with the fused kernel) and it announces the next one, whose plan runs on the second
        # stream beside these phases -- what the timed steps do.
        acc = {}
        reps = 10
        big = args.batch * (args.num_neg + 2) > 32768     # (smaller batches take the two-launch step: no plan, no look-ahead)
        nb = len(batches)
        for s in range(reps):
            a, b, c = batches[(2 * s) % nb], batches[(2 * s + 1) % nb], batches[(2 * s + 2) % nb]
            if big:
                trainer.step(*a, next_batch=b)
            ph = trainer.profile_step(*b, next_batch=c if big else None)
            if big:
                trainer.step(*c)     # consumes the plan the profiled step prepared
            for k, v in ph.items():
                acc[k] = acc.get(k, 0.0) + v / reps
        ab = algorithmic_bytes(args, batches)
        mine = {k: acc[k] for k in ("fused_fwd_bwd", "item_update", "user_update")}
        dom = max(mine, key=mine.get)
        achieved = ab[dom] / (acc[dom] * 1e-3) / 1e9
        out["roofline"] = {
            "bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBPS,
            # the committed PMC passes (tools/pmc_collect.sh) were taken on the default SGD workload
            "traffic": load_pmc_traffic(dom) if default_workload(args) else None,
            "algorithmic_bytes_per_launch": ab[dom], "avg_ms": acc[dom],
        }
        out["phases_ms"] = {k: round(v, 4) for k, v in acc.items()}
        if big:
            # Every kernel of the step with NOTHING beside it: a step that plans its own batch on ONE stream
            # (rc_bprmf_step_pipeline(2)) -- the plan's two halves, and what the dominant kernel reaches by itself.
            from rechorus_amd import _lib as _rl
            lib = _rl.load()
            alone = {}
            prev = lib.rc_bprmf_step_pipeline(2)
            try:
                for s in range(reps):
                    for k, v in trainer.profile_step(*batches[s % nb]).items():
                        alone[k] = alone.get(k, 0.0) + v / reps
            finally:
                lib.rc_bprmf_step_pipeline(prev)
            out["plan_ms"] = {"front_partition_bitmap": round(alone["sort_items"], 4), "bucket_pass": round(alone["segment_heads"], 4)}
            out["phases_alone_ms"] = {k: round(alone[k], 4) for k in ("fused_fwd_bwd", "item_update", "user_update")}
            out["roofline"]["note"] = ("live hipEvents in a steady-state step of the timed loop: the batch's plan was prepared ahead, the "
                                       "plan of the following batch runs on the second stream beside the kernel")
            out["roofline"]["alone"] = {"avg_ms": alone[dom], "achieved": ab[dom] / (alone[dom] * 1e-3) / 1e9,
                                        "frac": ab[dom] / (alone[dom] * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                                        "note": "same kernel with nothing beside it (bucket plan on one stream)"}
        if dom == "fused_fwd_bwd" and args.emb_size in (32, 64, 128):
            # what THIS box delivers for the kernel's access mix (rc_bench_mix: the batch's own ids, every occurrence reads its
            # row, the single-occurrence share writes it back, no arithmetic), in the currency of `achieved`: the kernel's
            # algorithmic bytes over the time the bare accesses take
            try:
                import ctypes as C
                from rechorus_amd import _lib as _rl
                ids = batches[0][1].reshape(-1)
                wf = ab["single_items"] / float(ids.numel())
                sink = torch.zeros(1, dtype=torch.float32, device=device)
                ms = C.c_float(0.0)
                _rl.call("rc_bench_mix", C.c_void_p(trainer.I.data_ptr()), args.emb_size, C.c_void_p(ids.data_ptr()), ids.numel(),
                         C.c_float(wf), 10, C.c_void_p(sink.data_ptr()), C.byref(ms), C.c_void_p(torch.cuda.current_stream().cuda_stream))
                ceiling = ab[dom] / (ms.value * 1e-3) / 1e9
                out["roofline"]["box_ceiling_gbps"] = ceiling
                out["roofline"]["frac_of_box_ceiling"] = achieved / ceiling
                out["roofline"]["box_ceiling_note"] = (f"rc_bench_mix on this box, in this run: the access mix of the kernel without its arithmetic "
                                                       f"({ids.numel()} row reads of {4 * args.emb_size} B, {wf:.2f} of them written back, int64 ids) takes "
                                                       f"{ms.value:.4f} ms; ceiling = the kernel's algorithmic bytes over that time")
            except Exception as e:
                out["roofline"]["box_ceiling_note"] = "rc_bench_mix failed: " + repr(e)[-200:]
        out["phases_gbps"] = {k: round(ab[k] / (acc[k] * 1e-3) / 1e9, 1)
                              for k in ("fused_fwd_bwd", "item_update", "user_update") if acc.get(k, 0) > 0}
        out["uniq_rows_per_step"] = {"items": ab["uniq_items"], "users": ab["uniq_users"],
                                     "items_single": ab["single_items"], "items_multi": ab["multi_items"],
                                     "multi_item_occurrences": ab["multi_item_occurrences"]}
        # compulsory bytes of the whole fused fwd+bwd+SGD step, rows once per
        # distinct row: read + write each touched row, ids, pred
        whole = 2 * (ab["uniq_items"] + ab["uniq_users"]) * 4 * args.emb_size + \
            8 * args.batch * (args.num_neg + 2) + 4 * args.batch * (args.num_neg + 1)
        out["step_effective_gbps"] = whole / (out["ms_per_step"] * 1e-3) / 1e9

    timed = getattr(trainer, "dp", None) or trainer
    if world > 1 and args.parallel == "sharded" and not args.no_roofline and hasattr(timed, "timing_ms"):
        # where a multi-GPU step spends its time (cuda events on every rank, outside the timed region):
        # collectives are inside the phases, so this also shows what xGMI costs
        timed.timing = []
        acc, reps = {}, 3
        for s in range(reps):
            run_step(args.warmup + args.steps + s)   # (continues the announced batch sequence)
            for k, v in timed.timing_ms().items():
                acc[k] = acc.get(k, 0.0) + v / reps
        timed.timing = None
        if rank == 0:
            out["sharded_phases_ms"] = {k: round(v, 4) for k, v in acc.items()}
    if world > 1 and rank == 0 and getattr(trainer, "wire", None):
        # bytes rank 0 moved over the links in its last step (ids out, rows in, gradient rows out), after the
        # per-destination de-duplication of the lookups (rechorus_amd/sharded.py::_Route)
        out["sharded_wire_bytes_rank0"] = trainer.wire

    if rank == 0 and world == 1 and not args.no_roofline and isinstance(trainer, (engine.NeumfTrainer, engine.SasrecTrainer)):
        out.update(model_roofline(args, trainer, batches, engine))

    if args.workload == "deepfm":
        out["metric"] = "labelled rows/sec (CTR: one tuple = one (user, item, context, label) row)"
    if rank == 0 and world == 1 and args.workload == "deepfm":
        if not args.no_roofline:
            out.update(deepfm_roofline(args, trainer, batches, engine, ms_per_step=out["ms_per_step"]))
        if not args.no_cpu_baseline:
            cpu_b = [({k: (v.cpu() if torch.is_tensor(v) else v) for k, v in batches[s % len(batches)][0].items()},)
                     for s in range(args.cpu_steps + 1)]
            out["cpu_baseline"] = cpu_baseline_deepfm(args, cpu_b)

    # MFMA-bound legs: what the matrix pipes of THIS box sustain (rc_bench_mfma: back-to-back fp32 MFMAs on every SIMD, no operand
    # traffic), measured in this run, beside the datasheet peak the fraction is quoted on
    rl = out.get("roofline")
    if rank == 0 and world == 1 and isinstance(rl, dict) and (rl.get("bound") == "mfma" or "frac_of_mfma_peak" in rl):
        try:
            import ctypes as C
            from rechorus_amd import _lib as _rl
            sink = torch.zeros(1, dtype=torch.float32, device=device)
            tf = C.c_float(0.0)
            _rl.call("rc_bench_mfma", 20000, C.c_void_p(sink.data_ptr()), C.byref(tf), C.c_void_p(torch.cuda.current_stream(device).cuda_stream))
            rl["box_mfma_tflops"] = round(float(tf.value), 1)
            ach_tf = rl["achieved"] if rl.get("unit") == "TFLOP/s" else rl["frac_of_mfma_peak"] * F32_MFMA_PEAK_TFLOPS
            rl["frac_of_box_mfma"] = ach_tf / float(tf.value)
            rl["box_mfma_note"] = ("rc_bench_mfma on this box, in this run: v_mfma_f32_32x32x2_f32 back to back on every SIMD (two waves, four "
                                   "accumulators each, no operand traffic)")
        except Exception as e:   # the ceiling is context, not the measurement
            rl["box_mfma_note"] = "rc_bench_mfma failed: " + repr(e)[-200:]

    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.workload == "bprmf":
        out["cpu_baseline"] = cpu_baseline(args)
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.workload in ("neumf", "sasrec"):
        n_cpu = args.cpu_steps + 1
        out["cpu_baseline"] = cpu_baseline_model(args, [tuple(t.cpu() for t in batches[s % len(batches)]) for s in range(n_cpu)])

    return out


def build_info():
    """what the loaded library was built from (rechorus_amd/librechorus_hip.build.json, written by csrc/build.py) + the sha256 of
    the file this process loaded"""
    try:
        from rechorus_amd.csrc import build as B
        info = json.load(open(B.INFO_PATH)) if os.path.exists(B.INFO_PATH) else {}
        sha = B.sha256_of(B.LIB_PATH)
        return {"flags": info.get("flags"), "hipcc --version": info.get("hipcc_version"), "so_sha256": sha,
                "so_matches_build_record": info.get("so_sha256") == sha, "arch": B.ARCH}
    except Exception as e:
        return {"error": repr(e)[-200:]}


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.gpus > 1 and world == 1:
        # not started by a launcher: be the launcher.  The contract workload's ranks first; then -- its own rank group, so that a
        # failure there cannot take the contract line with it -- the config-4 leg, attached as secondary.neumf_100M
        if not secondary_wanted(args):
            raise SystemExit(launch_ranks(args.gpus)[0])
        rc, text = launch_ranks(args.gpus, sys.argv[1:] + ["--no-secondary"], capture=True)
        line = _last_json(text)
        if rc != 0 or line is None:
            sys.stdout.write(text or "")
            raise SystemExit(rc or 1)
        rc2, text2 = launch_ranks(args.gpus, ["--gpus", str(args.gpus), "--dist-backend", args.dist_backend] + SHARDED_LEG, capture=True)
        j2 = _last_json(text2)
        line["secondary"] = {"neumf_100M": _summary(j2) if (rc2 == 0 and j2) else {"failed": (text2 or "")[-300:], "rc": rc2}}
        print(json.dumps(line))
        return
    if os.environ.get("RC_BENCH_LAUNCH_ONLY") == "1":
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        return launch_check(rank, world, "gloo")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    # RC_BENCH_ONE_DEVICE=1 + --dist-backend gloo: smoke-test the N>1 code path on a 1-GPU box
    # (all ranks on cuda:0, collectives staged through the host) -- not a measurement mode
    if os.environ.get("RC_BENCH_ONE_DEVICE") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dist_backend == "nccl":  # RCCL over xGMI
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(args.dist_backend, rank=rank, world_size=world)

    out = measure(args, rank, world, device, dist)
    if rank == 0:
        out["build"] = build_info()
    if rank == 0 and world == 1 and secondary_wanted(args):
        out["secondary"] = secondary_single_gpu(args)
    if world > 1 and secondary_wanted(args):
        # launched by torch.distributed.run: the config-4 leg on the same ranks (the self-launcher runs it as its own rank group)
        sec = secondary_sharded_in_process(args, rank, world, device, dist)
        if rank == 0:
            out["secondary"] = sec
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
