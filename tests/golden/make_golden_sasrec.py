"""Golden vectors for SASRec FROM THE REFERENCE (models/sequential/SASRec.py + utils/layers.py),
build container only:   PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_sasrec.py

sasrecdrop_*: the reference model in TRAINING mode with --dropout p, the two nn.Dropout modules of every
TransformerLayer (utils/layers.py:104,114) swapped for modules that apply a GIVEN keep-and-scale mask -- the
counter-based one of rc_sasrec_batch_fwd_dropout (oracle/sasrec_oracle.dropout_keep), since torch's own random
stream cannot be reproduced by another implementation.  Everything else (the order of operations, LayerNorm,
autograd) is the reference's."""
import os
import sys
from types import SimpleNamespace

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_golden import HERE, _import_reference, _runner_args  # noqa: E402


def make_case(name, n_items, d, n_layers, n_heads, hist_max, B, K, seed):
    torch, _, BaseRunner = _import_reference()
    from models.sequential.SASRec import SASRec
    torch.manual_seed(seed)
    rng = np.random.default_rng(seed)
    args = SimpleNamespace(device=torch.device("cpu"), model_path="", buffer=1, num_neg=K, dropout=0, test_all=0,
                           emb_size=d, num_layers=n_layers, num_heads=n_heads, history_max=hist_max)
    corpus = SimpleNamespace(n_users=10, n_items=n_items)
    model = SASRec(args, corpus)
    with torch.no_grad():  # std 0.01 init keeps attention uniform and ReLUs half-dead: scale up
        for n, p in model.named_parameters():
            if "layer_norm" not in n:
                p.mul_(12.0)
            else:
                p.add_(torch.randn_like(p) * 0.1)
    P0 = {k: v.detach().numpy().copy() for k, v in model.state_dict().items()}
    out = {"meta": np.array([n_items, d, n_layers, n_heads, hist_max, B, K, seed], dtype=np.int64)}
    for k, v in P0.items():
        out["P0/" + k] = v

    def batch():
        lengths = rng.integers(1, hist_max + 1, size=B).astype(np.int64)
        lengths[0] = hist_max
        L = int(lengths.max())
        hist = np.zeros((B, L), dtype=np.int64)
        for b in range(B):
            hist[b, :lengths[b]] = rng.integers(1, n_items, size=lengths[b])
        iid = rng.integers(1, n_items, size=(B, 1 + K)).astype(np.int64)
        return hist, lengths, iid
    b1, b2 = batch(), batch()
    for tag, (h, ln, i) in (("", b1), ("2", b2)):
        out["hist" + tag], out["len" + tag], out["iid" + tag] = h, ln, i

    def feed(h, ln, i):
        return {"history_items": torch.from_numpy(h), "lengths": torch.from_numpy(ln), "item_id": torch.from_numpy(i),
                "user_id": torch.zeros(len(ln), dtype=torch.long), "batch_size": len(ln), "phase": "train"}

    model.zero_grad()
    o = model(feed(*b1))
    pred = o["prediction"]
    pred.retain_grad()
    loss = model.loss(o)
    loss.backward()
    out["pred"], out["loss"], out["gpred"] = pred.detach().numpy().copy(), np.float32(loss.item()), pred.grad.numpy().copy()
    for k, p in model.named_parameters():
        out["G/" + k] = p.grad.numpy().copy()

    for opt_name, lr, l2 in (("SGD", 0.05, 1e-3), ("Adam", 1e-3, 1e-4)):
        m = SASRec(args, corpus)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in P0.items()})
        runner = BaseRunner(_runner_args(BaseRunner, opt_name, lr, l2))
        m.optimizer = runner._build_optimizer(m)
        tag = "{}_l2{:g}".format(opt_name, l2)
        losses = []
        for bt in (b1, b2):
            m.optimizer.zero_grad()
            od = m(feed(*bt))
            ls = m.loss(od)
            ls.backward()
            m.optimizer.step()
            losses.append(ls.item())
        for k, v in m.state_dict().items():
            out["{}/{}".format(tag, k)] = v.detach().numpy().copy()
        out[tag + "_losses"] = np.array(losses, dtype=np.float32)
        out[tag + "_hyper"] = np.array([lr, l2], dtype=np.float64)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) >> 10, "KiB")


def make_dropout_case(name, n_items, d, n_layers, n_heads, hist_max, B, K, p, seed):
    torch, _, _ = _import_reference()
    from models.sequential.SASRec import SASRec
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from oracle import sasrec_oracle
    torch.manual_seed(seed)
    rng = np.random.default_rng(seed)
    args = SimpleNamespace(device=torch.device("cpu"), model_path="", buffer=1, num_neg=K, dropout=p, test_all=0,
                           emb_size=d, num_layers=n_layers, num_heads=n_heads, history_max=hist_max)
    model = SASRec(args, SimpleNamespace(n_users=10, n_items=n_items))
    with torch.no_grad():
        for n, q in model.named_parameters():
            if "layer_norm" not in n:
                q.mul_(12.0)
            else:
                q.add_(torch.randn_like(q) * 0.1)
    lengths = rng.integers(1, hist_max + 1, size=B).astype(np.int64)
    lengths[0] = hist_max
    L = int(lengths.max())
    hist = np.zeros((B, L), dtype=np.int64)
    for b in range(B):
        hist[b, :lengths[b]] = rng.integers(1, n_items, size=lengths[b])
    iid = rng.integers(1, n_items, size=(B, 1 + K)).astype(np.int64)
    mask_seed = int(rng.integers(1, 2 ** 62))
    keep = sasrec_oracle.dropout_keep(mask_seed, lengths, L, d, n_layers, p)

    class GivenMask(torch.nn.Module):
        def __init__(self, m):
            super().__init__()
            self.m = torch.from_numpy(m)

        def forward(self, x):
            return x * self.m

    for l, block in enumerate(model.transformer_block):
        assert isinstance(block.dropout1, torch.nn.Dropout) and block.dropout1.p == p
        block.dropout1, block.dropout2 = GivenMask(keep[2 * l]), GivenMask(keep[2 * l + 1])
    model.train()
    out = {"meta": np.array([n_items, d, n_layers, n_heads, hist_max, B, K, seed], dtype=np.int64),
           "p": np.float32(p), "mask_seed": np.int64(mask_seed), "hist": hist, "len": lengths, "iid": iid}
    for k, v in model.state_dict().items():
        out["P0/" + k] = v.detach().numpy().copy()
    model.zero_grad()
    o = model({"history_items": torch.from_numpy(hist), "lengths": torch.from_numpy(lengths), "item_id": torch.from_numpy(iid),
               "user_id": torch.zeros(B, dtype=torch.long), "batch_size": B, "phase": "train"})
    pred = o["prediction"]
    pred.retain_grad()
    loss = model.loss(o)
    loss.backward()
    out["pred"], out["loss"], out["gpred"] = pred.detach().numpy().copy(), np.float32(loss.item()), pred.grad.numpy().copy()
    for k, q in model.named_parameters():
        out["G/" + k] = q.grad.numpy().copy()
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) >> 10, "KiB")


CASES = [
    # name,                 n_items, d, layers, heads, hist_max, B, K, seed
    ("sasrec_d64_l1_h1",       80, 64, 1, 1, 20, 12, 9, 31),
    ("sasrec_d64_l1_h4_L50",  120, 64, 1, 4, 50, 8, 19, 32),
    ("sasrec_d64_l2_h2",       80, 64, 2, 2, 12, 10, 5, 33),
    ("sasrec_d32_l1_h4",       60, 32, 1, 4, 7, 9, 3, 34),
    ("sasrec_d64_l1_h2_L100", 150, 64, 1, 2, 100, 9, 7, 35),    # more than 64 positions: one block, the one-row path's 128-row tile
    # outside the register-resident kernels' envelope: the shape-generic layers of csrc/seq_layers.hip
    ("sasrec_d128_l1_h4",     100, 128, 1, 4, 20, 10, 7, 36),
    ("sasrec_d128_l2_h8_L100", 150, 128, 2, 8, 100, 6, 5, 37),
    ("sasrec_d64_l3_h4_L200", 200, 64, 3, 4, 200, 5, 5, 38),   # history beyond 128, three blocks
    ("sasrec_d48_l2_h3",       70, 48, 2, 3, 15, 9, 4, 39),    # an emb_size that is no power of two, three heads of 16
    ("sasrec_d64_l5_h2",       60, 64, 5, 2, 10, 7, 3, 40),    # five blocks
]

DROP_CASES = [
    # name,                        n_items, d, layers, heads, hist_max, B, K, p, seed
    ("sasrecdrop_d64_l1_h4_p0.2",     90, 64, 1, 4, 50, 10, 9, 0.2, 41),
    ("sasrecdrop_d32_l2_h2_p0.5",     60, 32, 2, 2, 11, 12, 4, 0.5, 42),
    ("sasrecdrop_d64_l2_h4_L100_p0.2", 120, 64, 2, 4, 100, 8, 6, 0.2, 43),    # dropout with a history beyond 64 and two blocks
    ("sasrecdrop_d128_l1_h2_p0.3",    90, 128, 1, 2, 30, 8, 6, 0.3, 44),
]

if __name__ == "__main__":
    only = [a.split("=", 1)[1] for a in sys.argv if a.startswith("--only=")]
    if only:   # one named case (the committed fixtures of the others stay as they are)
        for c in CASES:
            if c[0] in only:
                make_case(*c)
        for c in DROP_CASES:
            if c[0] in only:
                make_dropout_case(*c)
        sys.exit(0)
    if "--dropout-only" not in sys.argv:
        for c in CASES:
            make_case(*c)
    for c in DROP_CASES:
        make_dropout_case(*c)
