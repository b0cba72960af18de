"""Golden vectors for the NeuMF head FROM THE REFERENCE (models/general/NeuMF.py), build container
only:   PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_neumf.py
Stores params, a batch, prediction, loss, autograd grads of every parameter, and the parameters
after two fit() iterations with Adam / SGD through the reference's own optimizer construction.
neumfdrop_*: the reference model in TRAINING mode with --dropout p, its nn.Dropout swapped for a module
that applies a GIVEN keep mask with nn.Dropout's arithmetic (zero, or scale by 1/(1-p)); the mask is the
counter-based one of rc_neumf_fwd_dropout (oracle/neumf_oracle.dropout_keep), since torch's own random
stream is not something another implementation can reproduce."""
import os
import sys
from types import SimpleNamespace

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_golden import HERE, _import_reference, _runner_args  # noqa: E402  (numpy alias shim too)


def make_case(name, n_users, n_items, d, layers, B, K, seed):
    torch, _, BaseRunner = _import_reference()
    from models.general.NeuMF import NeuMF
    torch.manual_seed(seed)
    rng = np.random.default_rng(seed)
    args = SimpleNamespace(device=torch.device("cpu"), model_path="", buffer=1, num_neg=K, dropout=0,
                           test_all=0, emb_size=d, layers=str(layers))
    corpus = SimpleNamespace(n_users=n_users, n_items=n_items)
    model = NeuMF(args, corpus)
    # the reference initialises everything with std 0.01, which leaves ReLU inputs ~1e-3 and the
    # MLP path nearly linear; scale up so that both ReLU branches and the mask matter
    with torch.no_grad():
        for p in model.parameters():
            p.mul_(20.0)
    P0 = {k: v.detach().numpy().copy() for k, v in model.state_dict().items()}
    out = {"meta": np.array([n_users, n_items, d, B, K, seed] + list(layers), dtype=np.int64)}
    for k, v in P0.items():
        out["P0/" + k] = v

    def batch():
        uid = rng.integers(1, n_users, size=B).astype(np.int64)
        iid = rng.integers(1, n_items, size=(B, 1 + K)).astype(np.int64)
        iid[:, 0] = iid[:, 0] % 9 + 1  # duplicates
        return uid, iid
    (uid, iid), (uid2, iid2) = batch(), batch()
    out.update(uid=uid, iid=iid, uid2=uid2, iid2=iid2)

    def feed(u, i):
        return {"user_id": torch.from_numpy(u), "item_id": torch.from_numpy(i), "batch_size": len(u), "phase": "train"}

    model.zero_grad()
    o = model(feed(uid, iid))
    pred = o["prediction"]
    pred.retain_grad()
    loss = model.loss(o)
    loss.backward()
    out["pred"], out["loss"], out["gpred"] = pred.detach().numpy().copy(), np.float32(loss.item()), pred.grad.numpy().copy()
    for k, p in model.named_parameters():
        out["G/" + k] = p.grad.numpy().copy()

    for opt_name, lr, l2 in (("SGD", 0.05, 1e-3), ("Adam", 1e-3, 1e-4)):
        m = NeuMF(args, corpus)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in P0.items()})
        runner = BaseRunner(_runner_args(BaseRunner, opt_name, lr, l2))
        m.optimizer = runner._build_optimizer(m)
        tag = "{}_l2{:g}".format(opt_name, l2)
        losses = []
        for step, (u, i) in enumerate(((uid, iid), (uid2, iid2)), 1):
            m.optimizer.zero_grad()
            od = m(feed(u, i))
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


def make_dropout_case(name, n_users, n_items, d, l1, B, K, p, seed):
    torch, _, _ = _import_reference()
    from models.general.NeuMF import NeuMF
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from oracle import neumf_oracle
    torch.manual_seed(seed)
    rng = np.random.default_rng(seed)
    args = SimpleNamespace(device=torch.device("cpu"), model_path="", buffer=1, num_neg=K, dropout=p,
                           test_all=0, emb_size=d, layers=str([l1]))
    model = NeuMF(args, SimpleNamespace(n_users=n_users, n_items=n_items))
    with torch.no_grad():
        for q in model.parameters():
            q.mul_(20.0)
    model.train()
    # nn.Dropout's arithmetic on a fresh tensor: survivors are x * 1/(1-p) in fp32
    x = torch.rand(4096) + 0.5
    y = model.dropout_layer(x)
    scale = np.float32(1) / (np.float32(1) - np.float32(p))
    assert torch.equal(y[y != 0], (x * float(scale))[y != 0]) and abs(float((y == 0).float().mean()) - p) < 0.05
    mask_seed = 0x1234ABCD5678 + seed
    keep = neumf_oracle.dropout_keep(mask_seed, B * (1 + K), l1, p)

    class GivenMask(torch.nn.Module):
        def forward(self, h):  # h [B, C, l1]
            return h * torch.from_numpy(keep).view(h.shape)
    model.dropout_layer = GivenMask()
    out = {"meta": np.array([n_users, n_items, d, B, K, seed, l1], dtype=np.int64), "p": np.float32(p),
           "mask_seed": np.int64(mask_seed), "keep_rate": np.float32((keep != 0).mean())}
    for k, v in model.state_dict().items():
        out["P0/" + k] = v.detach().numpy().copy()
    uid = rng.integers(1, n_users, size=B).astype(np.int64)
    iid = rng.integers(1, n_items, size=(B, 1 + K)).astype(np.int64)
    out.update(uid=uid, iid=iid)
    o = model({"user_id": torch.from_numpy(uid), "item_id": torch.from_numpy(iid), "batch_size": B, "phase": "train"})
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
    ("neumf_d64_l64_k4", 30, 90, 64, [64], 40, 4, 21),
    ("neumf_d128_l64_k4", 20, 60, 128, [64], 30, 4, 22),
    ("neumf_d32_l32_k9", 20, 60, 32, [32], 16, 9, 23),
    ("neumf_d64_l64_k99", 20, 300, 64, [64], 5, 99, 24),
]

# towers with several hidden layers (--layers '[64,32]' ...): the mirror runs them on rc_linear_fwd / rc_linear_bwd
ML_CASES = [
    ("neumfml_d32_l64x32_k4", 25, 80, 32, [64, 32], 36, 4, 41),
    ("neumfml_d64_l128x64x32_k9", 20, 70, 64, [128, 64, 32], 20, 9, 42),
]

DROPOUT_CASES = [
    ("neumfdrop_d64_l64_k4_p0.2", 30, 90, 64, 64, 40, 4, 0.2, 31),
    ("neumfdrop_d32_l128_k9_p0.5", 20, 60, 32, 128, 16, 9, 0.5, 32),
]

if __name__ == "__main__":
    if "--ml-only" in sys.argv:
        for c in ML_CASES:
            make_case(*c)
        sys.exit(0)
    for c in CASES + ML_CASES:
        make_case(*c)
    for c in DROPOUT_CASES:
        make_dropout_case(*c)
