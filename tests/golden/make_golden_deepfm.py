"""Golden vectors for the FM / WideDeep / DeepFM heads FROM THE REFERENCE (models/context/*.py),
build container only:   PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_deepfm.py
Stores params, two batches with user / item / situation features, prediction, loss, autograd grads
of every parameter, and the parameters after two fit() iterations with Adam / SGD through the
reference's own optimizer construction."""
import importlib
import os
import sys
from types import SimpleNamespace

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_golden import HERE, _import_reference, _runner_args  # noqa: E402  (numpy alias shim too)

USER_F, ITEM_F, SIT_F = ["u_age_c", "u_gender_c"], ["i_category_c"], ["c_hour_c", "c_weekday_c"]
VOCAB = {"u_age_c": 7, "u_gender_c": 3, "i_category_c": 11, "c_hour_c": 24, "c_weekday_c": 7}
# MIND's field set (data/MIND_Large/MIND-large.ipynb cells 10-17: situation features hour / weekday / period and the NUMERIC
# c_day_f = days since the first impression, an integer column read from the csv as int64; item_meta: category, subcategory).
# Feature names are sorted inside their groups like helpers/ContextReader.py:43-50 does.
MIND = dict(user=[], item=["i_category_c", "i_subcategory_c"], sit=["c_day_f", "c_hour_c", "c_period_c", "c_weekday_c"],
            vocab={"i_category_c": 18, "i_subcategory_c": 300, "c_hour_c": 24, "c_period_c": 9, "c_weekday_c": 7},
            numeric={"c_day_f": ("int64", 7)})
# the same with the day as a float64 column (what pandas yields once a value has a fraction) and a numeric ITEM feature ([B, C])
MIND_FLOAT = dict(MIND, item=["i_age_f", "i_category_c", "i_subcategory_c"],
                  numeric={"c_day_f": ("float64", 7), "i_age_f": ("float64", 3)})


def make_case(name, cls_name, n_users, n_items, d, layers, B, K, loss_n, seed, spec=None):
    USER_F, ITEM_F, SIT_F, VOCAB, NUMERIC = globals()["USER_F"], globals()["ITEM_F"], globals()["SIT_F"], globals()["VOCAB"], {}
    if spec is not None:
        USER_F, ITEM_F, SIT_F, VOCAB, NUMERIC = spec["user"], spec["item"], spec["sit"], spec["vocab"], spec["numeric"]
    torch, _, BaseRunner = _import_reference()
    module = {"FM": "FM", "WideDeep": "WideDeep", "DeepFM": "DeepFM"}[cls_name.replace("CTR", "").replace("TopK", "")]
    cls = getattr(importlib.import_module("models.context." + module), cls_name)
    ctr = cls_name.endswith("CTR")
    torch.manual_seed(seed)
    rng = np.random.default_rng(seed)
    args = SimpleNamespace(device=torch.device("cpu"), model_path="", buffer=1, num_neg=K, dropout=0,
                           test_all=0, emb_size=d, layers=str(layers), loss_n=loss_n)
    fmax = dict(VOCAB, user_id=n_users, item_id=n_items)
    for f, (_, top) in NUMERIC.items():     # helpers/ContextReader.py:52-53 records max + 1 for every feature; unused for '*_f'
        fmax[f] = top
    corpus = SimpleNamespace(n_users=n_users, n_items=n_items, user_feature_names=USER_F, item_feature_names=ITEM_F,
                             situation_feature_names=SIT_F, feature_max=fmax)
    model = cls(args, corpus)
    with torch.no_grad():  # std 0.01 init leaves everything linear; scale up (see make_golden_neumf.py)
        for p in model.parameters():
            p.mul_(20.0)
    P0 = {k: v.detach().numpy().copy() for k, v in model.state_dict().items()}
    C = 1 if ctr else 1 + K
    out = {"meta": np.array([n_users, n_items, d, B, C, seed] + list(layers), dtype=np.int64),
           "fields": np.array(model.context_features)}
    for k, v in P0.items():
        out["P0/" + k] = v
    def column(f, size):
        if f not in NUMERIC:
            return rng.integers(0, VOCAB[f], size=size).astype(np.int64)
        dtype, top = NUMERIC[f]
        if dtype == "int64":
            return rng.integers(0, top, size=size).astype(np.int64)
        return (rng.random(size=size) * (top - 1)).astype(np.float64)

    item_cols = {f: column(f, n_items) for f in ITEM_F}     # item_meta: one value per item

    def batch():
        b = {"user_id": rng.integers(1, n_users, size=B).astype(np.int64), "item_id": rng.integers(1, n_items, size=(B, C)).astype(np.int64)}
        b["item_id"][:, 0] = b["item_id"][:, 0] % 5 + 1  # duplicates
        for f in USER_F + SIT_F:
            b[f] = column(f, B)
        for f in ITEM_F:
            b[f] = item_cols[f][b["item_id"]]
        if ctr:
            b["label"] = rng.integers(0, 2, size=(B, 1)).astype(np.int64)
        return b

    batches = [batch(), batch()]
    for n, b in enumerate(batches, 1):
        for k, v in b.items():
            out["b%d/%s" % (n, k)] = v

    def feed(b):
        f = {k: torch.from_numpy(v) for k, v in b.items()}
        f.update(batch_size=B, phase="train")
        return f

    model.zero_grad()
    o = model(feed(batches[0]))
    pred = o["prediction"]
    pred.retain_grad()
    loss = model.loss(o)
    loss.backward()
    out["pred"], out["loss"], out["gpred"] = pred.detach().numpy().copy(), np.float32(loss.item()), pred.grad.numpy().copy()
    for k, p in model.named_parameters():
        out["G/" + k] = p.grad.numpy().copy()

    for opt_name, lr, l2 in (("SGD", 0.05, 1e-3), ("Adam", 1e-3, 1e-4)):
        m = cls(args, corpus)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in P0.items()})
        runner = BaseRunner(_runner_args(BaseRunner, opt_name, lr, l2))
        m.optimizer = runner._build_optimizer(m)
        tag = "{}_l2{:g}".format(opt_name, l2)
        losses = []
        for b in batches:
            m.optimizer.zero_grad()
            ls = m.loss(m(feed(b)))
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


CASES = [
    ("deepfm_ctr_d64", "DeepFMCTR", 30, 50, 64, [64, 32], 48, 0, "BCE", 31),
    ("deepfm_ctr_d16", "DeepFMCTR", 20, 40, 16, [32], 33, 0, "BCE", 32),
    ("deepfm_fm_ctr_d32", "FMCTR", 20, 40, 32, [], 40, 0, "BCE", 33),
    ("deepfm_wd_ctr_d32", "WideDeepCTR", 20, 40, 32, [64], 40, 0, "BCE", 34),
    ("deepfm_topk_d64_k4", "DeepFMTopK", 25, 60, 64, [64], 24, 4, "BPR", 35),
    ("deepfm_fm_topk_d128_k9", "FMTopK", 15, 80, 128, [], 10, 9, "BPR", 36),
    # MIND's field set: numeric c_day_f beside the categorical fields (models/context/FM.py:38-41,47-48,51-52)
    ("deepfm_mind_ctr_d64", "DeepFMCTR", 40, 60, 64, [64, 32], 48, 0, "BCE", 41, MIND),
    ("deepfm_mind_fm_ctr_d32", "FMCTR", 30, 50, 32, [], 40, 0, "BCE", 42, MIND_FLOAT),
    ("deepfm_mind_wd_topk_d32_k3", "WideDeepTopK", 25, 40, 32, [64], 20, 3, "BPR", 43, MIND_FLOAT),
    ("deepfm_mind_topk_d16_k4", "DeepFMTopK", 25, 40, 16, [32], 24, 4, "BPR", 44, MIND),
]

if __name__ == "__main__":
    only = sys.argv[1:]
    for c in CASES:
        if not only or c[0] in only:
            make_case(*c)
