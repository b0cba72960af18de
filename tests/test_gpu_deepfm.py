"""GPU parity of the FM-family path: rc_fm_second_order_* and rc_bce_prob_fwd_bwd vs the numpy oracle,
the narrow-row (d <= 4) segmented sum behind the [vocab, 1] first-order tables, and the mirror's
FM / WideDeep / DeepFM model files vs the reference's own outputs (tests/golden/deepfm_*.npz)."""
import argparse
import os
import re
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT, assert_close, assert_update_close, load_golden
from oracle import deepfm_oracle as DO
from test_oracle_deepfm import CASES, batch, cancel_floor, params

pytestmark = pytest.mark.gpu

PLUGIN = os.path.join(ROOT, "rechorus_amd", "rechorus")
if PLUGIN not in sys.path:
    sys.path.insert(0, PLUGIN)


@pytest.fixture(scope="module")
def eng(cuda):
    from rechorus_amd import engine
    return engine


@pytest.mark.parametrize("d", [16, 32, 64, 128, 24, 5, 200])
def test_fm_second_order_vs_oracle(d, cuda, eng):
    rng = np.random.default_rng(d)
    for shape, F in (((1,), 1), ((63,), 2), ((33, 5), 7), ((1000,), 39), ((7, 100), 3)):
        V = rng.normal(0, 0.5, size=shape + (F, d)).astype(np.float32)
        gout = rng.normal(size=shape).astype(np.float32)
        Vd = torch.from_numpy(V).to(cuda)
        out = eng.fm_second_order(Vd)
        assert out.shape == shape
        # cancellation: 0.5((sum v)^2 - sum v^2) is a difference of two O(F d) sums
        floor = 4e-7 * float((V.astype(np.float64) ** 2).sum(axis=(-1, -2)).max())
        assert_close(out.cpu().numpy(), DO.fm_second_order(V), what=f"fm2 F={F}", abs_floor=floor)
        dV = eng.fm_second_order_bwd(Vd, torch.from_numpy(gout).to(cuda))
        assert_close(dV.cpu().numpy(), DO.fm_second_order_bwd(V, gout), what=f"dV F={F}", atol_scale=1e-6)
    empty = eng.fm_second_order(torch.empty((0, 3, d), device=cuda))
    assert empty.shape == (0,)


def test_fm_second_order_autograd_and_unsupported_width(cuda):
    from rechorus_amd import nn as hnn
    rng = np.random.default_rng(1)
    for d in (64, 24):  # 24: the generic-width kernel
        V = torch.from_numpy(rng.normal(size=(9, 4, 5, d)).astype(np.float32)).to(cuda).requires_grad_(True)
        w = torch.from_numpy(rng.normal(size=(9, 4)).astype(np.float32)).to(cuda)
        (hnn.fm_second_order(V) * w).sum().backward()
        want = DO.fm_second_order_bwd(V.detach().cpu().numpy(), w.cpu().numpy())
        assert_close(V.grad.cpu().numpy(), want, what=f"autograd d={d}", atol_scale=2e-6)


def test_bce_vs_oracle_and_torch_clamps(cuda, eng):
    rng = np.random.default_rng(2)
    for n in (1, 77, 4096, 100003):
        p = (1.0 / (1.0 + np.exp(-rng.normal(0, 3, size=n)))).astype(np.float32)
        y = rng.integers(0, 2, size=n).astype(np.float32)
        loss, gp = eng.bce_prob(torch.from_numpy(p).to(cuda), torch.from_numpy(y).to(cuda))
        assert_close(loss.cpu().numpy()[0], DO.bce(p, y), what=f"bce n={n}", rtol=2e-5)
        assert_close(gp.cpu().numpy(), DO.bce_grad(p, y), what=f"dbce n={n}", rtol=2e-5)
    # saturated probabilities: torch clamps log at -100 and p(1-p) at 1e-12
    p = torch.tensor([0.0, 1.0, 1e-30, 1.0, 0.0, 1e-8], device=cuda)
    y = torch.tensor([1.0, 0.0, 1.0, 1.0, 0.0, 0.0], device=cuda)
    loss, gp = eng.bce_prob(p, y)
    pt = p.detach().clone().requires_grad_(True)
    want = torch.nn.functional.binary_cross_entropy(pt, y)
    want.backward()
    assert_close(loss.cpu().numpy()[0], want.item(), what="clamped loss")
    assert_close(gp.cpu().numpy(), pt.grad.cpu().numpy(), what="clamped grad")
    loss, none = eng.bce_prob(p, y, need_grad=False)
    assert none is None and abs(loss.item() - want.item()) < 1e-4


@pytest.mark.parametrize("d", [1, 2, 3, 4])
def test_narrow_tables_with_hot_rows(d, cuda):
    """[vocab, 1] first-order tables: a handful of rows, each hit thousands of times"""
    from rechorus_amd.nn import HipEmbedding
    rng = np.random.default_rng(d)
    for n_rows, shape in ((3, (5000,)), (24, (4096, 1)), (7, (300, 11)), (1000, (50,))):
        emb = HipEmbedding(n_rows, d).to(cuda)
        ids = rng.integers(0, n_rows, size=shape).astype(np.int64)
        ids.reshape(-1)[:3] = n_rows - 1
        out = emb(torch.from_numpy(ids).to(cuda))
        assert np.array_equal(out.detach().cpu().numpy(), emb.weight.detach().cpu().numpy()[ids])
        coef = rng.normal(size=shape + (d,)).astype(np.float32)
        (out * torch.from_numpy(coef).to(cuda)).sum().backward()
        want = np.zeros((n_rows, d), dtype=np.float64)
        np.add.at(want, ids.reshape(-1), coef.reshape(-1, d).astype(np.float64))
        # thousands of signed terms per row: the floor is the round-off of that sum, not of the result
        floor = 2e-7 * float(np.abs(coef).sum()) / n_rows
        assert_close(emb.weight.grad.cpu().numpy(), want, what=f"narrow grad rows={n_rows}", abs_floor=floor)


# ---- model files ---------------------------------------------------------------------------------------

def _build(case, g, cuda, loss_n=None):
    import importlib
    cls_name = {"deepfm_ctr": "DeepFMCTR", "deepfm_fm_ctr": "FMCTR", "deepfm_wd_ctr": "WideDeepCTR",
                "deepfm_topk": "DeepFMTopK", "deepfm_fm_topk": "FMTopK", "deepfm_wd_topk": "WideDeepTopK"}[
        re.match(r"(.*?)_d\d+", case).group(1).replace("_mind", "")]
    module = cls_name.replace("CTR", "").replace("TopK", "")
    cls = getattr(importlib.import_module("models.context." + module), cls_name)
    n_users, n_items, d, B, C = (int(x) for x in g["meta"][:5])
    layers = [int(x) for x in g["meta"][6:]]
    ctr = cls_name.endswith("CTR")
    args = argparse.Namespace(device=cuda, model_path="", buffer=1, num_neg=C - 1, dropout=0, test_all=0, emb_size=d,
                              layers=str(layers), loss_n="BCE" if ctr else "BPR")
    # the corpus the golden's model was built on, read back from the golden itself: feature groups by prefix (sorted inside their
    # groups, helpers/ContextReader.py:43-50), vocabularies = the tables' row counts (numeric '*_f' features own no table)
    fields = [str(f) for f in g["fields"]]
    group = lambda pre: [f for f in fields if f.startswith(pre)]
    fmax = {f: int(g["P0/context_embedding.%s.weight" % f].shape[0]) if DO.is_categorical(f) else 7 for f in fields}
    corpus = argparse.Namespace(n_users=n_users, n_items=n_items, user_feature_names=group("u_"), item_feature_names=group("i_"),
                                situation_feature_names=group("c_"), feature_max=fmax)
    model = cls(args, corpus)
    assert model.context_features == fields
    sd = {k[3:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("P0/")}
    assert set(sd) == set(model.state_dict()), "state_dict keys differ from the reference's"
    model.load_state_dict(sd)
    return model.to(cuda), B


def case_is_ctr(case):
    return "_ctr_" in case


def _feed(g, n, B, cuda):
    f = {k: torch.from_numpy(v).to(cuda) for k, v in batch(g, n).items()}
    f.update(batch_size=B, phase="train")
    return f


@pytest.mark.parametrize("case", CASES)
def test_model_file_matches_reference(case, cuda):
    g = load_golden(case)
    model, B = _build(case, g, cuda)
    b = batch(g, 1)
    # In training mode the CTR variants compute sum-of-terms + sigmoid + BCE in ONE kernel (rc_ctr_head_fwd_bwd): the
    # probability leaves it without a graph of its own, so d loss / d prediction is checked on the op-by-op path (eval mode;
    # dropout is 0 in the goldens) and the fused training path against prediction, loss and every parameter gradient.
    for mode in ("eval", "train"):
        model.train(mode == "train")
        model.zero_grad()
        out = model(_feed(g, 1, B, cuda))
        fused = "loss" in out
        assert fused == (mode == "train" and case_is_ctr(case)), (mode, case)
        if not fused:
            out["prediction"].retain_grad()
        assert_close(out["prediction"].detach().cpu().numpy(), g["pred"], what=mode + " prediction", rtol=2e-5)
        loss = model.loss(out)
        loss.backward()
        assert_close(loss.item(), g["loss"], what=mode + " loss", rtol=2e-5)
        if not fused:
            assert_close(out["prediction"].grad.cpu().numpy(), g["gpred"], what="dloss/dprediction", rtol=2e-5)
        for name, p in model.named_parameters():
            assert_close(p.grad.cpu().numpy(), g["G/" + name], what=mode + " grad " + name, rtol=2e-5, atol_scale=5e-5,
                         abs_floor=cancel_floor(name, g, b))


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("tag,opt", [("SGD_l20.001", "SGD"), ("Adam_l20.0001", "Adam")])
def test_two_fit_iterations_match_reference(case, tag, opt, cuda):
    """model(batch) -> loss -> backward -> HipOptimizer.step twice, vs the reference's own fit()"""
    from helpers.BaseRunner import BaseRunner
    g = load_golden(case)
    lr, l2 = (float(x) for x in g[tag + "_hyper"])
    model, B = _build(case, g, cuda)
    a = BaseRunner.parse_runner_args(argparse.ArgumentParser()).parse_args([])
    a.train, a.log_file, a.optimizer, a.lr, a.l2, a.engine = 1, "/tmp/rechorus_amd_test/log.txt", opt, lr, l2, "dense"
    model.optimizer = BaseRunner(a)._build_optimizer(model)
    for step in (1, 2):
        model.optimizer.zero_grad()
        loss = model.loss(model(_feed(g, step, B, cuda)))
        loss.backward()
        model.optimizer.step()
        assert_close(loss.item(), g[tag + "_losses"][step - 1], what=f"loss {step}", rtol=5e-5)
    P0, want = params(g), params(g, tag + "/")
    ex = 1e-3 * lr if opt == "Adam" else 0.0
    for name, p in model.state_dict().items():
        # Adam normalises analytically-zero gradients (see cancel_floor) to +-lr: not comparable
        if opt == "Adam" and cancel_floor(name, g, batch(g, 1)) > 0:
            assert np.abs(p.cpu().numpy() - P0[name]).max() <= 2.5 * lr
            continue
        # elements whose first-step gradient is summation-order noise (|g| < 1e-7 in the reference's own autograd: e.g. a hidden unit
        # whose BPR gradients cancel over the row's candidates) are normalised by Adam to a step of size ~lr in either direction
        ill = ((np.abs(g["G/" + name]) < 1e-7) & (g["G/" + name] != 0)) if opt == "Adam" else None     # (0: a table row the batch did not touch)
        assert ill is None or ill.mean() <= 0.02, (name, float(ill.mean()))
        assert_update_close(p.cpu().numpy(), P0[name], want[name], what=name, extra_atol=ex, outlier_atol=2 * lr, exclude=ill)


# ---- CLI ------------------------------------------------------------------------------------------------

@pytest.fixture(scope="module")
def ctx_root(tmp_path_factory):
    from synth_data import make_context_dataset
    root = str(tmp_path_factory.mktemp("ctxdata"))
    make_context_dataset(root, "ctr", n_users=300, n_items=150, per_user=20, ctr=True, seed=3)
    make_context_dataset(root, "topk", n_users=300, n_items=150, per_user=12, ctr=False, seed=4)
    return root


def test_cli_deepfm_ctr(ctx_root, tmp_path, cuda):
    import main
    log = str(tmp_path / "log" / "run.txt")
    res = main.run(["--model_name", "DeepFM", "--model_mode", "CTR", "--emb_size", "16", "--layers", "[32]", "--lr", "5e-3",
                    "--l2", "0", "--loss_n", "BCE", "--dataset", "ctr", "--path", ctx_root + "/", "--epoch", "8",
                    "--batch_size", "256", "--num_workers", "0", "--regenerate", "1", "--metric", "AUC,ACC,LOG_LOSS",
                    "--include_item_features", "1", "--include_user_features", "1", "--include_situation_features", "1",
                    "--log_file", log, "--model_path", str(tmp_path / "model" / "m.pt"), "--save_final_results", "1"])
    text = open(log).read()
    rec = (tmp_path / "log" / "run" / "rec-DeepFMCTR-test.csv").read_text().splitlines()
    assert rec[0].split("\t") == ["user_id", "item_id", "pCTR", "label"] and len(rec) > 100
    losses = [float(x) for x in re.findall(r"Epoch \d+\s+loss=([0-9.]+)", text)]
    assert len(losses) >= 2 and losses[-1] < losses[0], losses
    before = float(re.search(r"Test Before Training: \(.*?AUC:([0-9.]+)", text).group(1))
    after = float(re.search(r"AUC:([0-9.]+)", res["test"]).group(1))
    assert after > 0.6 and after > before, (before, after)  # clicks follow user-age x item-category affinity


def test_cli_fm_topk_with_context(ctx_root, tmp_path, cuda):
    import main
    log = str(tmp_path / "log" / "run.txt")
    res = main.run(["--model_name", "FM", "--model_mode", "TopK", "--emb_size", "32", "--lr", "5e-3", "--l2", "0",
                    "--dataset", "topk", "--path", ctx_root + "/", "--epoch", "6", "--num_neg", "2", "--batch_size", "256",
                    "--num_workers", "0", "--regenerate", "1", "--topk", "5,10", "--include_item_features", "1",
                    "--include_user_features", "1", "--log_file", log, "--model_path", str(tmp_path / "model" / "m.pt"),
                    "--save_final_results", "0"])
    text = open(log).read()
    losses = [float(x) for x in re.findall(r"Epoch \d+\s+loss=([0-9.]+)", text)]
    assert len(losses) >= 2 and losses[-1] < losses[0], losses
    before = float(re.search(r"Test Before Training: \(HR@5:([0-9.]+)", text).group(1))
    after = float(re.search(r"HR@5:([0-9.]+)", res["test"]).group(1))
    assert after > before, (before, after)


def test_gather_fields_equals_per_field_gathers(cuda):
    """rc_gather_fields (one launch, table-pointer array) == stacking F rc_gather_rows; its composite-key
    backward == F separate embedding backwards"""
    from rechorus_amd import nn as hnn
    rng = np.random.default_rng(12)
    for d, B, C in ((64, 33, 5), (1, 40, 1), (6, 7, 3), (16, 1, 1)):
        vocab = [11, 300, 5, 70, 2]
        per_row = [True, False, True, False, False]
        tables = [torch.from_numpy(rng.normal(size=(v, d)).astype(np.float32)).to(cuda).requires_grad_(True) for v in vocab]
        ids = [torch.from_numpy(rng.integers(0, v, size=(B,) if pr else (B, C)).astype(np.int64)).to(cuda)
               for v, pr in zip(vocab, per_row)]
        out = hnn.gather_fields(tables, ids, C)
        assert out.shape == (B, C, len(vocab), d)
        want = torch.stack([t.detach()[x] if x.dim() == 2 else t.detach()[x][:, None, :].expand(-1, C, -1)
                            for t, x in zip(tables, ids)], dim=-2)
        assert torch.equal(out.detach(), want)
        w = torch.from_numpy(rng.normal(size=(B, C, len(vocab), d)).astype(np.float32)).to(cuda)
        (out * w).sum().backward()
        for f, (t, x) in enumerate(zip(tables, ids)):
            G = np.zeros(t.shape, dtype=np.float64)
            wf = w[:, :, f, :].cpu().numpy().astype(np.float64)
            xi = x.cpu().numpy()
            if xi.ndim == 1:
                np.add.at(G, xi, wf.sum(axis=1))
            else:
                np.add.at(G, xi.reshape(-1), wf.reshape(-1, d))
            assert_close(t.grad.cpu().numpy(), G, what=f"field {f} grad d={d}", abs_floor=1e-6 * float(np.abs(wf).sum()) / t.shape[0])


def _numeric_values(rng, dtype, shape):
    if dtype == torch.int64:
        return torch.from_numpy(rng.integers(0, 7, size=shape).astype(np.int64))
    return torch.from_numpy((rng.random(size=shape) * 6).astype(np.float64)).to(dtype)


@pytest.mark.parametrize("d,B,C", [(64, 48, 1), (16, 33, 5), (6, 7, 3), (32, 1024, 1), (64, 3000, 1), (32, 700, 4), (128, 9000, 2)])
def test_numeric_fields_ride_in_the_gather_and_get_linear_gradients(d, B, C, cuda, monkeypatch):
    """rc_gather_fields_mixed / rc_numeric_field_grads: a field list with numeric features (nn.Linear(1, d) / nn.Linear(1, 1) on
    the value, models/context/FM.py:38-41,47-48,51-52) in int64 / float32 / float64, per row and per candidate, against torch's
    own Embedding / Linear ops in float64 -- one gather launch, one weighted column sum, every grouping route (small plan,
    sort with the numeric tail, more than one chunk of 1,024 rows)"""
    from rechorus_amd import _lib, engine, nn as hnn
    rng = np.random.default_rng(d * 1000 + B)
    #         kind                vocab  per_row
    spec = [(engine.FIELD_IDS, 11, True), (engine.FIELD_I64, 0, True), (engine.FIELD_IDS, 300, False), (engine.FIELD_F64, 0, False),
            (engine.FIELD_IDS, 5, True), (engine.FIELD_F32, 0, True), (engine.FIELD_IDS, 70, False)]
    F = len(spec)
    mk = lambda shape: torch.from_numpy(rng.normal(0, 0.5, size=shape).astype(np.float32)).to(cuda).requires_grad_(True)
    vec = [mk((v, d)) if k == engine.FIELD_IDS else mk((d, 1)) for k, v, _ in spec]
    lin = [mk((v, 1)) if k == engine.FIELD_IDS else mk((1, 1)) for k, v, _ in spec]
    dt = {engine.FIELD_I64: torch.int64, engine.FIELD_F64: torch.float64, engine.FIELD_F32: torch.float32}
    ids = [(torch.from_numpy(rng.integers(0, v, size=(B,) if pr else (B, C)).astype(np.int64)) if k == engine.FIELD_IDS
            else _numeric_values(rng, dt[k], (B,) if pr else (B, C))).to(cuda) for k, v, pr in spec]
    kinds = [k for k, _, _ in spec]
    assert [engine.FIELD_IDS if k == engine.FIELD_IDS else engine.field_kind(x) for k, x in zip(kinds, ids)] == kinds
    names = []
    real = _lib.call
    monkeypatch.setattr(_lib, "call", lambda fn, *a: names.append(fn) or real(fn, *a))
    V, L = hnn.gather_fields_pair(vec, lin, ids, C, kinds=kinds)
    wv, wl = torch.randn_like(V), torch.randn_like(L)
    ((V * wv).sum() + (L * wl).sum()).backward()
    monkeypatch.undo()
    # the Linear weights' gradients: riding in the small route's row-sums launch, or rc_numeric_field_grads on its own (large batches)
    rides = B * C * F <= 8192 and d in (16, 32, 64, 128)
    # small batches: the grouping of the composite keys runs inside the gather's launch (rc_gather_fields_fused), the backward is ONE launch
    assert names.count("rc_gather_fields_fused" if rides else "rc_gather_fields_mixed") == 1
    assert names.count("rc_small_row_sums_planned") == (1 if rides else 0) and names.count("rc_numeric_field_grads") == (0 if rides else 1), names
    assert not any(n in names for n in ("rc_gather_fields", "rc_gather_fields_pair", "rc_gather_rows", "rc_small_row_sums_pair_numeric",
                                        "rc_small_row_sums_pair", "rc_small_row_sums")), names
    assert V.shape == (B, C, F, d) and L.shape == (B, C, F, 1)
    # what the reference computes, field by field (FM.py:47-55), in float64 from the same fp32 inputs
    bc = lambda t: t if t.dim() == 3 else t.unsqueeze(-2).expand(-1, C, -1)
    v64 = [t.detach().double().requires_grad_(True) for t in vec]
    l64 = [t.detach().double().requires_grad_(True) for t in lin]
    look = lambda W, x, k: W[x] if k == engine.FIELD_IDS else torch.nn.functional.linear(x.float().double().unsqueeze(-1), W)
    V64 = torch.stack([bc(look(W, x, k)) for W, x, k in zip(v64, ids, kinds)], dim=-2)
    L64 = torch.stack([bc(look(W, x, k)) for W, x, k in zip(l64, ids, kinds)], dim=-2)
    ((V64 * wv.double()).sum() + (L64 * wl.double()).sum()).backward()
    assert_close(V.detach().cpu().numpy(), V64.detach().cpu().numpy(), what="field vectors", rtol=2e-7, atol_scale=0)
    assert_close(L.detach().cpu().numpy(), L64.detach().cpu().numpy(), what="first-order values", rtol=2e-7, atol_scale=0)
    for f, (k, _, _) in enumerate(spec):
        n_terms = B * C / (vec[f].shape[0] if k == engine.FIELD_IDS else 1)
        for fam, got, want, w in (("vec", vec[f].grad, v64[f].grad, wv), ("lin", lin[f].grad, l64[f].grad, wl)):
            assert got.shape == want.shape
            # a sum of n signed terms: the floor is the round-off of that sum (numeric fields: B * C terms of |x| <= 6)
            floor = 3e-7 * (6.0 if k != engine.FIELD_IDS else 1.0) * float(w.abs().max()) * n_terms ** 0.5 * 4
            assert_close(got.cpu().numpy(), want.cpu().numpy(), what=f"field {f} ({fam}) kind {k}", rtol=1e-5, atol_scale=1e-5, abs_floor=floor)
    # only the first-order family reaches the loss (WideDeep's wide part alone): the other family's gradients are absent
    for t in vec + lin:
        t.grad = None
    _, L3 = hnn.gather_fields_pair(vec, lin, ids, C, kinds=kinds)
    (L3 * wl).sum().backward()
    for f in range(F):
        assert vec[f].grad is None or not bool(vec[f].grad.any())
        assert_close(lin[f].grad.cpu().numpy(), l64[f].grad.cpu().numpy(), what=f"first-order only, field {f}", rtol=1e-5, atol_scale=1e-5,
                     abs_floor=3e-7 * 6.0 * float(wl.abs().max()) * (B * C) ** 0.5 * 4)


@pytest.mark.parametrize("d,B,C,numeric,tower", [(64, 1024, 1, True, True), (64, 1024, 1, False, True), (16, 37, 5, True, True),
                                                  (32, 700, 2, True, False), (128, 100, 3, False, False), (64, 5000, 1, True, True),
                                                  (24, 50, 2, True, True)])
def test_fm_term_and_plan_inside_the_gather_launch(d, B, C, numeric, tower, cuda, monkeypatch):
    """rc_gather_fields_fused / rc_small_row_sums_planned: the FM pairwise term (models/context/FM.py:61) formed in the gather's
    launch is rc_fm_second_order_fwd's value bit for bit; its backward folded into the row sums (on the grouping the gather's launch
    left behind) gives the gradients of the three-launch route -- rc_fm_second_order_bwd_add, grouping, row sums -- bit for bit,
    and both agree with torch's own ops in float64.  tower: a second consumer of the field vectors (DeepFM) sends a gradient too."""
    from rechorus_amd import _lib, engine, nn as hnn
    rng = np.random.default_rng(d * 7919 + B + C)
    spec = [(engine.FIELD_IDS, 11, True), (engine.FIELD_I64 if numeric else engine.FIELD_IDS, 0 if numeric else 9, True),
            (engine.FIELD_IDS, 300, False), (engine.FIELD_IDS, 5, True), (engine.FIELD_F32 if numeric else engine.FIELD_IDS, 0 if numeric else 40, False),
            (engine.FIELD_IDS, 70, False)]
    F = len(spec)
    mk = lambda shape: torch.from_numpy(rng.normal(0, 0.5, size=shape).astype(np.float32)).to(cuda).requires_grad_(True)
    vec = [mk((v, d)) if k == engine.FIELD_IDS else mk((d, 1)) for k, v, _ in spec]
    lin = [mk((v, 1)) if k == engine.FIELD_IDS else mk((1, 1)) for k, v, _ in spec]
    dt = {engine.FIELD_I64: torch.int64, engine.FIELD_F32: torch.float32}
    ids = [(torch.from_numpy(rng.integers(0, v, size=(B,) if pr else (B, C)).astype(np.int64)) if k == engine.FIELD_IDS
            else _numeric_values(rng, dt[k], (B,) if pr else (B, C))).to(cuda) for k, v, pr in spec]
    kinds = [k for k, _, _ in spec] if numeric else None
    names = []
    real = _lib.call
    monkeypatch.setattr(_lib, "call", lambda fn, *a: names.append(fn) or real(fn, *a))
    V, L, fm = hnn.gather_fields_pair(vec, lin, ids, C, kinds=kinds, fm=True)
    wv, wl, wf = torch.randn_like(V) * (1.0 if tower else 0.0), torch.randn_like(L), torch.randn_like(fm)
    loss = (L * wl).sum() + (fm * wf).sum()
    if tower:
        loss = loss + (V * wv).sum()
    loss.backward()
    monkeypatch.undo()
    small, lanes = B * C * F <= 8192, d in (16, 32, 64, 128)
    if lanes:
        assert names.count("rc_gather_fields_fused") == 1 and "rc_fm_second_order_fwd" not in names, names
    else:
        assert names.count("rc_fm_second_order_fwd") == 1 and "rc_gather_fields_fused" not in names, names
    if small and lanes:     # one launch forward, one launch backward
        assert names.count("rc_small_row_sums_planned") == 1 and "rc_fm_second_order_bwd" not in names and "rc_fm_second_order_bwd_add" not in names, names
        assert not any(n.startswith("rc_small_row_sums") and n != "rc_small_row_sums_planned" and not n.endswith("_bytes") and not n.endswith("_supported") for n in names), names
    else:
        assert names.count("rc_fm_second_order_bwd_add" if tower else "rc_fm_second_order_bwd") == 1, names
    got = {"vec": [t.grad.clone() for t in vec], "lin": [t.grad.clone() for t in lin]}
    # ---- the separate kernels on the same inputs: bit for bit
    assert torch.equal(fm, engine.fm_second_order(V.detach())), "FM term differs from rc_fm_second_order_fwd"
    for t in vec + lin:
        t.grad = None
    V2, L2 = hnn.gather_fields_pair(vec, lin, ids, C, kinds=kinds)
    assert torch.equal(V2, V) and torch.equal(L2, L)
    fm2, flat = hnn.fm_second_order_and_flat(V2)
    loss2 = (L2 * wl).sum() + (fm2 * wf).sum()
    if tower:
        loss2 = loss2 + (flat.view(V2.shape) * wv).sum()
    loss2.backward()
    for f in range(F):
        assert torch.equal(got["vec"][f], vec[f].grad), f"field {f}: vector gradient differs from the separate kernels"
        assert torch.equal(got["lin"][f], lin[f].grad), f"field {f}: first-order gradient differs from the separate kernels"
    if small and lanes:
        # ---- the grouping the gather's launch leaves behind against the plan launch of rc_small_row_sums_pair(_numeric)
        kd = [k for k, _, _ in spec]
        num = [f for f in range(F) if kd[f] != engine.FIELD_IDS]
        with torch.no_grad():
            Vd, Ld, cid, offs, _, _, ws = engine.gather_fields([t.detach() for t in vec], ids, C, tables1=[t.detach() for t in lin], kinds=kd,
                                                              numeric_key=-1, plan=True)
            n, n_rows = cid.numel(), offs[-1]
            gv, gl = torch.randn(n, d, device=cuda), torch.randn(n, 1, device=cuda)
            riding = ([ids[f] for f in num], num, F, C) if num else None
            a = engine.small_row_sums_planned(ws, n, n_rows, gv, gl, d, (F, B, C), numeric=riding)
            b = engine.small_row_sums_pair(cid, n_rows, gv, gl, numeric=riding)
            for x, y in zip(a[:2], b[:2]):
                assert torch.equal(x, y), "row sums on the gather's plan differ from the plan launch's"
            for xs, ys in zip(a[2:], b[2:]):
                assert all(torch.equal(x, y) for x, y in zip(xs, ys))
    # ---- torch's own ops in float64
    bc = lambda t: t if t.dim() == 3 else t.unsqueeze(-2).expand(-1, C, -1)
    v64 = [t.detach().double().requires_grad_(True) for t in vec]
    l64 = [t.detach().double().requires_grad_(True) for t in lin]
    kk = [k for k, _, _ in spec]
    look = lambda W, x, k: W[x] if k == engine.FIELD_IDS else torch.nn.functional.linear(x.float().double().unsqueeze(-1), W)
    V64 = torch.stack([bc(look(W, x, k)) for W, x, k in zip(v64, ids, kk)], dim=-2)
    L64 = torch.stack([bc(look(W, x, k)) for W, x, k in zip(l64, ids, kk)], dim=-2)
    fm64 = 0.5 * (V64.sum(dim=-2).pow(2) - V64.pow(2).sum(dim=-2)).sum(dim=-1)
    ((V64 * wv.double()).sum() + (L64 * wl.double()).sum() + (fm64 * wf.double()).sum()).backward()
    scale = float(V64.abs().max()) ** 2 * F * d
    assert_close(fm.detach().cpu().numpy(), fm64.detach().cpu().numpy(), what="FM term", rtol=1e-5, atol_scale=0, abs_floor=3e-7 * scale)
    for f, (k, _, _) in enumerate(spec):
        n_terms = B * C / (vec[f].shape[0] if k == engine.FIELD_IDS else 1)
        gmax = float(wv.abs().max()) + float(wf.abs().max()) * F * float(V64.abs().max())
        floor = 3e-7 * (6.0 if k != engine.FIELD_IDS else 1.0) * gmax * max(n_terms, 1.0) ** 0.5 * 4
        assert_close(got["vec"][f].cpu().numpy(), v64[f].grad.cpu().numpy(), what=f"field {f} vectors", rtol=1e-5, atol_scale=1e-5, abs_floor=floor)
        assert_close(got["lin"][f].cpu().numpy(), l64[f].grad.cpu().numpy(), what=f"field {f} first-order", rtol=1e-5, atol_scale=1e-5,
                     abs_floor=3e-7 * 6.0 * float(wl.abs().max()) * max(n_terms, 1.0) ** 0.5 * 4)


def test_numeric_field_model_path_uses_no_torch_stack_or_cat(cuda, monkeypatch):
    """with a '*_f' feature among the fields the FM family stays on the one-launch gather, the fused FM term and the one-kernel
    CTR head: no torch.stack / torch.cat / nn.Linear call on the training path (the round-5 route did all three)"""
    from rechorus_amd import _lib
    g = load_golden("deepfm_mind_ctr_d64")
    model, B = _build("deepfm_mind_ctr_d64", g, cuda)
    model.train()
    names = []
    real = _lib.call
    monkeypatch.setattr(_lib, "call", lambda fn, *a: names.append(fn) or real(fn, *a))

    def forbidden(*a, **k):
        raise AssertionError("torch.stack / torch.cat / F.linear on the numeric-field path")
    monkeypatch.setattr(torch, "stack", forbidden)
    monkeypatch.setattr(torch, "cat", forbidden)
    monkeypatch.setattr(torch.nn.functional, "linear", forbidden)
    out = model(_feed(g, 1, B, cuda))
    model.loss(out).backward()
    monkeypatch.undo()
    assert "loss" in out, "the one-kernel CTR head did not run"
    # (B = 48: the small route -- the gather's launch also forms the FM term and groups the keys, ONE row-sums launch backward)
    assert names.count("rc_gather_fields_fused") == 1 and names.count("rc_small_row_sums_planned") == 1, names
    assert not any(n in names for n in ("rc_numeric_field_grads", "rc_fm_second_order_fwd", "rc_fm_second_order_bwd", "rc_fm_second_order_bwd_add",
                                        "rc_small_row_sums_pair_numeric", "rc_gather_fields_mixed")), names
    assert any(n.startswith("rc_ctr_head_fwd_bwd") for n in names), names
    assert_close(out["loss"].item(), g["loss"], what="loss", rtol=2e-5)


def test_bce_ranking_kernel_matches_the_reference(cuda):
    """rc_bce_ranking_fwd_bwd through ContextModel.loss (loss_n='BCE') vs the reference's loss / autograd"""
    from models.BaseContextModel import ContextModel
    g = load_golden("context_bce_ranking")
    for i in range(4):
        p = torch.from_numpy(g["%d/pred" % i]).to(cuda).requires_grad_(True)
        loss = ContextModel.loss(argparse.Namespace(loss_n="BCE"), {"prediction": p})
        loss.backward()
        assert_close(loss.item(), g["%d/loss" % i], what="loss", rtol=2e-5)
        assert_close(p.grad.cpu().numpy(), g["%d/gpred" % i], what="grad", rtol=2e-5, atol_scale=2e-5)


def test_two_table_families_share_keys_and_grouping(cuda, monkeypatch):
    """the [vocab, d] vectors and the [vocab, 1] first-order weights of the FM family are gathered with the same id tensors
    (models/context/FM.py:44-57): ONE gather launch (rc_gather_fields_fused: the grouping of the composite keys runs beside it), and
    in the backward pass ONE row-sums launch for both dense gradients (rc_small_row_sums_planned) -- values and gradients
    bit-identical to two separate gathers"""
    from rechorus_amd import _lib, nn as hnn
    rng = np.random.default_rng(4)
    for vocab, B, C in (([50, 7, 300], 96, 1), ([11, 300, 5, 70, 2], 33, 5), ([40000, 9], 1024, 1)):
        mk = lambda d: [torch.from_numpy(rng.normal(size=(v, d)).astype(np.float32)).to(cuda).requires_grad_(True) for v in vocab]
        vec, lin = mk(64), mk(1)
        ids = [torch.from_numpy(rng.integers(0, v, size=(B,) if k % 2 == 0 else (B, C)).astype(np.int64)).to(cuda) for k, v in enumerate(vocab)]
        names = []
        real = _lib.call
        monkeypatch.setattr(_lib, "call", lambda fn, *a: names.append(fn) or real(fn, *a))
        V, L = hnn.gather_fields_pair(vec, lin, ids, C)
        wv, wl = torch.randn_like(V), torch.randn_like(L)
        ((V * wv).sum() + (L * wl).sum()).backward()
        monkeypatch.undo()
        assert names.count("rc_gather_fields_fused") == 1 and "rc_gather_fields" not in names and "rc_gather_fields_pair" not in names
        assert names.count("rc_small_row_sums_planned") == 1 and "rc_small_row_sums" not in names and "rc_small_row_sums_pair" not in names, names
        got = [t.grad.clone() for t in vec + lin]
        for t in vec + lin:
            t.grad = None
        V2 = hnn.gather_fields(vec, [x.clone() for x in ids], C)      # the one-family node
        L2 = hnn.gather_fields(lin, [x.clone() for x in ids], C)
        assert torch.equal(V2, V) and torch.equal(L2, L)
        ((V2 * wv).sum() + (L2 * wl).sum()).backward()
        for a, t in zip(got[:len(vec)], vec):
            assert torch.equal(a, t.grad)
        for a, t in zip(got[len(vec):], lin):      # (the one-float-wide sums run in another fixed order inside the pair launch)
            assert_close(a.cpu().numpy(), t.grad.cpu().numpy(), what="first-order gradient", rtol=1e-6, atol_scale=1e-6)
        # only one of the two outputs used: the other family's gradient is absent, not garbage
        for t in vec + lin:
            t.grad = None
        V3, _ = hnn.gather_fields_pair(vec, lin, ids, C)
        (V3 * wv).sum().backward()
        assert all(torch.equal(a, t.grad) for a, t in zip(got[:len(vec)], vec))
        assert all(t.grad is None or not bool(t.grad.any()) for t in lin)


def test_ctr_head_one_workgroup_equals_the_two_kernel_head(cuda, monkeypatch):
    """rc_ctr_head_fwd_bwd_sums / rc_ctr_head_bwd (probabilities, loss mean, sum gz and the backward fan-out in two launches) against
    rc_ctr_head_fwd_bwd + reduce + autograd's mul / sum / expand: same probabilities and gradients, loss to rounding"""
    from rechorus_amd import engine, nn as hnn
    rng = np.random.default_rng(8)
    for n, F, n_terms in ((1024, 8, 2), (37, 3, 1), (5000, 1, 0)):
        res = []
        for one_wg in (True, False):
            monkeypatch.setattr(engine, "CTR_HEAD_ONE_WG_MAX", 65536 if one_wg else 0)
            bias = torch.tensor([0.05], device=cuda, requires_grad=True)
            lin = torch.from_numpy(rng.normal(0, 0.3, (n, 1, F, 1)).astype(np.float32)).to(cuda).requires_grad_(True)
            terms = [torch.from_numpy(rng.normal(0, 1.0, (n, 1)).astype(np.float32)).to(cuda).requires_grad_(True) for _ in range(n_terms)]
            label = torch.from_numpy(rng.integers(0, 2, n).astype(np.int64)).to(cuda)
            rng = np.random.default_rng(8)      # (same inputs for the second variant)
            p, loss = hnn.ctr_head(bias, lin, label, terms)
            (loss * 1.7).backward()
            res.append((p.detach(), loss.detach(), bias.grad, lin.grad, [t.grad for t in terms]))
            rng = np.random.default_rng(8)
        (pa, la, ba, ga, ta), (pb, lb, bb, gb, tb) = res
        assert torch.equal(pa, pb)
        assert_close(la.cpu().numpy(), lb.cpu().numpy(), what="loss", rtol=1e-6)
        assert_close(ba.cpu().numpy(), bb.cpu().numpy(), what="bias grad", rtol=1e-5, atol_scale=1e-6)
        assert torch.equal(ga, gb) and all(torch.equal(x, y) for x, y in zip(ta, tb))
