"""GPU: north star "existing model files drop in" for the heads of the hot path (BPRMF / NeuMF / SASRec, and the FM / WideDeep /
DeepFM context family at the end of this file).  Model files written the way a user of the reference writes them -- plain torch
layers over the plugin's task bases, tests/user_models/ (this repository's own code; the reference's files never leave
/root/reference, and tests/test_dropin_heads_cpu.py checks dropin.py's table of THEIR syntax trees there) -- are given to the
plugin's main.py through RECHORUS_MODEL_DIRS.  rechorus_amd/dropin.py recognises the head by structure + a probe batch, and the run
(i) trains through the fused one-call fit() iteration (rc_bprmf_train_step_ahead / rc_neumf_train_step* / engine.SasrecTrainer),
(ii) leaves a checkpoint that is BIT-IDENTICAL to the one the plugin's own class of the same name leaves from the same seed, with
the reference's state_dict keys, and (iii) an edited model file whose forward no longer computes the head keeps its own route.
A file whose syntax tree is not in dropin's table is bound only with dropout 0 (a probe batch cannot verify a stochastic forward);
the dropout cases below register the user file's tree the way the reference's are listed.

Reference: src/models/general/BPRMF.py:34-63, NeuMF.py:56-76, src/models/sequential/SASRec.py:51-86, src/main.py:164-166."""
import argparse
import os
import re
import sys

import pytest
import torch

from conftest import ROOT
from synth_data import make_dataset

pytestmark = pytest.mark.gpu

PLUGIN = os.path.join(ROOT, "rechorus_amd", "rechorus")
FIX = os.path.join(ROOT, "tests", "user_models")
if PLUGIN not in sys.path:
    sys.path.insert(0, PLUGIN)


def _list_as_known(monkeypatch, sub, name, mode=""):
    """put the user file's forward syntax tree into dropin's table (what tests/golden/make_reference_forward_hashes.py does for the
    reference's files): the file then binds with dropout > 0 too"""
    import main
    from rechorus_amd import dropin
    monkeypatch.setenv("RECHORUS_MODEL_DIRS", os.path.join(FIX, sub))
    h = dropin.forward_hash(main.find_class("model", (name, mode)))
    table = {k: set(v) for k, v in dropin.KNOWN_FORWARD_HASHES.items()}
    table[name + mode].add(h)
    monkeypatch.setattr(dropin, "KNOWN_FORWARD_HASHES", table)


@pytest.fixture(scope="module")
def dataset_root(tmp_path_factory):
    root = str(tmp_path_factory.mktemp("data"))
    make_dataset(root, "synth", n_users=400, n_items=300, per_user=14, seed=1)
    return root


CASES = [
    ("general", "BPRMF", ["--emb_size", "32", "--optimizer", "SGD", "--lr", "40", "--l2", "0"], "BprmfTrainer", r"rc_bprmf_train_step_ahead"),
    ("general", "NeuMF", ["--emb_size", "32", "--layers", "[32]", "--lr", "5e-3", "--l2", "1e-6", "--dropout", "0"], "NeumfTrainer",
     r"rc_neumf_train_step(_marked)?"),
    ("sequential", "SASRec", ["--emb_size", "32", "--num_layers", "1", "--num_heads", "2", "--history_max", "10", "--lr", "3e-3",
                              "--l2", "1e-6", "--dropout", "0"], "SasrecTrainer", r"rc_sasrec\w*"),
    # the reference's own NeuMF command line (docs/demo_scripts_results/Topk_Amazon.sh:8): --dropout 0.2 -> the mask inside the fused
    # kernel (the file's syntax tree is listed first: _list_as_known)
    ("general", "NeuMF", ["--emb_size", "64", "--layers", "[64]", "--lr", "5e-4", "--l2", "1e-7", "--dropout", "0.2"], "NeumfTrainer",
     r"rc_neumf_train_step_dropout"),
]


def _run(model_args, name, dataset_root, out, monkeypatch, model_dir, known=None):
    import main
    from rechorus_amd import _lib, engine, nn as hnn
    if known:
        _list_as_known(monkeypatch, *known)
    monkeypatch.setattr(hnn, "_DROP_SEED_GEN", None)    # both runs draw the same first dropout seed (the generator follows --random_seed)
    if model_dir:
        monkeypatch.setenv("RECHORUS_MODEL_DIRS", model_dir)
    else:
        monkeypatch.delenv("RECHORUS_MODEL_DIRS", raising=False)
    names, steps = set(), {}
    real_call = _lib.call

    def call(fn_name, *a):
        names.add(fn_name)
        return real_call(fn_name, *a)
    monkeypatch.setattr(_lib, "call", call)
    for tr in ("BprmfTrainer", "NeumfTrainer", "SasrecTrainer"):
        cls = getattr(engine, tr)
        real = cls.step

        def step(self, *a, _real=real, _tr=tr, **kw):
            steps[_tr] = steps.get(_tr, 0) + 1
            return _real(self, *a, **kw)
        monkeypatch.setattr(cls, "step", step)
    log = str(out / "log" / "run.txt")
    res = main.run(["--model_name", name] + model_args +
                   ["--dataset", "synth", "--path", dataset_root + "/", "--epoch", "3", "--num_neg", "4", "--batch_size", "128",
                    "--num_workers", "0", "--engine", "rowwise", "--regenerate", "1", "--random_seed", "7", "--log_file", log,
                    "--model_path", str(out / "model" / "m.pt"), "--topk", "5,10", "--save_final_results", "0"])
    monkeypatch.undo()
    return res, open(log).read(), torch.load(str(out / "model" / "m.pt"), map_location="cpu"), names, steps


@pytest.mark.parametrize("sub,name,model_args,trainer,entry", CASES)
def test_a_users_model_file_trains_through_the_fused_step(sub, name, model_args, trainer, entry, dataset_root, tmp_path, monkeypatch, cuda):
    (tmp_path / "ref").mkdir(), (tmp_path / "mirror").mkdir()
    stochastic = float(model_args[model_args.index("--dropout") + 1]) > 0 if "--dropout" in model_args else False
    res_a, text_a, sd_a, names_a, steps_a = _run(model_args, name, dataset_root, tmp_path / "ref", monkeypatch, os.path.join(FIX, sub),
                                                 known=(sub, name) if stochastic else None)
    assert "Recognised the %s head" % name in text_a, text_a[-1500:]
    assert ("(edited forward, verified on a probe batch)" in text_a) == (not stochastic)
    assert "Adopted" in text_a
    assert steps_a.get(trainer, 0) > 3, steps_a
    assert any(re.fullmatch(entry, n) for n in names_a), sorted(names_a)
    losses = [float(x) for x in re.findall(r"Epoch \d+\s+loss=([0-9.]+)", text_a)]
    assert len(losses) >= 2 and losses[-1] < losses[0], losses
    res_b, text_b, sd_b, names_b, steps_b = _run(model_args, name, dataset_root, tmp_path / "mirror", monkeypatch, None)
    assert "Recognised the" not in text_b and steps_b.get(trainer, 0) == steps_a[trainer]
    # the reference's own state_dict keys, and the plugin's class of the same name ends on the same bits
    assert set(sd_a) == set(sd_b) and not any("drop_seed" in k for k in sd_a)
    for k in sd_a:
        assert torch.equal(sd_a[k], sd_b[k]), k
    assert res_a == res_b
    # the model-file class is still the user's (found first in RECHORUS_MODEL_DIRS), only its head moved
    assert re.search(r"HR@5:([0-9.]+)", res_a["test"])


def test_an_edited_head_keeps_its_own_route(dataset_root, tmp_path, monkeypatch, cuda):
    """a BPRMF.py whose forward was changed (scores doubled): state_dict and attributes still look like the head, the probe batch
    says otherwise -> no binding, the model trains on its own forward over the adopted tables"""
    import main
    from rechorus_amd import dropin, nn as hnn
    src = open(os.path.join(FIX, "general", "BPRMF.py")).read()
    d = tmp_path / "edited"
    d.mkdir()
    line = "users = self.u_embeddings(feed_dict['user_id'])"
    assert line in src
    (d / "BPRMF.py").write_text(src.replace(line, "users = 2 * self.u_embeddings(feed_dict['user_id'])"))
    monkeypatch.setenv("RECHORUS_MODEL_DIRS", str(d))
    cls = main.find_class("model", ("BPRMF", ""))
    args = cls.parse_model_args(argparse.ArgumentParser()).parse_args(["--emb_size", "32"])
    args.device, args.model_path, args.buffer = cuda, "", 1
    torch.manual_seed(0)
    model = cls(args, argparse.Namespace(n_users=20, n_items=50)).to(cuda)
    assert hnn.adopt_embeddings(model) == 2
    assert dropin._kind(model) == "BPRMF"
    assert dropin.bind_known_head(model) is None
    assert "does not reproduce its forward on a probe batch" in dropin.last_miss_reason      # (main.py's log carries the reason)
    assert type(model) is cls and not hasattr(model, "hip_train_step") and "drop_seed" not in dict(model.named_buffers())
    # ... and the unedited file binds, with the model file's class still underneath
    monkeypatch.setenv("RECHORUS_MODEL_DIRS", os.path.join(FIX, "general"))
    cls2 = main.find_class("model", ("BPRMF", ""))
    torch.manual_seed(0)
    m2 = cls2(args, argparse.Namespace(n_users=20, n_items=50)).to(cuda)
    hnn.adopt_embeddings(m2)
    assert dropin.bind_known_head(m2) == "BPRMF" and dropin.last_miss_reason is None
    assert isinstance(m2, cls2) and type(m2).__name__ == "BPRMF" and hasattr(m2, "hip_train_step") and m2.candidate_permutation_equivariant
    assert dropin.bind_known_head(m2) is None       # idempotent: already bound


# ---- the context family: models/context/FM.py, WideDeep.py, DeepFM.py ---------------------------------------------------------

@pytest.fixture(scope="module")
def ctx_root(tmp_path_factory):
    from synth_data import make_context_dataset
    root = str(tmp_path_factory.mktemp("ctx"))
    make_context_dataset(root, "ctr", n_users=90, n_items=70, per_user=12, ctr=True, seed=3)
    make_context_dataset(root, "topk", n_users=90, n_items=70, per_user=10, ctr=False, seed=4)
    make_context_dataset(root, "ctrf", n_users=90, n_items=70, per_user=12, ctr=True, seed=5, numeric=True)     # + c_day_f (int64), i_age_f (float64)
    make_context_dataset(root, "topkf", n_users=90, n_items=70, per_user=10, ctr=False, seed=6, numeric=True)
    return root


CTX_CASES = [
    # (name, mode[:f = the dataset with numeric features], flags, entry points that must run, entry points that must NOT, CTR head)
    ("DeepFM", "CTR", ["--emb_size", "16", "--layers", "[32,16]", "--dropout", "0.2"], {"rc_gather_fields_fused", "rc_small_row_sums_planned"},
     {"rc_fm_second_order_fwd", "rc_fm_second_order_bwd", "rc_fm_second_order_bwd_add", "rc_gather_fields_pair"}, r"rc_ctr_head_fwd_bwd(_sums)?"),
    ("DeepFM", "TopK", ["--emb_size", "16", "--layers", "[32]", "--dropout", "0"], {"rc_gather_fields_fused", "rc_small_row_sums_planned"},
     {"rc_fm_second_order_fwd", "rc_fm_second_order_bwd_add"}, None),
    ("FM", "CTR", ["--emb_size", "16"], {"rc_gather_fields_fused", "rc_small_row_sums_planned"}, {"rc_fm_second_order_fwd", "rc_fm_second_order_bwd"},
     r"rc_ctr_head_fwd_bwd(_sums)?"),
    ("WideDeep", "CTR", ["--emb_size", "16", "--layers", "[32]", "--dropout", "0.1"], {"rc_gather_fields_fused", "rc_small_row_sums_planned"}, set(),
     r"rc_ctr_head_fwd_bwd(_sums)?"),
    # the same heads over a field list with numeric features (Linear(1, d) on c_day_f / i_age_f, as models/context/FM.py:38-41 does)
    ("DeepFM", "CTR:f", ["--emb_size", "16", "--layers", "[32,16]", "--dropout", "0.2"], {"rc_gather_fields_fused", "rc_small_row_sums_planned"},
     {"rc_numeric_field_grads", "rc_fm_second_order_bwd_add", "rc_gather_fields_mixed"}, r"rc_ctr_head_fwd_bwd(_sums)?"),
    ("FM", "TopK:f", ["--emb_size", "16"], {"rc_gather_fields_fused", "rc_small_row_sums_planned"}, {"rc_numeric_field_grads", "rc_fm_second_order_fwd"}, None),
]


def _run_ctx(name, mode, model_args, ctx_root, out, monkeypatch, model_dir, known=False):
    import main
    from rechorus_amd import _lib, nn as hnn
    mode, _, numeric = mode.partition(":")
    if known:
        _list_as_known(monkeypatch, "context", name, mode)
    monkeypatch.setattr(hnn, "_DROP_SEED_GEN", None)
    if model_dir:
        monkeypatch.setenv("RECHORUS_MODEL_DIRS", model_dir)
    else:
        monkeypatch.delenv("RECHORUS_MODEL_DIRS", raising=False)
    names = set()
    real_call = _lib.call

    def call(fn_name, *a):
        names.add(fn_name)
        return real_call(fn_name, *a)
    monkeypatch.setattr(_lib, "call", call)
    log = str(out / "log" / "run.txt")
    ctr = mode == "CTR"
    res = main.run(["--model_name", name, "--model_mode", mode] + model_args +
                   ["--dataset", ("ctr" if ctr else "topk") + numeric, "--path", ctx_root + "/", "--epoch", "3", "--num_neg", "3", "--batch_size", "64",
                    "--num_workers", "0", "--regenerate", "1", "--random_seed", "11", "--log_file", log, "--lr", "2e-3", "--l2", "1e-6",
                    "--loss_n", "BCE" if ctr else "BPR", "--metric", "AUC,ACC" if ctr else "NDCG,HR", "--include_item_features", "1",
                    "--include_user_features", "1", "--include_situation_features", "1",
                    "--model_path", str(out / "model" / "m.pt"), "--topk", "5,10", "--save_final_results", "0"])
    monkeypatch.undo()
    return res, open(log).read(), torch.load(str(out / "model" / "m.pt"), map_location="cpu"), names


@pytest.mark.parametrize("name,mode,model_args,entries,absent,head_entry", CTX_CASES)
def test_a_users_context_model_file_reaches_the_fused_head(name, mode, model_args, entries, absent, head_entry, ctx_root, tmp_path, monkeypatch, cuda):
    """a user's FM.py / WideDeep.py / DeepFM.py through main.py: the head is recognised, training runs on the one-launch field gather
    (FM term and key grouping inside it), ONE row-sums launch backward and (CTR, --loss_n BCE) the one-kernel CTR head; the run
    learns, and ends where the plugin's class of the same name ends (same metric keys, losses within a few percent: the two
    constructors draw their initial weights in different orders, so the runs are not the same run -- the bit-for-bit statement is
    test_a_bound_context_model_file_steps_like_the_plugins_class below)"""
    (tmp_path / "user").mkdir(), (tmp_path / "mirror").mkdir()
    stochastic = float(model_args[model_args.index("--dropout") + 1]) > 0 if "--dropout" in model_args else False
    res_a, text_a, sd_a, names_a = _run_ctx(name, mode, model_args, ctx_root, tmp_path / "user", monkeypatch, os.path.join(FIX, "context"),
                                            known=stochastic)
    assert "Recognised the %s%s head" % (name, mode.partition(":")[0]) in text_a, text_a[-1500:]
    assert entries <= names_a and not (absent & names_a), sorted(names_a)
    if head_entry:
        assert any(re.fullmatch(head_entry, n) for n in names_a), sorted(names_a)
    losses = [float(x) for x in re.findall(r"Epoch \d+\s+loss=([0-9.]+)", text_a)]
    assert len(losses) >= 2 and losses[-1] < losses[0], losses
    res_b, text_b, sd_b, names_b = _run_ctx(name, mode, model_args, ctx_root, tmp_path / "mirror", monkeypatch, None)
    assert "Recognised the" not in text_b
    assert set(sd_a) == set(sd_b) and all(sd_a[k].shape == sd_b[k].shape for k in sd_a)
    assert entries <= names_b and not (absent & names_b), sorted(names_b)        # the same entry points either way
    losses_b = [float(x) for x in re.findall(r"Epoch \d+\s+loss=([0-9.]+)", text_b)]
    assert len(losses_b) == len(losses) and abs(losses_b[-1] - losses[-1]) < 0.08 * losses[-1], (losses, losses_b)


def _ctx_model(name, mode, flags, model_dir, monkeypatch, cuda, numeric):
    import main
    if model_dir:
        monkeypatch.setenv("RECHORUS_MODEL_DIRS", model_dir)
    else:
        monkeypatch.delenv("RECHORUS_MODEL_DIRS", raising=False)
    cls = main.find_class("model", (name, mode))
    args = cls.parse_model_args(argparse.ArgumentParser()).parse_args(flags + ["--loss_n", "BCE" if mode == "CTR" else "BPR"])
    args.device, args.model_path, args.buffer = cuda, "", 1
    args.include_item_features = args.include_user_features = args.include_situation_features = 1
    fmax = {"user_id": 40, "item_id": 90, "i_cat_c": 7, "u_grp_c": 4, "c_hour_c": 24}
    corpus = argparse.Namespace(n_users=40, n_items=90, feature_max=fmax, item_feature_names=["i_cat_c"] + (["i_price_f"] if numeric else []),
                                user_feature_names=["u_grp_c"], situation_feature_names=["c_hour_c"] + (["c_day_f"] if numeric else []))
    return cls, cls(args, corpus).to(cuda)


@pytest.mark.parametrize("name,mode,flags,numeric", [("DeepFM", "CTR", ["--emb_size", "16", "--layers", "[32,16]", "--dropout", "0.2"], True),
                                                     ("FM", "TopK", ["--emb_size", "32"], True),
                                                     ("WideDeep", "CTR", ["--emb_size", "16", "--layers", "[32]", "--dropout", "0"], False),
                                                     ("DeepFM", "TopK", ["--emb_size", "64", "--layers", "[64]", "--dropout", "0"], False)])
def test_a_bound_context_model_file_steps_like_the_plugins_class(name, mode, flags, numeric, monkeypatch, cuda):
    """a user's model file, bound, and the plugin's class of the same name started from the SAME weights take the same training
    steps (same batches, HipOptimizer Adam): parameters bit-identical after every step, under the same state_dict keys"""
    from rechorus_amd import dropin, nn as hnn
    stochastic = float(flags[flags.index("--dropout") + 1]) > 0 if "--dropout" in flags else False
    if stochastic:
        _list_as_known(monkeypatch, "context", name, mode)
    monkeypatch.setattr(hnn, "_DROP_SEED_GEN", None)
    torch.manual_seed(3)
    cls_u, user = _ctx_model(name, mode, flags, os.path.join(FIX, "context"), monkeypatch, cuda, numeric)
    assert hnn.adopt_embeddings(user) > 0
    assert dropin.bind_known_head(user) == name + mode, dropin.last_miss_reason
    monkeypatch.setattr(hnn, "_DROP_SEED_GEN", None)      # (the plugin's tower draws the same first dropout seed)
    cls_p, mirror = _ctx_model(name, mode, flags, None, monkeypatch, cuda, numeric)
    assert cls_p.__module__.startswith("models.") and isinstance(user, cls_u) and not isinstance(user, cls_p)
    assert set(user.state_dict()) == set(mirror.state_dict())
    mirror.load_state_dict(user.state_dict())
    for k, b in mirror.named_buffers():        # dropout seeds are no state_dict entries: same stream for both
        dict(user.named_buffers())[k].copy_(b)
    g = torch.Generator().manual_seed(5)
    B, C = 48, (1 if mode == "CTR" else 4)
    opts = [hnn.HipOptimizer(m.customize_parameters(), "Adam", 2e-3, 1e-6) for m in (user, mirror)]
    for step in range(4):
        feed = {"user_id": torch.randint(0, 40, (B,), generator=g), "item_id": torch.randint(0, 90, (B, C), generator=g),
                "i_cat_c": torch.randint(0, 7, (B, C), generator=g), "u_grp_c": torch.randint(0, 4, (B,), generator=g),
                "c_hour_c": torch.randint(0, 24, (B,), generator=g)}
        if numeric:
            feed["i_price_f"] = torch.rand((B, C), generator=g) * 3
            feed["c_day_f"] = torch.randint(0, 7, (B,), generator=g)
        if mode == "CTR":
            feed["label"] = torch.randint(0, 2, (B, 1), generator=g)
        feed = {k: v.to(cuda) for k, v in feed.items()}
        feed.update(batch_size=B, phase="train")
        losses = []
        for m, opt in zip((user, mirror), opts):
            m.train()
            out = m(dict(feed))
            loss = m.loss(out)
            for p in m.parameters():
                p.grad = None
            loss.backward()
            opt.step()
            losses.append(float(loss))
        assert losses[0] == losses[1], (step, losses)
        sd_u, sd_p = user.state_dict(), mirror.state_dict()
        for k in sd_u:
            assert torch.equal(sd_u[k], sd_p[k]), (step, k)


def test_an_edited_context_head_keeps_its_own_route(tmp_path, monkeypatch, cuda):
    """a DeepFM.py whose forward drops the FM term still has DeepFM's parameters and class name; the probe batch disagrees with the
    fused head -> no binding"""
    import main
    from rechorus_amd import dropin, nn as hnn
    src = open(os.path.join(FIX, "context", "DeepFM.py")).read()
    d = tmp_path / "edited"
    d.mkdir()
    edited = src.replace("return first + pairwise + self.deep_layers(", "return first + self.deep_layers(")
    assert edited != src
    (d / "DeepFM.py").write_text(edited.replace("os.path.dirname(os.path.abspath(__file__))", repr(os.path.join(FIX, "context"))))
    fmax = {"user_id": 20, "item_id": 50, "i_cat_c": 7, "u_grp_c": 4, "c_hour_c": 24}
    corpus = argparse.Namespace(n_users=20, n_items=50, feature_max=fmax, item_feature_names=["i_cat_c"], user_feature_names=["u_grp_c"],
                                situation_feature_names=["c_hour_c"])
    built = []
    for model_dir in (str(d), os.path.join(FIX, "context")):
        monkeypatch.setenv("RECHORUS_MODEL_DIRS", model_dir)
        cls = main.find_class("model", ("DeepFM", "CTR"))
        args = cls.parse_model_args(argparse.ArgumentParser()).parse_args(["--emb_size", "16", "--layers", "[32]", "--dropout", "0", "--loss_n", "BCE"])
        args.device, args.model_path, args.buffer = cuda, "", 1
        args.include_item_features = args.include_user_features = args.include_situation_features = 1
        torch.manual_seed(0)
        model = cls(args, corpus).to(cuda)
        hnn.adopt_embeddings(model)
        assert dropin._context_kind(model) == "DeepFMCTR"
        built.append((cls, model))
    (cls_e, edited_model), (cls_r, user_model) = built
    assert dropin.bind_known_head(edited_model) is None and type(edited_model) is cls_e
    assert "does not reproduce its forward on a probe batch" in dropin.last_miss_reason
    assert dropin.bind_known_head(user_model) == "DeepFMCTR"
    assert isinstance(user_model, cls_r) and type(user_model).__name__ == "DeepFMCTR" and type(user_model)._rc_bound_head == "DeepFMCTR"
    assert dropin.bind_known_head(user_model) is None     # idempotent
