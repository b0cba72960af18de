"""GPU: north star "existing model files drop in" for the heads of the hot path (BPRMF / NeuMF / SASRec, and the FM / WideDeep /
DeepFM context family at the end of this file).  The reference's OWN BPRMF.py / NeuMF.py /
SASRec.py (verbatim copies, tests/golden/reference_models/, checked against their manifest by tests/test_dropin_heads_cpu.py)
are given to the plugin's main.py through RECHORUS_MODEL_DIRS.  rechorus_amd/dropin.py recognises the head, and the run
(i) trains through the fused one-call fit() iteration (rc_bprmf_train_step_ahead / rc_neumf_train_step* / engine.SasrecTrainer),
(ii) leaves a checkpoint that is BIT-IDENTICAL to the one the plugin's own class of the same name leaves from the same seed, with
the reference's state_dict keys, and (iii) an edited model file whose forward no longer computes the head keeps its own route.

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
FIX = os.path.join(ROOT, "tests", "golden", "reference_models")
if PLUGIN not in sys.path:
    sys.path.insert(0, PLUGIN)


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
    # the reference's own NeuMF command line (docs/demo_scripts_results/Topk_Amazon.sh:8): --dropout 0.2 -> the mask inside the fused kernel
    ("general", "NeuMF", ["--emb_size", "64", "--layers", "[64]", "--lr", "5e-4", "--l2", "1e-7", "--dropout", "0.2"], "NeumfTrainer",
     r"rc_neumf_train_step_dropout"),
]


def _run(model_args, name, dataset_root, out, monkeypatch, model_dir):
    import main
    from rechorus_amd import _lib, engine, nn as hnn
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
def test_unmodified_reference_model_file_trains_through_the_fused_step(sub, name, model_args, trainer, entry, dataset_root, tmp_path,
                                                                        monkeypatch, cuda):
    (tmp_path / "ref").mkdir(), (tmp_path / "mirror").mkdir()
    res_a, text_a, sd_a, names_a, steps_a = _run(model_args, name, dataset_root, tmp_path / "ref", monkeypatch, os.path.join(FIX, sub))
    assert "Recognised the %s head" % name in text_a, text_a[-1500:]
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
    # the model-file class is still the reference's (found first in RECHORUS_MODEL_DIRS), only its head moved
    assert re.search(r"HR@5:([0-9.]+)", res_a["test"])


def test_an_edited_head_keeps_its_own_route(dataset_root, tmp_path, monkeypatch, cuda):
    """a BPRMF.py whose forward was changed (scores doubled): state_dict and attributes still look like the head, the probe batch
    says otherwise -> no binding, the model trains on its own forward over the adopted tables"""
    import main
    from rechorus_amd import dropin, nn as hnn
    src = open(os.path.join(FIX, "general", "BPRMF.py")).read()
    d = tmp_path / "edited"
    d.mkdir()
    (d / "BPRMF.py").write_text(src.replace("cf_u_vectors = self.u_embeddings(u_ids)", "cf_u_vectors = 2 * self.u_embeddings(u_ids)"))
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
    ("DeepFM", "CTR", ["--emb_size", "16", "--layers", "[32,16]", "--dropout", "0.2"], {"rc_gather_fields_pair", "rc_fm_second_order_bwd_add"},
     r"rc_ctr_head_fwd_bwd(_sums)?"),
    ("DeepFM", "TopK", ["--emb_size", "16", "--layers", "[32]", "--dropout", "0"], {"rc_gather_fields_pair", "rc_fm_second_order_bwd_add"}, None),
    ("FM", "CTR", ["--emb_size", "16"], {"rc_gather_fields_pair", "rc_fm_second_order_fwd"}, r"rc_ctr_head_fwd_bwd(_sums)?"),
    ("WideDeep", "CTR", ["--emb_size", "16", "--layers", "[32]", "--dropout", "0.1"], {"rc_gather_fields_pair"}, r"rc_ctr_head_fwd_bwd(_sums)?"),
    # the same heads over a field list with numeric features (models/context/FM.py:38-41: Linear(1, d) on c_day_f / i_age_f)
    ("DeepFM", "CTR:f", ["--emb_size", "16", "--layers", "[32,16]", "--dropout", "0.2"],
     {"rc_gather_fields_mixed", "rc_small_row_sums_pair_numeric", "rc_fm_second_order_bwd_add"}, r"rc_ctr_head_fwd_bwd(_sums)?"),
    ("FM", "TopK:f", ["--emb_size", "16"], {"rc_gather_fields_mixed", "rc_small_row_sums_pair_numeric", "rc_fm_second_order_fwd"}, None),
]


def _run_ctx(name, mode, model_args, ctx_root, out, monkeypatch, model_dir):
    import main
    from rechorus_amd import _lib, nn as hnn
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
    mode, _, numeric = mode.partition(":")
    ctr = mode == "CTR"
    res = main.run(["--model_name", name, "--model_mode", mode] + model_args +
                   ["--dataset", ("ctr" if ctr else "topk") + numeric, "--path", ctx_root + "/", "--epoch", "3", "--num_neg", "3", "--batch_size", "64",
                    "--num_workers", "0", "--regenerate", "1", "--random_seed", "11", "--log_file", log, "--lr", "2e-3", "--l2", "1e-6",
                    "--loss_n", "BCE" if ctr else "BPR", "--metric", "AUC,ACC" if ctr else "NDCG,HR", "--include_item_features", "1",
                    "--include_user_features", "1", "--include_situation_features", "1",
                    "--model_path", str(out / "model" / "m.pt"), "--topk", "5,10", "--save_final_results", "0"])
    monkeypatch.undo()
    return res, open(log).read(), torch.load(str(out / "model" / "m.pt"), map_location="cpu"), names


@pytest.mark.parametrize("name,mode,model_args,entries,head_entry", CTX_CASES)
def test_unmodified_reference_context_model_file_reaches_the_fused_head(name, mode, model_args, entries, head_entry, ctx_root, tmp_path,
                                                                         monkeypatch, cuda):
    """the reference's OWN FM.py / WideDeep.py / DeepFM.py through main.py: the head is recognised, training runs on the one-launch
    field gathers, the fused FM term and (CTR, --loss_n BCE) the one-kernel CTR head, and the checkpoint equals the one the plugin's
    class of the same name leaves from the same seed, bit for bit, under the reference's state_dict keys"""
    (tmp_path / "ref").mkdir(), (tmp_path / "mirror").mkdir()
    res_a, text_a, sd_a, names_a = _run_ctx(name, mode, model_args, ctx_root, tmp_path / "ref", monkeypatch, os.path.join(FIX, "context"))
    assert "Recognised the %s%s head" % (name, mode.partition(":")[0]) in text_a, text_a[-1500:]
    assert entries <= names_a, sorted(names_a)
    if head_entry:
        assert any(re.fullmatch(head_entry, n) for n in names_a), sorted(names_a)
    losses = [float(x) for x in re.findall(r"Epoch \d+\s+loss=([0-9.]+)", text_a)]
    assert len(losses) >= 2 and losses[-1] < losses[0], losses
    res_b, text_b, sd_b, names_b = _run_ctx(name, mode, model_args, ctx_root, tmp_path / "mirror", monkeypatch, None)
    assert "Recognised the" not in text_b
    assert set(sd_a) == set(sd_b)
    for k in sd_a:
        assert torch.equal(sd_a[k], sd_b[k]), k
    assert res_a == res_b


def test_an_edited_context_head_keeps_its_own_route(tmp_path, monkeypatch, cuda):
    """a DeepFM.py whose forward drops the FM term still has DeepFM's parameters and class name; the probe batch disagrees with the
    fused head -> no binding"""
    import main
    from rechorus_amd import dropin
    src = open(os.path.join(FIX, "context", "DeepFM.py")).read()
    d = tmp_path / "edited"
    d.mkdir()
    edited = src.replace("fm_prediction = fm_vectors.sum(dim=-1) + linear_vectors", "fm_prediction = linear_vectors")
    assert edited != src
    (d / "DeepFM.py").write_text(edited)
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
        assert dropin._context_kind(model) == "DeepFMCTR"
        built.append((cls, model))
    (cls_e, edited_model), (cls_r, ref_model) = built
    assert dropin.bind_known_head(edited_model) is None and type(edited_model) is cls_e
    assert dropin.bind_known_head(ref_model) == "DeepFMCTR"
    assert isinstance(ref_model, cls_r) and type(ref_model).__name__ == "DeepFMCTR" and type(ref_model)._rc_bound_head == "DeepFMCTR"
    assert dropin.bind_known_head(ref_model) is None     # idempotent
