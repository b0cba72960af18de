"""CPU: the known-head recognition of rechorus_amd/dropin.py on the reference's own model files (verbatim copies under
tests/golden/reference_models/, tests/golden/make_reference_model_copies.py): the fixtures are what the manifest says (and, in
the build container, byte-identical to /root/reference), their forward syntax trees are the listed ones, the structural
recogniser names the three heads and rejects near misses.  Binding itself needs the GPU (tests/test_gpu_reference_heads.py)."""
import argparse
import hashlib
import json
import os
import sys

import pytest
import torch

from conftest import ROOT

FIX = os.path.join(ROOT, "tests", "golden", "reference_models")
PLUGIN = os.path.join(ROOT, "rechorus_amd", "rechorus")
CASES = [("general", "BPRMF", ["--emb_size", "32"]), ("general", "NeuMF", ["--emb_size", "32", "--layers", "[64]", "--dropout", "0.2"]),
         ("sequential", "SASRec", ["--emb_size", "32", "--num_layers", "2", "--num_heads", "2", "--history_max", "7"])]


def _build(sub, name, argv, monkeypatch, model_dir=None):
    monkeypatch.setattr(sys, "dont_write_bytecode", True)
    monkeypatch.setenv("RECHORUS_MODEL_DIRS", model_dir or os.path.join(FIX, sub))
    if PLUGIN not in sys.path:
        monkeypatch.syspath_prepend(PLUGIN)
    import main
    cls = main.find_class("model", (name, ""))
    args = cls.parse_model_args(argparse.ArgumentParser()).parse_args(argv)
    args.device, args.model_path, args.buffer = torch.device("cpu"), "", 1
    torch.manual_seed(0)
    return cls, cls(args, argparse.Namespace(n_users=20, n_items=50))


def test_fixtures_are_the_reference_files():
    man = json.load(open(os.path.join(FIX, "MANIFEST.json")))
    assert sorted(man) == ["context/DeepFM.py", "context/FM.py", "context/WideDeep.py", "general/BPRMF.py", "general/NeuMF.py",
                           "sequential/SASRec.py"]
    for rel, rec in man.items():
        data = open(os.path.join(FIX, rel), "rb").read()
        assert hashlib.sha256(data).hexdigest() == rec["sha256"], rel
        ref = os.path.join("/root/reference/src/models", rel)
        if os.path.exists(ref):      # build container: byte-identical to the reference
            assert open(ref, "rb").read() == data, rel
        assert b"rechorus_amd" not in data and b"HipEmbedding" not in data


@pytest.mark.parametrize("sub,name,argv", CASES)
def test_reference_model_files_are_recognised(sub, name, argv, monkeypatch):
    from rechorus_amd import dropin
    cls, model = _build(sub, name, argv, monkeypatch)
    assert os.path.realpath(cls.__init__.__globals__["__file__"]).startswith(os.path.realpath(FIX))
    h = dropin.forward_hash(cls)
    assert h in dropin.KNOWN_FORWARD_HASHES[name]
    assert h == json.load(open(os.path.join(FIX, "MANIFEST.json")))[sub + "/" + name + ".py"]["forward_hash"]
    assert dropin._kind(model) == name
    assert not hasattr(model, "hip_train_step")
    assert dropin.bind_known_head(model) is None and not hasattr(model, "hip_train_step")   # CPU model: nothing is bound
    assert "not on the GPU" in dropin.last_miss_reason          # ... and main.py's log says why
    # a model file that overrides a training hook the fused step would bypass (per-group optimizer settings, e.g. the reference's
    # Chorus.py:179-196) keeps its own route whatever its forward looks like
    sub_cls = type(name, (cls,), {"customize_parameters": lambda self: [{"params": list(self.parameters()), "lr": 1e-2}],
                                  "__module__": cls.__module__})
    hooked = sub_cls.__new__(sub_cls)
    hooked.__dict__.update(model.__dict__)
    assert dropin.bind_known_head(hooked) is None and "customize_parameters" in dropin.last_miss_reason
    # the plugin's own class of the same name brings its fused step and is left alone
    monkeypatch.delenv("RECHORUS_MODEL_DIRS")
    import main
    mirror = main.find_class("model", (name, ""))
    assert hasattr(mirror, "hip_train_step") and mirror.__module__.startswith("models.")


def test_near_misses_are_not_recognised(monkeypatch, tmp_path):
    from rechorus_amd import dropin
    _, bpr = _build("general", "BPRMF", ["--emb_size", "32"], monkeypatch)
    bpr.extra = torch.nn.Linear(4, 4)                     # one more parameter than the head has
    assert dropin._kind(bpr) is None
    _, neu = _build("general", "NeuMF", ["--emb_size", "32", "--layers", "[64,16]"], monkeypatch)
    assert dropin._kind(neu) == "NeuMF"                   # (towers of any depth are the NeuMF head; the mirror picks the kernels)
    neu.dropout_layer = torch.nn.Identity()
    assert dropin._kind(neu) is None
    _, sas = _build("sequential", "SASRec", ["--emb_size", "32", "--num_heads", "2", "--history_max", "7"], monkeypatch)
    sas.num_heads = 3                                     # 32 is not divisible by 3 heads
    assert dropin._kind(sas) is None
    # an edited forward is not one of the known syntax trees (comments and whitespace do not count as edits)
    src = open(os.path.join(FIX, "general", "BPRMF.py")).read()
    d1, d2 = tmp_path / "a", tmp_path / "b"
    d1.mkdir(), d2.mkdir()
    (d1 / "BPRMF.py").write_text(src.replace("cf_u_vectors = self.u_embeddings(u_ids)", "cf_u_vectors = self.u_embeddings(u_ids)  # a comment"))
    (d2 / "BPRMF.py").write_text(src.replace("cf_u_vectors = self.u_embeddings(u_ids)", "cf_u_vectors = 2 * self.u_embeddings(u_ids)"))
    cls1, _ = _build("general", "BPRMF", ["--emb_size", "32"], monkeypatch, str(d1))
    cls2, _ = _build("general", "BPRMF", ["--emb_size", "32"], monkeypatch, str(d2))
    assert dropin.forward_hash(cls1) in dropin.KNOWN_FORWARD_HASHES["BPRMF"]
    assert dropin.forward_hash(cls2) not in dropin.KNOWN_FORWARD_HASHES["BPRMF"]


CTX_CASES = [("FM", mode, argv) for mode in ("CTR", "TopK") for argv in (["--emb_size", "16"],)] + \
            [(name, mode, ["--emb_size", "16", "--layers", "[32,8]", "--dropout", "0.2"]) for name in ("WideDeep", "DeepFM") for mode in ("CTR", "TopK")]


def _build_context(name, mode, argv, monkeypatch, numeric=False):
    monkeypatch.setattr(sys, "dont_write_bytecode", True)
    monkeypatch.setenv("RECHORUS_MODEL_DIRS", os.path.join(FIX, "context"))
    if PLUGIN not in sys.path:
        monkeypatch.syspath_prepend(PLUGIN)
    import main
    cls = main.find_class("model", (name, mode))
    args = cls.parse_model_args(argparse.ArgumentParser()).parse_args(argv + ["--loss_n", "BCE" if mode == "CTR" else "BPR"])
    args.device, args.model_path, args.buffer = torch.device("cpu"), "", 1
    args.include_item_features = args.include_user_features = args.include_situation_features = 1
    fmax = {"user_id": 20, "item_id": 50, "i_cat_c": 7, "u_grp_c": 4, "c_hour_c": 24}
    corpus = argparse.Namespace(n_users=20, n_items=50, feature_max=fmax, item_feature_names=["i_cat_c"] + (["i_price_f"] if numeric else []),
                                user_feature_names=["u_grp_c"], situation_feature_names=["c_hour_c"])
    torch.manual_seed(0)
    return cls, cls(args, corpus)


@pytest.mark.parametrize("name,mode,argv", CTX_CASES)
def test_reference_context_model_files_are_recognised(name, mode, argv, monkeypatch):
    """the reference's own FM.py / WideDeep.py / DeepFM.py: class found in the fixture, forward syntax trees listed, the structural
    recogniser names the head (also with a numeric field), the mixin carries the plugin's head methods and no construction"""
    from rechorus_amd import dropin
    cls, model = _build_context(name, mode, argv, monkeypatch)
    kind = name + mode
    assert os.path.realpath(sys.modules.get(cls.__module__, None).__file__ if cls.__module__ in sys.modules else
                            cls.forward.__globals__["__file__"]).startswith(os.path.realpath(FIX))
    h = dropin.forward_hash(cls)
    assert h in dropin.KNOWN_FORWARD_HASHES[kind]
    assert h == json.load(open(os.path.join(FIX, "MANIFEST.json")))["context/" + name + ".py"]["forward_hashes"][kind]
    assert dropin._context_kind(model) == kind
    assert dropin.bind_known_head(model) is None and getattr(type(model), "_rc_bound_head", None) is None   # CPU model: nothing is bound
    mix = dropin._context_mixin(kind)
    for meth in ("forward", "_get_embeddings_FM", "_fused_fields", "_head_terms", "_rows_opt"):
        assert meth in mix.__dict__, meth
    assert not any(n.startswith("_define") or n.startswith("parse_model_args") or n == "__init__" for n in mix.__dict__)
    _, with_numeric = _build_context(name, mode, argv, monkeypatch, numeric=True)
    assert dropin._context_kind(with_numeric) == kind
    feed = dropin._context_probe_feed(with_numeric, kind, torch.device("cpu"))
    assert feed["i_price_f"].dtype == torch.float32 and feed["i_cat_c"].shape == feed["item_id"].shape and feed["u_grp_c"].dim() == 1
    if name == "FM":      # (FM.py builds its own nn.Embedding tables; WideDeep.py / DeepFM.py inherit the plugin's FMBase: GPU-only tables)
        with torch.no_grad():
            with_numeric.eval()
            out = with_numeric(dict(feed))["prediction"]       # the model file's own forward runs on the probe feed
        assert out.numel() == feed["item_id"].numel()
    # near misses: one more parameter; a class of another name
    model.extra = torch.nn.Linear(3, 3)
    assert dropin._context_kind(model) is None
    del model.extra
    model.__class__ = type("Something" + mode, (cls,), {})
    assert dropin._context_kind(model) is None
