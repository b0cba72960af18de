"""CPU: the known-head recognition of rechorus_amd/dropin.py.
  * its table of known forward syntax trees is the reference's: checked against the reference's OWN model files where they lie
    (/root/reference/src/models -- build container only; nothing of them is kept in this repository);
  * the structural recogniser names the heads of model files written by a user (tests/user_models/: plain torch layers over the
    plugin's task bases, this repository's own code) and rejects near misses.
Binding itself needs the GPU (tests/test_gpu_model_file_heads.py)."""
import argparse
import difflib
import os
import sys

import pytest
import torch

from conftest import ROOT

FIX = os.path.join(ROOT, "tests", "user_models")
REF = "/root/reference/src/models"
PLUGIN = os.path.join(ROOT, "rechorus_amd", "rechorus")
FILES = ["context/DeepFM.py", "context/FM.py", "context/WideDeep.py", "general/BPRMF.py", "general/NeuMF.py", "sequential/SASRec.py"]
CASES = [("general", "BPRMF", ["--emb_size", "32"]), ("general", "NeuMF", ["--emb_size", "32", "--layers", "[64]", "--dropout", "0.2"]),
         ("sequential", "SASRec", ["--emb_size", "32", "--num_layers", "2", "--num_heads", "2", "--history_max", "7"])]
needs_reference = pytest.mark.skipif(not os.path.isdir(REF), reason="the reference's model files exist in the build container only")


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


def _code_lines(path):
    return [l.strip() for l in open(path).read().splitlines() if l.strip() and not l.strip().startswith("#")]


def test_user_model_files_are_plain_torch_and_nobodys_copy():
    for rel in FILES:
        text = open(os.path.join(FIX, rel)).read()
        assert "rechorus_amd" not in text and "HipEmbedding" not in text and "hnn." not in text, rel
        ref = os.path.join(REF, rel)
        if os.path.exists(ref):      # build container: far from the reference's file of the same name
            ratio = difflib.SequenceMatcher(None, _code_lines(os.path.join(FIX, rel)), _code_lines(ref)).ratio()
            assert ratio < 0.35, (rel, ratio)


@needs_reference
@pytest.mark.parametrize("sub,name,argv", CASES)
def test_known_forward_hashes_are_the_reference_files(sub, name, argv, monkeypatch):
    """the reference's own BPRMF.py / NeuMF.py / SASRec.py, imported where they lie: class found there, forward syntax tree listed"""
    from rechorus_amd import dropin
    cls, model = _build(sub, name, argv, monkeypatch, os.path.join(REF, sub))
    assert os.path.realpath(cls.__init__.__globals__["__file__"]).startswith(os.path.realpath(REF))
    assert dropin.forward_hash(cls) in dropin.KNOWN_FORWARD_HASHES[name]
    assert dropin._kind(model) == name


@pytest.mark.parametrize("sub,name,argv", CASES)
def test_user_model_files_are_recognised_by_structure(sub, name, argv, monkeypatch):
    from rechorus_amd import dropin
    cls, model = _build(sub, name, argv, monkeypatch)
    assert os.path.realpath(cls.__init__.__globals__["__file__"]).startswith(os.path.realpath(FIX))
    h = dropin.forward_hash(cls)
    assert h is not None and h not in dropin.KNOWN_FORWARD_HASHES[name]      # (not the reference's syntax tree: the probe decides)
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
    cls0, bpr = _build("general", "BPRMF", ["--emb_size", "32"], monkeypatch)
    bpr.extra = torch.nn.Linear(4, 4)                     # one more parameter than the head has
    assert dropin._kind(bpr) is None
    _, neu = _build("general", "NeuMF", ["--emb_size", "32", "--layers", "[64,16]"], monkeypatch)
    assert dropin._kind(neu) == "NeuMF"                   # (towers of any depth are the NeuMF head; the mirror picks the kernels)
    neu.dropout_layer = torch.nn.Identity()
    assert dropin._kind(neu) is None
    _, sas = _build("sequential", "SASRec", ["--emb_size", "32", "--num_heads", "2", "--history_max", "7"], monkeypatch)
    sas.num_heads = 3                                     # 32 is not divisible by 3 heads
    assert dropin._kind(sas) is None
    # the syntax-tree hash: comments and whitespace do not count as edits, a changed expression does
    src = open(os.path.join(FIX, "general", "BPRMF.py")).read()
    line = "users = self.u_embeddings(feed_dict['user_id'])"
    assert line in src
    d1, d2 = tmp_path / "a", tmp_path / "b"
    d1.mkdir(), d2.mkdir()
    (d1 / "BPRMF.py").write_text(src.replace(line, line + "  # a comment"))
    (d2 / "BPRMF.py").write_text(src.replace(line, "users = 2 * self.u_embeddings(feed_dict['user_id'])"))
    cls1, _ = _build("general", "BPRMF", ["--emb_size", "32"], monkeypatch, str(d1))
    cls2, _ = _build("general", "BPRMF", ["--emb_size", "32"], monkeypatch, str(d2))
    assert dropin.forward_hash(cls1) == dropin.forward_hash(cls0)
    assert dropin.forward_hash(cls2) != dropin.forward_hash(cls0)


CTX_CASES = [("FM", mode, argv) for mode in ("CTR", "TopK") for argv in (["--emb_size", "16"],)] + \
            [(name, mode, ["--emb_size", "16", "--layers", "[32,8]", "--dropout", "0.2"]) for name in ("WideDeep", "DeepFM") for mode in ("CTR", "TopK")]


def _build_context(name, mode, argv, monkeypatch, numeric=False, model_dir=None):
    monkeypatch.setattr(sys, "dont_write_bytecode", True)
    monkeypatch.setenv("RECHORUS_MODEL_DIRS", model_dir or os.path.join(FIX, "context"))
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


@needs_reference
@pytest.mark.parametrize("name,mode,argv", CTX_CASES)
def test_known_context_forward_hashes_are_the_reference_files(name, mode, argv, monkeypatch):
    """the reference's own FM.py / WideDeep.py / DeepFM.py, imported where they lie: forward syntax trees listed, head named"""
    from rechorus_amd import dropin
    cls, model = _build_context(name, mode, argv, monkeypatch, model_dir=os.path.join(REF, "context"))
    assert os.path.realpath(cls.forward.__globals__["__file__"]).startswith(os.path.realpath(REF))
    assert dropin.forward_hash(cls) in dropin.KNOWN_FORWARD_HASHES[name + mode]
    assert dropin._context_kind(model) == name + mode


@pytest.mark.parametrize("name,mode,argv", CTX_CASES)
def test_user_context_model_files_are_recognised_by_structure(name, mode, argv, monkeypatch):
    """a user's FM.py / WideDeep.py / DeepFM.py: the structural recogniser names the head (also with a numeric field), the mixin
    carries the plugin's head methods and no construction, the model file's own forward runs on the probe feed"""
    from rechorus_amd import dropin
    cls, model = _build_context(name, mode, argv, monkeypatch)
    kind = name + mode
    assert os.path.realpath(cls.forward.__globals__["__file__"]).startswith(os.path.realpath(FIX))
    h = dropin.forward_hash(cls)
    assert h is not None and h not in dropin.KNOWN_FORWARD_HASHES[kind]
    assert dropin._context_kind(model) == kind
    assert dropin.bind_known_head(model) is None and getattr(type(model), "_rc_bound_head", None) is None   # CPU model: nothing is bound
    mix = dropin._context_mixin(kind)
    for meth in ("forward", "_get_embeddings_FM", "_fused_fields", "_head_terms", "_rows_opt"):
        assert meth in mix.__dict__, meth
    assert mix.__dict__["fm_term"] is (name != "WideDeep")
    assert not any(n.startswith("_define") or n.startswith("parse_model_args") or n == "__init__" for n in mix.__dict__)
    _, with_numeric = _build_context(name, mode, argv, monkeypatch, numeric=True)
    assert dropin._context_kind(with_numeric) == kind
    feed = dropin._context_probe_feed(with_numeric, kind, torch.device("cpu"))
    assert feed["i_price_f"].dtype == torch.float32 and feed["i_cat_c"].shape == feed["item_id"].shape and feed["u_grp_c"].dim() == 1
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
