"""Build container only (needs /root/reference; skipped on the GPU box, where it does not exist): the drop-in claim of the
plugin surface checked with a model FILE FROM THE REFERENCE, unmodified -- src/models/sequential/GRU4Rec.py is found
through RECHORUS_MODEL_DIRS, imports the mirror's models.BaseModel in place of the reference's, parses its own flags,
is constructed, has its nn.Embedding table counted by adopt_embeddings (state_dict keys unchanged) and runs its forward.
Nothing is written under /root/reference (no byte code either)."""
import argparse
import copy
import os
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT

REF_MODELS = "/root/reference/src/models/sequential"
PLUGIN = os.path.join(ROOT, "rechorus_amd", "rechorus")
pytestmark = pytest.mark.skipif(not os.path.exists(os.path.join(REF_MODELS, "GRU4Rec.py")), reason="reference tree not present")


@pytest.fixture()
def plugin_main(monkeypatch):
    monkeypatch.setattr(sys, "dont_write_bytecode", True)      # the reference tree stays untouched
    monkeypatch.setenv("RECHORUS_MODEL_DIRS", REF_MODELS)
    if PLUGIN not in sys.path:
        monkeypatch.syspath_prepend(PLUGIN)
    import main
    return main


def test_reference_gru4rec_file_resolves_against_the_mirror(plugin_main):
    from models.BaseModel import SequentialModel as MirrorSequential
    from rechorus_amd import nn as hnn
    cls = plugin_main.find_class("model", ("GRU4Rec", ""))
    assert os.path.realpath(sys.modules[cls.__module__].__file__ if cls.__module__ in sys.modules else
                            cls.__init__.__globals__["__file__"]).startswith("/root/reference/")
    assert issubclass(cls, MirrorSequential)                    # the reference file's base class import hit the mirror
    assert cls.reader == "SeqReader" and cls.runner == "BaseRunner"
    # its own flags, through the reference's hook chain (parse_model_args -> SequentialModel -> GeneralModel -> BaseModel)
    parser = cls.parse_model_args(argparse.ArgumentParser())
    args = parser.parse_args(["--emb_size", "24", "--hidden_size", "40", "--history_max", "7", "--num_neg", "3"])
    assert (args.emb_size, args.hidden_size, args.history_max, args.num_neg) == (24, 40, 7, 3)
    args.device, args.model_path, args.buffer = torch.device("cpu"), "", 1
    corpus = argparse.Namespace(n_users=30, n_items=120)
    torch.manual_seed(0)
    model = cls(args, corpus)
    assert set(dict(model.named_parameters())) >= {"i_embeddings.weight", "rnn.weight_ih_l0", "out.weight"}
    # one nn.Embedding table would move onto the engine; checkpoint keys stay what the reference wrote
    adopted = copy.deepcopy(model)
    assert hnn.adopt_embeddings(adopted) == 1 and type(adopted.i_embeddings).__name__ == "HipEmbedding"
    assert set(adopted.state_dict()) == set(model.state_dict())
    # forward + the inherited BPR loss on the CPU (plain torch), as the reference would
    rng = np.random.default_rng(0)
    B, L, C = 16, 7, 4
    lengths = rng.integers(1, L + 1, size=B)
    hist = rng.integers(1, 120, size=(B, L)) * (np.arange(L)[None, :] < lengths[:, None])
    feed = {"user_id": torch.from_numpy(rng.integers(1, 30, size=B)), "item_id": torch.from_numpy(rng.integers(1, 120, size=(B, C))),
            "history_items": torch.from_numpy(hist), "lengths": torch.from_numpy(lengths), "batch_size": B, "phase": "train"}
    out = model(feed)
    assert tuple(out["prediction"].shape) == (B, C) and bool(torch.isfinite(out["prediction"]).all())
    # the inherited loss is the engine's HIP kernel: on CPU tensors it refuses loudly (no CPU path) -- the rest of the
    # iteration (loss, backward, optimizer, main.py end to end) is what tests/test_gpu_dropin.py runs on the GPU with a
    # model file written to the same surface
    with pytest.raises(RuntimeError, match="no CPU path"):
        model.loss(out)
    assert not os.path.exists(os.path.join(REF_MODELS, "__pycache__"))


REF_GENERAL = "/root/reference/src/models/general"


@pytest.mark.parametrize("subdir,name,argv,n_tables", [
    ("general", "BPRMF", ["--emb_size", "32", "--num_neg", "3"], 2),
    ("general", "NeuMF", ["--emb_size", "32", "--layers", "[64]", "--num_neg", "3", "--dropout", "0"], 4),
    ("sequential", "SASRec", ["--emb_size", "32", "--num_layers", "1", "--num_heads", "2", "--history_max", "7", "--num_neg", "3",
                              "--dropout", "0"], 2),
])
def test_the_references_own_headline_model_files_drop_in(subdir, name, argv, n_tables, monkeypatch):
    """north star: "existing model files drop in".  The reference's OWN BPRMF.py / NeuMF.py / SASRec.py (the three heads of the hot
    path), unmodified, found through RECHORUS_MODEL_DIRS ahead of the mirror's files of the same name: they import the mirror's
    base classes and layers, parse their flags, are constructed, hand their nn.Embedding tables to the engine with the
    reference's state_dict keys, and their forward (plain torch on the CPU here) agrees with the numpy oracle of the same head --
    the arithmetic the HIP kernels are held to on the GPU."""
    from oracle import bprmf_oracle as BO, neumf_oracle as NO, sasrec_oracle as SO
    monkeypatch.setattr(sys, "dont_write_bytecode", True)
    ref_dir = "/root/reference/src/models/" + subdir
    monkeypatch.setenv("RECHORUS_MODEL_DIRS", ref_dir)
    if PLUGIN not in sys.path:
        monkeypatch.syspath_prepend(PLUGIN)
    import main
    from rechorus_amd import nn as hnn
    cls = main.find_class("model", (name, ""))
    assert os.path.realpath(cls.__init__.__globals__["__file__"]) == os.path.join(ref_dir, name + ".py")
    base = "SequentialModel" if subdir == "sequential" else "GeneralModel"
    import models.BaseModel as mirror_base
    assert issubclass(cls, getattr(mirror_base, base)) and mirror_base.__file__.startswith(PLUGIN)
    args = cls.parse_model_args(argparse.ArgumentParser()).parse_args(argv)
    args.device, args.model_path, args.buffer = torch.device("cpu"), "", 1
    n_users, n_items = 25, 90
    torch.manual_seed(3)
    model = cls(args, argparse.Namespace(n_users=n_users, n_items=n_items))
    model.eval()
    with torch.no_grad():
        for p in model.parameters():
            p.mul_(20.0)        # (the reference's 0.01-std init leaves the ReLU / softmax inputs near 0)
    keys = set(model.state_dict())
    adopted = copy.deepcopy(model)
    assert hnn.adopt_embeddings(adopted) == n_tables and set(adopted.state_dict()) == keys
    rng = np.random.default_rng(1)
    B, C, L = 12, 4, 7
    uid, iid = rng.integers(1, n_users, size=B), rng.integers(1, n_items, size=(B, C))
    feed = {"user_id": torch.from_numpy(uid), "item_id": torch.from_numpy(iid), "batch_size": B, "phase": "train"}
    P = {k: v.detach().numpy() for k, v in model.state_dict().items()}
    if name == "SASRec":
        lengths = rng.integers(1, L + 1, size=B)
        hist = rng.integers(1, n_items, size=(B, L)) * (np.arange(L)[None, :] < lengths[:, None])
        feed.update(history_items=torch.from_numpy(hist), lengths=torch.from_numpy(lengths))
        want = SO.forward(P, hist, lengths, iid, args.num_heads)
    elif name == "NeuMF":
        want = NO.forward(P, uid, iid)[0]
    else:
        want = BO.gather_dot(P["u_embeddings.weight"], P["i_embeddings.weight"], uid, iid)
    with torch.no_grad():
        pred = model(feed)["prediction"].numpy()
    assert pred.shape == (B, C)
    np.testing.assert_allclose(pred, want, rtol=2e-5, atol=2e-5 * float(np.abs(want).max()))
    assert not os.path.exists(os.path.join(ref_dir, "__pycache__"))
