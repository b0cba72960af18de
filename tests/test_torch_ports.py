"""CPU: the torch-CPU ports that bench.py times as `cpu_baseline` (oracle/torch_port.py, kind "port": the reference
itself does not exist on the GPU box) reproduce the reference's own outputs -- prediction, loss and every parameter
gradient of the golden fixtures generated from /root/reference (tests/golden/make_golden_neumf.py, _sasrec.py)."""
import glob
import os

import numpy as np
import pytest
import torch

from conftest import ROOT, assert_close
from oracle.torch_port import NeumfTorchPort, SasrecTorchPort

GOLD = os.path.join(ROOT, "tests", "golden")


def _load_state(model, g):
    sd = {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("P0/")}
    missing, unexpected = model.load_state_dict(sd, strict=True)
    assert not missing and not unexpected


def _check_grads(model, g, what):
    for name, p in model.named_parameters():
        key = "G/" + name
        if key in g.files:
            scale = max(float(np.abs(g[k]).max()) for k in g.files if k.startswith("G/") and g[k].ndim == g[key].ndim)
            assert_close(p.grad.numpy(), g[key], abs_floor=1e-6 * scale, what=f"{what} grad {name}")


@pytest.mark.parametrize("path", sorted(p for p in glob.glob(os.path.join(GOLD, "neumf_d*.npz"))))
def test_neumf_port_matches_the_reference(path):
    g = np.load(path)
    n_users, d = g["P0/mf_u_embeddings.weight"].shape
    n_items = g["P0/mf_i_embeddings.weight"].shape[0]
    l1 = g["P0/mlp.0.weight"].shape[0]
    m = NeumfTorchPort(n_users, n_items, d, layers=(l1,))
    _load_state(m, g)
    pred = m(torch.from_numpy(g["uid"]), torch.from_numpy(g["iid"]))
    assert_close(pred.detach().numpy(), g["pred"], what="pred")
    loss = m.loss(pred)
    assert_close(float(loss), float(g["loss"]), what="loss")
    loss.backward()
    _check_grads(m, g, os.path.basename(path))


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLD, "sasrec_d*.npz"))))
def test_sasrec_port_matches_the_reference(path):
    g = np.load(path)
    n_items, d = g["P0/i_embeddings.weight"].shape
    hist_max = g["P0/p_embeddings.weight"].shape[0] - 1
    n_layers = 1 + max(int(k.split(".")[1]) for k in g.files if k.startswith("P0/transformer_block."))
    n_heads = int(g["meta"][list(g["meta_names"]).index("n_heads")]) if "meta_names" in g.files else None
    if n_heads is None:  # file name carries it: sasrec_d64_l1_h4[_L50].npz
        n_heads = int(os.path.basename(path).split("_h")[1].split("_")[0].split(".")[0])
    m = SasrecTorchPort(n_items, d, hist_max, n_layers=n_layers, n_heads=n_heads)
    _load_state(m, g)
    pred = m(torch.from_numpy(g["hist"]), torch.from_numpy(g["len"]), torch.from_numpy(g["iid"]))
    assert_close(pred.detach().numpy(), g["pred"], what="pred")
    loss = m.loss(pred)
    assert_close(float(loss), float(g["loss"]), what="loss")
    loss.backward()
    _check_grads(m, g, os.path.basename(path))


@pytest.mark.parametrize("name", ["deepfm_ctr_d64", "deepfm_ctr_d16", "deepfm_mind_ctr_d64"])   # (mind: with the numeric c_day_f)
def test_deepfm_ctr_port_matches_the_reference(name):
    from oracle.torch_port import DeepfmCtrTorchPort
    g = np.load(os.path.join(GOLD, name + ".npz"), allow_pickle=True)
    fields = [str(f) for f in g["fields"]]
    fmax = {f: g[f"P0/context_embedding.{f}.weight"].shape[0] for f in fields}
    d = g["P0/context_embedding.item_id.weight"].shape[1]
    hidden = [g[k].shape[0] for k in sorted((k for k in g.files if k.startswith("P0/deep_layers.mlp.") and k.endswith(".weight")),
                                            key=lambda k: int(k.split(".")[2]))][:-1]
    m = DeepfmCtrTorchPort(fields, fmax, d, layers=hidden)
    _load_state(m, g)
    feed = {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("b1/")}
    pred = m(feed)
    assert_close(pred.detach().numpy(), g["pred"], what="pred")
    loss = m.loss(pred, feed["label"])
    assert_close(float(loss.detach()), float(g["loss"]), what="loss")
    loss.backward()
    _check_grads(m, g, name)
