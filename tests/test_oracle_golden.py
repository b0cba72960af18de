"""CPU: the numpy oracle must reproduce what the reference itself computed (golden fixtures
made by tests/golden/make_golden.py from /root/reference/src).  This is what pins the oracle."""
import numpy as np
import pytest

from conftest import assert_close, assert_update_close, golden_cases, load_golden
from oracle import bprmf_oracle as O


@pytest.mark.parametrize("case", golden_cases())
def test_forward_loss_grad(case):
    g = load_golden(case)
    U, I, uid, iid = g["U0"], g["I0"], g["uid"], g["iid"]
    pred = O.gather_dot(U, I, uid, iid)
    assert_close(pred, g["pred"], what="pred")
    assert_close(O.bpr_loss(g["pred"]), g["loss"], what="loss")
    assert_close(O.bpr_loss_grad(g["pred"]), g["gpred"], what="gpred")
    loss, _, gp, GU, GI = O.bprmf_dense_grads(U, I, uid, iid)
    assert_close(loss, g["loss"], what="loss(e2e)")
    assert_close(GU, g["GU"], what="GU")
    assert_close(GI, g["GI"], what="GI")


@pytest.mark.parametrize("case", golden_cases())
@pytest.mark.parametrize("tag,opt", [("SGD_l20", "SGD"), ("SGD_l20.001", "SGD"), ("Adam_l20", "Adam"),
                                     ("Adam_l20.0001", "Adam"), ("Adagrad_l20.0001", "Adagrad")])
def test_dense_optimizer_steps(case, tag, opt):
    """two reference fit() iterations (dense torch.optim semantics) == oracle rowwise=False"""
    g = load_golden(case)
    lr, l2 = g[tag + "_hyper"]
    U, I = g["U0"].copy(), g["I0"].copy()
    sU, sI = O.new_state(U, opt), O.new_state(I, opt)
    for step, (u, i) in enumerate(((g["uid"], g["iid"]), (g["uid2"], g["iid2"])), 1):
        loss, _ = O.bprmf_train_step(U, I, sU, sI, u, i, opt=opt, lr=lr, l2=l2, step=step,
                                     rowwise=False)
        assert_close(loss, g[tag + "_losses"][step - 1], what=f"loss step {step}")
        ex = 1e-3 * lr if opt in ("Adam", "Adagrad") else 0.0
        assert_update_close(U, g["U0"], g[f"{tag}_U{step}"], what=f"dU step {step}", extra_atol=ex)
        assert_update_close(I, g["I0"], g[f"{tag}_I{step}"], what=f"dI step {step}", extra_atol=ex)


def test_rowwise_sgd_l2_zero_equals_dense():
    """with l2 = 0, SGD on touched rows only IS the dense step (untouched rows have zero grad)"""
    g = load_golden("bprmf_k99_d64_zipf")
    U, I = g["U0"].copy(), g["I0"].copy()
    O.bprmf_train_step(U, I, {}, {}, g["uid"], g["iid"], opt="SGD", lr=0.05, l2=0.0, rowwise=True)
    assert_close(U, g["SGD_l20_U1"], what="U")
    assert_close(I, g["SGD_l20_I1"], what="I")


def test_metrics_known_answer():
    pred = np.array([[0.9, 0.1, 0.5], [0.2, 0.3, 0.1], [0.5, 0.5, 0.5]], dtype=np.float32)
    r = O.evaluate_method(pred, [1, 2], ["HR", "NDCG"])
    # ranks (ties count against the target, helpers/BaseRunner.py:63): 1, 2, 3
    assert r["HR@1"] == pytest.approx(1 / 3)
    assert r["HR@2"] == pytest.approx(2 / 3)
    assert r["NDCG@2"] == pytest.approx((1.0 + 1 / np.log2(3)) / 3)


@pytest.mark.parametrize("case", ["bprmf_k1_d64", "bprmf_k99_d64_zipf", "bprmf_k5_d48"])
@pytest.mark.parametrize("tag,opt", [("SGD_l20.001", "SGD"), ("Adam_l20.0001", "Adam")])
def test_torch_port_matches_reference(case, tag, opt):
    """the torch-CPU port that bench.py times as cpu_baseline reproduces the reference's fit()"""
    import torch
    from oracle.torch_port import BprmfTorchPort
    g = load_golden(case)
    lr, l2 = (float(x) for x in g[tag + "_hyper"])
    n_users, n_items, d = g["U0"].shape[0], g["I0"].shape[0], g["U0"].shape[1]
    m = BprmfTorchPort(n_users, n_items, d)
    with torch.no_grad():
        m.u_embeddings.weight.copy_(torch.from_numpy(g["U0"]))
        m.i_embeddings.weight.copy_(torch.from_numpy(g["I0"]))
    optim = m.make_optimizer(opt, lr, l2)
    for step, (u, i) in enumerate(((g["uid"], g["iid"]), (g["uid2"], g["iid2"])), 1):
        loss = m.fit_step(optim, torch.from_numpy(u), torch.from_numpy(i))
        assert_close(loss.numpy(), g[tag + "_losses"][step - 1], what=f"loss step {step}")
    ex = 1e-3 * lr if opt == "Adam" else 0.0
    assert_update_close(m.u_embeddings.weight.detach().numpy(), g["U0"], g[tag + "_U2"], what="dU", extra_atol=ex)
    assert_update_close(m.i_embeddings.weight.detach().numpy(), g["I0"], g[tag + "_I2"], what="dI", extra_atol=ex)


@pytest.mark.parametrize("tag,lr,l2", [("lr1_l20.0001", 1.0, 1e-4), ("lr0.001_l20", 1e-3, 0.0)])
def test_oracle_adadelta_matches_the_reference(tag, lr, l2):
    """--optimizer Adadelta: three fit() iterations of the reference (tests/golden/make_golden_adadelta.py) vs the
    numpy restatement of torch.optim.Adadelta's single-tensor path (dense: every row of both tables steps)"""
    import os
    from conftest import ROOT
    g = np.load(os.path.join(ROOT, "tests", "golden", "adadelta_bprmf_d64.npz"))
    U, I = g["U0"].copy(), g["I0"].copy()
    sU, sI = O.new_state(U, "Adadelta"), O.new_state(I, "Adadelta")
    for step in (1, 2, 3):
        loss, _ = O.bprmf_train_step(U, I, sU, sI, g[f"uid{step}"], g[f"iid{step}"], opt="Adadelta", lr=lr, l2=l2, step=step,
                                     rowwise=False)
        assert_close(loss, g[tag + "_losses"][step - 1], what=f"loss {step}")
        assert_update_close(U, g["U0"], g[f"{tag}_U{step}"], what=f"dU {step}", extra_atol=1e-3 * lr * 1e-3)
        assert_update_close(I, g["I0"], g[f"{tag}_I{step}"], what=f"dI {step}", extra_atol=1e-3 * lr * 1e-3)
