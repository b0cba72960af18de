"""GPU, BASELINE.json configs[1] sizes (BPRMF d=64, K=99, 10M-item catalogue): the oracle
cannot finish these in seconds, so parity is checked through size-independent properties of
the training step (consistency between independent kernels, conservation of the SGD update,
untouched rows, permutation invariance, sortedness, bit-reproducibility) plus an oracle
comparison on a random SAMPLE of tuples and rows."""
import numpy as np
import pytest
import torch

from conftest import assert_close, assert_update_close
from oracle import bprmf_oracle as O

pytestmark = pytest.mark.gpu

N_ITEMS, N_USERS, D, B, K = 10_000_001, 1_000_001, 64, 8192, 99
LR = 200.0  # large on purpose: every update must be >> 1 fp32 ulp of the weights (see conservation test)


@pytest.fixture(scope="module")
def world(cuda):
    from rechorus_amd import engine
    g = torch.Generator(device=cuda)
    g.manual_seed(0)
    U = torch.empty((N_USERS, D), device=cuda).normal_(0, 0.01, generator=g)
    I = torch.empty((N_ITEMS, D), device=cuda).normal_(0, 0.01, generator=g)
    rng = np.random.default_rng(0)
    # Zipf(1.0) users / positives over seeded rank->id permutations, uniform negatives
    # (models/BaseModel.py:207), as BASELINE.md section 3 prescribes
    def zipf(n_rows, size):
        ranks = np.exp(rng.uniform(0, np.log(n_rows - 1), size=size)).astype(np.int64)  # ~1/r
        return (ranks * 2654435761 % (n_rows - 1)) + 1
    uid = torch.from_numpy(zipf(N_USERS, B)).to(cuda)
    pos = zipf(N_ITEMS, B)
    neg = rng.integers(1, N_ITEMS, size=(B, K))
    iid = torch.from_numpy(np.concatenate([pos[:, None], neg], axis=1)).to(cuda)
    return dict(eng=engine, U=U, I=I, uid=uid, iid=iid)


def test_fused_agrees_with_unfused_kernels(world):
    e = world["eng"]
    U, I, uid, iid = (world[k] for k in ("U", "I", "uid", "iid"))
    pred, loss_vec, gpred, ugrad = e.bprmf_fwd_bwd(U, I, uid, iid)
    pred2 = e.gather_dot(U, I, uid, iid)
    assert_close(pred.cpu().numpy(), pred2.cpu().numpy(), rtol=1e-6, atol_scale=1e-6, what="pred")
    loss2, lv2, gp2 = e.bpr_loss(pred2)
    assert_close(loss_vec.cpu().numpy(), lv2.cpu().numpy(), what="loss_vec")
    assert_close(gpred.cpu().numpy(), gp2.cpu().numpy(), what="gpred")
    # oracle on a sample of tuples
    idx = np.random.default_rng(1).choice(B, size=64, replace=False)
    u_s, i_s = uid[idx].cpu().numpy(), iid[idx].cpu().numpy()
    rows_u, inv_u = np.unique(u_s, return_inverse=True)
    rows_i, inv_i = np.unique(i_s, return_inverse=True)
    Us, Is = U[rows_u].cpu().numpy(), I[rows_i].cpu().numpy()
    want = O.gather_dot(Us, Is, inv_u, inv_i.reshape(i_s.shape))
    assert_close(pred[idx].cpu().numpy(), want, what="pred(sample)")
    assert_close(gpred[idx].cpu().numpy(), O.bpr_loss_grad(want, inv_b=1.0 / B), what="gpred(sample)")
    gu, _ = O.bprmf_row_grads(Us, Is, inv_u, inv_i.reshape(i_s.shape), O.bpr_loss_grad(want, inv_b=1.0 / B))
    assert_close(ugrad[idx].cpu().numpy(), gu, what="ugrad(sample)")


def test_sorted_keys_and_permutation(world):
    e = world["eng"]
    iid = world["iid"]
    keys, perm = e.sort_ids(iid, N_ITEMS)
    k = keys.to(torch.int64) & 0xFFFFFFFF
    p = perm.to(torch.int64) & 0xFFFFFFFF
    assert bool((k[1:] >= k[:-1]).all()), "keys not sorted"
    assert torch.equal(iid.reshape(-1)[p], k), "perm does not map ids to sorted keys"
    same = k[1:] == k[:-1]
    assert bool((p[1:][same] > p[:-1][same]).all()), "sort is not stable"
    assert torch.equal(torch.sort(p).values, torch.arange(p.numel(), device=p.device))


def test_sgd_step_conservation_untouched_rows_and_reproducibility(world):
    e = world["eng"]
    U0, I0, uid, iid = (world[k] for k in ("U", "I", "uid", "iid"))
    _, _, gpred, ugrad = e.bprmf_fwd_bwd(U0, I0, uid, iid, want_pred=False)
    results = []
    for _ in range(2):
        U, I = U0.clone(), I0.clone()
        tr = e.BprmfTrainer(U, I, opt="SGD", lr=LR, l2=0.0)
        loss = tr.step(uid, iid)
        results.append((U, I, loss.clone()))
    U, I, loss = results[0]
    assert torch.equal(U, results[1][0]) and torch.equal(I, results[1][1]), "not bit-reproducible"
    assert torch.equal(loss, results[1][2])
    # conservation: sum over rows of the update == -lr * sum over occurrences of the row grads.
    # LR is chosen so every per-element update is >> 1 ulp of the weights (with a realistic lr
    # the K tiny negative updates round away individually and the sum is dominated by fp32
    # rounding, in the reference too).  Tolerance: 1e-5 of the absolute-value sum of the
    # contributions + the rounding of n_touched fp32 stores.
    ulp = float(np.finfo(np.float32).eps) * float(I0.abs().max())
    contrib = gpred.double().abs().sum(1)[:, None] * U0[uid].double().abs()
    dI = (I.double() - I0.double()).sum(0)
    want_dI = -LR * (gpred.double().sum(1)[:, None] * U0[uid].double()).sum(0)
    n_touched = int(torch.unique(iid).numel())
    tol_I = 1e-5 * LR * contrib.sum(0) + 8 * np.sqrt(n_touched) * ulp
    assert bool(((dI - want_dI).abs() <= tol_I).all()), f"sum dI off by {(dI - want_dI).abs().max().item():.3e}"
    dU = (U.double() - U0.double()).sum(0)
    want_dU = -LR * ugrad.double().sum(0)
    tol_U = 1e-5 * LR * ugrad.double().abs().sum(0) + 8 * np.sqrt(B) * ulp
    assert bool(((dU - want_dU).abs() <= tol_U).all()), f"sum dU off by {(dU - want_dU).abs().max().item():.3e}"
    assert float(dU.abs().max()) > 100 * float(tol_U.max()), "conservation check is vacuous"
    # rows outside the batch are bit-identical
    touched = torch.zeros(N_ITEMS, dtype=torch.bool, device=I.device)
    touched[iid.reshape(-1)] = True
    assert torch.equal(I[~touched], I0[~touched])
    assert int((I[touched] != I0[touched]).any(dim=1).sum()) > 0.99 * int(touched.sum())
    # oracle on a sample of touched item rows: rebuild their gradient from (gpred, U0, uid)
    rows = torch.unique(iid.reshape(-1))[:: max(1, int(touched.sum()) // 200)][:200]
    flat = iid.reshape(-1)
    for r in rows[:50].tolist():
        occ = torch.nonzero(flat == r).reshape(-1)
        g = (gpred.reshape(-1)[occ][:, None] * U0[uid[occ // (K + 1)]]).sum(0)
        want = I0[r] - LR * g
        assert_update_close(I[r].cpu().numpy(), I0[r].cpu().numpy(), want.cpu().numpy(), what=f"item row {r}")


def test_negative_permutation_invariance(world):
    """permuting the K negatives inside each tuple changes nothing but fp summation order"""
    e = world["eng"]
    U0, I0, uid, iid = (world[k] for k in ("U", "I", "uid", "iid"))
    g = torch.Generator(device=iid.device)
    g.manual_seed(3)
    order = torch.argsort(torch.rand((B, K), device=iid.device, generator=g), dim=1) + 1
    iid_p = torch.cat([iid[:, :1], torch.gather(iid, 1, order)], dim=1).contiguous()
    outs = []
    for ids in (iid, iid_p):
        U, I = U0.clone(), I0.clone()
        loss = e.BprmfTrainer(U, I, opt="SGD", lr=LR).step(uid, ids).clone()
        outs.append((U, I, loss))
    assert_close(outs[1][2].cpu().numpy(), outs[0][2].cpu().numpy(), rtol=1e-6, what="loss")
    for a, b, a0, nm in ((outs[0][0], outs[1][0], U0, "U"), (outs[0][1], outs[1][1], I0, "I")):
        d0, d1 = (a - a0), (b - a0)
        err = (d0 - d1).abs().max().item()
        assert err <= 1e-5 * d0.abs().max().item() + 1e-9, f"{nm}: update changed by {err}"


def test_adam_rowwise_step_sample_vs_oracle(world):
    e = world["eng"]
    U0, I0, uid, iid = (world[k] for k in ("U", "I", "uid", "iid"))
    U, I = U0.clone(), I0.clone()
    tr = e.BprmfTrainer(U, I, opt="Adam", lr=1e-3, l2=1e-6)
    tr.step(uid, iid)
    _, _, gpred, ugrad = e.bprmf_fwd_bwd(U0, I0, uid, iid, want_pred=False)
    flat = iid.reshape(-1)
    for r in torch.unique(flat)[::40000][:20].tolist():
        occ = torch.nonzero(flat == r).reshape(-1)
        g = (gpred.reshape(-1)[occ][:, None] * U0[uid[occ // (K + 1)]]).sum(0).cpu().numpy()
        W = I0[r].cpu().numpy()[None].copy()
        st = {"m": np.zeros_like(W), "v": np.zeros_like(W)}
        O.opt_step_dense(W, g[None], st, "Adam", 1e-3, 1e-6, step=1)
        assert_update_close(I[r].cpu().numpy()[None], I0[r].cpu().numpy()[None], W, what=f"row {r}",
                            extra_atol=1e-6)
        assert_close(tr.mI[r].cpu().numpy()[None], st["m"], what="exp_avg", atol_scale=1e-4)


def test_step_ahead_at_the_bench_batch_sample_vs_oracle(world, cuda):
    """The configuration bench.py times: B = 65,536 tuples, K = 99, 10 M items, SGD, every step announcing the next batch
    (rc_bprmf_train_step_ahead consuming a plan prepared beside the previous step's updates).  The step that RUNS FROM A
    PREPARED PLAN is checked against the oracle arithmetic on sampled item / user rows, rebuilt from the unfused kernels'
    gpred / ugrad on the pre-step tables, plus: untouched rows bit-identical, and bit-equality with the same two steps
    without look-ahead."""
    e = world["eng"]
    U0, I0 = world["U"], world["I"]
    Bb = 65_536
    g = torch.Generator(device=cuda)
    g.manual_seed(11)
    def batch():
        ranks_u = torch.exp(torch.rand(Bb, generator=g, device=cuda, dtype=torch.float64) * np.log(N_USERS - 1)).to(torch.int64).clamp_(1, N_USERS - 1)
        ranks_p = torch.exp(torch.rand(Bb, generator=g, device=cuda, dtype=torch.float64) * np.log(N_ITEMS - 1)).to(torch.int64).clamp_(1, N_ITEMS - 1)
        uid = (ranks_u * 2654435761) % (N_USERS - 1) + 1
        pos = (ranks_p * 2654435761) % (N_ITEMS - 1) + 1
        neg = torch.randint(1, N_ITEMS, (Bb, K), generator=g, device=cuda)
        return uid.contiguous(), torch.cat([pos[:, None], neg], dim=1).contiguous()
    b1, b2 = batch(), batch()
    lr = 50.0
    runs = []
    for ahead in (True, False):
        U, I = U0.clone(), I0.clone()
        tr = e.BprmfTrainer(U, I, opt="SGD", lr=lr, l2=0.0)
        tr.step(*b1, next_batch=b2 if ahead else None)
        if ahead:
            assert tr._ticket.generation != 0 and tr._ticket.B == Bb and tr._ticket.flavour == 1, "no plan was prepared"
            U1, I1 = U.clone(), I.clone()           # tables between the steps (the stream orders the copy behind step 1)
            assert tr._generation_of(*b2) == tr._ticket.generation, "step 2 would not consume the prepared plan"
        loss2 = tr.step(*b2).clone()
        runs.append((U, I, loss2))
    (U, I, loss), (Ub, Ib, lossb) = runs
    assert torch.equal(U, Ub) and torch.equal(I, Ib) and torch.equal(loss, lossb), "look-ahead changed the result"
    uid, iid = b2
    _, lv, gpred, ugrad = e.bprmf_fwd_bwd(U1, I1, uid, iid, want_pred=False)
    assert_close(loss.cpu().numpy()[0], float(lv.double().mean()), rtol=1e-5, what="loss of the step run from the prepared plan")
    flat = iid.reshape(-1)
    cnt = torch.bincount(flat, minlength=N_ITEMS)
    touched = cnt > 0
    assert torch.equal(I[~touched], I1[~touched])
    # sampled item rows: singletons (updated inside the fused kernel), doubles / triples (plan-driven update) and the
    # hottest rows (chunked path)
    order = torch.argsort(cnt, descending=True)
    picks = torch.cat([order[:8], torch.nonzero(cnt == 1).reshape(-1)[:: 40000][:40], torch.nonzero(cnt == 2).reshape(-1)[:: 15000][:40],
                       torch.nonzero(cnt == 3).reshape(-1)[:: 4000][:20]])
    assert int((cnt[picks] == 1).sum()) >= 20 and int((cnt[picks] == 2).sum()) >= 20 and int(cnt[picks].max()) > 32
    # the numpy ORACLE (oracle/bprmf_oracle.py) on sampled tuples of this very batch: 100 tuples spread over the batch plus every
    # tuple that touches a picked row with at most three occurrences -- their scores' gradient, loss rows and user gradients are
    # compared directly, and the expected value of those item rows is rebuilt from the oracle's numbers alone
    few = picks[cnt[picks] <= 3]
    occ_few = torch.nonzero(torch.isin(flat, few)).reshape(-1)
    tsel = torch.unique(torch.cat([torch.arange(0, Bb, Bb // 100, device=cuda), occ_few // (K + 1)]))
    u_np, i_np = uid[tsel].cpu().numpy(), iid[tsel].cpu().numpy()
    U_rows, I_rows = U1[uid[tsel]].cpu().numpy(), I1[iid[tsel]].cpu().numpy()           # [T, d], [T, C, d]
    pred_o = np.einsum("td,tcd->tc", U_rows.astype(np.float64), I_rows.astype(np.float64)).astype(np.float32)
    g_o = O.bpr_loss_grad(pred_o, inv_b=1.0 / Bb)
    assert_close(gpred[tsel].cpu().numpy(), g_o, what="gpred of sampled tuples vs the oracle")
    assert_close(lv[tsel].cpu().numpy(), O.bpr_loss_rows(pred_o)[0], what="loss rows of sampled tuples vs the oracle")
    ug_o = np.einsum("tc,tcd->td", g_o.astype(np.float64), I_rows.astype(np.float64))
    assert_close(ugrad[tsel].cpu().numpy(), ug_o, what="user gradients of sampled tuples vs the oracle")
    where_t = {int(t): k for k, t in enumerate(tsel.tolist())}
    gp = gpred.reshape(-1).double()
    for r in picks.tolist():
        occ = torch.nonzero(flat == r).reshape(-1)
        if int(cnt[r]) <= 3:     # from the oracle's gradients only
            gsum = np.zeros(U_rows.shape[1])
            for o in occ.tolist():
                k = where_t[o // (K + 1)]
                gsum += float(g_o[k, o % (K + 1)]) * U_rows[k].astype(np.float64)
            want = (I1[r].double().cpu().numpy() - lr * gsum).astype(np.float32)
            assert_update_close(I[r].cpu().numpy(), I1[r].cpu().numpy(), want, what=f"item row {r} (n={int(cnt[r])}) vs the oracle")
            continue
        gsum = (gp[occ][:, None] * U1[uid[occ // (K + 1)]].double()).sum(0)
        want = (I1[r].double() - lr * gsum).float()
        assert_update_close(I[r].cpu().numpy(), I1[r].cpu().numpy(), want.cpu().numpy(), what=f"item row {r} (n={int(cnt[r])})")
    ucnt = torch.bincount(uid, minlength=N_USERS)
    uorder = torch.argsort(ucnt, descending=True)
    for r in torch.cat([uorder[:6], torch.nonzero(ucnt == 1).reshape(-1)[::300][:30]]).tolist():
        rows = torch.nonzero(uid == r).reshape(-1)
        want = (U1[r].double() - lr * ugrad[rows].double().sum(0)).float()
        assert_update_close(U[r].cpu().numpy(), U1[r].cpu().numpy(), want.cpu().numpy(), what=f"user row {r} (n={int(ucnt[r])})")
    assert torch.equal(U[ucnt == 0], U1[ucnt == 0])


def test_bench_contract_line(cuda):
    import os
    """bench.py prints ONE JSON line with the driver's keys, the roofline object and the CPU baseline"""
    import json
    import subprocess
    import sys
    from conftest import ROOT
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "4", "--warmup", "2", "--batch", "4096",
                        "--items", "300001", "--users", "30001", "--cpu-steps", "1"], cwd=ROOT, capture_output=True, text=True,
                       timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    out = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in out, k
    assert out["n_gpus"] == 1 and out["steps"] == 4 and out["warmup"] == 2 and out["higher_is_better"] is True
    assert out["vs_baseline"] is None and out["dtype"] == "f32" and out["data"] == "synthetic" and "workload" in out["config"]
    r = out["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert "traffic" in r and r["achieved"] > 0
    c = out["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == out["unit"] and c["sample"]
    assert out["value"] > c["value"]


@pytest.mark.parametrize("workload,extra", [("sasrec", ["--batch", "512", "--hist", "50", "--num-neg", "9"]),
                                            ("neumf", ["--batch", "2048", "--items", "100001", "--users", "10001", "--num-neg", "4",
                                                       "--emb-size", "64"])])
def test_bench_other_workloads(workload, extra, cuda):
    """BASELINE configs[2] / configs[3] through the same bench.py line (no roofline object: the committed PMC
    passes are for the contract workload)"""
    import json
    import os
    import subprocess
    import sys
    from conftest import ROOT
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", workload, "--steps", "3", "--warmup", "2",
                        "--no-cpu-baseline"] + extra, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    out = json.loads([l for l in p.stdout.splitlines() if l.strip()][-1])
    assert out["metric"] == "ranked (1+K)-tuples/sec" and out["value"] > 0 and out["n_gpus"] == 1
    assert workload.lower() in out["config"]["workload"].lower() and np.isfinite(out["final_loss"])
