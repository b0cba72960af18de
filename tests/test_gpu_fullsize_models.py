"""GPU, the shapes bench.py's `secondary` legs time (BASELINE.json configs[2], [3], [4]): the fused NeuMF step at B = 65,536 on
10,000,001- and 100,000,001-row item tables, the hipGraph-replayed SASRec step at B = 4,096 / history 50 / K = 99 / 4 heads, the
DeepFM step at B = 131,072.  Unlike the contract workload's K = 99 (tests/test_gpu_fullsize.py: the oracle on a sample) these
batches are small enough for the numpy oracle to run on the WHOLE batch once the touched rows are compacted into small tables,
so every score, loss row, gradient and updated table row of the step is compared, not a sample.

Reference: src/models/general/NeuMF.py:56-76, src/models/sequential/SASRec.py:51-86, src/models/context/DeepFM.py:18-41,
src/models/BaseModel.py:175-189, 262-274, src/helpers/BaseRunner.py:187-208."""
import argparse

import numpy as np
import pytest
import torch

from conftest import assert_close, assert_update_close

# why these comparisons keep more than conftest.TOL_CAP: a dense-layer gradient at the bench's batch is a sum of 65,536 x 5 (131,072)
# signed fp32 terms, in a different order than numpy's; observed at most 0.7 of the allowance (profiles/r09_tolerances.txt)
BIG_SUM = "whole-batch fp32 sums at B = 65,536 / 131,072"

pytestmark = pytest.mark.gpu


def _zipf(n_rows, size, gen, dev):
    """bench.py's id stream: Zipf(1) ranks mapped to ids by a fixed bijection"""
    u = torch.rand(size, generator=gen, device=dev, dtype=torch.float64)
    ranks = torch.exp(u * np.log(n_rows - 1)).to(torch.int64).clamp_(1, n_rows - 1)
    return (ranks * 2654435761) % (n_rows - 1) + 1


@pytest.fixture(scope="module")
def eng(cuda):
    from rechorus_amd import engine
    return engine


# ---- NeuMF (configs[3]) -----------------------------------------------------------------------------------------------------

@pytest.mark.parametrize("n_items,n_users", [(10_000_001, 1_000_001), (100_000_001, 1_000_001)])
def test_neumf_fused_step_at_the_bench_shape_whole_batch_vs_oracle(n_items, n_users, cuda, eng):
    """NeumfTrainer (row-wise SGD) at d = 128, hidden 64, K = 4, B = 65,536 -- the `secondary.neumf` / `neumf_100M` shapes: row
    offsets beyond 2^32 bytes (and, at 100 M rows, beyond 2^32 floats x 12), the hashed plan geometry, Zipf rows with thousands
    of occurrences, the look-ahead marking passes.  The step that is CHECKED is the second one, which runs from the flags and
    the bucket plan prepared beside the first step's table updates.  oracle/neumf_oracle.py + bprmf_oracle.bpr_loss_* run on
    the whole batch over tables compacted to the touched rows; compared: the loss, every loss row, every per-tuple user
    gradient, the dense gradients, every touched row of the four tables after the step (single-occurrence rows are updated
    inside the fused kernel, the others by the plan's pair update, the hottest by its chunked path), untouched rows bit for bit."""
    from oracle import bprmf_oracle as BO
    from oracle import neumf_oracle as NO
    d, l1, B, K = 128, 64, 65_536, 4
    C = K + 1
    LR = 100.0   # SGD is linear in lr: large enough that a single negative's update is many ulps of the row it lands on
    gen = torch.Generator(device=cuda)
    gen.manual_seed(7)
    mk = lambda std, *shape: torch.empty(shape, device=cuda).normal_(0, std, generator=gen)
    P = {"mf_u": mk(0.1, n_users, d), "mlp_u": mk(0.1, n_users, d), "mf_i": mk(0.1, n_items, d), "mlp_i": mk(0.1, n_items, d),
         "W1": mk(0.15, l1, 2 * d), "b1": mk(0.1, l1), "w_out": mk(0.3, d + l1)}

    def batch():
        uid = _zipf(n_users, (B,), gen, cuda)
        pos = _zipf(n_items, (B, 1), gen, cuda)
        neg = torch.randint(1, n_items, (B, K), generator=gen, device=cuda)
        return uid.contiguous(), torch.cat([pos, neg], dim=1).contiguous()
    b1, b2 = batch(), batch()
    if n_items > 50_000_000:
        assert int(b2[1].max()) > 90_000_000, "the batch does not reach the top of the table"
    assert eng.neumf_train_step_supported(C, d, l1)
    tr = eng.NeumfTrainer(P, opt="SGD", lr=LR, l2=0.0, rowwise=True)
    tr.timing = None
    tr.step(*b1, next_batch=b2)
    ahead = getattr(tr, "_ahead", None)
    assert ahead is not None and ahead["key"] == tr._batch_key(*b2), "step 2 would not run from the prepared flags / plan"
    uid, iid = b2
    ui, inv_i = torch.unique(iid, return_inverse=True)
    uu, inv_u = torch.unique(uid, return_inverse=True)
    cnt = torch.bincount(inv_i.reshape(-1))
    assert int((cnt == 1).sum()) > 200_000 and int(cnt.max()) > 256, "the batch lacks single rows or a chunked hot row"
    snap = {k: P[k][uu if k.endswith("_u") else ui].clone() for k in ("mf_u", "mlp_u", "mf_i", "mlp_i")}
    dense0 = {k: P[k].clone() for k in ("W1", "b1", "w_out")}
    probe = torch.randint(0, n_items, (2_000_000,), generator=gen, device=cuda)
    probe = probe[~torch.isin(probe, ui)]
    probe_u = torch.randint(0, n_users, (200_000,), generator=gen, device=cuda)
    probe_u = probe_u[~torch.isin(probe_u, uu)]
    before = {k: P[k][probe_u if k.endswith("_u") else probe].clone() for k in ("mf_u", "mlp_u", "mf_i", "mlp_i")}
    loss = tr.step(*b2)
    torch.cuda.synchronize()
    assert getattr(tr, "_ahead", None) is None
    for m in tr._marks:
        assert not m[(4 * n_items + 255) // 256 * 256:].any(), "multi-occurrence flags left behind"
    for k, t in before.items():
        assert torch.equal(P[k][probe_u if k.endswith("_u") else probe], t), f"{k}: rows outside the batch moved"

    NAMES = {"mf_u": "mf_u_embeddings.weight", "mlp_u": "mlp_u_embeddings.weight", "mf_i": "mf_i_embeddings.weight",
             "mlp_i": "mlp_i_embeddings.weight"}
    Pn = {NAMES[k]: v.cpu().numpy() for k, v in snap.items()}
    Pn.update({"mlp.0.weight": dense0["W1"].cpu().numpy(), "mlp.0.bias": dense0["b1"].cpu().numpy(),
               "prediction.weight": dense0["w_out"].cpu().numpy()[None]})
    uid_c, iid_c = inv_u.cpu().numpy(), inv_i.cpu().numpy().reshape(B, C)
    pred, _ = NO.forward(Pn, uid_c, iid_c)
    rows = BO.bpr_loss_rows(pred)[0]
    gp = BO.bpr_loss_grad(pred)
    _, G = NO.backward(Pn, uid_c, iid_c, gp)
    out = tr._fused_out[1]
    tol = 3e-5
    assert_close(out["loss_vec"].cpu().numpy(), rows, what="loss rows", atol_scale=tol)
    assert_close(float(loss.item()), float(rows.astype(np.float64).mean()), rtol=1e-5, what="loss")
    assert_close(out["W1"].cpu().numpy(), G["mlp.0.weight"], what="dW1", rtol=2e-5, atol_scale=5e-5, loose=BIG_SUM)
    assert_close(out["b1"].cpu().numpy(), G["mlp.0.bias"], what="db1", rtol=2e-5, atol_scale=5e-5, loose=BIG_SUM)
    assert_close(out["w_out"].cpu().numpy(), G["prediction.weight"][0], what="dw_out", rtol=2e-5, atol_scale=5e-5, loose=BIG_SUM)
    for key, tab in (("gu_mf", "mf_u"), ("gu_mlp", "mlp_u")):
        T = np.zeros(Pn[NAMES[tab]].shape, dtype=np.float64)
        np.add.at(T, uid_c, out[key].cpu().numpy().astype(np.float64))
        assert_close(T, G[NAMES[tab]], what="user gradient rows summed per user: " + tab, atol_scale=tol)
    cnt_np = cnt.cpu().numpy()
    for tab in ("mf_i", "mlp_i", "mf_u", "mlp_u"):
        idx = uu if tab.endswith("_u") else ui
        W0 = Pn[NAMES[tab]]
        Wref = W0.copy()
        BO.opt_step_dense(Wref, G[NAMES[tab]], {}, "SGD", LR, 0.0, step=2)
        Wn = P[tab][idx].cpu().numpy()
        if tab.endswith("_i"):
            for what, sel in (("single-occurrence rows (fused kernel)", cnt_np == 1), ("rows with 2..32 occurrences (plan)", (cnt_np >= 2) & (cnt_np <= 32)),
                              ("hot rows (chunks)", cnt_np > 32)):
                assert sel.any(), what
                assert_update_close(Wn[sel], W0[sel], Wref[sel], what=f"{tab}: {what}")
                assert (np.abs(Wn[sel] - W0[sel]).max(axis=1) > 0).mean() > 0.99, f"{tab}: {what} did not move"
        else:
            assert_update_close(Wn, W0, Wref, what=tab)
    for k, name in (("W1", "mlp.0.weight"), ("b1", "mlp.0.bias"), ("w_out", "prediction.weight")):
        Wref = Pn[name].copy()
        BO.opt_step_dense(Wref, G[name], {}, "SGD", LR, 0.0, step=2)
        assert_update_close(P[k].cpu().numpy().reshape(Wref.shape), Pn[name], Wref, what=k, rtol=1e-4, extra_atol=5e-5 * LR * float(np.abs(G[name]).max()))


# ---- SASRec (configs[2]) ----------------------------------------------------------------------------------------------------

def test_sasrec_graph_replayed_step_at_the_bench_shape_whole_batch_vs_oracle(cuda, eng):
    """SasrecTrainer(graph=True, row-wise SGD) at the `secondary.sasrec` shape: B = 4,096, history_max 50, K = 99, d = 64, 4 heads, one
    block, an 8,714-row catalogue (Grocery_and_Gourmet_Food), Zipf histories.  Two eager steps, the capture, then replays; the SECOND
    REPLAY is checked against oracle/sasrec_oracle.py run on the whole batch from the parameters as they stood before it: the
    loss and every parameter after the step (item table, position table, all block parameters)."""
    from oracle import bprmf_oracle as BO
    from oracle import sasrec_oracle as SO
    from rechorus_amd import graph as hgraph
    from test_gpu_sasrec import LAYER_NAMES, _random_sasrec, to_dev
    if not hgraph.usable():
        pytest.skip("hipGraph replay not enabled in this process")
    B, L, d, n_heads, K, n_items = 4096, 50, 64, 4, 99, 8714
    LR = 0.25
    rng = np.random.default_rng(5)
    P0 = _random_sasrec(rng, n_items, d, 1, L)
    Pd = to_dev(P0, 1, cuda)
    gen = torch.Generator(device=cuda)
    gen.manual_seed(3)
    batches = []
    for _ in range(4):
        lengths = torch.randint(1, L + 1, (B,), generator=gen, device=cuda)
        hist = _zipf(n_items, (B, L), gen, cuda)
        hist = (hist * (torch.arange(L, device=cuda)[None, :] < lengths[:, None])).contiguous()
        pos = _zipf(n_items, (B, 1), gen, cuda)
        neg = torch.randint(1, n_items, (B, K), generator=gen, device=cuda)
        batches.append((hist, lengths, torch.cat([pos, neg], dim=1).contiguous()))
    tr = eng.SasrecTrainer(Pd, n_heads, opt="SGD", lr=LR, l2=0.0, rowwise=True, graph=True)
    for b in batches[:3]:
        tr.step(*b)
    torch.cuda.synchronize()
    assert len(tr._graphs) == 1, "the third step should have been captured and replayed"

    def state():
        S = {"i_embeddings.weight": Pd["item_emb"], "p_embeddings.weight": Pd["pos_emb"]}
        S.update({"transformer_block.0." + v: Pd["layers"][0][k] for k, v in LAYER_NAMES.items()})
        return {k: t.cpu().numpy().copy() for k, t in S.items()}
    before = state()
    loss = float(tr.step(*batches[3])[0])
    torch.cuda.synchronize()
    after = state()
    hist, lengths, iid = (t.cpu().numpy() for t in batches[3])
    pred = SO.forward(before, hist, lengths, iid, n_heads)
    assert_close(loss, float(BO.bpr_loss_rows(pred)[0].astype(np.float64).mean()), rtol=1e-5, what="loss of the replayed step")
    _, G = SO.backward(before, hist, lengths, iid, n_heads, BO.bpr_loss_grad(pred))
    for name, W0 in before.items():
        if name.endswith("k_linear.bias"):
            # softmax is shift-invariant along the keys: this gradient is exactly 0 in exact arithmetic, round-off in fp32
            assert float(np.abs(after[name] - W0).max()) <= LR * 1e-6
            continue
        Wref = W0.copy()
        BO.opt_step_dense(Wref, G[name], {}, "SGD", LR, 0.0, step=4)
        assert_update_close(after[name], W0, Wref, what=name, rtol=1e-4, extra_atol=5e-5 * LR * float(np.abs(G[name]).max()))
        assert not np.array_equal(after[name], W0), name


# ---- DeepFM (configs[4]) ----------------------------------------------------------------------------------------------------

def test_deepfm_step_at_the_large_bench_batch_whole_batch_vs_oracle(cuda):
    """DeepFMCTR exactly as bench.py's `secondary.deepfm_b131072` builds it (8 fields, emb_size 64, MLP [512, 64], MIND-like
    vocabularies) at B = 131,072, dropout 0 (the oracle has no mask stream for this model): probabilities, the BCE loss and every
    parameter gradient -- the 128 x 128-tile GEMMs, split-K weight gradients, the chained dX products, the tiled narrow-table sums --
    against oracle/deepfm_oracle.py on the whole batch."""
    import bench
    from oracle import deepfm_oracle as DO
    B = 131_072
    args = argparse.Namespace(emb_size=64, mlp="[512,64]", lr=5e-4, l2=0.0, opt="Adam", batch=B, pool=1, dropout=0.0)
    w = bench.DeepfmBench(args, cuda)
    (f,), = w.batches(args, cuda, seed=5)
    m = w.model
    m.train()
    m.optimizer.zero_grad()
    out = m(f)
    loss = m.loss(out)
    loss.backward()
    torch.cuda.synchronize()
    Pn = {k: v.detach().cpu().numpy() for k, v in m.state_dict().items()}
    feats = {k: f[k].cpu().numpy() for k in m.context_features}
    labels = f["label"].cpu().numpy().astype(np.float32)
    p, want_loss, G = DO.ctr_loss_and_grads(Pn, feats, labels, "DeepFM", list(m.context_features))
    assert_close(out["prediction"].detach().cpu().numpy().reshape(-1), p, what="probabilities", rtol=1e-5, atol_scale=1e-5)
    assert_close(float(loss.item()), float(want_loss), rtol=1e-5, what="BCE loss")
    scale = max(float(np.abs(v).max()) for k, v in G.items() if k.startswith("deep_layers"))
    # ReLU flips: of the 67 M hidden pre-activations of this batch a dozen lie within fp32 round-off of 0, and a unit that is on in
    # one summation order and off in the other adds or drops ONE row's dz in that unit's bias / weight gradient -- at most
    # |d loss / d logit| <= 1 / B times the absolute weights between the unit and the output.  Two flips per unit are allowed.
    lin_ids = sorted(int(n.split(".")[2]) for n in Pn if n.startswith("deep_layers.mlp.") and n.endswith(".weight"))
    reach, flip = np.abs(Pn["deep_layers.mlp.%d.weight" % lin_ids[-1]]), {}
    for idx in reversed(lin_ids[:-1]):
        flip[idx] = 2.0 * float(reach.max()) / B
        reach = reach @ np.abs(Pn["deep_layers.mlp.%d.weight" % idx])
    for name, prm in m.named_parameters():
        # dense layers at 5e-5; table rows at 1e-4 of the table's largest gradient: an element of a rarely seen row is ONE 512-long fp32
        # dot product (dh = dz W) plus the FM term, which nearly cancel in places -- summation-order noise of a different K order
        table = name.startswith("context_embedding") or name.startswith("linear_embedding")
        assert_close(prm.grad.cpu().numpy().reshape(G[name].shape), G[name], what="grad " + name, rtol=2e-5, atol_scale=1e-4 if table else 5e-5, loose=BIG_SUM,
                     abs_floor=(1e-6 * scale + (flip.get(int(name.split(".")[2]), 0.0) if name.endswith(".bias") else 0.0))
                     if name.startswith("deep_layers") else 0.0)
