"""GPU parity of the SASRec encoder kernels vs the reference's own outputs
(tests/golden/sasrec_*.npz): forward scores, every parameter gradient, two fit() iterations."""
import numpy as np
import pytest
import torch

from conftest import assert_close, assert_update_close, load_golden
from test_oracle_sasrec import CASES as ALL_CASES, DROP_CASES as ALL_DROP_CASES, params


def in_envelope(case):
    """the register-resident / MFMA encoders of csrc/sasrec.hip and sasrec_batch.hip take this golden's shape (emb_size 32 / 64, up to
    four blocks, history <= 64, or <= 128 with one block and no dropout); the other goldens exercise the shape-generic layers:
    tests/test_gpu_seq_layers.py"""
    g = load_golden(case)
    d, n_layers = int(g["meta"][1]), int(g["meta"][2])
    L = g["hist"].shape[1]
    return d in (32, 64) and n_layers <= 4 and (L <= 64 or (L <= 128 and n_layers == 1 and "p" not in g))


CASES = [c for c in ALL_CASES if in_envelope(c)]
DROP_CASES = [c for c in ALL_DROP_CASES if in_envelope(c)]

pytestmark = pytest.mark.gpu

LAYER_NAMES = {"Wq": "masked_attn_head.q_linear.weight", "bq": "masked_attn_head.q_linear.bias",
               "Wk": "masked_attn_head.k_linear.weight", "bk": "masked_attn_head.k_linear.bias",
               "Wv": "masked_attn_head.v_linear.weight", "bv": "masked_attn_head.v_linear.bias",
               "ln1w": "layer_norm1.weight", "ln1b": "layer_norm1.bias", "W1": "linear1.weight", "b1": "linear1.bias",
               "W2": "linear2.weight", "b2": "linear2.bias", "ln2w": "layer_norm2.weight", "ln2b": "layer_norm2.bias"}


def to_dev(P, n_layers, cuda):
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(cuda)
    return {"item_emb": t(P["i_embeddings.weight"]), "pos_emb": t(P["p_embeddings.weight"]),
            "layers": [{k: t(P["transformer_block.%d.%s" % (l, v)]) for k, v in LAYER_NAMES.items()}
                       for l in range(n_layers)]}


@pytest.fixture(scope="module")
def eng(cuda):
    from rechorus_amd import engine
    return engine


@pytest.mark.parametrize("impl", ["sequence", "batch"])
@pytest.mark.parametrize("case", CASES)
def test_sasrec_forward_backward(case, impl, cuda, eng):
    """both implementations of the encoder (one workgroup per sequence / batch-level row-space kernels) against
    the reference's own outputs"""
    g = load_golden(case)
    n_layers, n_heads = int(g["meta"][2]), int(g["meta"][3])
    P = to_dev(params(g), n_layers, cuda)
    hist, lengths, iid = (torch.from_numpy(g[k]).to(cuda) for k in ("hist", "len", "iid"))
    B, L = g["hist"].shape
    C, d = g["iid"].shape[1], P["item_emb"].shape[1]
    assert eng.sasrec_supported(d, n_layers, n_heads, L)
    if L > 64 and impl == "sequence":   # more than 64 positions: the batch encoder's one-row path only (one block, no dropout)
        with pytest.raises(Exception, match="not supported"):
            eng.sasrec_fwd(P["item_emb"], P["pos_emb"], P["layers"], n_heads, hist, lengths, save=True, impl=impl)
        return
    hv, xsave = eng.sasrec_fwd(P["item_emb"], P["pos_emb"], P["layers"], n_heads, hist, lengths, save=True, impl=impl)
    rows = torch.arange(B, device=cuda)
    pred = eng.gather_dot(hv, P["item_emb"], rows, iid)
    assert_close(pred.cpu().numpy(), g["pred"], what="pred")      # north star: 1e-5 relative fp32
    hv2, _ = eng.sasrec_fwd(P["item_emb"], P["pos_emb"], P["layers"], n_heads, hist, lengths, save=False, impl=impl)
    assert torch.equal(hv, hv2)

    gpred = torch.from_numpy(g["gpred"]).to(cuda)
    dhv = eng.weighted_row_sum(P["item_emb"], iid, gpred)
    g_hist, dg = eng.sasrec_bwd(P["layers"], n_heads, lengths, xsave, dhv)
    G = params(g, "G/")
    floor = 1e-6 * max(float(np.abs(v).max()) for v in G.values())
    for l in range(n_layers):
        for k, name in LAYER_NAMES.items():
            assert_close(dg[l][k].cpu().numpy(), G["transformer_block.%d.%s" % (l, name)], what=f"layer {l} d{k}",
                         rtol=2e-5, atol_scale=5e-5, abs_floor=floor)
    ids = torch.cat([iid.reshape(-1), hist.reshape(-1)])
    keys, perm = eng.sort_ids(ids, P["item_emb"].shape[0])
    GI = torch.zeros_like(P["item_emb"])
    eng.segmented_update2(keys, perm, hv, g_hist.view(-1, d), B * C, coef=gpred.reshape(-1), div=C, dense_grad=GI)
    assert_close(GI.cpu().numpy(), G["i_embeddings.weight"], what="d item_emb", rtol=2e-5, atol_scale=5e-5)
    valid = (hist > 0).to(torch.int64)
    position = ((lengths[:, None] - torch.arange(L, device=cuda)[None, :]) * valid).contiguous()
    GP = eng.embedding_dense_backward(g_hist, position, P["pos_emb"].shape[0])
    assert_close(GP.cpu().numpy(), G["p_embeddings.weight"], what="d pos_emb", rtol=2e-5, atol_scale=5e-5)
    GP2 = eng.sasrec_pos_grad(g_hist, lengths, P["pos_emb"].shape[0])  # the dedicated kernel the trainers use
    assert_close(GP2.cpu().numpy(), G["p_embeddings.weight"], what="d pos_emb (rc_sasrec_pos_grad)", rtol=2e-5, atol_scale=5e-5)


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("tag,opt", [("SGD_l20.001", "SGD"), ("Adam_l20.0001", "Adam")])
def test_sasrec_two_fit_iterations_match_reference(case, tag, opt, cuda, eng):
    g = load_golden(case)
    lr, l2 = (float(x) for x in g[tag + "_hyper"])
    n_layers, n_heads = int(g["meta"][2]), int(g["meta"][3])
    P0 = params(g)
    P = to_dev(P0, n_layers, cuda)
    tr = eng.SasrecTrainer(P, n_heads, opt=opt, lr=lr, l2=l2, rowwise=False)
    for step, sfx in enumerate(("", "2"), 1):
        hist, lengths, iid = (torch.from_numpy(g[k + sfx]).to(cuda) for k in ("hist", "len", "iid"))
        loss = tr.step(hist, lengths, iid)
        assert_close(loss.cpu().numpy()[0], g[tag + "_losses"][step - 1], what=f"loss {step}", rtol=1e-5)
    want = params(g, tag + "/")
    ex = 2e-3 * lr if opt == "Adam" else 0.0
    G1 = params(g, "G/")     # the reference's autograd gradients of the first batch
    checks = [("item_emb", "i_embeddings.weight", P["item_emb"]), ("pos_emb", "p_embeddings.weight", P["pos_emb"])]
    for l in range(n_layers):
        checks += [(f"L{l}.{k}", "transformer_block.%d.%s" % (l, v), P["layers"][l][k]) for k, v in LAYER_NAMES.items()]
    n_ill = n_all = 0
    for what, name, t in checks:
        if opt == "Adam" and name.endswith("k_linear.bias"):
            # d/d(key bias) is exactly 0 in exact arithmetic (softmax is shift invariant along the
            # keys); in fp32 it is ~1e-10 round-off whose SIGN Adam turns into +-lr steps -- in the
            # reference as well.  Not a comparable quantity; bounded by lr instead.
            assert float(np.abs(t.cpu().numpy() - P0[name]).max()) <= 2.5 * lr
            continue
        if opt == "Adam":
            # Adam normalises a gradient to a step of about lr whatever its size: an element whose first gradient is summation-order
            # noise (|g| < 1e-7, against eps = 1e-8) is not a comparable quantity -- those elements are LISTED and excluded, they
            # must be few, and every other element is held to the tolerance without an outlier allowance
            decay = 0.0 if "bias" in name else l2      # (the reference's 'bias' group has no weight decay, BaseModel.py:64-73)
            ill = np.abs(G1[name] + decay * P0[name]) < 1e-7
            n_ill += int((ill & (np.abs(want[name] - P0[name]) > 0)).sum())
            n_all += ill.size
            assert_update_close(t.cpu().numpy(), P0[name], want[name], what=what, extra_atol=ex, rtol=2e-4, exclude=ill, strict=True)
        else:
            assert_update_close(t.cpu().numpy(), P0[name], want[name], what=what, rtol=2e-4)
    # (the excluded elements that actually moved -- mostly biases of hidden units that a single row barely activates -- stay a
    #  fraction of a per cent of the model)
    assert n_ill <= 0.005 * max(n_all, 1), f"{n_ill} of {n_all} elements excluded as ill-conditioned"


def _random_sasrec(rng, n_items, d, n_layers, L):
    P = {"i_embeddings.weight": rng.normal(0, 0.3, (n_items, d)), "p_embeddings.weight": rng.normal(0, 0.3, (L + 1, d))}
    for l in range(n_layers):
        pre = "transformer_block.%d." % l
        for nm in ("masked_attn_head.q_linear", "masked_attn_head.k_linear", "masked_attn_head.v_linear", "linear1", "linear2"):
            P[pre + nm + ".weight"] = rng.normal(0, 0.15, (d, d))
            P[pre + nm + ".bias"] = rng.normal(0, 0.1, d)
        for nm in ("layer_norm1", "layer_norm2"):
            P[pre + nm + ".weight"] = 1 + rng.normal(0, 0.1, d)
            P[pre + nm + ".bias"] = rng.normal(0, 0.1, d)
    return {k: v.astype(np.float32) for k, v in P.items()}


@pytest.mark.parametrize("impl", ["sequence", "batch"])
def test_sasrec_length_buckets_vs_oracle(impl, cuda, eng):
    """history_max > 32: the batch is split on the device into sequences of <= 32 items (32-row LDS geometry)
    and longer ones (64 rows), two launches per pass.  700 sequences (three rounds of the compaction kernel)
    with lengths on both sides of the boundary vs the oracle; the short ones must equal, bit for bit, what a
    history_max = 32 call computes for them."""
    from oracle import sasrec_oracle as SO
    rng = np.random.default_rng(17)
    B, L, d, n_layers, n_heads, C, n_items = 700, 50, 64, 2, 2, 3, 500
    P = _random_sasrec(rng, n_items, d, n_layers, L)
    lengths = rng.integers(1, L + 1, size=B).astype(np.int64)
    lengths[:6] = (32, 33, 1, 50, 31, 34)
    hist = rng.integers(1, n_items, size=(B, L)).astype(np.int64) * (np.arange(L)[None, :] < lengths[:, None])
    iid = rng.integers(1, n_items, size=(B, C)).astype(np.int64)
    gpred = rng.normal(size=(B, C)).astype(np.float32)
    Pd = to_dev(P, n_layers, cuda)
    h_d, l_d, i_d = (torch.from_numpy(x).to(cuda) for x in (hist, lengths, iid))
    hv, xsave = eng.sasrec_fwd(Pd["item_emb"], Pd["pos_emb"], Pd["layers"], n_heads, h_d, l_d, save=True, impl=impl)
    want_pred, cache = SO.forward(P, hist, lengths, iid, n_heads, keep=True)
    assert_close(hv.cpu().numpy(), cache["hv"], what="hv", rtol=2e-5, atol_scale=3e-5)
    gp_d = torch.from_numpy(gpred).to(cuda)
    dhv = eng.weighted_row_sum(Pd["item_emb"], i_d, gp_d)
    g_hist, dg = eng.sasrec_bwd(Pd["layers"], n_heads, l_d, xsave, dhv)
    _, G = SO.backward(P, hist, lengths, iid, n_heads, gpred)
    floor = 1e-6 * max(float(np.abs(v).max()) for v in G.values())
    for l in range(n_layers):
        for k, name in LAYER_NAMES.items():
            assert_close(dg[l][k].cpu().numpy(), G["transformer_block.%d.%s" % (l, name)], what=f"layer {l} d{k}",
                         rtol=3e-5, atol_scale=1e-4, abs_floor=floor)
    valid = (h_d > 0).to(torch.int64)
    position = ((l_d[:, None] - torch.arange(L, device=cuda)[None, :]) * valid).contiguous()
    GP = eng.embedding_dense_backward(g_hist, position, Pd["pos_emb"].shape[0])
    assert_close(GP.cpu().numpy(), G["p_embeddings.weight"], what="d pos_emb", rtol=3e-5, atol_scale=1e-4)
    # the short bucket runs the same geometry as a history_max = 32 call
    short = np.nonzero(lengths <= 32)[0]
    assert 0 < len(short) < B
    hs = torch.from_numpy(np.ascontiguousarray(hist[short][:, :32])).to(cuda)
    ls = torch.from_numpy(lengths[short]).to(cuda)
    hv32, _ = eng.sasrec_fwd(Pd["item_emb"], Pd["pos_emb"], Pd["layers"], n_heads, hs, ls, impl=impl)
    if impl == "sequence":  # (the batch kernels tile the row space, so a row's tile-mates differ between the calls)
        assert torch.equal(hv32, hv[torch.from_numpy(short).to(cuda)])
    else:
        assert_close(hv32.cpu().numpy(), hv[torch.from_numpy(short).to(cuda)].cpu().numpy(), what="hv32", rtol=1e-6, atol_scale=1e-6)
    # deterministic
    hv_b, xs_b = eng.sasrec_fwd(Pd["item_emb"], Pd["pos_emb"], Pd["layers"], n_heads, h_d, l_d, save=True, impl=impl)
    g_hist_b, dg_b = eng.sasrec_bwd(Pd["layers"], n_heads, l_d, xs_b, dhv)
    assert torch.equal(hv, hv_b) and torch.equal(g_hist, g_hist_b)
    assert all(torch.equal(dg[l][k], dg_b[l][k]) for l in range(n_layers) for k in LAYER_NAMES)


@pytest.mark.parametrize("d,n_heads,L,B", [(64, 4, 128, 90), (64, 1, 65, 33), (32, 2, 100, 140), (32, 4, 128, 21)])
def test_sasrec_more_than_64_positions_vs_oracle(d, n_heads, L, B, cuda, eng, monkeypatch):
    """64 < history_max <= 128 with ONE block and no dropout: the batch encoder's one-row path with a 128-row tile (two keys per
    lane) is the whole encoder.  Output, history gradient (incl. zero rows past the length), position-table and parameter
    gradients against the numpy oracle, with empty, single-item, 64 / 65-item and full histories; other routes refuse the shape."""
    from oracle import sasrec_oracle as SO
    rng = np.random.default_rng(L * 7 + d + n_heads)
    n_layers, C, n_items = 1, 4, 300
    assert eng.sasrec_supported(d, n_layers, n_heads, L) and not eng.sasrec_supported(d, 2, n_heads, L)
    assert not eng.sasrec_supported(d, n_layers, n_heads, L, dropout=0.2) and not eng.sasrec_supported(d, n_layers, n_heads, 129)
    P = _random_sasrec(rng, n_items, d, n_layers, L)
    lengths = rng.integers(1, L + 1, size=B).astype(np.int64)
    lengths[:6] = (L, 0, 1, 64, 65, L - 1)
    hist = rng.integers(1, n_items, size=(B, L)).astype(np.int64) * (np.arange(L)[None, :] < lengths[:, None])
    iid = rng.integers(1, n_items, size=(B, C)).astype(np.int64)
    gpred = rng.normal(size=(B, C)).astype(np.float32)
    Pd = to_dev(P, n_layers, cuda)
    h_d, l_d, i_d = (torch.from_numpy(x).to(cuda) for x in (hist, lengths, iid))
    hv, xsave = eng.sasrec_fwd(Pd["item_emb"], Pd["pos_emb"], Pd["layers"], n_heads, h_d, l_d, save=True)
    assert xsave.impl == "batch"
    live = lengths > 0   # (the oracle, like the reference, reads position -1 of an all-padding row; the engine defines 0)
    _, cache = SO.forward(P, hist[live], lengths[live], iid[live], n_heads, keep=True)
    assert_close(hv.cpu().numpy()[live], cache["hv"], what="hv", rtol=2e-5, atol_scale=3e-5)
    assert np.all(hv.cpu().numpy()[~live] == 0)
    gp_d = torch.from_numpy(gpred).to(cuda)
    dhv = eng.weighted_row_sum(Pd["item_emb"], i_d, gp_d)
    g_hist, dg = eng.sasrec_bwd(Pd["layers"], n_heads, l_d, xsave, dhv)
    _, G = SO.backward(P, hist[live], lengths[live], iid[live], n_heads, gpred[live])
    floor = 1e-6 * max(float(np.abs(v).max()) for v in G.values())
    for k, name in LAYER_NAMES.items():
        assert_close(dg[0][k].cpu().numpy(), G["transformer_block.0.%s" % name], what=f"d{k}", rtol=3e-5, atol_scale=1e-4, abs_floor=floor)
    gh = g_hist.cpu().numpy()
    assert np.all(gh[np.arange(L)[None, :] >= lengths[:, None]] == 0)
    valid = (h_d > 0).to(torch.int64)
    position = ((l_d[:, None] - torch.arange(L, device=cuda)[None, :]) * valid).contiguous()
    GP = eng.embedding_dense_backward(g_hist, position, Pd["pos_emb"].shape[0])
    assert_close(GP.cpu().numpy(), G["p_embeddings.weight"], what="d pos_emb", rtol=3e-5, atol_scale=1e-4)
    GI = eng.embedding_dense_backward(g_hist, h_d, n_items)
    want_items = G["i_embeddings.weight"].copy()
    np.subtract.at(want_items, iid[live].reshape(-1), (gpred[live][:, :, None] * cache["hv"][:, None, :]).reshape(-1, d))   # minus the candidates' part
    assert_close(GI.cpu().numpy()[1:], want_items[1:], what="d item_emb (history part)", rtol=3e-5, atol_scale=1e-4)
    hv_b, xs_b = eng.sasrec_fwd(Pd["item_emb"], Pd["pos_emb"], Pd["layers"], n_heads, h_d, l_d, save=True)
    g_hist_b, _ = eng.sasrec_bwd(Pd["layers"], n_heads, l_d, xs_b, dhv)
    assert torch.equal(hv, hv_b) and torch.equal(g_hist, g_hist_b)
    for bad in ({"impl": "sequence"}, {"drop_p": 0.3, "seed": torch.zeros(1, dtype=torch.int64, device=cuda)}):
        with pytest.raises(Exception):
            eng.sasrec_fwd(Pd["item_emb"], Pd["pos_emb"], Pd["layers"], n_heads, h_d, l_d, save=True, **bad)
    monkeypatch.setenv("RC_SAS_LAST_ROW", "0")
    with pytest.raises(Exception, match="one-row path"):
        eng.sasrec_fwd(Pd["item_emb"], Pd["pos_emb"], Pd["layers"], n_heads, h_d, l_d, save=True, impl="batch")


def test_sasrec_batch_kernels_edge_shapes_vs_sequence_kernels(cuda, eng):
    """d = 32 and 64, 1..3 layers, empty histories, a single row, history_max = 20 (one attention geometry) and 64"""
    rng = np.random.default_rng(23)
    for d, n_layers, n_heads, L, B in ((32, 1, 4, 20, 70), (64, 3, 1, 64, 37), (32, 2, 2, 50, 300), (64, 1, 4, 7, 1)):
        n_items = 200
        P = _random_sasrec(rng, n_items, d, n_layers, L)
        lengths = rng.integers(0, L + 1, size=B).astype(np.int64)
        lengths[0] = L
        if B > 3:
            lengths[1], lengths[2] = 0, 1
        hist = rng.integers(1, n_items, size=(B, L)).astype(np.int64) * (np.arange(L)[None, :] < lengths[:, None])
        Pd = to_dev(P, n_layers, cuda)
        h_d, l_d = torch.from_numpy(hist).to(cuda), torch.from_numpy(lengths).to(cuda)
        dhv = torch.from_numpy(rng.normal(size=(B, d)).astype(np.float32)).to(cuda)
        out = {}
        for impl in ("sequence", "batch"):
            hv, saved = eng.sasrec_fwd(Pd["item_emb"], Pd["pos_emb"], Pd["layers"], n_heads, h_d, l_d, save=True, impl=impl)
            g_hist, dg = eng.sasrec_bwd(Pd["layers"], n_heads, l_d, saved, dhv)
            out[impl] = (hv.cpu().numpy(), g_hist.cpu().numpy(), [{k: v.cpu().numpy() for k, v in g.items()} for g in dg])
        what = f"d={d} layers={n_layers} heads={n_heads} L={L} B={B}"
        assert_close(out["batch"][0], out["sequence"][0], what=what + " hv", rtol=2e-5, atol_scale=2e-5)
        assert_close(out["batch"][1], out["sequence"][1], what=what + " g_hist", rtol=5e-5, atol_scale=5e-5)
        floor = 1e-6 * max(float(np.abs(v).max()) for g in out["sequence"][2] for v in g.values())
        for l in range(n_layers):
            for k in LAYER_NAMES:
                assert_close(out["batch"][2][l][k], out["sequence"][2][l][k], what=f"{what} layer {l} d{k}", rtol=5e-5,
                             atol_scale=1e-4, abs_floor=floor)
        assert np.all(out["batch"][0][lengths == 0] == 0) and np.all(out["batch"][1][lengths == 0] == 0)


@pytest.mark.parametrize("d,n_layers,n_heads,L,B,fits", [
    (64, 2, 4, 64, 260, True), (64, 1, 4, 50, 300, True), (64, 1, 2, 32, 90, True), (64, 1, 1, 16, 90, True),
    (32, 2, 2, 50, 130, True), (32, 1, 1, 20, 70, True), (64, 1, 4, 9, 5, True),
    (64, 1, 2, 50, 130, False), (32, 1, 4, 20, 70, False)])
def test_sasrec_register_attention_equals_lds_attention(d, n_layers, n_heads, L, B, fits, cuda, eng, monkeypatch):
    """the register-resident attention (one wave per (sequence, head), 16 x 16 x 4 MFMA tiles, sas_attn_reg.hpp) against the LDS-tile
    kernels of rounds 1-3 (RC_SAS_REG_ATTN=0) on every tile count 1..4, lengths on both sides of each 16-row boundary, empty
    histories; shapes outside its envelope (d_k = 8, or too many operand tiles) must take the LDS kernels either way"""
    monkeypatch.setenv("RC_SAS_LAST_ROW", "0")   # every layer on all rows: the attention kernels under test serve the last layer too
    rng = np.random.default_rng(1000 * d + 10 * L + n_heads)
    n_items = 300
    P = _random_sasrec(rng, n_items, d, n_layers, L)
    lengths = rng.integers(0, L + 1, size=B).astype(np.int64)
    marks = [x for x in (L, 0, 1, 15, 16, 17, 31, 32, 33, 47, 48, 49, 63, 64) if x <= L][:B]
    lengths[:len(marks)] = marks
    hist = rng.integers(1, n_items, size=(B, L)).astype(np.int64) * (np.arange(L)[None, :] < lengths[:, None])
    Pd = to_dev(P, n_layers, cuda)
    h_d, l_d = torch.from_numpy(hist).to(cuda), torch.from_numpy(lengths).to(cuda)
    dhv = torch.from_numpy(rng.normal(size=(B, d)).astype(np.float32)).to(cuda)
    out = {}
    for mode in ("1", "0", "1"):
        monkeypatch.setenv("RC_SAS_REG_ATTN", mode)
        hv, saved = eng.sasrec_fwd(Pd["item_emb"], Pd["pos_emb"], Pd["layers"], n_heads, h_d, l_d, save=True, impl="batch")
        g_hist, dg = eng.sasrec_bwd(Pd["layers"], n_heads, l_d, saved, dhv)
        torch.cuda.synchronize()
        res = (hv.cpu().numpy(), g_hist.cpu().numpy(), [{k: v.cpu().numpy() for k, v in g.items()} for g in dg])
        if mode in out:   # bit-reproducible
            assert np.array_equal(res[0], out[mode][0]) and np.array_equal(res[1], out[mode][1])
            assert all(np.array_equal(res[2][l][k], out[mode][2][l][k]) for l in range(n_layers) for k in LAYER_NAMES)
        out[mode] = res
    what = f"d={d} layers={n_layers} heads={n_heads} L={L} B={B}"
    if not fits:
        assert np.array_equal(out["1"][0], out["0"][0]) and np.array_equal(out["1"][1], out["0"][1]), what
        return
    assert not np.array_equal(out["1"][1], out["0"][1]), what + ": the switch had no effect"
    assert_close(out["1"][0], out["0"][0], what=what + " hv", rtol=2e-5, atol_scale=2e-5)
    assert_close(out["1"][1], out["0"][1], what=what + " g_hist", rtol=5e-5, atol_scale=5e-5)
    floor = 1e-6 * max(float(np.abs(v).max()) for g in out["0"][2] for v in g.values())
    for l in range(n_layers):
        for k in LAYER_NAMES:
            assert_close(out["1"][2][l][k], out["0"][2][l][k], what=f"{what} layer {l} d{k}", rtol=5e-5, atol_scale=1e-4,
                         abs_floor=floor)
    assert np.all(out["1"][0][lengths == 0] == 0) and np.all(out["1"][1][lengths == 0] == 0)


@pytest.mark.parametrize("d,n_rows,n_a,C,n_b,opt", [(64, 300, 4000, 5, 9000, None), (64, 97, 2000, 4, 1500, "Adam"),
                                                      (16, 50, 600, 3, 700, "SGD"), (128, 40, 900, 2, 0, "Adagrad"),
                                                      (32, 1000, 3000, 6, 5000, None)])
def test_segmented_update_rows_equals_head_list_route(d, n_rows, n_a, C, n_b, opt, cuda, eng, monkeypatch):
    """rc_segmented_update_rows (one wave per table row, every row collecting many occurrences) against a float64 sum
    and against rc_segmented_update2's head list: coef * src[o / C] occurrences followed by plain src2 rows, rows that
    never occur, a row past the 192-occurrence hand-over to the chunked path, dense-gradient and optimizer outputs"""
    rng = np.random.default_rng(d + n_rows)
    ids_a = rng.integers(0, n_rows, size=(n_a, C)).astype(np.int64)
    ids_a[ids_a == 3] = 4                       # row 3 never occurs
    ids_b = rng.integers(0, n_rows, size=n_b).astype(np.int64)
    ids_b[ids_b == 3] = 5
    ids_b[: n_b // 3] = 7                       # a hot row (chunked path)
    coef = rng.normal(size=(n_a, C)).astype(np.float32)
    src = rng.normal(size=(n_a, d)).astype(np.float32)
    src2 = rng.normal(size=(max(n_b, 1), d)).astype(np.float32)
    ids = np.concatenate([ids_a.reshape(-1), ids_b])
    want = np.zeros((n_rows, d), np.float64)
    np.add.at(want, ids_a.reshape(-1), (coef.reshape(-1, 1).astype(np.float64) * np.repeat(src, C, axis=0)))
    if n_b:
        np.add.at(want, ids_b, src2[:n_b].astype(np.float64))
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(cuda)
    keys, perm = eng.sort_ids(t(ids), n_rows)
    W0 = rng.normal(size=(n_rows, d)).astype(np.float32)
    out = {}
    for mode in ("1", "0", "1"):
        monkeypatch.setattr(eng, "_SEG_ROWS", mode == "1")
        if opt is None:
            G = torch.zeros((n_rows, d), device=cuda)
            eng.segmented_update2(keys, perm, t(src), t(src2), n_a * C, coef=t(coef).reshape(-1), div=C, dense_grad=G)
            res = (G.cpu().numpy(),)
        else:
            W, m, v = t(W0), torch.zeros((n_rows, d), device=cuda), torch.zeros((n_rows, d), device=cuda)
            h = eng.make_hyper(opt, lr=0.01, l2=1e-4, step=3)
            eng.segmented_update2(keys, perm, t(src), t(src2), n_a * C, hyper=h, W=W, m=m if opt != "SGD" else None,
                                  v=v if opt == "Adam" else None, coef=t(coef).reshape(-1), div=C)
            res = (W.cpu().numpy(), m.cpu().numpy(), v.cpu().numpy())
        if mode in out:
            assert all(np.array_equal(a, b) for a, b in zip(res, out[mode])), "not reproducible"
        out[mode] = res
    if opt is None:
        scale = np.abs(want).max()
        assert np.abs(out["1"][0] - want).max() <= 2e-6 * scale * np.sqrt(len(ids) / n_rows)
        assert np.all(out["1"][0][3] == 0)
    else:
        assert np.array_equal(out["1"][0][3], W0[3])    # a row without occurrences does not move (row-wise update)
        assert not np.array_equal(out["1"][0][7], W0[7])
    for a, b in zip(out["1"], out["0"]):
        assert_close(a, b, what=f"rows route vs head list (d={d}, opt={opt})", rtol=2e-5, atol_scale=2e-5)


@pytest.mark.parametrize("d,n_rows,B,C,L,opt", [(64, 300, 700, 5, 12, None), (64, 97, 500, 4, 3, "Adam"), (16, 50, 200, 3, 4, "SGD"),
                                                 (128, 40, 450, 2, 0, "Adagrad"), (32, 1000, 600, 6, 9, None),
                                                 (64, 8714, 1100, 100, 50, "SGD"), (64, 12288, 2100, 64, 7, None)])
def test_rows_plan_equals_the_sorted_rows_route(d, n_rows, B, C, L, opt, cuda, eng, monkeypatch):
    """rc_rows_plan_build / rc_rows_plan_update (row bounds from a counting sort of the id tensors themselves) against the route it
    replaces: the grouping is the stable sort's (keys ascending, occurrence numbers ascending inside a row; padding slots of the
    history windows left out except the batch's first, which closes row 0's list), and the update is bit-identical to
    rc_segmented_update_rows on the radix-sorted keys.  Hot rows (chunked path), rows that never occur, no history at all (L = 0),
    the table-size limit (12,288 rows), several tiles of 4,096 occurrences, ids outside the table (counted, no part)."""
    rng = np.random.default_rng(d + n_rows + B)
    ids_a = rng.integers(1, n_rows, size=(B, C)).astype(np.int64)
    ids_a[ids_a == 3] = 4                       # row 3 never occurs
    ids_a[: B // 3, 0] = 7                      # a hot row (chunked path)
    lengths = rng.integers(0, L + 1, size=B).astype(np.int64)
    hist = rng.integers(1, n_rows, size=(B, max(L, 1))).astype(np.int64)[:, :L]
    hist[hist == 3] = 5
    hist = hist * (np.arange(L)[None, :] < lengths[:, None])
    n_a, n_b = B * C, B * L
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(cuda)
    assert eng.rows_plan_supported(n_rows, n_a + n_b, d)
    rp = eng.RowsPlan(t(ids_a), t(hist) if L else None, t(lengths) if L else None, n_rows, d, tag=f"test_rows_plan_{d}_{n_rows}")
    keys, perm, start, end, status = (x.cpu().numpy().astype(np.int64) for x in rp.views())
    # the grouping
    pad = (np.arange(L)[None, :] >= lengths[:, None]).reshape(-1)
    occ_key = np.concatenate([ids_a.reshape(-1), hist.reshape(-1)])
    live = np.concatenate([np.ones(n_a, bool), ~pad])
    order = np.argsort(occ_key[live], kind="stable")
    want_perm = np.nonzero(live)[0][order]
    want_keys = occ_key[live][order]
    if pad.any():                               # the first padding slot closes row 0's list
        first = n_a + int(np.argmax(pad))
        n0 = int((want_keys == 0).sum())
        want_perm = np.concatenate([want_perm[:n0], [first], want_perm[n0:]])
        want_keys = np.concatenate([want_keys[:n0], [0], want_keys[n0:]])
    n_live = len(want_perm)
    assert np.array_equal(perm[:n_live], want_perm)
    cnt = np.bincount(want_keys, minlength=n_rows)
    assert np.array_equal(end - start, cnt) and np.array_equal(start, np.cumsum(cnt) - cnt) and status[0] == 0
    hot = np.nonzero(cnt > 192)[0]
    assert (len(hot) or B * C < 1500) and np.array_equal(keys[start[hot]], hot)   # (all the route keeps of the keys: the hot rows' numbers)
    # the update, against rc_segmented_update_rows on the radix-sorted keys of the same occurrences (padding parked behind the table)
    coef = rng.normal(size=(B, C)).astype(np.float32)
    src = rng.normal(size=(B, d)).astype(np.float32)
    src2 = rng.normal(size=(max(n_b, 1), d)).astype(np.float32)
    src2[:n_b][pad] = 0                         # the encoder's backward leaves zero rows at the padding slots
    parked = occ_key.copy()
    parked[~live] = n_rows
    if pad.any():
        parked[first] = 0
    skeys, sperm = eng.sort_ids(t(parked), n_rows + 1)
    monkeypatch.setattr(eng, "_SEG_ROWS_MIN_PER_ROW", 0)
    W0 = rng.normal(size=(n_rows, d)).astype(np.float32)
    res = {}
    for route in ("plan", "sorted", "plan"):
        if opt is None:
            G = torch.zeros((n_rows, d), device=cuda)
            if route == "plan":
                rp.update(t(src), coef=t(coef).reshape(-1), div=C, src2=t(src2), dense_grad=G)
            else:
                eng.segmented_update2(skeys, sperm, t(src), t(src2), n_a, coef=t(coef).reshape(-1), div=C, dense_grad=G)
            got = (G.cpu().numpy(),)
        else:
            W, m, v = t(W0), torch.zeros((n_rows, d), device=cuda), torch.zeros((n_rows, d), device=cuda)
            h = eng.make_hyper(opt, lr=0.01, l2=1e-4, step=3)
            kw = dict(hyper=h, W=W, m=m if opt != "SGD" else None, v=v if opt == "Adam" else None, coef=t(coef).reshape(-1), div=C)
            if route == "plan":
                rp.update(t(src), src2=t(src2), **kw)
            else:
                eng.segmented_update2(skeys, sperm, t(src), t(src2), n_a, **kw)
            got = (W.cpu().numpy(), m.cpu().numpy(), v.cpu().numpy())
        if route in res:
            assert all(np.array_equal(a, b) for a, b in zip(got, res[route])), "not reproducible"
        res[route] = got
    for a, b in zip(res["plan"], res["sorted"]):
        assert np.array_equal(a, b), f"rows plan vs sorted rows route (d={d}, opt={opt})"
    if opt is not None:
        assert np.array_equal(res["plan"][0][3], W0[3]) and not np.array_equal(res["plan"][0][7], W0[7])
    # ids outside the table take no part and are counted
    bad = ids_a.copy()
    bad[0, 0], bad[1, 1] = n_rows, -2
    rp2 = eng.RowsPlan(t(bad), t(hist) if L else None, t(lengths) if L else None, n_rows, d, tag=f"test_rows_plan_bad_{d}_{n_rows}")
    k2, p2, s2, e2, st2 = (x.cpu().numpy().astype(np.int64) for x in rp2.views())
    assert st2[0] == 2 and int((e2 - s2).sum()) == n_live - 2


@pytest.mark.parametrize("opt,rowwise", [("SGD", True), ("Adam", True), ("Adam", False)])
def test_sasrec_trainer_rows_plan_equals_sorted_route(opt, rowwise, cuda, eng, monkeypatch):
    """SasrecTrainer with the counting-sort row bounds (default) against RC_SAS_ROWS_PLAN=0 (radix sort + glue): three steps leave
    every parameter bit-identical, on one stream and on two"""
    rng = np.random.default_rng(11)
    B, L, d, n_layers, n_heads, C, n_items = 600, 50, 64, 1, 4, 20, 400
    P = _random_sasrec(rng, n_items, d, n_layers, L)
    batches = []
    for _ in range(3):
        lengths = rng.integers(1, L + 1, size=B).astype(np.int64)
        hist = rng.integers(1, n_items, size=(B, L)).astype(np.int64) * (np.arange(L)[None, :] < lengths[:, None])
        iid = rng.integers(1, n_items, size=(B, C)).astype(np.int64)
        batches.append(tuple(torch.from_numpy(x).to(cuda) for x in (hist, lengths, iid)))
    out = {}
    for overlap in (True, False):
        for plan in (True, False):
            monkeypatch.setattr(eng, "_SAS_ROWS_PLAN", plan)
            monkeypatch.setattr(eng, "_SAS_OVERLAP", overlap)
            monkeypatch.setattr(eng, "_SAS_OVERLAP_MIN", 0)
            Pd = to_dev(P, n_layers, cuda)
            tr = eng.SasrecTrainer(Pd, n_heads, opt=opt, lr=1e-3, l2=1e-5, rowwise=rowwise)
            losses = [float(tr.step(*b)[0]) for b in batches]
            torch.cuda.synchronize()
            out[(overlap, plan)] = (losses, Pd["item_emb"].cpu().numpy(), Pd["pos_emb"].cpu().numpy())
    ref = out[(False, False)]
    assert not np.array_equal(ref[1], P["i_embeddings.weight"])
    for k, got in out.items():
        assert got[0] == ref[0] and np.array_equal(got[1], ref[1]) and np.array_equal(got[2], ref[2]), k


@pytest.mark.parametrize("opt,overlap", [("SGD", True), ("Adagrad", False), ("Adam", True)])
def test_sasrec_trainer_graph_replay_equals_eager(opt, overlap, cuda, eng, monkeypatch):
    """SasrecTrainer(graph=True) replays the step from a hipGraph (both streams captured, the batch copied into static
    buffers): seven steps over different batches -- two eager, the capture, four replays -- leave the loss sequence and every
    parameter bit-identical to the eager trainer.  Adam: the replayed trainer keeps the step count in device memory and forms the
    bias corrections in the kernels (rc_segmented_update_rows_dev, rc_dense_update_multi_dev) -- the same double-precision
    expressions as the host's, compared to rounding."""
    from rechorus_amd import graph as hgraph
    if not hgraph.usable():
        pytest.skip("hipGraph replay not enabled in this process")
    rng = np.random.default_rng(17)
    B, L, d, n_layers, n_heads, C, n_items = 500, 50, 64, 1, 4, 20, 400
    P = _random_sasrec(rng, n_items, d, n_layers, L)
    batches = []
    for _ in range(7):
        lengths = rng.integers(0, L + 1, size=B).astype(np.int64)
        hist = rng.integers(1, n_items, size=(B, L)).astype(np.int64) * (np.arange(L)[None, :] < lengths[:, None])
        iid = rng.integers(1, n_items, size=(B, C)).astype(np.int64)
        batches.append(tuple(torch.from_numpy(x).to(cuda) for x in (hist, lengths, iid)))
    monkeypatch.setattr(eng, "_SAS_OVERLAP", overlap)
    monkeypatch.setattr(eng, "_SAS_OVERLAP_MIN", 0)
    out = {}
    for graph in (False, True):
        Pd = to_dev(P, n_layers, cuda)
        tr = eng.SasrecTrainer(Pd, n_heads, opt=opt, lr=1e-2, l2=1e-5, rowwise=True, graph=graph)
        losses = [float(tr.step(*b)[0]) for b in batches]
        torch.cuda.synchronize()
        assert (len(tr._graphs) == 1) == graph
        out[graph] = (losses, Pd["item_emb"].cpu().numpy(), Pd["pos_emb"].cpu().numpy(),
                      [{k: v.cpu().numpy() for k, v in lay.items()} for lay in Pd["layers"]])
    assert len(set(out[True][0])) == len(batches)
    if opt == "Adam":
        assert np.allclose(out[True][0], out[False][0], rtol=1e-5, atol=0)
        assert_close(out[True][1], out[False][1], what="item table", rtol=1e-5, atol_scale=1e-6)
        assert_close(out[True][2], out[False][2], what="position table", rtol=1e-5, atol_scale=1e-6)
        for l in range(n_layers):
            for k in LAYER_NAMES:
                if k == "bk":
                    # the key bias does not reach the output (a per-query constant under the softmax): its gradient is rounding
                    # noise, which Adam normalises to steps of size lr in either direction -- bounded, not reproducible to rounding
                    assert float(np.abs(out[True][3][l][k] - out[False][3][l][k]).max()) <= 2 * 1e-2 * len(batches)
                    continue
                assert_close(out[True][3][l][k], out[False][3][l][k], what=f"layer {l} {k}", rtol=1e-5, atol_scale=1e-6)
        assert not np.array_equal(out[True][1], P["i_embeddings.weight"])
    else:
        assert out[True][0] == out[False][0]
        assert np.array_equal(out[True][1], out[False][1]) and np.array_equal(out[True][2], out[False][2])
        assert all(np.array_equal(out[True][3][l][k], out[False][3][l][k]) for l in range(n_layers) for k in LAYER_NAMES)
    with pytest.raises(ValueError):
        eng.SasrecTrainer(to_dev(P, n_layers, cuda), n_heads, opt="Adam", rowwise=False, graph=True)


def test_sasrec_trainer_graph_adam_counts_replayed_steps_on_the_host_too(cuda, eng, monkeypatch):
    """Adam under replay keeps its step count in device memory; a batch shape that leaves the captured route (the short last
    batch of an epoch: fewer occurrences than the one-wave-per-row update wants) runs eagerly with the HOST's count -- which
    therefore has to follow the replays.  A stale count (t = 3 after a dozen replays) scales that step by the wrong bias
    corrections; compared with the eager trainer on the same batch sequence."""
    from rechorus_amd import graph as hgraph
    if not hgraph.usable():
        pytest.skip("hipGraph replay not enabled in this process")
    rng = np.random.default_rng(23)
    L, d, n_layers, n_heads, C, n_items = 50, 64, 1, 4, 20, 400
    P = _random_sasrec(rng, n_items, d, n_layers, L)

    def batch(B):
        lengths = rng.integers(1, L + 1, size=B).astype(np.int64)
        hist = rng.integers(1, n_items, size=(B, L)).astype(np.int64) * (np.arange(L)[None, :] < lengths[:, None])
        iid = rng.integers(1, n_items, size=(B, C)).astype(np.int64)
        return tuple(torch.from_numpy(x).to(cuda) for x in (hist, lengths, iid))
    batches = [batch(500) for _ in range(9)] + [batch(24)] + [batch(500) for _ in range(2)]
    monkeypatch.setattr(eng, "_SAS_OVERLAP_MIN", 0)
    out = {}
    for graph in (False, True):
        Pd = to_dev(P, n_layers, cuda)
        tr = eng.SasrecTrainer(Pd, n_heads, opt="Adam", lr=1e-2, l2=1e-5, rowwise=True, graph=graph)
        assert not graph or not tr._dev_step_route(batches[9][0], batches[9][2])   # the short batch is off the captured route
        losses = [float(tr.step(*b)[0]) for b in batches]
        torch.cuda.synchronize()
        assert tr.step_count == len(batches)
        if graph:
            assert int(tr._step_dev.item()) == len(batches) and len(tr._graphs) == 1
        out[graph] = (losses, Pd["item_emb"].cpu().numpy(), Pd["pos_emb"].cpu().numpy())
    assert np.allclose(out[True][0], out[False][0], rtol=1e-5, atol=0)
    assert_close(out[True][1], out[False][1], what="item table", rtol=1e-5, atol_scale=1e-6)
    assert_close(out[True][2], out[False][2], what="position table", rtol=1e-5, atol_scale=1e-6)


@pytest.mark.parametrize("rowwise", [False, True])
def test_sasrec_trainer_two_streams_equal_one_stream(rowwise, cuda, eng, monkeypatch):
    """SasrecTrainer sorts the batch's ids beside the encoder and forms the position-table gradient beside the item-table
    update on a second stream; the kernels and their inputs are the same, so three steps leave every parameter bit-identical
    to the one-stream order (RC_SAS_OVERLAP=0)"""
    rng = np.random.default_rng(5)
    B, L, d, n_layers, n_heads, C, n_items = 600, 50, 64, 2, 4, 20, 400
    P = _random_sasrec(rng, n_items, d, n_layers, L)
    batches = []
    for _ in range(3):
        lengths = rng.integers(1, L + 1, size=B).astype(np.int64)
        hist = rng.integers(1, n_items, size=(B, L)).astype(np.int64) * (np.arange(L)[None, :] < lengths[:, None])
        iid = rng.integers(1, n_items, size=(B, C)).astype(np.int64)
        batches.append(tuple(torch.from_numpy(x).to(cuda) for x in (hist, lengths, iid)))
    out = {}
    for mode in (True, False):
        monkeypatch.setattr(eng, "_SAS_OVERLAP", mode)
        monkeypatch.setattr(eng, "_SAS_OVERLAP_MIN", 0)
        Pd = to_dev(P, n_layers, cuda)
        tr = eng.SasrecTrainer(Pd, n_heads, opt="Adam", lr=1e-3, l2=1e-5, rowwise=rowwise)
        losses = [float(tr.step(*b)[0]) for b in batches]
        torch.cuda.synchronize()
        assert (tr._side is not None) == mode
        out[mode] = (losses, Pd["item_emb"].cpu().numpy(), Pd["pos_emb"].cpu().numpy(),
                     [{k: v.cpu().numpy() for k, v in lay.items()} for lay in Pd["layers"]])
    assert out[True][0] == out[False][0]
    assert np.array_equal(out[True][1], out[False][1]) and np.array_equal(out[True][2], out[False][2])
    assert all(np.array_equal(out[True][3][l][k], out[False][3][l][k]) for l in range(n_layers) for k in LAYER_NAMES)
    # the one-wave-per-row item update parks the padding occurrences behind the table; the head-list route sums them (zeros) into
    # row 0: same tables up to the summation order of a row's occurrences
    monkeypatch.setattr(eng, "_SEG_ROWS", False)
    Pd = to_dev(P, n_layers, cuda)
    tr = eng.SasrecTrainer(Pd, n_heads, opt="Adam", lr=1e-3, l2=1e-5, rowwise=rowwise)
    for b in batches[:1]:
        tr.step(*b)
    monkeypatch.setattr(eng, "_SEG_ROWS", True)
    Pd2 = to_dev(P, n_layers, cuda)
    tr2 = eng.SasrecTrainer(Pd2, n_heads, opt="Adam", lr=1e-3, l2=1e-5, rowwise=rowwise)
    for b in batches[:1]:
        tr2.step(*b)
    torch.cuda.synchronize()
    d_ref = Pd["item_emb"].cpu().numpy() - P["i_embeddings.weight"]
    d_got = Pd2["item_emb"].cpu().numpy() - P["i_embeddings.weight"]
    # Adam normalises the step: an element whose gradient is rounding noise may move by lr either way -- allow a few
    assert float((np.abs(d_got - d_ref) > 2e-5).mean()) < 5e-3 and float(np.abs(d_got - d_ref).max()) <= 2.5e-3
    assert np.array_equal(d_got[0] != 0, d_ref[0] != 0)   # row 0 (padding id) is touched by both routes or by neither


@pytest.mark.parametrize("d,n_layers,n_heads,L,B", [(64, 2, 4, 50, 1400), (64, 1, 2, 20, 90), (32, 2, 2, 50, 1400), (32, 1, 1, 7, 3)])
def test_sasrec_16_row_projection_kernels_equal_lds_tile_kernels(d, n_layers, n_heads, L, B, cuda, eng, monkeypatch):
    """the QKV projection and the dX = dZ + dQ Wq + dK Wk + dV Wv sum on 16 x 16 x 4 tiles with operands straight from global
    memory (sb_qkv16_kernel / sb_sum3_16_kernel) against the LDS-tile 32 x 32 x 2 kernels (RC_SAS_ROWS16=0); 16-wave workgroups
    (B * history_max >= 65,536 rows) and 4-wave ones, ragged last tiles, empty histories"""
    monkeypatch.setenv("RC_SAS_LAST_ROW", "0")   # (as above: the all-rows kernels under test)
    rng = np.random.default_rng(77 * d + B)
    n_items = 300
    P = _random_sasrec(rng, n_items, d, n_layers, L)
    lengths = rng.integers(0, L + 1, size=B).astype(np.int64)
    lengths[0] = L
    hist = rng.integers(1, n_items, size=(B, L)).astype(np.int64) * (np.arange(L)[None, :] < lengths[:, None])
    Pd = to_dev(P, n_layers, cuda)
    h_d, l_d = torch.from_numpy(hist).to(cuda), torch.from_numpy(lengths).to(cuda)
    dhv = torch.from_numpy(rng.normal(size=(B, d)).astype(np.float32)).to(cuda)
    out = {}
    for mode in ("1", "0", "1"):
        monkeypatch.setenv("RC_SAS_ROWS16", mode)
        hv, saved = eng.sasrec_fwd(Pd["item_emb"], Pd["pos_emb"], Pd["layers"], n_heads, h_d, l_d, save=True, impl="batch")
        g_hist, dg = eng.sasrec_bwd(Pd["layers"], n_heads, l_d, saved, dhv)
        torch.cuda.synchronize()
        res = (hv.cpu().numpy(), g_hist.cpu().numpy(), [{k: v.cpu().numpy() for k, v in g.items()} for g in dg])
        if mode in out:
            assert np.array_equal(res[0], out[mode][0]) and np.array_equal(res[1], out[mode][1])
        out[mode] = res
    what = f"d={d} layers={n_layers} heads={n_heads} L={L} B={B}"
    assert not np.array_equal(out["1"][1], out["0"][1]), what + ": the switch had no effect"
    assert_close(out["1"][0], out["0"][0], what=what + " hv", rtol=2e-5, atol_scale=2e-5)
    assert_close(out["1"][1], out["0"][1], what=what + " g_hist", rtol=5e-5, atol_scale=5e-5)
    floor = 1e-6 * max(float(np.abs(v).max()) for g in out["0"][2] for v in g.values())
    for l in range(n_layers):
        for k in LAYER_NAMES:
            assert_close(out["1"][2][l][k], out["0"][2][l][k], what=f"{what} layer {l} d{k}", rtol=5e-5, atol_scale=1e-4, abs_floor=floor)
    assert np.all(out["1"][0][lengths == 0] == 0) and np.all(out["1"][1][lengths == 0] == 0)


# (d, layers, heads, L, B, versions of the last-row path the shape is eligible for)
LAST_ROW_CASES = [(64, 1, 4, 50, 700, "12"), (64, 2, 4, 50, 300, "12"), (64, 1, 1, 20, 90, "12"), (32, 1, 2, 64, 130, "12"),
                  (64, 3, 2, 7, 40, "12"), (64, 1, 4, 2, 9, "1"), (32, 1, 4, 12, 50, "2"), (64, 2, 2, 3, 33, "12"),
                  (64, 1, 4, 4, 21, "1"), (32, 2, 1, 64, 70, "12"), (64, 1, 2, 64, 260, "12")]


@pytest.mark.parametrize("d,n_layers,n_heads,L,B,eligible", LAST_ROW_CASES)
def test_sasrec_last_row_path_equals_all_rows(d, n_layers, n_heads, L, B, eligible, cuda, eng, monkeypatch):
    """Only position len - 1 of the last block is consumed (SASRec.py:76): the batch encoder runs the last block for one query
    row per sequence and mirrors it in the backward -- RC_SAS_LAST_ROW=1: k / v on all rows, one attention row per head;
    =2 (default): without keys and values at all (csrc/sas_last_row.hpp: scores = (Wk_h^T q_h) . x_j, ctx = Wv_h sum_j p_j x_j;
    with one block the rows come straight from the tables and the gradient rows are written padded).  Against the all-rows
    path (RC_SAS_LAST_ROW=0): hv, the history gradient and every parameter gradient, empty and single-item histories included."""
    rng = np.random.default_rng(31 * d + L + B)
    n_items = 300
    P = _random_sasrec(rng, n_items, d, n_layers, L)
    lengths = rng.integers(0, L + 1, size=B).astype(np.int64)
    lengths[:4] = (L, 0, 1, min(2, L))
    hist = rng.integers(1, n_items, size=(B, L)).astype(np.int64) * (np.arange(L)[None, :] < lengths[:, None])
    Pd = to_dev(P, n_layers, cuda)
    h_d, l_d = torch.from_numpy(hist).to(cuda), torch.from_numpy(lengths).to(cuda)
    dhv = torch.from_numpy(rng.normal(size=(B, d)).astype(np.float32)).to(cuda)
    out = {}
    for mode in ("2", "1", "0", "2", "1"):
        monkeypatch.setenv("RC_SAS_LAST_ROW", mode)
        hv, saved = eng.sasrec_fwd(Pd["item_emb"], Pd["pos_emb"], Pd["layers"], n_heads, h_d, l_d, save=True, impl="batch")
        g_hist, dg = eng.sasrec_bwd(Pd["layers"], n_heads, l_d, saved, dhv)
        torch.cuda.synchronize()
        res = (hv.cpu().numpy(), g_hist.cpu().numpy(), [{k: v.cpu().numpy() for k, v in g.items()} for g in dg])
        if mode in out:   # run to run: bit for bit
            assert np.array_equal(res[0], out[mode][0]) and np.array_equal(res[1], out[mode][1])
            assert all(np.array_equal(res[2][l][k], out[mode][2][l][k]) for l in range(n_layers) for k in LAYER_NAMES)
        out[mode] = res
    floor = 1e-6 * max(float(np.abs(v).max()) for g in out["0"][2] for v in g.values())
    for mode in ("1", "2"):
        what = f"d={d} layers={n_layers} heads={n_heads} L={L} B={B} last-row version {mode}"
        if mode in eligible:
            assert not np.array_equal(out[mode][1], out["0"][1]), what + ": the switch had no effect"
        if "1" in eligible and "2" in eligible:
            assert not np.array_equal(out["1"][1], out["2"][1]), what + ": versions 1 and 2 ran the same kernels"
        assert_close(out[mode][0], out["0"][0], what=what + " hv", rtol=2e-5, atol_scale=2e-5)
        assert_close(out[mode][1], out["0"][1], what=what + " g_hist", rtol=5e-5, atol_scale=5e-5)
        for l in range(n_layers):
            for k in LAYER_NAMES:
                assert_close(out[mode][2][l][k], out["0"][2][l][k], what=f"{what} layer {l} d{k}", rtol=5e-5, atol_scale=1e-4, abs_floor=floor)
        assert np.all(out[mode][0][lengths == 0] == 0) and np.all(out[mode][1][lengths == 0] == 0)
        pad = np.arange(L)[None, :] >= lengths[:, None]
        assert np.all(out[mode][1][pad] == 0), what + ": gradient rows past the length"


def test_sasrec_pos_grad_chunks(cuda, eng):
    """more than 256 sequences: several chunks per position + the chunk reduction; vs the generic sort + segmented sum"""
    rng = np.random.default_rng(4)
    for B, L, d in ((2500, 50, 64), (1030, 7, 32), (3, 20, 64)):
        lengths = torch.from_numpy(rng.integers(0, L + 1, size=B).astype(np.int64)).to(cuda)
        g_hist = torch.from_numpy(rng.normal(size=(B, L, d)).astype(np.float32)).to(cuda)
        g_hist = (g_hist * (torch.arange(L, device=cuda)[None, :, None] < lengths[:, None, None])).contiguous()
        position = ((lengths[:, None] - torch.arange(L, device=cuda)[None, :]).clamp(min=0)).contiguous()
        want = eng.embedding_dense_backward(g_hist, position, L + 3)
        got = eng.sasrec_pos_grad(g_hist, lengths, L + 3)
        assert torch.equal(got[0], torch.zeros_like(got[0])) and torch.equal(got[L + 1:], torch.zeros_like(got[L + 1:]))
        assert_close(got[1:].cpu().numpy(), want[1:].cpu().numpy(), what=f"pos grad B={B}", rtol=1e-5, atol_scale=2e-5)
        assert torch.equal(got, eng.sasrec_pos_grad(g_hist, lengths, L + 3))


@pytest.mark.parametrize("case", DROP_CASES)
def test_sasrec_training_mode_dropout_matches_reference(case, cuda, eng):
    """rc_sasrec_batch_fwd_dropout / _bwd_dropout vs the reference in training mode with its nn.Dropout modules
    swapped for the same counter-based mask (tests/golden/sasrecdrop_*.npz)"""
    g = load_golden(case)
    n_layers, n_heads = int(g["meta"][2]), int(g["meta"][3])
    P = to_dev(params(g), n_layers, cuda)
    hist, lengths, iid = (torch.from_numpy(g[k]).to(cuda) for k in ("hist", "len", "iid"))
    B, L = g["hist"].shape
    C, d = g["iid"].shape[1], P["item_emb"].shape[1]
    p = float(g["p"])
    seed = torch.tensor([int(g["mask_seed"])], dtype=torch.int64, device=cuda)
    hv, xsave = eng.sasrec_fwd(P["item_emb"], P["pos_emb"], P["layers"], n_heads, hist, lengths, save=True, drop_p=p, seed=seed)
    assert xsave.impl == "batch"
    rows = torch.arange(B, device=cuda)
    pred = eng.gather_dot(hv, P["item_emb"], rows, iid)
    assert_close(pred.cpu().numpy(), g["pred"], what="pred", atol_scale=2e-5)
    # p = 0 through the dropout entry point is the plain kernel bit for bit; another seed gives another mask
    hv0, _ = eng.sasrec_fwd(P["item_emb"], P["pos_emb"], P["layers"], n_heads, hist, lengths, impl="batch")
    hv0d, _ = eng.sasrec_fwd(P["item_emb"], P["pos_emb"], P["layers"], n_heads, hist, lengths, impl="batch", drop_p=0.0, seed=seed)
    assert torch.equal(hv0, hv0d) and not torch.equal(hv0, hv)
    hv_other, _ = eng.sasrec_fwd(P["item_emb"], P["pos_emb"], P["layers"], n_heads, hist, lengths, drop_p=p, seed=seed + 1)
    assert not torch.equal(hv_other, hv)
    with pytest.raises(ValueError):
        eng.sasrec_fwd(P["item_emb"], P["pos_emb"], P["layers"], n_heads, hist, lengths, impl="sequence", drop_p=p, seed=seed)

    gpred = torch.from_numpy(g["gpred"]).to(cuda)
    dhv = eng.weighted_row_sum(P["item_emb"], iid, gpred)
    g_hist, dg = eng.sasrec_bwd(P["layers"], n_heads, lengths, xsave, dhv, drop_p=p, seed=seed)
    G = params(g, "G/")
    floor = 1e-6 * max(float(np.abs(v).max()) for v in G.values())
    for l in range(n_layers):
        for k, name in LAYER_NAMES.items():
            assert_close(dg[l][k].cpu().numpy(), G["transformer_block.%d.%s" % (l, name)], what=f"layer {l} d{k}",
                         rtol=2e-5, atol_scale=5e-5, abs_floor=floor)
    ids = torch.cat([iid.reshape(-1), hist.reshape(-1)])
    keys, perm = eng.sort_ids(ids, P["item_emb"].shape[0])
    GI = torch.zeros_like(P["item_emb"])
    eng.segmented_update2(keys, perm, hv, g_hist.view(-1, d), B * C, coef=gpred.reshape(-1), div=C, dense_grad=GI)
    assert_close(GI.cpu().numpy(), G["i_embeddings.weight"], what="d item_emb", rtol=2e-5, atol_scale=5e-5)
    GP = eng.sasrec_pos_grad(g_hist, lengths, P["pos_emb"].shape[0])
    assert_close(GP.cpu().numpy(), G["p_embeddings.weight"], what="d pos_emb", rtol=2e-5, atol_scale=5e-5)


def test_sasrec_trainer_with_dropout_draws_a_fresh_mask_per_step(cuda, eng):
    """SasrecTrainer(dropout=p): the device seed is bumped every step (two steps from the same state differ from
    two steps without dropout, and a re-run with the same seed reproduces them bit for bit)"""
    rng = np.random.default_rng(5)
    n_items, d, L, B, K = 200, 64, 20, 300, 7  # B * L >= 4096 or not: dropout forces the batch kernels anyway
    def fresh():
        r = np.random.default_rng(6)
        t = lambda *s: torch.from_numpy(r.normal(0, 0.1, s).astype(np.float32)).to(cuda)
        lay = {k: (t(d, d) if k.startswith("W") else t(d)) for k in eng.SAS_LAYER_KEYS}
        lay["ln1w"] += 1
        lay["ln2w"] += 1
        return {"item_emb": t(n_items, d), "pos_emb": t(L + 1, d), "layers": [lay]}
    lengths = torch.from_numpy(rng.integers(1, L + 1, B)).to(cuda)
    hist = torch.zeros((B, L), dtype=torch.int64, device=cuda)
    for b in range(B):
        hist[b, :int(lengths[b])] = torch.from_numpy(rng.integers(1, n_items, int(lengths[b]))).to(cuda)
    iid = torch.from_numpy(rng.integers(1, n_items, (B, 1 + K))).to(cuda)
    runs = []
    for p, seed in ((0.3, 11), (0.3, 11), (0.3, 12), (0.0, 11)):
        P = fresh()
        tr = eng.SasrecTrainer(P, 2, opt="SGD", lr=0.05, rowwise=True, dropout=p, seed=seed)
        losses = [float(tr.step(hist, lengths, iid).item()) for _ in range(2)]
        runs.append((losses, P["item_emb"].clone()))
    assert runs[0][0] == runs[1][0] and torch.equal(runs[0][1], runs[1][1])
    assert runs[0][0] != runs[2][0] and runs[0][0] != runs[3][0]
    assert all(np.isfinite(x) for r in runs for x in r[0])
