"""torch.nn-shaped front of the HIP engine: what a ReChorus model file touches.

`HipEmbedding` is a drop-in for `nn.Embedding` (same `.weight` Parameter, same state_dict key)
whose forward and backward run in librechorus_hip.so; `bprmf_scores` / `bpr_loss` are the fused
head and loss as autograd Functions, so the reference's call order
`model(batch) -> model.loss(out) -> loss.backward() -> optimizer.step()`
(helpers/BaseRunner.py:193-206) keeps working unchanged while every FLOP is HIP.

Gradients reaching a table are DENSE `[n_rows, d]` tensors like autograd's
(aten::embedding_dense_backward semantics), built by sort + segmented sum (no atomics), so any
torch optimizer or `HipOptimizer` (rc_dense_update, exact torch.optim maths) can consume them.
The large-table row-wise path bypasses autograd entirely (`engine.BprmfTrainer`).
"""
import os

import torch
import torch.nn as nn

from . import engine


class _EmbeddingFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, weight, ids):
        ctx.save_for_backward(ids)
        ctx.n_rows = weight.shape[0]
        return engine.gather_rows(weight.detach(), ids)

    @staticmethod
    def backward(ctx, grad_out):
        (ids,) = ctx.saved_tensors
        # a [B] or [B, <= 2] id tensor is a user / candidate list (the reference's default --num_neg 1): a row repeats only by
        # popularity, the two-launch small route fits.  A wider one may be a padded history window, whose padding id collects
        # thousands of occurrences: those keep the plan / sort routes
        route = "small" if (ids.dim() <= 1 or ids.shape[-1] <= 2) else None
        g = engine.embedding_dense_backward(grad_out.contiguous(), ids, ctx.n_rows, route=route)
        return g, None


class HipEmbedding(nn.Module):
    """nn.Embedding(num_embeddings, embedding_dim) on the HIP engine (reference call sites:
    models/general/BPRMF.py:31-32,39-40 and the 88 other `nn.Embedding` definitions)."""

    def __init__(self, num_embeddings, embedding_dim):
        super().__init__()
        self.num_embeddings, self.embedding_dim = num_embeddings, embedding_dim
        self.weight = nn.Parameter(torch.empty(num_embeddings, embedding_dim))
        nn.init.normal_(self.weight)  # nn.Embedding's default; models re-init via init_weights

    def forward(self, ids):
        if not self.weight.is_cuda:
            raise RuntimeError("HipEmbedding runs on the GPU only (move the model to cuda)")
        return _EmbeddingFn.apply(self.weight, ids.contiguous())

    def extra_repr(self):
        return f"{self.num_embeddings}, {self.embedding_dim} [HIP]"


_DROP_SEED_GEN = None


def fresh_drop_seed():
    """int64 [1] seed for a module's counter-based dropout masks, drawn from a generator of its OWN (seeded from torch's
    initial seed, so runs with the same --random_seed repeat) -- the global RNG stream that initialises the parameters is
    left exactly as the reference leaves it."""
    global _DROP_SEED_GEN
    if _DROP_SEED_GEN is None:
        _DROP_SEED_GEN = torch.Generator()
        _DROP_SEED_GEN.manual_seed((torch.initial_seed() * 6364136223846793005 + 1442695040888963407) % (2 ** 63))
    return torch.randint(0, 2 ** 62, (1,), dtype=torch.int64, generator=_DROP_SEED_GEN)


def adopt_embeddings(model):
    """Make an unmodified ReChorus model file run its table lookups on the HIP engine: every plain `nn.Embedding`
    submodule (the reference defines all 88 of its tables that way, sparse=False, no padding_idx / max_norm) is
    replaced by a `HipEmbedding` that SHARES its `weight` Parameter -- state_dict keys, init_weights, optimizer
    parameter groups and checkpoints stay what the model file expects.  Returns the number of tables adopted;
    embeddings with options HipEmbedding does not implement are left on torch."""
    n = 0
    for parent in list(model.modules()):
        for name, child in list(parent.named_children()):
            if type(child) is not nn.Embedding:
                continue
            if child.sparse or child.padding_idx is not None or child.max_norm is not None or child.scale_grad_by_freq:
                continue
            hip = HipEmbedding.__new__(HipEmbedding)
            nn.Module.__init__(hip)
            hip.num_embeddings, hip.embedding_dim = child.num_embeddings, child.embedding_dim
            hip.weight = child.weight
            setattr(parent, name, hip)
            n += 1
    return n


class _BprmfScoreFn(torch.autograd.Function):
    """prediction[b,c] = <U[uid[b]], I[iid[b,c]]>  (models/general/BPRMF.py:39-42)."""

    @staticmethod
    def forward(ctx, U, I, uid, iid):
        ctx.save_for_backward(U, I, uid, iid)
        return engine.gather_dot(U.detach(), I.detach(), uid, iid)

    @staticmethod
    def backward(ctx, gpred):
        U, I, uid, iid = ctx.saved_tensors
        Ud, Id = U.detach(), I.detach()
        gpred = gpred.contiguous()
        C = iid.shape[1]
        # dL/dI[r] = sum_{(b,c): iid[b,c]=r} g[b,c] * U[uid[b]]  (rows rebuilt on the fly)
        keys, perm = engine.sort_ids(iid, I.shape[0])
        GI = torch.zeros_like(Id)
        engine.segmented_update(keys, perm, Ud, coef=gpred.reshape(-1), src_index=uid, div=C,
                                dense_grad=GI)
        # dL/dU[r] = sum_{b: uid[b]=r} sum_c g[b,c] * I[iid[b,c]]
        ug = engine.weighted_row_sum(Id, iid, gpred)
        GU = engine.embedding_dense_backward(ug, uid, U.shape[0])
        return GU, GI, None, None


def bprmf_scores(U, I, uid, iid):
    return _BprmfScoreFn.apply(U, I, uid.contiguous(), iid.contiguous())


class _BprLossFn(torch.autograd.Function):
    """GeneralModel.loss (models/BaseModel.py:182-185), closed-form backward."""

    @staticmethod
    def forward(ctx, pred):
        loss, _, gpred = engine.bpr_loss(pred.detach().contiguous())
        ctx.save_for_backward(gpred)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, grad_loss):
        (gpred,) = ctx.saved_tensors
        return gpred * grad_loss


def bpr_loss(pred):
    return _BprLossFn.apply(pred)


class _ListLossFn(torch.autograd.Function):
    """list-wise losses of ImpressionModel.loss on the HIP engine (rc_list_loss_fwd_bwd): the list-level BPR family with
    its re-weighting variants, listnet, softmaxCE, attention_rank; closed-form backward.  kind: engine.list_kind(loss_n)"""

    @staticmethod
    def forward(ctx, pred, target, max_pos, kind):
        loss, gpred = engine.list_loss(pred.detach().contiguous(), target.contiguous(), max_pos, kind)
        ctx.save_for_backward(gpred)
        ctx.per_row = kind == engine.LIST_KINDS["BPRsimple"]   # the reference leaves this one unreduced: [B]
        return loss if ctx.per_row else loss.reshape(())

    @staticmethod
    def backward(ctx, grad_loss):
        (gpred,) = ctx.saved_tensors
        return gpred * (grad_loss[:, None] if ctx.per_row else grad_loss), None, None, None


def list_loss(pred, target, max_pos, kind):
    return _ListLossFn.apply(pred, target, max_pos, kind)


class _FmSecondOrderFn(torch.autograd.Function):
    """0.5 * sum_k ((sum_f v)^2 - sum_f v^2) over stacked field vectors (models/context/FM.py:61)."""

    @staticmethod
    def forward(ctx, V):
        Vc = V.detach().contiguous()
        ctx.save_for_backward(Vc)
        return engine.fm_second_order(Vc)

    @staticmethod
    def backward(ctx, gout):
        (Vc,) = ctx.saved_tensors
        return engine.fm_second_order_bwd(Vc, gout.contiguous())


def fm_second_order(V):
    if not V.is_cuda:
        raise RuntimeError("fm_second_order runs on the GPU only (no CPU path)")
    return _FmSecondOrderFn.apply(V)  # float4 lane-group kernels for emb_size 16/32/64/128, a generic kernel otherwise


class _FmTapFn(torch.autograd.Function):
    """the FM pairwise term AND the flattened field vectors for a second consumer (DeepFM's deep tower, models/context/DeepFM.py:
    19-28) as one autograd node: its backward forms d fm2 / dV on top of the gradient the second consumer sends back in ONE pass
    (rc_fm_second_order_bwd_add) -- as two nodes autograd writes both gradients out and adds them (three more passes over
    [B, C, F, d])."""

    @staticmethod
    def forward(ctx, V):
        Vc = V.detach().contiguous()
        ctx.save_for_backward(Vc)
        flat = Vc.view(*Vc.shape[:-2], Vc.shape[-2] * Vc.shape[-1])
        return engine.fm_second_order(Vc), flat

    @staticmethod
    def backward(ctx, g_fm, g_flat):
        (Vc,) = ctx.saved_tensors
        if g_fm is None:
            return None if g_flat is None else g_flat.reshape(Vc.shape)
        add = None if g_flat is None else g_flat.contiguous().view(Vc.shape)
        return engine.fm_second_order_bwd(Vc, g_fm.contiguous(), add=add)


def fm_second_order_and_flat(V):
    """(fm_second_order(V), V.flatten(start_dim=-2)) as one autograd node"""
    if not V.is_cuda:
        raise RuntimeError("fm_second_order_and_flat runs on the GPU only (no CPU path)")
    return _FmTapFn.apply(V)


class _FieldGatherFn(torch.autograd.Function):
    """stacked field vectors [B, C, F, d] of F embedding tables (models/context/FM.py:49-52); the backward
    builds all F dense gradients with one sort + one segmented sum over the composite (field, id) key."""

    # Two table families gathered with the SAME id tensors (FM-family models: the [vocab, d] vectors and the [vocab, 1] first-order
    # weights of every field, models/context/FM.py:44-57) share the composite keys and, in the backward pass, their sort: the
    # last forward's (ids, key tensor) and the last backward's sorted keys are remembered.  The entries hold the tensors
    # themselves (identity + version are compared, never addresses: a freed tensor's address can come back).
    _last_cid = None      # (ids tuple, versions, n_cand, offsets, cid)
    _last_sort = None     # (cid, n_rows, keys, perm)
    _last_small = None    # (cid, n_rows): the small route's grouping of these keys sits in the workspace "edb_small_fields"

    @staticmethod
    def forward(ctx, n_cand, n_fields, *args):
        ids, tables = args[:n_fields], args[n_fields:]
        ids_c = [x.contiguous() for x in ids]
        offs_now = [0]
        for t in tables:
            offs_now.append(offs_now[-1] + t.shape[0])
        # The caches live for ONE forward / backward of a model: a hit consumes the entry, a backward drops it, and nothing is
        # remembered without autograd (an evaluation loop over a static id buffer rewritten through raw pointers -- which
        # does not bump _version -- can therefore never meet a stale key tensor).
        last = _FieldGatherFn._last_cid
        _FieldGatherFn._last_cid = None
        hit = (last is not None and len(last[0]) == len(ids_c) and all(a is b for a, b in zip(last[0], ids_c))
               and last[1] == tuple(x._version for x in ids_c) and last[2] == n_cand and last[3] == offs_now)
        out, cid, offs = engine.gather_fields([t.detach() for t in tables], ids_c, n_cand, want_cid=not hit)
        if hit:
            cid = last[4]
        else:
            _FieldGatherFn._last_sort = None
            _FieldGatherFn._last_small = None
            if torch.is_grad_enabled():
                _FieldGatherFn._last_cid = (tuple(ids_c), tuple(x._version for x in ids_c), n_cand, offs_now, cid)
        ctx.cid, ctx.offs = cid, offs
        # a field whose vocabulary is small against the batch makes hot rows: the sort-driven reduction handles any skew
        n_ids = cid.numel() // max(1, n_fields)
        # (a field value occurs at most once per row: up to 8,192 ids in all the small route fits whatever the vocabularies)
        ctx.route = "small" if cid.numel() <= 8192 else ("sort" if min(t.shape[0] for t in tables) * 8 <= n_ids else None)
        return out

    @staticmethod
    def backward(ctx, gout):
        offs = ctx.offs
        presorted = None
        _FieldGatherFn._last_cid = None     # the forward this belongs to is over
        if ctx.route == "sort":
            last = _FieldGatherFn._last_sort
            if last is not None and last[0] is ctx.cid and last[1] == offs[-1]:
                presorted = last[2:]
                _FieldGatherFn._last_sort = None    # second (last) user of this sort
            else:
                presorted = engine.sort_ids(ctx.cid.reshape(-1), offs[-1])
                _FieldGatherFn._last_sort = (ctx.cid, offs[-1]) + tuple(presorted)
        again = False
        if ctx.route == "small" and engine.small_route_ok(ctx.cid.numel(), offs[-1], gout.shape[-1]):
            # the small route groups the composite keys in its first launch: the second family gathered with the same keys
            # (same key tensor, nothing else used that workspace in between: it has a tag of its own) runs the sums only
            last = _FieldGatherFn._last_small
            again = last is not None and last[0] is ctx.cid and last[1] == offs[-1]
            _FieldGatherFn._last_small = None if again else (ctx.cid, offs[-1])
        G = engine.embedding_dense_backward(gout.contiguous(), ctx.cid, offs[-1], route=ctx.route, presorted=presorted,
                                            small_again=again, small_tag="edb_small_fields")  # virtual concatenated table
        grads = tuple(G[offs[f]:offs[f + 1]] for f in range(len(offs) - 1))
        return (None, None) + (None,) * len(grads) + grads


def gather_fields(tables, ids, n_cand):
    """tables: F HipEmbedding weights of equal width; ids: F int64 tensors ([B] or [B, C])"""
    return _FieldGatherFn.apply(n_cand, len(tables), *ids, *tables)


class _FieldGatherPairFn(torch.autograd.Function):
    """the TWO table families the FM models gather with the same ids (models/context/FM.py:44-57: field vectors [vocab, d] and
    first-order weights [vocab, 1]) as one autograd node: one gather launch forward (rc_gather_fields_pair); backward, ONE
    grouping of the composite (field, id) keys serves both dense gradients -- the small route's plan (rc_small_row_sums +
    rc_small_row_sums_again on one zero-filled buffer), or one sort -- instead of one per family.
    Numeric fields (kinds[f] != 0; FM.py:38-41,47-48: Linear(1, d) / Linear(1, 1) on the feature's value) ride in the same gather
    launch (rc_gather_fields_mixed: their "row" is x * W[:, 0]); they own no row of the concatenated table, their occurrences
    take no part in the grouping, and their weight gradients are one weighted column sum (rc_numeric_field_grads)."""

    @staticmethod
    def forward(ctx, n_cand, n_fields, rows_opt, kinds, want_fm, bump, *args):
        ids = [x.contiguous() for x in args[:n_fields]]
        tables, tables1 = args[n_fields:2 * n_fields], args[2 * n_fields:]
        kinds = tuple(kinds) if kinds is not None else (engine.FIELD_IDS,) * n_fields
        cat = [f for f in range(n_fields) if kinds[f] == engine.FIELD_IDS]
        num = [f for f in range(n_fields) if kinds[f] != engine.FIELD_IDS]
        d = tables[cat[0]].shape[1] if cat else tables[0].shape[0]
        rows = ids[0].shape[0] * n_cand
        n = rows * n_fields
        n_rows = sum(tables[f].shape[0] for f in cat)
        mark = None
        if rows_opt is not None and cat and n <= 8192 and d % 4 == 0 and engine.small_route_ok(n, n_rows, d):
            # rows mode (HipOptimizer.rows_begin): the gather stamps the looked-up rows, and the backward pass below hands their row
            # sums to the optimizer in a scratch that nothing zero-fills, instead of a dense gradient
            mark = rows_opt.rows_begin([tables[f] for f in cat], [tables1[f] for f in cat])
        n_ids = rows
        ctx.route = "small" if n <= 8192 else ("sort" if cat and min(tables[f].shape[0] for f in cat) * 8 <= n_ids else None)
        if num and not (ctx.route == "small" and engine.small_route_ok(n, max(n_rows, 1), d)):
            ctx.route = "sort"    # (the other groupings of embedding_dense_backward have no key that stands for "no row")
        # what the numeric occurrences carry in cid: the small route and the bucket plan skip negative keys; a sort puts the key
        # past the last row behind every real row (the grouping below stops in front of that tail)
        numeric_key = n_rows if (ctx.route == "sort" and num) else -1
        fusable = d in (16, 32, 64, 128)
        # small batches: the backward pass's grouping of the composite keys depends on the ids alone -- it runs in the gather's
        # launch (rc_gather_fields_fused) instead of as a launch of its own in front of the row sums
        plan = bool(fusable and cat and ctx.route == "small" and any(ctx.needs_input_grad) and engine.small_route_ok(n, n_rows, d))
        fm_here = bool(want_fm and fusable)
        S = fm = plan_ws = None
        if plan or fm_here:
            V, L, cid, offs, fm, S, plan_ws = engine.gather_fields([t.detach() for t in tables], ids, n_cand, want_cid=True,
                                                                   tables1=[t.detach() for t in tables1], mark=mark, kinds=kinds,
                                                                   numeric_key=numeric_key, fm=fm_here, plan=plan, bump=bump)
        else:
            V, L, cid, offs = engine.gather_fields([t.detach() for t in tables], ids, n_cand, want_cid=True,
                                                   tables1=[t.detach() for t in tables1], mark=mark,
                                                   kinds=kinds if num else None, numeric_key=numeric_key)
        if want_fm and fm is None:
            fm = engine.fm_second_order(V)      # (a width without the float4 lane-group tiling: its own launch)
        ctx.rows_opt = rows_opt if mark is not None else None
        ctx.cid, ctx.offs, ctx.d = cid, offs, d
        ctx.kinds, ctx.num, ctx.n_cand = kinds, num, n_cand
        ctx.values = [ids[f] for f in num]
        ctx.want_fm, ctx.plan_ws = bool(want_fm), plan_ws
        ctx.has_S = S is not None
        ctx.set_materialize_grads(False)
        if want_fm:
            ctx.save_for_backward(*((V, S) if S is not None else (V,)))
            return V, L, fm
        return V, L

    @staticmethod
    def backward(ctx, gV, gL, g_fm=None):
        offs, d = ctx.offs, ctx.d
        n_rows, n = offs[-1], ctx.cid.numel()
        F = len(offs) - 1
        gV = None if gV is None else gV.contiguous()
        gL = None if gL is None else gL.contiguous()
        num = ctx.num
        gw = gw1 = None
        small = ctx.route == "small" and n_rows > 0 and d % 4 == 0 and engine.small_route_ok(n, n_rows, d)
        planned = small and ctx.plan_ws is not None and gL is not None
        # the FM term's backward: added to the tower's gradient rows where the row sums read them (planned small route: never
        # written out), else one pass of its own (rc_fm_second_order_bwd_add)
        tap = None
        if ctx.want_fm and g_fm is not None:
            saved = ctx.saved_tensors
            Vs = saved[0]
            rideable = (not num) or (16 <= d <= 128 and len(num) <= engine.SMALL_NUMERIC_MAX)
            if planned and ctx.has_S and rideable:
                tap = (Vs.view(n // F, F, d), saved[1].view(n // F, d), g_fm.contiguous().view(-1))
            else:
                gV = engine.fm_second_order_bwd(Vs, g_fm.contiguous(), add=None if gV is None else gV.view(Vs.shape))
        have_v = gV is not None or tap is not None
        # small batches: the numeric fields' weight gradients ride in the row-sums launch of the small route (a launch of their own
        # is ~13 us of a replayed DeepFM step at B = 1,024); otherwise rc_numeric_field_grads on its own
        ride = (num and have_v and gL is not None and small and 16 <= d <= 128
                and len(num) <= engine.SMALL_NUMERIC_MAX)
        riding = (ctx.values, num, F, ctx.n_cand) if ride else None
        if num and not ride and (gV is not None or gL is not None):
            gw, gw1 = engine.numeric_field_grads(None if gV is None else gV.view(n // F, F, d), None if gL is None else gL.view(n // F, F),
                                                 ctx.values, num, F, ctx.n_cand, d)

        def numeric(grads, family):
            """the tables' gradient tuple with the numeric fields' Linear weights filled in"""
            if family is None:
                return grads
            out = list(grads)
            for j, f in enumerate(num):
                out[f] = family[j]
            return tuple(out)

        def row_sums(into=None):
            """(Gv, Gl) of the small route; the numeric fields' gradients land in gw / gw1 when they ride"""
            nonlocal gw, gw1
            if planned and have_v:
                res = engine.small_row_sums_planned(ctx.plan_ws, n, n_rows, None if gV is None else gV.view(n, d), gL.view(n, 1), d,
                                                    (F, n // (F * ctx.n_cand), ctx.n_cand), into=into, numeric=riding, fm=tap)
            else:
                res = engine.small_row_sums_pair(ctx.cid, n_rows, gV.view(n, d), gL.view(n, 1), into=into, numeric=riding)
            if ride:
                gw, gw1 = res[2:]
            return res[:2]

        lead = (None,) * 6
        if ctx.rows_opt is not None:
            if not have_v or gL is None:
                raise RuntimeError("gather_fields_pair (rows mode): both table families must reach the loss")
            Gv, Gl = row_sums(into=ctx.rows_opt.rows_scratch())
            ctx.rows_opt.rows_grads(Gv, Gl)
            return lead + (None,) * F + numeric((None,) * F, gw) + numeric((None,) * F, gw1)
        if n_rows == 0:       # numeric fields only
            Gv = Gl = None
        elif have_v and gL is not None and small:
            Gv, Gl = row_sums()
        else:
            presorted = None
            if ctx.route == "sort":
                # the numeric occurrences (key n_rows) sort behind every row of the virtual table: the grouping takes the head
                keys, perm = engine.sort_ids(ctx.cid.reshape(-1), n_rows + (1 if num else 0))
                n_cat = (n // F) * (F - len(num))
                presorted = (keys[:n_cat], perm[:n_cat])
            Gv = None if gV is None else engine.embedding_dense_backward(gV, ctx.cid, n_rows, route=ctx.route, presorted=presorted)
            Gl = None if gL is None else engine.embedding_dense_backward(gL, ctx.cid, n_rows, route=ctx.route, presorted=presorted)
        cat = lambda G, f: None if (G is None or ctx.kinds[f] != engine.FIELD_IDS) else G[offs[f]:offs[f + 1]]
        gv = numeric(tuple(cat(Gv, f) for f in range(F)), gw)
        gl = numeric(tuple(cat(Gl, f) for f in range(F)), gw1)
        return lead + (None,) * F + gv + gl


def gather_fields_pair(tables, tables1, ids, n_cand, rows_opt=None, kinds=None, fm=False, bump=None):
    """-> (field vectors [B, C, F, d], first-order values [B, C, F, 1]) of the two table families looked up with the same ids.
    fm=True: -> (vectors, values, FM pairwise term [B, C]) -- models/context/FM.py:61's 0.5 sum_k ((sum_f v)^2 - sum_f v^2) formed
    in the gather's launch; its backward is added to the vectors' other gradient (DeepFM's tower) where the row sums read it.
    bump: an int64 [1] device counter whose next engine.step_increment the gather's launch performs early, where it has a slot for
    it (the dropout seed of the tower that consumes the vectors: one one-thread launch fewer per step); nothing the gather launches
    may read it.
    rows_opt: the HipOptimizer that owns the tables, when this forward is part of a whole training step whose optimizer.step()
    follows (graph.GraphedStep sets it): small batches then take the optimizer's rows mode.
    kinds: per field engine.FIELD_IDS (a table looked up by ids) or the value type of a numeric field (engine.field_kind): then
    tables[f] / tables1[f] are the weights of its Linear(1, d) / Linear(1, 1) and ids[f] holds the feature's values"""
    return _FieldGatherPairFn.apply(n_cand, len(tables), rows_opt, None if kinds is None else tuple(kinds), bool(fm), bump, *ids, *tables, *tables1)


class _BceProbFn(torch.autograd.Function):
    """nn.BCELoss on probabilities (models/BaseModel.py:259-267), closed-form backward."""

    @staticmethod
    def forward(ctx, p, y):
        loss, gp = engine.bce_prob(p.detach().contiguous(), y.detach().contiguous().float())
        ctx.save_for_backward(gp)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, grad_loss):
        (gp,) = ctx.saved_tensors
        return gp * grad_loss, None


def bce_loss(p, y):
    return _BceProbFn.apply(p, y)


UNIT_LOSS_GRAD = None     # float32 scalar buffer holding 1.0 that the step in progress seeds loss.backward() with (graph.GraphedStep)


class _CtrHeadFn(torch.autograd.Function):
    """the CTR head of the context models in training: logit = bias + first-order sum (+ pairwise term) (+ MLP output), sigmoid,
    nn.BCELoss -- one kernel forward, the closed-form d loss / d logit fanned out to the terms backward."""

    @staticmethod
    def forward(ctx, bias, lin, label, *terms):
        n = label.numel()
        t = [x.detach().reshape(-1).contiguous() for x in terms]
        lin2 = lin.detach().reshape(n, -1).contiguous()
        ctx.lin_shape, ctx.term_shapes = lin.shape, [x.shape for x in terms]
        ctx.one_wg = n <= engine.CTR_HEAD_ONE_WG_MAX and lin2.numel() == lin.numel()
        ctx.set_materialize_grads(False)    # p is not differentiable: no [n] block of zeros is filled for it in every backward pass
        if ctx.one_wg:
            # one workgroup: probabilities, loss MEAN and sum gz (the bias gradient) in the same launch.  Inside a whole training
            # step whose backward is seeded with the constant 1 (graph.GraphedStep announces its buffer: UNIT_LOSS_GRAD) the launch
            # also leaves the backward fan-out and takes Adam's pending step-count increment along
            ctx.full = UNIT_LOSS_GRAD is not None and UNIT_LOSS_GRAD.device == lin2.device
            res = engine.ctr_head_sums(bias.detach(), lin2, t[0] if len(t) > 0 else None, t[1] if len(t) > 1 else None,
                                       label.reshape(-1).contiguous(), full=ctx.full)
            p, sums, gz = res[:3]
            ctx.save_for_backward(gz, sums, *res[3:])
            ctx.mark_non_differentiable(p)
            return p, sums[0].reshape(())
        p, loss, gz = engine.ctr_head(bias.detach(), lin2, t[0] if len(t) > 0 else None,
                                      t[1] if len(t) > 1 else None, label.reshape(-1).contiguous())
        ctx.save_for_backward(gz)
        ctx.mark_non_differentiable(p)
        return p, loss.reshape(())

    @staticmethod
    def backward(ctx, _gp, g_loss):
        if g_loss is None:      # the loss did not reach the objective
            return (None, None, None) + (None,) * len(ctx.term_shapes)
        if ctx.one_wg:
            gz, sums = ctx.saved_tensors[:2]
            n = gz.shape[0]
            unit = UNIT_LOSS_GRAD
            if ctx.full and unit is not None and g_loss.data_ptr() == unit.data_ptr() and g_loss.numel() == 1:
                g_lin, g_bias = ctx.saved_tensors[2:]       # the forward launch wrote them for exactly this seed
                return (g_bias, g_lin.view(ctx.lin_shape), None) + tuple(gz.reshape(sh) for sh in ctx.term_shapes)
            # one launch: g = gz * g_loss for the terms, the contiguous [n, F] block of the first-order weights, the bias gradient
            g, g_lin, g_bias = engine.ctr_head_bwd(gz, sums, g_loss.reshape(1).float().contiguous(), ctx.lin_shape.numel() // n)
            return (g_bias, g_lin.view(ctx.lin_shape), None) + tuple(g.reshape(sh) for sh in ctx.term_shapes)
        (gz,) = ctx.saved_tensors
        g = gz * g_loss
        n = g.shape[0]
        g_lin = g.reshape(n, *([1] * (len(ctx.lin_shape) - 1))).expand(ctx.lin_shape)
        return (g.sum().reshape(1), g_lin, None) + tuple(g.reshape(sh) for sh in ctx.term_shapes)


def ctr_head(bias, lin, label, terms):
    """-> (p [n] probabilities, BCE loss scalar); lin: first-order values [n, ..., F] (summed over the trailing dims per row),
    terms: up to two tensors of n elements added to the logit"""
    if not lin.is_cuda:
        raise RuntimeError("ctr_head runs on the GPU only (no CPU path)")
    assert len(terms) <= 2
    return _CtrHeadFn.apply(bias, lin, label, *terms)


class _BceRankingFn(torch.autograd.Function):
    """ContextModel.loss, loss_n 'BCE' (models/BaseContextModel.py:53-56), closed-form backward"""

    @staticmethod
    def forward(ctx, pred):
        loss, gpred = engine.bce_ranking(pred.detach().contiguous())
        ctx.save_for_backward(gpred)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, grad_loss):
        (gpred,) = ctx.saved_tensors
        return gpred * grad_loss


def bce_ranking_loss(pred):
    return _BceRankingFn.apply(pred)


class _NeumfFn(torch.autograd.Function):
    """NeuMF head, one hidden layer (models/general/NeuMF.py:61-75): rc_neumf_fwd / rc_neumf_bwd."""

    @staticmethod
    def forward(ctx, mf_u, mf_i, mlp_u, mlp_i, W1, b1, w_out, uid, iid, drop_p=0.0, seed=None):
        P = {"mf_u": mf_u.detach(), "mf_i": mf_i.detach(), "mlp_u": mlp_u.detach(), "mlp_i": mlp_i.detach(),
             "W1": W1.detach().contiguous(), "b1": b1.detach().contiguous(),
             "w_out": w_out.detach().reshape(-1).contiguous()}
        ctx.P, ctx.uid, ctx.iid, ctx.wshape = P, uid, iid, w_out.shape
        ctx.drop = (drop_p, seed)  # the backward regenerates the mask from the same device seed
        return engine.neumf_fwd(P, uid, iid, drop_p, seed)

    @staticmethod
    def backward(ctx, gpred):
        P, uid, iid = ctx.P, ctx.uid, ctx.iid
        rows, dense = engine.neumf_bwd(P, uid, iid, gpred.contiguous(), *ctx.drop)
        uid_occ = uid.repeat_interleave(iid.shape[1])
        ku, pu = engine.sort_ids(uid_occ, P["mf_u"].shape[0])
        ki, pi = engine.sort_ids(iid, P["mf_i"].shape[0])

        # the mf / mlp tables of a side share ids: one sort and one head list serve both
        _, hu, nhu = engine.segment_heads(ku, pu, want_single=False)
        _, hi, nhi = engine.segment_heads(ki, pi, want_single=False)

        def dense_pair(ta, tb, ga, gb, keys, perm, heads, n_heads):  # one pass over the ids for both tables
            Ga, Gb = torch.zeros_like(P[ta]), torch.zeros_like(P[tb])
            if engine.segmented_pair_supported(P[ta].shape[1]):
                engine.segmented_update_pair(keys, perm, rows[ga], rows[gb], dense_grad=(Ga, Gb), heads=heads, n_heads=n_heads)
            else:
                engine.segmented_update(keys, perm, rows[ga], dense_grad=Ga, heads=heads, n_heads=n_heads)
                engine.segmented_update(keys, perm, rows[gb], dense_grad=Gb, heads=heads, n_heads=n_heads)
            return Ga, Gb
        g_mf_u, g_mlp_u = dense_pair("mf_u", "mlp_u", "g_mf_u", "g_mlp_u", ku, pu, hu, nhu)
        g_mf_i, g_mlp_i = dense_pair("mf_i", "mlp_i", "g_mf_i", "g_mlp_i", ki, pi, hi, nhi)
        return (g_mf_u, g_mf_i, g_mlp_u, g_mlp_i,
                dense["W1"], dense["b1"], dense["w_out"].view(ctx.wshape), None, None, None, None)


def neumf_scores(mf_u, mf_i, mlp_u, mlp_i, W1, b1, w_out, uid, iid, drop_p=0.0, seed=None):
    """drop_p > 0 (training): hidden-layer dropout inside the kernels; `seed` int64 [1] device tensor the caller
    bumps once per forward (engine.step_increment)"""
    return _NeumfFn.apply(mf_u, mf_i, mlp_u, mlp_i, W1, b1, w_out, uid.contiguous(), iid.contiguous(), drop_p, seed)


_SAS_ATTRS = (("Wq", "masked_attn_head.q_linear.weight"), ("bq", "masked_attn_head.q_linear.bias"),
              ("Wk", "masked_attn_head.k_linear.weight"), ("bk", "masked_attn_head.k_linear.bias"),
              ("Wv", "masked_attn_head.v_linear.weight"), ("bv", "masked_attn_head.v_linear.bias"),
              ("ln1w", "layer_norm1.weight"), ("ln1b", "layer_norm1.bias"), ("W1", "linear1.weight"),
              ("b1", "linear1.bias"), ("W2", "linear2.weight"), ("b2", "linear2.bias"),
              ("ln2w", "layer_norm2.weight"), ("ln2b", "layer_norm2.bias"))


def _get_attr(mod, path):
    for part in path.split("."):
        mod = getattr(mod, part)
    return mod


class _SasrecEncodeFn(torch.autograd.Function):
    """SASRec encoder (models/sequential/SASRec.py:58-76): rc_sasrec_fwd / rc_sasrec_bwd.
    Inputs: item table, position table, then 14 parameters per block in engine.SAS_LAYER_KEYS order."""

    @staticmethod
    def forward(ctx, item_emb, pos_emb, n_heads, hist, lengths, drop_p, seed, *flat):
        n_layers = len(flat) // 14
        layers = [{k: flat[14 * l + j].detach().contiguous() for j, (k, _) in enumerate(_SAS_ATTRS)}
                  for l in range(n_layers)]
        need_grad = any(t.requires_grad for t in (item_emb, pos_emb) + tuple(flat))
        hv, xsave = engine.sasrec_fwd(item_emb.detach(), pos_emb.detach(), layers, n_heads, hist, lengths,
                                      save=need_grad, drop_p=drop_p, seed=seed)
        ctx.layers, ctx.n_heads, ctx.hist, ctx.lengths, ctx.xsave = layers, n_heads, hist, lengths, xsave
        ctx.drop_p, ctx.seed = drop_p, seed
        ctx.n_items, ctx.n_pos = item_emb.shape[0], pos_emb.shape[0]
        return hv

    @staticmethod
    def backward(ctx, dhv):
        hist, lengths = ctx.hist, ctx.lengths
        g_hist, dgrads = engine.sasrec_bwd(ctx.layers, ctx.n_heads, lengths, ctx.xsave, dhv.contiguous(),
                                           drop_p=ctx.drop_p, seed=ctx.seed)
        GI = engine.embedding_dense_backward(g_hist, hist, ctx.n_items)
        GP = engine.sasrec_pos_grad(g_hist, lengths, ctx.n_pos)
        flat = [g[k].contiguous() for g in dgrads for k, _ in _SAS_ATTRS]
        return (GI, GP, None, None, None, None, None) + tuple(flat)


def sasrec_layer_params(blocks):
    """[{engine.SAS_LAYER_KEYS name: parameter storage}] of utils.layers.TransformerLayer blocks (no copies)"""
    return [{k: _get_attr(b, path).data for k, path in _SAS_ATTRS} for b in blocks]


def sasrec_encode(item_emb, pos_emb, blocks, n_heads, hist, lengths, drop_p=0.0, seed=None):
    """blocks: nn.ModuleList of utils.layers.TransformerLayer (parameters only are used).  drop_p > 0 (training):
    dropout1 / dropout2 of every block inside the batch-level kernels; `seed` int64 [1] device tensor the caller
    bumps once per forward (engine.step_increment)"""
    flat = [_get_attr(b, path) for b in blocks for _, path in _SAS_ATTRS]
    return _SasrecEncodeFn.apply(item_emb, pos_emb, n_heads, hist.contiguous(), lengths.contiguous(), float(drop_p), seed, *flat)


# ---- the encoder layer by layer, any shape (csrc/seq_layers.hip + the GEMMs of csrc/mlp.hip) ---------------------------------

class _SeqEmbedFn(torch.autograd.Function):
    """item rows + position rows of a right-padded history (SASRec.py:58-66) as B * L rows; the backward is the two dense
    embedding gradients (the rows' gradients are zero on the padding)"""

    @staticmethod
    def forward(ctx, item_emb, pos_emb, hist, lengths):
        ctx.hist, ctx.lengths = hist, lengths
        ctx.n_items, ctx.n_pos = item_emb.shape[0], pos_emb.shape[0]
        return engine.seq_embed(item_emb.detach(), pos_emb.detach(), hist, lengths)

    @staticmethod
    def backward(ctx, gX):
        B, L = ctx.hist.shape
        g = gX.contiguous()
        GI = engine.embedding_dense_backward(g.view(B, L, -1), ctx.hist, ctx.n_items)
        GP = engine.seq_pos_grad(g, ctx.lengths, B, L, ctx.n_pos)
        return GI, GP, None, None


class _SeqAttentionFn(torch.autograd.Function):
    """scaled_dot_product_attention of every head (utils/layers.py:52-63) on Q / K / V [B * L, H * dk]; nothing of size L x L is
    kept: the backward recomputes the probabilities from the saved log-sum-exp"""

    @staticmethod
    def forward(ctx, q, k, v, off, B, L, H, mask, causal):
        qc, kc, vc = q.detach().contiguous(), k.detach().contiguous(), v.detach().contiguous()
        out, lse = engine.seq_attention_fwd(qc, kc, vc, off, B, L, H, mask=mask, causal=causal)
        ctx.save_for_backward(qc, kc, vc, lse)
        ctx.args = (off, B, L, H, mask, causal)
        return out

    @staticmethod
    def backward(ctx, dctx):
        qc, kc, vc, lse = ctx.saved_tensors
        off, B, L, H, mask, causal = ctx.args
        dQ, dK, dV = engine.seq_attention_bwd(qc, kc, vc, off, B, L, H, lse, dctx.contiguous(), mask=mask, causal=causal)
        return dQ, dK, dV, None, None, None, None, None, None


class _AddLayerNormFn(torch.autograd.Function):
    """LayerNorm(dropout(branch) + residual) (utils/layers.py:110,117) in one kernel each way; the dropout mask is the
    counter-based stream of the batch encoder (never stored)"""

    @staticmethod
    def forward(ctx, a, r, w, b, off, L, drop_p, seed, site):
        ac = a.detach().contiguous()
        y, xhat, rstd = engine.seq_add_layernorm_fwd(ac, None if r is None else r.detach().contiguous(), w.detach(), b.detach(), off, L,
                                                     drop_p, seed, site)
        ctx.save_for_backward(xhat, rstd, w.detach())
        ctx.args = (off, L, drop_p, seed, site, r is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        xhat, rstd, w = ctx.saved_tensors
        off, L, drop_p, seed, site, has_r = ctx.args
        dA, dR, dw, db = engine.seq_add_layernorm_bwd(dy.contiguous(), xhat, rstd, w, off, L, drop_p, seed, site, need_dR=has_r)
        return dA, dR, dw, db, None, None, None, None, None


class _SeqPickLastFn(torch.autograd.Function):
    """hv[b] = x[b, len_b - 1] (SASRec.py:76)"""

    @staticmethod
    def forward(ctx, x, lengths, B, L):
        ctx.args = (lengths, B, L)
        return engine.seq_pick_last(x.detach().contiguous(), lengths, B, L)

    @staticmethod
    def backward(ctx, dhv):
        lengths, B, L = ctx.args
        return engine.seq_pick_last_bwd(dhv.contiguous(), lengths, B, L), None, None, None


def sasrec_encode_layers(item_emb, pos_emb, blocks, n_heads, hist, lengths, drop_p=0.0, seed=None):
    """The SASRec encoder (SASRec.py:58-76) block by block for the shapes outside the register-resident kernels' envelope
    (engine.sasrec_supported): embeddings, every Linear (fp32 MFMA GEMMs, bias / ReLU in the epilogue), the attention of every
    head, LayerNorm(dropout(.) + residual) -- all HIP kernels, autograd only strings them together.  -> hv [B, d], the output
    row at position length - 1 of every sequence"""
    hist, lengths = hist.contiguous(), lengths.contiguous()
    B, L = hist.shape
    off = engine.seq_offsets(lengths, L)
    x = _SeqEmbedFn.apply(item_emb, pos_emb, hist, lengths)
    for l, blk in enumerate(blocks):
        a = blk.masked_attn_head
        q = linear(x, a.q_linear.weight, a.q_linear.bias)
        k = linear(x, a.k_linear.weight, a.k_linear.bias)
        v = linear(x, a.v_linear.weight, a.v_linear.bias)
        c = _SeqAttentionFn.apply(q, k, v, off, B, L, n_heads, None, True)        # causal mask only (SASRec.py:69-70)
        y1 = _AddLayerNormFn.apply(c, x, blk.layer_norm1.weight, blk.layer_norm1.bias, off, L, float(drop_p), seed, 2 * l)
        h = linear(y1, blk.linear1.weight, blk.linear1.bias, relu=True)
        f = linear(h, blk.linear2.weight, blk.linear2.bias)
        x = _AddLayerNormFn.apply(f, y1, blk.layer_norm2.weight, blk.layer_norm2.bias, off, L, float(drop_p), seed, 2 * l + 1)
    return _SeqPickLastFn.apply(x, lengths, B, L)


def sasrec_layers_supported(d, n_heads, L):
    """sasrec_encode_layers covers this shape: emb_size a multiple of 4 up to 1,024 whose heads divide it, history up to 1,024"""
    return d % 4 == 0 and 4 <= d <= 1024 and n_heads >= 1 and d % n_heads == 0 and engine.seq_attention_supported(L, d // n_heads)


class _LinearFn(torch.autograd.Function):
    """drop(relu(x W^T + b)) as one fp32 MFMA GEMM with the epilogue fused (rc_linear_fwd); backward = mask + two GEMMs
    (rc_linear_bwd).  The saved output is its own ReLU / dropout mask."""

    @staticmethod
    def forward(ctx, x, W, b, relu, drop_p, seed, site):
        xs = x.detach().reshape(-1, x.shape[-1]).contiguous()
        Wd = W.detach().contiguous()
        y = engine.linear_fwd(xs, Wd, None if b is None else b.detach().contiguous(), relu, drop_p, seed, site)
        ctx.save_for_backward(xs, Wd, y if (relu or drop_p > 0) else None)
        ctx.drop_p, ctx.has_b, ctx.xshape = float(drop_p), b is not None, x.shape
        ctx.need_dx = x.requires_grad
        return y.view(*x.shape[:-1], W.shape[0])

    @staticmethod
    def backward(ctx, dy):
        xs, Wd, y = ctx.saved_tensors
        dyf = dy.reshape(-1, dy.shape[-1]).contiguous()
        dX, dW, db = engine.linear_bwd(xs, Wd, y, dyf, ctx.drop_p, need_dx=ctx.need_dx, need_db=ctx.has_b)
        return (dX.view(ctx.xshape) if dX is not None else None), dW, db, None, None, None, None


def linear(x, W, b=None, relu=False, drop_p=0.0, seed=None, site=0):
    """nn.Linear (+ ReLU + training-mode dropout) on the HIP engine; x [..., K]"""
    if not x.is_cuda:
        raise RuntimeError("linear runs on the GPU only (no CPU path)")
    if float(drop_p) > 0.0 and not relu:
        # the backward pass takes the saved output as its own mask (y > 0): without the ReLU a kept negative output would
        # get a zero gradient -- refuse instead of returning wrong gradients (mlp_plan never builds this combination)
        raise ValueError("linear: dropout needs relu=True (the kernels' dropout mask is the sign of the saved output)")
    return _LinearFn.apply(x, W, b, bool(relu), float(drop_p), seed, int(site))


def mlp_plan(modules):
    """[(nn.Linear, relu?, dropout p)] when `modules` (an nn.Sequential's children) is a chain of Linear [-> ReLU] [-> Dropout]
    groups -- what MLP_Block builds without norms / Dice -- else None (the caller keeps torch's modules)"""
    plan, i, mods = [], 0, list(modules)
    while i < len(mods):
        if type(mods[i]) is not nn.Linear:
            return None
        lin, relu, p = mods[i], False, 0.0
        i += 1
        if i < len(mods) and type(mods[i]) is nn.ReLU:
            relu, i = True, i + 1
        if i < len(mods) and type(mods[i]) is nn.Dropout:
            if not relu:
                return None  # the kernels' mask is the saved output: dropout without ReLU would lose the sign information
            p, i = float(mods[i].p), i + 1
        plan.append((lin, relu, p))
    return plan or None


class _MlpFn(torch.autograd.Function):
    """a whole mlp_plan as ONE autograd node: forward = one rc_linear_fwd per layer; backward = rc_linear_bwd_chain per layer,
    top down, where the dX product of layer i + 1 applies the ReLU / dropout mask of layer i in its epilogue (the saved output of
    layer i is the input of layer i + 1), so no layer but the top one makes a separate masking pass over [M, width]."""

    @staticmethod
    def _tail(spec, params, M):
        """the last hidden layer + a width -> 1 output layer as the two tower-tail kernels (csrc/tower_tail.hip)?"""
        n = len(spec)
        if n < 2:
            return False
        (relu2, _, _), (relu3, p3, _) = spec[n - 2], spec[n - 1]
        W2, W3 = params[2 * (n - 2)], params[2 * (n - 1)]
        return bool(relu2 and not relu3 and p3 == 0.0 and W3.shape[0] == 1 and engine.tower_tail_supported(M, W2.shape[1], W2.shape[0]))

    @staticmethod
    def forward(ctx, x, spec, seed, *params):
        # spec: tuple of (relu, drop_p, has_bias) per layer; params: W_0, b_0 | None, W_1, ...
        xs = x.detach().reshape(-1, x.shape[-1]).contiguous()
        saved, h = [], xs
        n = len(spec)
        tail = _MlpFn._tail(spec, params, xs.shape[0])
        for site, (relu, p, has_b) in enumerate(spec):
            W = params[2 * site].detach().contiguous()
            b = params[2 * site + 1].detach().contiguous() if has_b else None
            if tail and site == n - 2:
                # one kernel for this layer and the output layer: H2 (saved: the output layer's input and this layer's mask), z
                W3 = params[2 * (n - 1)].detach().contiguous()
                b3 = params[2 * (n - 1) + 1].detach().contiguous() if spec[n - 1][2] else None
                if all(t.data_ptr() % 16 == 0 for t in (h, W, W3)):   # (float4 operand loads; an offset view keeps the GEMM route)
                    H2, z = engine.tower_tail_fwd(h, W, b, W3, b3, p, seed, site)
                    saved += [h, W, H2, W3]
                    h = z
                    break
                tail = False
            y = engine.linear_fwd(h, W, b, relu, p, seed, site)
            saved += [h, W]
            h = y
        ctx.save_for_backward(*saved, h)
        ctx.spec, ctx.xshape, ctx.need_dx, ctx.tail = spec, x.shape, x.requires_grad, tail
        return h.view(*x.shape[:-1], h.shape[-1])

    @staticmethod
    def backward(ctx, dy):
        spec = ctx.spec
        *saved, y_top = ctx.saved_tensors
        n = len(spec)
        dz = dy.reshape(-1, dy.shape[-1]).contiguous()
        grads = [None] * (2 * n)
        top = n - 1
        if ctx.tail:
            X2, W2, H2, W3 = saved[2 * (n - 2)], saved[2 * (n - 2) + 1], saved[2 * (n - 1)], saved[2 * (n - 1) + 1]
            below = spec[n - 3] if n > 2 else None
            x_act = below is not None and (below[0] or below[1] > 0)
            dX, dW2, db2, dW3, db3 = engine.tower_tail_bwd(X2, W2, W3, H2, dz, spec[n - 2][1], need_dx=(n > 2 or ctx.need_dx), x_act=x_act,
                                                           x_drop_p=below[1] if x_act else 0.0, need_db2=spec[n - 2][2], need_db3=spec[n - 1][2])
            grads[2 * (n - 2)], grads[2 * (n - 2) + 1], grads[2 * (n - 1)], grads[2 * (n - 1) + 1] = dW2, db2, dW3, db3
            dz, top = dX, n - 3
        # (the weight-gradient products on a second stream beside the dX chain -- rc_linear_bwd_chain takes dX / dW separately --
        #  measured SLOWER inside the replayed graph at B = 1,024: 0.389 against 0.377 ms, profiles/r05h_gemm_probe.txt)
        for i in range(top, -1, -1):
            X, W = saved[2 * i], saved[2 * i + 1]
            relu, p, has_b = spec[i]
            # the top layer masks its own dY (Y = its saved output); every other layer receives a dY that the layer above
            # already masked in its dX product (x_act)
            y = y_top if (i == n - 1 and (relu or p > 0)) else None
            below = spec[i - 1] if i > 0 else None
            x_act = below is not None and (below[0] or below[1] > 0)
            need_dx = i > 0 or ctx.need_dx
            dX, dW, db = engine.linear_bwd(X, W, y, dz, p, need_dx=need_dx, need_db=has_b, x_act=x_act,
                                           x_drop_p=below[1] if x_act else 0.0)
            grads[2 * i], grads[2 * i + 1] = dW, db
            dz = dX
        return (dz.view(ctx.xshape) if dz is not None else None, None, None, *grads)


def mlp_forward(x, plan, training, seed):
    """run a mlp_plan (Linear [-> ReLU] [-> Dropout] groups) as one autograd node (dropout only in training mode; `seed` bumped
    by the caller)"""
    if not x.is_cuda:
        raise RuntimeError("mlp_forward runs on the GPU only (no CPU path)")
    spec = tuple((bool(relu), float(p) if training else 0.0, lin.bias is not None) for lin, relu, p in plan)
    params = []
    for lin, _, _ in plan:
        params += [lin.weight, lin.bias]
    return _MlpFn.apply(x, spec, seed, *params)


class HipOptimizer:
    """torch.optim.{SGD,Adam,Adagrad,Adadelta} semantics (dense, weight decay per param group) executed by
    rc_dense_update.  Built by the runner in place of `eval('torch.optim.X')`
    (helpers/BaseRunner.py:110-114); accepts the same param-group list
    (`model.customize_parameters()`, models/BaseModel.py:64-73)."""

    def __init__(self, param_groups, name="Adam", lr=1e-3, weight_decay=0.0, capturable=False):
        if name not in ("SGD", "Adam", "Adagrad", "Adadelta"):
            raise ValueError(f"HipOptimizer: optimizer {name!r} not built (SGD, Adam, Adagrad, Adadelta)")
        self.name, self.lr = name, lr
        # capturable: Adam's step count lives on the device, so step() can be recorded in a hipGraph
        self.capturable = capturable
        self._step_dev = None
        self.param_groups = []
        for g in param_groups:
            g = dict(g)
            g.setdefault("lr", lr)
            g.setdefault("weight_decay", weight_decay)
            g["params"] = list(g["params"])
            self.param_groups.append(g)
        self.state = {}
        self.step_count = 0
        self._rows = None           # the rows-mode step in flight (rows_begin .. step)
        self._rows_state = None     # its persistent buffers: row flags, the touched rows' gradient scratch

    # ---- rows mode: Adam over embedding tables of which a small batch touches a fraction of a percent ----------------------
    # torch.optim.Adam on a dense gradient updates every row every step (helpers/BaseRunner.py:110-114,206), so the pass over
    # tables and state cannot be skipped -- but the dense GRADIENT can.  At batch_size 1024 DeepFM's tables are 71 MB of which
    # 10 K rows have a gradient: the step zero-filled 71 MB, wrote row sums into it and read all of it back.  Here the gather
    # stamps the rows it looks up with the step's number, the backward pass writes the row sums into a scratch that is never
    # cleared, and the update reads a gradient only where the stamp matches (g = 0 elsewhere): rc_dense_update_rows_dev,
    # bit-identical to the dense step.  (Splitting the pass -- unstamped rows on a second stream beside forward / backward,
    # stamped rows at the end -- was measured and lost: profiles/r08_rows_adam_split_vs_single.txt.)
    def rows_ok(self):
        return self.name == "Adam" and self.capturable and os.environ.get("RC_ROWS_ADAM", "1") != "0"

    def _hyper_of(self, p):
        for g in self.param_groups:
            if any(q is p for q in g["params"]):
                return engine.make_hyper(self.name, lr=g["lr"], l2=g["weight_decay"], step=max(self.step_count, 1))
        return None

    def rows_begin(self, tables, tables1):
        """-> (row flags, step count on the device) for the gather to stamp, or None (rows mode not available: dense step)"""
        if not self.rows_ok():
            return None
        if self._rows is not None:
            raise RuntimeError("HipOptimizer rows mode: the previous training forward was not followed by backward + step()")
        params = list(tables) + list(tables1)
        hypers = [self._hyper_of(p) for p in params]
        if any(h is None for h in hypers):
            return None
        dev = params[0].device
        key = tuple(id(p) for p in params)
        st = self._rows_state
        capturing = torch.cuda.is_current_stream_capturing()    # (buffers are created by an eager step)
        if st is None or st["key"] != key:
            if capturing:
                return None
            n_rows, d = sum(t.shape[0] for t in tables), tables[0].shape[1]
            st = self._rows_state = {"key": key, "flags": torch.zeros(n_rows, dtype=torch.int32, device=dev),
                                     "G": torch.empty(n_rows * (d + 1), dtype=torch.float32, device=dev)}
        if self._step_dev is None:
            if capturing:
                return None
            self._step_dev = torch.full((1,), self.step_count, dtype=torch.int64, device=dev)
        for p in params:
            ps = self.state.setdefault(p, {})
            for k in ("m", "v"):
                if k not in ps:
                    if capturing:
                        return None
                    ps[k] = torch.zeros_like(p)
        offs, run = [], 0
        for t in tables:
            offs.append(run)
            run += t.shape[0]
        offs.append(run)
        self._rows = {"params": params, "hypers": hypers, "offs": offs, "F": len(tables), "G": None}
        # step() increments the count in front of its update launch; nothing between this gather (which reads count + 1) and that
        # launch reads it, so a tower's dropout-seed increment may take it along (engine.defer_increment)
        engine.defer_increment(self._step_dev)
        return st["flags"], self._step_dev

    def _rows_items(self, Gs):
        r, st = self._rows, self._rows_state
        F, offs, flags = r["F"], r["offs"], st["flags"]
        items = []
        for i, (p, h) in enumerate(zip(r["params"], r["hypers"])):
            f = i % F
            ps = self.state[p]
            items.append((p.data, Gs[i // F][offs[f]:offs[f + 1]], h, ps["m"], ps["v"], flags[offs[f]:offs[f + 1]], p.shape[1]))
        return items

    def rows_abort(self):
        """forget a rows-mode step that did not reach step() (the caller's step raised).  The tables took no update, but the gather
        stamped its rows with the number the NEXT step will carry too (the count was not incremented): the stamps are cleared, or
        rows of the abandoned batch would read stale gradient rows"""
        if self._rows is not None:
            self._rows = None
            folded = engine.take_deferred(self._step_dev)
            if self._rows_state is not None and not torch.cuda.is_current_stream_capturing():
                self._rows_state["flags"].zero_()
                if folded:      # the count went up with a tower's seed although no update followed: the step is forgotten
                    self._step_dev.sub_(1)

    def rows_scratch(self):
        return self._rows_state["G"]

    def rows_grads(self, Gv, Gl):
        self._rows["G"] = (Gv, Gl)

    def zero_grad(self, set_to_none=True):
        for g in self.param_groups:
            for p in g["params"]:
                p.grad = None

    @torch.no_grad()
    def step(self):
        """every parameter with a gradient in ONE rc_dense_update_multi call (per 36 tensors)"""
        items = []
        dev = None
        rows = self._rows
        if rows is not None and rows["G"] is None:
            raise RuntimeError("HipOptimizer rows mode: step() without the backward pass of the training forward")
        if rows is not None:      # (validated before anything is counted: a refused step leaves the optimizer as it was)
            mine = {id(q) for q in rows["params"]}
            if any(p.grad is not None and id(p) in mine for g in self.param_groups for p in g["params"]):
                raise RuntimeError("HipOptimizer rows mode: a table of the rows-mode gather also received a dense gradient")
        self.step_count += 1
        for g in self.param_groups:
            h = engine.make_hyper(self.name, lr=g["lr"], l2=g["weight_decay"], step=max(self.step_count, 1))
            for p in g["params"]:
                if p.grad is None:
                    continue
                st = self.state.setdefault(p, {})
                m = v = None
                # (state created once: `st.setdefault(k, torch.zeros_like(p))` would allocate and zero-fill a tensor of the
                # parameter's size on EVERY step just to throw it away -- 48 fill kernels per DeepFM step, 24 % of its
                # replayed GPU time in profiles/r02z_deepfm_kernel_stats.csv)
                if self.name in ("Adam", "Adagrad", "Adadelta"):   # exp_avg / state_sum / square_avg
                    if "m" not in st:
                        st["m"] = torch.zeros_like(p)
                    m = st["m"]
                if self.name in ("Adam", "Adadelta"):              # exp_avg_sq / acc_delta
                    if "v" not in st:
                        st["v"] = torch.zeros_like(p)
                    v = st["v"]
                grad = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                items.append((p.data, grad, h, m, v))
                dev = p.device
        if rows is not None:
            # the tables (gradient rows where stamped, g = 0 elsewhere) and every other parameter in one launch
            everything = [(w, gr, h, m, v, None, 0) for w, gr, h, m, v in items] + self._rows_items(rows["G"])
            if not engine.take_deferred(self._step_dev):      # (else a tower's seed increment took the count along)
                engine.step_increment(self._step_dev)
            engine.dense_update_rows(everything, self._step_dev, touched=2)
            self._rows = None
            return
        if self.capturable and items:
            if self._step_dev is None:  # (allocated outside any capture: the first step runs eagerly)
                self._step_dev = torch.full((1,), self.step_count - 1, dtype=torch.int64, device=dev)
            engine.dense_update_multi(items, self.name, step_dev=self._step_dev)
        else:
            engine.dense_update_multi(items, self.name)
