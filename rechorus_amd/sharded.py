"""Row-sharded BPRMF training across W GPUs (one process per GPU, torch.distributed over RCCL/xGMI).

The reference is single-device; this layer is new.  Tables are sharded by row,
owner(id) = id mod W, local row = id div W (hashing the Zipf head across ranks).  A tuple touches
1+K item rows but only ONE user row, and xGMI is per-link bound (~64 GB/s per direction per peer),
so rows are not moved to the tuples (25.6 KB per tuple at K=99, d=64); the tuples' user rows are
moved to the item rows' owners instead ("owner computes"):

  1. fetch   user rows of the local batch from their owners        all_to_all  (ids, rows)
  2. share   all_gather the batch's user rows  Uall [W*B, d]        all_gather  (256 B / tuple / peer)
  3. route   every candidate occurrence (tuple index, local row) to the item's owner   all_to_all
  4. score   owner: s = <Uall[t], I_loc[row]>;  scores return home   all_to_all  (4 B / occurrence)
  5. loss    home: BPR loss + g = dL/ds on its own tuples; loss all_reduce;  g goes back to owners
  6. update  owner: item rows  += opt(sum_occ g*Uall[t])  (segmented, atomic-free, local);
             partial user grads  PUG[t] = sum_{occ at this owner} g*I_loc[row]
  7. reduce  reduce_scatter PUG -> ugrad of the home's tuples;  route to the user rows' owners,
             owner applies the segmented update.

≈5 KB of traffic per tuple instead of ≈45 KB for moving rows both ways at K = 99, W = 8.  The all_gather /
reduce_scatter pair costs 2 (W-1) d 4 bytes per tuple whatever K is, though, so with FEW candidates per tuple
(BASELINE's headline configuration has K = 1) moving the rows is the cheaper plan and the step switches to it
(`mode="auto"`, ShardedBprmf._plan):

  rows travel:  ids -> owners, owners gather and return the 1 + (1+K) rows of every tuple; home runs the fused
                gather-dot / loss / backward kernel on the compact row blocks; per-occurrence row gradients
                return along the same routes; owners apply the segmented update.   2 (2+K) d 4 bytes / tuple,
                independent of W.

With W = 1 every collective is the identity and the arithmetic equals engine.BprmfTrainer's (same kernels).

Local arithmetic goes through an `ops` object: `HipOps` (the C ABI, default) on GPUs; the CPU tests
inject an oracle-backed implementation so the routing can be verified with gloo, world_size 2.
"""
import numpy as np
import torch
import torch.distributed as dist


class _SortPlan:
    """The interface of engine.Plan the sharded steps use -- row_sums / update on list "a", .single -- on the radix sort +
    segmented sum (rc_sort_ids, rc_segmented_update): for id lists the bucket plan has no geometry for and row widths without a
    plan kernel.  Same results up to the summation order inside a row (both fixed)."""

    def __init__(self, e, ids, n_rows, want_single=False):
        self.e, self.n_a, self.n_b = e, ids.numel(), 0
        self.keys, self.perm = e.sort_ids(ids.reshape(-1), n_rows)
        self.single = None
        if want_single:   # uint8 per position: 1 = the row occurs once in the list (updated by the caller, skipped here)
            self.single, _, _ = e.segment_heads(self.keys, self.perm, want_single=True, want_heads=False)

    def _src(self, coef, src, src_index, div, src2):
        if src2 is not None:
            return dict(src=src2)
        return dict(src=src, coef=coef, src_index=src_index, div=div)

    def row_sums(self, side, out, coef=None, src=None, src_index=None, div=1, src2=None):
        self.e.segmented_update(self.keys, self.perm, dense_grad=out, **self._src(coef, src, src_index, div, src2))
        return out

    def update(self, side, W, hyper, m=None, v=None, coef=None, src=None, src_index=None, div=1, src2=None):
        self.e.segmented_update(self.keys, self.perm, hyper=hyper, W=W, m=m, v=v, skip_singletons=self.single is not None,
                                **self._src(coef, src, src_index, div, src2))


class HipOps:
    """local kernels of the sharded step, all through librechorus_hip.so"""

    def __init__(self):
        from . import engine
        self.e = engine

    def gather_rows_pair(self, Wa, Wb, rows):
        """(Wa[rows] | Wb[rows]) as one block: two tables that share the row ids (rc_gather_rows_pair)"""
        if rows.numel() == 0:
            return torch.empty((0, 2 * Wa.shape[1]), dtype=torch.float32, device=Wa.device)
        return self.e.gather_rows_pair(Wa, Wb, rows.contiguous())

    def gather_rows(self, W, rows):
        return self.e.gather_rows(W, rows)

    def dot_rows(self, Uall, t_idx, I_loc, rows):
        return self.e.gather_dot(Uall, I_loc, t_idx, rows.reshape(-1, 1)).reshape(-1)

    def bpr_loss(self, pred, inv_b):
        _, loss_vec, g = self.e.bpr_loss(pred, inv_b=inv_b)
        return loss_vec, g

    # Grouping occurrences by row is the bucket plan wherever it has a geometry and a kernel for the row width (engine.Plan:
    # id-range or hashed buckets, no device-wide sort, no host round trip); lists beyond that (more than ~8.4 M ids under a hashed
    # geometry, widths other than 16 / 32 / 64 / 128 / 256) take the radix sort + segmented sum behind the same interface
    # (_SortPlan).  Tags keep the cached buffers of plans that are alive together apart.
    def _grouping(self, ids, n_rows, d, tag, list_single_a=True):
        if d in (16, 32, 64, 128, 256) and self.e.plan_supported(ids.numel(), 0, n_rows, 0):
            return self.e.Plan(ids, n_rows, tag=tag, list_single_a=list_single_a)
        return _SortPlan(self.e, ids, n_rows, want_single=not list_single_a)

    def partial_user_grads(self, I_loc, rows, g, t_idx, n_tuples):
        out = torch.zeros((n_tuples, I_loc.shape[1]), dtype=torch.float32, device=I_loc.device)
        if t_idx.numel():
            self._grouping(t_idx, n_tuples, I_loc.shape[1], "pug").row_sums("a", out, coef=g, src=I_loc, src_index=rows, div=1)
        return out

    def sum_rows_by_index(self, rows, index, n_out):
        """out[k] = sum of rows[p] over p with index[p] = k, in ascending p (atomic-free segmented sum)"""
        out = torch.zeros((n_out, rows.shape[1]), dtype=torch.float32, device=rows.device)
        if index.numel():
            self._grouping(index, n_out, rows.shape[1], "srbi").row_sums("a", out, src2=rows.contiguous())
        return out

    def unique(self, ids, n_rows):
        """distinct ids (in the plan's order) + inverse index; reads the count back (the sharded steps call it one step
        ahead, on their side stream)"""
        return self.e.unique_ids(ids, n_rows, tag="route.unique")

    def unique_many(self, lists):
        """unique() of several (ids, n_rows) lists with ONE host synchronisation for all their counts (a sync per list otherwise:
        eight per sharded NeuMF step, each a wait for the side stream's small kernels)"""
        begun = [self.e.unique_ids_begin(ids, n_rows, tag="route.unique") for ids, n_rows in lists]
        if any(b is None for b in begun):
            return [self.unique(ids, n_rows) for ids, n_rows in lists]
        counts = torch.cat([b[2] for b in begun]).tolist()       # the one device -> host copy
        return [(b[0][:int(c)], b[1]) for b, c in zip(begun, counts)]

    def prepare_rows(self, rows, n_rows, tag="rows", d=64):
        """grouping of a row-id list (bucket plan, or the sorted route where the plan has no geometry / no kernel for width d),
        reusable by several update_rows calls on tables that share the ids"""
        if rows.numel() == 0:
            return None
        return self._grouping(rows, n_rows, d, tag)

    def update_rows(self, W, state, rows, src, hyper, coef=None, src_index=None, prep=None):
        """W[r] <- opt(W[r], sum_{o: rows[o]=r} coef[o] * src[src_index[o] or o])"""
        if rows.numel() == 0:
            return
        plan = prep if prep is not None else self.prepare_rows(rows, W.shape[0], d=W.shape[1])
        if coef is None and src_index is None:
            plan.update("a", W, hyper, m=state.get("m"), v=state.get("v"), src2=src)
        else:
            plan.update("a", W, hyper, m=state.get("m"), v=state.get("v"), coef=coef, src=src, src_index=src_index, div=1)

    def update_rows_pair(self, Wa, Wb, sa, sb, rows, src_a, src_b, hyper, prep):
        """two tables that share `rows` (NeuMF's mf / mlp embeddings) in one pass; False if the width has no pair kernel.
        src_b=None: src_a is the [n, 2 d] block of both gradients side by side, read where it lies"""
        if rows.numel() == 0:
            return True
        if not self.e.segmented_pair_supported(Wa.shape[1]) or isinstance(prep, _SortPlan):
            return False
        prep.update_pair("a", Wa, Wb, src_a, src_b, hyper, ma=sa.get("m"), va=sa.get("v"), mb=sb.get("m"), vb=sb.get("v"))
        return True

    pair_block_updates = True     # update_rows_pair takes the (d a | d b) block itself

    # ---- csrc/owner_step.hip
    def route(self, ids, world, tuple_base=None, div=1):
        """stable grouping by owner -> (order int32 [n], counts int64 [world] (device), payload int64 [n]);
        payload = local rows (users) or (tuple << 32 | local row) messages (items, tuple_base given)"""
        return self.e.route_by_owner(ids, world, tuple_base, div)

    def unpack(self, recv):
        return self.e.owner_unpack(recv)

    def prepare_owner(self, rows, n_rows):
        """bucket plan of the received rows: singleton flags + the multi-occurrence rows listed (independent of the
        scores: issued while they travel)"""
        return self._grouping(rows, n_rows, 64, "owner", list_single_a=False)   # (rc_owner_backward itself: widths with a kernel)

    def owner_backward(self, I_loc, state, rows, g, t_idx, t32, Uall, n_tuples, hyper, prep):
        """partial user grads + the item-row update in two passes over the received occurrences"""
        if not self.e.owner_backward_supported(I_loc.shape[1]):
            pug = self.partial_user_grads(I_loc, rows, g, t_idx, n_tuples)
            self.update_rows(I_loc, state, rows, Uall, hyper, coef=g, src_index=t_idx)
            return pug
        pug = self.e.owner_backward(I_loc, state.get("m"), state.get("v"), Uall, t32, rows, g, prep.single, n_tuples, hyper)
        prep.update("a", I_loc, hyper, m=state.get("m"), v=state.get("v"), coef=g, src=Uall, src_index=t_idx, div=1)
        return pug

    def owner_backward_split(self, I_loc, state, rows, g, t_idx, t32, Uall, n_tuples, hyper, prep):
        """owner_backward in two parts: -> (pug, finish).  The partial user gradients are complete on return (they
        go into the reduce_scatter at once); finish() runs the segmented update of the multi-occurrence item rows,
        which the caller issues while that collective is in flight."""
        if not self.e.owner_backward_supported(I_loc.shape[1]):
            pug = self.partial_user_grads(I_loc, rows, g, t_idx, n_tuples)
            return pug, lambda: self.update_rows(I_loc, state, rows, Uall, hyper, coef=g, src_index=t_idx)
        pug = self.e.owner_backward(I_loc, state.get("m"), state.get("v"), Uall, t32, rows, g, prep.single, n_tuples, hyper)

        def finish():
            prep.update("a", I_loc, hyper, m=state.get("m"), v=state.get("v"), coef=g, src=Uall, src_index=t_idx, div=1)
        return pug, finish

    # ---- NeuMF head on rows that were moved to the tuples (ShardedNeumf) ------------------------------
    def neumf_fwd(self, P, uid, iid):
        return self.e.neumf_fwd(P, uid, iid)

    def neumf_bwd(self, P, uid, iid, gpred):
        return self.e.neumf_bwd(P, uid, iid, gpred)

    def neumf_head(self, urows, irows, P, B, C, inv_b):
        """forward + BPR loss + backward of the head on the fetched row blocks [rows, mf | mlp] in ONE kernel
        (rc_neumf_head_fwd_bwd) -> (loss_vec, gu [B, 2d], gi [B C, 2d], dense grads), or None where the fused kernel has no
        instance (the caller then runs neumf_fwd / bpr_loss / neumf_bwd)"""
        d, l1 = urows.shape[1] // 2, P["W1"].shape[0]
        if not self.e._NEUMF_FUSED or C < 2 or not self.e.neumf_train_step_supported(C, d, l1):
            return None
        loss_vec, gu, gi, dense, _ = self.e.neumf_head_fwd_bwd(urows, irows, P["W1"], P["b1"], P["w_out"], B, C, inv_b)
        return loss_vec, gu, gi, dense

    # ---- owner-computed item half of the NeuMF hidden layer (ShardedNeumf, item_half = "owner") -----------------
    def item_half_fwd(self, mlp_rows, W1i):
        """zi [n, l1] = mlp_rows [n, d] W1i^T: what travels to the tuples instead of the mlp_i rows (rc_linear_fwd, no bias)"""
        return self.e.linear_fwd(mlp_rows, W1i)

    def item_half_bwd(self, mlp_rows, W1i, dz):
        """-> (d mlp_i [n, d] = dz W1i, this rank's share of dW1i [l1, d] = dz^T mlp_rows)  (rc_linear_bwd)"""
        if mlp_rows.shape[0] == 0:
            return torch.zeros_like(mlp_rows), torch.zeros_like(W1i)
        dX, dW, _ = self.e.linear_bwd(mlp_rows, W1i, None, dz, need_db=False, ws_tag="item_half_bwd")
        return dX, dW

    def neumf_zhead(self, urows, irows, P, B, C, inv_b):
        """the head on (mf_u | mlp_u) user blocks and (mf_i | zi) item blocks (csrc/neumf_zhead.hip + the user half's GEMMs)
        -> (loss_vec, gu [B, 2d], gi [B C, d + l1] = (d mf_i | dz), {"W1u", "b1", "w_out"}) or None (shape not covered)"""
        d, l1 = urows.shape[1] // 2, P["W1"].shape[0]
        if not self.e.neumf_zhead_supported(C, d, l1):
            return None
        loss_vec, gu, gi, dense, _ = self.e.neumf_zhead(urows, irows, P["W1"], P["b1"], P["w_out"], B, C, inv_b)
        return loss_vec, gu, gi, dense

    def dense_update(self, W, G, hyper, state):
        self.e.dense_update(W, G, hyper, state.get("m"), state.get("v"))

    # ---- BPRMF on rows that were moved to the tuples (ShardedBprmf, mode "rows") ----------------------
    def rows_fwd_bwd(self, Ub, Ib, pos_u, pos_i, inv_b):
        """fused gather-dot + BPR loss + backward on compact row blocks -> (loss_vec [B], g [B,C], ugrad [B,d])"""
        _, loss_vec, g, ugrad = self.e.bprmf_fwd_bwd(Ub, Ib, pos_u, pos_i, inv_b=inv_b, want_pred=False)
        return loss_vec, g, ugrad

    def scaled_rows(self, src, idx, coef):
        """out[o] = coef[o] * src[idx[o]]  (the item-row gradient g * u of every occurrence, in send order)"""
        n = idx.numel()
        return self.e.weighted_row_sum(src, idx.reshape(n, 1), coef.reshape(n, 1))

    def make_hyper(self, **kw):
        return self.e.make_hyper(**kw)

    def new_state(self, W, opt):
        st = {}
        if opt in ("Adam", "Adagrad"):
            st["m"] = torch.zeros_like(W)
        if opt == "Adam":
            st["v"] = torch.zeros_like(W)
        return st


def _is_nccl(group):
    return dist.get_backend(group) == "nccl"


def _all_to_all_v(out, inp, out_splits, in_splits, group):
    """all_to_all_single with split sizes.  RCCL does it natively; gloo (CPU tests) has no
    all_to_all, so it is emulated there with point-to-point sends."""
    if _is_nccl(group):
        dist.all_to_all_single(out, inp, output_split_sizes=out_splits, input_split_sizes=in_splits, group=group)
        return
    if inp.is_cuda:  # gloo moves host memory: stage through the CPU (test configurations only)
        out_h = torch.empty(out.shape, dtype=out.dtype)
        _all_to_all_v(out_h, inp.cpu(), out_splits, in_splits, group)
        out.copy_(out_h)
        return
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    ins = list(torch.split(inp, in_splits)) if inp.shape[0] else [inp[:0]] * world
    outs = list(torch.split(out, out_splits)) if out.shape[0] else [out[:0]] * world
    reqs = []
    for peer in range(world):
        if peer == rank:
            outs[peer].copy_(ins[peer])
            continue
        if in_splits[peer]:
            reqs.append(dist.isend(ins[peer].contiguous(), peer, group=group))
    for peer in range(world):
        if peer != rank and out_splits[peer]:
            buf = torch.empty_like(outs[peer])
            dist.recv(buf, peer, group=group)
            outs[peer].copy_(buf)
    for r in reqs:
        r.wait()


def _exchange(send, send_counts, group=None, recv_counts=None):
    """variable all_to_all along dim 0: `send` is grouped by destination (send_counts per rank).
    Returns (recv, recv_counts).  One host sync for the counts unless recv_counts is given."""
    world = dist.get_world_size(group)
    if recv_counts is None:
        sc = torch.as_tensor(send_counts, dtype=torch.int64, device=send.device)
        rc = torch.empty_like(sc)
        _all_to_all_v(rc, sc, [1] * world, [1] * world, group)
        recv_counts = rc.tolist()
    recv = torch.empty((sum(recv_counts),) + tuple(send.shape[1:]), dtype=send.dtype, device=send.device)
    _all_to_all_v(recv, send.contiguous(), list(recv_counts), list(send_counts), group)
    return recv, recv_counts


def _exchange_async(send, send_counts, recv_counts, group=None, out=None):
    """_exchange with known split sizes whose transfer may stay in flight (RCCL): -> _Pending of the received rows.
    Other backends (the gloo emulation of the CPU tests) complete it on the spot.
    out: where the rows are received (a contiguous slice of a larger buffer the caller assembles from several exchanges)"""
    shape = (sum(recv_counts),) + tuple(send.shape[1:])
    if out is not None and (tuple(out.shape) != shape or out.dtype != send.dtype or not out.is_contiguous()):
        raise ValueError("_exchange_async: out does not fit the rows to receive")
    recv = out if out is not None else torch.empty(shape, dtype=send.dtype, device=send.device)
    if _is_nccl(group):
        work = dist.all_to_all_single(recv, send.contiguous(), output_split_sizes=list(recv_counts),
                                      input_split_sizes=list(send_counts), group=group, async_op=True)
        return _Pending(recv, work)
    _all_to_all_v(recv, send.contiguous(), list(recv_counts), list(send_counts), group)
    return _Pending(recv)


class _Counts:
    """split sizes of several exchanges, travelling: the tiny all_to_all is enqueued at construction, the device -> host
    copy is asynchronous (pinned buffer + event), get() waits for THAT event only.  A step that prepares the NEXT batch's
    routes before it enqueues its own heavy kernels (`next_batch=` of the sharded steps) finds the sizes on the host when
    the next step starts, so the host never waits for the device to drain and keeps enqueueing one step ahead."""

    def __init__(self, count_tensors, group=None):
        world = dist.get_world_size(group)
        self.k = len(count_tensors)
        sc = torch.stack([c.to(torch.int64) for c in count_tensors], dim=1).contiguous()  # [world, k]
        rc = torch.empty_like(sc)
        _all_to_all_v(rc, sc, [1] * world, [1] * world, group)
        both = torch.cat([sc, rc], dim=1)
        self.event = None
        if both.is_cuda:
            self.host = torch.empty(both.shape, dtype=both.dtype, pin_memory=True)
            self.host.copy_(both, non_blocking=True)
            self.event = torch.cuda.Event()
            self.event.record()
        else:
            self.host = both
        self._lists = None

    def get(self):
        """-> ([send lists], [recv lists]); blocks only until the copy of these sizes has landed"""
        if self._lists is None:
            if self.event is not None:
                self.event.synchronize()
            both, k = self.host.tolist(), self.k
            self._lists = ([[row[j] for row in both] for j in range(k)], [[row[k + j] for row in both] for j in range(k)])
        return self._lists


def _exchange_counts(count_tensors, group=None):
    """one all_to_all for the split sizes of several exchanges -> ([send lists], [recv lists]); ONE host sync"""
    return _Counts(count_tensors, group).get()


def _exchange_back(payload, recv_counts, send_counts, group=None):
    """reverse direction of a previous _exchange: payload is ordered like what was received"""
    out, _ = _exchange(payload, recv_counts, group, recv_counts=send_counts)
    return out


def _all_gather_rows(x, world, group):
    out = torch.empty((world * x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
    if _is_nccl(group):
        dist.all_gather_into_tensor(out, x.contiguous(), group=group)
    else:
        xh = x.cpu().contiguous()
        parts = [torch.empty_like(xh) for _ in range(world)]
        dist.all_gather(parts, xh, group=group)
        out.copy_(torch.cat(parts, dim=0))
    return out


class _Pending:
    """a collective that may still be in flight on RCCL's stream: wait() makes the current stream wait for it"""

    def __init__(self, out, work=None):
        self.out, self.work = out, work

    def wait(self):
        if self.work is not None:
            self.work.wait()
            self.work = None
        return self.out


def _all_gather_rows_async(x, world, group):
    if not _is_nccl(group):
        return _Pending(_all_gather_rows(x, world, group))
    out = torch.empty((world * x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
    return _Pending(out, dist.all_gather_into_tensor(out, x.contiguous(), group=group, async_op=True))


def _reduce_scatter_rows_async(x, world, group):
    if not _is_nccl(group):
        return _Pending(_reduce_scatter_rows(x, world, group))
    out = torch.empty((x.shape[0] // world,) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
    return _Pending(out, dist.reduce_scatter_tensor(out, x.contiguous(), group=group, async_op=True))


def _all_reduce_sum(x, group):
    if _is_nccl(group) or not x.is_cuda:
        dist.all_reduce(x, group=group)
        return x
    h = x.cpu()
    dist.all_reduce(h, group=group)
    x.copy_(h)
    return x


def _reduce_scatter_rows(x, world, group):
    """sum over ranks of x [world*n, ...]; rank r keeps rows [r*n, (r+1)*n)"""
    n = x.shape[0] // world
    if _is_nccl(group):
        out = torch.empty((n,) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
        dist.reduce_scatter_tensor(out, x.contiguous(), group=group)
        return out
    y = _all_reduce_sum(x.clone(), group)
    r = dist.get_rank(group)
    return y[r * n:(r + 1) * n].clone()


def ops_unique(ops, ids, n_rows):
    """(distinct ids, inverse index) through the ops object: HipOps = bucket plan + rc_plan_distinct (distinct ids in the
    plan's order); the CPU tests' oracle ops bring their own.  Nothing in this module sorts."""
    return ops.unique(ids, n_rows)


def _record_stream(obj, stream):
    """tensors of a prepared-routes structure were allocated on the side stream and are consumed on `stream`"""
    if torch.is_tensor(obj):
        if obj.is_cuda:
            obj.record_stream(stream)
    elif isinstance(obj, dict):
        for v in obj.values():
            _record_stream(v, stream)
    elif isinstance(obj, (list, tuple)):
        for v in obj:
            _record_stream(v, stream)
    elif type(obj).__name__ == "_Route":      # (routes completed ahead of their step: ids exchanged on the side stream)
        _record_stream(vars(obj), stream)


class _Timed:
    """per-phase cuda events of the last step (self.timing = [] switches it on; bench.py prints them as sharded_phases_ms)"""
    timing = None

    def _mark(self, label):
        if self.timing is not None:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            self.timing.append((label, ev))

    def timing_ms(self):
        """per-phase milliseconds of the last step; synchronises.  Phases that occur several times (micro-batches) add up."""
        torch.cuda.synchronize()
        t, out = self.timing, {}
        for i in range(1, len(t)):
            out[t[i][0]] = out.get(t[i][0], 0.0) + t[i - 1][1].elapsed_time(t[i][1])
        return out


class _LookAhead(_Timed):
    """Shared by the sharded steps: `step(batch, next_batch=...)` routes the NEXT batch before this step's kernels.
    The routing needs host-visible sizes (the number of distinct ids, the split sizes of the exchanges); on the caller's stream
    those device -> host copies would wait for everything enqueued before them, i.e. for the previous step.  So the
    look-ahead runs on a SIDE stream that depends only on the start of the previous step (the next batch's ids must
    exist by then: pooled batches, or a loader running at least one step ahead): its host waits are a few small
    kernels long, and the caller's stream picks the prepared routes up through an event."""

    def _routes_of(self, uid, iid, next_batch):
        ahead, self._ahead = getattr(self, "_ahead", None), None
        if ahead is not None:
            # The announced batch is identified by the tensor OBJECTS (strong references are kept in the record, so
            # an address cannot have been reused by another batch) and their version counters.  Bringing anything
            # else is an error, not a silent re-route: the ranks would disagree about hit / miss and issue different
            # collective sequences (one extra count exchange on the missing rank = a hang).
            au, ai, vu, vi = ahead["batch"]
            if not (au is uid and ai is iid and vu == uid._version and vi == iid._version):
                raise RuntimeError("sharded step: the batch announced with next_batch= has to be the one the next step() "
                                   "brings, unchanged, on every rank (announce nothing if that is not known)")
            prep = ahead
            if prep.get("ready") is not None:
                main = torch.cuda.current_stream(uid.device)
                main.wait_event(prep["ready"])
                _record_stream(prep, main)
        else:
            prep = self._prepare(uid, iid)
            # routed on the caller's stream (first step, or no look-ahead): the side stream's routing of the next batch
            # shares the engine's cached scratch buffers with it and must start after it
            self._mark_step_begin(uid)
        # the NEXT batch is routed when this step's own launches are enqueued (_look_ahead, called at the end of step()): the host
        # side of the routing -- a millisecond of Python and small launches at the config-4 shape -- then runs while the GPU works on
        # this step, instead of in front of its first kernel
        self._next_batch = next_batch if (next_batch is not None and (self.world > 1 or getattr(self, "force_exchange", False))) else None
        return prep

    def _look_ahead(self):
        nb, self._next_batch = getattr(self, "_next_batch", None), None
        if nb is not None:
            self._ahead = self._prepare_ahead(*nb)

    def _prepare_ahead(self, uid, iid):
        if not uid.is_cuda:
            prep = self._prepare(uid, iid)
            prep["batch"] = (uid, iid, uid._version, iid._version)
            return prep
        if getattr(self, "_side", None) is None:
            self._side = torch.cuda.Stream(device=uid.device)
        begin = getattr(self, "_step_begin", None)
        if begin is not None:
            self._side.wait_event(begin)
        with torch.cuda.stream(self._side):
            prep = self._prepare(uid, iid)
            finish = getattr(self, "_finish_routes", None)
            if finish is not None:
                # the id exchanges of the routes too: they depend on the batch alone.  The host waits here for the split sizes -- a few
                # small kernels on this stream, while the main stream still works on the step whose launches were just enqueued
                finish(prep)
            done = torch.cuda.Event()
            done.record(self._side)
        prep["ready"] = done
        prep["batch"] = (uid, iid, uid._version, iid._version)
        return prep

    def _mark_step_begin(self, ref):
        if ref.is_cuda:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(ref.device))
            self._step_begin = ev


class ShardedBprmf(_LookAhead):
    def __init__(self, n_users, n_items, emb_size, opt="SGD", lr=1e-3, l2=0.0, device=None, ops=None,
                 group=None, init_std=0.01, seed=0, force_exchange=False, timing=False, mode="auto", dedup_users=True):
        if mode not in ("auto", "owner", "rows"):
            raise ValueError("ShardedBprmf mode must be auto | owner | rows, got {!r}".format(mode))
        self.mode = mode
        # owner plan: only the DISTINCT user rows of a rank's batch travel (fetch, all_gather, gradient reduction); under
        # Zipf users 27 K of 65 K at the bench shape, i.e. 58 % less on the two largest transfers of the step
        self.dedup_users = bool(dedup_users)
        self.wire = None   # bytes over the links of the last owner-plan step (this rank)
        self.group = group
        self.force_exchange = force_exchange  # W = 1 through the general path (loop-back profiling)
        self.timing = [] if timing else None   # [(label, cuda event)] of the last step
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.n_users, self.n_items, self.d = n_users, n_items, emb_size
        self.ops = ops if ops is not None else HipOps()
        self.opt, self.lr, self.l2 = opt, lr, l2
        self.device = device
        W = self.world
        self.rows_u = (n_users + W - 1) // W
        self.rows_i = (n_items + W - 1) // W
        g = torch.Generator(device=device if device is not None else "cpu")
        g.manual_seed(seed * 1000 + self.rank)
        self.U = torch.empty((self.rows_u, emb_size), device=device).normal_(0, init_std, generator=g)
        self.I = torch.empty((self.rows_i, emb_size), device=device).normal_(0, init_std, generator=g)
        self.sU = self.ops.new_state(self.U, opt)
        self.sI = self.ops.new_state(self.I, opt)
        self.step_count = 0

    # ---- shard <-> global helpers (tests, checkpoints) --------------------------------------
    def load_global(self, U, I):
        """take this rank's rows out of full tables (global row id = local*W + rank)"""
        W, r = self.world, self.rank
        self.U.zero_()
        self.I.zero_()
        u, i = U[r::W], I[r::W]
        self.U[: u.shape[0]].copy_(u)
        self.I[: i.shape[0]].copy_(i)

    def gather_global(self):
        """all ranks -> full [n_users,d], [n_items,d] tables (gather-on-save)"""
        out = []
        for shard, n in ((self.U, self.n_users), (self.I, self.n_items)):
            if self.world == 1:
                out.append(shard[:n].clone())
                continue
            allg = _all_gather_rows(shard, self.world, self.group).view(self.world, shard.shape[0], self.d)
            full = allg.permute(1, 0, 2).reshape(-1, self.d)  # global row l*W + k  <-  rank k, local l
            out.append(full[:n].clone())
        return out

    # ---- one training step ------------------------------------------------------------------------
    def _prepare(self, uid, iid):
        """group the batch's ids by owner for the plan of this shape and START the exchange of the split sizes"""
        B, C = iid.shape
        plan = self._plan(C)
        inv = None
        counts = []
        if plan == "owner" and self.dedup_users:
            uniq, inv = ops_unique(self.ops, uid, self.n_users)   # distinct users of this rank's batch + inverse index
            ru = self._route(uniq)
            # every rank needs every rank's number of distinct users (sizes of the variable "all_gather" / "reduce_scatter")
            counts.append(torch.full((self.world,), uniq.numel(), dtype=torch.int64, device=uid.device))
        else:
            ru = self._route(uid)
        ri = self._route(iid.reshape(-1)) if plan == "rows" else self._route(iid.reshape(-1), tuple_base=self.rank * B, div=C)
        return {"plan": plan, "u": ru, "i": ri, "inv": inv,
                "counts": _Counts([ru[1], ri[1]] + counts, self.group)}

    def step(self, uid, iid, next_batch=None):
        """uid [B] int64, iid [B, C] int64: this rank's tuples (same B on every rank).
        Returns the GLOBAL mean loss as a python-free device tensor [1].
        next_batch = (uid, iid) of the following step (optional): routed and its split sizes exchanged ahead of this
        step's kernels, so the host finds them ready and never waits for the device (see _Counts)."""
        W, ops = self.world, self.ops
        B, C = iid.shape
        n_tuples = W * B
        self.step_count += 1
        hyper = ops.make_hyper(opt=self.opt, lr=self.lr, l2=self.l2, step=self.step_count)
        if W == 1 and not self.force_exchange:
            return self._step_single(uid, iid, hyper)
        if self.timing is not None:
            self.timing.clear()
        mark = self._mark
        mark("start")
        self._mark_step_begin(uid)
        prep = self._routes_of(uid, iid, next_batch)
        if prep["plan"] == "rows":
            return self._step_rows(uid, iid, hyper, prep)

        # 0. user ids and candidate occurrences grouped by owner; ONE exchange of all split sizes (a step ahead with next_batch)
        order_u, _, req_local = prep["u"]
        order_i, _, packed = prep["i"]
        inv = prep["inv"]
        sends, recvs = prep["counts"].get()
        (cnt_u, cnt_i), (rcnt_u, rcnt_i) = sends[:2], recvs[:2]
        dev = uid.device
        mark("0 group by owner")

        # 1. fetch the batch's (distinct) user rows from their owners
        req_u, _ = _exchange(req_local, cnt_u, self.group, recv_counts=rcnt_u)
        rows_back = _exchange_back(ops.gather_rows(self.U, req_u), rcnt_u, cnt_u, self.group)
        n_mine = int(order_u.numel())                     # B, or this rank's number of distinct users
        Ub = torch.empty((n_mine, self.d), dtype=torch.float32, device=dev)
        Ub[order_u] = rows_back
        mark("1 fetch user rows")

        # 3. route candidate occurrences to the item's owner: (global tuple index, local row).  Every
        #    source sends in batch order, so the owner receives them grouped by tuple (t_idx ascending)
        recv, _ = _exchange(packed, cnt_i, self.group, recv_counts=rcnt_i)
        # 2. every owner needs every tuple's user row: this gather (the largest transfer of the step, with the
        #    reduction of phase 7) is issued now and travels while the owner unpacks and sorts what it received.
        #    With dedup_users only the distinct rows of every rank travel (a variable all_to_all, every rank sending
        #    its rows to all), plus 8 B per tuple for the tuple -> distinct-row index; the per-tuple block the
        #    owner-side kernels read is rebuilt locally (rc_gather_rows).
        if inv is None:
            Uall_pending = _all_gather_rows_async(Ub, W, self.group)
        else:
            d_all = recvs[2]                              # distinct users of every rank
            Uall_pending = _exchange_async(Ub.repeat(W, 1), [n_mine] * W, d_all, self.group)
            inv_all = _all_gather_rows(inv.contiguous(), W, self.group)        # [W * B] positions in the source's distinct list
            offs = torch.tensor([0] + list(np.cumsum(d_all))[:-1], dtype=torch.int64, device=dev)
            uidx_all = (inv_all + offs.repeat_interleave(B)).contiguous()     # row of the gathered distinct block per tuple
            self.wire = {"user_rows_gather_in": 4 * self.d * (sum(d_all) - n_mine), "user_grads_reduce_out": 4 * self.d * n_mine * (W - 1),
                         "tuple_index_in": 8 * B * (W - 1), "without_dedup_each_way": 4 * self.d * B * (W - 1)}
        t_idx, rows, t32 = self._unpack(recv)
        prep = ops.prepare_owner(rows, self.I.shape[0]) if hasattr(ops, "prepare_owner") else None
        mark("3 route occurrences + owner sort")
        Uall = Uall_pending.wait()
        if inv is not None:
            Uall = ops.gather_rows(Uall, uidx_all)        # [W * B, d], local
        mark("2 all_gather user rows (exposed part)")

        # 4. owner scores its rows; scores go home
        scores = ops.dot_rows(Uall, t_idx, self.I, rows)
        s_home = _exchange_back(scores, rcnt_i, cnt_i, self.group)
        pred = torch.empty(B * C, dtype=torch.float32, device=uid.device)
        pred[order_i] = s_home
        pred = pred.view(B, C)
        mark("4 score + scores home")

        # 5. loss and dL/dscore at home (mean over the GLOBAL batch)
        loss_vec, g = ops.bpr_loss(pred, 1.0 / n_tuples)
        loss = _all_reduce_sum(loss_vec.sum().reshape(1) / n_tuples, self.group)
        g_own, _ = _exchange(g.reshape(-1)[order_i], cnt_i, self.group, recv_counts=rcnt_i)  # routing of step 3
        mark("5 loss + g to owners")

        # 6. owner: partial user grads (from pre-step item rows) and the item-row update;
        # 7. the partial user grads are summed at the tuples' home (reduce_scatter, in flight while the owner
        #    updates its multi-occurrence item rows) and routed to the user rows' owners
        def reduce_users(pug):
            """partial user gradients [W * B, d] (per tuple, this owner's item rows) -> in flight towards the tuples' home"""
            if inv is None:
                return _reduce_scatter_rows_async(pug, W, self.group)
            # one row per DISTINCT user of every source rank (fixed-order segmented sum), then a variable all_to_all:
            # block r goes to rank r; home adds the W blocks it receives in rank order
            pug_d = ops.sum_rows_by_index(pug, uidx_all, int(sum(d_all)))
            return _exchange_async(pug_d, d_all, [n_mine] * W, self.group)

        if hasattr(ops, "owner_backward_split"):
            pug, finish = ops.owner_backward_split(self.I, self.sI, rows, g_own, t_idx, t32, Uall, n_tuples, hyper, prep)
            ugrad_pending = reduce_users(pug)
            finish()
        else:
            if hasattr(ops, "owner_backward"):
                pug = ops.owner_backward(self.I, self.sI, rows, g_own, t_idx, t32, Uall, n_tuples, hyper, prep)
            else:
                pug = ops.partial_user_grads(self.I, rows, g_own, t_idx, n_tuples)
                ops.update_rows(self.I, self.sI, rows, Uall, hyper, coef=g_own, src_index=t_idx)
            ugrad_pending = reduce_users(pug)
        mark("6 owner backward + item-row update")
        ugrad = ugrad_pending.wait()
        if inv is not None:
            ugrad = ugrad.view(W, n_mine, self.d).sum(dim=0)   # blocks in rank order: a fixed summation order
        ug_own, _ = _exchange(ugrad[order_u], cnt_u, self.group, recv_counts=rcnt_u)
        ops.update_rows(self.U, self.sU, req_u, ug_own, hyper)
        mark("7 user grads reduce + update")
        self._look_ahead()
        return loss

    def _plan(self, C):
        """which side travels: bytes over the links per tuple, (W-1)/W of the traffic of either plan"""
        if self.mode != "auto":
            return self.mode
        W, row = self.world, 4 * self.d
        rows_plan = 2 * (1 + C) * row * (W - 1) / W                      # fetch + gradient push of 1 + C rows
        owner_plan = 2 * (W - 1) * row + (2 * row + 16 * C) * (W - 1) / W  # all_gather + reduce_scatter + the small routes
        return "rows" if rows_plan < owner_plan else "owner"

    def _step_rows(self, uid, iid, hyper, prep):
        """the rows travel to the tuples (few candidates per tuple); see the module docstring"""
        W, ops, group = self.world, self.ops, self.group
        B, C = iid.shape
        n_tuples = W * B
        dev = uid.device
        mark = self._mark
        # 0. both lookups grouped by owner; ONE exchange of all split sizes
        order_u, _, loc_u = prep["u"]
        order_i, _, loc_i = prep["i"]
        (cnt_u, cnt_i), (rcnt_u, rcnt_i) = prep["counts"].get()
        mark("0 group by owner")

        # 1. owners serve the rows; they come back in SEND order (grouped by owner), the kernels below
        #    address them through the inverse permutation instead of un-permuting 256 B rows
        req_u, _ = _exchange(loc_u, cnt_u, group, recv_counts=rcnt_u)
        req_i, _ = _exchange(loc_i, cnt_i, group, recv_counts=rcnt_i)
        back_u = _exchange_back(ops.gather_rows(self.U, req_u), rcnt_u, cnt_u, group)  # [B, d]
        back_i = _exchange_back(ops.gather_rows(self.I, req_i), rcnt_i, cnt_i, group)  # [B*C, d]
        pos_u = torch.empty(B, dtype=torch.int64, device=dev)
        pos_u[order_u] = torch.arange(B, device=dev)
        pos_i = torch.empty(B * C, dtype=torch.int64, device=dev)
        pos_i[order_i] = torch.arange(B * C, device=dev)
        mark("1 fetch rows")

        # 2. home: scores, loss (mean over the GLOBAL batch), dL/dscore and the user-row gradients
        if hasattr(ops, "rows_fwd_bwd"):
            loss_vec, g, ugrad = ops.rows_fwd_bwd(back_u, back_i, pos_u, pos_i.view(B, C), 1.0 / n_tuples)
        else:
            t_idx = torch.arange(B, device=dev).repeat_interleave(C)
            pred = ops.dot_rows(back_u, pos_u[t_idx], back_i, pos_i).view(B, C)
            loss_vec, g = ops.bpr_loss(pred, 1.0 / n_tuples)
            ugrad = ops.partial_user_grads(back_i, pos_i, g.reshape(-1), t_idx, B)
        loss = _all_reduce_sum(loss_vec.sum().reshape(1) / n_tuples, group)
        mark("2 score + loss + backward")

        # 3. row gradients, produced directly in send order, go to the owners
        order_i64 = order_i.to(torch.int64)
        src_pos, coef = pos_u[order_i64 // C], g.reshape(-1)[order_i64]
        if hasattr(ops, "scaled_rows"):
            gi_send = ops.scaled_rows(back_u, src_pos, coef)
        else:
            gi_send = coef[:, None] * back_u[src_pos]
        gu_send = ops.gather_rows(ugrad, order_u.to(torch.int64))
        own_u, _ = _exchange(gu_send, cnt_u, group, recv_counts=rcnt_u)
        own_i, _ = _exchange(gi_send, cnt_i, group, recv_counts=rcnt_i)
        mark("3 push row gradients")

        # 4. owners: atomic-free segmented update of the rows they serve
        ops.update_rows(self.I, self.sI, req_i, own_i, hyper)
        ops.update_rows(self.U, self.sU, req_u, own_u, hyper)
        mark("4 owner updates")
        self._look_ahead()
        return loss

    def _route(self, ids, tuple_base=None, div=1):
        """(order, counts tensor [W], payload grouped by owner): HIP counting sort, or torch for test ops"""
        return self.ops.route(ids, self.world, tuple_base, div)

    def _unpack(self, recv):
        if hasattr(self.ops, "unpack"):
            return self.ops.unpack(recv)
        t_idx = (recv >> 32).contiguous()
        return t_idx, (recv & 0xFFFFFFFF).contiguous(), t_idx.to(torch.int32)

    def _step_single(self, uid, iid, hyper):
        """W = 1: the same arithmetic without any exchange (reference point for the tests)"""
        ops = self.ops
        B, C = iid.shape
        flat = iid.reshape(-1)
        t_idx = torch.arange(B, device=uid.device).repeat_interleave(C)
        Ub = ops.gather_rows(self.U, uid)
        pred = ops.dot_rows(Ub, t_idx, self.I, flat).view(B, C)
        loss_vec, g = ops.bpr_loss(pred, 1.0 / B)
        pug = ops.partial_user_grads(self.I, flat, g.reshape(-1), t_idx, B)
        ops.update_rows(self.I, self.sI, flat, Ub, hyper, coef=g.reshape(-1), src_index=t_idx)
        ops.update_rows(self.U, self.sU, uid, pug, hyper)
        return loss_vec.sum().reshape(1) / B


# ---- "move the rows": generic row-sharded tables, used where a tuple touches few, wide rows (NeuMF) -------------

class _Route:
    """where the ids of one lookup live: send order, split sizes both ways, the rows the owner has to read.

    dedup (default): only the DISTINCT ids of the lookup travel -- one row back per distinct id,
    expanded to lookup order at home (rc_gather_rows by the inverse index), and one gradient row out per distinct id:
    the per-position gradient rows are summed per id at home in a fixed order (the atomic-free segmented sum that
    stands in for embedding_dense_backward) before they are pushed.  Under Zipf users / positives that is most of
    the user-side traffic (27 K distinct of 65 K at the bench shape) and every repeated positive."""

    def __init__(self, ids, world, ops, group, grouped=None, splits=None, dedup=True, prepared=None, n_rows=None):
        if prepared is None:
            prepared = self.prepare(ids, world, ops, dedup, n_rows) if grouped is None else (grouped, None, ids.numel())
        (order, counts, local), self.inverse, self.n_lookup = prepared[:3]
        self.back_index = prepared[3] if len(prepared) > 3 else None    # lookup position -> row of the block that comes back
        if splits is None:
            (self.send,), (self.recv,) = _exchange_counts([counts], group)
        else:  # split sizes exchanged by the caller together with those of other lookups (one host sync for all)
            self.send, self.recv = splits
        self.order, self.group = order, group
        self.n_sent = int(order.numel())
        rank = dist.get_rank(group)
        self.remote_out = sum(self.send) - self.send[rank]   # ids this rank asks OTHER ranks for
        self.remote_in = sum(self.recv) - self.recv[rank]    # rows this rank serves to other ranks
        self.req, _ = _exchange(local, self.send, group, recv_counts=self.recv)  # local rows this rank must serve

    @staticmethod
    def prepare_many(lists, world, ops, dedup=True):
        """prepare() of several (ids, n_rows) lookups; with ops.unique_many the de-duplications share ONE host synchronisation"""
        if dedup and hasattr(ops, "unique_many") and len(lists) > 1:
            uniq = ops.unique_many([(ids.reshape(-1), n_rows) for ids, n_rows in lists])
            return [_Route.prepare(ids, world, ops, True, n_rows, unique=u) for (ids, n_rows), u in zip(lists, uniq)]
        return [_Route.prepare(ids, world, ops, dedup, n_rows) for ids, n_rows in lists]

    @staticmethod
    def prepare(ids, world, ops, dedup=True, n_rows=None, unique=None):
        """-> ((send order, per-owner counts, local rows in send order), inverse index | None, lookup length);
        n_rows: size of the (global) id space, needed for the de-duplication's bucket plan; unique: (distinct ids, inverse) where the
        caller has them already (prepare_many)"""
        n = ids.numel()
        inverse = None
        if dedup:
            ids, inverse = unique if unique is not None else ops_unique(ops, ids, n_rows)
        grouped = _Route.group_by_owner(ids, world, ops)
        # The rows come back in SEND order (row k answers the k-th id sent, id number order[k]) and are wanted in lookup order:
        # lookup position j reads row inv_order[inverse[j]].  The composed index is formed here, with the route (off the critical
        # path under look-ahead), so that a fetched block is expanded by ONE gather and the gradient rows are summed straight into
        # send order -- instead of a row-wide index_put into send-independent order plus a gather each way.
        order = grouped[0].long()
        inv_order = torch.empty_like(order)
        inv_order[order] = torch.arange(order.numel(), device=order.device, dtype=torch.int64)
        back_index = inv_order if inverse is None else inv_order[inverse.long()]
        return grouped, inverse, n, back_index

    @staticmethod
    def group_by_owner(ids, world, ops):
        """-> (send order, per-owner counts (device tensor), local rows in send order)"""
        return ops.route(ids, world, None, 1)

    def _expand(self, back, ops):
        if self.back_index is not None:
            return ops.gather_rows(back, self.back_index)
        out = torch.empty_like(back)
        out[self.order] = back
        return out if self.inverse is None else ops.gather_rows(out, self.inverse)

    def _reduce(self, grads, ops):
        if self.back_index is not None and self.inverse is not None:
            return ops.sum_rows_by_index(grads, self.back_index, self.n_sent)     # per-id sums, already in send order
        if self.inverse is not None:
            grads = ops.sum_rows_by_index(grads, self.inverse, self.n_sent)
        return grads[self.order]

    def _serve(self, tables, ops):
        """the rows of `tables` (same row space) this rank owes the others, side by side"""
        if len(tables) == 2 and hasattr(ops, "gather_rows_pair") and tables[0].shape[1] == tables[1].shape[1] and tables[0].shape[1] % 4 == 0:
            return ops.gather_rows_pair(tables[0], tables[1], self.req)
        return torch.cat([ops.gather_rows(T, self.req) for T in tables], dim=1)

    def fetch_async(self, tables, ops):
        """fetch() whose rows-back transfer may stay in flight: -> _Pending; finish with rows_in_lookup_order()"""
        return _exchange_async(self._serve(tables, ops), self.recv, self.send, self.group)

    def rows_in_lookup_order(self, back, ops):
        return self._expand(back, ops)

    def fetch_block(self, served, ops):
        """fetch() of a block the caller built for self.req itself (rows of several tables side by side, or values computed
        from them at the owner: ShardedNeumf's (mf_i | W1i mlp_i))"""
        return self._expand(_exchange_back(served, self.recv, self.send, self.group), ops)

    def fetch_block_async(self, served):
        return _exchange_async(served, self.recv, self.send, self.group)

    def push_async(self, grads, ops, out=None):
        return _exchange_async(self._reduce(grads, ops), self.send, self.recv, self.group, out=out)

    def fetch(self, tables, ops):
        """rows of `tables` (this rank's shards, same row space) for the ids of the lookup, in lookup order"""
        return self._expand(_exchange_back(self._serve(tables, ops), self.recv, self.send, self.group), ops)

    def push(self, grads, ops):
        """per-lookup-position gradient rows -> the owners, aligned with self.req (one row per id sent)"""
        own, _ = _exchange(self._reduce(grads, ops), self.send, self.group, recv_counts=self.recv)
        return own

    def wire_bytes(self, row_bytes):
        """bytes this rank moves over the links for this lookup: ids out, rows in, gradient rows out (row_bytes each way)"""
        return {"ids_out": 8 * self.remote_out, "rows_in": row_bytes * self.remote_out, "grads_out": row_bytes * self.remote_out,
                "rows_served_out": row_bytes * self.remote_in, "lookups": self.n_lookup, "ids_sent": self.n_sent}


class ShardedNeumf(_LookAhead):
    """NeuMF (one hidden layer) with its four tables sharded by row over W ranks (BASELINE config 4: d = 128,
    K = 4, 100 M items).  A tuple touches 2 user rows and 2(1+K) item rows of 512 B each, so here the ROWS
    travel: ids are routed to their owners (rc_route_by_owner + all_to_all), owners gather
    (rc_gather_rows) and send the rows back, the MFMA head kernels (rc_neumf_fwd / rc_neumf_bwd) run on the
    compact per-batch row blocks with positional ids, per-occurrence row gradients return along the same
    route and the owners apply the atomic-free segmented update.  The MLP is replicated: its gradients are
    all-reduced and every rank takes the same dense step.  One step equals single-table training on the
    concatenated global batch (tests/test_sharded_gloo.py)."""

    TABLES = ("mf_u", "mlp_u", "mf_i", "mlp_i")

    def __init__(self, n_users, n_items, emb_size, hidden, opt="SGD", lr=1e-3, l2=0.0, device=None, ops=None,
                 group=None, init_std=0.01, seed=0, micro_batches=1, dedup=True, item_half="auto", force_exchange=False):
        self.micro_batches = max(1, int(micro_batches))
        self.force_exchange = bool(force_exchange)   # W = 1 through the general exchange path (loop-back profiling of one rank alone)
        self.dedup = bool(dedup)
        self._item_half_arg = item_half
        self.wire = None  # {"ids_out", "rows_in", "grads_out", ...} bytes of the last step (this rank), W > 1
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.n_users, self.n_items, self.d, self.l1 = n_users, n_items, emb_size, hidden
        self.ops = ops if ops is not None else HipOps()
        self.opt, self.lr, self.l2, self.device = opt, lr, l2, device
        W = self.world
        g = torch.Generator(device=device if device is not None else "cpu")
        g.manual_seed(seed * 1000 + self.rank)
        mk = lambda *shape: torch.empty(shape, device=device).normal_(0, init_std, generator=g)
        ru, ri = (n_users + W - 1) // W, (n_items + W - 1) // W
        self.P = {"mf_u": mk(ru, emb_size), "mlp_u": mk(ru, emb_size), "mf_i": mk(ri, emb_size), "mlp_i": mk(ri, emb_size)}
        g.manual_seed(seed * 1000 + 999)  # replicated parameters: identical on every rank
        self.P.update(W1=mk(hidden, 2 * emb_size), b1=mk(hidden), w_out=mk(emb_size + hidden))
        self.state = {k: self.ops.new_state(v, opt) for k, v in self.P.items()}
        self.step_count = 0
        # "owner": the item half of the hidden layer is computed where the item rows live -- NeuMF.py:61 tiles the user ids over
        # the candidates, so W1 [mlp_u ; mlp_i] = W1u mlp_u + W1i mlp_i: an owner returns (mf_i | zi = W1i mlp_i), d + hidden floats
        # instead of 2 d, receives (d mf_i | dz) back, and forms d mlp_i = W1i^T dz and its share of dW1i = dz^T mlp_i itself.
        # Fewer bytes each way whenever hidden < emb_size (config 4: 192 instead of 256 floats per distinct item id, -25 %).
        if item_half not in ("auto", "owner", "rows"):
            raise ValueError("item_half must be auto | owner | rows")
        if item_half == "owner" and not ((self.world > 1 or self.force_exchange) and all(hasattr(self.ops, m) for m in ("item_half_fwd", "item_half_bwd", "neumf_zhead"))):
            raise ValueError("item_half='owner' needs W > 1 and ops with item_half_fwd / item_half_bwd / neumf_zhead")
        # "auto" = "rows" for now: on one GPU the owner's three small-K GEMMs per chunk (0.087 ms) and the lighter home head
        # (0.11 against the MFMA head's 0.15 ms) add 0.05 ms per chunk and save 9.6 MB each way -- even at ~400 GB/s per GPU, and
        # no link rate has been measured yet (profiles/r09_sharded_neumf_item_half.txt); RC_SHARDED_ITEM_HALF=owner selects it
        self.item_half = "owner" if item_half == "owner" else "rows"

    def _item_row_floats(self):
        """floats per distinct item id on the wire, each way"""
        return self.d + self.l1 if self.item_half == "owner" else 2 * self.d

    def load_global(self, P):
        """full tables / MLP -> this rank's shards (global row = local * W + rank)"""
        W, r = self.world, self.rank
        for k in self.TABLES:
            shard = P[k][r::W]
            self.P[k].zero_()
            self.P[k][: shard.shape[0]].copy_(shard)
        for k in ("W1", "b1", "w_out"):
            self.P[k].copy_(P[k].reshape(self.P[k].shape))

    def gather_global(self):
        out = {}
        for k in self.TABLES:
            n = self.n_users if k.endswith("_u") else self.n_items
            shard = self.P[k]
            if self.world == 1:
                out[k] = shard[:n].clone()
                continue
            allg = _all_gather_rows(shard, self.world, self.group).view(self.world, shard.shape[0], self.d)
            out[k] = allg.permute(1, 0, 2).reshape(-1, self.d)[:n].clone()
        for k in ("W1", "b1", "w_out"):
            out[k] = self.P[k].clone()
        return out

    # ---- routing of a batch, separable from its step (look-ahead) -------------------------------------------------
    def _prepare(self, uid, iid):
        """group the batch's ids by owner (per micro-batch chunk) and START the exchange of the split sizes"""
        W, ops = self.world, self.ops
        M = self.micro_batches if ((W > 1 or self.force_exchange) and self.micro_batches > 1 and iid.shape[0] >= self.micro_batches) else 1
        uc, ic = (torch.chunk(uid, M), torch.chunk(iid, M)) if M > 1 else ((uid,), (iid,))
        flat = _Route.prepare_many([x for u, i in zip(uc, ic) for x in ((u, self.n_users), (i.reshape(-1), self.n_items))], W, ops, self.dedup)
        grouped = [(flat[2 * k], flat[2 * k + 1]) for k in range(len(uc))]
        counts = _Counts([g[0][1] for pair in grouped for g in pair], self.group)
        return {"uc": uc, "ic": ic, "grouped": grouped, "counts": counts}

    def _finish_routes(self, prep):
        """complete the routes of a prepared batch on the current (side) stream: split sizes from the host, ids to their owners"""
        W, ops, group = self.world, self.ops, self.group
        sends, recvs = prep["counts"].get()
        routes = []
        for k, (gu, gi) in enumerate(prep["grouped"]):
            ru = _Route(prep["uc"][k], W, ops, group, prepared=gu, splits=(sends[2 * k], recvs[2 * k]))
            rv = _Route(prep["ic"][k].reshape(-1), W, ops, group, prepared=gi, splits=(sends[2 * k + 1], recvs[2 * k + 1]))
            routes.append((ru, rv))
        prep["routes"] = routes

    def step(self, uid, iid, next_batch=None):
        """uid [B], iid [B, C]: this rank's tuples (same B everywhere) -> global mean loss, device tensor [1].
        next_batch = (uid, iid) of the FOLLOWING step (optional): its ids are grouped by owner and the exchange of its
        split sizes is started before this step's kernels are enqueued, so the next step finds the sizes on the host and
        the host never waits for the device (otherwise: one blocking device -> host copy per step)."""
        W, ops, d = self.world, self.ops, self.d
        B, C = iid.shape
        n_tuples = W * B
        self.step_count += 1
        hyper = ops.make_hyper(opt=self.opt, lr=self.lr, l2=self.l2, step=self.step_count)
        hyper0 = ops.make_hyper(opt=self.opt, lr=self.lr, l2=0.0, step=self.step_count)  # 'bias': no weight decay
        dev = uid.device
        if self.timing is not None:
            self.timing.clear()
        mark = self._mark
        mark("start")
        self._mark_step_begin(uid)
        multi = W > 1 or self.force_exchange
        if multi and self.micro_batches > 1 and B >= self.micro_batches:
            return self._step_pipelined(uid, iid, hyper, hyper0, self._routes_of(uid, iid, next_batch))
        if not multi:
            ru = rv = None
            urows = torch.cat([ops.gather_rows(self.P[k], uid) for k in ("mf_u", "mlp_u")], dim=1)
            irows = torch.cat([ops.gather_rows(self.P[k], iid.reshape(-1)) for k in ("mf_i", "mlp_i")], dim=1)
        else:
            prep = self._routes_of(uid, iid, next_batch)
            (pu_, pv_), = prep["grouped"]
            sends, recvs = prep["counts"].get()   # split sizes of both lookups (on the host already with look-ahead)
            if prep.get("routes") is not None:
                (ru, rv), = prep["routes"]
            else:
                ru = _Route(uid, W, ops, self.group, splits=(sends[0], recvs[0]), prepared=pu_)
                rv = _Route(iid.reshape(-1), W, ops, self.group, splits=(sends[1], recvs[1]), prepared=pv_)
            self._account([ru, rv])
            mark("route")
            urows = ru.fetch([self.P["mf_u"], self.P["mlp_u"]], ops)      # [B, 2d]
            if self.item_half == "owner":
                W1i = self.P["W1"][:, d:].contiguous()
                mlp_req = ops.gather_rows(self.P["mlp_i"], rv.req)          # the rows this rank serves (kept for the backward)
                served = torch.cat([ops.gather_rows(self.P["mf_i"], rv.req), ops.item_half_fwd(mlp_req, W1i)], dim=1)
                irows = rv.fetch_block(served, ops)                         # [B*C, d + l1] = (mf_i | W1i mlp_i)
            else:
                irows = rv.fetch([self.P["mf_i"], self.P["mlp_i"]], ops)      # [B*C, 2d]
        mark("fetch_rows")
        loss_vec, gu, gi, dense = self._head(urows, irows, B, C, n_tuples, mark)
        loss = loss_vec.sum().reshape(1) / n_tuples
        if multi:
            loss = _all_reduce_sum(loss, self.group)
            own_u, own_i, req_u, req_i = ru.push(gu, ops), rv.push(gi, ops), ru.req, rv.req
            if self.item_half == "owner":     # (d mf_i | dz) arrived: d mlp_i and this rank's share of dW1i are formed here
                dmlp, dW1i = ops.item_half_bwd(mlp_req, W1i, own_i[:, d:].contiguous())
                own_i = torch.cat([own_i[:, :d], dmlp], dim=1)
                dense["W1"] = torch.cat([dense.pop("W1u"), dW1i], dim=1)
            flat = torch.cat([dense[k].reshape(-1) for k in ("W1", "b1", "w_out")])
            flat = _all_reduce_sum(flat, self.group)  # replicated MLP: summed gradients, identical step everywhere
            o = 0
            for k in ("W1", "b1", "w_out"):
                n = dense[k].numel()
                dense[k] = flat[o:o + n].view(dense[k].shape)
                o += n
        else:
            own_u, own_i, req_u, req_i = gu, gi, uid, iid.reshape(-1)
        mark("push_grads")
        shared = hasattr(ops, "prepare_rows")  # one sort + head list per side, used by its mf and mlp table
        prep_u = ops.prepare_rows(req_u, self.P["mf_u"].shape[0], tag="rows.u") if shared else None
        prep_i = ops.prepare_rows(req_i, self.P["mf_i"].shape[0], tag="rows.i") if shared else None
        for ta, tb, own, req, prep in (("mf_u", "mlp_u", own_u, req_u, prep_u), ("mf_i", "mlp_i", own_i, req_i, prep_i)):
            if (shared and getattr(ops, "pair_block_updates", False) and own.shape[1] == 2 * d and own.is_contiguous() and not isinstance(prep, _SortPlan)
                    and ops.update_rows_pair(self.P[ta], self.P[tb], self.state[ta], self.state[tb], req, own, None, hyper, prep)):
                continue      # (the received (d mf | d mlp) block read where it lies: no contiguous halves)
            ga, gb = own[:, :d].contiguous(), own[:, d:].contiguous()
            if shared and hasattr(ops, "update_rows_pair") and ops.update_rows_pair(
                    self.P[ta], self.P[tb], self.state[ta], self.state[tb], req, ga, gb, hyper, prep):
                continue
            kw = {"prep": prep} if shared else {}
            ops.update_rows(self.P[ta], self.state[ta], req, ga, hyper, **kw)
            ops.update_rows(self.P[tb], self.state[tb], req, gb, hyper, **kw)
        mark("table_update")
        for k in ("W1", "b1", "w_out"):
            ops.dense_update(self.P[k], dense[k].contiguous(), hyper0 if k == "b1" else hyper, self.state[k])
        mark("dense_update")
        self._look_ahead()
        return loss

    def _head(self, urows, irows, B, C, n_tuples, mark):
        """the head on the fetched row blocks urows [B, 2d], irows [B C, 2d] (layout [mf | mlp]) -> per-tuple loss, the user
        and item gradient blocks in the same layout, the MLP's dense gradients.  ONE kernel where the fused step has an instance
        (HipOps.neumf_head: the rows are read in place through their stride, the user half of the layer and the user gradients are
        per-tuple work, the loss stays in registers); otherwise -- other widths, the oracle-backed ops of the CPU tests --
        forward kernel, loss kernel, backward kernel on contiguous copies with positional ids."""
        ops, d, dev = self.ops, self.d, urows.device
        if self.item_half == "owner":     # irows = (mf_i | zi): dense carries "W1u" (the user half; the owners add theirs)
            out = ops.neumf_zhead(urows, irows, self.P, B, C, 1.0 / n_tuples)
            if out is None:
                raise RuntimeError("ShardedNeumf(item_half='owner'): no head kernel for C=%d d=%d hidden=%d" % (C, d, self.l1))
            mark("head_fwd_loss")
            mark("head_bwd")
            return out
        fused = ops.neumf_head(urows, irows, self.P, B, C, 1.0 / n_tuples) if hasattr(ops, "neumf_head") else None
        if fused is not None:
            mark("head_fwd_loss")
            mark("head_bwd")
            return fused
        loc = {"mf_u": urows[:, :d].contiguous(), "mlp_u": urows[:, d:].contiguous(),
               "mf_i": irows[:, :d].contiguous(), "mlp_i": irows[:, d:].contiguous(),
               "W1": self.P["W1"], "b1": self.P["b1"], "w_out": self.P["w_out"]}
        pos_u = torch.arange(B, device=dev)
        pos_i = torch.arange(B * C, device=dev).view(B, C)
        pred = ops.neumf_fwd(loc, pos_u, pos_i)
        loss_vec, g = ops.bpr_loss(pred, 1.0 / n_tuples)
        mark("head_fwd_loss")
        rows, dense = ops.neumf_bwd(loc, pos_u, pos_i, g)
        gu = torch.cat([rows["g_mf_u"].view(B, C, d).sum(dim=1), rows["g_mlp_u"].view(B, C, d).sum(dim=1)], dim=1)
        gi = torch.cat([rows["g_mf_i"], rows["g_mlp_i"]], dim=1)
        mark("head_bwd")
        return loss_vec, gu, gi, dense

    def _account(self, routes):
        """bytes over the links of this step (this rank): a user route moves (mf_u | mlp_u) rows, 2 d floats; an item route
        (mf_i | mlp_i), or (mf_i | W1i mlp_i) = d + hidden floats with the owner-computed item half"""
        tot = {}
        for k_route, r in enumerate(routes):      # routes alternate user, item (one pair per chunk)
            floats = 2 * self.d if k_route % 2 == 0 else self._item_row_floats()
            for k, v in r.wire_bytes(floats * 4).items():
                tot[k] = tot.get(k, 0) + v
        self.wire = tot

    def _step_pipelined(self, uid, iid, hyper, hyper0, prep):
        """The same step with the local batch cut into `micro_batches` chunks: the row fetch of chunk k+1 and the
        gradient push of chunk k-1 are in flight (RCCL's stream) while the head kernels of chunk k run.  Every chunk
        is scored against the pre-step parameters and the owners apply ONE update per table over the gradients of
        all chunks, so the result equals the unpipelined step (tests/test_sharded_gloo.py).  All split sizes of
        all chunks travel in one exchange (prep["counts"], started by _prepare -- a step earlier with look-ahead)."""
        W, ops, d, group = self.world, self.ops, self.d, self.group
        B, C = iid.shape
        n_tuples = W * B
        dev = uid.device
        uc, ic, grouped = prep["uc"], prep["ic"], prep["grouped"]
        M = len(uc)
        sends, recvs = prep["counts"].get()
        routes = []
        mark = self._mark
        mark("route")

        owner = self.item_half == "owner"
        W1i = self.P["W1"][:, d:].contiguous() if owner else None
        mlp_reqs = []

        ready_routes = prep.get("routes")      # (completed ahead of this step, look-ahead: _finish_routes)

        def start(k):  # routes of chunk k, its rows requested (transfers may stay in flight)
            if ready_routes is not None:
                ru, rv = ready_routes[k]
            else:
                ru = _Route(uc[k], W, ops, group, prepared=grouped[k][0], splits=(sends[2 * k], recvs[2 * k]))
                rv = _Route(ic[k].reshape(-1), W, ops, group, prepared=grouped[k][1], splits=(sends[2 * k + 1], recvs[2 * k + 1]))
            mark("route_ids")     # (the id exchanges of the two routes)
            if owner:
                mlp_req = ops.gather_rows(self.P["mlp_i"], rv.req)
                mlp_reqs.append(mlp_req)
                served_i = torch.cat([ops.gather_rows(self.P["mf_i"], rv.req), ops.item_half_fwd(mlp_req, W1i)], dim=1)
            else:
                served_i = rv._serve([self.P["mf_i"], self.P["mlp_i"]], ops)     # (mf | mlp) blocks from one kernel where the ops have it
            served_u = ru._serve([self.P["mf_u"], self.P["mlp_u"]], ops)
            mark("serve_rows")    # (owner side, local: row gathers, and the item half of the hidden layer where the owners compute it)
            routes.append((ru, rv, ru.fetch_block_async(served_u), rv.fetch_block_async(served_i)))

        start(0)
        loss = torch.zeros(1, dtype=torch.float32, device=dev)
        dense_sum, pushes = None, []
        # the gradient rows the other ranks push to this one arrive chunk by chunk: they are received straight into their slices of
        # ONE block per side (the owner-side update reads that block), not into a buffer per chunk that a 171 MB cat assembles later
        tot_u, tot_i = [sum(recvs[2 * k]) for k in range(M)], [sum(recvs[2 * k + 1]) for k in range(M)]
        own_u = torch.empty((sum(tot_u), 2 * d), dtype=torch.float32, device=dev)
        own_i = torch.empty((sum(tot_i), self._item_row_floats()), dtype=torch.float32, device=dev)
        off_u, off_i = 0, 0
        for k in range(M):
            if k + 1 < M:
                start(k + 1)
            ru, rv, pu, pv = routes[k]
            urows, irows = ru.rows_in_lookup_order(pu.wait(), ops), rv.rows_in_lookup_order(pv.wait(), ops)
            mark("fetch_rows")
            Bk = uc[k].shape[0]
            loss_vec, gu, gi, dense = self._head(urows, irows, Bk, C, n_tuples, mark)
            loss = loss + loss_vec.sum().reshape(1) / n_tuples
            pushes.append((ru.push_async(gu, ops, out=own_u[off_u:off_u + tot_u[k]]), rv.push_async(gi, ops, out=own_i[off_i:off_i + tot_i[k]])))
            off_u, off_i = off_u + tot_u[k], off_i + tot_i[k]
            flat = torch.cat([dense[n].reshape(-1) for n in (("W1u" if owner else "W1"), "b1", "w_out")])
            dense_sum = flat if dense_sum is None else dense_sum + flat
        loss = _all_reduce_sum(loss, group)
        self._account([r for pair in routes for r in pair[:2]])
        for pu_, pv_ in pushes:
            pu_.wait()
            pv_.wait()
        req_u = torch.cat([r[0].req for r in routes])
        req_i = torch.cat([r[1].req for r in routes])
        mark("push_grads")
        if owner:   # (d mf_i | dz) of every chunk arrived: ONE pair of GEMMs forms d mlp_i and this rank's share of dW1i
            dmlp, dW1i = ops.item_half_bwd(torch.cat(mlp_reqs), W1i, own_i[:, d:].contiguous())
            own_i = torch.cat([own_i[:, :d], dmlp], dim=1)
            n_u = self.l1 * d
            dW1 = torch.cat([dense_sum[:n_u].view(self.l1, d), dW1i], dim=1)
            dense_sum = torch.cat([dW1.reshape(-1), dense_sum[n_u:]])
            mark("owner_bwd")
        dense_sum = _all_reduce_sum(dense_sum, group)
        mark("dense_allreduce")
        shared = hasattr(ops, "prepare_rows")
        prep_u = ops.prepare_rows(req_u, self.P["mf_u"].shape[0], tag="rows.u") if shared else None
        prep_i = ops.prepare_rows(req_i, self.P["mf_i"].shape[0], tag="rows.i") if shared else None
        for ta, tb, own, req, prep in (("mf_u", "mlp_u", own_u, req_u, prep_u), ("mf_i", "mlp_i", own_i, req_i, prep_i)):
            if (shared and getattr(ops, "pair_block_updates", False) and own.shape[1] == 2 * d and own.is_contiguous() and not isinstance(prep, _SortPlan)
                    and ops.update_rows_pair(self.P[ta], self.P[tb], self.state[ta], self.state[tb], req, own, None, hyper, prep)):
                continue      # (the received (d mf | d mlp) block read where it lies: no contiguous halves)
            ga, gb = own[:, :d].contiguous(), own[:, d:].contiguous()
            if shared and hasattr(ops, "update_rows_pair") and ops.update_rows_pair(
                    self.P[ta], self.P[tb], self.state[ta], self.state[tb], req, ga, gb, hyper, prep):
                continue
            kw = {"prep": prep} if shared else {}
            ops.update_rows(self.P[ta], self.state[ta], req, ga, hyper, **kw)
            ops.update_rows(self.P[tb], self.state[tb], req, gb, hyper, **kw)
        mark("table_update")
        o = 0
        for n in ("W1", "b1", "w_out"):
            cnt = self.P[n].numel()
            ops.dense_update(self.P[n], dense_sum[o:o + cnt].view(self.P[n].shape).contiguous(), hyper0 if n == "b1" else hyper,
                             self.state[n])
            o += cnt
        mark("dense_update")
        self._look_ahead()
        return loss


class DataParallelDense(_Timed):
    """Data-parallel leg for models whose parameters are replicated (BASELINE configs[4]: DeepFM-CTR on MIND, tables of
    269 K users / 9.4 K items fit every GPU): each rank runs model(batch) -> loss -> backward on ITS batch, the dense
    gradients of all parameters are summed over the ranks in ONE flat all-reduce (RCCL ring over xGMI; gloo in the CPU
    tests) and divided by the world size -- the gradient of the mean loss over the global batch -- and every rank
    takes the same optimizer step (the reference's dense torch.optim semantics, helpers/BaseRunner.py:110-114,206).

        dp = DataParallelDense(model)            # after model.optimizer has been built
        loss = dp.step(batch)                    # in place of the loop body of BaseRunner.fit

    Byte model: the all-reduce moves 2 (W-1)/W x the parameter bytes per rank and step, 73 MB of
    dense gradients for config 5 whatever the batch size -- the user table's dense gradient dominates; exchanging only
    the touched rows (ids + gradient rows, all_gather) is the next step."""

    def __init__(self, model, group=None, loss_of=None):
        self.model, self.group = model, group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.loss_of = loss_of or (lambda m, b: m.loss(m(b)))
        self.bytes_per_step = 0

    def step(self, batch):
        m = self.model
        if self.timing is not None:
            self.timing.clear()
        self._mark("start")
        m.optimizer.zero_grad()
        loss = self.loss_of(m, batch)
        loss.backward()
        self._mark("forward_backward")
        if self.world > 1:
            ps = [p for p in m.parameters() if p.grad is not None]
            flat = torch.cat([p.grad.reshape(-1) for p in ps])
            flat = _all_reduce_sum(flat, self.group) / self.world
            self.bytes_per_step = 2 * (self.world - 1) * flat.numel() * 4 // self.world
            o = 0
            for p in ps:
                n = p.grad.numel()
                p.grad.copy_(flat[o:o + n].view_as(p.grad))
                o += n
            loss = _all_reduce_sum(loss.detach().reshape(1), self.group) / self.world
        self._mark("gradient_allreduce")
        m.optimizer.step()
        self._mark("optimizer")
        return loss.detach().reshape(1)
