"""Tensor-level access to the HIP engine: torch tensors are only typed device buffers
(`data_ptr()`), every FLOP happens in librechorus_hip.so on torch's current HIP stream.

Nothing here falls back to torch arithmetic: a missing library, a CPU tensor or an
unsupported shape raises.
"""
import ctypes as C
import os

import torch

from . import _lib
from ._lib import OptHyper, OPT_BY_NAME


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t, dtype, name, allow_none=False):
    if t is None:
        if allow_none:
            return C.c_void_p(0)
        raise ValueError(f"{name} is required")
    if not t.is_cuda:
        raise ValueError(f"{name} must live on the GPU (got {t.device}); the HIP engine has no CPU path")
    if t.dtype != dtype:
        raise TypeError(f"{name} must be {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name} must be contiguous")
    return C.c_void_p(t.data_ptr())


def make_hyper(opt="SGD", lr=1e-3, l2=0.0, beta1=0.9, beta2=0.999, eps=None, step=1):
    """rc_opt_hyper with torch.optim's defaults (Adam eps 1e-8, Adagrad eps 1e-10, Adadelta rho 0.9 / eps 1e-6;
    Adadelta's rho travels in beta1 and is built for dense steps only)."""
    if opt not in OPT_BY_NAME:
        raise ValueError(f"optimizer {opt!r} not supported by the HIP engine (SGD, Adam, Adagrad, Adadelta)")
    if eps is None:
        eps = {"Adagrad": 1e-10, "Adadelta": 1e-6}.get(opt, 1e-8)
    return OptHyper(OPT_BY_NAME[opt], 0, float(lr), float(l2), float(beta1), float(beta2),
                    float(eps), int(step))


_ws_cache = {}
_ws_retired = []


def workspace(nbytes, device, tag="default"):
    """Cached uint8 scratch buffer (grow-only) per (device, tag).  A buffer that is outgrown is RETIRED, not
    freed: a captured hipGraph (rechorus_amd/graph.py) holds raw pointers into the buffers that were current
    at capture time, and a later, larger request (another batch shape, an evaluation pass) must not pull
    that memory from under its replays.  Growth is geometric, so retired buffers total less than the
    live one."""
    key = (str(device), tag)
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        if buf is not None:
            _ws_retired.append(buf)
            nbytes = max(int(nbytes), 2 * buf.numel())
        buf = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)
        _ws_cache[key] = buf
    return buf


# ---- forward ---------------------------------------------------------------------------

def gather_rows(W, ids):
    """nn.Embedding forward: W[ids] (reference: models/general/BPRMF.py:39-40)."""
    ids_flat = ids.reshape(-1)
    d = W.shape[1]
    out = torch.empty((ids_flat.numel(), d), dtype=torch.float32, device=W.device)
    _lib.call("rc_gather_rows", _ptr(W, torch.float32, "W"), d, _ptr(ids_flat, torch.int64, "ids"),
              ids_flat.numel(), _ptr(out, torch.float32, "out"), _stream())
    return out.view(*ids.shape, d)


def gather_rows_pair(Wa, Wb, ids):
    """(Wa[ids] | Wb[ids]) as one [n, 2 d] block (rc_gather_rows_pair): two tables that share the ids"""
    ids_flat = ids.reshape(-1)
    d = Wa.shape[1]
    if Wb.shape[1] != d or d % 4:
        raise ValueError("gather_rows_pair: two tables of the same width, a multiple of 4")
    out = torch.empty((ids_flat.numel(), 2 * d), dtype=torch.float32, device=Wa.device)
    _lib.call("rc_gather_rows_pair", _ptr(Wa, torch.float32, "Wa"), _ptr(Wb, torch.float32, "Wb"), d, _ptr(ids_flat, torch.int64, "ids"),
              ids_flat.numel(), _ptr(out, torch.float32, "out"), _stream())
    return out


def gather_dot(U, I, uid, iid):
    """pred[b,c] = <U[uid[b]], I[iid[b,c]]> (reference: models/general/BPRMF.py:39-42)."""
    B, Cn = iid.shape
    d = U.shape[1]
    if I.shape[1] != d:
        raise ValueError("user and item tables need the same emb_size")
    pred = torch.empty((B, Cn), dtype=torch.float32, device=U.device)
    _lib.call("rc_gather_dot_fwd", _ptr(U, torch.float32, "U"), _ptr(I, torch.float32, "I"),
              _ptr(uid, torch.int64, "uid"), _ptr(iid, torch.int64, "iid"), B, Cn, d,
              _ptr(pred, torch.float32, "pred"), _stream())
    return pred


def weighted_row_sum(W, ids, coef):
    """out[b] = sum_c coef[b,c] * W[ids[b,c]]  (user-row gradient of the GMF dot)."""
    B, Cn = ids.shape
    d = W.shape[1]
    out = torch.empty((B, d), dtype=torch.float32, device=W.device)
    _lib.call("rc_weighted_row_sum", _ptr(W, torch.float32, "W"), _ptr(ids, torch.int64, "ids"),
              _ptr(coef, torch.float32, "coef"), B, Cn, d, _ptr(out, torch.float32, "out"), _stream())
    return out


# ---- loss ------------------------------------------------------------------------------

def reduce_sum(x, scale=1.0):
    out = torch.empty(1, dtype=torch.float32, device=x.device)
    _lib.call("rc_reduce_sum", _ptr(x, torch.float32, "x"), x.numel(), float(scale),
              _ptr(out, torch.float32, "out"), _stream())
    return out


def bpr_loss(pred, inv_b=None, need_grad=True):
    """GeneralModel.loss (models/BaseModel.py:182-185) -> (loss[1], loss_vec[B], gpred|None)."""
    B, Cn = pred.shape
    if inv_b is None:
        inv_b = 1.0 / B
    loss_vec = torch.empty(B, dtype=torch.float32, device=pred.device)
    gpred = torch.empty_like(pred) if need_grad else None
    _lib.call("rc_bpr_loss_fwd_bwd", _ptr(pred, torch.float32, "pred"), B, Cn, float(inv_b),
              _ptr(loss_vec, torch.float32, "loss_vec"),
              _ptr(gpred, torch.float32, "gpred", allow_none=True), _stream())
    return reduce_sum(loss_vec, inv_b), loss_vec, gpred


def bprmf_fwd_bwd(U, I, uid, iid, inv_b=None, want_pred=True):
    """Fused gather + dot + BPR loss + backward to rows.
    Returns (pred|None, loss_vec[B], gpred[B,C], ugrad[B,d])."""
    B, Cn = iid.shape
    d = U.shape[1]
    if inv_b is None:
        inv_b = 1.0 / B
    dev = U.device
    pred = torch.empty((B, Cn), dtype=torch.float32, device=dev) if want_pred else None
    loss_vec = torch.empty(B, dtype=torch.float32, device=dev)
    gpred = torch.empty((B, Cn), dtype=torch.float32, device=dev)
    ugrad = torch.empty((B, d), dtype=torch.float32, device=dev)
    _lib.call("rc_bprmf_fwd_bwd", _ptr(U, torch.float32, "U"), _ptr(I, torch.float32, "I"),
              _ptr(uid, torch.int64, "uid"), _ptr(iid, torch.int64, "iid"), B, Cn, d, float(inv_b),
              _ptr(pred, torch.float32, "pred", allow_none=True),
              _ptr(loss_vec, torch.float32, "loss_vec"), _ptr(gpred, torch.float32, "gpred"),
              _ptr(ugrad, torch.float32, "ugrad"), _stream())
    return pred, loss_vec, gpred, ugrad


# ---- sort + segmented update --------------------------------------------------------------

def sort_ids(ids, n_rows):
    """Stable sort of a flat id list -> (keys uint32-as-int32 tensor, perm)."""
    ids_flat = ids.reshape(-1)
    n = ids_flat.numel()
    dev = ids.device
    keys = torch.empty(n, dtype=torch.int32, device=dev)  # bit pattern is uint32
    perm = torch.empty(n, dtype=torch.int32, device=dev)
    nbytes = _lib.load().rc_sort_workspace_bytes(n)
    ws = workspace(nbytes, dev, "sort")
    _lib.call("rc_sort_ids", _ptr(ids_flat, torch.int64, "ids"), n, int(n_rows),
              _ptr(keys, torch.int32, "keys"), _ptr(perm, torch.int32, "perm"),
              C.c_void_p(ws.data_ptr()), ws.numel(), _stream())
    return keys, perm


def bucket_plan(ids_a, range_a, ids_b=None, range_b=0, list_single_a=True):
    """rc_bucket_plan: occurrences of the ids grouped by row without a device-wide sort.
    -> dict(rows_a int32[n_rows_a, 4] (row, start, n, 0), rows_b, occ int32[n_a + n_b], single uint8[n_a] | None)
    (the uint32 fields are returned as int32 bit patterns; row counts are read back, so this wrapper synchronises --
    the training step uses the plan through rc_bprmf_train_step without any host round trip)."""
    a = ids_a.reshape(-1)
    dev = a.device
    n_a = a.numel()
    b = ids_b.reshape(-1) if ids_b is not None else None
    n_b = b.numel() if b is not None else 0
    lib = _lib.load()
    if not lib.rc_bucket_plan_supported(n_a, n_b, int(range_a), int(range_b)):
        raise _lib.RechorusHipError("rc_bucket_plan_supported", -4, "id ranges too wide for one bucket level")
    rows_a = torch.zeros((max(n_a, 1), 4), dtype=torch.int32, device=dev)
    rows_b = torch.zeros((max(n_b, 1), 4), dtype=torch.int32, device=dev)
    cnt = torch.zeros(2, dtype=torch.int32, device=dev)
    occ = torch.full((max(n_a + n_b, 1),), -1, dtype=torch.int32, device=dev)
    single = None
    if not list_single_a:
        single = torch.empty(max(lib.rc_bucket_plan_flags_bytes(n_a), 1), dtype=torch.uint8, device=dev)
    ws = workspace(lib.rc_bucket_plan_workspace_bytes(n_a, n_b), dev, "plan")
    _lib.call("rc_bucket_plan", _ptr(a, torch.int64, "ids_a") if n_a else None, n_a, int(range_a),
              _ptr(b, torch.int64, "ids_b") if n_b else None, n_b, int(range_b), 1 if list_single_a else 0,
              C.c_void_p(single.data_ptr()) if single is not None else None,
              C.c_void_p(rows_a.data_ptr()), C.c_void_p(cnt.data_ptr()),
              C.c_void_p(rows_b.data_ptr()), C.c_void_p(cnt[1:].data_ptr()),
              C.c_void_p(occ.data_ptr()), C.c_void_p(ws.data_ptr()), ws.numel(), _stream())
    na, nb = (int(x) for x in cnt.tolist())
    return {"rows_a": rows_a[:na], "rows_b": rows_b[:nb], "occ": occ[:n_a + n_b],
            "single": None if single is None else single[:n_a]}


def bucket_multi_bitmap(ids, n_rows):
    """rc_bucket_multi_bitmap: int32 words, bit (id & 31) of word id >> 5 = 1 iff row id occurs at least twice in `ids`
    (words of id ranges the batch does not touch are unspecified: the buffer is NOT zero-filled)."""
    a = ids.reshape(-1)
    lib = _lib.load()
    n = a.numel()
    bm = torch.empty(max(lib.rc_bucket_bitmap_bytes(int(n_rows)) // 4, 1), dtype=torch.int32, device=a.device)
    ws = workspace(lib.rc_bucket_plan_workspace_bytes(n, 0), a.device, "plan")
    _lib.call("rc_bucket_multi_bitmap", _ptr(a, torch.int64, "ids") if n else None, n, int(n_rows), C.c_void_p(bm.data_ptr()),
              C.c_void_p(ws.data_ptr()), ws.numel(), _stream())
    return bm


class Plan:
    """Device-side result of rc_bucket_plan (no host round trip: the row counts stay in device memory and the consumers
    read them there).  Buffers are cached per (device, tag) and reused every step -- two plans that are alive at the same
    time need different tags.  list_single_a=False: rows of list a that occur once are not listed but flagged in
    self.single (uint8 per position), for consumers that update them elsewhere (rc_owner_backward).
    Negative ids take no part."""

    def __init__(self, ids_a, range_a, ids_b=None, range_b=0, tag="plan", list_single_a=True):
        a = ids_a.reshape(-1)
        b = ids_b.reshape(-1) if ids_b is not None else None
        dev = a.device
        self.n_a, self.n_b = a.numel(), (b.numel() if b is not None else 0)
        lib = _lib.load()
        if not lib.rc_bucket_plan_supported(self.n_a, self.n_b, int(range_a), int(range_b)):
            raise _lib.RechorusHipError("rc_bucket_plan_supported", -4, "no plan geometry for these id ranges / list lengths")
        n = self.n_a + self.n_b
        na16, nb16 = 16 * max(self.n_a, 1), 16 * max(self.n_b, 1)
        r256 = lambda x: (x + 255) // 256 * 256
        o_b = r256(na16)
        o_occ = o_b + r256(nb16)
        o_cnt = o_occ + r256(4 * max(n, 1))
        o_single = o_cnt + 256
        n_single = 0 if list_single_a else max(int(lib.rc_bucket_plan_flags_bytes(self.n_a)), 1)
        buf = workspace(o_single + n_single, dev, tag + ".out")
        self.rows_a, self.rows_b = buf[:na16], buf[o_b:o_b + nb16]
        self.occ, self.cnt = buf[o_occ:o_occ + 4 * max(n, 1)], buf[o_cnt:o_cnt + 8]
        self.single = None if list_single_a else buf[o_single:o_single + n_single][:self.n_a]
        self.tag, self.device = tag, dev
        ws = workspace(lib.rc_bucket_plan_workspace_bytes(self.n_a, self.n_b), dev, tag + ".ws")
        p = lambda t: C.c_void_p(t.data_ptr())
        _lib.call("rc_bucket_plan", _ptr(a, torch.int64, "ids_a") if self.n_a else None, self.n_a, int(range_a),
                  _ptr(b, torch.int64, "ids_b") if self.n_b else None, self.n_b, int(range_b), 1 if list_single_a else 0,
                  None if list_single_a else p(buf[o_single:]),
                  p(self.rows_a), p(self.cnt), p(self.rows_b), p(self.cnt[4:]), p(self.occ), p(ws), ws.numel(), _stream())
        self._ws = ws
        if _PLAN_CHECK:     # RC_PLAN_CHECK=1: every plan reads its status word back (a host sync per plan: debugging only)
            self.check()

    def status(self):
        """the plan's status word (rc_bucket_plan_status_ptr; synchronises): 0 ok, 1 an id outside its table, 2 hashed bucket overflow"""
        if self.n_a + self.n_b == 0:
            return 0
        addr = _lib.load().rc_bucket_plan_status_ptr(C.c_void_p(self._ws.data_ptr()), self.n_a, self.n_b)
        off = (addr - self._ws.data_ptr()) // 4
        return int(self._ws.view(torch.int32)[off].item())

    def check(self):
        st = self.status()
        if st:
            raise _lib.RechorusHipError("rc_bucket_plan", -1, {1: "an id lies outside its table (nn.Embedding would raise)",
                                                               2: "a hashed bucket overflowed: the plan is incomplete"}.get(st, str(st)))

    def row_sums(self, side, out, coef=None, src=None, src_index=None, div=1, src2=None, n_split=None):
        """rc_plan_row_sums on list `side`: out[row] = summed gradient row of every listed row (other rows untouched)"""
        d = out.shape[1]
        n = self.n_a + self.n_b
        if n == 0:
            return out
        rows, cnt, _ = self._side(side)
        if n_split is None:
            n_split = n if src2 is None else 0
        ws = workspace(_lib.load().rc_plan_update_workspace_bytes(n, d), out.device, self.tag + ".upd")
        f32 = torch.float32
        p = lambda t: C.c_void_p(t.data_ptr())
        _lib.call("rc_plan_row_sums", _ptr(out, f32, "out"), d, p(rows), p(cnt), p(self.occ), n, _ptr(coef, f32, "coef", True),
                  _ptr(src, f32, "src", True), _ptr(src_index, torch.int64, "src_index", True), int(div),
                  _ptr(src2, f32, "src2", True), int(n_split), p(ws), ws.numel(), _stream())
        return out

    def distinct(self, side):
        """-> (uniq int64 [list length] (the first *count entries are valid), inverse int64 [list length], count int32 [1] on the
        device): the distinct ids of the list in the plan's record order and every position's index into them"""
        rows, cnt, base = self._side(side)
        n_list = self.n_a if side == "a" else self.n_b
        uniq = torch.empty(max(n_list, 1), dtype=torch.int64, device=self.device)
        inverse = torch.empty(max(n_list, 1), dtype=torch.int64, device=self.device)
        p = lambda t: C.c_void_p(t.data_ptr())
        _lib.call("rc_plan_distinct", p(rows), p(cnt), p(self.occ), base, n_list, p(uniq), p(inverse), _stream())
        return uniq, inverse[:n_list], cnt[:4].view(torch.int32)

    def _side(self, side):
        return (self.rows_a, self.cnt, 0) if side == "a" else (self.rows_b, self.cnt[4:], self.n_a)

    def update_pair(self, side, Wa, Wb, src_a, src_b, hyper, ma=None, va=None, mb=None, vb=None, ws_tag=""):
        """rc_plan_update_pair on list `side` ('a' | 'b'): two tables sharing the ids, per-occurrence gradient rows.
        ws_tag: suffix of the scratch buffer's cache key -- two updates that run at the same time on two streams need two.
        src_b=None: src_a is ONE block [n, >= 2 d] whose rows hold (gradient of table a | gradient of table b) side by side
        (rc_plan_update_pair_block: the block is read where it lies, no contiguous halves)"""
        d = Wa.shape[1]
        if src_b is None:
            if src_a.dim() != 2 or src_a.stride(1) != 1 or src_a.shape[1] < 2 * d or src_a.stride(0) % 4:
                raise ValueError("update_pair: a block source is [n, >= 2 d] with unit column stride and a row stride that is a multiple of 4")
            n = self.n_a + self.n_b
            rows, cnt, base = self._side(side)
            ws = workspace(_lib.load().rc_plan_update_workspace_bytes(n, 2 * d), Wa.device, self.tag + ".upd" + ws_tag)
            f32 = torch.float32
            p = lambda t: C.c_void_p(t.data_ptr())
            zc = getattr(self, "upd_counters", None)
            c = zc.pop(side) if (zc is not None and side in zc) else None
            if c is not None:
                c.record_stream(torch.cuda.current_stream(c.device))
            _lib.call("rc_plan_update_pair_block", _ptr(Wa, f32, "W_a"), _ptr(ma, f32, "m_a", True), _ptr(va, f32, "v_a", True),
                      _ptr(Wb, f32, "W_b"), _ptr(mb, f32, "m_b", True), _ptr(vb, f32, "v_b", True), d, p(rows), p(cnt), p(self.occ), n,
                      C.c_void_p(src_a.data_ptr()), int(src_a.stride(0)), base, C.byref(hyper), p(c) if c is not None else None,
                      p(ws), ws.numel(), _stream())
            return
        n = self.n_a + self.n_b
        rows, cnt, base = self._side(side)
        ws = workspace(_lib.load().rc_plan_update_workspace_bytes(n, 2 * d), Wa.device, self.tag + ".upd" + ws_tag)
        f32 = torch.float32
        p = lambda t: C.c_void_p(t.data_ptr())
        zc = getattr(self, "upd_counters", None)
        if zc is not None and side in zc:
            # the update's ticket counters were zero-filled with the plan (prezero_update_counters: beside other work, on the plan's
            # stream); one use each -- the memset in front of the update, a launch on the step's critical path, is left out
            c = zc.pop(side)
            c.record_stream(torch.cuda.current_stream(c.device))
            _lib.call("rc_plan_update_pair_zeroed", _ptr(Wa, f32, "W_a"), _ptr(ma, f32, "m_a", True), _ptr(va, f32, "v_a", True),
                      _ptr(Wb, f32, "W_b"), _ptr(mb, f32, "m_b", True), _ptr(vb, f32, "v_b", True), d, p(rows), p(cnt), p(self.occ), n,
                      _ptr(src_a, f32, "src_a"), _ptr(src_b, f32, "src_b"), base, C.byref(hyper), p(c), p(ws), ws.numel(), _stream())
            return
        _lib.call("rc_plan_update_pair", _ptr(Wa, f32, "W_a"), _ptr(ma, f32, "m_a", True), _ptr(va, f32, "v_a", True),
                  _ptr(Wb, f32, "W_b"), _ptr(mb, f32, "m_b", True), _ptr(vb, f32, "v_b", True), d, p(rows), p(cnt), p(self.occ), n,
                  _ptr(src_a, f32, "src_a"), _ptr(src_b, f32, "src_b"), base, C.byref(hyper), p(ws), ws.numel(), _stream())

    def prezero_update_counters(self):
        """zero-fill the ticket counters of one later update_pair per side NOW, on the current stream (the plan's: beside the work the
        plan is built beside) -- update_pair then runs without its own memset"""
        z = torch.zeros(16, dtype=torch.int32, device=self.occ.device)
        self.upd_counters = {"a": z[:8], "b": z[8:]}

    def update(self, side, W, hyper, m=None, v=None, coef=None, src=None, src_index=None, div=1, src2=None, n_split=None):
        """rc_plan_update on list `side`: gradient sources as in segmented_update2 (positions of list b start at n_a)"""
        d = W.shape[1]
        n = self.n_a + self.n_b
        rows, cnt, _ = self._side(side)
        if n_split is None:
            n_split = n if src2 is None else 0
        ws = workspace(_lib.load().rc_plan_update_workspace_bytes(n, d), W.device, self.tag + ".upd")
        f32 = torch.float32
        p = lambda t: C.c_void_p(t.data_ptr())
        _lib.call("rc_plan_update", _ptr(W, f32, "W"), _ptr(m, f32, "m", True), _ptr(v, f32, "v", True), d, p(rows), p(cnt),
                  p(self.occ), n, _ptr(coef, f32, "coef", True), _ptr(src, f32, "src", True),
                  _ptr(src_index, torch.int64, "src_index", True), int(div), _ptr(src2, f32, "src2", True), int(n_split),
                  C.byref(hyper), p(ws), ws.numel(), _stream())


def plan_supported(n_a, n_b, range_a, range_b):
    return bool(_lib.load().rc_bucket_plan_supported(int(n_a), int(n_b), int(range_a), int(range_b)))


def segment_heads(keys, perm, only_multi=False, want_single=True, want_heads=True):
    """One pass over sorted ids -> (single uint8[n]|None, heads int32[n]|None, n_heads int32[1]|None)."""
    n = keys.numel()
    dev = keys.device
    single = torch.empty(n, dtype=torch.uint8, device=dev) if want_single else None
    heads = torch.empty(max(n, 1), dtype=torch.int32, device=dev) if want_heads else None
    n_heads = torch.zeros(1, dtype=torch.int32, device=dev) if want_heads else None
    _lib.call("rc_segment_heads", _ptr(keys, torch.int32, "keys"), _ptr(perm, torch.int32, "perm"), n,
              1 if only_multi else 0, _ptr(single, torch.uint8, "single", True),
              _ptr(heads, torch.int32, "heads", True), _ptr(n_heads, torch.int32, "n_heads", True),
              _stream())
    return single, heads, n_heads


def mark_singletons(keys, perm):
    """uint8 flag per occurrence: 1 iff its row occurs exactly once in the batch."""
    return segment_heads(keys, perm, want_heads=False)[0]


def bprmf_fwd_bwd_update(U, I, uid, iid, single, hyper, mI=None, vI=None, inv_b=None, want_pred=False, multi=None):
    """Fused fwd/loss/bwd that also applies the optimizer to single-occurrence item rows: `single` = uint8 flag per batch
    position (segment_heads), or `multi` = the bitmap over item ids of bucket_multi_bitmap (single = None)."""
    B, Cn = iid.shape
    d = U.shape[1]
    if inv_b is None:
        inv_b = 1.0 / B
    dev = U.device
    f32 = torch.float32
    pred = torch.empty((B, Cn), dtype=f32, device=dev) if want_pred else None
    loss_vec = torch.empty(B, dtype=f32, device=dev)
    gpred = torch.empty((B, Cn), dtype=f32, device=dev)
    ugrad = torch.empty((B, d), dtype=f32, device=dev)
    _lib.call("rc_bprmf_fwd_bwd_update_bitmap" if multi is not None else "rc_bprmf_fwd_bwd_update", _ptr(U, f32, "U"), _ptr(I, f32, "I"),
              _ptr(mI, f32, "mI", True), _ptr(vI, f32, "vI", True),
              _ptr(uid, torch.int64, "uid"), _ptr(iid, torch.int64, "iid"),
              _ptr(multi, torch.int32, "multi") if multi is not None else _ptr(single, torch.uint8, "single"),
              B, Cn, d, float(inv_b), C.byref(hyper),
              _ptr(pred, f32, "pred", True), _ptr(loss_vec, f32, "loss_vec"),
              _ptr(gpred, f32, "gpred"), _ptr(ugrad, f32, "ugrad"), _stream())
    return pred, loss_vec, gpred, ugrad


def segmented_update(keys, perm, src, hyper=None, W=None, m=None, v=None, coef=None,
                     src_index=None, div=1, dense_grad=None, skip_singletons=False, heads=None,
                     n_heads=None):
    """rc_segmented_update: per distinct row r, grad_r = sum coef[o]*src[srow(o)], then either
    write dense_grad[r] or apply the optimizer to W[r] (and m, v) in place."""
    n_occ = keys.numel()
    d = src.shape[-1]
    dev = keys.device
    lib = _lib.load()
    ws = workspace(lib.rc_segmented_workspace_bytes(n_occ, d), dev, "seg")
    hp = C.byref(hyper) if hyper is not None else None
    _lib.call("rc_segmented_update",
              _ptr(W, torch.float32, "W", allow_none=True),
              _ptr(m, torch.float32, "m", allow_none=True),
              _ptr(v, torch.float32, "v", allow_none=True), d,
              _ptr(keys, torch.int32, "keys"), _ptr(perm, torch.int32, "perm"), n_occ,
              _ptr(coef, torch.float32, "coef", allow_none=True), _ptr(src, torch.float32, "src"),
              _ptr(src_index, torch.int64, "src_index", allow_none=True), int(div), hp,
              _ptr(dense_grad, torch.float32, "dense_grad", allow_none=True),
              _ptr(heads, torch.int32, "heads", True), _ptr(n_heads, torch.int32, "n_heads", True),
              _lib.RC_SEG_SKIP_SINGLETONS if skip_singletons else 0,
              C.c_void_p(ws.data_ptr()), ws.numel(), _stream())


def segmented_update_pair(keys, perm, src_a, src_b, hyper=None, W=(None, None), m=(None, None), v=(None, None),
                          dense_grad=(None, None), heads=None, n_heads=None):
    """rc_segmented_update_pair: two tables with the same ids (NeuMF's mf / mlp embeddings) in one pass.
    src_a, src_b [n_occ, d] per-occurrence gradient rows; W / m / v / dense_grad are (table a, table b) pairs."""
    n_occ = keys.numel()
    d = src_a.shape[-1]
    f32 = torch.float32
    ws = workspace(_lib.load().rc_segmented_workspace_bytes(n_occ, 2 * d), keys.device, "seg")
    _lib.call("rc_segmented_update_pair", _ptr(W[0], f32, "W_a", True), _ptr(m[0], f32, "m_a", True), _ptr(v[0], f32, "v_a", True),
              _ptr(W[1], f32, "W_b", True), _ptr(m[1], f32, "m_b", True), _ptr(v[1], f32, "v_b", True), d,
              _ptr(keys, torch.int32, "keys"), _ptr(perm, torch.int32, "perm"), n_occ, _ptr(src_a, f32, "src_a"),
              _ptr(src_b, f32, "src_b"), C.byref(hyper) if hyper is not None else None,
              _ptr(dense_grad[0], f32, "dense_grad_a", True), _ptr(dense_grad[1], f32, "dense_grad_b", True),
              _ptr(heads, torch.int32, "heads", True), _ptr(n_heads, torch.int32, "n_heads", True),
              C.c_void_p(ws.data_ptr()), ws.numel(), _stream())


def segmented_pair_supported(d):
    return 2 * d in (16, 32, 64, 128, 256)


_EDB_PLAN_MIN = int(os.environ.get("RC_EDB_PLAN_MIN", "8192"))
_EDB_SMALL = os.environ.get("RC_EDB_SMALL", "1") != "0"   # embedding_dense_backward of a small id list: rc_small_row_sums
_EDB_SMALL_MAX = int(os.environ.get("RC_EDB_SMALL_MAX", "8192"))
# A/B switches: RC_TABLE_UPDATE=sort puts every trainer's table update behind the radix sort, =plan behind the bucket plan.
# Default: NeuMF on the plan (hashed buckets: 0.33 M lookups over 10 M - 100 M rows); SASRec behind the sort -- 0.6 M
# occurrences over 8.7 K rows are all hot rows, where the sort-driven update measured faster (table_update 0.27 ms against
# 0.44 ms through the narrow-bucket plan, profiles/r03d_bench_sasrec*.json)
_USE_PLAN = os.environ.get("RC_TABLE_UPDATE", "auto") != "sort"
# sorted ids of a table whose rows collect many occurrences each: rc_segmented_update_rows (RC_SEG_ROWS=0: the head-list route)
_SEG_ROWS = os.environ.get("RC_SEG_ROWS", "1") != "0"
# SasrecTrainer: id sort beside the encoder, position gradient beside the item update, on a second stream (RC_SAS_OVERLAP=0: one stream)
_SAS_OVERLAP = os.environ.get("RC_SAS_OVERLAP", "1") != "0"
_PLAN_CHECK = os.environ.get("RC_PLAN_CHECK", "0") == "1"        # every engine.Plan verifies its status word (host sync; debugging)
_NEUMF_OVERLAP = os.environ.get("RC_NEUMF_OVERLAP", "1") != "0"   # NeumfTrainer: bucket plan beside the head kernels
_NEUMF_FUSED = os.environ.get("RC_NEUMF_FUSED", "1") != "0"       # NeumfTrainer: rc_neumf_train_step (A/B against the three-kernel step)
_SAS_OVERLAP_MIN = int(os.environ.get("RC_SAS_OVERLAP_MIN", "131072"))   # candidate + history occurrences of the batch
_SEG_ROWS_MIN_PER_ROW = int(os.environ.get("RC_SEG_ROWS_MIN_PER_ROW", "8"))
_SASREC_PLAN = os.environ.get("RC_TABLE_UPDATE", "auto") == "plan"
# SasrecTrainer on the one-wave-per-row route: row bounds from a counting sort of item_id + history_items (RowsPlan) instead of the
# radix sort and its tensor glue (RC_SAS_ROWS_PLAN=0: the sorted route, same results bit for bit)
_SAS_ROWS_PLAN = os.environ.get("RC_SAS_ROWS_PLAN", "1") != "0"
# the two-stream schedule of SasrecTrainer (SasrecTrainer._step_item_stream): one stream owns everything about the item table -- the
# plan beside the encoder, then the table update as soon as the history rows' gradient is complete (rc_sasrec_batch_bwd_part) -- while
# the other finishes the encoder's parameter gradients, the position gradient and the dense step: ONE join at the end of the step.
# "2" (default): the encoder rides the caller's stream; "3": the item-table work does; "1": the round-5 schedule (plan on the side
# stream, joined before the table update on the main stream, position gradient on the side).  Same box, config 3, ms per replayed
# step: "1" 0.240, "2" 0.219, "3" 0.225 (profiles/r09_sasrec_two_stream_schedules.txt)
_SAS_SCHED = os.environ.get("RC_SAS_SCHED", "2")


def unique_ids(ids, n_rows, tag="unique"):
    """distinct ids + inverse index through the bucket plan (rc_bucket_plan + rc_plan_distinct) -- what
    torch.unique(ids, return_inverse=True) gives, except that the distinct ids come in the plan's order, not sorted.
    Reads the count back (one host sync, like torch.unique): callers on a hot path run it on a side stream."""
    flat = ids.reshape(-1)
    if flat.numel() == 0:
        return flat.clone(), flat.clone()
    if not plan_supported(flat.numel(), 0, n_rows, 0):   # no plan geometry for this list: torch's sort-based unique (sorted ids)
        return torch.unique(flat, return_inverse=True)
    uniq, inverse, cnt = Plan(flat, n_rows, tag=tag).distinct("a")
    return uniq[:int(cnt.item())], inverse


def unique_ids_begin(ids, n_rows, tag="unique"):
    """unique_ids without the host sync: -> (uniq_full int64 [n] (the first count entries are valid), inverse, count int32 [1] on the
    device -- a copy: the plan's own counter is overwritten by the next plan with this tag), or None where the list has no plan
    geometry (the caller falls back to unique_ids).  Several lists' counts can then be read back with ONE synchronisation."""
    flat = ids.reshape(-1)
    if flat.numel() == 0 or not plan_supported(flat.numel(), 0, n_rows, 0):
        return None
    uniq, inverse, cnt = Plan(flat, n_rows, tag=tag).distinct("a")
    return uniq, inverse, cnt[:1].clone()


def small_route_ok(n_ids, n_rows, d):
    """embedding_dense_backward(route="small") takes rc_small_row_sums for this shape"""
    return bool(_EDB_SMALL and 0 < n_ids <= _EDB_SMALL_MAX and _lib.load().rc_small_row_sums_supported(int(n_ids), int(n_rows), int(d)))


def embedding_dense_backward(grad_out, ids, n_rows, route=None, presorted=None, small_again=False, small_tag="edb_small"):
    """aten::embedding_dense_backward: G [n_rows, d] = index_add of the per-occurrence gradient rows, in ascending
    position order per row (no float atomics) -- bucket plan + rc_plan_row_sums; radix sort + segmented sum where no
    plan geometry exists.  presorted = sort_ids(ids, n_rows) of a caller that sorted the same ids already (route "sort").
    route "small" with small_again=True: the caller vouches that the preceding call on workspace `small_tag` grouped these very
    ids (a second table family gathered with the same ids): only the row sums run (rc_small_row_sums_again)."""
    d = grad_out.shape[-1]
    flat = ids.reshape(-1)
    G = torch.zeros((n_rows, d), dtype=torch.float32, device=grad_out.device)
    if flat.numel() == 0:
        return G
    go = grad_out.reshape(-1, d).contiguous()
    # route="small": a small id list WITHOUT very hot rows (the reference's own batch sizes: user / candidate ids, the fields of a
    # CTR row) in two launches -- 128 workgroups group the ids in LDS, one lane-group per touched row sums its occurrences
    # (rc_small_row_sums).  Up to 8,192 ids a plan workgroup's buffers hold the whole list, so the grouping cannot leave LDS; a row
    # with thousands of occurrences (the padding id of a padded history: measured 0.27 against 0.11 s per SASRec epoch) is still
    # summed by ONE wave there, which is why the caller has to ask for this route
    n_ids = flat.numel()
    if route == "small" and small_route_ok(n_ids, n_rows, d):
        ws = workspace(_lib.load().rc_small_row_sums_workspace_bytes(n_ids), go.device, small_tag)
        if small_again:
            _lib.call("rc_small_row_sums_again", n_ids, int(n_rows), _ptr(go, torch.float32, "grad_out"), d,
                      _ptr(G, torch.float32, "G"), C.c_void_p(ws.data_ptr()), ws.numel(), _stream())
        else:
            _lib.call("rc_small_row_sums", _ptr(flat, torch.int64, "ids"), n_ids, int(n_rows), _ptr(go, torch.float32, "grad_out"), d,
                      _ptr(G, torch.float32, "G"), C.c_void_p(ws.data_ptr()), ws.numel(), _stream())
        return G
    # (below a few thousand ids both routes are a handful of latency-bound launches; the plan pays off with the batch)
    # route="sort": id lists with very hot rows (the categorical fields of the CTR models: 131,072 occurrences of 7 weekdays) --
    # a plan bucket counts its ids with LDS atomics, which such a row serialises (15.8 ms per DeepFM step at B = 131,072)
    if route != "sort" and d in (16, 32, 64, 128, 256) and flat.numel() >= _EDB_PLAN_MIN and plan_supported(flat.numel(), 0, n_rows, 0):
        return Plan(flat, n_rows, tag="edb").row_sums("a", G, src2=go)
    keys, perm = presorted if presorted is not None else sort_ids(flat, n_rows)
    segmented_update(keys, perm, go, dense_grad=G)
    return G


SMALL_NUMERIC_MAX = 4      # numeric fields that can ride in the small route's row-sums launch (kSmallNumeric)


def small_row_sums_pair(cid, n_rows, src_a, src_b, into=None, numeric=None):
    """two dense gradients [n_rows, d_a], [n_rows, d_b] of per-occurrence rows src_a [n, d_a], src_b [n, d_b] that share their ids:
    ONE zero fill (both live in one buffer), ONE grouping (rc_small_row_sums, then rc_small_row_sums_again for the second).
    into: a float buffer of n_rows * (d_a + d_b) elements that is NOT zero-filled -- only the rows of `cid` are written (the
    kernels assign row sums), for a consumer that reads only those (dense_update_rows(touched=True))"""
    n = cid.numel()
    d_a, d_b = src_a.shape[1], src_b.shape[1]
    G = torch.zeros(n_rows * (d_a + d_b), dtype=torch.float32, device=src_a.device) if into is None else into
    Ga, Gb = G[:n_rows * d_a].view(n_rows, d_a), G[n_rows * d_a:].view(n_rows, d_b)
    ws = workspace(_lib.load().rc_small_row_sums_workspace_bytes(n), src_a.device, "edb_small_pair")
    flat = cid.reshape(-1)
    if numeric is not None:
        # numeric = (values, fields, F, n_cand): the weight gradients of the numeric fields of the same gradient blocks (src_a = gV
        # [rows * F, d], src_b = gL [rows * F, 1]) by extra workgroups of the row-sums launch -> (Ga, Gb, dW list, dw1 list)
        values, fields, F, n_cand = numeric
        J = len(values)
        if not (d_b == 1 and 16 <= d_a <= 128 and d_a % 4 == 0 and 1 <= J <= SMALL_NUMERIC_MAX):
            raise ValueError("small_row_sums_pair: the numeric fields ride with d in 16 .. 128 and at most %d of them" % SMALL_NUMERIC_MAX)
        f32 = torch.float32
        B = values[0].shape[0]
        dW = torch.empty((J, d_a, 1), dtype=f32, device=src_a.device)
        dw1 = torch.empty((J, 1, 1), dtype=f32, device=src_a.device)
        val_arr = (C.c_void_p * J)(*[_ptr(x, x.dtype, "values").value for x in values])
        per_row = (C.c_int * J)(*[1 if x.dim() == 1 else 0 for x in values])
        kind_arr = (C.c_int * J)(*[field_kind(x) for x in values])
        field_arr = (C.c_int * J)(*[int(f) for f in fields])
        dW_arr = (C.c_void_p * J)(*[dW[j].data_ptr() for j in range(J)])
        dw1_arr = (C.c_void_p * J)(*[dw1[j].data_ptr() for j in range(J)])
        _lib.call("rc_small_row_sums_pair_numeric", _ptr(flat, torch.int64, "ids"), n, int(n_rows), _ptr(src_a, f32, "src_a"), d_a,
                  C.c_void_p(Ga.data_ptr()), _ptr(src_b, f32, "src_b"), C.c_void_p(Gb.data_ptr()), val_arr, per_row, kind_arr, field_arr, J,
                  int(F), B, int(n_cand), dW_arr, dw1_arr, C.c_void_p(ws.data_ptr()), ws.numel(), _stream())
        return Ga, Gb, [dW[j] for j in range(J)], [dw1[j] for j in range(J)]
    if d_b == 1 and d_a >= 16:     # the one-float-wide table rides in the vectors' row-sums launch
        _lib.call("rc_small_row_sums_pair", _ptr(flat, torch.int64, "ids"), n, int(n_rows), _ptr(src_a, torch.float32, "src_a"), d_a,
                  C.c_void_p(Ga.data_ptr()), _ptr(src_b, torch.float32, "src_b"), C.c_void_p(Gb.data_ptr()), C.c_void_p(ws.data_ptr()),
                  ws.numel(), _stream())
        return Ga, Gb
    _lib.call("rc_small_row_sums", _ptr(flat, torch.int64, "ids"), n, int(n_rows), _ptr(src_a, torch.float32, "src_a"), d_a,
              C.c_void_p(Ga.data_ptr()), C.c_void_p(ws.data_ptr()), ws.numel(), _stream())
    _lib.call("rc_small_row_sums_again", n, int(n_rows), _ptr(src_b, torch.float32, "src_b"), d_b, C.c_void_p(Gb.data_ptr()),
              C.c_void_p(ws.data_ptr()), ws.numel(), _stream())
    return Ga, Gb


def small_row_sums_planned(plan_ws, n, n_rows, src_a, src_b, d, shape, into=None, numeric=None, fm=None):
    """small_row_sums_pair on a grouping that gather_fields(plan=True) built during the forward pass (rc_small_row_sums_planned: ONE
    launch): src_a [n, d] | None, src_b [n, 1]; shape = (F, B, n_cand) with n = B * n_cand * F; numeric = (values, fields, F, n_cand)
    as small_row_sums_pair;
    fm = (V [rows, F, d], S [rows, d], g [rows]): the FM pairwise term's backward is added to src_a's rows where they are read
    (src_a may then be None) -> (Ga, Gb) or (Ga, Gb, dW list, dw1 list)"""
    dev, f32 = src_b.device, torch.float32
    G = torch.zeros(n_rows * (d + 1), dtype=f32, device=dev) if into is None else into
    Ga, Gb = G[:n_rows * d].view(n_rows, d), G[n_rows * d:].view(n_rows, 1)
    J = 0
    val_arr = per_row = kind_arr = field_arr = dW_arr = dw1_arr = dW = dw1 = None
    F, B, n_cand = shape
    if numeric is not None:
        values, fields = numeric[:2]
        J = len(values)
        if not (1 <= J <= SMALL_NUMERIC_MAX):
            raise ValueError("small_row_sums_planned: at most %d numeric fields ride along" % SMALL_NUMERIC_MAX)
        dW = torch.empty((J, d, 1), dtype=f32, device=dev)
        dw1 = torch.empty((J, 1, 1), dtype=f32, device=dev)
        val_arr = (C.c_void_p * J)(*[_ptr(x, x.dtype, "values").value for x in values])
        per_row = (C.c_int * J)(*[1 if x.dim() == 1 else 0 for x in values])
        kind_arr = (C.c_int * J)(*[field_kind(x) for x in values])
        field_arr = (C.c_int * J)(*[int(f) for f in fields])
        dW_arr = (C.c_void_p * J)(*[dW[j].data_ptr() for j in range(J)])
        dw1_arr = (C.c_void_p * J)(*[dw1[j].data_ptr() for j in range(J)])
    fm_V = fm_S = fm_g = None
    if fm is not None:
        fm_V, fm_S, fm_g = fm
    _lib.call("rc_small_row_sums_planned", n, int(n_rows), _ptr(src_a, f32, "src_a", True), int(d), C.c_void_p(Ga.data_ptr()),
              _ptr(src_b, f32, "src_b"), C.c_void_p(Gb.data_ptr()), val_arr, per_row, kind_arr, field_arr, J, int(F), int(B), int(n_cand),
              dW_arr, dw1_arr, _ptr(fm_V, f32, "fm_V", True), _ptr(fm_S, f32, "fm_S", True), _ptr(fm_g, f32, "fm_g", True),
              C.c_void_p(plan_ws.data_ptr()), plan_ws.numel(), _stream())
    if J:
        return Ga, Gb, [dW[j] for j in range(J)], [dw1[j] for j in range(J)]
    return Ga, Gb


def dense_update(W, G, hyper, m=None, v=None):
    """Exact torch.optim step over a whole tensor (helpers/BaseRunner.py:206)."""
    _lib.call("rc_dense_update", _ptr(W, torch.float32, "W"), _ptr(G, torch.float32, "G"),
              _ptr(m, torch.float32, "m", allow_none=True),
              _ptr(v, torch.float32, "v", allow_none=True), W.numel(), C.byref(hyper), _stream())


def dense_update_multi(items, opt, step_dev=None, increment=True):
    """items: list of (W, G, hyper, m | None, v | None) -> one launch per 36 tensors (rc_dense_update_multi).
    step_dev: int64 device tensor [1] holding Adam's step count (hipGraph-capturable mode): it is incremented
    on the stream (unless the caller already did: increment=False), then read by the update kernel"""
    T = len(items)
    if T == 0:
        return
    f32 = torch.float32
    Wa = (C.c_void_p * T)(*[_ptr(w, f32, "W").value for w, _, _, _, _ in items])
    Ga = (C.c_void_p * T)(*[_ptr(g, f32, "G").value for _, g, _, _, _ in items])
    Ma = (C.c_void_p * T)(*[_ptr(m, f32, "m", True).value for _, _, _, m, _ in items])
    Va = (C.c_void_p * T)(*[_ptr(v, f32, "v", True).value for _, _, _, _, v in items])
    na = (C.c_int64 * T)(*[w.numel() for w, _, _, _, _ in items])
    ha = (OptHyper * T)(*[h for _, _, h, _, _ in items])
    if step_dev is None:
        _lib.call("rc_dense_update_multi", Wa, Ga, Ma, Va, na, ha, T, _stream())
        return
    if increment:
        _lib.call("rc_step_increment", _ptr(step_dev, torch.int64, "step_dev"), _stream())
    _lib.call("rc_dense_update_multi_dev", Wa, Ga, Ma, Va, na, ha, T, _ptr(step_dev, torch.int64, "step_dev"), _stream())


def dense_update_rows(items, step_dev, touched=2, max_blocks=0):
    """items: list of (W, G | None, hyper, m, v, flags | None, row_w) -> rc_dense_update_rows_dev: Adam over W without a dense
    gradient -- touched=2: every row, g = G's row where the row's int32 flag equals the step and 0 elsewhere (the rest of G is
    never read); touched=1 / 0: only the stamped rows (from G) / only the others (g = 0).  flags None = the whole tensor from G.
    The step count was already incremented for this step.  max_blocks > 0: grid cap."""
    T = len(items)
    if T == 0:
        return
    f32 = torch.float32
    Wa = (C.c_void_p * T)(*[_ptr(it[0], f32, "W").value for it in items])
    Ga = (C.c_void_p * T)(*[_ptr(it[1], f32, "G", True).value for it in items])
    Ma = (C.c_void_p * T)(*[_ptr(it[3], f32, "m").value for it in items])
    Va = (C.c_void_p * T)(*[_ptr(it[4], f32, "v").value for it in items])
    na = (C.c_int64 * T)(*[it[0].numel() for it in items])
    Fa = (C.c_void_p * T)(*[_ptr(it[5], torch.int32, "flags", True).value for it in items])
    ra = (C.c_int * T)(*[int(it[6]) for it in items])
    ha = (OptHyper * T)(*[it[2] for it in items])
    _lib.call("rc_dense_update_rows_dev", Wa, Ga, Ma, Va, na, Fa, ra, ha, T, int(touched), int(max_blocks),
              _ptr(step_dev, torch.int64, "step_dev"), _stream())


# ---- whole BPRMF step -----------------------------------------------------------------------

class BprmfTrainer:
    """Row-wise-optimizer BPRMF training on device tables U [n_users,d], I [n_items,d].

    One `step(uid, iid)` = one iteration of BaseRunner.fit's batch loop
    (helpers/BaseRunner.py:193-206) for models/general/BPRMF.py, as one C-ABI call.
    """

    def __init__(self, U, I, opt="SGD", lr=1e-3, l2=0.0, beta1=0.9, beta2=0.999, eps=None):
        if U.shape[1] != I.shape[1]:
            raise ValueError("user and item tables need the same emb_size")
        self.U, self.I = U, I
        self.d = U.shape[1]
        self.opt = opt
        self.hyper = make_hyper(opt, lr, l2, beta1, beta2, eps, step=0)
        self.mU = self.vU = self.mI = self.vI = None
        if opt in ("Adam", "Adagrad"):
            self.mU, self.mI = torch.zeros_like(U), torch.zeros_like(I)
        if opt == "Adam":
            self.vU, self.vI = torch.zeros_like(U), torch.zeros_like(I)
        self.loss = torch.zeros(1, dtype=torch.float32, device=U.device)
        self._ws = None
        self._ws_shape = None
        # look-ahead across steps: the library records what it prepared in THIS caller-owned ticket (no hidden state)
        self._ticket = _lib.StepTicket()
        self._gen = 0            # generation ids handed out for announced batches
        self._announced = None   # (uid, iid, uid._version, iid._version, generation) of the batch announced last

    def _workspace(self, B, Cn):
        if self._ws_shape != (B, Cn):
            nbytes = _lib.load().rc_bprmf_step_workspace_bytes(B, Cn, self.d)
            if nbytes == 0:
                raise _lib.RechorusHipError("rc_bprmf_step_workspace_bytes", -1, "bad shape")
            if self._ws is None or self._ws.numel() < nbytes:
                if self._ws is not None:
                    self._forget_ahead()
                    _ws_retired.append(self._ws)  # a captured hipGraph may still point into it
                self._ws = torch.empty(nbytes, dtype=torch.uint8, device=self.U.device)
            self._ws_shape = (B, Cn)
        return self._ws

    def _forget_ahead(self):
        """a plan prepared by step(next_batch=...) lives in this trainer's workspace: drop it before the memory can be reused"""
        self._announced = None
        if getattr(self, "_ticket", None) is not None and self._ticket.generation != 0:
            try:
                _lib.call("rc_bprmf_step_ahead_reset", C.byref(self._ticket), _stream())
            except Exception:  # interpreter shutdown: the library / torch may be gone already
                pass

    def __del__(self):
        self._forget_ahead()

    def _generation_of(self, uid, iid):
        """generation id of (uid, iid) if it IS the batch announced by the previous step: the same tensor objects (strong
        references are held, so an address cannot have been reused) with unchanged version counters (an in-place refill
        through torch bumps them).  0 = unknown: the step plans the batch itself."""
        a = self._announced
        if a is not None and a[0] is uid and a[1] is iid and a[2] == uid._version and a[3] == iid._version:
            return a[4]
        return 0

    def step(self, uid, iid, inv_b=None, pred=None, phase_ms=None, next_batch=None, generation=None, next_generation=None):
        """Runs one training step; returns the device loss tensor (shape [1], no sync).
        phase_ms: optional ctypes float[8] to receive per-phase hipEvent timings.
        next_batch = (uid, iid) of the FOLLOWING step (same shapes): the bucket plan of those ids is prepared beside this
        step's row updates (rc_bprmf_train_step_ahead); the tensors must stay alive and unchanged until that step.
        generation / next_generation: the caller's ids of this / the following batch's CONTENTS (non-zero ints that
        change with every new batch) -- needed when batches are written into the same buffers by something torch's
        version counters do not see (a kernel writing through data_ptr()); left None, the trainer identifies the
        announced batch by tensor identity + version.  Pass them always or never."""
        B, Cn = iid.shape
        if inv_b is None:
            inv_b = 1.0 / B
        ws = self._workspace(B, Cn)
        self.hyper.step += 1
        f32 = torch.float32
        if generation is None:
            generation = self._generation_of(uid, iid)
        nu = ni = None
        ngen = 0
        if next_batch is not None and tuple(next_batch[1].shape) == (B, Cn) and tuple(next_batch[0].shape) == (B,):
            nu, ni = next_batch
            if next_generation is None:
                self._gen += 1
                ngen = self._gen
            else:
                ngen = int(next_generation)
        self._announced = (nu, ni, nu._version, ni._version, ngen) if nu is not None else None
        _lib.call("rc_bprmf_train_step_ahead",
                  _ptr(self.U, f32, "U"), _ptr(self.I, f32, "I"),
                  _ptr(self.mU, f32, "mU", True), _ptr(self.vU, f32, "vU", True),
                  _ptr(self.mI, f32, "mI", True), _ptr(self.vI, f32, "vI", True),
                  _ptr(uid, torch.int64, "uid"), _ptr(iid, torch.int64, "iid"), int(generation),
                  _ptr(nu, torch.int64, "next_uid", True), _ptr(ni, torch.int64, "next_iid", True), ngen,
                  C.byref(self._ticket), B, Cn, self.d,
                  self.U.shape[0], self.I.shape[0], C.byref(self.hyper), float(inv_b),
                  _ptr(self.loss, f32, "loss"), _ptr(pred, f32, "pred", True),
                  C.c_void_p(ws.data_ptr()), ws.numel(), _stream(),
                  phase_ms if phase_ms is not None else None)
        return self.loss

    def profile_step(self, uid, iid, next_batch=None):
        """One step with hipEvent phase timing -> dict of milliseconds (synchronises).  With next_batch (and this batch
        announced by the previous step) the profiled step is a step of the steady state: its plan was prepared ahead, the
        plan of the following batch runs on the second stream beside its phases."""
        buf = (C.c_float * 8)()
        self.step(uid, iid, phase_ms=buf, next_batch=next_batch)
        names = ["sort_items", "sort_users", "fused_fwd_bwd", "loss_mean", "item_update",
                 "user_update", "total", "segment_heads"]
        return {n: float(buf[i]) for i, n in enumerate(names)}


# ---- NeuMF head ----------------------------------------------------------------------------------

def neumf_supported(d, l1):
    return bool(_lib.load().rc_neumf_supported(int(d), int(l1)))


def _neumf_ptrs(P):
    f32 = torch.float32
    return [_ptr(P[k], f32, k) for k in ("mf_u", "mf_i", "mlp_u", "mlp_i", "W1", "b1", "w_out")]


def _drop_args(drop_p, seed):
    """(p, device pointer to the 64-bit mask seed); seed = int64 tensor [1] on the device, read by the kernels"""
    if not drop_p:
        return C.c_float(0.0), C.c_void_p(0)
    if seed is None:
        raise ValueError("dropout needs a device seed tensor (int64 [1])")
    return C.c_float(float(drop_p)), _ptr(seed, torch.int64, "seed")


_deferred_inc = None      # [counter, folded?]: an increment somebody has promised to make before its next reader runs


def defer_increment(counter):
    """The caller owes `counter[0] += 1` some time before its own later kernel reads the counter, and nothing in between reads it:
    the next step_increment of ANOTHER counter takes it along in the same launch (rc_step_increment2: in a replayed graph a launch
    costs ~4.6 us whatever it does; a DeepFM step at B = 1,024 bumps the tower's dropout seed and Adam's step count).  The caller
    settles with take_deferred before that later kernel."""
    global _deferred_inc
    _deferred_inc = [counter, False]


def take_deferred(counter):
    """settle a defer_increment: True = some step_increment folded it in (the counter is incremented), False = nobody did (the caller
    increments itself).  Either way the promise is gone."""
    global _deferred_inc
    d, _deferred_inc = _deferred_inc, None
    return d is not None and d[0] is counter and d[1]


_bumped_early = []        # counters a carrier launch has already incremented for their owner's NEXT step_increment call


def clear_bumped_early():
    """forget early increments nobody collected (a step that raised between the carrier launch and the owner's step_increment)"""
    del _bumped_early[:]


def pending_deferred():
    """the counter of an unsettled defer_increment that no launch has taken along yet, or None -- a launch with a slot for a counter
    it does not read may take it (fold_deferred after the launch is enqueued)"""
    d = _deferred_inc
    return d[0] if (d is not None and not d[1]) else None


def fold_deferred(counter):
    d = _deferred_inc
    if d is not None and d[0] is counter:
        d[1] = True


def step_increment(counter):
    """counter[0] += 1 on the device (rc_step_increment): Adam's step, the dropout seed; capturable"""
    for k, c in enumerate(_bumped_early):
        if c is counter:      # an earlier launch of this step (the field gather) incremented it on the owner's behalf
            del _bumped_early[k]
            return
    d = _deferred_inc
    if d is not None and not d[1] and d[0] is not counter and d[0].device == counter.device:
        _lib.call("rc_step_increment2", _ptr(counter, torch.int64, "counter"), _ptr(d[0], torch.int64, "deferred counter"), _stream())
        d[1] = True
        return
    _lib.call("rc_step_increment", _ptr(counter, torch.int64, "counter"), _stream())


def neumf_fwd(P, uid, iid, drop_p=0.0, seed=None):
    """P: dict mf_u, mf_i, mlp_u, mlp_i [rows,d], W1 [l1,2d], b1 [l1], w_out [d+l1] -> pred [B,C]
    (models/general/NeuMF.py:61-75).  drop_p > 0: training-mode dropout on the hidden layer, mask drawn from
    the counter-based stream keyed by seed[0] (rc_neumf_fwd_dropout)."""
    B, Cn = iid.shape
    d, l1 = P["mf_u"].shape[1], P["W1"].shape[0]
    pred = torch.empty((B, Cn), dtype=torch.float32, device=iid.device)
    _lib.call("rc_neumf_fwd_dropout", *_neumf_ptrs(P), _ptr(uid, torch.int64, "uid"), _ptr(iid, torch.int64, "iid"),
              B, Cn, d, l1, *_drop_args(drop_p, seed), _ptr(pred, torch.float32, "pred"), _stream())
    return pred


def neumf_bwd(P, uid, iid, gpred, drop_p=0.0, seed=None):
    """-> (per-occurrence row grads dict g_mf_u, g_mf_i, g_mlp_u, g_mlp_i [B*C,d], dense grads dict W1, b1, w_out);
    drop_p / seed as given to the forward this is the backward of (the mask is regenerated, not stored)"""
    B, Cn = iid.shape
    d, l1 = P["mf_u"].shape[1], P["W1"].shape[0]
    dev, f32 = iid.device, torch.float32
    rows = {k: torch.empty((B * Cn, d), dtype=f32, device=dev) for k in ("g_mf_u", "g_mf_i", "g_mlp_u", "g_mlp_i")}
    dense = {"W1": torch.empty_like(P["W1"]), "b1": torch.empty_like(P["b1"]), "w_out": torch.empty_like(P["w_out"])}
    ws = workspace(_lib.load().rc_neumf_workspace_bytes(B, Cn, d, l1), dev, "neumf")
    _lib.call("rc_neumf_bwd_dropout", *_neumf_ptrs(P), _ptr(uid, torch.int64, "uid"), _ptr(iid, torch.int64, "iid"),
              _ptr(gpred, f32, "gpred"), B, Cn, d, l1, *_drop_args(drop_p, seed),
              *[_ptr(rows[k], f32, k) for k in ("g_mf_u", "g_mf_i", "g_mlp_u", "g_mlp_i")],
              _ptr(dense["W1"], f32, "dW1"), _ptr(dense["b1"], f32, "db1"), _ptr(dense["w_out"], f32, "dw_out"),
              C.c_void_p(ws.data_ptr()), ws.numel(), _stream())
    return rows, dense


def neumf_train_step_supported(Cn, d, l1):
    return bool(_lib.load().rc_neumf_train_step_supported(int(Cn), int(d), int(l1)))


def neumf_fused_step_selected(Cn, d, l1, opt="SGD"):
    """would NeumfTrainer.step take the one-kernel step (rc_neumf_train_step) for Cn candidates per tuple?  Everything its
    selection tests except the batch's plan geometry (known only with the batch)"""
    return bool(_USE_PLAN and _NEUMF_FUSED and opt in ("SGD", "Adam", "Adagrad") and Cn >= 2 and segmented_pair_supported(d)
                and neumf_train_step_supported(Cn, d, l1))


def neumf_mark_rows(iid, n_items, marks, unmark=False):
    """rc_neumf_mark_rows / rc_neumf_unmark_rows on torch's current stream: the single / multi-occurrence flags of a batch's item ids"""
    _lib.call("rc_neumf_unmark_rows" if unmark else "rc_neumf_mark_rows", _ptr(iid, torch.int64, "iid"), iid.numel(), int(n_items),
              C.c_void_p(marks.data_ptr()), _stream())


def neumf_train_step(P, state, uid, iid, hyper, marks, out, inv_b=None, pred=None, marked=False, drop_p=0.0, seed=None):
    """rc_neumf_train_step: forward + BPR loss + backward + in-place update of single-occurrence item rows.
    state: {table: {"m": .., "v": ..}} of the optimizer; marks: uint8 buffer of rc_neumf_train_step_marks_bytes(n_items), zeroed
    once (every call leaves it ready for the next; marked=True: prepared by neumf_mark_rows for this very batch, cleared by the caller); out: dict of preallocated buffers loss_vec [B], g_mf_i / g_mlp_i [B C, d], gu_mf / gu_mlp [B, d], W1 / b1 / w_out
    gradients.  drop_p > 0: training-mode dropout on the hidden layer inside the kernel (rc_neumf_train_step_dropout; `seed` int64 [1]
    on the device, the mask stream of neumf_fwd / neumf_bwd).  Returns nothing: the caller finishes the step with the plan's pair
    updates and the dense update."""
    B, Cn = iid.shape
    d, l1 = P["mf_u"].shape[1], P["W1"].shape[0]
    f32 = torch.float32
    lib = _lib.load()
    ws = workspace(lib.rc_neumf_train_step_workspace_bytes(B, Cn, d, l1), iid.device, "neumf_step")
    st = lambda t, k: _ptr(state[t].get(k), f32, k + "_" + t, True)
    head = [_ptr(P["mf_u"], f32, "mf_u"), _ptr(P["mf_i"], f32, "mf_i"), _ptr(P["mlp_u"], f32, "mlp_u"),
            _ptr(P["mlp_i"], f32, "mlp_i"), st("mf_i", "m"), st("mf_i", "v"), st("mlp_i", "m"), st("mlp_i", "v"),
            _ptr(P["W1"], f32, "W1"), _ptr(P["b1"], f32, "b1"), _ptr(P["w_out"], f32, "w_out"),
            _ptr(uid, torch.int64, "uid"), _ptr(iid, torch.int64, "iid"), B, Cn, d, l1, int(P["mf_i"].shape[0]),
            C.c_void_p(marks.data_ptr())]
    tail = [_ptr(out["loss_vec"], f32, "loss_vec"), _ptr(pred, f32, "pred", True),
            _ptr(out["g_mf_i"], f32, "g_mf_i"), _ptr(out["g_mlp_i"], f32, "g_mlp_i"), _ptr(out["gu_mf"], f32, "gu_mf"),
            _ptr(out["gu_mlp"], f32, "gu_mlp"), _ptr(out["W1"], f32, "dW1"), _ptr(out["b1"], f32, "db1"),
            _ptr(out["w_out"], f32, "dw_out"), C.c_void_p(ws.data_ptr()), ws.numel(), _stream()]
    inv = float(1.0 / B if inv_b is None else inv_b)
    if drop_p:
        _lib.call("rc_neumf_train_step_dropout", *head, 1 if marked else 0, C.byref(hyper), inv, *_drop_args(drop_p, seed), *tail)
    else:
        _lib.call("rc_neumf_train_step_marked" if marked else "rc_neumf_train_step", *head, C.byref(hyper), inv, *tail)


def neumf_head_fwd_bwd(urows, irows, W1, b1, w_out, B, Cn, inv_b, want_pred=False):
    """rc_neumf_head_fwd_bwd on fetched row blocks: urows [B, 2d] = [mf_u | mlp_u] of every tuple's user, irows [B C, 2d] =
    [mf_i | mlp_i] of every candidate (the layout the sharded step's exchanges deliver).  -> (loss_vec [B], gu [B, 2d],
    gi [B C, 2d], {W1, b1, w_out gradients}, pred | None): the gradient blocks in the same [mf | mlp] layout, ready to travel back."""
    d2 = urows.shape[1]
    d, l1 = d2 // 2, W1.shape[0]
    dev, f32 = urows.device, torch.float32
    if irows.shape != (B * Cn, d2) or urows.shape[0] != B:
        raise ValueError("neumf_head_fwd_bwd: row blocks do not match B, C")
    key = ("nhead", B, Cn, d2, str(dev))
    ids = _ws_cache.get(key)
    if ids is None:   # positional ids: tuple b uses user row b and item rows b C .. b C + C - 1
        ids = _ws_cache[key] = (torch.arange(B, device=dev), torch.arange(B * Cn, device=dev).view(B, Cn))
    loss_vec = torch.empty(B, dtype=f32, device=dev)
    gu = torch.empty((B, d2), dtype=f32, device=dev)
    gi = torch.empty((B * Cn, d2), dtype=f32, device=dev)
    pred = torch.empty((B, Cn), dtype=f32, device=dev) if want_pred else None
    dense = {"W1": torch.empty_like(W1), "b1": torch.empty_like(b1), "w_out": torch.empty_like(w_out)}
    ws = workspace(_lib.load().rc_neumf_train_step_workspace_bytes(B, Cn, d, l1), dev, "neumf_step")
    _ptr(urows, f32, "urows"); _ptr(irows, f32, "irows")
    off = lambda t, k: C.c_void_p(t.data_ptr() + 4 * k)
    _lib.call("rc_neumf_head_fwd_bwd", off(urows, 0), off(urows, d), d2, off(irows, 0), off(irows, d), d2,
              _ptr(W1, f32, "W1"), _ptr(b1, f32, "b1"), _ptr(w_out, f32, "w_out"), _ptr(ids[0], torch.int64, "uid"),
              _ptr(ids[1], torch.int64, "iid"), B, Cn, d, l1, float(inv_b), _ptr(loss_vec, f32, "loss_vec"),
              _ptr(pred, f32, "pred", True), off(gi, 0), off(gi, d), d2, off(gu, 0), off(gu, d), d2,
              _ptr(dense["W1"], f32, "dW1"), _ptr(dense["b1"], f32, "db1"), _ptr(dense["w_out"], f32, "dw_out"),
              C.c_void_p(ws.data_ptr()), ws.numel(), _stream())
    return loss_vec, gu, gi, dense, pred


def neumf_zhead_supported(Cn, d, l1):
    return bool(_lib.load().rc_neumf_zhead_supported(int(Cn), int(d), int(l1)))


def neumf_zhead(urows, irows, W1, b1, w_out, B, Cn, inv_b, want_pred=False):
    """the head of a sharded NeuMF step whose item half of the hidden layer came from the rows' owners (csrc/neumf_zhead.hip):
    urows [B, 2d] = [mf_u | mlp_u], irows [B C, d + l1] = [mf_i | zi = W1i mlp_i].  zu = W1u mlp_u + b1 and the backward of the
    user half are GEMMs (rc_linear_fwd / rc_linear_bwd), everything per candidate is rc_neumf_zhead_fwd_bwd.
    -> (loss_vec [B], gu [B, 2d] = [d mf_u | d mlp_u], gi [B C, d + l1] = [d mf_i | dz], {W1u [l1, d], b1, w_out gradients}, pred | None)"""
    d = urows.shape[1] // 2
    l1 = W1.shape[0]
    dev, f32 = urows.device, torch.float32
    if irows.shape != (B * Cn, d + l1) or urows.shape != (B, 2 * d):
        raise ValueError("neumf_zhead: row blocks do not match B, C, d, hidden")
    W1u = W1[:, :d].contiguous()
    mlp_u = urows[:, d:].contiguous()
    zu = linear_fwd(mlp_u, W1u, b1)
    loss_vec = torch.empty(B, dtype=f32, device=dev)
    gu = torch.empty((B, 2 * d), dtype=f32, device=dev)
    gi = torch.empty((B * Cn, d + l1), dtype=f32, device=dev)
    dzu = torch.empty((B, l1), dtype=f32, device=dev)
    dw_out = torch.empty(d + l1, dtype=f32, device=dev)
    pred = torch.empty((B, Cn), dtype=f32, device=dev) if want_pred else None
    ws = workspace(_lib.load().rc_neumf_zhead_workspace_bytes(d, l1), dev, "neumf_zhead")
    _lib.call("rc_neumf_zhead_fwd_bwd", _ptr(urows, f32, "urows"), 2 * d, _ptr(zu, f32, "zu"), _ptr(irows, f32, "irows"), d + l1,
              _ptr(w_out, f32, "w_out"), B, Cn, d, l1, float(inv_b), _ptr(loss_vec, f32, "loss_vec"), _ptr(pred, f32, "pred", True),
              _ptr(gi, f32, "gi"), d + l1, _ptr(gu, f32, "gu"), 2 * d, _ptr(dzu, f32, "dzu"), _ptr(dw_out, f32, "dw_out"),
              C.c_void_p(ws.data_ptr()), ws.numel(), _stream())
    dmlp_u, dW1u, db1 = linear_bwd(mlp_u, W1u, None, dzu, ws_tag="zhead_bwd")
    gu[:, d:] = dmlp_u
    return loss_vec, gu, gi, {"W1u": dW1u, "b1": db1, "w_out": dw_out}, pred


class _PhaseTimer:
    """Optional per-phase timing of a trainer step with events on the launch stream (torch's current stream is the
    stream every kernel of the step is enqueued on).  trainer.timing = {} switches it on; read with phases_ms()."""

    def __init__(self, owner, name):
        self.t = getattr(owner, "timing", None)
        self.name = name

    def __enter__(self):
        if self.t is not None:
            self.a = torch.cuda.Event(enable_timing=True)
            self.a.record()

    def __exit__(self, *exc):
        if self.t is not None:
            b = torch.cuda.Event(enable_timing=True)
            b.record()
            self.t.setdefault(self.name, []).append((self.a, b))


def phases_ms(trainer):
    """MEDIAN milliseconds per phase of the steps recorded since trainer.timing = {} (synchronises).  (The mean until round 6: one
    eager step of twenty that meets an allocation on a side stream -- 68 ms once -- made a 0.05 ms phase read 3.4 ms.)"""
    torch.cuda.synchronize()
    out = {}
    for k, v in (trainer.timing or {}).items():
        t = sorted(a.elapsed_time(b) for a, b in v)
        out[k] = t[len(t) // 2] if len(t) % 2 else 0.5 * (t[len(t) // 2 - 1] + t[len(t) // 2])
    return out


class NeumfTrainer:
    """One BaseRunner.fit iteration for NeuMF (single hidden layer) on device tensors:
    forward (MFMA) -> BPR loss -> backward (MFMA) -> row-wise segmented update of the four tables
    -> dense optimizer step of W1, b1, w_out.  P as in neumf_fwd; updated in place.
    dropout > 0: hidden-layer dropout with a fresh mask per step (device seed counter, bumped every step)."""

    def __init__(self, P, opt="Adam", lr=1e-3, l2=0.0, rowwise=True, dropout=0.0, seed=0):
        self.P, self.opt, self.lr, self.l2, self.rowwise = P, opt, lr, l2, rowwise
        self.dropout = float(dropout)
        self.seed = torch.tensor([seed], dtype=torch.int64, device=P["W1"].device) if self.dropout > 0 else None
        self.state = {}
        for k, t in P.items():
            st = {}
            if opt in ("Adam", "Adagrad"):
                st["m"] = torch.zeros_like(t)
            if opt == "Adam":
                st["v"] = torch.zeros_like(t)
            self.state[k] = st
        self.step_count = 0
        self.loss = None
        self._side = None
        NeumfTrainer._n_trainers += 1
        self._serial = NeumfTrainer._n_trainers     # plan workspaces are cached by tag: one set per trainer

    _n_trainers = 0

    def __del__(self):
        try:
            # the trainer's plan workspaces leave the cache -- once the side streams that wrote them have drained (a block freed
            # with side-stream work in flight could be handed out again on the main stream)
            for st in (getattr(self, "_side", None), getattr(self, "_side2", None)):
                if st is not None:
                    st.synchronize()
            for key in [k for k in _ws_cache if isinstance(k[1], str) and k[1].startswith("neumf%d." % self._serial)]:
                _ws_cache.pop(key, None)
        except Exception:
            pass

    def step(self, uid, iid, next_batch=None):
        """next_batch = (uid, iid) of the FOLLOWING call (the very tensors it will bring, unmodified until then): the fused step
        builds their bucket plan beside this step's table updates, off the critical path."""
        P = self.P
        B, Cn = iid.shape
        self.step_count += 1
        if self.seed is not None:
            step_increment(self.seed)
        pair_ok = segmented_pair_supported(P["mf_u"].shape[1])
        n_u, n_i = P["mf_u"].shape[0], P["mf_i"].shape[0]
        use_plan = self.rowwise and pair_ok and _USE_PLAN and plan_supported(iid.numel(), uid.numel(), n_i, n_u)
        # the bucket plan needs only the ids: on a second stream it runs beside the head kernels (large batches; a small step is
        # bound by the host's launch rate and the stream switches cost more than they return)
        if (use_plan and _NEUMF_FUSED and self.opt in ("SGD", "Adam", "Adagrad") and Cn >= 2
                and neumf_train_step_supported(Cn, P["mf_u"].shape[1], P["W1"].shape[0])):
            return self._step_fused(uid, iid, next_batch)
        if not neumf_supported(P["mf_u"].shape[1], P["W1"].shape[0]):
            # (a tower that exists only inside the one-kernel step, e.g. hidden 16: say why this call cannot take it instead of
            # failing with RC_ERR_UNSUPPORTED inside rc_neumf_fwd)
            raise RuntimeError("NeumfTrainer: emb_size {} / hidden {} is a shape of the one-kernel step (rc_neumf_train_step) only, which this call "
                               "cannot take: {} candidates per tuple (needs >= 2 and the kernel's LDS budget: rc_neumf_train_step_supported), "
                               "plan {} (RC_TABLE_UPDATE, row-wise updates, a plan geometry for {} + {} ids), RC_NEUMF_FUSED={} -- use --engine "
                               "dense for this configuration".format(P["mf_u"].shape[1], P["W1"].shape[0], Cn, "on" if use_plan else "off",
                                                                      iid.numel(), uid.numel(), int(_NEUMF_FUSED)))
        overlap = use_plan and _NEUMF_OVERLAP and iid.is_cuda and iid.numel() >= _SAS_OVERLAP_MIN
        plan = plan_done = main = None
        if overlap:
            if self._side is None:
                self._side = torch.cuda.Stream(device=iid.device)
            main, side = torch.cuda.current_stream(iid.device), self._side
            side.wait_stream(main)   # the batch is ready; last step's readers of the plan buffers are done
            with torch.cuda.stream(side):
                plan = Plan(iid, n_i, uid, n_u, tag="neumf")
                plan_done = side.record_event()
        with _PhaseTimer(self, "head_fwd"):
            pred = neumf_fwd(P, uid, iid, self.dropout, self.seed)
        with _PhaseTimer(self, "loss"):
            self.loss, _, gpred = bpr_loss(pred)
        with _PhaseTimer(self, "head_bwd"):
            rows, dense = neumf_bwd(P, uid, iid, gpred, self.dropout, self.seed)
        h = make_hyper(self.opt, lr=self.lr, l2=self.l2, step=self.step_count)
        h0 = make_hyper(self.opt, lr=self.lr, l2=0.0, step=self.step_count)  # 'bias' params: no weight decay
        if use_plan:
            # ONE bucket plan of both id lists (round 3: hashed buckets where the id space is wide and sparse -- 0.33 M item
            # lookups over 10 M - 100 M rows -- so the cost follows the keys, not the id range; round 2's id-range
            # buckets cost as much as the radix sort here and the sort stayed), then one pair update per side: the
            # mf / mlp tables of a side share ids, records and positions.  The user side is planned per TUPLE: the head
            # kernel's per-candidate user gradients are summed over a tuple's candidates first (fixed order c = 0..C-1), so
            # a hot user contributes B_u occurrences, not C * B_u.
            with _PhaseTimer(self, "sort"):
                if overlap:
                    main.wait_event(plan_done)
                else:
                    plan = Plan(iid, n_i, uid, n_u, tag="neumf")
            with _PhaseTimer(self, "table_update"):
                key = (B, Cn, str(uid.device))
                if getattr(self, "_sum_idx", (None,))[0] != key:
                    self._sum_idx = (key, torch.arange(B * Cn, device=uid.device).view(B, Cn), torch.ones((B, Cn), device=uid.device))
                _, pos, ones = self._sum_idx
                gu_a, gu_b = weighted_row_sum(rows["g_mf_u"], pos, ones), weighted_row_sum(rows["g_mlp_u"], pos, ones)
                for side, ta, tb, ga, gb in (("b", "mf_u", "mlp_u", gu_a, gu_b), ("a", "mf_i", "mlp_i", rows["g_mf_i"], rows["g_mlp_i"])):
                    sa, sb = self.state[ta], self.state[tb]
                    plan.update_pair(side, P[ta], P[tb], ga, gb, h, ma=sa.get("m"), va=sa.get("v"),
                                     mb=sb.get("m"), vb=sb.get("v"))
        else:
            self._step_tables_sorted(P, uid.repeat_interleave(Cn), iid, rows, h, pair_ok)
        with _PhaseTimer(self, "dense_update"):
            dense_update_multi([(P[k], dense[k], h0 if k == "b1" else h, self.state[k].get("m"), self.state[k].get("v"))
                                for k in ("W1", "b1", "w_out")], self.opt)
        return self.loss

    @staticmethod
    def _batch_key(uid, iid):
        """identity of a batch's id tensors (storage, shape and torch's in-place version counters): a plan prepared for them is
        used only if the following call brings exactly these tensors, untouched"""
        return (uid.data_ptr(), iid.data_ptr(), tuple(uid.shape), tuple(iid.shape), uid._version, iid._version)

    def _step_fused(self, uid, iid, next_batch=None):
        """The step on rc_neumf_train_step (csrc/neumf_step.hip): ONE kernel for forward, loss, backward and the in-place update
        of single-occurrence item rows, then two pair updates from the bucket plan of the batch (item side: multi-occurrence rows
        only; user side: one gradient row per tuple) on two streams.  The fused kernel leaves no register for a co-resident wave,
        so a plan built beside it would wait for its workgroups to retire: the plan of the NEXT batch is built beside this
        step's table updates instead (next_batch), the batch's own plan only when nobody announced it."""
        P = self.P
        B, Cn = iid.shape
        dev = iid.device
        d = P["mf_u"].shape[1]
        n_u, n_i = P["mf_u"].shape[0], P["mf_i"].shape[0]
        if getattr(self, "_marks", None) is None:   # two buffers: this batch's flags and, prepared beside this step's updates, the next batch's
            nbytes = max(int(_lib.load().rc_neumf_train_step_marks_bytes(n_i)), 1)
            self._marks = [torch.zeros(nbytes, dtype=torch.uint8, device=dev) for _ in range(2)]
        key = (B, Cn, str(dev))
        if getattr(self, "_fused_out", (None,))[0] != key:
            e = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)
            self._fused_out = (key, {"loss_vec": e(B), "g_mf_i": e(B * Cn, d), "g_mlp_i": e(B * Cn, d), "gu_mf": e(B, d), "gu_mlp": e(B, d),
                                     "W1": torch.empty_like(P["W1"]), "b1": torch.empty_like(P["b1"]), "w_out": torch.empty_like(P["w_out"])})
        out = self._fused_out[1]
        two_streams = _NEUMF_OVERLAP and iid.numel() >= _SAS_OVERLAP_MIN
        main = torch.cuda.current_stream(dev)
        if two_streams and self._side is None:
            # RC_NEUMF_PLAN_PRIORITY=1: the plan's stream above the others (its small latency-bound kernels then go first where the
            # update kernels compete for the same CUs)
            prio = -1 if os.environ.get("RC_NEUMF_PLAN_PRIORITY", "0") == "1" else 0
            self._side, self._side2 = torch.cuda.Stream(device=dev, priority=prio), torch.cuda.Stream(device=dev)
        ahead = getattr(self, "_ahead", None)
        self._ahead = None
        plan = plan_done = marks_done = None
        # which of the two flag buffers / plan workspaces this step uses travels WITH the announcement (not with step_count:
        # a non-fused step in between, or a second trainer, must not shift it); with nothing pending both buffers are clean
        buf = 0
        if ahead is not None and ahead["key"] == self._batch_key(uid, iid):
            plan, plan_done, marks_done, buf = ahead["plan"], ahead["plan_done"], ahead["marks_done"], ahead["buf"]
        elif ahead is not None:
            # a batch was announced and another one came: its prepared flags have to go before that buffer is marked again
            with torch.cuda.stream(self._side):
                neumf_mark_rows(ahead["iid"], n_i, self._marks[ahead["buf"]], unmark=True)
            main.wait_stream(self._side)
        ahead = None    # (drops the reference to the announced id tensors kept for the side stream's reads)
        tag = lambda k: "neumf%d.%d" % (self._serial, k)
        if plan is None and two_streams:
            self._side.wait_stream(main)
            with torch.cuda.stream(self._side):
                plan = Plan(iid, n_i, uid, n_u, tag=tag(buf), list_single_a=False)
                plan.prezero_update_counters()
                plan_done = self._side.record_event()
            iid.record_stream(self._side)
            uid.record_stream(self._side)
        h = make_hyper(self.opt, lr=self.lr, l2=self.l2, step=self.step_count)
        h0 = make_hyper(self.opt, lr=self.lr, l2=0.0, step=self.step_count)
        with _PhaseTimer(self, "fused_step"):
            if marks_done is not None:      # flags prepared beside the previous step's updates
                main.wait_event(marks_done)
                neumf_train_step(P, self.state, uid, iid, h, self._marks[buf], out, marked=True, drop_p=self.dropout, seed=self.seed)
            else:
                neumf_train_step(P, self.state, uid, iid, h, self._marks[buf], out, drop_p=self.dropout, seed=self.seed)
        if marks_done is not None and not two_streams:
            # prepared flags are ALWAYS cleared by the step that consumed them -- also a short step (a ragged last batch of an
            # epoch announced by a large one) that runs on one stream: a flag left behind would make a later single occurrence
            # of that row look like a multiple one, and its update would be dropped
            neumf_mark_rows(iid, n_i, self._marks[buf], unmark=True)
        loss_done = None
        if two_streams:
            self._side.wait_stream(main)    # behind the fused kernel: beside the updates below
            with torch.cuda.stream(self._side):
                # the batch mean of the per-tuple losses (one workgroup) FIRST on the plan's stream: nothing of this step waits for it
                # (main joins it at the very end), and the streams that carry the two table updates start those at once.  (Round 5
                # had it in front of the user-side update: 28 us on the critical path of the step.)
                self.loss = reduce_sum(out["loss_vec"], 1.0 / B)
                loss_done = self._side.record_event()
        if two_streams and (next_batch is not None or marks_done is not None):
            with torch.cuda.stream(self._side):
                if marks_done is not None:
                    neumf_mark_rows(iid, n_i, self._marks[buf], unmark=True)
                    iid.record_stream(self._side)   # the runner drops the batch when step() returns; the allocator must not
                                                    # hand its block out while the side stream still reads it
                if next_batch is not None:
                    # the following batch's flags and plan; their buffers alternate with this batch's
                    nu, ni = next_batch
                    neumf_mark_rows(ni, n_i, self._marks[buf ^ 1])
                    nmarks_done = self._side.record_event()
                    nplan = Plan(ni, n_i, nu, n_u, tag=tag(buf ^ 1), list_single_a=False)
                    nplan.prezero_update_counters()
                    ni.record_stream(self._side)
                    nu.record_stream(self._side)
                    self._ahead = {"key": self._batch_key(nu, ni), "plan": nplan, "plan_done": self._side.record_event(),
                                   "marks_done": nmarks_done, "iid": ni, "buf": buf ^ 1}
        if two_streams:
            self._side2.wait_stream(main)
        else:
            with _PhaseTimer(self, "loss"):
                self.loss = reduce_sum(out["loss_vec"], 1.0 / B)
        with _PhaseTimer(self, "sort"):
            if plan is None:
                plan = Plan(iid, n_i, uid, n_u, tag=tag(buf), list_single_a=False)
            elif plan_done is not None:
                main.wait_event(plan_done)
        with _PhaseTimer(self, "table_update"):
            sides = (("a", "mf_i", "mlp_i", out["g_mf_i"], out["g_mlp_i"]), ("b", "mf_u", "mlp_u", out["gu_mf"], out["gu_mlp"]))

            def upd(side_, ta, tb, ga, gb):
                sa, sb = self.state[ta], self.state[tb]
                plan.update_pair(side_, P[ta], P[tb], ga, gb, h, ma=sa.get("m"), va=sa.get("v"), mb=sb.get("m"), vb=sb.get("v"), ws_tag=side_)
            if two_streams:   # item tables and user tables are disjoint: the two updates run side by side
                if plan_done is not None:
                    self._side2.wait_event(plan_done)
                with torch.cuda.stream(self._side2):
                    upd(*sides[1])
                upd(*sides[0])
                main.wait_stream(self._side2)
            else:
                upd(*sides[0])
                upd(*sides[1])
        with _PhaseTimer(self, "dense_update"):
            dense_update_multi([(P[k], out[k], h0 if k == "b1" else h, self.state[k].get("m"), self.state[k].get("v"))
                                for k in ("W1", "b1", "w_out")], self.opt)
        if loss_done is not None:
            main.wait_event(loss_done)      # (long complete: the caller reads the loss on this stream)
        return self.loss

    def _step_tables_sorted(self, P, uid_occ, iid, rows, h, pair_ok):
        """the table updates behind a radix sort (dense-gradient mode = the reference's exact optimizer semantics, widths
        without a pair kernel, id spaces no plan geometry covers)"""
        with _PhaseTimer(self, "sort"):
            ku, pu = sort_ids(uid_occ, P["mf_u"].shape[0])
            ki, pi = sort_ids(iid, P["mf_i"].shape[0])
            # the mf / mlp tables of a side share ids: one sort, ONE head list and ONE update pass serve both
            _, hu, nhu = segment_heads(ku, pu, want_single=False)
            _, hi, nhi = segment_heads(ki, pi, want_single=False)
        _upd = _PhaseTimer(self, "table_update")
        _upd.__enter__()
        for ta, tb, ga, gb, keys, perm, hd, nh in (("mf_u", "mlp_u", "g_mf_u", "g_mlp_u", ku, pu, hu, nhu),
                                                   ("mf_i", "mlp_i", "g_mf_i", "g_mlp_i", ki, pi, hi, nhi)):
            sa, sb = self.state[ta], self.state[tb]
            G = (torch.zeros_like(P[ta]), torch.zeros_like(P[tb])) if not self.rowwise else (None, None)
            if pair_ok and self.rowwise:
                segmented_update_pair(keys, perm, rows[ga], rows[gb], hyper=h, W=(P[ta], P[tb]), m=(sa.get("m"), sb.get("m")),
                                      v=(sa.get("v"), sb.get("v")), heads=hd, n_heads=nh)
            elif pair_ok:
                segmented_update_pair(keys, perm, rows[ga], rows[gb], dense_grad=G, heads=hd, n_heads=nh)
            else:
                for k, (tab, grad) in enumerate(((ta, ga), (tb, gb))):
                    st = self.state[tab]
                    if self.rowwise:
                        segmented_update(keys, perm, rows[grad], hyper=h, W=P[tab], m=st.get("m"), v=st.get("v"), heads=hd,
                                         n_heads=nh)
                    else:
                        segmented_update(keys, perm, rows[grad], dense_grad=G[k], heads=hd, n_heads=nh)
            if not self.rowwise:
                dense_update(P[ta], G[0], h, sa.get("m"), sa.get("v"))
                dense_update(P[tb], G[1], h, sb.get("m"), sb.get("v"))
        _upd.__exit__()


# ---- dense layers (csrc/mlp.hip) -------------------------------------------------------------------------

def linear_fwd(X, W, b=None, relu=False, drop_p=0.0, seed=None, site=0):
    """Y = drop(relu(X W^T + b)) on the fp32 MFMA GEMM (rc_linear_fwd); X [M, K], W [N, K] (nn.Linear.weight), b [N] | None.
    drop_p > 0: training-mode dropout keyed by seed[0] (int64 [1] on the device) and the layer index `site`."""
    f32 = torch.float32
    M, K = X.shape
    N = W.shape[0]
    Y = torch.empty((M, N), dtype=f32, device=X.device)
    ws = workspace(_lib.load().rc_linear_fwd_workspace_bytes(M, N, K), X.device, "linear_fwd")   # split-K planes of a small batch
    _lib.call("rc_linear_fwd_ws", _ptr(X, f32, "X"), _ptr(W, f32, "W"), _ptr(b, f32, "b", True), M, N, K, 1 if relu else 0,
              *_drop_args(drop_p, seed), C.c_uint32(int(site)), _ptr(Y, f32, "Y"), C.c_void_p(ws.data_ptr()), ws.numel(), _stream())
    return Y


def linear_bwd(X, W, Y, dY, drop_p=0.0, need_dx=True, need_db=True, x_act=False, x_drop_p=0.0, need_dw=True, ws_tag="linear_bwd"):
    """backward of linear_fwd -> (dX | None, dW, db | None).  Y: the layer's saved output when it went through
    relu (+ dropout) -- it is its own mask -- or None for a plain Linear (or when dY is already masked, see below).
    x_act: X is the drop(relu(.)) output of the layer below (dropout x_drop_p): dX comes out multiplied by that layer's mask
    in the product's epilogue (rc_linear_bwd_chain), and the layer below is called with Y=None.
    need_dw=False: only dX (the weight gradient comes from a second call, e.g. on another stream with its own ws_tag)."""
    f32 = torch.float32
    M, K = X.shape
    N = W.shape[0]
    dev = X.device
    dX = torch.empty((M, K), dtype=f32, device=dev) if need_dx else None
    dW = torch.empty((N, K), dtype=f32, device=dev) if need_dw else None
    db = torch.empty(N, dtype=f32, device=dev) if (need_db and need_dw) else None
    ws = workspace(_lib.load().rc_linear_bwd_workspace_bytes(M, N, K), dev, ws_tag)
    _lib.call("rc_linear_bwd_chain", _ptr(X, f32, "X"), _ptr(W, f32, "W"), _ptr(Y, f32, "Y", True), _ptr(dY, f32, "dY"), M, N, K,
              C.c_float(float(drop_p)), 1 if (x_act and need_dx) else 0, C.c_float(float(x_drop_p)), _ptr(dX, f32, "dX", True),
              _ptr(dW, f32, "dW", True), _ptr(db, f32, "db", True), C.c_void_p(ws.data_ptr()), ws.numel(), _stream())
    return dX, dW, db


_TOWER_TAIL = os.environ.get("RC_TOWER_TAIL", "1") != "0"      # A/B switch: the tail of a tower as two kernels (csrc/tower_tail.hip)
_TOWER_TAIL_MAX_M = int(os.environ.get("RC_TOWER_TAIL_MAX_M", "8192"))    # larger batches fill the GEMM tiles (B = 16,384: 0.875 ms with the tail kernels, 0.839 without)


def tower_tail_supported(M, K, N2):
    """the last hidden layer [K -> N2] + the output layer [N2 -> 1] of a tower run as rc_tower_tail_fwd / _bwd"""
    return bool(_TOWER_TAIL and 1 <= M <= _TOWER_TAIL_MAX_M and _lib.load().rc_tower_tail_supported(int(M), int(K), int(N2)))


def tower_tail_fwd(X, W2, b2, w3, b3, drop_p=0.0, seed=None, site=0):
    """-> (H2 [M, N2] = drop(relu(X W2^T + b2)), z [M, 1] = H2 w3^T + b3); w3 [1, N2] (nn.Linear(N2, 1).weight), b3 [1] | None;
    dropout mask as linear_fwd with layer index `site`"""
    f32 = torch.float32
    M, K = X.shape
    N2 = W2.shape[0]
    H2 = torch.empty((M, N2), dtype=f32, device=X.device)
    z = torch.empty((M, 1), dtype=f32, device=X.device)
    _lib.call("rc_tower_tail_fwd", _ptr(X, f32, "X"), _ptr(W2, f32, "W2"), _ptr(b2, f32, "b2", True), _ptr(w3, f32, "w3"), _ptr(b3, f32, "b3", True),
              M, K, N2, *_drop_args(drop_p, seed), C.c_uint32(int(site)), _ptr(H2, f32, "H2"), _ptr(z, f32, "z"), _stream())
    return H2, z


def tower_tail_bwd(X, W2, w3, H2, dz, drop_p=0.0, need_dx=True, x_act=False, x_drop_p=0.0, need_db2=True, need_db3=True):
    """backward of tower_tail_fwd given dz [M, 1] -> (dX | None, dW2, db2 | None, dW3 [1, N2], db3 [1] | None); x_act: X is the
    drop(relu(.)) output of the layer below and dX comes out multiplied by its mask (as linear_bwd)"""
    f32 = torch.float32
    M, K = X.shape
    N2 = W2.shape[0]
    dev = X.device
    dX = torch.empty((M, K), dtype=f32, device=dev) if need_dx else None
    dW2 = torch.empty((N2, K), dtype=f32, device=dev)
    db2 = torch.empty(N2, dtype=f32, device=dev) if need_db2 else None
    dW3 = torch.empty((1, N2), dtype=f32, device=dev)
    db3 = torch.empty(1, dtype=f32, device=dev) if need_db3 else None
    ws = workspace(_lib.load().rc_tower_tail_workspace_bytes(M, K, N2), dev, "tower_tail")
    _lib.call("rc_tower_tail_bwd", _ptr(X, f32, "X"), _ptr(W2, f32, "W2"), _ptr(w3, f32, "w3"), _ptr(H2, f32, "H2"), _ptr(dz, f32, "dz"), M, K, N2,
              C.c_float(float(drop_p)), 1 if (x_act and need_dx) else 0, C.c_float(float(x_drop_p)), _ptr(dX, f32, "dX", True),
              _ptr(dW2, f32, "dW2"), _ptr(db2, f32, "db2", True), _ptr(dW3, f32, "dw3"), _ptr(db3, f32, "db3", True),
              C.c_void_p(ws.data_ptr()), ws.numel(), _stream())
    return dX, dW2, db2, dW3, db3


# ---- SASRec encoder ---------------------------------------------------------------------------------

SAS_LAYER_KEYS = ("Wq", "bq", "Wk", "bk", "Wv", "bv", "ln1w", "ln1b", "W1", "b1", "W2", "b2", "ln2w", "ln2b")
SAS_NO_DECAY = ("bq", "bk", "bv", "ln1b", "b1", "b2", "ln2b")  # names containing 'bias' (BaseModel.py:64-73)


SASREC_CORE_MAX_HIS = 64   # history_max every encoder path covers; up to 128: one block without dropout (the batch encoder's one-row path)


def sasrec_supported(d, n_layers, n_heads, L, dropout=0.0):
    """the fused encoder covers the shape (rc_sasrec_supported); training-mode dropout needs the all-rows kernels (history <= 64)"""
    if dropout > 0.0 and int(L) > SASREC_CORE_MAX_HIS:
        return False
    return bool(_lib.load().rc_sasrec_supported(int(d), int(n_layers), int(n_heads), int(L)))


def _sas_ptr_table(layers):
    tab = (C.c_void_p * (14 * len(layers)))()
    for l, lay in enumerate(layers):
        for k, name in enumerate(SAS_LAYER_KEYS):
            tab[14 * l + k] = _ptr(lay[name], torch.float32, f"layer{l}.{name}").value
    return tab


class SasSaved:
    """what one forward pass keeps for its backward: layer inputs only (`sequence` kernels, the backward recomputes
    the rest) or every intermediate activation over the compact row space (`batch` kernels)"""

    def __init__(self, impl, data, B, n_layers, L, d):
        self.impl, self.data, self.B, self.n_layers, self.L, self.d = impl, data, B, n_layers, L, d


SASREC_BATCH_MIN_ROWS = 4096  # B * history_max from which the batch-level kernels are used (launch-bound below)


def _sasrec_one_row_encoder(d, n_heads, n_layers, L):
    """the batch encoder's K / V-free last-row path (csrc/sas_last_row.hpp; sb_last_row_mode) serves the WHOLE encoder:
    one block, no dropout, head count 1 / 2 / 4, history_max 3 .. 128"""
    if d is None or int(os.environ.get("RC_SAS_LAST_ROW", "2")) < 2 or os.environ.get("RC_SAS_FUSED_BLOCK") == "0" \
            or os.environ.get("RC_SAS_ROWS16") == "0":
        return False
    return (n_layers == 1 and d in (32, 64) and n_heads in (1, 2, 4) and 3 <= L <= 128 and L >= n_heads + 1
            and (d // n_heads) % (d * d // 256) == 0)


def _sasrec_impl(B, L, impl, d=None, n_heads=None, n_layers=None):
    """auto: the batch-level kernels whenever their one-row path is the whole encoder (4 + 7 launches that touch each history
    row once: 0.18 against 0.28 ms per step at B = 256, history_max = 20, and ahead at every size measured, B = 128 .. 4096);
    otherwise the per-sequence kernels (3 launches per pass) while the whole batch is ONE round of resident workgroups in the
    32-row geometry (B <= 512, history_max <= 32), the batch-level kernels (~35 launches, 2-3x the throughput) beyond"""
    impl = impl or os.environ.get("RC_SASREC_IMPL", "auto")
    if impl == "auto":
        if _sasrec_one_row_encoder(d, n_heads, n_layers, L):
            return "batch"
        if B * L < SASREC_BATCH_MIN_ROWS or (B <= 512 and L <= 32):
            return "sequence"
        return "batch"
    if impl not in ("batch", "sequence"):
        raise ValueError("SASRec impl must be auto | batch | sequence, got {!r}".format(impl))
    return impl


def sasrec_fwd(item_emb, pos_emb, layers, n_heads, hist, lengths, save=False, impl=None, drop_p=0.0, seed=None):
    """-> (hv [B,d], SasSaved|None): encoder output at position length-1 (SASRec.py:58-76).
    impl: 'sequence' (csrc/sasrec.hip, one workgroup per sequence), 'batch' (csrc/sasrec_batch.hip, row-space
    kernels), default by batch size.  drop_p > 0: training-mode dropout of both residual branches of every layer
    (utils/layers.py:104-117), mask from the counter-based stream keyed by seed[0] (int64 [1] on the device); only
    the batch-level kernels carry it (rc_sasrec_batch_fwd_dropout), so dropout selects them."""
    B, L = hist.shape
    d = item_emb.shape[1]
    dev, f32 = hist.device, torch.float32
    hv = torch.empty((B, d), dtype=f32, device=dev)
    lib = _lib.load()
    if drop_p > 0.0:
        if impl == "sequence":
            raise ValueError("SASRec dropout is implemented by the batch-level kernels only (impl='batch')")
        impl = "batch"
    impl = _sasrec_impl(B, L, impl, *((d, int(n_heads), len(layers)) if drop_p == 0.0 else ()))
    if impl == "batch":
        # eval passes reuse one scratch state; training passes own theirs until the backward has run
        n_state = lib.rc_sasrec_batch_state_floats(B, L, d, len(layers))
        state = (torch.empty(n_state, dtype=f32, device=dev) if save
                 else workspace(4 * n_state, dev, "sasrec_state").view(f32)[:n_state])
        ws = workspace(lib.rc_sasrec_batch_workspace_bytes(B, L, d, len(layers)), dev, "sasrec_batch")
        _lib.call("rc_sasrec_batch_fwd_dropout", _ptr(item_emb, f32, "item_emb"), _ptr(pos_emb, f32, "pos_emb"),
                  _sas_ptr_table(layers), len(layers), int(n_heads), _ptr(hist, torch.int64, "hist"),
                  _ptr(lengths, torch.int64, "lengths"), B, L, d, *_drop_args(drop_p, seed), _ptr(hv, f32, "hv"),
                  _ptr(state, f32, "state"), C.c_void_p(ws.data_ptr()), ws.numel(), _stream())
        return hv, (SasSaved("batch", state, B, len(layers), L, d) if save else None)
    xsave = torch.empty((B, len(layers), L, d), dtype=f32, device=dev) if save else None
    ws = workspace(lib.rc_sasrec_workspace_bytes(B, d, len(layers)), dev, "sasrec")
    _lib.call("rc_sasrec_fwd", _ptr(item_emb, f32, "item_emb"), _ptr(pos_emb, f32, "pos_emb"),
              _sas_ptr_table(layers), len(layers), int(n_heads), _ptr(hist, torch.int64, "hist"),
              _ptr(lengths, torch.int64, "lengths"), B, L, d, _ptr(hv, f32, "hv"),
              _ptr(xsave, f32, "xsave", True), C.c_void_p(ws.data_ptr()), ws.numel(), _stream())
    return hv, (SasSaved("sequence", xsave, B, len(layers), L, d) if save else None)


def _sas_dense_views(dense, n_layers, d):
    grads = []
    for l in range(n_layers):
        off, g = 0, {}
        for name in SAS_LAYER_KEYS:
            n = d * d if name.startswith("W") else d
            g[name] = dense[l, off:off + n].view(d, d) if name.startswith("W") else dense[l, off:off + n]
            off += n
        grads.append(g)
    return grads


def sasrec_bwd(layers, n_heads, lengths, saved, dhv, drop_p=0.0, seed=None, split=False):
    """-> (g_hist [B,L,d], list of per-layer dicts of dense gradients); `saved` from sasrec_fwd(save=True);
    drop_p / seed as given to the forward this is the backward of (the mask is regenerated, not stored).
    split=True (batch-level kernels): only the launches up to the one that completes g_hist are enqueued (rc_sasrec_batch_bwd_part,
    part 1) -> (g_hist, grads, finish): the caller forks whatever waits for g_hist alone, then calls finish() on the same stream
    for the rest (the dense gradient views are complete after it)"""
    B, n_layers, L, d = saved.B, saved.n_layers, saved.L, saved.d
    dev, f32 = dhv.device, torch.float32
    g_hist = torch.empty((B, L, d), dtype=f32, device=dev)
    lib = _lib.load()
    pl = lib.rc_sasrec_dense_param_count(d)
    dense = torch.empty((n_layers, pl), dtype=f32, device=dev)
    if saved.impl == "batch":
        ws = workspace(lib.rc_sasrec_batch_workspace_bytes(B, L, d, n_layers), dev, "sasrec_batch")
        if split:
            args = (_sas_ptr_table(layers), n_layers, int(n_heads), _ptr(lengths, torch.int64, "lengths"), B, L, d, *_drop_args(drop_p, seed),
                    _ptr(saved.data, f32, "state"), _ptr(dhv, f32, "dhv"), _ptr(g_hist, f32, "g_hist"), _ptr(dense, f32, "dense"),
                    C.c_void_p(ws.data_ptr()), ws.numel())
            _lib.call("rc_sasrec_batch_bwd_part", *args, 1, _stream())
            keep = (layers, lengths, saved, dhv, g_hist, dense, ws)      # alive until part 2 is enqueued
            return g_hist, _sas_dense_views(dense, n_layers, d), (lambda: (_lib.call("rc_sasrec_batch_bwd_part", *args, 2, _stream()), keep)[0])
        _lib.call("rc_sasrec_batch_bwd_dropout", _sas_ptr_table(layers), n_layers, int(n_heads),
                  _ptr(lengths, torch.int64, "lengths"), B, L, d, *_drop_args(drop_p, seed), _ptr(saved.data, f32, "state"),
                  _ptr(dhv, f32, "dhv"), _ptr(g_hist, f32, "g_hist"), _ptr(dense, f32, "dense"), C.c_void_p(ws.data_ptr()),
                  ws.numel(), _stream())
    else:
        if drop_p > 0.0:
            raise ValueError("SASRec dropout needs the batch-level kernels")
        ws = workspace(lib.rc_sasrec_workspace_bytes(B, d, n_layers), dev, "sasrec")
        _lib.call("rc_sasrec_bwd", _sas_ptr_table(layers), n_layers, int(n_heads), _ptr(lengths, torch.int64, "lengths"),
                  B, L, d, _ptr(saved.data, f32, "xsave"), _ptr(dhv, f32, "dhv"), _ptr(g_hist, f32, "g_hist"),
                  _ptr(dense, f32, "dense"), C.c_void_p(ws.data_ptr()), ws.numel(), _stream())
    if split:
        return g_hist, _sas_dense_views(dense, n_layers, d), (lambda: None)
    return g_hist, _sas_dense_views(dense, n_layers, d)


def sasrec_pos_grad(g_hist, lengths, n_pos):
    """dense gradient [n_pos, d] of the position table from the history-row gradients g_hist [B, L, d]
    (position id = length - index, SASRec.py:64)"""
    B, L, d = g_hist.shape
    out = torch.empty((n_pos, d), dtype=torch.float32, device=g_hist.device)
    ws = workspace(_lib.load().rc_sasrec_pos_grad_workspace_bytes(B, L, d), g_hist.device, "sasrec_pos")
    _lib.call("rc_sasrec_pos_grad", _ptr(g_hist, torch.float32, "g_hist"), _ptr(lengths, torch.int64, "lengths"), B, L, d,
              int(n_pos), _ptr(out, torch.float32, "grad_pos"), C.c_void_p(ws.data_ptr()), ws.numel(), _stream())
    return out


# ---- shape-generic sequence-encoder layers (csrc/seq_layers.hip) ------------------------------------------------------

def seq_offsets(lengths, L):
    """off int32 [B + 1]: exclusive prefix sums of min(len_b, L) -- valid row counts and compact row indices of the padded batch"""
    B = lengths.numel()
    off = torch.empty(B + 1, dtype=torch.int32, device=lengths.device)
    _lib.call("rc_seq_offsets", _ptr(lengths, torch.int64, "lengths"), B, int(L), _ptr(off, torch.int32, "off"), _stream())
    return off


def seq_embed(item_emb, pos_emb, hist, lengths):
    """X [B * L, d] = item rows + position rows (position id = length - index) on valid rows, 0 on the padding (SASRec.py:58-66)"""
    B, L = hist.shape
    d = item_emb.shape[1]
    f32, i64 = torch.float32, torch.int64
    X = torch.empty((B * L, d), dtype=f32, device=hist.device)
    _lib.call("rc_seq_embed_fwd", _ptr(item_emb, f32, "item_emb"), _ptr(pos_emb, f32, "pos_emb"), _ptr(hist, i64, "hist"),
              _ptr(lengths, i64, "lengths"), B, L, d, _ptr(X, f32, "X"), _stream())
    return X


def seq_pick_last(X, lengths, B, L):
    """hv [B, d] = X[b * L + len_b - 1] (SASRec.py:76)"""
    d = X.shape[1]
    hv = torch.empty((B, d), dtype=torch.float32, device=X.device)
    _lib.call("rc_seq_pick_last_fwd", _ptr(X, torch.float32, "X"), _ptr(lengths, torch.int64, "lengths"), B, int(L), d, _ptr(hv, torch.float32, "hv"), _stream())
    return hv


def seq_pick_last_bwd(dhv, lengths, B, L):
    d = dhv.shape[1]
    dX = torch.empty((B * L, d), dtype=torch.float32, device=dhv.device)
    _lib.call("rc_seq_pick_last_bwd", _ptr(dhv, torch.float32, "dhv"), _ptr(lengths, torch.int64, "lengths"), B, int(L), d, _ptr(dX, torch.float32, "dX"), _stream())
    return dX


def seq_pos_grad(dX, lengths, B, L, n_pos):
    """dense gradient [n_pos, d] of the position table from the rows' gradients dX [B * L, d] (position id = length - index)"""
    d = dX.shape[1]
    out = torch.empty((n_pos, d), dtype=torch.float32, device=dX.device)
    _lib.call("rc_seq_pos_grad", _ptr(dX, torch.float32, "dX"), _ptr(lengths, torch.int64, "lengths"), B, int(L), d, int(n_pos),
              _ptr(out, torch.float32, "grad_pos"), _stream())
    return out


def seq_attention_supported(L, dk):
    return bool(_lib.load().rc_seq_attention_supported(int(L), int(dk)))


def _seq_attn_head(Q, K, V, off, mask, causal, B, L, H):
    f32 = torch.float32
    dk = Q.shape[1] // H
    mb = 0
    if mask is not None:
        if mask.dtype != torch.uint8 or mask.shape[-2:] != (L, L) or mask.numel() not in (L * L, B * L * L):
            raise ValueError("seq_attention: mask must be uint8 [L, L] or [B, L, L]")
        mb = 1 if mask.numel() == B * L * L and B > 1 else 0
    return (_ptr(Q, f32, "Q"), _ptr(K, f32, "K"), _ptr(V, f32, "V"), _ptr(off, torch.int32, "off", True), _ptr(mask, torch.uint8, "mask", True),
            mb, 1 if causal else 0, B, L, H, dk)


def seq_attention_fwd(Q, K, V, off, B, L, H, mask=None, causal=True):
    """Q, K, V [B * L, H * dk] -> (ctx [B * L, H * dk], lse [B, H, L])  (utils/layers.py:52-63, per head)"""
    f32 = torch.float32
    ctx = torch.empty_like(Q)
    lse = torch.empty((B, H, L), dtype=f32, device=Q.device)
    _lib.call("rc_seq_attention_fwd", *_seq_attn_head(Q, K, V, off, mask, causal, B, L, H), _ptr(ctx, f32, "ctx"), _ptr(lse, f32, "lse"), _stream())
    return ctx, lse


def seq_attention_bwd(Q, K, V, off, B, L, H, lse, dctx, mask=None, causal=True):
    """-> (dQ, dK, dV); probabilities recomputed from lse"""
    f32 = torch.float32
    dQ, dK, dV = torch.empty_like(Q), torch.empty_like(Q), torch.empty_like(Q)
    Dv = torch.empty((B, H, L), dtype=f32, device=Q.device)
    _lib.call("rc_seq_attention_bwd", *_seq_attn_head(Q, K, V, off, mask, causal, B, L, H), _ptr(lse, f32, "lse"), _ptr(dctx, f32, "dctx"),
              _ptr(Dv, f32, "Dv"), _ptr(dQ, f32, "dQ"), _ptr(dK, f32, "dK"), _ptr(dV, f32, "dV"), _stream())
    return dQ, dK, dV


def seq_add_layernorm_fwd(A, R, w, b, off, L, drop_p=0.0, seed=None, site=0):
    """Y = LayerNorm(dropout(A) + R) over the rows of A [rows, d] -> (Y, xhat, rstd)  (utils/layers.py:110,117)"""
    f32 = torch.float32
    rows, d = A.shape
    Y, xhat = torch.empty_like(A), torch.empty_like(A)
    rstd = torch.empty(rows, dtype=f32, device=A.device)
    _lib.call("rc_seq_add_layernorm_fwd", _ptr(A, f32, "A"), _ptr(R, f32, "R", True), _ptr(w, f32, "w"), _ptr(b, f32, "b"),
              _ptr(off, torch.int32, "off", True), rows, int(L), d, *_drop_args(drop_p, seed), C.c_uint32(int(site)), _ptr(Y, f32, "Y"),
              _ptr(xhat, f32, "xhat"), _ptr(rstd, f32, "rstd"), _stream())
    return Y, xhat, rstd


def seq_add_layernorm_bwd(dY, xhat, rstd, w, off, L, drop_p=0.0, seed=None, site=0, need_dA=True, need_dR=True):
    """-> (dA | None, dR | None, dw [d], db [d])"""
    f32 = torch.float32
    rows, d = dY.shape
    dA = torch.empty_like(dY) if need_dA else None
    dR = torch.empty_like(dY) if need_dR else None
    dw, db = torch.empty(d, dtype=f32, device=dY.device), torch.empty(d, dtype=f32, device=dY.device)
    ws = workspace(_lib.load().rc_seq_add_layernorm_bwd_workspace_bytes(d), dY.device, "seq_ln_bwd")
    _lib.call("rc_seq_add_layernorm_bwd", _ptr(dY, f32, "dY"), _ptr(xhat, f32, "xhat"), _ptr(rstd, f32, "rstd"), _ptr(w, f32, "w"),
              _ptr(off, torch.int32, "off", True), rows, int(L), d, *_drop_args(drop_p, seed), C.c_uint32(int(site)), _ptr(dA, f32, "dA", True),
              _ptr(dR, f32, "dR", True), _ptr(dw, f32, "dw"), _ptr(db, f32, "db"), C.c_void_p(ws.data_ptr()), ws.numel(), _stream())
    return dA, dR, dw, db


def seg_rows_route(n_occ, n_rows, d):
    """True iff segmented_update2 takes rc_segmented_update_rows (one wave per table row: every row collects many occurrences).
    That route ignores keys >= n_rows, which lets a caller park occurrences that must not take part (padding) behind the table."""
    return _SEG_ROWS and d in (16, 32, 64, 128, 256) and n_occ >= _SEG_ROWS_MIN_PER_ROW * n_rows


def segmented_update2(keys, perm, src, src2, n_split, hyper=None, W=None, m=None, v=None, coef=None,
                      src_index=None, div=1, dense_grad=None, step_dev=None):
    """rc_segmented_update2: occurrences >= n_split take plain rows src2[o - n_split].
    step_dev: int64 device tensor [1] with Adam's step count (hipGraph-capturable; one-wave-per-row route only)"""
    n_occ = keys.numel()
    d = src.shape[-1]
    f32 = torch.float32
    table = W if W is not None else dense_grad
    n_rows = table.shape[0]
    if seg_rows_route(n_occ, n_rows, d):
        # every row collects many occurrences (a small catalogue under a large batch): one wave per table row
        ws = workspace(_lib.load().rc_segmented_rows_workspace_bytes(n_rows, n_occ, d), keys.device, "seg_rows")
        head = (_ptr(W, f32, "W", True), _ptr(m, f32, "m", True), _ptr(v, f32, "v", True), d,
                n_rows, _ptr(keys, torch.int32, "keys"), _ptr(perm, torch.int32, "perm"), n_occ,
                _ptr(coef, f32, "coef", True), _ptr(src, f32, "src"), _ptr(src_index, torch.int64, "src_index", True),
                int(div), _ptr(src2, f32, "src2"), int(n_split), C.byref(hyper) if hyper is not None else None)
        tail = (_ptr(dense_grad, f32, "dense_grad", True), C.c_void_p(ws.data_ptr()), ws.numel(), _stream())
        if step_dev is None:
            _lib.call("rc_segmented_update_rows", *head, *tail)
        else:
            _lib.call("rc_segmented_update_rows_dev", *head, _ptr(step_dev, torch.int64, "step_dev"), *tail)
        return
    if step_dev is not None:
        raise RuntimeError("segmented_update2(step_dev=...): the device-side step count is carried by the one-wave-per-row "
                           "update only (seg_rows_route: a small catalogue under a large batch)")
    ws = workspace(_lib.load().rc_segmented_workspace_bytes(n_occ, d), keys.device, "seg")
    _lib.call("rc_segmented_update2", _ptr(W, f32, "W", True), _ptr(m, f32, "m", True), _ptr(v, f32, "v", True), d,
              _ptr(keys, torch.int32, "keys"), _ptr(perm, torch.int32, "perm"), n_occ,
              _ptr(coef, f32, "coef", True), _ptr(src, f32, "src"), _ptr(src_index, torch.int64, "src_index", True),
              int(div), _ptr(src2, f32, "src2"), int(n_split), 0, 0, C.byref(hyper) if hyper is not None else None,
              _ptr(dense_grad, f32, "dense_grad", True), None, None, 0, C.c_void_p(ws.data_ptr()), ws.numel(), _stream())


def rows_plan_supported(n_rows, n_occ, d):
    return bool(_lib.load().rc_rows_plan_supported(int(n_rows), int(n_occ), int(d)))


_rows_plan_zeroed = set()


class RowsPlan:
    """rc_rows_plan_build: the occurrences of a small table's rows grouped by row, straight from the batch's id tensors
    (ids_a [.., C] then ids_b [B, L]; lengths_b marks the padding slots of ids_b, which take no part) -- three launches, no sort;
    update(): rc_rows_plan_update, the one-wave-per-row reduction of segmented_update2 on that grouping (bit-identical to it)."""

    def __init__(self, ids_a, ids_b, lengths_b, n_rows, d, tag="rows_plan"):
        self.n_a, self.n_b = ids_a.numel(), (ids_b.numel() if ids_b is not None else 0)
        self.n_occ, self.n_rows, self.d = self.n_a + self.n_b, int(n_rows), int(d)
        dev = ids_a.device
        lib = _lib.load()
        self.ws = workspace(lib.rc_rows_plan_workspace_bytes(self.n_rows, self.n_occ, self.d), dev, tag)
        if self.ws.data_ptr() not in _rows_plan_zeroed:   # status counter; everything else is written before it is read
            self.ws.zero_()
            _rows_plan_zeroed.add(self.ws.data_ptr())
        L = ids_b.shape[-1] if (ids_b is not None and lengths_b is not None) else 1
        _lib.call("rc_rows_plan_build", _ptr(ids_a, torch.int64, "ids_a"), self.n_a, _ptr(ids_b, torch.int64, "ids_b", True), self.n_b,
                  _ptr(lengths_b, torch.int64, "lengths_b", True), int(L), self.n_rows, self.d, C.c_void_p(self.ws.data_ptr()),
                  self.ws.numel(), _stream())

    def update(self, src, hyper=None, W=None, m=None, v=None, coef=None, src_index=None, div=1, src2=None, dense_grad=None,
               step_dev=None):
        f32 = torch.float32
        _lib.call("rc_rows_plan_update", _ptr(W, f32, "W", True), _ptr(m, f32, "m", True), _ptr(v, f32, "v", True), self.d, self.n_rows,
                  self.n_occ, _ptr(coef, f32, "coef", True), _ptr(src, f32, "src"), _ptr(src_index, torch.int64, "src_index", True),
                  int(div), _ptr(src2, f32, "src2", True), self.n_a, C.byref(hyper) if hyper is not None else None,
                  _ptr(step_dev, torch.int64, "step_dev", True), _ptr(dense_grad, f32, "dense_grad", True),
                  C.c_void_p(self.ws.data_ptr()), self.ws.numel(), _stream())

    def views(self):
        """(keys, perm, start, end, status) as int32 tensors copied out of the workspace (tests)"""
        ptrs = [C.c_void_p() for _ in range(5)]
        _lib.call("rc_rows_plan_views", C.c_void_p(self.ws.data_ptr()), self.n_rows, self.n_occ, self.d, *[C.byref(p) for p in ptrs])
        base = self.ws.data_ptr()
        words = self.ws.view(torch.int32)
        sizes = [self.n_occ + 1, self.n_occ + 1, self.n_rows, self.n_rows, 1]
        return tuple(words[(p.value - base) // 4:(p.value - base) // 4 + n].clone() for p, n in zip(ptrs, sizes))


class SasrecTrainer:
    """One BaseRunner.fit iteration for SASRec on device tensors.
    P = {"item_emb": [n_items,d], "pos_emb": [max_his+1,d], "layers": [dict(SAS_LAYER_KEYS) ...]}.
    dropout > 0: both residual branches of every layer dropped with a fresh mask per step (device seed counter,
    bumped every step; batch-level kernels)."""

    GRAPH_WARMUP = 2   # eager steps per batch shape before the capture: workspaces and optimizer state exist by then

    def __init__(self, P, n_heads, opt="Adam", lr=1e-3, l2=0.0, rowwise=False, dropout=0.0, seed=0, graph=False):
        """graph=True: from the third step of a batch shape on, the step (about 45 launches on two streams, no host
        synchronisation, grids that depend on shapes only) is replayed from a hipGraph with the batch copied into static
        buffers -- at config 3 the eager step is bound by the host's launch rate (0.42 ms enqueue against 0.27 ms of GPU work).
        Same kernels, same results.  With Adam the step count is kept in device memory (capturable semantics: bias corrections
        formed in the kernels) and the item table must take the one-wave-per-row update (seg_rows_route), rowwise=True.
        The loss tensor a replayed step returns is the graph's own buffer: the next replay overwrites it (clone it to keep it)."""
        self.P, self.n_heads, self.opt, self.lr, self.l2, self.rowwise = P, n_heads, opt, lr, l2, rowwise
        self.dropout = float(dropout)
        self.seed = torch.tensor([seed], dtype=torch.int64, device=P["item_emb"].device) if self.dropout > 0 else None
        self.step_count = 0
        self.loss = None
        self.state = {}
        self._side = None
        self.graph = bool(graph)
        # Adam under replay: the step count lives in device memory (torch.optim.Adam(capturable=True) semantics); carried by the
        # row-wise one-wave-per-row item update and the multi-tensor dense step
        self._step_dev = None
        if self.graph and opt == "Adam":
            if not rowwise:
                raise ValueError("SasrecTrainer(graph=True, opt='Adam') needs rowwise=True (the dense item-table step takes the "
                                 "step count as a kernel argument)")
            self._step_dev = torch.zeros(1, dtype=torch.int64, device=P["item_emb"].device)
        self._graphs, self._graph_seen = {}, {}

    def _st(self, t):
        st = self.state.get(t.data_ptr())
        if st is None:
            st = {}
            if self.opt in ("Adam", "Adagrad"):
                st["m"] = torch.zeros_like(t)
            if self.opt == "Adam":
                st["v"] = torch.zeros_like(t)
            self.state[t.data_ptr()] = st
        return st

    def _side_stream(self, dev):
        """the trainer's second stream: the id sort (needs only the batch) runs beside the encoder, the position-table
        gradient beside the item-table update -- all of them latency-bound launches that leave most of the chip idle"""
        if self._side is None:
            self._side = torch.cuda.Stream(device=dev)
        return self._side

    def _dev_step_route(self, hist, iid):
        """Adam with the step count in device memory: carried by the one-wave-per-row item update (seg_rows_route) only"""
        if self._step_dev is None:
            return False
        I = self.P["item_emb"]
        n_occ = iid.numel() + hist.numel()
        use_plan = _SASREC_PLAN and n_occ >= _EDB_PLAN_MIN and plan_supported(n_occ, 0, I.shape[0], 0)
        return not use_plan and seg_rows_route(n_occ, I.shape[0], I.shape[1])

    def step(self, hist, lengths, iid):
        if not (self.graph and hist.is_cuda) or getattr(self, "timing", None) is not None \
                or (self._step_dev is not None and not self._dev_step_route(hist, iid)):
            return self._step(hist, lengths, iid)
        from . import graph as hgraph
        if not hgraph.usable():
            raise RuntimeError("SasrecTrainer(graph=True): hipGraph replay needs DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 before HIP "
                               "initialises (rechorus_amd/graph.py); import rechorus_amd before touching the GPU")
        # lr / l2 / the optimizer are kernel arguments frozen into a captured step: they are part of the key
        key = (tuple(hist.shape), tuple(iid.shape), self.opt, float(self.lr), float(self.l2))
        entry = self._graphs.get(key)
        if entry is None:
            seen = self._graph_seen.get(key, 0)
            if seen < self.GRAPH_WARMUP:
                self._graph_seen[key] = seen + 1
                return self._step(hist, lengths, iid)
            # one flat id buffer, the three inputs are views into it: the per-step copy is one launch
            flat = torch.cat([hist.reshape(-1), lengths.reshape(-1), iid.reshape(-1)])
            n0, n1 = hist.numel(), hist.numel() + lengths.numel()
            static = (flat[:n0].view(hist.shape), flat[n0:n1].view(lengths.shape), flat[n1:].view(iid.shape), flat)
            torch.cuda.synchronize(hist.device)
            g = torch.cuda.CUDAGraph()
            cap = torch.cuda.Stream(device=hist.device)
            with torch.cuda.stream(cap):
                with torch.cuda.graph(g, stream=cap):
                    self._step(*static[:3])
            self.step_count -= 1   # the capture recorded a step, it did not train one (its device-side increments are graph nodes)
            entry = self._graphs[key] = (g, static, self.loss)
            torch.cuda.synchronize(hist.device)
            # (the capture recorded the step without running it: this batch is trained by the replay below)
        g, static, loss = entry
        _lib.call("rc_stage_batch", _ptr(hist, torch.int64, "hist"), hist.numel(), _ptr(lengths, torch.int64, "lengths"), lengths.numel(),
                  _ptr(iid, torch.int64, "iid"), iid.numel(), _ptr(static[3], torch.int64, "static"), _stream())
        g.replay()
        # the host's count follows every trained step -- replayed or eager -- so that a batch shape that leaves the captured route
        # (the short last batch of an epoch) steps Adam with the right bias corrections
        self.step_count += 1
        self.loss = loss
        return loss

    def _sorted_occurrences(self, hist, lengths, iid):
        """candidate + history ids, sorted.  On the one-wave-per-row route the padding slots of the history windows (id 0, zero
        gradient rows: half of B * history_max occurrences of ONE row, a 400-chunk hot row) are parked behind the table (key =
        n_items, ignored there) except the first of them, which keeps row 0 among the touched rows exactly as before -- a sum of
        zero rows is zero either way."""
        I = self.P["item_emb"]
        n_items, d = I.shape
        L = hist.shape[1]
        n_occ = iid.numel() + hist.numel()
        if not seg_rows_route(n_occ, n_items, d):
            return sort_ids(torch.cat([iid.reshape(-1), hist.reshape(-1)]), n_items)
        pad = (torch.arange(L, device=hist.device)[None, :] >= lengths[:, None]).reshape(-1)
        hid = torch.where(pad, hist.new_full((), n_items), hist.reshape(-1))
        first_pad = pad.to(torch.int32).argmax().reshape(1)   # position of the first padding slot (0 if there is none)
        hid.scatter_(0, first_pad, hist.reshape(-1).gather(0, first_pad))   # (tensor-indexed assignment would read the index back: a host sync)
        return sort_ids(torch.cat([iid.reshape(-1), hid]), n_items + 1)

    def _item_table_update(self, sorted_ids, rows_plan, hv, gpred, g_hist, h, step_dev, B, Cn):
        """candidate occurrences (g * hv, rebuilt on the fly) + history occurrences (g_hist rows) of the item table, grouped by
        `sorted_ids` (a RowsPlan, or (keys, perm) of the sort): row-wise optimizer step, or dense gradient + dense step"""
        I = self.P["item_emb"]
        d = I.shape[1]
        st = self._st(I)
        if rows_plan:
            src = dict(coef=gpred.reshape(-1), div=Cn, src2=g_hist.view(-1, d))
            if self.rowwise:
                sorted_ids.update(hv, hyper=h, W=I, m=st.get("m"), v=st.get("v"), step_dev=step_dev, **src)
            else:
                G = torch.zeros_like(I)
                sorted_ids.update(hv, dense_grad=G, **src)
                dense_update(I, G, h, st.get("m"), st.get("v"))
        elif self.rowwise:
            keys, perm = sorted_ids
            segmented_update2(keys, perm, hv, g_hist.view(-1, d), B * Cn, hyper=h, W=I, m=st.get("m"), v=st.get("v"),
                              coef=gpred.reshape(-1), div=Cn, step_dev=step_dev)
        else:
            keys, perm = sorted_ids
            G = torch.zeros_like(I)
            segmented_update2(keys, perm, hv, g_hist.view(-1, d), B * Cn, coef=gpred.reshape(-1), div=Cn, dense_grad=G)
            dense_update(I, G, h, st.get("m"), st.get("v"))

    def _dense_step(self, Gp, dgrads, h, h0, step_dev):
        """position table + every block parameter: one launch"""
        P = self.P
        Pe, layers = P["pos_emb"], P["layers"]
        st = self._st(Pe)
        items = [(Pe, Gp, h, st.get("m"), st.get("v"))]
        for lay, g in zip(layers, dgrads):
            for name in SAS_LAYER_KEYS:
                st = self._st(lay[name])
                items.append((lay[name], g[name].contiguous(), h0 if name in SAS_NO_DECAY else h, st.get("m"), st.get("v")))
        dense_update_multi(items, self.opt, step_dev=step_dev, increment=False)

    def _score_loss(self, hv, iid, B):
        # scores, BPR loss, d loss / d pred and d loss / d hv in ONE pass over the candidate rows: the fused BPRMF kernel
        # with the encoder output as the "user" row (SASRec.py:80-81, BaseModel.py:182-185); three launches and two
        # more passes over the [B, C] rows before
        # (one tensor per batch size, kept for the trainer's life: a captured step holds its ADDRESS -- replacing it when another
        #  batch size comes by, e.g. the short last batch of an epoch, would leave the graph of the first size reading freed memory)
        rows_by = self.__dict__.setdefault("_rows_by", {})
        key_rows = (B, str(hv.device))
        if key_rows not in rows_by:
            rows_by[key_rows] = torch.arange(B, device=hv.device)
        _, loss_vec, gpred, dhv = bprmf_fwd_bwd(hv, self.P["item_emb"], rows_by[key_rows], iid, want_pred=False)
        return loss_vec, gpred, dhv

    def _step_item_stream(self, hist, lengths, iid, h, h0, step_dev, rows_plan, encoder_first):
        """The step on two streams, one of which owns everything about the ITEM TABLE: the grouping of the batch's ids beside the
        encoder, then -- as soon as the history rows' gradient is complete (rc_sasrec_batch_bwd_part) -- the table update, while the
        other stream finishes the encoder's parameter gradients, the position gradient and the dense step.  One fork, one
        hand-over (g_hist ready), one join.  (Round 5: the table update ran on the encoder's stream behind a second join, the
        position gradient on the side: two more cross-queue edges of ~11 us each on the critical path of a replayed step.)
        encoder_first: the encoder rides the caller's stream (in a captured step: the launch stream, the other branch starts behind a
        cross-queue barrier); else the item-table work does, and the LAST kernels of the step -- the table update's -- are on the
        launch stream, where the graph ends without waiting for another queue."""
        P = self.P
        I, Pe, layers = P["item_emb"], P["pos_emb"], P["layers"]
        B, L = hist.shape
        Cn = iid.shape[1]
        d = I.shape[1]
        cur, other = torch.cuda.current_stream(hist.device), self._side_stream(hist.device)
        enc, tab = (cur, other) if encoder_first else (other, cur)
        # the batch is ready; last step's users of the other stream's buffers are done (the step ends with a join)
        other.wait_event(cur.record_event())
        box = {}

        def plan():
            with torch.cuda.stream(tab):
                box["ids"] = RowsPlan(iid, hist, lengths, I.shape[0], d, tag="sasrec_rows") if rows_plan else self._sorted_occurrences(hist, lengths, iid)

        def forward():
            with torch.cuda.stream(enc), _PhaseTimer(self, "encoder_fwd"):
                box["hv"], box["x"] = sasrec_fwd(I, Pe, layers, self.n_heads, hist, lengths, save=True, drop_p=self.dropout, seed=self.seed)

        for part in ((forward, plan) if encoder_first else (plan, forward)):   # (capture order: see encoder_first)
            part()
        hv = box["hv"]
        with torch.cuda.stream(enc):
            with _PhaseTimer(self, "score_loss"):
                loss_vec, gpred, dhv = self._score_loss(hv, iid, B)
            with _PhaseTimer(self, "encoder_bwd"):
                g_hist, dgrads, finish_bwd = sasrec_bwd(layers, self.n_heads, lengths, box["x"], dhv, drop_p=self.dropout, seed=self.seed, split=True)
            ready = enc.record_event()     # g_hist, gpred, hv: what the table update reads
            with _PhaseTimer(self, "dense_update"):
                finish_bwd()               # the encoder's parameter gradients ...
                Gp = sasrec_pos_grad(g_hist, lengths, Pe.shape[0])     # ... the position table's, and the dense step of both
                self._dense_step(Gp, dgrads, h, h0, step_dev)
            # the batch mean of the losses last on this stream: it ends ~20 us before the item-table stream does, and 5 us in front of
            # the table update were 5 us of the step
            self.loss = reduce_sum(loss_vec, 1.0 / B)
        with torch.cuda.stream(tab):
            tab.wait_event(ready)
            with _PhaseTimer(self, "table_update"):
                self._item_table_update(box["ids"], rows_plan, hv, gpred, g_hist, h, step_dev, B, Cn)
        cur.wait_stream(other)             # the step's one join
        return self.loss

    def _step(self, hist, lengths, iid):
        P = self.P
        I, Pe, layers = P["item_emb"], P["pos_emb"], P["layers"]
        B, L = hist.shape
        Cn = iid.shape[1]
        d = I.shape[1]
        self.step_count += 1
        h = make_hyper(self.opt, lr=self.lr, l2=self.l2, step=self.step_count)
        h0 = make_hyper(self.opt, lr=self.lr, l2=0.0, step=self.step_count)
        if self.seed is not None:
            step_increment(self.seed)
        if self._step_dev is not None:
            step_increment(self._step_dev)
        n_occ = B * Cn + hist.numel()
        use_plan = _SASREC_PLAN and n_occ >= _EDB_PLAN_MIN and plan_supported(n_occ, 0, I.shape[0], 0)
        # Adam under replay: the kernels read the step count from device memory where the route carries it (else this step takes
        # the host's count -- both advance every step -- and is not captured)
        step_dev = self._step_dev if self._dev_step_route(hist, iid) else None
        # (small batches are bound by the host's launch rate: the extra stream switches cost more than the overlap returns --
        #  B = 256: 0.39 against 0.30 ms; B = 4096: 0.66 against 0.72 ms)
        overlap = _SAS_OVERLAP and hist.is_cuda and not use_plan and n_occ >= _SAS_OVERLAP_MIN
        sorted_ids = sort_done = main = side = None
        # one wave per table row with the rows' bounds from a counting sort of the id tensors themselves (no radix sort, no glue)
        rows_plan = (_SAS_ROWS_PLAN and hist.is_cuda and not use_plan and seg_rows_route(n_occ, I.shape[0], d)
                     and rows_plan_supported(I.shape[0], n_occ, d))
        if overlap and _SAS_SCHED in ("2", "3"):
            return self._step_item_stream(hist, lengths, iid, h, h0, step_dev, rows_plan, encoder_first=_SAS_SCHED == "2")

        if overlap:      # RC_SAS_SCHED=1, the round-5 schedule: plan on the side stream, joined before the table update
            main, side = torch.cuda.current_stream(hist.device), self._side_stream(hist.device)
            side.wait_stream(main)   # the batch is ready; last step's readers of the side stream's buffers are done
            with torch.cuda.stream(side):
                sorted_ids = RowsPlan(iid, hist, lengths, I.shape[0], d, tag="sasrec_rows") if rows_plan else self._sorted_occurrences(hist, lengths, iid)
                sort_done = side.record_event()
        with _PhaseTimer(self, "encoder_fwd"):
            hv, xsave = sasrec_fwd(I, Pe, layers, self.n_heads, hist, lengths, save=True, drop_p=self.dropout, seed=self.seed)
        with _PhaseTimer(self, "score_loss"):
            loss_vec, gpred, dhv = self._score_loss(hv, iid, B)
            if not overlap:
                self.loss = reduce_sum(loss_vec, 1.0 / B)   # (two streams: the mean is formed on the side stream below)
        with _PhaseTimer(self, "encoder_bwd"):
            g_hist, dgrads = sasrec_bwd(layers, self.n_heads, lengths, xsave, dhv, drop_p=self.dropout, seed=self.seed)
        # item table: candidate occurrences (g * hv, rebuilt on the fly) + history occurrences (g_hist rows)
        Gp = None
        if overlap:
            side.wait_stream(main)   # g_hist is ready
            with torch.cuda.stream(side):
                self.loss = reduce_sum(loss_vec, 1.0 / B)   # nothing on the main stream waits for the mean of the loss
                Gp = sasrec_pos_grad(g_hist, lengths, Pe.shape[0])
        _upd = _PhaseTimer(self, "table_update")
        _upd.__enter__()
        st = self._st(I)
        if use_plan:
            # bucket plan of candidate + history ids.  The padding slots of the history windows (id 0, zero gradient rows:
            # half of B * history_max occurrences of ONE row) are marked "takes no part" (negative id) except the first of
            # them, which keeps row 0 among the touched rows exactly as before -- a sum of zero rows is zero either way.
            L = hist.shape[1]
            pad = torch.arange(L, device=hist.device)[None, :] >= lengths[:, None]
            hid = torch.where(pad, hist.new_full((), -1), hist).reshape(-1)
            first_pad = pad.reshape(-1).to(torch.int32).argmax().reshape(1)   # position of the first padding slot (0 if there is none)
            hid.scatter_(0, first_pad, hist.reshape(-1).gather(0, first_pad))   # (no tensor-indexed assignment: it syncs the host)
            plan = Plan(torch.cat([iid.reshape(-1), hid]), I.shape[0], tag="sasrec")
            src = dict(coef=gpred.reshape(-1), src=hv, div=Cn, src2=g_hist.view(-1, d), n_split=B * Cn)
            if self.rowwise:
                plan.update("a", I, h, m=st.get("m"), v=st.get("v"), **src)
            else:
                G = plan.row_sums("a", torch.zeros_like(I), **src)
                dense_update(I, G, h, st.get("m"), st.get("v"))
        else:
            if overlap:
                main.wait_event(sort_done)
            elif rows_plan:
                sorted_ids = RowsPlan(iid, hist, lengths, I.shape[0], d, tag="sasrec_rows")
            else:
                sorted_ids = self._sorted_occurrences(hist, lengths, iid)
            self._item_table_update(sorted_ids, rows_plan, hv, gpred, g_hist, h, step_dev, B, Cn)
        _upd.__exit__()
        # position table (tiny): dense gradient, dense step
        _dns = _PhaseTimer(self, "dense_update")
        _dns.__enter__()
        if overlap:
            main.wait_stream(side)
        else:
            Gp = sasrec_pos_grad(g_hist, lengths, Pe.shape[0])
        self._dense_step(Gp, dgrads, h, h0, step_dev)
        _dns.__exit__()
        return self.loss


# ---- list-wise softmax cross-entropy -------------------------------------------------------------------

def softmax_ce(pred, target, max_pos, need_grad=True):
    """ImpressionModel.loss with loss_n='softmaxCE' (models/BaseImpressionModel.py:96-107).
    pred [B,n] fp32, target [B,n] int64 in {1,0,-1}; -> (loss [1], gpred | None)"""
    B, n = pred.shape
    dev, f32 = pred.device, torch.float32
    loss_vec = torch.empty(B, dtype=f32, device=dev)
    h_sum = torch.empty(1, dtype=f32, device=dev)
    gpred = torch.empty_like(pred) if need_grad else None
    _lib.call("rc_softmax_ce_fwd_bwd", _ptr(pred, f32, "pred"), _ptr(target, torch.int64, "target"), B, n, int(max_pos),
              _ptr(loss_vec, f32, "loss_vec"), _ptr(h_sum, f32, "h_sum"), _ptr(gpred, f32, "gpred", True), _stream())
    return reduce_sum(loss_vec, 1.0), gpred


def list_bpr(pred, target, max_pos, hard=False, need_grad=True):
    """ImpressionModel.loss with loss_n 'BPR' / 'BPRhard' (models/BaseImpressionModel.py:50-89)
    -> (loss [1], gpred | None)"""
    B, n = pred.shape
    f32 = torch.float32
    loss_vec = torch.empty(B, dtype=f32, device=pred.device)
    gpred = torch.empty_like(pred) if need_grad else None
    _lib.call("rc_list_bpr_fwd_bwd", _ptr(pred, f32, "pred"), _ptr(target, torch.int64, "target"), B, n, int(max_pos),
              1 if hard else 0, 1.0 / B, _ptr(loss_vec, f32, "loss_vec"), _ptr(gpred, f32, "gpred", True), _stream())
    return reduce_sum(loss_vec, 1.0 / B), gpred


LIST_KINDS = {"BPR": 0, "BPRhard": 1, "BPRafter": 2, "BPRhardafter": 3, "BPRbefore": 4, "BPRhardbefore": 5, "listnet": 6,
              "softmaxCE": 7, "attention_rank": 8, "BPRsimple": 9}


def list_kind(loss_n):
    """ImpressionModel's --loss_n -> rc_list_kind, by the reference's own substring rules (models/BaseImpressionModel.py:50-89:
    any name containing 'BPR'; 'hard' / 'after' / 'before' / 'simple' anywhere in it, in the reference's elif order); None
    for unknown names"""
    if "BPR" in loss_n:
        if "simple" in loss_n and "after" not in loss_n and "before" not in loss_n:
            return LIST_KINDS["BPRsimple"]
        base = "BPRhard" if "hard" in loss_n else "BPR"
        return LIST_KINDS[base + ("after" if "after" in loss_n else ("before" if "before" in loss_n else ""))]
    return LIST_KINDS.get(loss_n)


def list_loss(pred, target, max_pos, kind, need_grad=True):
    """rc_list_loss_fwd_bwd: any list-wise loss of ImpressionModel.loss -> (loss [1] -- [B] unreduced for 'BPR...simple' --,
    gpred | None); kind from list_kind()"""
    B, n = pred.shape
    dev, f32 = pred.device, torch.float32
    loss_vec = torch.empty(B, dtype=f32, device=dev)
    h_sum = torch.empty(1, dtype=f32, device=dev)
    gpred = torch.empty_like(pred) if need_grad else None
    _lib.call("rc_list_loss_fwd_bwd", _ptr(pred, f32, "pred"), _ptr(target, torch.int64, "target"), B, n, int(max_pos), int(kind),
              1.0 if kind == LIST_KINDS["BPRsimple"] else 1.0 / B, _ptr(loss_vec, f32, "loss_vec"), _ptr(h_sum, f32, "h_sum"),
              _ptr(gpred, f32, "gpred", True), _stream())
    if kind == LIST_KINDS["BPRsimple"]:
        return loss_vec, gpred        # unreduced rows [B] (reference :83); gpred[b] = d loss_vec[b] / d pred[b]
    return reduce_sum(loss_vec, 1.0 / B if kind <= 5 else 1.0), gpred


# ---- factorization-machine term, BCE ----------------------------------------------------------------------

def fm_second_order(V):
    """V [..., F, d] stacked field vectors -> 0.5*sum_k((sum_f v)^2 - sum_f v^2), shape [...]
    (models/context/FM.py:61, DeepFM.py:22-23)"""
    F, d = V.shape[-2], V.shape[-1]
    n = V.numel() // (F * d)
    out = torch.empty(V.shape[:-2], dtype=torch.float32, device=V.device)
    _lib.call("rc_fm_second_order_fwd", _ptr(V, torch.float32, "V"), n, F, d, _ptr(out, torch.float32, "out"), _stream())
    return out


def fm_second_order_bwd(V, gout, add=None):
    """d fm_second_order / dV times gout, plus `add` (the same vectors' gradient through another consumer) when given"""
    F, d = V.shape[-2], V.shape[-1]
    n = V.numel() // (F * d)
    dV = torch.empty_like(V)
    if add is None:
        _lib.call("rc_fm_second_order_bwd", _ptr(V, torch.float32, "V"), _ptr(gout, torch.float32, "gout"), n, F, d,
                  _ptr(dV, torch.float32, "dV"), _stream())
    else:
        _lib.call("rc_fm_second_order_bwd_add", _ptr(V, torch.float32, "V"), _ptr(gout, torch.float32, "gout"), n, F, d,
                  _ptr(add, torch.float32, "add"), _ptr(dV, torch.float32, "dV"), _stream())
    return dV


def bce_ranking(pred, need_grad=True):
    """ContextModel.loss with loss_n 'BCE' (models/BaseContextModel.py:53-56) -> (loss [1], gpred | None)"""
    B, Cn = pred.shape
    f32 = torch.float32
    loss_vec = torch.empty(B, dtype=f32, device=pred.device)
    gpred = torch.empty_like(pred) if need_grad else None
    _lib.call("rc_bce_ranking_fwd_bwd", _ptr(pred, f32, "pred"), B, Cn, 1.0 / B, _ptr(loss_vec, f32, "loss_vec"),
              _ptr(gpred, f32, "gpred", True), _stream())
    return reduce_sum(loss_vec, 1.0 / B), gpred


FIELD_IDS, FIELD_F32, FIELD_F64, FIELD_I64 = 0, 1, 2, 3   # enum rc_field_kind


def field_kind(values):
    """rc_field_kind of a numeric feature's value tensor (what `feed_dict[f].float()` of models/context/FM.py:47-48 starts from)"""
    kind = {torch.float32: FIELD_F32, torch.float64: FIELD_F64, torch.int64: FIELD_I64}.get(values.dtype)
    if kind is None:
        raise ValueError("numeric field values must be float32, float64 or int64, got {}".format(values.dtype))
    return kind


def gather_fields(tables, ids, n_cand, want_cid=True, tables1=None, mark=None, kinds=None, numeric_key=-1, fm=False, plan=False, bump=None):
    """tables: list of F [vocab_f, d] tensors; ids: list of F int64 tensors, [B] (per-row field) or [B, C]
    -> (out [B, C, F, d], cid [B, C, F] | None, row_offset list): all field lookups in one launch.
    tables1: F [vocab_f, 1] tables looked up with the same ids (rc_gather_fields_pair) -> (out, out1 [B, C, F, 1], cid, offsets)
    mark = (row_flags int32 [sum of vocab sizes], step_dev int64 [1]): the looked-up rows are stamped with the number of the
    step in progress, step_dev + 1 (rc_gather_fields_pair_mark, for dense_update_rows)
    kinds: per field FIELD_IDS or the value type of a NUMERIC field (rc_gather_fields_mixed; models/context/FM.py:38-41,47-48):
    tables[f] is then the Linear(1, d) weight [d, 1], tables1[f] the Linear(1, 1) weight [1, 1], ids[f] the feature's values;
    such a field owns no row of the concatenated table and its occurrences carry `numeric_key` in cid
    fm / plan (rc_gather_fields_fused; d in 16 / 32 / 64 / 128, tables1 given): the same launch also forms the FM pairwise term
    [B, C] + the field sums [B, C, d], and / or groups the composite keys for the backward pass's row sums (small batches) into a
    fresh workspace -> (out, out1, cid, offsets, fm_term | None, fm_sum | None, plan_ws | None); bump: an int64 [1] device counter
    the same launch increments (the caller promises nothing in the launch reads it; recorded for step_increment: bumped_early)"""
    F = len(tables)
    kinds = [FIELD_IDS] * F if kinds is None else [int(k) for k in kinds]
    mixed = any(k != FIELD_IDS for k in kinds)
    d = next((t.shape[1] for t, k in zip(tables, kinds) if k == FIELD_IDS), tables[0].shape[0])
    B = ids[0].shape[0]
    dev, f32, i64 = tables[0].device, torch.float32, torch.int64
    out = torch.empty((B, n_cand, F, d), dtype=f32, device=dev)
    cid = torch.empty((B, n_cand, F), dtype=i64, device=dev) if want_cid else None
    offs, run = [], 0
    for t, k in zip(tables, kinds):
        if tuple(t.shape[-2:]) != ((t.shape[0], d) if k == FIELD_IDS else (d, 1)):
            raise ValueError("gather_fields: all tables need the same width (a numeric field's weight is [d, 1])")
        offs.append(run)
        run += t.shape[0] if k == FIELD_IDS else 0
    for x, k in zip(ids, kinds):
        if k != FIELD_IDS and field_kind(x) != k:
            raise ValueError("gather_fields: kind {} does not match the value dtype {}".format(k, x.dtype))
    tab_arr = (C.c_void_p * F)(*[_ptr(t, f32, "table").value for t in tables])
    ids_arr = (C.c_void_p * F)(*[_ptr(x, i64 if k == FIELD_IDS else x.dtype, "ids").value for x, k in zip(ids, kinds)])
    per_row = (C.c_int * F)(*[1 if x.dim() == 1 else 0 for x in ids])
    off_arr = (C.c_int64 * F)(*offs)
    out1 = tab1_arr = None
    if tables1 is not None:
        if len(tables1) != F or any(t1.shape != ((t.shape[0], 1) if k == FIELD_IDS else (1, 1)) for t, t1, k in zip(tables, tables1, kinds)):
            raise ValueError("gather_fields: the second family must be [vocab_f, 1] tables of the same vocabularies")
        out1 = torch.empty((B, n_cand, F, 1), dtype=f32, device=dev)
        tab1_arr = (C.c_void_p * F)(*[_ptr(t, f32, "table1").value for t in tables1])
    flags = step_dev = None
    if mark is not None:
        flags, step_dev = mark
        if tables1 is None:
            raise ValueError("gather_fields: row flags come with the pair gather")
        if flags.numel() != run:
            raise ValueError("gather_fields: one row flag per row of the concatenated tables")
    if fm or plan:
        if tables1 is None or d not in (16, 32, 64, 128):
            raise ValueError("gather_fields: the FM term / the plan ride with the pair gather at d in 16 / 32 / 64 / 128")
        kind_arr = (C.c_int * F)(*kinds)
        fm_term = torch.empty((B, n_cand), dtype=f32, device=dev) if fm else None
        fm_sum = torch.empty((B, n_cand, d), dtype=f32, device=dev) if fm else None
        plan_ws = None
        if plan:
            # not the name-keyed workspace cache: the plan lives from this forward to its backward
            plan_ws = torch.empty(_lib.load().rc_small_row_sums_workspace_bytes(B * n_cand * F), dtype=torch.uint8, device=dev)
        _lib.call("rc_gather_fields_fused", tab_arr, tab1_arr, ids_arr, per_row, kind_arr, int(numeric_key), off_arr, F, B, int(n_cand), d,
                  _ptr(out, f32, "out"), _ptr(out1, f32, "out1"), _ptr(cid, i64, "cid", True), _ptr(flags, torch.int32, "row_flags", True),
                  _ptr(step_dev, i64, "step_dev", True), 1, _ptr(fm_term, f32, "fm_out", True), _ptr(fm_sum, f32, "fm_sum", True),
                  C.c_void_p(plan_ws.data_ptr()) if plan else None, plan_ws.numel() if plan else 0, _ptr(bump, i64, "bump", True), _stream())
        if bump is not None:
            _bumped_early.append(bump)
        return out, out1, cid, offs + [run], fm_term, fm_sum, plan_ws
    if bump is not None:
        raise ValueError("gather_fields: a counter rides in the fused launch only (fm / plan)")
    if mixed:
        kind_arr = (C.c_int * F)(*kinds)
        _lib.call("rc_gather_fields_mixed", tab_arr, tab1_arr, ids_arr, per_row, kind_arr, int(numeric_key), off_arr, F, B, int(n_cand), d,
                  _ptr(out, f32, "out"), _ptr(out1, f32, "out1", True), _ptr(cid, i64, "cid", True), _ptr(flags, torch.int32, "row_flags", True),
                  _ptr(step_dev, i64, "step_dev", True), 1, _stream())
    elif mark is not None:
        _lib.call("rc_gather_fields_pair_mark", tab_arr, tab1_arr, ids_arr, per_row, off_arr, F, B, int(n_cand), d,
                  _ptr(out, f32, "out"), _ptr(out1, f32, "out1"), _ptr(cid, i64, "cid", True), _ptr(flags, torch.int32, "row_flags"),
                  _ptr(step_dev, i64, "step_dev"), 1, _stream())
    elif tables1 is not None:
        _lib.call("rc_gather_fields_pair", tab_arr, tab1_arr, ids_arr, per_row, off_arr, F, B, int(n_cand), d, _ptr(out, f32, "out"),
                  _ptr(out1, f32, "out1"), _ptr(cid, i64, "cid", True), _stream())
    else:
        _lib.call("rc_gather_fields", tab_arr, ids_arr, per_row, off_arr, F, B, int(n_cand), d, _ptr(out, f32, "out"),
                  _ptr(cid, i64, "cid", True), _stream())
    if tables1 is not None:
        return out, out1, cid, offs + [run]
    return out, cid, offs + [run]


def numeric_field_grads(gV, gL, values, fields, n_fields, n_cand, d):
    """weight gradients of the numeric fields (rc_numeric_field_grads): gV [n, F, d] | None, gL [n, F] | None per-occurrence
    gradient blocks, values[j] / fields[j] the value tensor and field index of numeric field j
    -> (list of dW [d, 1] | None, list of dw1 [1, 1] | None)"""
    J = len(values)
    src = gV if gV is not None else gL
    dev, f32 = src.device, torch.float32
    B = values[0].shape[0]
    n = B * n_cand
    dW = torch.empty((J, d, 1), dtype=f32, device=dev) if gV is not None else None
    dw1 = torch.empty((J, 1, 1), dtype=f32, device=dev) if gL is not None else None
    kinds = [field_kind(x) for x in values]
    val_arr = (C.c_void_p * J)(*[_ptr(x, x.dtype, "values").value for x in values])
    per_row = (C.c_int * J)(*[1 if x.dim() == 1 else 0 for x in values])
    kind_arr = (C.c_int * J)(*kinds)
    field_arr = (C.c_int * J)(*[int(f) for f in fields])
    dW_arr = (C.c_void_p * J)(*[dW[j].data_ptr() for j in range(J)]) if dW is not None else None
    dw1_arr = (C.c_void_p * J)(*[dw1[j].data_ptr() for j in range(J)]) if dw1 is not None else None
    nbytes = _lib.load().rc_numeric_field_grads_workspace_bytes(n, J, d)
    ws = workspace(nbytes, dev, "numeric_fields")
    _lib.call("rc_numeric_field_grads", _ptr(gV, f32, "gV", True), _ptr(gL, f32, "gL", True), val_arr, per_row, kind_arr, field_arr, J,
              int(n_fields), B, int(n_cand), int(d), dW_arr, dw1_arr, C.c_void_p(ws.data_ptr()), ws.numel(), _stream())
    return (None if dW is None else [dW[j] for j in range(J)]), (None if dw1 is None else [dw1[j] for j in range(J)])


def bce_prob(p, y, need_grad=True):
    """nn.BCELoss()(p, y) on probabilities (models/BaseModel.py:259-267) -> (loss [1], dloss/dp | None)"""
    n = p.numel()
    f32 = torch.float32
    loss_vec = torch.empty(n, dtype=f32, device=p.device)
    gp = torch.empty_like(p) if need_grad else None
    _lib.call("rc_bce_prob_fwd_bwd", _ptr(p, f32, "p"), _ptr(y, f32, "y"), n, 1.0 / n, _ptr(loss_vec, f32, "loss_vec"),
              _ptr(gp, f32, "gp", True), _stream())
    return reduce_sum(loss_vec, 1.0 / n), gp


def ctr_head(bias, lin, term1, term2, label):
    """rc_ctr_head_fwd_bwd: p = sigmoid(bias + lin.sum(-1) (+ term1) (+ term2)), BCE loss [1] and d loss / d logit [n].
    lin [n, F] fp32; term1 / term2 [n] | None; label int64 [n]"""
    n, F = lin.shape
    f32 = torch.float32
    p = torch.empty(n, dtype=f32, device=lin.device)
    loss_vec = torch.empty(n, dtype=f32, device=lin.device)
    gz = torch.empty(n, dtype=f32, device=lin.device)
    _lib.call("rc_ctr_head_fwd_bwd", _ptr(bias, f32, "bias"), _ptr(lin, f32, "lin"), int(F), _ptr(term1, f32, "term1", True),
              _ptr(term2, f32, "term2", True), _ptr(label, torch.int64, "label"), n, _ptr(p, f32, "p"), _ptr(loss_vec, f32, "loss_vec"),
              _ptr(gz, f32, "gz"), _stream())
    return p, reduce_sum(loss_vec, 1.0 / n), gz


CTR_HEAD_ONE_WG_MAX = 65536   # rows the one-workgroup head (rc_ctr_head_fwd_bwd_sums) takes


def ctr_head_sums(bias, lin, term1, term2, label, full=False):
    """ctr_head in one workgroup that also forms the loss mean and sum gz: -> (p [n], sums [2] = (loss, sum gz), gz [n]).
    full=True (rc_ctr_head_fwd_full): the same launch leaves the backward fan-out for a seed gradient of exactly one --
    -> (p, sums, gz, g_lin [n, F], g_bias [1]) -- and takes a pending deferred counter increment along (Adam's step count)"""
    n, F = lin.shape
    f32 = torch.float32
    p = torch.empty(n, dtype=f32, device=lin.device)
    loss_vec = torch.empty(n, dtype=f32, device=lin.device)
    gz = torch.empty(n, dtype=f32, device=lin.device)
    sums = torch.empty(2, dtype=f32, device=lin.device)
    if full:
        g_lin = torch.empty((n, F), dtype=f32, device=lin.device)
        g_bias = torch.empty(1, dtype=f32, device=lin.device)
        bump = pending_deferred()
        if bump is not None and bump.device != lin.device:
            bump = None
        _lib.call("rc_ctr_head_fwd_full", _ptr(bias, f32, "bias"), _ptr(lin, f32, "lin"), int(F), _ptr(term1, f32, "term1", True),
                  _ptr(term2, f32, "term2", True), _ptr(label, torch.int64, "label"), n, _ptr(p, f32, "p"), _ptr(loss_vec, f32, "loss_vec"),
                  _ptr(gz, f32, "gz"), _ptr(sums, f32, "sums"), _ptr(g_lin, f32, "g_lin"), _ptr(g_bias, f32, "g_bias"),
                  _ptr(bump, torch.int64, "bump", True), _stream())
        if bump is not None:
            fold_deferred(bump)
        return p, sums, gz, g_lin, g_bias
    _lib.call("rc_ctr_head_fwd_bwd_sums", _ptr(bias, f32, "bias"), _ptr(lin, f32, "lin"), int(F), _ptr(term1, f32, "term1", True),
              _ptr(term2, f32, "term2", True), _ptr(label, torch.int64, "label"), n, _ptr(p, f32, "p"), _ptr(loss_vec, f32, "loss_vec"),
              _ptr(gz, f32, "gz"), _ptr(sums, f32, "sums"), _stream())
    return p, sums, gz


def ctr_head_bwd(gz, sums, g_loss, F):
    """backward fan-out of the head: -> (g [n] = gz * g_loss, g_lin [n, F] (contiguous), g_bias [1])"""
    n = gz.shape[0]
    f32 = torch.float32
    g = torch.empty(n, dtype=f32, device=gz.device)
    g_lin = torch.empty((n, F), dtype=f32, device=gz.device)
    g_bias = torch.empty(1, dtype=f32, device=gz.device)
    _lib.call("rc_ctr_head_bwd", _ptr(gz, f32, "gz"), _ptr(sums, f32, "sums"), _ptr(g_loss, f32, "g_loss"), n, int(F), _ptr(g, f32, "g"),
              _ptr(g_lin, f32, "g_lin"), _ptr(g_bias, f32, "g_bias"), _stream())
    return g, g_lin, g_bias


# ---- batch assembly on the device (csrc/sampler.hip) -------------------------------------------------------

def sample_negatives(users, K, n_items, clicked_ptr=None, clicked_items=None, seed=0, base_index=0, out=None):
    """neg [n, K] int64: uniform over [1, n_items) minus the user's clicked set (models/BaseModel.py:206-214)"""
    n = users.numel()
    i64 = torch.int64
    neg = out if out is not None else torch.empty((n, K), dtype=i64, device=users.device)
    _lib.call("rc_sample_negatives", _ptr(users, i64, "users"), n, int(K), int(n_items),
              _ptr(clicked_ptr, i64, "clicked_ptr", True), _ptr(clicked_items, i64, "clicked_items", True),
              C.c_uint64(int(seed) & (2**64 - 1)), C.c_uint64(int(base_index)), _ptr(neg, i64, "neg"), _stream())
    return neg


def assemble_candidates(idx, users, items, neg):
    """(user_id [B], item_id [B, 1+K]) for the rows idx of a training set (models/BaseModel.py:192-203)"""
    B = idx.numel()
    K = 0 if neg is None else neg.shape[1]
    i64 = torch.int64
    u = torch.empty(B, dtype=i64, device=idx.device)
    cand = torch.empty((B, 1 + K), dtype=i64, device=idx.device)
    _lib.call("rc_assemble_candidates", _ptr(idx, i64, "idx"), B, K, _ptr(users, i64, "users"), _ptr(items, i64, "items"),
              _ptr(neg, i64, "neg", True), _ptr(u, i64, "users_out"), _ptr(cand, i64, "cand_out"), _stream())
    return u, cand


def gather_history(idx, users, position, his_ptr, his_items, L, his_times=None):
    """-> (history_items [B, L] right-padded with 0, history_times | None, lengths [B])  (models/BaseModel.py:236-245)"""
    B = idx.numel() if idx is not None else users.numel()
    i64 = torch.int64
    dev = users.device
    hist = torch.empty((B, L), dtype=i64, device=dev)
    times = torch.empty((B, L), dtype=i64, device=dev) if his_times is not None else None
    lengths = torch.empty(B, dtype=i64, device=dev)
    _lib.call("rc_gather_history", _ptr(idx, i64, "idx", True), B, int(L), _ptr(users, i64, "users"),
              _ptr(position, i64, "position"), _ptr(his_ptr, i64, "his_ptr"), _ptr(his_items, i64, "his_items"),
              _ptr(his_times, i64, "his_times", True), _ptr(hist, i64, "hist"), _ptr(times, i64, "times", True),
              _ptr(lengths, i64, "lengths"), _stream())
    return hist, times, lengths


# ---- evaluation (csrc/eval_rank.hip) ---------------------------------------------------------------------------

def target_rank(pred):
    """rank of column 0 among the row's candidates, ties against it (helpers/BaseRunner.py:62-63) -> int32 [n]"""
    n, Cn = pred.shape
    rank = torch.empty(n, dtype=torch.int32, device=pred.device)
    _lib.call("rc_target_rank", _ptr(pred, torch.float32, "pred"), n, Cn, _ptr(rank, torch.int32, "rank"), _stream())
    return rank


def full_catalogue_rank_supported(d):
    return bool(_lib.load().rc_full_catalogue_rank_supported(int(d)))


def full_catalogue_rank(Uvec, I, users, targets, clicked_ptr=None, clicked_items=None):
    """--test_all rank of the target among ALL items, clicked items masked -> (rank int32 [N], target_score [N])"""
    N, d = Uvec.shape
    i64, f32 = torch.int64, torch.float32
    rank = torch.empty(N, dtype=torch.int32, device=Uvec.device)
    tscore = torch.empty(N, dtype=f32, device=Uvec.device)
    _lib.call("rc_full_catalogue_rank", _ptr(Uvec, f32, "Uvec"), _ptr(I, f32, "I"), _ptr(users, i64, "users", True),
              _ptr(targets, i64, "targets"), N, I.shape[0], d, _ptr(clicked_ptr, i64, "clicked_ptr", True),
              _ptr(clicked_items, i64, "clicked_items", True), _ptr(tscore, f32, "target_score"),
              _ptr(rank, torch.int32, "rank"), _stream())
    return rank, tscore


def rank_metrics(rank, topk, metrics):
    """HR@k / NDCG@k from gt ranks (helpers/BaseRunner.py:64-76); a few scalar reductions, one D2H copy"""
    rank = rank.to(torch.float64)
    keys, vals = [], []
    for k in topk:
        hit = (rank <= k).to(torch.float64)
        for metric in metrics:
            if metric == "HR":
                vals.append(hit.mean())
            elif metric == "NDCG":
                vals.append((hit / torch.log2(rank + 1)).mean())
            else:
                raise ValueError("Undefined evaluation metric: {}.".format(metric))
            keys.append("{}@{}".format(metric, k))
    out = torch.stack(vals).cpu().numpy() if vals else []
    return {k: float(v) for k, v in zip(keys, out)}


def list_metrics_supported(n, max_pos, n_k):
    return bool(_lib.load().rc_list_metrics_supported(int(n), int(max_pos), int(n_k)))


def list_metrics(pred, pos_num, neg_num, max_pos, topk, want_mean=True):
    """ImpressionRunner's HR / NDCG / MAP @k (helpers/ImpressionRunner.py:18-66,74-133) on the device:
    pred [N, n] fp32, pos_num [N] int64 | None, neg_num [N] int64 -> (per_row float64 [N, 3, K], mean float64 [3, K] | None),
    metric order NDCG, MAP, HR"""
    N, n = pred.shape
    K = len(topk)
    dev, f64, i64 = pred.device, torch.float64, torch.int64
    per_row = torch.empty((N, 3, K), dtype=f64, device=dev)
    mean = torch.empty((3, K), dtype=f64, device=dev) if want_mean else None
    ks = (C.c_int * K)(*[int(k) for k in topk])
    _lib.call("rc_list_metrics", _ptr(pred, torch.float32, "pred") if N else None, _ptr(pos_num, i64, "pos_num", True),
              _ptr(neg_num, i64, "neg_num") if N else None, N, int(n), int(max_pos), ks, K, _ptr(per_row, f64, "per_row") if N else None,
              _ptr(mean, f64, "mean", True), _stream())
    return per_row, mean


# ---- row-sharded step: local kernels (csrc/owner_step.hip) --------------------------------------------------

def route_by_owner(ids, world, tuple_base=None, div=1):
    """stable grouping of ids by id % world -> (order int32 [n], counts int64 [world] on the device, payload
    int64 [n]: local rows id // world, or (tuple_base + position // div) << 32 | local row)"""
    ids = ids.reshape(-1)
    n = ids.numel()
    dev, i64 = ids.device, torch.int64
    order = torch.empty(n, dtype=torch.int32, device=dev)
    payload = torch.empty(n, dtype=i64, device=dev)
    counts = torch.empty(world, dtype=i64, device=dev)
    lib = _lib.load()
    ws = workspace(lib.rc_route_workspace_bytes(n, world), dev, "route")
    packed = tuple_base is not None
    _lib.call("rc_route_by_owner", _ptr(ids, i64, "ids"), n, int(world), int(tuple_base or 0), int(div),
              _ptr(order, torch.int32, "order"), _ptr(payload if packed else None, i64, "packed", True),
              _ptr(None if packed else payload, i64, "local_row", True), _ptr(counts, i64, "counts"),
              C.c_void_p(ws.data_ptr()), ws.numel(), _stream())
    return order, counts, payload


def owner_unpack(recv):
    """(tuple << 32 | row) messages -> (t_idx int64, rows int64, t32 int32-typed uint32)"""
    n = recv.numel()
    dev, i64 = recv.device, torch.int64
    t_idx, rows = torch.empty(n, dtype=i64, device=dev), torch.empty(n, dtype=i64, device=dev)
    t32 = torch.empty(n, dtype=torch.int32, device=dev)
    _lib.call("rc_owner_unpack", _ptr(recv, i64, "packed"), n, _ptr(t_idx, i64, "t_idx"), _ptr(rows, i64, "rows"),
              _ptr(t32, torch.int32, "t32"), _stream())
    return t_idx, rows, t32


def owner_backward_supported(d):
    return int(d) in (16, 32, 64, 128, 256)


def owner_backward(I, mI, vI, Uall, t32, rows, g, single, n_tuples, hyper):
    """pug [n_tuples, d] = per-tuple sum of g * I[row]; rows flagged in `single` are updated in place"""
    n = rows.numel()
    d = I.shape[1]
    f32 = torch.float32
    pug = torch.empty((n_tuples, d), dtype=f32, device=I.device)
    lib = _lib.load()
    ws = workspace(lib.rc_owner_backward_workspace_bytes(n), I.device, "owner")
    _lib.call("rc_owner_backward", _ptr(I, f32, "I"), _ptr(mI, f32, "mI", True), _ptr(vI, f32, "vI", True), d,
              _ptr(Uall, f32, "Uall"), _ptr(t32, torch.int32, "t32"), _ptr(rows, torch.int64, "rows"),
              _ptr(g, f32, "g"), _ptr(single, torch.uint8, "single", True), n, int(n_tuples),
              C.byref(hyper) if hyper is not None else None, _ptr(pug, f32, "pug"),
              C.c_void_p(ws.data_ptr()), ws.numel(), _stream())
    return pug
