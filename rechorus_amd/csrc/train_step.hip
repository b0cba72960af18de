// train_step.hip -- one BaseRunner.fit iteration for BPRMF as a single C-ABI call
// (reference: helpers/BaseRunner.py:193-206 around models/general/BPRMF.py:34-45 and
// models/BaseModel.py:182-185).  All phases are enqueued on the caller's stream with no
// host synchronisation (grid sizes depend only on B, C), so the call can be captured in a
// hipGraph; with `phase_ms` it brackets the phases with hipEvents and synchronises.
//
// Three pipelines (rc_bprmf_step_pipeline):
//   small batches (<= 32,768 row ids)   two launches: small_step.hip
//   bucket plan (default otherwise)     partition -> [flags] -> fused kernel || per-bucket pass on a second stream
//                                       -> row updates; with rc_bprmf_train_step_ahead the front of the NEXT batch runs
//                                       beside this step's row updates
//   sort pipeline (wide id spaces)      joint radix sort -> segment heads -> fused kernel -> segmented updates
// phase_ms slots (all pipelines): [0] sort / partition (+ flags) [7] segment heads / per-bucket pass when on the caller's
// stream [2] fused kernel [3] loss mean (0 when folded into the last update launch) [4] item-row update [5] user-row update.
//
// Ordering constraints: the item-row update reads U (pre-step values, to rebuild g*U[u]) so it runs
// before the user rows are rewritten; the fused kernel reads both tables before either is updated.
#include "common.hpp"
#include "plan.hpp"

namespace rc {
int plan_bprmf_step_updates(float* U, float* mU, float* vU, float* I, float* mI, float* vI, int d, const int64_t* uid,
                            int C, int64_t n_i, int64_t B, const float* gpred, const float* ugrad,
                            const rc_plan_row* rows_i, const uint32_t* n_rows_i, const rc_plan_row* rows_u,
                            const uint32_t* n_rows_u, const uint32_t* occ, uint32_t* counters,
                            const PlanLongWs& lw,
                            const rc_opt_hyper* h, const float* loss_vec, float loss_scale, float* loss_out,
                            hipStream_t s, hipEvent_t* ev_items_done);
int plan_prepare();
// small batches (small_step.hip): two launches
size_t small_step_extra_bytes(int64_t n, int64_t B, int d);
bool small_step_supported(int64_t n_i, int64_t B, int64_t n_items, int64_t n_users, int d);
int small_step_launch(float* U, float* I, float* mU, float* vU, float* mI, float* vI, const int64_t* uid, const int64_t* iid,
                      int B, int C, int d, int64_t n_items, const rc_opt_hyper* h, float inv_b, float* loss_out, float* pred,
                      float* gpred, float* ugrad, float* loss_vec, void* extra, hipStream_t s, hipEvent_t* ev_mid);
}

// The second stream of the step.  The bucket plan's per-bucket pass (row records + grouped positions) is index work
// that only the updates need; the fused kernel needs at most the singleton flags.  So the step forks: the fused
// kernel runs on the caller's stream while plan_launch_back runs on this side stream, and the updates wait for both.
// Created once per process (non-blocking: the caller's stream may be the legacy null stream); fork / join are
// event dependencies, so the call stays capturable in a hipGraph.  RC_BPRMF_STEP=serial keeps everything on one stream.
namespace {
struct StepSide {
  hipStream_t stream = nullptr;
  hipEvent_t fork = nullptr, join = nullptr, fork2 = nullptr, front_done = nullptr;
  bool ok = false;
};
StepSide& step_side() {
  static StepSide sd = [] {
    StepSide x;
    if (hipStreamCreateWithFlags(&x.stream, hipStreamNonBlocking) == hipSuccess &&
        hipEventCreateWithFlags(&x.fork, hipEventDisableTiming) == hipSuccess &&
        hipEventCreateWithFlags(&x.join, hipEventDisableTiming) == hipSuccess &&
        hipEventCreateWithFlags(&x.fork2, hipEventDisableTiming) == hipSuccess &&
        hipEventCreateWithFlags(&x.front_done, hipEventDisableTiming) == hipSuccess)
      x.ok = true;
    return x;
  }();
  return sd;
}

// Look-ahead across steps (rc_bprmf_train_step_ahead): the plan's FRONT of the next batch (histogram, partition,
// singleton flags: 0.155 ms on the critical path of the SGD step at config 2) runs on the side stream beside THIS step's
// row updates, so that the next call starts with its fused kernel.  What was prepared, for which batch and workspace:
struct Ahead {
  bool valid = false;
  const void* ws = nullptr;
  const int64_t* uid = nullptr;
  const int64_t* iid = nullptr;
  int B = 0, C = 0, d = 0, slot = 0;
  int64_t n_users = 0, n_items = 0;
};
Ahead& step_ahead() {
  static Ahead a;
  return a;
}
}  // namespace

using namespace rc;

namespace {
struct StepWs {
  uint32_t* keys_i;
  uint32_t* perm_i;
  uint32_t* keys_u;
  uint32_t* perm_u;
  float* gpred;
  float* ugrad;
  float* loss_vec;
  uint8_t* single;
  uint32_t* heads_i;
  uint32_t* n_heads_i;
  void* sort_ws;
  size_t sort_ws_bytes;
  void* seg_ws;
  size_t seg_ws_bytes;
  // bucket-plan step (the default where the id space allows it)
  PlanWs plan;
  PlanLongWs plan_long;
  rc_plan_row* rows_i;
  rc_plan_row* rows_u;
  uint32_t* occ;
  void* small_extra;     // small-batch step (small_step.hip): user-row snapshot, per-workgroup row / position segments
  uint32_t* counters2;   // second counter block: the look-ahead front of the next batch zeroes / fills the one this step does not use
  size_t total;
};

StepWs carve_step_ws(void* base, int B, int C, int d) {
  const size_t n_i = (size_t)B * C;
  Carver cv(base);
  StepWs w;
  // item and user ids are sorted together: [0, n_i) is the item segment, [n_i, n_i+B) the users'
  w.keys_i = cv.take<uint32_t>(n_i + (size_t)B);
  w.perm_i = cv.take<uint32_t>(n_i + (size_t)B);
  w.keys_u = w.keys_i + n_i;
  w.perm_u = w.perm_i + n_i;
  w.gpred = cv.take<float>(n_i);
  w.ugrad = cv.take<float>((size_t)B * d);
  w.loss_vec = cv.take<float>((size_t)B);
  const size_t plan_off = cv.off;
  w.single = cv.take<uint8_t>(n_i);
  w.heads_i = cv.take<uint32_t>(n_i);
  w.n_heads_i = cv.take<uint32_t>(1);
  w.sort_ws_bytes = rc_sort_workspace_bytes((int64_t)n_i + B);
  w.sort_ws = cv.take<char>(w.sort_ws_bytes);
  w.seg_ws_bytes = rc_segmented_workspace_bytes((int64_t)n_i, d);
  w.seg_ws = cv.take<char>(w.seg_ws_bytes);
  // the two pipelines never run in the same call: the plan buffers overlay the sort / segment scratch
  // (everything after loss_vec)
  Carver pv(base);
  pv.off = plan_off;
  w.single = pv.take<uint8_t>(align_up(n_i, (size_t)kPlanTile));
  {
    const PlanWs pw = carve_plan_ws(base ? reinterpret_cast<char*>(base) + pv.off : nullptr, (int64_t)n_i + B);
    w.plan = pw;
    pv.off += align_up(pw.total, 256);
    const PlanLongWs lw = carve_plan_long_ws(base ? reinterpret_cast<char*>(base) + pv.off : nullptr, (int64_t)n_i + B, d);
    w.plan_long = lw;
    pv.off += align_up(lw.total, 256);
  }
  w.rows_i = pv.take<rc_plan_row>(n_i);
  w.rows_u = pv.take<rc_plan_row>((size_t)B);
  w.occ = pv.take<uint32_t>(n_i + (size_t)B);
  w.counters2 = pv.take<uint32_t>(PC_N);
  w.total = cv.off > pv.off ? cv.off : pv.off;
  // the small-batch step's buffers overlay the same region (the pipelines never run in the same call)
  w.small_extra = base ? reinterpret_cast<char*>(base) + plan_off : nullptr;
  if ((int64_t)n_i + B <= 32768) {
    const size_t need = plan_off + small_step_extra_bytes((int64_t)n_i + B, B, d);
    if (need > w.total) w.total = need;
  }
  return w;
}
}  // namespace

// 0 = automatic (two-launch step for small batches; else the bucket plan, its per-bucket pass on a second stream behind
// the fused kernel), 1 = always the sort pipeline, 2 = bucket plan on ONE stream, 3 = bucket plan on two streams whatever
// the batch size; initial value from RC_BPRMF_STEP=sort|serial|plan
static int& step_pipeline() {
  static int mode = [] {
    const char* v = getenv("RC_BPRMF_STEP");
    return (v && strcmp(v, "sort") == 0) ? 1 : ((v && strcmp(v, "serial") == 0) ? 2 : ((v && strcmp(v, "plan") == 0) ? 3 : 0));
  }();
  return mode;
}

extern "C" int rc_bprmf_step_pipeline(int mode) {
  const int prev = step_pipeline();
  if (mode >= 0 && mode <= 3) step_pipeline() = mode;
  return prev;
}

extern "C" size_t rc_bprmf_step_workspace_bytes(int B, int C, int d) {
  if (B < 1 || C < 1 || d < 1) return 0;
  return carve_step_ws(nullptr, B, C, d).total;
}

static int train_step_impl(float* U, float* I, float* mU, float* vU, float* mI, float* vI,
                           const int64_t* uid, const int64_t* iid, int B, int C, int d,
                           int64_t n_users, int64_t n_items, const rc_opt_hyper* h,
                           float inv_b, float* loss_out, float* pred, void* ws,
                           size_t ws_bytes, rc_stream_t stream, float* phase_ms,
                           const int64_t* next_uid, const int64_t* next_iid) {
  RC_REQUIRE(U && I && uid && iid && h && loss_out && ws, "rc_bprmf_train_step: null pointer");
  RC_REQUIRE(B >= 1 && C >= 2 && d >= 1, "rc_bprmf_train_step: bad shape B=%d C=%d d=%d", B, C, d);
  RC_REQUIRE((int64_t)B * C < ((int64_t)1 << 31), "rc_bprmf_train_step: B*C too large");
  RC_REQUIRE(U != I, "rc_bprmf_train_step: user and item tables must be distinct");
  const StepWs w = carve_step_ws(ws, B, C, d);
  if (ws_bytes < w.total)
    return fail(RC_ERR_WORKSPACE, "rc_bprmf_train_step: workspace %zu < %zu", ws_bytes, w.total);
  hipStream_t s = as_stream(stream);
  const int64_t n_i = (int64_t)B * C;
  // a front prepared ahead by the previous call: usable when it was made for exactly this batch and workspace;
  // in every case its kernels (side stream) have to be finished before this call touches the plan buffers
  Ahead& ahead = step_ahead();
  const bool ahead_hit = ahead.valid && ahead.ws == ws && ahead.uid == uid && ahead.iid == iid && ahead.B == B && ahead.C == C &&
                         ahead.d == d && ahead.n_users == n_users && ahead.n_items == n_items;
  const int slot = ahead_hit ? ahead.slot : 0;
  if (ahead.valid) RC_HIP(hipStreamWaitEvent(s, step_side().front_done, 0));
  ahead.valid = false;

  constexpr int kMarks = 8;
  hipEvent_t ev[kMarks];
  const bool prof = phase_ms != nullptr;
  if (prof)
    for (int i = 0; i < kMarks; ++i) RC_HIP(hipEventCreate(&ev[i]));
#define RC_MARK(i)                                \
  do {                                            \
    if (prof) RC_HIP(hipEventRecord(ev[i], s));   \
  } while (0)

  // The singleton fast path (update single-occurrence item rows inside the fused kernel) pays for
  // SGD only: with optimizer state the m/v rows have to be fetched at the kernel's tail, where nothing
  // hides their latency (measured at config 2, Adam: 2.98 ms/step fused vs 2.31 ms through the
  // segmented update, which already streams 6 row-units per touched row at the HBM rate).
#if defined(RC_FUSED_UPD_NEVER)
  const bool fused_upd = false;
#elif defined(RC_FUSED_UPD_ALWAYS)
  const bool fused_upd = rc_bprmf_fused_supported(d, C) != 0;
#else
  const bool fused_upd = rc_bprmf_fused_supported(d, C) != 0 && h->opt == RC_OPT_SGD;
#endif
  // Pipeline choice: the bucket plan (bucket_plan.hip + plan_update.hip, 8 launches) where the register-resident
  // fused kernel exists and the joint id space fits one bucket level; otherwise (and with RC_BPRMF_STEP=sort)
  // the round-1 pipeline: joint radix sort -> segment heads -> fused -> segmented updates.
  const bool force_sort = step_pipeline() == 1;
  const PlanGeom geom = plan_geometry(n_i, B, n_items, n_users);
  const bool fused_ok = rc_bprmf_fused_supported(d, C) != 0;
  // Small batches (<= 32,768 row ids, e.g. the reference's default B = 256 with K = 99): two launches (small_step.hip).
  if (step_pipeline() == 0 && fused_ok && small_step_supported(n_i, B, n_items, n_users, d) &&
      reinterpret_cast<uintptr_t>(U) % 16 == 0 && reinterpret_cast<uintptr_t>(I) % 16 == 0) {
    RC_MARK(0);
    RC_MARK(1);
    RC_MARK(2);
    RC_MARK(3);
    RC_TRY(small_step_launch(U, I, mU, vU, mI, vI, uid, iid, B, C, d, n_items, h, inv_b, loss_out, pred, w.gpred, w.ugrad,
                             w.loss_vec, w.small_extra, s, prof ? &ev[4] : nullptr));  // marks 4, 5
    RC_MARK(6);
    RC_MARK(7);
  } else
  if (!force_sort && geom.ok && fused_ok && (d == 16 || d == 32 || d == 64 || d == 128)) {
    RC_TRY(plan_prepare());
    PlanArgs pa;
    memset(&pa, 0, sizeof(pa));
    pa.ids_a = iid; pa.ids_b = uid; pa.n_a = (uint32_t)n_i; pa.n = (uint32_t)(n_i + B);
    pa.range_a = n_items; pa.range_b = n_users;
    pa.g = geom;
    pa.w = w.plan;
    uint32_t* ctr = slot ? w.counters2 : w.plan.counters;   // (the other block belongs to a look-ahead front)
    pa.w.counters = ctr;
    pa.list_single_a = fused_upd ? 0 : 1;
    pa.single_a = fused_upd ? w.single : nullptr;
    pa.rows_a = w.rows_i; pa.rows_b = w.rows_u;
    pa.n_rows_a = &ctr[PC_ROWS_A]; pa.n_rows_b = &ctr[PC_ROWS_B];
    pa.occ = w.occ;
    StepSide& side = step_side();
    const bool two_streams = (step_pipeline() == 0 || step_pipeline() == 3) && side.ok;
    RC_MARK(0);
    if (two_streams) {
      // caller's stream: partition (+ singleton flags when the fused kernel updates them) -> fused kernel
      // side stream:     per-bucket pass (row records, grouped positions), joined before the updates
      // (without the singleton fast path the fused kernel needs nothing from the plan: all of it runs on the side stream)
      pa.flags_done = fused_upd ? 1 : 0;
      if (fused_upd && !ahead_hit) RC_TRY(plan_launch_front(pa, true, s));   // (ahead_hit: done beside the previous step's updates)
      RC_MARK(1);
      RC_HIP(hipEventRecord(side.fork, s));
      RC_HIP(hipStreamWaitEvent(side.stream, side.fork, 0));
      if (!fused_upd) RC_TRY(plan_launch_front(pa, false, side.stream));
      RC_TRY(plan_launch_back(pa, side.stream));
      RC_HIP(hipEventRecord(side.join, side.stream));
    } else {
      RC_TRY(plan_launch(pa, s, prof ? &ev[1] : nullptr));   // ev[1]: after the partition, before the bucket kernel
    }
    RC_MARK(2);
    RC_MARK(3);
    if (fused_upd)
      RC_TRY(rc_bprmf_fwd_bwd_update(U, I, mI, vI, uid, iid, w.single, B, C, d, inv_b, h, pred, w.loss_vec, w.gpred,
                                     w.ugrad, stream));
    else
      RC_TRY(rc_bprmf_fwd_bwd(U, I, uid, iid, B, C, d, inv_b, pred, w.loss_vec, w.gpred, w.ugrad, stream));
    if (two_streams) RC_HIP(hipStreamWaitEvent(s, side.join, 0));
    if (two_streams && fused_upd && next_uid != nullptr && next_iid != nullptr && !prof) {
      // Look-ahead: the front of the NEXT batch on the side stream, beside this step's row updates.  It writes the
      // histogram / bucketed keys / flags (all consumed by this step already: the fused kernel and the per-bucket pass are
      // behind the join above) and zeroes the OTHER counter block.  Not under stream capture (the next call's wait
      // on front_done would cross graphs).
      hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
      if (hipStreamIsCapturing(s, &cap) == hipSuccess && cap == hipStreamCaptureStatusNone) {
        PlanArgs pn = pa;
        pn.ids_a = next_iid; pn.ids_b = next_uid;
        uint32_t* nctr = slot ? w.plan.counters : w.counters2;
        pn.w.counters = nctr;
        pn.n_rows_a = &nctr[PC_ROWS_A]; pn.n_rows_b = &nctr[PC_ROWS_B];
        RC_HIP(hipEventRecord(side.fork2, s));
        RC_HIP(hipStreamWaitEvent(side.stream, side.fork2, 0));
        RC_TRY(plan_launch_front(pn, true, side.stream));
        RC_HIP(hipEventRecord(side.front_done, side.stream));
        ahead.valid = true; ahead.ws = ws; ahead.uid = next_uid; ahead.iid = next_iid; ahead.B = B; ahead.C = C; ahead.d = d;
        ahead.n_users = n_users; ahead.n_items = n_items; ahead.slot = 1 - slot;
      }
    }
    RC_MARK(4);
    RC_MARK(5);  // (the loss mean is one workgroup of the last update launch)
    RC_TRY(plan_bprmf_step_updates(U, mU, vU, I, mI, vI, d, uid, C, n_i, B, w.gpred, w.ugrad, w.rows_i, pa.n_rows_a,
                                   w.rows_u, pa.n_rows_b, w.occ, ctr, w.plan_long, h, w.loss_vec, inv_b,
                                   loss_out, s, prof ? &ev[6] : nullptr));
    RC_MARK(7);
  } else {
  RC_MARK(0);
  // one joint radix sort: keys = item id | n_items + user id (all user keys sort after all item keys)
  RC_REQUIRE(n_items + n_users <= ((int64_t)1 << 32), "rc_bprmf_train_step: n_items + n_users exceeds 2^32");
  RC_TRY(rc_sort_ids2(iid, n_i, uid, B, n_items, n_items + n_users, w.keys_i, w.perm_i, w.sort_ws,
                      w.sort_ws_bytes, stream));
  RC_MARK(1);
  if (fused_upd)
    RC_TRY(rc_segment_heads(w.keys_i, w.perm_i, n_i, 1, w.single, w.heads_i, w.n_heads_i, stream));
  RC_MARK(2);
  RC_MARK(3);  // (the user ids were sorted with the item ids)
  if (fused_upd)
    RC_TRY(rc_bprmf_fwd_bwd_update(U, I, mI, vI, uid, iid, w.single, B, C, d, inv_b, h, pred,
                                   w.loss_vec, w.gpred, w.ugrad, stream));
  else
    RC_TRY(rc_bprmf_fwd_bwd(U, I, uid, iid, B, C, d, inv_b, pred, w.loss_vec, w.gpred, w.ugrad,
                            stream));
  RC_MARK(4);
  RC_TRY(rc_reduce_sum(w.loss_vec, B, inv_b, loss_out, stream));
  RC_MARK(5);
  // item rows: grad_r = sum_{(b,c): iid[b,c]=r} g[b,c] * U[uid[b]]
  RC_TRY(rc_segmented_update(I, mI, vI, d, w.keys_i, w.perm_i, n_i, w.gpred, U, uid, C, h,
                             nullptr, fused_upd ? w.heads_i : nullptr,
                             fused_upd ? w.n_heads_i : nullptr,
                             fused_upd ? RC_SEG_SKIP_SINGLETONS : 0, w.seg_ws, w.seg_ws_bytes,
                             stream));
  RC_MARK(6);
  // user rows: grad_r = sum_{b: uid[b]=r} ugrad[b]
  RC_TRY(rc_segmented_update2(U, mU, vU, d, w.keys_u, w.perm_u, B, nullptr, w.ugrad, nullptr, 1, nullptr, B,
                              /*key_base=*/n_items, /*occ_base=*/n_i, h, nullptr, nullptr, nullptr, 0,
                              w.seg_ws, w.seg_ws_bytes, stream));
  RC_MARK(7);
  }
#undef RC_MARK

  if (prof) {
    RC_HIP(hipEventSynchronize(ev[kMarks - 1]));
    // event i opens: 0 sort items, 1 mark singletons, 2 sort users, 3 fused, 4 loss mean,
    // 5 item update, 6 user update; reported in the header's slot order
    const int slot[7] = {0, 7, 1, 2, 3, 4, 5};
    for (int i = 0; i < 7; ++i) RC_HIP(hipEventElapsedTime(&phase_ms[slot[i]], ev[i], ev[i + 1]));
    RC_HIP(hipEventElapsedTime(&phase_ms[6], ev[0], ev[kMarks - 1]));
    for (int i = 0; i < kMarks; ++i) RC_HIP(hipEventDestroy(ev[i]));
  }
  return RC_OK;
}

extern "C" int rc_bprmf_train_step(float* U, float* I, float* mU, float* vU, float* mI, float* vI,
                                   const int64_t* uid, const int64_t* iid, int B, int C, int d,
                                   int64_t n_users, int64_t n_items, const rc_opt_hyper* h,
                                   float inv_b, float* loss_out, float* pred, void* ws,
                                   size_t ws_bytes, rc_stream_t stream, float* phase_ms) {
  return train_step_impl(U, I, mU, vU, mI, vI, uid, iid, B, C, d, n_users, n_items, h, inv_b, loss_out, pred, ws, ws_bytes, stream,
                         phase_ms, nullptr, nullptr);
}

extern "C" int rc_bprmf_train_step_ahead(float* U, float* I, float* mU, float* vU, float* mI, float* vI,
                                         const int64_t* uid, const int64_t* iid, const int64_t* next_uid,
                                         const int64_t* next_iid, int B, int C, int d, int64_t n_users, int64_t n_items,
                                         const rc_opt_hyper* h, float inv_b, float* loss_out, float* pred, void* ws,
                                         size_t ws_bytes, rc_stream_t stream) {
  return train_step_impl(U, I, mU, vU, mI, vI, uid, iid, B, C, d, n_users, n_items, h, inv_b, loss_out, pred, ws, ws_bytes, stream,
                         nullptr, next_uid, next_iid);
}

// Forget a prepared front (the owner of the workspace goes away or re-allocates it): `stream` is made to wait for the
// side stream's writes into that workspace, so that whatever reuses the memory afterwards is ordered behind them.
extern "C" int rc_bprmf_step_ahead_reset(rc_stream_t stream) {
  Ahead& ahead = step_ahead();
  if (ahead.valid) {
    RC_HIP(hipStreamWaitEvent(as_stream(stream), step_side().front_done, 0));
    ahead.valid = false;
  }
  return RC_OK;
}
