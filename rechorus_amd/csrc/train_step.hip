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
#include <mutex>

#include "common.hpp"
#include "plan.hpp"

namespace rc {
int plan_bprmf_step_updates(float* U, float* mU, float* vU, float* I, float* mI, float* vI, int d, const int64_t* uid,
                            int C, int64_t n_i, int64_t B, const float* gpred, const float* ugrad,
                            const rc_plan_row* rows_i, const uint32_t* n_rows_i, const rc_plan_row* rows_u,
                            const uint32_t* n_rows_u, const uint32_t* occ, uint32_t* counters,
                            const PlanLongWs& lw, bool long_planned,
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
// that only the updates need; the fused kernel needs at most the multi-occurrence bitmap.  So the step forks: the fused
// kernel runs on the caller's stream while plan_launch_back runs on this side stream, and the updates wait for both.
// One record per device, created on first use (non-blocking stream: the caller's stream may be the legacy null
// stream); fork / join are event dependencies, so the call stays capturable in a hipGraph.  RC_BPRMF_STEP=serial keeps
// everything on one stream.  These are resources (a stream, four events), not batch state: what was prepared for which
// batch lives in the CALLER's rc_step_ticket.
namespace {
struct StepSide {
  hipStream_t stream = nullptr;
  hipEvent_t fork = nullptr, join = nullptr, fork2 = nullptr, front_done = nullptr;
  bool tried = false, ok = false;
};
constexpr int kMaxDevices = 64;
StepSide* step_side(int* device_out = nullptr) {
  static StepSide sides[kMaxDevices];
  static std::mutex mu;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return nullptr;
  if (device_out) *device_out = dev;
  std::lock_guard<std::mutex> lock(mu);
  StepSide& x = sides[dev];
  if (!x.tried) {
    x.tried = true;
    // RC_SIDE_PRIO=high: the side stream's workgroups are dispatched ahead of the caller's whenever a CU has room
    // (the plan kernels are latency-bound index work that has to squeeze in beside bandwidth-bound row kernels)
    const char* pr = getenv("RC_SIDE_PRIO");
    bool made = false;
    if (pr && strcmp(pr, "high") == 0) {
      int least = 0, greatest = 0;
      if (hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess)
        made = hipStreamCreateWithPriority(&x.stream, hipStreamNonBlocking, greatest) == hipSuccess;
    }
    x.ok = (made || hipStreamCreateWithFlags(&x.stream, hipStreamNonBlocking) == hipSuccess) &&
           hipEventCreateWithFlags(&x.fork, hipEventDisableTiming) == hipSuccess &&
           hipEventCreateWithFlags(&x.join, hipEventDisableTiming) == hipSuccess &&
           hipEventCreateWithFlags(&x.fork2, hipEventDisableTiming) == hipSuccess &&
           hipEventCreateWithFlags(&x.front_done, hipEventDisableTiming) == hipSuccess;
  }
  return x.ok ? &x : nullptr;
}
// where the look-ahead plan of the next batch is forked off: 1 = at the start of the step (default: the plan's chain of
// latency-bound kernels stretches about 2.5 x beside the bandwidth-bound row kernels and needs the whole step as its
// window -- 0.988 -> 0.945 ms/step at config 2, profiles/r03d_ab_overlap.txt), 0 = behind the fused kernel (beside this
// step's row updates only); RC_AHEAD_FORK=late selects 0
bool ahead_bitmap_in_bucket() {   // RC_AHEAD_BITMAP=separate: the bitmap kernel of the front for the look-ahead plan as well (A/B)
  static int on = [] {
    const char* v = getenv("RC_AHEAD_BITMAP");
    return (v && strcmp(v, "separate") == 0) ? 0 : 1;
  }();
  return on != 0;
}
int ahead_fork_mode() {
  static int mode = [] {
    const char* v = getenv("RC_AHEAD_FORK");
    return (v && strcmp(v, "late") == 0) ? 0 : 1;
  }();
  return mode;
}
// what the look-ahead prepares: 0 = the whole plan (default), 1 = only its front (partition + bitmap: what the fused
// kernel needs); the per-bucket pass then runs on the side stream beside the fused kernel of the step that uses it
// (RC_AHEAD_PART=front)
int ahead_part_mode() {
  static int mode = [] {
    const char* v = getenv("RC_AHEAD_PART");
    return (v && strcmp(v, "front") == 0) ? 1 : 0;
  }();
  return mode;
}
}  // namespace

using namespace rc;

namespace {
// one complete bucket plan of a batch: partition buffers, row records, grouped positions, multi-occurrence bitmap.
// Two slots: the step works from one while the look-ahead writes the plan of the following batch into the other.
struct PlanSlot {
  PlanWs plan;
  rc_plan_row* rows_i;
  rc_plan_row* rows_u;
  uint32_t* occ;
  uint32_t* bitmap;
  PlanLongWs plan_long;   // hot rows of this plan: long-row records, chunk list (+ the chunk partial sums of its updates)
};
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
  PlanSlot slot[2];
  void* small_extra;     // small-batch step (small_step.hip): user-row snapshot, per-workgroup row / position segments
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
  // the pipelines never run in the same call: the plan buffers overlay the sort / segment scratch
  // (everything after loss_vec)
  Carver pv(base);
  pv.off = plan_off;
  for (int k = 0; k < 2; ++k) {
    const PlanWs pw = carve_plan_ws(base ? reinterpret_cast<char*>(base) + pv.off : nullptr, (int64_t)n_i + B);
    w.slot[k].plan = pw;
    pv.off += align_up(pw.total, 256);
    w.slot[k].rows_i = pv.take<rc_plan_row>(n_i);
    w.slot[k].rows_u = pv.take<rc_plan_row>((size_t)B);
    w.slot[k].occ = pv.take<uint32_t>(n_i + (size_t)B);
    w.slot[k].bitmap = pv.take<uint32_t>(kPlanBitmapWords);
    const PlanLongWs lw = carve_plan_long_ws(base ? reinterpret_cast<char*>(base) + pv.off : nullptr, (int64_t)n_i + B, d);
    w.slot[k].plan_long = lw;
    pv.off += align_up(lw.total, 256);
  }
  w.total = cv.off > pv.off ? cv.off : pv.off;
  // the small-batch step's buffers overlay the same region (the pipelines never run in the same call)
  w.small_extra = base ? reinterpret_cast<char*>(base) + plan_off : nullptr;
  if ((int64_t)n_i + B <= 32768) {
    const size_t need = plan_off + small_step_extra_bytes((int64_t)n_i + B, B, d);
    if (need > w.total) w.total = need;
  }
  return w;
}

// the plan of one batch in slot `k`
PlanArgs slot_plan_args(const StepWs& w, int k, const int64_t* uid, const int64_t* iid, int64_t n_i, int B, int64_t n_users,
                        int64_t n_items, const PlanGeom& geom, bool fused_upd) {
  PlanArgs pa;
  memset(&pa, 0, sizeof(pa));
  const PlanSlot& sl = w.slot[k];
  pa.ids_a = iid; pa.ids_b = uid; pa.n_a = (uint32_t)n_i; pa.n = (uint32_t)(n_i + B);
  pa.range_a = n_items; pa.range_b = n_users;
  pa.g = geom;
  pa.w = sl.plan;
  pa.list_single_a = fused_upd ? 0 : 1;
  pa.single_a = nullptr;
  pa.bitmap_a = fused_upd ? sl.bitmap : nullptr;
  pa.flags_done = 1;   // singleton information, where needed, is the bitmap
  pa.rows_a = sl.rows_i; pa.rows_b = sl.rows_u;
  pa.n_rows_a = &sl.plan.counters[PC_ROWS_A]; pa.n_rows_b = &sl.plan.counters[PC_ROWS_B];
  pa.occ = sl.occ;
  pa.emit_long = 1;
  pa.lw = sl.plan_long;
  return pa;
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

static bool ticket_matches(const rc_step_ticket* t, uint64_t generation, const void* ws, int device, int B, int C, int d,
                           int64_t n_users, int64_t n_items, int flavour) {
  return t != nullptr && generation != 0 && t->generation == generation && t->ws == reinterpret_cast<uintptr_t>(ws) &&
         t->device == device && t->B == B && t->C == C && t->d == d && t->n_users == n_users && t->n_items == n_items &&
         t->flavour == flavour && (t->slot == 0 || t->slot == 1);
}

static int train_step_impl(float* U, float* I, float* mU, float* vU, float* mI, float* vI,
                           const int64_t* uid, const int64_t* iid, int B, int C, int d,
                           int64_t n_users, int64_t n_items, const rc_opt_hyper* h,
                           float inv_b, float* loss_out, float* pred, void* ws,
                           size_t ws_bytes, rc_stream_t stream, float* phase_ms,
                           uint64_t generation, const int64_t* next_uid, const int64_t* next_iid, uint64_t next_generation,
                           rc_step_ticket* ticket) {
  RC_REQUIRE(U && I && uid && iid && h && loss_out && ws, "rc_bprmf_train_step: null pointer");
  RC_REQUIRE(B >= 1 && C >= 2 && d >= 1, "rc_bprmf_train_step: bad shape B=%d C=%d d=%d", B, C, d);
  RC_REQUIRE((int64_t)B * C < ((int64_t)1 << 31), "rc_bprmf_train_step: B*C too large");
  RC_REQUIRE(U != I, "rc_bprmf_train_step: user and item tables must be distinct");
  const StepWs w = carve_step_ws(ws, B, C, d);
  if (ws_bytes < w.total)
    return fail(RC_ERR_WORKSPACE, "rc_bprmf_train_step: workspace %zu < %zu", ws_bytes, w.total);
  hipStream_t s = as_stream(stream);
  const int64_t n_i = (int64_t)B * C;

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
  // (a hashed plan geometry -- very wide / sparse id spaces -- has no id-indexed bitmap: every row is listed then)
  const bool want_bitmap = rc_bprmf_fused_supported(d, C) != 0 && h->opt == RC_OPT_SGD;   // -> id-range buckets if at all possible
  const PlanGeom geom = plan_geometry(n_i, B, n_items, n_users, want_bitmap ? 0 : -1);
#if defined(RC_FUSED_UPD_NEVER)
  const bool fused_upd = false;
#elif defined(RC_FUSED_UPD_ALWAYS)
  const bool fused_upd = rc_bprmf_fused_supported(d, C) != 0 && !geom.hashed && !geom.narrow;
#else
  // (narrow geometry = dense batch, several occurrences per row of the table: hardly any row occurs once, the singleton
  //  fast path has nothing to win there)
  const bool fused_upd = rc_bprmf_fused_supported(d, C) != 0 && h->opt == RC_OPT_SGD && !geom.hashed && !geom.narrow;
#endif
  // what a prepared plan contains: 1 = bitmap + multi rows, 2 = every row listed, 3 = only the front of flavour 1
  const int flavour = fused_upd ? (ahead_part_mode() == 1 ? 3 : 1) : 2;

  // A plan prepared ahead by an earlier call (rc_step_ticket, caller-owned): usable when it was made for exactly this
  // batch -- the caller's generation id, not a pointer, says so -- workspace, geometry and plan flavour.  In every case
  // the side stream's writes into the workspace have to be finished before this call touches the plan buffers.
  int device = 0;
  StepSide* side = step_side(&device);
  bool ahead_hit = false;
  int slot = 0;
  if (ticket != nullptr && ticket->generation != 0) {
    RC_REQUIRE(ticket->device == device, "rc_bprmf_train_step_ahead: the ticket was prepared on device %d, current device %d",
               ticket->device, device);
    RC_REQUIRE(side != nullptr, "rc_bprmf_train_step_ahead: side stream unavailable on device %d", device);
    ahead_hit = ticket_matches(ticket, generation, ws, device, B, C, d, n_users, n_items, flavour);
    if (ahead_hit) slot = ticket->slot;
    RC_HIP(hipStreamWaitEvent(s, side->front_done, 0));
    ticket->generation = 0;
  }

  // Pipeline choice: the bucket plan (bucket_plan.hip + plan_update.hip) where the register-resident
  // fused kernel exists and the joint id space fits one bucket level; otherwise (and with RC_BPRMF_STEP=sort)
  // the round-1 pipeline: joint radix sort -> segment heads -> fused -> segmented updates.
  const bool force_sort = step_pipeline() == 1;
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
    const PlanArgs pa = slot_plan_args(w, slot, uid, iid, n_i, B, n_users, n_items, geom, fused_upd);
    const bool two_streams = (step_pipeline() == 0 || step_pipeline() == 3) && side != nullptr;
    // Look-ahead: the WHOLE plan of the next batch (partition, bitmap, row records, grouped positions) into the other
    // slot, on the side stream.  Not under stream capture (the next call's wait on front_done would cross graphs).
    // (Also in profiling mode: a profiled step is then exactly a step of the steady state, the next plan beside it.)
    bool look_ahead = two_streams && ticket != nullptr && next_generation != 0 && next_uid != nullptr && next_iid != nullptr;
    if (look_ahead) {
      hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
      look_ahead = hipStreamIsCapturing(s, &cap) == hipSuccess && cap == hipStreamCaptureStatusNone;
    }
    auto launch_ahead = [&]() -> int {
      PlanArgs pn = slot_plan_args(w, 1 - slot, next_uid, next_iid, n_i, B, n_users, n_items, geom, fused_upd);
      RC_HIP(hipEventRecord(side->fork2, s));
      RC_HIP(hipStreamWaitEvent(side->stream, side->fork2, 0));
      // a plan prepared as a whole: the bucket kernel writes the multi-occurrence bitmap from the counts it holds anyway
      // (the separate bitmap launch zeroes and counts every bucket a second time: 0.06 ms beside the row kernels)
      const bool whole = flavour != 3 && fused_upd && !geom.hashed && !geom.narrow && ahead_bitmap_in_bucket();
      pn.bitmap_in_bucket = whole ? 1 : 0;
      RC_TRY(plan_launch_front(pn, fused_upd && !whole, side->stream));
      if (flavour != 3) RC_TRY(plan_launch_back(pn, side->stream));
      RC_HIP(hipEventRecord(side->front_done, side->stream));
      ticket->generation = next_generation;
      ticket->ws = reinterpret_cast<uintptr_t>(ws);
      ticket->slot = 1 - slot; ticket->device = device; ticket->B = B; ticket->C = C; ticket->d = d;
      ticket->flavour = flavour; ticket->n_users = n_users; ticket->n_items = n_items;
      return RC_OK;
    };
    RC_MARK(0);
    if (ahead_hit && flavour == 3) {
      // the front is prepared; the per-bucket pass runs on the side stream beside this step's fused kernel
      RC_MARK(1);
      RC_HIP(hipEventRecord(side->fork, s));
      RC_HIP(hipStreamWaitEvent(side->stream, side->fork, 0));
      RC_TRY(plan_launch_back(pa, side->stream));
      RC_HIP(hipEventRecord(side->join, side->stream));
    } else if (ahead_hit) {
      // the plan is complete (prepared beside the previous step): this step starts with its fused kernel
      RC_MARK(1);
    } else if (two_streams) {
      // caller's stream: partition (+ bitmap when the fused kernel updates singleton rows) -> fused kernel
      // side stream:     per-bucket pass (row records, grouped positions), joined before the updates
      // (without the singleton fast path the fused kernel needs nothing from the plan: all of it runs on the side stream)
      if (fused_upd) RC_TRY(plan_launch_front(pa, true, s));
      RC_MARK(1);
      RC_HIP(hipEventRecord(side->fork, s));
      RC_HIP(hipStreamWaitEvent(side->stream, side->fork, 0));
      if (!fused_upd) RC_TRY(plan_launch_front(pa, false, side->stream));
      RC_TRY(plan_launch_back(pa, side->stream));
      RC_HIP(hipEventRecord(side->join, side->stream));
    } else {
      RC_TRY(plan_launch_front(pa, fused_upd, s));
      RC_MARK(1);   // after the partition (+ bitmap), before the bucket kernel
      RC_TRY(plan_launch_back(pa, s));
    }
    if (look_ahead && ahead_fork_mode() == 1) RC_TRY(launch_ahead());
    RC_MARK(2);
    RC_MARK(3);
    if (fused_upd)
      RC_TRY(rc_bprmf_fwd_bwd_update_bitmap(U, I, mI, vI, uid, iid, w.slot[slot].bitmap, B, C, d, inv_b, h, pred, w.loss_vec,
                                            w.gpred, w.ugrad, stream));
    else
      RC_TRY(rc_bprmf_fwd_bwd(U, I, uid, iid, B, C, d, inv_b, pred, w.loss_vec, w.gpred, w.ugrad, stream));
    if (two_streams && (!ahead_hit || flavour == 3)) RC_HIP(hipStreamWaitEvent(s, side->join, 0));
    if (look_ahead && ahead_fork_mode() == 0) RC_TRY(launch_ahead());
    RC_MARK(4);
    RC_MARK(5);  // (the loss mean is one workgroup of the last update launch)
    RC_TRY(plan_bprmf_step_updates(U, mU, vU, I, mI, vI, d, uid, C, n_i, B, w.gpred, w.ugrad, pa.rows_a, pa.n_rows_a,
                                   pa.rows_b, pa.n_rows_b, pa.occ, pa.w.counters, pa.lw, true, h, w.loss_vec, inv_b,
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
                         phase_ms, 0, nullptr, nullptr, 0, nullptr);
}

extern "C" int rc_bprmf_train_step_ahead(float* U, float* I, float* mU, float* vU, float* mI, float* vI,
                                         const int64_t* uid, const int64_t* iid, uint64_t generation,
                                         const int64_t* next_uid, const int64_t* next_iid, uint64_t next_generation,
                                         rc_step_ticket* ticket, int B, int C, int d, int64_t n_users, int64_t n_items,
                                         const rc_opt_hyper* h, float inv_b, float* loss_out, float* pred, void* ws,
                                         size_t ws_bytes, rc_stream_t stream, float* phase_ms) {
  RC_REQUIRE(ticket != nullptr, "rc_bprmf_train_step_ahead: ticket missing (caller-owned rc_step_ticket, zero-initialised)");
  return train_step_impl(U, I, mU, vU, mI, vI, uid, iid, B, C, d, n_users, n_items, h, inv_b, loss_out, pred, ws, ws_bytes, stream,
                         phase_ms, generation, next_uid, next_iid, next_generation, ticket);
}

// Forget a prepared plan (the owner of the workspace goes away or re-allocates it): `stream` is made to wait for the
// side stream's writes into that workspace, so that whatever reuses the memory afterwards is ordered behind them.
extern "C" int rc_bprmf_step_ahead_reset(rc_step_ticket* ticket, rc_stream_t stream) {
  RC_REQUIRE(ticket != nullptr, "rc_bprmf_step_ahead_reset: ticket missing");
  if (ticket->generation != 0) {
    int cur = 0;
    RC_HIP(hipGetDevice(&cur));
    if (cur != ticket->device) RC_HIP(hipSetDevice(ticket->device));
    StepSide* side = step_side();
    int rc_ = RC_OK;
    if (side != nullptr && hipStreamWaitEvent(as_stream(stream), side->front_done, 0) != hipSuccess)
      rc_ = fail(RC_ERR_HIP, "rc_bprmf_step_ahead_reset: hipStreamWaitEvent failed");
    if (cur != ticket->device) (void)hipSetDevice(cur);
    ticket->generation = 0;
    return rc_;
  }
  return RC_OK;
}
