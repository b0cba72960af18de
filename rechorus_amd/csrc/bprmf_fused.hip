// bprmf_fused.hip -- phase A of a BPRMF training step: gather + GMF dot + BPR loss +
// backward to per-tuple row gradients, in ONE pass over HBM.
//
// Reference arithmetic: models/general/BPRMF.py:34-45 (gather, broadcast mul, sum),
// models/BaseModel.py:182-185 (loss) and the MulBackward/SumBackward nodes autograd
// derives from them.
//
// Layout.  A tuple's C = 1+K candidate rows are the unit of work.  A row of D fp32 is
// owned by LPR = D/4 consecutive lanes (one float4 each -> one coalesced D*4-byte
// segment per row); a tuple is owned by GS such lane-groups (S = LPR*GS lanes), each
// group holding CPL = ceil(C/GS) candidate rows *in registers* (CPL float4 per lane).
// D=64, C=100: LPR=16, GS=4, CPL=25 -> the whole 25.6 KB candidate block of a tuple
// lives in one wave's VGPRs (100 VGPRs/lane), so the backward pass (user-row gradient
// sum_c g_c * I_c, which needs every row again after the softmax over all K negatives is
// known) re-reads nothing from HBM, and the only cross-lane traffic is DPP row reductions +
// a few ds_bpermute across the GS groups.  The only LDS use is a strip of C floats per tuple that
// transposes the scores so that the loss is evaluated one lane per candidate (fused_body.hpp).
// Small C packs several tuples per wave (C=2: two 32-lane tuples).
//
// Singleton rows (MODE != MODE_NONE).  An item row that occurs exactly once in the batch is
// read by nobody else in this step, and its complete gradient g_c * U[u] is known right
// here, where the row itself is still in registers: the kernel applies the optimizer and
// writes the new row, so the row crosses HBM once in each direction per step (the
// compulsory traffic).  `single[o]` comes from the bucket plan (bucket_plan.hip: plan_flags_kernel;
// rc_segment_heads on the sorted ids in the sort pipeline); rows with several occurrences are left to
// the plan-driven row update (plan_update.hip; seg_update.hip in the sort pipeline).  The kernel body
// lives in fused_body.hpp: small_front_kernel below runs it beside the small-batch plan workgroups.
#include "fused_body.hpp"
#include "small_plan.hpp"

namespace rc {

template <int D, int GS, int CPL, int MODE>
__global__ __launch_bounds__(kBlock, (fused_min_waves<CPL, MODE>())) void bprmf_fwd_bwd_kernel(
    const float* __restrict__ U, const float* I,
    const int64_t* __restrict__ uid, const int64_t* __restrict__ iid, int B, int C,
    float inv_b, float* __restrict__ pred, float* __restrict__ loss_vec,
    float* __restrict__ gpred, float* __restrict__ ugrad, FusedUpd upd) {
  bprmf_fwd_bwd_body<D, GS, CPL, MODE, kBlock>(U, I, uid, iid, B, C, inv_b, pred, loss_vec, gpred, ugrad, upd,
                                               (int64_t)blockIdx.x, nullptr);
}

// First launch of the small-batch step (small_step.hip): workgroups [0, kSmallPlanWgs) group the batch's row ids
// (small_plan.hpp), the others run the fused forward / loss / backward (read-only: every touched row is updated by the
// second launch) and snapshot the batch's user rows.  The two parts are independent, so one launch runs them side by side.
template <int D, int GS, int CPL>
__global__ __launch_bounds__(kSmallThreads) void small_front_kernel(
    const float* __restrict__ U, const float* I, const int64_t* __restrict__ uid, const int64_t* __restrict__ iid,
    int B, int C, float inv_b, float* __restrict__ pred, float* __restrict__ loss_vec, float* __restrict__ gpred,
    float* __restrict__ ugrad, float* __restrict__ ub, SmallPlanArgs plan) {
  extern __shared__ __attribute__((aligned(16))) unsigned char small_smem[];
  if (blockIdx.x < (unsigned)kSmallPlanWgs) {
    small_plan_block(plan, blockIdx.x, small_smem);
    return;
  }
  FusedUpd none;
  none.I = nullptr; none.M = nullptr; none.V = nullptr; none.single = nullptr;
  bprmf_fwd_bwd_body<D, GS, CPL, MODE_NONE, kSmallThreads>(U, I, uid, iid, B, C, inv_b, pred, loss_vec, gpred, ugrad, none,
                                                           (int64_t)blockIdx.x - kSmallPlanWgs, ub);
}

// Any d, any 2 <= C <= kGenericMaxC: one wave per tuple, scores staged in LDS, candidate
// rows re-gathered (from L2/MALL) for the backward sum.  Correctness fall-back.
constexpr int kGenericMaxC = 4096;
constexpr int kGenericMaxDChunks = 8;  // d <= 512

__global__ __launch_bounds__(kBlock) void bprmf_fwd_bwd_generic_kernel(
    const float* __restrict__ U, const float* __restrict__ I,
    const int64_t* __restrict__ uid, const int64_t* __restrict__ iid, int B, int C, int d,
    float inv_b, float* __restrict__ pred, float* __restrict__ loss_vec,
    float* __restrict__ gpred, float* __restrict__ ugrad) {
  extern __shared__ float smem[];  // [waves per block][C]
  const int lane = threadIdx.x & 63;
  const int wib = threadIdx.x >> 6;
  const int64_t t_raw = (int64_t)blockIdx.x * (kBlock / 64) + wib;
  const bool tv = t_raw < B;  // wave-uniform; no early return (block barriers below)
  const int64_t t = tv ? t_raw : (int64_t)B - 1;
  float* sp = smem + (size_t)wib * C;
  const float* ur = U + uid[t] * d;
  const int64_t* ids = iid + t * C;
  for (int c = 0; c < C; ++c) {
    const float* ir = I + ids[c] * d;
    float a = 0.f;
    for (int k = lane; k < d; k += 64) a = fmaf(ur[k], ir[k], a);
    a = wave_allreduce_sum(a);
    if (lane == 0) sp[c] = a;
  }
  __syncthreads();
  const float pos = sp[0];
  float mx = -INFINITY;
  for (int c = 1 + lane; c < C; c += 64) mx = fmaxf(mx, sp[c]);
  mx = wave_allreduce_max(mx);
  float se = 0.f;
  for (int c = 1 + lane; c < C; c += 64) se += expf(sp[c] - mx);
  se = wave_allreduce_sum(se);
  const float inv_se = 1.0f / se;
  float P = 0.f, A = 0.f;
  for (int c = 1 + lane; c < C; c += 64) {
    const float w = expf(sp[c] - mx) * inv_se;
    const float s = sigmoidf_(pos - sp[c]);
    P = fmaf(w, s, P);
    A = fmaf(w, s * (1.0f - s), A);
  }
  P = wave_allreduce_sum(P);
  A = wave_allreduce_sum(A);
  const BprRow br = bpr_row(P, inv_b);
  if (tv && lane == 0) loss_vec[t] = br.loss;
  __syncthreads();  // all lanes have read sp[] as scores
  for (int c = lane; c < C; c += 64) {
    const float pc = sp[c];
    if (pred && tv) pred[t * C + c] = pc;
    float g;
    if (c == 0) {
      g = br.dLdP * A;
    } else {
      const float w = expf(pc - mx) * inv_se;
      const float s = sigmoidf_(pos - pc);
      g = br.dLdP * bpr_dP_dneg(w, s, P);
    }
    if (tv) gpred[t * C + c] = g;
    sp[c] = g;
  }
  __syncthreads();
  float acc[kGenericMaxDChunks];
#pragma unroll
  for (int q = 0; q < kGenericMaxDChunks; ++q) acc[q] = 0.f;
  for (int c = 0; c < C; ++c) {
    const float* ir = I + ids[c] * d;
    const float g = sp[c];
#pragma unroll
    for (int q = 0; q < kGenericMaxDChunks; ++q) {
      const int k = lane + 64 * q;
      if (k < d) acc[q] = fmaf(g, ir[k], acc[q]);
    }
  }
#pragma unroll
  for (int q = 0; q < kGenericMaxDChunks; ++q) {
    const int k = lane + 64 * q;
    if (tv && k < d) ugrad[t * d + k] = acc[q];
  }
}

struct FusedCall {
  const float* U;
  const float* I;
  const int64_t* uid;
  const int64_t* iid;
  int B, C;
  float inv_b;
  float* pred;
  float* loss_vec;
  float* gpred;
  float* ugrad;
  int mode;  // MODE_NONE or the optimizer mode of the singleton update
  FusedUpd upd;
  hipStream_t s;
  const SmallPlanArgs* small;  // non-null: launch small_front_kernel (plan workgroups + this kernel's body)
  float* ub;
};

template <int D, int GS, int CPL>
static int launch_fused(const FusedCall& f) {
  constexpr int TPW = 64 / ((D / 4) * GS);
  constexpr int TPB = TPW * (kBlock / 64);
  const int blocks = (f.B + TPB - 1) / TPB;
  if (f.small) {
    constexpr int TPB_S = TPW * (kSmallThreads / 64);
    const int fblocks = (f.B + TPB_S - 1) / TPB_S;
    auto kern = small_front_kernel<D, GS, CPL>;
    static bool attr_done = false;  // per instantiation
    if (!attr_done) {
      RC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kSmallLdsBytes));
      attr_done = true;
    }
    hipLaunchKernelGGL(kern, dim3(kSmallPlanWgs + fblocks), dim3(kSmallThreads), kSmallLdsBytes, f.s, f.U, f.I, f.uid, f.iid,
                       f.B, f.C, f.inv_b, f.pred, f.loss_vec, f.gpred, f.ugrad, f.ub, *f.small);
    RC_LAUNCH_CHECK();
    return RC_OK;
  }
#define RC_GO(MODE_)                                                                          \
  hipLaunchKernelGGL((bprmf_fwd_bwd_kernel<D, GS, CPL, MODE_>), dim3(blocks), dim3(kBlock), 0, \
                     f.s, f.U, f.I, f.uid, f.iid, f.B, f.C, f.inv_b, f.pred, f.loss_vec,       \
                     f.gpred, f.ugrad, f.upd)
  switch (f.mode) {
    case MODE_SGD: RC_GO(MODE_SGD); break;
    case MODE_ADAM: RC_GO(MODE_ADAM); break;
    case MODE_ADAGRAD: RC_GO(MODE_ADAGRAD); break;
    default: RC_GO(MODE_NONE); break;
  }
#undef RC_GO
  RC_LAUNCH_CHECK();
  return RC_OK;
}

// pick (GS, CPL) for C candidates when a wave offers G = 64/LPR lane-groups
template <int D>
static int dispatch_fused(const FusedCall& f, bool* handled) {
  constexpr int G = 64 / (D / 4);
  const int C = f.C;
  *handled = true;
#define RC_FUSED(GS_, CPL_) return launch_fused<D, GS_, CPL_>(f)
  if (G >= 2 && C <= 2) { RC_FUSED((G >= 2 ? 2 : 1), 1); }
  if (G >= 4 && C <= 4) { RC_FUSED((G >= 4 ? 4 : 1), 1); }
  if (G >= 8 && C <= 8) { RC_FUSED((G >= 8 ? 8 : 1), 1); }
  if (G >= 16 && C <= 16) { RC_FUSED((G >= 16 ? 16 : 1), 1); }
  const int cpl = (C + G - 1) / G;
  if (cpl <= 1) { RC_FUSED(G, 1); }
  if (cpl <= 2) { RC_FUSED(G, 2); }
  if (cpl <= 3) { RC_FUSED(G, 3); }
  if (cpl <= 4) { RC_FUSED(G, 4); }
  if (cpl <= 6) { RC_FUSED(G, 6); }
  if (cpl <= 8) { RC_FUSED(G, 8); }
  if (cpl <= 13) { RC_FUSED(G, 13); }
  if (cpl <= 16) { RC_FUSED(G, 16); }
  if (cpl <= 20) { RC_FUSED(G, 20); }
  if (cpl <= 25) { RC_FUSED(G, 25); }
  if (cpl <= 32) { RC_FUSED(G, 32); }
#undef RC_FUSED
  *handled = false;
  return RC_OK;
}

static bool register_path_ok(int d, int C) {
  if (d != 16 && d != 32 && d != 64 && d != 128) return false;
  const int G = 64 / (d / 4);
  return (C + G - 1) / G <= 32;
}

static int run_fused(const FusedCall& f, int d, bool* handled) {
  *handled = false;
  switch (d) {
    case 16: return dispatch_fused<16>(f, handled);
    case 32: return dispatch_fused<32>(f, handled);
    case 64: return dispatch_fused<64>(f, handled);
    case 128: return dispatch_fused<128>(f, handled);
    default: return RC_OK;
  }
}

}  // namespace rc

using namespace rc;

static bool aligned16(const void* p) { return reinterpret_cast<uintptr_t>(p) % 16 == 0; }

extern "C" int rc_bprmf_fused_supported(int d, int C) { return register_path_ok(d, C) ? 1 : 0; }

extern "C" int rc_bprmf_fwd_bwd(const float* U, const float* I, const int64_t* uid,
                                const int64_t* iid, int B, int C, int d, float inv_b,
                                float* pred, float* loss_vec, float* gpred, float* ugrad,
                                rc_stream_t stream) {
  if (B == 0) return RC_OK;
  RC_REQUIRE(U && I && uid && iid && loss_vec && gpred && ugrad, "rc_bprmf_fwd_bwd: null pointer");
  RC_REQUIRE(B > 0 && C >= 2 && d >= 1,
             "rc_bprmf_fwd_bwd: need C >= 2 (one negative), got B=%d C=%d d=%d", B, C, d);
  hipStream_t s = as_stream(stream);
  if (aligned16(U) && aligned16(I) && aligned16(ugrad) && register_path_ok(d, C)) {
    FusedCall f;
    memset(&f, 0, sizeof(f));
    f.U = U; f.I = I; f.uid = uid; f.iid = iid; f.B = B; f.C = C; f.inv_b = inv_b;
    f.pred = pred; f.loss_vec = loss_vec; f.gpred = gpred; f.ugrad = ugrad;
    f.mode = MODE_NONE; f.s = s;
    bool handled = false;
    const int rc_ = run_fused(f, d, &handled);
    if (handled) return rc_;
  }
  if (C > kGenericMaxC || d > 64 * kGenericMaxDChunks)
    return fail(RC_ERR_UNSUPPORTED,
                "rc_bprmf_fwd_bwd: no kernel for C=%d d=%d (generic path: C<=%d, d<=%d)", C, d,
                kGenericMaxC, 64 * kGenericMaxDChunks);
  const int blocks = (B + (kBlock / 64) - 1) / (kBlock / 64);
  const size_t shmem = (size_t)(kBlock / 64) * C * sizeof(float);
  hipLaunchKernelGGL(bprmf_fwd_bwd_generic_kernel, dim3(blocks), dim3(kBlock), shmem, s, U, I,
                     uid, iid, B, C, d, inv_b, pred, loss_vec, gpred, ugrad);
  RC_LAUNCH_CHECK();
  return RC_OK;
}

static int fwd_bwd_update_impl(const char* who, const float* U, float* I, float* mI, float* vI, const int64_t* uid,
                               const int64_t* iid, const uint8_t* single, const uint32_t* multi, int B, int C, int d,
                               float inv_b, const rc_opt_hyper* h, float* pred, float* loss_vec, float* gpred, float* ugrad,
                               rc_stream_t stream) {
  if (B == 0) return RC_OK;
  RC_REQUIRE(U && I && uid && iid && (single || multi) && h && loss_vec && gpred && ugrad, "%s: null pointer", who);
  RC_REQUIRE(B > 0 && C >= 2 && d >= 1, "%s: bad shape B=%d C=%d d=%d", who, B, C, d);
  if (!register_path_ok(d, C))
    return fail(RC_ERR_UNSUPPORTED,
                "%s: no register-resident kernel for d=%d C=%d "
                "(check rc_bprmf_fused_supported; use rc_bprmf_fwd_bwd + rc_segmented_update)", who, d, C);
  FusedCall f;
  memset(&f, 0, sizeof(f));
  RC_TRY(fill_opt_scalars(h, &f.upd.o));
  f.mode = mode_of(h);
  RC_REQUIRE(f.mode != MODE_ADAM || (mI && vI), "%s: Adam needs mI and vI", who);
  RC_REQUIRE(f.mode != MODE_ADAGRAD || mI, "%s: Adagrad needs mI", who);
  RC_REQUIRE(aligned16(U) && aligned16(I) && aligned16(ugrad) && aligned16(mI) && aligned16(vI),
             "%s: tables must be 16-byte aligned", who);
  f.U = U; f.I = I; f.uid = uid; f.iid = iid; f.B = B; f.C = C; f.inv_b = inv_b;
  f.pred = pred; f.loss_vec = loss_vec; f.gpred = gpred; f.ugrad = ugrad;
  f.upd.I = I; f.upd.M = mI; f.upd.V = vI; f.upd.single = single; f.upd.multi = multi;
  f.s = as_stream(stream);
  bool handled = false;
  const int rc_ = run_fused(f, d, &handled);
  if (!handled) return fail(RC_ERR_UNSUPPORTED, "%s: dispatch failed", who);
  return rc_;
}

extern "C" int rc_bprmf_fwd_bwd_update(const float* U, float* I, float* mI, float* vI,
                                       const int64_t* uid, const int64_t* iid,
                                       const uint8_t* single, int B, int C, int d, float inv_b,
                                       const rc_opt_hyper* h, float* pred, float* loss_vec,
                                       float* gpred, float* ugrad, rc_stream_t stream) {
  if (B != 0) RC_REQUIRE(single, "rc_bprmf_fwd_bwd_update: null pointer");
  return fwd_bwd_update_impl("rc_bprmf_fwd_bwd_update", U, I, mI, vI, uid, iid, single, nullptr, B, C, d, inv_b, h, pred,
                             loss_vec, gpred, ugrad, stream);
}

extern "C" int rc_bprmf_fwd_bwd_update_bitmap(const float* U, float* I, float* mI, float* vI,
                                              const int64_t* uid, const int64_t* iid,
                                              const uint32_t* multi, int B, int C, int d, float inv_b,
                                              const rc_opt_hyper* h, float* pred, float* loss_vec,
                                              float* gpred, float* ugrad, rc_stream_t stream) {
  if (B != 0) RC_REQUIRE(multi, "rc_bprmf_fwd_bwd_update_bitmap: null pointer");
  return fwd_bwd_update_impl("rc_bprmf_fwd_bwd_update_bitmap", U, I, mI, vI, uid, iid, nullptr, multi, B, C, d, inv_b, h, pred,
                             loss_vec, gpred, ugrad, stream);
}

namespace rc {
// first launch of the small-batch step; the caller (train_step.hip) checked rc_bprmf_fused_supported(d, C)
int small_front_launch(const float* U, const float* I, const int64_t* uid, const int64_t* iid, int B, int C, int d, float inv_b,
                       float* pred, float* loss_vec, float* gpred, float* ugrad, float* ub, const SmallPlanArgs& plan,
                       hipStream_t s) {
  RC_REQUIRE(aligned16(U) && aligned16(I) && aligned16(ugrad) && aligned16(ub) && register_path_ok(d, C),
             "small-batch step: unsupported shape or alignment (d=%d C=%d)", d, C);
  FusedCall f;
  memset(&f, 0, sizeof(f));
  f.U = U; f.I = I; f.uid = uid; f.iid = iid; f.B = B; f.C = C; f.inv_b = inv_b;
  f.pred = pred; f.loss_vec = loss_vec; f.gpred = gpred; f.ugrad = ugrad;
  f.mode = MODE_NONE; f.s = s; f.small = &plan; f.ub = ub;
  bool handled = false;
  const int rc_ = run_fused(f, d, &handled);
  if (!handled) return fail(RC_ERR_UNSUPPORTED, "small-batch step: dispatch failed (d=%d C=%d)", d, C);
  return rc_;
}
}  // namespace rc
