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
// known) re-reads nothing from HBM, LDS is not needed at all, and the only cross-lane
// traffic is DPP row reductions + a few ds_bpermute across the GS groups.
// Small C packs several tuples per wave (C=2: two 32-lane tuples).
//
// Singleton rows (MODE != MODE_NONE).  An item row that occurs exactly once in the batch is
// read by nobody else in this step, and its complete gradient g_c * U[u] is known right
// here, where the row itself is still in registers: the kernel applies the optimizer and
// writes the new row, so the row crosses HBM once in each direction per step (the
// compulsory traffic).  `single[o]` comes from rc_mark_singletons on the sorted ids; rows
// with several occurrences are left to the segmented update (seg_update.hip).
#include "bpr_math.hpp"
#include "common.hpp"
#include "opt_math.hpp"

namespace rc {

struct FusedUpd {     // singleton-row update (unused when MODE == MODE_NONE)
  float* I;           // the item table again, writable (no __restrict__: aliases the input)
  float* M;
  float* V;
  const uint8_t* single;
  OptScalars o;
};

// all-reduce over the S lanes (power of two, S-aligned) that own one tuple
template <int S>
__device__ __forceinline__ float tuple_allreduce_sum(float x) {
  x = row_allreduce_sum<(S < 16 ? S : 16)>(x);
  if (S >= 32) x += __shfl_xor(x, 16, 64);
  if (S >= 64) x += __shfl_xor(x, 32, 64);
  return x;
}
template <int S>
__device__ __forceinline__ float tuple_allreduce_max(float x) {
  if (S >= 2) x = fmaxf(x, dpp_mov<0xB1>(x));
  if (S >= 4) x = fmaxf(x, dpp_mov<0x4E>(x));
  if (S >= 8) x = fmaxf(x, dpp_mov<0x141>(x));
  if (S >= 16) x = fmaxf(x, dpp_mov<0x140>(x));
  if (S >= 32) x = fmaxf(x, __shfl_xor(x, 16, 64));
  if (S >= 64) x = fmaxf(x, __shfl_xor(x, 32, 64));
  return x;
}

// The loss of a tuple is scalar work on its C scores.  Every lane of a row's lane-group holds the same
// score after the DPP row reduction, so computing softmax / sigmoid per register slot repeats each
// exp / division LPR times (16 x at d = 64) and keeps three CPL-long arrays alive next to the row
// registers (198 VGPRs -> two waves per SIMD; measured 0.21 ms of VALU time per step that two waves
// cannot hide).  Here the scores are transposed through a C-float LDS strip per tuple: one lane per
// candidate evaluates the loss terms (ceil(C / S) per lane instead of CPL), the gradient scalars g_c
// return through the same strip and are broadcast-read by the lane-groups for the backward pass.
// Slot order in the strip: s = grp * CPL + j  <->  candidate c = j * GS + grp.
#ifndef RC_FUSED_MINW
#define RC_FUSED_MINW 3
#endif
// waves per SIMD the register allocation must allow: the candidate block (CPL float4 = 4 CPL VGPRs) is the
// floor; the stateful singleton paths keep their per-slot gradient scalars as well
template <int CPL_, int MODE_>
constexpr int fused_min_waves() {
  return (MODE_ == MODE_ADAM || MODE_ == MODE_ADAGRAD) ? (CPL_ >= 20 ? 2 : 3) : (CPL_ >= 26 ? 2 : (CPL_ >= 20 ? RC_FUSED_MINW : 4));
}
template <int D, int GS, int CPL, int MODE>
__global__ __launch_bounds__(kBlock, (fused_min_waves<CPL, MODE>())) void bprmf_fwd_bwd_kernel(
    const float* __restrict__ U, const float* I,
    const int64_t* __restrict__ uid, const int64_t* __restrict__ iid, int B, int C,
    float inv_b, float* __restrict__ pred, float* __restrict__ loss_vec,
    float* __restrict__ gpred, float* __restrict__ ugrad, FusedUpd upd) {
  constexpr int LPR = D / 4;
  constexpr int S = LPR * GS;
  static_assert(S <= 64 && (64 % S) == 0, "tuple must fit a wave");
  constexpr int TPW = 64 / S;          // tuples per wave
  constexpr int SLOTS = GS * CPL;      // strip length (>= C)
  constexpr int NPL = (SLOTS + S - 1) / S;  // candidates per lane in the loss phase
  __shared__ float strip_mem[(kBlock / 64) * TPW * SLOTS];

  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
  const int64_t t_raw = wave * TPW + lane / S;
  const bool tv = t_raw < B;
  const int64_t t = tv ? t_raw : (int64_t)B - 1;  // clamp: every lane stays in the shuffles
  const int sub = lane % S;
  const int grp = sub / LPR;
  const int l = sub % LPR;
  float* strip = strip_mem + ((threadIdx.x >> 6) * TPW + lane / S) * SLOTS;

  // ---- gather: user row, then this group's CPL candidate rows, all loads in flight
  const int64_t u = uid[t];
  const float4 u4 = reinterpret_cast<const float4*>(U + u * D)[l];
  const int64_t* ids = iid + t * C;
  float4 r[CPL];
#pragma unroll
  for (int j = 0; j < CPL; ++j) {
    const int c = j * GS + grp;
    const int64_t id = ids[c < C ? c : 0];  // slots past C re-read candidate 0; masked below
    r[j] = load_stream4(reinterpret_cast<const float4*>(I + id * D) + l);
  }
  unsigned smask = 0;  // bit j: candidate slot j of this group is a singleton row
  if (MODE != MODE_NONE) {
    if (S == 64 && GS == 4 && (C & 3) == 0) {
      // the tuple's C flag bytes as C/4 dwords in ONE coalesced load (lane k holds candidates 4k..4k+3); dword j is
      // then a wave-uniform value (readlane with a constant lane) whose byte `grp` is this group's candidate j*4+grp.
      // (25 separate byte loads per lane cost 26 us of the 0.56 ms kernel at config 2.)
      const uint32_t* f32p = reinterpret_cast<const uint32_t*>(upd.single + t * C);
      const uint32_t mine = (tv && lane < C / 4) ? f32p[lane] : 0u;
#pragma unroll
      for (int j = 0; j < CPL; ++j) {
        const uint32_t wj = (uint32_t)__builtin_amdgcn_readlane((int)mine, j);
        if ((wj >> (8 * grp)) & 0xFFu) smask |= 1u << j;
      }
    } else {
#pragma unroll
      for (int j = 0; j < CPL; ++j) {
        const int c = j * GS + grp;
        if (tv && c < C && upd.single[t * C + c]) smask |= 1u << j;
      }
    }
  }

  // ---- scores -> strip
#pragma unroll
  for (int j = 0; j < CPL; ++j) {
    const float pj = row_allreduce_sum<LPR>(dot4(u4, r[j]));
    if (l == 0) strip[grp * CPL + j] = pj;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

  // ---- loss, one lane per candidate: softmax over the negatives, P = sum w * sigmoid(pos - neg)
  float pc[NPL];
  int cc[NPL];
#pragma unroll
  for (int k = 0; k < NPL; ++k) {
    const int s = sub + k * S;
    const int c = (s % CPL) * GS + s / CPL;
    cc[k] = (s < SLOTS && c < C) ? c : -1;
    pc[k] = s < SLOTS ? strip[s] : 0.f;
  }
  const float pos = strip[0];  // candidate 0 = group 0, j = 0
  float mx = -INFINITY;
#pragma unroll
  for (int k = 0; k < NPL; ++k)
    if (cc[k] >= 1) mx = fmaxf(mx, pc[k]);
  mx = tuple_allreduce_max<S>(mx);
  float ew[NPL], sg[NPL];
  float se = 0.f;
#pragma unroll
  for (int k = 0; k < NPL; ++k) {
    ew[k] = cc[k] >= 1 ? expf(pc[k] - mx) : 0.f;
    se += ew[k];
  }
  se = tuple_allreduce_sum<S>(se);
  const float inv_se = 1.0f / se;
  float P = 0.f, A = 0.f;
#pragma unroll
  for (int k = 0; k < NPL; ++k) {
    ew[k] *= inv_se;  // softmax weight (0 on the positive and on masked slots)
    sg[k] = sigmoidf_(pos - pc[k]);
    P = fmaf(ew[k], sg[k], P);
    A = fmaf(ew[k], sg[k] * (1.0f - sg[k]), A);
  }
  P = tuple_allreduce_sum<S>(P);
  A = tuple_allreduce_sum<S>(A);
  const BprRow br = bpr_row(P, inv_b);
  if (tv && sub == 0) loss_vec[t] = br.loss;
#pragma unroll
  for (int k = 0; k < NPL; ++k) {
    const int s = sub + k * S;
    float g = br.dLdP * bpr_dP_dneg(ew[k], sg[k], P);
    if (cc[k] == 0) g = br.dLdP * A;
    if (cc[k] < 0) g = 0.f;
    if (s < SLOTS) strip[s] = g;
    if (tv && cc[k] >= 0) {
      gpred[t * C + cc[k]] = g;
      if (pred != nullptr) pred[t * C + cc[k]] = pc[k];
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

  // ---- backward: user-row gradient = sum_c g_c * I_c; single-occurrence item rows updated in place
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  float gs[CPL];  // stateful optimizers only
#pragma unroll
  for (int j = 0; j < CPL; ++j) {
    const int c = j * GS + grp;
    const float g = strip[grp * CPL + j];  // broadcast read, 0 on slots past C
    acc.x = fmaf(g, r[j].x, acc.x);
    acc.y = fmaf(g, r[j].y, acc.y);
    acc.z = fmaf(g, r[j].z, acc.z);
    acc.w = fmaf(g, r[j].w, acc.w);
    if (MODE == MODE_ADAM || MODE == MODE_ADAGRAD) gs[j] = g;  // row updates below
  }
  if (MODE == MODE_SGD) {
    // single-occurrence rows: all write-backs of the wave in one burst after the accumulation
#pragma unroll
    for (int j = 0; j < CPL; ++j) {
      if (smask & (1u << j)) {  // whole lane-group takes the branch together
        const int64_t id = ids[j * GS + grp];
        const float g = strip[grp * CPL + j];
        const float4 gi = make_float4(g * u4.x, g * u4.y, g * u4.z, g * u4.w);
        opt_row4<MODE>(upd.o, upd.I, upd.M, upd.V, (size_t)id * LPR + l, r[j], gi);
      }
    }
  }
  acc.x = groups_allreduce_sum<LPR, S>(acc.x);
  acc.y = groups_allreduce_sum<LPR, S>(acc.y);
  acc.z = groups_allreduce_sum<LPR, S>(acc.z);
  acc.w = groups_allreduce_sum<LPR, S>(acc.w);
  if (tv && grp == 0) reinterpret_cast<float4*>(ugrad + t * D)[l] = acc;

  if (MODE == MODE_ADAM || MODE == MODE_ADAGRAD) {
    // singleton rows under Adam / Adagrad: the m (and v) rows of GRP candidates are requested
    // together before any is used -- one memory round trip per batch instead of one per row
    constexpr int GRP = 5;
#pragma unroll
    for (int j0 = 0; j0 < CPL; j0 += GRP) {
      float4 mm[GRP], vv[GRP];
      size_t idx[GRP];
#pragma unroll
      for (int q = 0; q < GRP; ++q) {
        const int j = j0 + q;
        mm[q] = vv[q] = make_float4(0.f, 0.f, 0.f, 0.f);
        idx[q] = 0;
        if (j < CPL && ((smask >> j) & 1u)) {
          idx[q] = (size_t)ids[j * GS + grp] * LPR + l;
          mm[q] = load_stream4(reinterpret_cast<const float4*>(upd.M) + idx[q]);
          if (MODE == MODE_ADAM) vv[q] = load_stream4(reinterpret_cast<const float4*>(upd.V) + idx[q]);
        }
      }
#pragma unroll
      for (int q = 0; q < GRP; ++q) {
        const int j = j0 + q;
        if (j < CPL && ((smask >> j) & 1u)) {
          const float g = gs[j];
          const float4 gi = make_float4(g * u4.x, g * u4.y, g * u4.z, g * u4.w);
          float4 w = r[j];
          opt_apply4<MODE>(upd.o, w, mm[q], vv[q], gi);
          store_row4(reinterpret_cast<float4*>(upd.I) + idx[q], w);
          store_row4(reinterpret_cast<float4*>(upd.M) + idx[q], mm[q]);
          if (MODE == MODE_ADAM) store_row4(reinterpret_cast<float4*>(upd.V) + idx[q], vv[q]);
        }
      }
    }
  }
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
};

template <int D, int GS, int CPL>
static int launch_fused(const FusedCall& f) {
  constexpr int TPW = 64 / ((D / 4) * GS);
  constexpr int TPB = TPW * (kBlock / 64);
  const int blocks = (f.B + TPB - 1) / TPB;
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

extern "C" int rc_bprmf_fwd_bwd_update(const float* U, float* I, float* mI, float* vI,
                                       const int64_t* uid, const int64_t* iid,
                                       const uint8_t* single, int B, int C, int d, float inv_b,
                                       const rc_opt_hyper* h, float* pred, float* loss_vec,
                                       float* gpred, float* ugrad, rc_stream_t stream) {
  if (B == 0) return RC_OK;
  RC_REQUIRE(U && I && uid && iid && single && h && loss_vec && gpred && ugrad,
             "rc_bprmf_fwd_bwd_update: null pointer");
  RC_REQUIRE(B > 0 && C >= 2 && d >= 1, "rc_bprmf_fwd_bwd_update: bad shape B=%d C=%d d=%d", B, C, d);
  if (!register_path_ok(d, C))
    return fail(RC_ERR_UNSUPPORTED,
                "rc_bprmf_fwd_bwd_update: no register-resident kernel for d=%d C=%d "
                "(check rc_bprmf_fused_supported; use rc_bprmf_fwd_bwd + rc_segmented_update)", d, C);
  FusedCall f;
  memset(&f, 0, sizeof(f));
  RC_TRY(fill_opt_scalars(h, &f.upd.o));
  f.mode = mode_of(h);
  RC_REQUIRE(f.mode != MODE_ADAM || (mI && vI), "rc_bprmf_fwd_bwd_update: Adam needs mI and vI");
  RC_REQUIRE(f.mode != MODE_ADAGRAD || mI, "rc_bprmf_fwd_bwd_update: Adagrad needs mI");
  RC_REQUIRE(aligned16(U) && aligned16(I) && aligned16(ugrad) && aligned16(mI) && aligned16(vI),
             "rc_bprmf_fwd_bwd_update: tables must be 16-byte aligned");
  f.U = U; f.I = I; f.uid = uid; f.iid = iid; f.B = B; f.C = C; f.inv_b = inv_b;
  f.pred = pred; f.loss_vec = loss_vec; f.gpred = gpred; f.ugrad = ugrad;
  f.upd.I = I; f.upd.M = mI; f.upd.V = vI; f.upd.single = single;
  f.s = as_stream(stream);
  bool handled = false;
  const int rc_ = run_fused(f, d, &handled);
  if (!handled) return fail(RC_ERR_UNSUPPORTED, "rc_bprmf_fwd_bwd_update: dispatch failed");
  return rc_;
}
