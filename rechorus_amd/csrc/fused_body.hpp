// fused_body.hpp -- the register-resident BPRMF forward / loss / backward of one workgroup (see bprmf_fused.hip for
// the layout notes); shared by the stand-alone kernel and the small-batch step.
#pragma once
#include "bpr_math.hpp"
#include "common.hpp"
#include "opt_math.hpp"

namespace rc {

struct FusedUpd {     // singleton-row update (unused when MODE == MODE_NONE)
  float* I;           // the item table again, writable (no __restrict__: aliases the input)
  float* M;
  float* V;
  const uint8_t* single;   // flag byte per batch position (sort pipeline: rc_segment_heads), or
  const uint32_t* multi;   // bitmap over item ids, bit = 1 iff the row occurs at least twice in the batch (bucket plan)
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
// The body of the fused kernel for workgroup `block` of NT threads (the stand-alone kernel: NT = kBlock and its own
// workgroup index; the small-batch step, small_step.hip, runs it in the upper part of a merged grid).  ub: optional
// [B, D] snapshot of the batch's user rows (the small-batch step updates U and I in ONE later launch, so the item side
// must not read U any more).
template <int D, int GS, int CPL, int MODE, int NT>
__device__ __forceinline__ void bprmf_fwd_bwd_body(
    const float* __restrict__ U, const float* I,
    const int64_t* __restrict__ uid, const int64_t* __restrict__ iid, int B, int C,
    float inv_b, float* __restrict__ pred, float* __restrict__ loss_vec,
    float* __restrict__ gpred, float* __restrict__ ugrad, const FusedUpd& upd, int64_t block, float* __restrict__ ub) {
  constexpr int LPR = D / 4;
  constexpr int S = LPR * GS;
  static_assert(S <= 64 && (64 % S) == 0, "tuple must fit a wave");
  constexpr int TPW = 64 / S;          // tuples per wave
  constexpr int SLOTS = GS * CPL;      // strip length (>= C)
  constexpr int NPL = (SLOTS + S - 1) / S;  // candidates per lane in the loss phase
  __shared__ float strip_mem[(NT / 64) * TPW * SLOTS];

  const int lane = threadIdx.x & 63;
  const int64_t wave = block * (NT / 64) + (threadIdx.x >> 6);
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
  if (ub != nullptr && tv && grp == 0) reinterpret_cast<float4*>(ub + t * D)[l] = u4;
  const int64_t* ids = iid + t * C;
  float4 r[CPL];
#pragma unroll
  for (int j = 0; j < CPL; ++j) {
    const int c = j * GS + grp;
    const int64_t id = ids[c < C ? c : 0];  // slots past C re-read candidate 0; masked below
    r[j] = load_stream4(reinterpret_cast<const float4*>(I + id * D) + l);
  }
  unsigned smask = 0;  // bit j: candidate slot j of this group is a singleton row
  if (MODE != MODE_NONE && upd.multi != nullptr) {
    // bucket plan: the singleton information is a bitmap over item ids (1.25 MB at 10 M rows: L2-resident).  Lane i
    // looks up candidates i, i + 64, ... of its tuple (the id loads are coalesced and hit the lines the row gathers
    // above brought in), a ballot turns the bits into wave-uniform masks, and every lane-group picks its candidates'
    // bits out of them -- no per-position flag array exists in HBM any more.
    if (S == 64 && C <= 128) {
      const int c0 = lane, c1 = lane + 64;
      const int64_t ia = ids[c0 < C ? c0 : 0];
      const int64_t ib = ids[c1 < C ? c1 : 0];
      const uint32_t wa = upd.multi[ia >> 5];
      const uint32_t wb = C > 64 ? upd.multi[ib >> 5] : 0u;
      const uint64_t m_lo = __ballot(c0 < C && ((wa >> (ia & 31)) & 1u));
      const uint64_t m_hi = __ballot(c1 < C && ((wb >> (ib & 31)) & 1u));
#pragma unroll
      for (int j = 0; j < CPL; ++j) {
        const int c = j * GS + grp;
        const uint64_t bit = (c < 64 ? m_lo : m_hi) >> (c & 63);
        if (tv && c < C && !(bit & 1ull)) smask |= 1u << j;
      }
    } else {
#pragma unroll
      for (int j = 0; j < CPL; ++j) {
        const int c = j * GS + grp;
        const int64_t id = ids[c < C ? c : 0];
        if (tv && c < C && !((upd.multi[id >> 5] >> (id & 31)) & 1u)) smask |= 1u << j;
      }
    }
  } else if (MODE != MODE_NONE) {
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

}  // namespace rc
