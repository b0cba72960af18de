// numeric_grads.hpp -- weight gradients of a context model's numeric fields: the body of rc_numeric_field_grads (fm_bce.hip), shared
// with the small route's row-sums launch (small_step.hip), in which it rides as extra workgroups.
// A numeric field is nn.Linear(1, d, bias=False) applied to the feature's value (models/context/FM.py:38-41,47-48); autograd's
// Linear backward gives dW[:, 0] = sum_n x[n] * gV[n, f, :] and, for the first-order Linear(1, 1), dw1 = sum_n x[n] * gL[n, f].
// Weighted column sums over a strided slice of the per-occurrence gradient blocks: chunks of kNumericChunk rows per workgroup,
// every lane-group its rows in ascending order, the groups combined in a fixed order through LDS, the chunks by a second
// launch in ascending order (a batch of up to kNumericChunk rows: one launch writes the gradients themselves).  No atomics.
#pragma once
#include "common.hpp"

namespace rc {

constexpr int kMaxFields = 48;
constexpr int kNumericChunk = 1024;

// the value of a numeric field ('*_f' features, models/context/FM.py:47-48: feed_dict[f].float()) at batch position i
__device__ __forceinline__ float field_value(int kind, const void* p, int64_t i) {
  if (kind == RC_FIELD_F32) return static_cast<const float*>(p)[i];
  if (kind == RC_FIELD_F64) return (float)static_cast<const double*>(p)[i];
  return (float)static_cast<const int64_t*>(p)[i];
}

struct NumericSlot {       // one numeric field
  const void* values;
  float* dW;               // [d]
  float* dw1;              // [1]
  int kind, per_row, field;
};
// The FM pairwise term's backward folded into a consumer of the field vectors' gradient (rc_small_row_sums_planned): the gradient
// row of occurrence o = r F + f is  base[o] + g[r] * (S[r] - V[o])  (fm2_bwd_kernel's expression; base: the gradient through the
// other consumer of the field vectors -- DeepFM's deep tower -- or nothing), formed in registers where the row is read.
struct FmTap {
  const float* V;          // [n, F, d] the stacked field vectors | null: no FM term
  const float* S;          // [n, d] their sum over the fields (the gather wrote it)
  const float* g;          // [n] d loss / d fm
  uint32_t F;
  uint32_t magic_F;        // ceil(2^32 / F) (0: F = 1): o / F by one v_mul_hi_u32 for o < 32,768 occurrences
};
__device__ __forceinline__ float4 fm_tap4(const float4& base, float g, const float4& s, const float4& x) {
  return make_float4(base.x + g * (s.x - x.x), base.y + g * (s.y - x.y), base.z + g * (s.z - x.z), base.w + g * (s.w - x.w));
}

struct NumericCommon {
  const float* gV;         // [n, F, d] | null
  const float* gL;         // [n, F] | null
  FmTap fm;                // gV's rows take the FM term's backward on top (fm.V != null; gV may then be null)
  float* part;             // [chunks][n_numeric][d + 1] (several chunks only)
  int64_t n;               // B * C
  int n_numeric, F, C, d;
};

// chunk `chunk` of numeric slot j by a workgroup of NT threads, U rows per lane-group requested together; direct: the workgroup
// holds the whole batch and writes the gradients themselves
// split / n_splits: this workgroup forms columns [split, split + 1) * d / n_splits only (disjoint outputs: no combine across the
// splits) -- fewer lanes per row, more rows in flight per load instruction; split 0 also forms the first-order weight's sum
template <int VEC, int NT, int U>
__device__ __forceinline__ void numeric_slot_grads(const NumericSlot& sl, const NumericCommon& a, int j, uint32_t chunk, bool direct,
                                                   int split = 0, int n_splits = 1) {
  __shared__ float red[NT * VEC];
  __shared__ float red1[NT];
  const int f = sl.field;
  const int dq_all = a.d / VEC;
  const int q_lo = (int)((int64_t)split * dq_all / n_splits), q_hi = (int)((int64_t)(split + 1) * dq_all / n_splits);
  const int dq = q_hi - q_lo;
  if (dq <= 0) return;                            // (workgroup-uniform)
  const int lpr = dq < NT ? dq : NT;              // lanes per row
  const int slots = NT / lpr;                     // rows in flight per step
  const int tid = threadIdx.x, l = tid % lpr, rs = tid / lpr;
  const bool live = rs < slots;
  const uint32_t r0 = chunk * (uint32_t)kNumericChunk;     // (n < 2^31: 32-bit row arithmetic, no 64-bit division per row)
  const uint32_t r1 = direct ? (uint32_t)a.n : ((int64_t)r0 + kNumericChunk < a.n ? r0 + kNumericChunk : (uint32_t)a.n);
  const int kind = sl.kind;
  const uint32_t cdiv = sl.per_row ? (uint32_t)a.C : 1u;   // a per-row feature's value sits at row / C
  const void* xs = sl.values;
  float* part = a.part + ((size_t)chunk * a.n_numeric + j) * (a.d + 1);
  for (int c0 = 0; c0 < dq; c0 += lpr) {       // (one trip unless d > NT * VEC; workgroup-uniform)
    const int cq = c0 + l < dq ? c0 + l : dq - 1;
    const bool col = c0 + l < dq;
    float acc[VEC];
#pragma unroll
    for (int c = 0; c < VEC; ++c) acc[c] = 0.f;
    float acc1 = 0.f;
    const bool first = c0 == 0 && l == 0 && split == 0 && a.gL != nullptr;   // this lane also forms the first-order weight's sum
    if (live && (a.gV || (VEC == 4 && a.fm.V))) {
      for (uint32_t r = r0 + rs; r < r1; r += (uint32_t)slots * U) {
        float x[U], g1[U];
        float gv[U][VEC];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const uint32_t rr = r + (uint32_t)u * slots;
          const bool in = rr < r1;
          const uint32_t ra = in ? rr : r0;
          x[u] = in ? field_value(kind, xs, cdiv == 1u ? ra : ra / cdiv) : 0.f;
          const size_t at = ((size_t)ra * a.F + f) * a.d + (size_t)(q_lo + cq) * VEC;
          if (VEC == 4) {
            float4 t = a.gV ? *reinterpret_cast<const float4*>(a.gV + at) : make_float4(0.f, 0.f, 0.f, 0.f);
            if (a.fm.V)
              t = fm_tap4(t, a.fm.g[ra], *reinterpret_cast<const float4*>(a.fm.S + (size_t)ra * a.d + (size_t)(q_lo + cq) * VEC),
                          *reinterpret_cast<const float4*>(a.fm.V + at));
            gv[u][0] = t.x; gv[u][1 % VEC] = t.y; gv[u][2 % VEC] = t.z; gv[u][3 % VEC] = t.w;
          } else {
            gv[u][0] = a.gV[at];
          }
          g1[u] = (first && in) ? a.gL[(size_t)rr * a.F + f] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
          for (int c = 0; c < VEC; ++c) acc[c] += x[u] * gv[u][c];
          acc1 += x[u] * g1[u];
        }
      }
    } else if (live && first) {   // (only the first-order family reached the loss)
      for (uint32_t r = r0 + rs; r < r1; r += slots) acc1 += field_value(kind, xs, cdiv == 1u ? r : r / cdiv) * a.gL[(size_t)r * a.F + f];
    }
    __syncthreads();   // (the previous trip's reads of red[])
#pragma unroll
    for (int c = 0; c < VEC; ++c) red[tid * VEC + c] = acc[c];
    red1[tid] = acc1;
    __syncthreads();
    // the slots' sums in a fixed order, two levels: groups of eight consecutive slots (slot 8 g: 8 g + 1 .. 8 g + 7 added in order),
    // then slot 0 adds the group sums in order -- 7 + slots / 8 dependent LDS reads instead of slots - 1
    if (live && (rs & 7) == 0) {
      float t[VEC];
#pragma unroll
      for (int c = 0; c < VEC; ++c) t[c] = red[tid * VEC + c];
      float t1 = red1[tid];
      for (int q = rs + 1; q < rs + 8 && q < slots; ++q) {
#pragma unroll
        for (int c = 0; c < VEC; ++c) t[c] += red[(q * lpr + l) * VEC + c];
        t1 += red1[q * lpr + l];
      }
#pragma unroll
      for (int c = 0; c < VEC; ++c) red[tid * VEC + c] = t[c];
      red1[tid] = t1;
    }
    __syncthreads();
    if (live && rs == 0) {
      float t[VEC];
#pragma unroll
      for (int c = 0; c < VEC; ++c) t[c] = red[l * VEC + c];
      float t1 = red1[l];
      for (int q = 8; q < slots; q += 8) {   // fixed order: group 0, 1, ...
#pragma unroll
        for (int c = 0; c < VEC; ++c) t[c] += red[(q * lpr + l) * VEC + c];
        t1 += red1[q * lpr + l];
      }
      if ((a.gV || (VEC == 4 && a.fm.V)) && col) {
#pragma unroll
        for (int c = 0; c < VEC; ++c) {
          if (direct) sl.dW[(q_lo + cq) * VEC + c] = t[c];
          else part[(q_lo + cq) * VEC + c] = t[c];
        }
      }
      if (first) {
        if (direct) sl.dw1[0] = t1;
        else part[a.d] = t1;
      }
    }
  }
}

}  // namespace rc
