// sas_attn_reg.hpp -- causal self-attention of one (sequence, head) by ONE wave with every operand, score and
// probability in registers: no LDS tile, no workgroup barrier.  Included by sasrec_batch.hip after SbAttnArgs.
// Reference semantics: utils/layers.py:34-63 (MultiHeadAttention.scaled_dot_product_attention: QK^T / sqrt(d_k),
// causal mask, softmax over the keys, . V), one head = d_k consecutive columns of the q / k / v rows.
//
// Tiles are 16 x 16 on v_mfma_f32_16x16x4_f32 (lane l: A[i = l & 15][k = l >> 4], B[k = l >> 4][j = l & 15],
// C[4 (l >> 4) + r][l & 15], r = 0..3), which fits d_k = 16 and the 1..64-row histories exactly where the 32 x 32 form
// wastes half of every N = d_k product and three quarters of a 16-row history.  Two operand layouts per array X
// (lane = (i, g) = (l & 15, l >> 4)):
//   rows  X[16 t + i][16 c + 4 g + e], e = 0..3  -- one float4 per lane, tile and 16-column chunk.  The contraction
//         over the head's columns may run in any order, so MFMA step e pairs k-index g with column 16 c + 4 g + e.
//   cols  X[16 t + 4 g + r][16 c + i], r = 0..3  -- the A operand of the "weights . X" products: the weights come
//         straight out of a score tile's accumulator (row index 4 g + r of the tile = the k-index g of step r).
// S^T = K Q^T puts a query in a lane's column: softmax needs the lane's own registers and two cross-lane steps, and
// P^T is already the B operand of ctx^T = V^T P^T, whose accumulator is a float4 of ctx[query][16 c + 4 g ..].
// The backward needs the weights keyed by query as well as by key (dQ vs dK, dV): the S = Q K^T tiles are the same
// registers with A and B swapped; the per-query statistics (max, 1 / sum, D = sum_j P dP) cross over through 768 bytes
// of LDS owned by the wave (LDS operations of one wave execute in order).
#pragma once
#include "sas_mma.hpp"

namespace rc {

template <int D, int NT, int NC>
struct SasRegRows {
  float v[NT][NC][4];
  __device__ __forceinline__ void load(const float* __restrict__ X, int64_t r0, int n, int hc) {
    const int lane = threadIdx.x & 63, i = lane & 15, g = lane >> 4;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int row = 16 * t + i;
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row < n) x = *reinterpret_cast<const float4*>(X + (size_t)(r0 + row) * D + hc + 16 * c + 4 * g);
        v[t][c][0] = x.x; v[t][c][1] = x.y; v[t][c][2] = x.z; v[t][c][3] = x.w;
      }
    }
  }
};

template <int D, int NT, int NC>
struct SasRegCols {
  float v[NT][NC][4];
  __device__ __forceinline__ void load(const float* __restrict__ X, int64_t r0, int n, int hc) {
    const int lane = threadIdx.x & 63, i = lane & 15, g = lane >> 4;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * t + 4 * g + r;
#pragma unroll
        for (int c = 0; c < NC; ++c) v[t][c][r] = row < n ? X[(size_t)(r0 + row) * D + hc + 16 * c + i] : 0.f;
      }
  }
};

__device__ __forceinline__ float sas_groups_max(float x) {
  x = fmaxf(x, __shfl_xor(x, 16, 64));
  return fmaxf(x, __shfl_xor(x, 32, 64));
}
__device__ __forceinline__ float sas_groups_sum(float x) {
  x += __shfl_xor(x, 16, 64);
  return x + __shfl_xor(x, 32, 64);
}

// score tiles of query tile QT against key tiles 0..QT in the transposed layout: s[kt][r] = X_k[16 kt + 4 g + r] . X_q[16 QT + i]
template <int NT, int NC, int QT, typename RowsK, typename RowsQ>
__device__ __forceinline__ void sas_reg_scores_t(const RowsK& k, const RowsQ& q, sas_f32x4 (&s)[NT]) {
#pragma unroll
  for (int kt = 0; kt < NT; ++kt) s[kt] = sas_zero4();
#pragma unroll
  for (int c = 0; c < NC; ++c)
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int kt = 0; kt < NT; ++kt)
        if (kt <= QT) s[kt] = sas_mfma16(k.v[kt][c][e], q.v[QT][c][e], s[kt]);
}

// causal softmax over the keys of the lane's query (column i of tile QT): s -> probabilities, returns max and 1 / sum
template <int NT, int QT>
__device__ __forceinline__ void sas_reg_softmax_t(sas_f32x4 (&s)[NT], float sqrt_dk, float* m_out, float* rz_out) {
  const int lane = threadIdx.x & 63, i = lane & 15, g = lane >> 4;
  float m = -INFINITY;
#pragma unroll
  for (int kt = 0; kt < NT; ++kt)
    if (kt <= QT)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float x = sas_div_scale(s[kt][r], sqrt_dk);
        s[kt][r] = x;
        if (kt < QT || 4 * g + r <= i) m = fmaxf(m, x);
      }
  m = sas_groups_max(m);
  float z = 0.f;
#pragma unroll
  for (int kt = 0; kt < NT; ++kt)
    if (kt <= QT)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float e = (kt < QT || 4 * g + r <= i) ? expf(s[kt][r] - m) : 0.f;
        s[kt][r] = e;
        z += e;
      }
  z = sas_groups_sum(z);
  const float rz = 1.0f / z;
#pragma unroll
  for (int kt = 0; kt < NT; ++kt)
    if (kt <= QT)
#pragma unroll
      for (int r = 0; r < 4; ++r) s[kt][r] *= rz;
  *m_out = m;
  *rz_out = rz;
}

// out^T[16 c + 4 g + r'][column i] = sum over tiles kt <= QT, rows 4 g + r: X[16 kt + 4 g + r][16 c + i] * w[kt][r]
template <int NT, int NC, int QT, typename Cols>
__device__ __forceinline__ void sas_reg_weighted_t(const Cols& x, const sas_f32x4 (&w)[NT], sas_f32x4 (&o)[NC]) {
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    sas_f32x4 o0 = sas_zero4(), o1 = sas_zero4();   // two chains: a dependent MFMA waits 40 cycles, an independent one 32
#pragma unroll
    for (int kt = 0; kt < NT; ++kt)
      if (kt <= QT) {
        o0 = sas_mfma16(x.v[kt][c][0], w[kt][0], o0);
        o1 = sas_mfma16(x.v[kt][c][1], w[kt][1], o1);
        o0 = sas_mfma16(x.v[kt][c][2], w[kt][2], o0);
        o1 = sas_mfma16(x.v[kt][c][3], w[kt][3], o1);
      }
    o[c] = o0 + o1;
  }
}

template <int D, int NC>
__device__ __forceinline__ void sas_reg_store_rows(float* __restrict__ Y, int64_t r0, int n, int hc, int tile,
                                                   const sas_f32x4 (&o)[NC]) {
  const int lane = threadIdx.x & 63, i = lane & 15, g = lane >> 4;
  const int row = 16 * tile + i;
  if (row < n)
#pragma unroll
    for (int c = 0; c < NC; ++c)
      *reinterpret_cast<float4*>(Y + (size_t)(r0 + row) * D + hc + 16 * c + 4 * g) = make_float4(o[c][0], o[c][1], o[c][2], o[c][3]);
}

template <int NT, int NC, int QT, int D>
struct SasRegFwdStep {
  template <typename Rows, typename Cols>
  static __device__ __forceinline__ void run(const SbAttnArgs& a, const Rows& q, const Rows& k, const Cols& vt, int64_t r0, int n,
                                             int hc, float sqrt_dk) {
    sas_f32x4 s[NT];
    sas_reg_scores_t<NT, NC, QT>(k, q, s);
    float m, rz;
    sas_reg_softmax_t<NT, QT>(s, sqrt_dk, &m, &rz);
    sas_f32x4 o[NC];
    sas_reg_weighted_t<NT, NC, QT>(vt, s, o);
    sas_reg_store_rows<D, NC>(a.ctx, r0, n, hc, QT, o);
    if constexpr (QT + 1 < NT) SasRegFwdStep<NT, NC, QT + 1, D>::run(a, q, k, vt, r0, n, hc, sqrt_dk);
  }
};

template <int D, int NT, int NC>
__device__ __forceinline__ void sas_reg_attn_fwd_item(const SbAttnArgs& a, int64_t r0, int n, int hc, float sqrt_dk) {
  SasRegRows<D, NT, NC> q, k;
  SasRegCols<D, NT, NC> vt;
  k.load(a.k, r0, n, hc);
  q.load(a.q, r0, n, hc);
  vt.load(a.v, r0, n, hc);
  SasRegFwdStep<NT, NC, 0, D>::run(a, q, k, vt, r0, n, hc, sqrt_dk);
}

// ---- backward ------------------------------------------------------------------------------------------------------
// phase 1, per query tile (transposed layout): P^T, dP^T = V dCtx^T, D = sum_keys P dP, dS^T = P (dP - D) / sqrt(dk),
// dQ^T = K^T dS^T; the query's (max, 1 / sum, D) go to the wave's LDS strip.
template <int NT, int NC, int QT, int D>
struct SasRegBwdQ {
  template <typename Rows, typename Cols>
  static __device__ __forceinline__ void run(const SbAttnArgs& a, const Rows& q, const Rows& k, const Rows& v, const Rows& gd,
                                             const Cols& kt_, float* st, int64_t r0, int n, int hc, float sqrt_dk) {
    const int lane = threadIdx.x & 63, i = lane & 15, g = lane >> 4;
    sas_f32x4 s[NT], dp[NT];
    sas_reg_scores_t<NT, NC, QT>(k, q, s);
    sas_reg_scores_t<NT, NC, QT>(v, gd, dp);
    float m, rz;
    sas_reg_softmax_t<NT, QT>(s, sqrt_dk, &m, &rz);
    float dot = 0.f;
#pragma unroll
    for (int kt = 0; kt < NT; ++kt)
      if (kt <= QT)
#pragma unroll
        for (int r = 0; r < 4; ++r) dot = fmaf(s[kt][r], dp[kt][r], dot);   // P = 0 above the diagonal
    dot = sas_groups_sum(dot);
    if (g == 0) {
      st[16 * QT + i] = m;
      st[64 + 16 * QT + i] = rz;
      st[128 + 16 * QT + i] = dot;
    }
#pragma unroll
    for (int kt = 0; kt < NT; ++kt)
      if (kt <= QT)
#pragma unroll
        for (int r = 0; r < 4; ++r) s[kt][r] = sas_div_scale(s[kt][r] * (dp[kt][r] - dot), sqrt_dk);
    sas_f32x4 o[NC];
    sas_reg_weighted_t<NT, NC, QT>(kt_, s, o);
    sas_reg_store_rows<D, NC>(a.dq, r0, n, hc, QT, o);
    if constexpr (QT + 1 < NT) SasRegBwdQ<NT, NC, QT + 1, D>::run(a, q, k, v, gd, kt_, st, r0, n, hc, sqrt_dk);
  }
};

// phase 2, per key tile (lane's column = a key): for every query tile qt >= KT the S and dP tiles with the operands
// swapped, P and dS from the stored statistics, dV^T += dCtx^T P, dK^T += Q^T dS.
template <int NT, int NC, int KT, int D>
struct SasRegBwdK {
  template <typename Rows, typename Cols>
  static __device__ __forceinline__ void run(const SbAttnArgs& a, const Rows& q, const Rows& k, const Rows& v, const Rows& gd,
                                             const Cols& qt_, const Cols& gt_, const float* st, int64_t r0, int n, int hc,
                                             float sqrt_dk) {
    const int lane = threadIdx.x & 63, i = lane & 15, g = lane >> 4;
    sas_f32x4 dk_[NC], dv_[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) dk_[c] = dv_[c] = sas_zero4();
#pragma unroll
    for (int qt = 0; qt < NT; ++qt)
      if (qt >= KT) {
        sas_f32x4 s = sas_zero4(), dp = sas_zero4();
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            s = sas_mfma16(q.v[qt][c][e], k.v[KT][c][e], s);      // s[r] = S[query 16 qt + 4 g + r][key 16 KT + i]
            dp = sas_mfma16(gd.v[qt][c][e], v.v[KT][c][e], dp);
          }
        const float4 m4 = *reinterpret_cast<const float4*>(st + 16 * qt + 4 * g);
        const float4 z4 = *reinterpret_cast<const float4*>(st + 64 + 16 * qt + 4 * g);
        const float4 d4 = *reinterpret_cast<const float4*>(st + 128 + 16 * qt + 4 * g);
        const float mm[4] = {m4.x, m4.y, m4.z, m4.w}, zz[4] = {z4.x, z4.y, z4.z, z4.w}, dd[4] = {d4.x, d4.y, d4.z, d4.w};
        float p[4], ds[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int query = 16 * qt + 4 * g + r;
          const bool on = query < n && (qt > KT || i <= 4 * g + r);   // rows past the history feed the contraction: exact zeros
          p[r] = on ? expf(sas_div_scale(s[r], sqrt_dk) - mm[r]) * zz[r] : 0.f;
          ds[r] = sas_div_scale(p[r] * (dp[r] - dd[r]), sqrt_dk);
        }
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            dv_[c] = sas_mfma16(gt_.v[qt][c][r], p[r], dv_[c]);
            dk_[c] = sas_mfma16(qt_.v[qt][c][r], ds[r], dk_[c]);
          }
      }
    sas_reg_store_rows<D, NC>(a.dk, r0, n, hc, KT, dk_);
    sas_reg_store_rows<D, NC>(a.dv, r0, n, hc, KT, dv_);
    if constexpr (KT + 1 < NT) SasRegBwdK<NT, NC, KT + 1, D>::run(a, q, k, v, gd, qt_, gt_, st, r0, n, hc, sqrt_dk);
  }
};

template <int D, int NT, int NC>
__device__ __forceinline__ void sas_reg_attn_bwd_item(const SbAttnArgs& a, float* st, int64_t r0, int n, int hc, float sqrt_dk) {
  SasRegRows<D, NT, NC> q, k, v, gd;
  k.load(a.k, r0, n, hc);
  q.load(a.q, r0, n, hc);
  v.load(a.v, r0, n, hc);
  gd.load(a.dctx, r0, n, hc);
  {
    SasRegCols<D, NT, NC> kt_;
    kt_.load(a.k, r0, n, hc);
    SasRegBwdQ<NT, NC, 0, D>::run(a, q, k, v, gd, kt_, st, r0, n, hc, sqrt_dk);
  }
  __builtin_amdgcn_wave_barrier();   // the strip is written and read by this wave only; keep the order in the schedule
  {
    SasRegCols<D, NT, NC> qt_, gt_;
    qt_.load(a.q, r0, n, hc);
    gt_.load(a.dctx, r0, n, hc);
    SasRegBwdK<NT, NC, 0, D>::run(a, q, k, v, gd, qt_, gt_, st, r0, n, hc, sqrt_dk);
  }
}

// One wave per (sequence, head); a workgroup's four waves are four consecutive (sequence, head) items (the four heads of
// one sequence at n_heads = 4: the same 256-byte rows).  One workgroup per four items and no loop: the dispatcher
// balances the very uneven items (1 tile pair at <= 16 rows, 10 at > 48).
template <int D, int DK, bool BWD>
__global__ __launch_bounds__(256, 2) void sb_attn_reg_kernel(SbAttnArgs a) {
  constexpr int NC = DK / 16, MAXNT = 4 / NC;
  __shared__ __align__(16) float stats[4][192];
  const int wave = threadIdx.x >> 6;
  const int item = (int)blockIdx.x * 4 + wave;
  if (item >= a.B * a.n_heads) return;
  const int w = item / a.n_heads, hh = item % a.n_heads;
  const int b = a.seq_list ? a.seq_list[w] : w;
  const int n = sb_len(a.lengths, b, a.L);
  if (n == 0) return;
  const int64_t r0 = a.off[b];
  const int hc = hh * DK;
  const float sqrt_dk = sqrtf((float)DK);
  const int nt = (n + 15) >> 4;
  float* st = stats[wave];
  (void)st;
#define RC_SAS_REG_CASE(NT_)                                                              \
  if constexpr (NT_ <= MAXNT) {                                                           \
    if (nt == NT_) {                                                                      \
      if constexpr (BWD) sas_reg_attn_bwd_item<D, NT_, NC>(a, st, r0, n, hc, sqrt_dk);    \
      else sas_reg_attn_fwd_item<D, NT_, NC>(a, r0, n, hc, sqrt_dk);                      \
      return;                                                                             \
    }                                                                                     \
  }
  RC_SAS_REG_CASE(1)
  RC_SAS_REG_CASE(2)
  RC_SAS_REG_CASE(3)
  RC_SAS_REG_CASE(4)
#undef RC_SAS_REG_CASE
}

// the shapes the register kernels take: d_k in {16, 32, 64} with ceil(L / 16) * d_k / 16 <= 4 tiles of operands per array
inline bool sas_reg_attn_fits(int D, int n_heads, int L) {
  if (D % n_heads != 0) return false;
  const int dk = D / n_heads;
  if (dk != 16 && dk != 32 && dk != 64) return false;
  return ((L + 15) / 16) * (dk / 16) <= 4;
}

}  // namespace rc
