// seq_layers.hip -- the layers of the reference's sequence encoder as shape-generic kernels: any emb_size, any head
// count, any number of blocks, history up to 1,024 positions, training-mode dropout.
//
// Reference: utils/layers.py MultiHeadAttention.scaled_dot_product_attention :52-63 (scores / sqrt(d_k), mask -> -inf, softmax,
// NaN -> 0, @ V; no output projection :44-49), TransformerLayer.forward :102-118 (LayerNorm(dropout(branch) + residual) twice),
// models/sequential/SASRec.py:58-76 (item + position embeddings of a right-padded history, position id = length - index, the
// causal mask, padded rows zeroed).
//
// The register-resident / MFMA encoders of sasrec_batch.hip and sasrec.hip cover emb_size 32 / 64 and histories up to 64
// (128) positions -- SASRec's published settings.  Everything else used to run the plugin's torch layers; it runs here
// instead: the d x d projections and the FFN are rc_linear_fwd / rc_linear_bwd (mlp.hip, fp32 MFMA GEMMs of any shape), and
// this file holds what is not a GEMM.  Layout: the padded batch [B, L, d] as B * L rows; a row (b, i) is VALID iff
// i < min(len_b, L).  Rows that are not valid hold zeros at every layer boundary and zero gradients (the reference computes
// garbage there, multiplies it by 0 at the end -- SASRec.py:74 -- and no valid row ever attends to one: padding is on the
// right and the mask is causal), so GEMMs over all B * L rows add nothing for them.  `off` [B + 1] = exclusive prefix sums of
// min(len_b, L): the compact row index off[b] + i keys the dropout mask exactly as rc_sasrec_batch_fwd_dropout does
// (oracle/sasrec_oracle.dropout_keep), so both encoders draw the same mask from the same seed.
//
// One wave per attention row (query row forward / for dQ, key row for dK and dV): scores one lane per key, the weighted sums
// one lane per feature with the probabilities broadcast lane by lane; probabilities are recomputed in the backward pass from
// the saved log-sum-exp, nothing of size L x L ever exists in memory.  Fixed summation orders, no atomics.
#include "common.hpp"
#include "philox.hpp"

namespace rc {

constexpr float kSeqLnEps = 1e-5f;       // nn.LayerNorm default
constexpr int kSeqMaxL = 1024;           // 16 keys per lane
constexpr int kSeqMaxDk = 256;           // per-head width the LDS strips hold
constexpr int kSeqMaxD = 1024;           // LayerNorm row width (four float4 per lane)

// ---- off[b] = sum_{b' < b} min(len_b', L) --------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void seq_offsets_kernel(const int64_t* __restrict__ lengths, int64_t B, int L,
                                                           int32_t* __restrict__ off) {
  __shared__ int32_t wsum[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int32_t carry = 0;
  for (int64_t base = 0; base < B; base += 1024) {
    const int64_t b = base + tid;
    int64_t len = b < B ? lengths[b] : 0;
    const int32_t v = (int32_t)(len < 0 ? 0 : (len > L ? L : len));
    int32_t x = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int32_t y = __shfl_up(x, o, 64);
      if (lane >= o) x += y;
    }
    if (lane == 63) wsum[wave] = x;
    __syncthreads();
    int32_t before = carry, total = 0;
    for (int q = 0; q < 16; ++q) {
      if (q < wave) before += wsum[q];
      total += wsum[q];
    }
    if (b < B) off[b] = before + x - v;
    carry += total;
    __syncthreads();
  }
  if (tid == 0) off[B] = carry;
}

// ---- X[b, i, :] = item_emb[hist[b, i]] + pos_emb[len_b - i] on valid rows, 0 elsewhere (SASRec.py:58-66) ---------------------
template <int VEC>
__global__ __launch_bounds__(kBlock) void seq_embed_kernel(const float* __restrict__ item_emb, const float* __restrict__ pos_emb,
                                                           const int64_t* __restrict__ hist, const int64_t* __restrict__ lengths,
                                                           int64_t B, int L, int d, float* __restrict__ X) {
  const int dq = d / VEC;
  const int64_t total = B * L * dq;
  for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < total; e += (int64_t)gridDim.x * kBlock) {
    const int64_t row = e / dq;
    const int q = (int)(e - row * dq);
    const int64_t b = row / L;
    const int i = (int)(row - b * L);
    const int64_t len = lengths[b];
    const bool valid = i < len;
    if (VEC == 4) {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (valid) {
        const float4 a = reinterpret_cast<const float4*>(item_emb)[hist[row] * dq + q];
        const float4 p = reinterpret_cast<const float4*>(pos_emb)[(len - i) * dq + q];
        v = make_float4(a.x + p.x, a.y + p.y, a.z + p.z, a.w + p.w);
      }
      reinterpret_cast<float4*>(X)[e] = v;
    } else {
      X[e] = valid ? item_emb[hist[row] * dq + q] + pos_emb[(len - i) * dq + q] : 0.f;
    }
  }
}

// ---- hv[b] = X[b, len_b - 1] (SASRec.py:76) and its backward: dX = 0 except those rows ---------------------------------------
__global__ __launch_bounds__(kBlock) void seq_pick_last_kernel(const float* __restrict__ X, const int64_t* __restrict__ lengths, int64_t B,
                                                               int L, int d, float* __restrict__ hv) {
  for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < B * d; e += (int64_t)gridDim.x * kBlock) {
    const int64_t b = e / d;
    const int k = (int)(e - b * d);
    int64_t len = lengths[b];
    len = len > L ? L : len;
    hv[e] = len >= 1 ? X[((size_t)b * L + (len - 1)) * d + k] : 0.f;
  }
}
__global__ __launch_bounds__(kBlock) void seq_pick_last_bwd_kernel(const float* __restrict__ dhv, const int64_t* __restrict__ lengths,
                                                                   int64_t B, int L, int d, float* __restrict__ dX) {
  const int64_t total = B * L * d;
  for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < total; e += (int64_t)gridDim.x * kBlock) {
    const int64_t row = e / d;
    const int k = (int)(e - row * d);
    const int64_t b = row / L;
    const int i = (int)(row - b * L);
    int64_t len = lengths[b];
    len = len > L ? L : len;
    dX[e] = (i == len - 1) ? dhv[b * d + k] : 0.f;
  }
}

// ---- gradient of the position table: grad_pos[p] = sum_b dX[b, len_b - p] over the sequences with min(len_b, L) >= p ----------
// (SASRec.py:64: position id = length - index on valid rows; autograd's embedding backward behind it.)  One workgroup per
// position, lane-groups stride over the sequences in ascending order and are combined in a fixed order: any row width.
template <int VEC>
__global__ __launch_bounds__(kBlock) void seq_pos_grad_kernel(const float* __restrict__ dX, const int64_t* __restrict__ lengths, int64_t B,
                                                              int L, int d, int n_pos, float* __restrict__ grad_pos) {
  __shared__ float red[kBlock * VEC];
  const int p = blockIdx.x;
  const int dq = d / VEC;
  const int lpr = dq < kBlock ? dq : kBlock;
  const int slots = kBlock / lpr;
  const int tid = threadIdx.x, l = tid % lpr, rs = tid / lpr;
  for (int c0 = 0; c0 < dq; c0 += lpr) {       // (one trip unless d > 256 * VEC; workgroup-uniform)
    const int cq = c0 + l < dq ? c0 + l : dq - 1;
    float acc[VEC];
#pragma unroll
    for (int c = 0; c < VEC; ++c) acc[c] = 0.f;
    if (rs < slots && p >= 1 && p <= L) {
      for (int64_t b = rs; b < B; b += slots) {
        int64_t len = lengths[b];
        len = len > L ? L : len;
        if (len >= p) {
          const float* src = dX + ((size_t)b * L + (size_t)(len - p)) * d + (size_t)cq * VEC;
          if (VEC == 4) {
            const float4 t = *reinterpret_cast<const float4*>(src);
            acc[0] += t.x; acc[1 % VEC] += t.y; acc[2 % VEC] += t.z; acc[3 % VEC] += t.w;
          } else {
            acc[0] += src[0];
          }
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < VEC; ++c) red[tid * VEC + c] = acc[c];
    __syncthreads();
    if (rs == 0 && c0 + l < dq) {
#pragma unroll
      for (int c = 0; c < VEC; ++c) {
        float t = red[l * VEC + c];
        for (int q = 1; q < slots; ++q) t += red[(q * lpr + l) * VEC + c];
        grad_pos[(size_t)p * d + cq * VEC + c] = t;
      }
    }
  }
}

// ---- attention ---------------------------------------------------------------------------------------------------------
struct SeqAttnArgs {
  const float* Q;          // [B * L, H * dk], head h in columns [h * dk, (h + 1) * dk)
  const float* K;
  const float* V;
  const int32_t* off;      // [B + 1] | null (every row valid)
  const uint8_t* mask;     // [mask_batch ? B : 1][L][L], 0 = masked out | null
  int mask_batch, causal;
  int64_t B;
  int L, H, dk;
  float inv_sqrt;
  float* ctx;              // fwd out [B * L, H * dk]
  float* lse;              // [B, H, L] log-sum-exp of a row's scaled scores (0 for a row with no visible key)
  // backward
  const float* dctx;
  float* Dv;               // [B, H, L] sum_j p_ij dP_ij
  float* dQ;
  float* dK;
  float* dV;
};

// dot product of two dk-long vectors, one in LDS (broadcast reads), one a global row read by this lane alone; the row's loads
// are requested eight (four) float4 at a time -- one at a time the wave waits a memory latency per 16 bytes
__device__ __forceinline__ float seq_dot(const float* __restrict__ s, const float* __restrict__ g, int dk) {
  float acc = 0.f;
  if ((dk & 3) == 0 && (reinterpret_cast<uintptr_t>(g) & 15) == 0) {
    const float4* s4 = reinterpret_cast<const float4*>(s);
    const float4* g4 = reinterpret_cast<const float4*>(g);
    const int nq = dk >> 2;
    int q = 0;
    for (; q + 8 <= nq; q += 8) {
      float4 b[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) b[u] = g4[q + u];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const float4 a = s4[q + u];
        acc = fmaf(a.x, b[u].x, acc); acc = fmaf(a.y, b[u].y, acc); acc = fmaf(a.z, b[u].z, acc); acc = fmaf(a.w, b[u].w, acc);
      }
    }
    if (q + 4 <= nq) {
      float4 b[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) b[u] = g4[q + u];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float4 a = s4[q + u];
        acc = fmaf(a.x, b[u].x, acc); acc = fmaf(a.y, b[u].y, acc); acc = fmaf(a.z, b[u].z, acc); acc = fmaf(a.w, b[u].w, acc);
      }
      q += 4;
    }
    for (; q < nq; ++q) {
      const float4 a = s4[q], bb = g4[q];
      acc = fmaf(a.x, bb.x, acc); acc = fmaf(a.y, bb.y, acc); acc = fmaf(a.z, bb.z, acc); acc = fmaf(a.w, bb.w, acc);
    }
  } else {
    for (int k = 0; k < dk; ++k) acc = fmaf(s[k], g[k], acc);
  }
  return acc;
}

// acc += sum_{j < n} w_j * col[j * D], w_j living in lane (j & 63) of wreg[j >> 6]: the column's loads eight at a time, the
// weights broadcast lane by lane (ascending j: a fixed order).  Every lane of the wave calls it.
template <int TMAX>
__device__ __forceinline__ float seq_weighted_col(const float (&wreg)[TMAX], const float* __restrict__ col, int D, int n, float acc) {
#pragma unroll
  for (int t = 0; t < TMAX; ++t) {
    if (t * 64 >= n) break;
    const int jn = n - t * 64 < 64 ? n - t * 64 : 64;
    const float* c = col + (size_t)(t * 64) * D;
    int jj = 0;
    for (; jj + 8 <= jn; jj += 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = c[(size_t)(jj + u) * D];
#pragma unroll
      for (int u = 0; u < 8; ++u) acc = fmaf(__shfl(wreg[t], jj + u, 64), v[u], acc);
    }
    for (; jj < jn; ++jj) acc = fmaf(__shfl(wreg[t], jj, 64), c[(size_t)jj * D], acc);
  }
  return acc;
}

// the (b, h, row) of this wave; false past the end
struct SeqWave {
  int64_t b;
  int h, i, n;
  size_t base;      // element offset of (row 0 of sequence b, column h * dk)
  size_t stat;      // index of (b, h, 0) in lse / Dv
};
__device__ __forceinline__ bool seq_wave(const SeqAttnArgs& a, SeqWave* w) {
  const int64_t wid = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
  if (wid >= a.B * a.H * a.L) return false;
  w->i = (int)(wid % a.L);
  const int64_t bh = wid / a.L;
  w->h = (int)(bh % a.H);
  w->b = bh / a.H;
  w->n = a.off ? a.off[w->b + 1] - a.off[w->b] : a.L;
  w->base = (size_t)w->b * a.L * a.H * a.dk + (size_t)w->h * a.dk;
  w->stat = (size_t)bh * a.L;
  return true;
}
__device__ __forceinline__ bool seq_visible(const SeqAttnArgs& a, int64_t b, int i, int j) {
  return a.mask == nullptr || a.mask[(a.mask_batch ? (size_t)b * a.L * a.L : 0) + (size_t)i * a.L + j] != 0;
}

template <int TMAX>
__global__ __launch_bounds__(kBlock) void seq_attn_fwd_kernel(SeqAttnArgs a) {
  __shared__ __attribute__((aligned(16))) float strip[(kBlock / 64) * kSeqMaxDk];
  const int lane = threadIdx.x & 63;
  float* qs = strip + (threadIdx.x >> 6) * kSeqMaxDk;
  SeqWave w;
  if (!seq_wave(a, &w)) return;            // wave-uniform; only wave-level synchronisation below
  const int D = a.H * a.dk;
  float* out = a.ctx + w.base + (size_t)w.i * D;
  if (w.i >= w.n) {
    for (int k = lane; k < a.dk; k += 64) out[k] = 0.f;
    if (lane == 0) a.lse[w.stat + w.i] = 0.f;
    return;
  }
  for (int k = lane; k < a.dk; k += 64) qs[k] = a.Q[w.base + (size_t)w.i * D + k];
  __builtin_amdgcn_wave_barrier();
  const int jmax = a.causal ? w.i + 1 : w.n;
  float s[TMAX];
  float m = -INFINITY;
#pragma unroll
  for (int t = 0; t < TMAX; ++t) {
    const int j = t * 64 + lane;
    s[t] = -INFINITY;
    if (j < jmax && seq_visible(a, w.b, w.i, j)) s[t] = seq_dot(qs, a.K + w.base + (size_t)j * D, a.dk) * a.inv_sqrt;
    m = fmaxf(m, s[t]);
  }
  m = wave_allreduce_max(m);
  float l = 0.f;
#pragma unroll
  for (int t = 0; t < TMAX; ++t) {
    s[t] = (m == -INFINITY || s[t] == -INFINITY) ? 0.f : expf(s[t] - m);
    l += s[t];
  }
  l = wave_allreduce_sum(l);
  const float inv_l = l > 0.f ? 1.0f / l : 0.f;      // (a row with no visible key: softmax of all -inf is NaN -> 0, :61)
#pragma unroll
  for (int t = 0; t < TMAX; ++t) s[t] *= inv_l;
  if (lane == 0) a.lse[w.stat + w.i] = l > 0.f ? m + logf(l) : 0.f;
  for (int k0 = 0; k0 < a.dk; k0 += 64) {
    const int k = k0 + lane;
    const float* vcol = a.V + w.base + (k < a.dk ? k : 0);
    const float acc = seq_weighted_col<TMAX>(s, vcol, D, jmax, 0.f);
    if (k < a.dk) out[k] = acc;
  }
}

// dQ and Dv: one wave per query row
template <int TMAX>
__global__ __launch_bounds__(kBlock) void seq_attn_bwd_q_kernel(SeqAttnArgs a) {
  __shared__ __attribute__((aligned(16))) float strip[(kBlock / 64) * 2 * kSeqMaxDk];
  const int lane = threadIdx.x & 63;
  float* qs = strip + (threadIdx.x >> 6) * 2 * kSeqMaxDk;
  float* ds = qs + kSeqMaxDk;
  SeqWave w;
  if (!seq_wave(a, &w)) return;
  const int D = a.H * a.dk;
  float* out = a.dQ + w.base + (size_t)w.i * D;
  if (w.i >= w.n) {
    for (int k = lane; k < a.dk; k += 64) out[k] = 0.f;
    if (lane == 0) a.Dv[w.stat + w.i] = 0.f;
    return;
  }
  for (int k = lane; k < a.dk; k += 64) {
    qs[k] = a.Q[w.base + (size_t)w.i * D + k];
    ds[k] = a.dctx[w.base + (size_t)w.i * D + k];
  }
  __builtin_amdgcn_wave_barrier();
  const int jmax = a.causal ? w.i + 1 : w.n;
  const float lse = a.lse[w.stat + w.i];
  float p[TMAX], dp[TMAX];
  float dsum = 0.f;
#pragma unroll
  for (int t = 0; t < TMAX; ++t) {
    const int j = t * 64 + lane;
    p[t] = dp[t] = 0.f;
    if (j < jmax && seq_visible(a, w.b, w.i, j)) {
      p[t] = expf(seq_dot(qs, a.K + w.base + (size_t)j * D, a.dk) * a.inv_sqrt - lse);
      dp[t] = seq_dot(ds, a.V + w.base + (size_t)j * D, a.dk);
    }
    dsum = fmaf(p[t], dp[t], dsum);
  }
  dsum = wave_allreduce_sum(dsum);
  if (lane == 0) a.Dv[w.stat + w.i] = dsum;
#pragma unroll
  for (int t = 0; t < TMAX; ++t) p[t] = p[t] * (dp[t] - dsum) * a.inv_sqrt;      // dS / sqrt(dk)
  for (int k0 = 0; k0 < a.dk; k0 += 64) {
    const int k = k0 + lane;
    const float* kcol = a.K + w.base + (k < a.dk ? k : 0);
    const float acc = seq_weighted_col<TMAX>(p, kcol, D, jmax, 0.f);
    if (k < a.dk) out[k] = acc;
  }
}

// dK and dV: one wave per key row j; its queries are i in [j, n) under the causal mask, [0, n) otherwise
template <int TMAX>
__global__ __launch_bounds__(kBlock) void seq_attn_bwd_kv_kernel(SeqAttnArgs a) {
  __shared__ __attribute__((aligned(16))) float strip[(kBlock / 64) * 2 * kSeqMaxDk];
  const int lane = threadIdx.x & 63;
  float* ks = strip + (threadIdx.x >> 6) * 2 * kSeqMaxDk;
  float* vs = ks + kSeqMaxDk;
  SeqWave w;
  if (!seq_wave(a, &w)) return;
  const int D = a.H * a.dk;
  const int j = w.i;
  float* outk = a.dK + w.base + (size_t)j * D;
  float* outv = a.dV + w.base + (size_t)j * D;
  if (j >= w.n) {
    for (int k = lane; k < a.dk; k += 64) outk[k] = outv[k] = 0.f;
    return;
  }
  for (int k = lane; k < a.dk; k += 64) {
    ks[k] = a.K[w.base + (size_t)j * D + k];
    vs[k] = a.V[w.base + (size_t)j * D + k];
  }
  __builtin_amdgcn_wave_barrier();
  const int i0 = a.causal ? j : 0;
  const int cnt = w.n - i0;
  float p[TMAX], dsv[TMAX];
#pragma unroll
  for (int t = 0; t < TMAX; ++t) {
    const int i = i0 + t * 64 + lane;
    p[t] = dsv[t] = 0.f;
    if (i < w.n && seq_visible(a, w.b, i, j)) {
      const float pp = expf(seq_dot(ks, a.Q + w.base + (size_t)i * D, a.dk) * a.inv_sqrt - a.lse[w.stat + i]);
      const float dpp = seq_dot(vs, a.dctx + w.base + (size_t)i * D, a.dk);
      p[t] = pp;
      dsv[t] = pp * (dpp - a.Dv[w.stat + i]) * a.inv_sqrt;
    }
  }
  for (int k0 = 0; k0 < a.dk; k0 += 64) {
    const int k = k0 + lane;
    const float* qcol = a.Q + w.base + (size_t)i0 * D + (k < a.dk ? k : 0);
    const float* dcol = a.dctx + w.base + (size_t)i0 * D + (k < a.dk ? k : 0);
    float accv = 0.f, acck = 0.f;     // (both columns in one walk over the queries: two separate walks measured 18 % slower)
#pragma unroll
    for (int t = 0; t < TMAX; ++t) {
      if (t * 64 >= cnt) break;
      const int in = cnt - t * 64 < 64 ? cnt - t * 64 : 64;
      int ii = 0;
      for (; ii + 4 <= in; ii += 4) {
        float dv[4], qv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          dv[u] = dcol[(size_t)(t * 64 + ii + u) * D];
          qv[u] = qcol[(size_t)(t * 64 + ii + u) * D];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          accv = fmaf(__shfl(p[t], ii + u, 64), dv[u], accv);
          acck = fmaf(__shfl(dsv[t], ii + u, 64), qv[u], acck);
        }
      }
      for (; ii < in; ++ii) {
        accv = fmaf(__shfl(p[t], ii, 64), dcol[(size_t)(t * 64 + ii) * D], accv);
        acck = fmaf(__shfl(dsv[t], ii, 64), qcol[(size_t)(t * 64 + ii) * D], acck);
      }
    }
    if (k < a.dk) { outv[k] = accv; outk[k] = acck; }
  }
}

// ---- y = LayerNorm(dropout(A) + R) ---------------------------------------------------------------------------------------
struct SeqLnArgs {
  const float* A;          // branch [rows, d]
  const float* R;          // residual [rows, d] | null
  const float* w;
  const float* b;
  const int32_t* off;      // [rows / L + 1] | null (every row valid, compact index = row)
  int64_t rows;
  int L, d;
  const uint64_t* seed;    // device memory | null (no dropout)
  uint32_t thresh, site;
  float scale;
  float* Y;
  float* xhat;
  float* rstd;
  // backward
  const float* dY;
  float* dA;               // | null
  float* dR;               // | null
  float* part;             // [blocks][2][d] partial dw, db
};

// validity and compact index of a row
__device__ __forceinline__ bool seq_row(const SeqLnArgs& a, int64_t row, int64_t* r) {
  if (a.off == nullptr) { *r = row; return true; }
  const int64_t b = row / a.L;
  const int i = (int)(row - b * a.L);
  const int32_t o = a.off[b];
  *r = (int64_t)o + i;
  return i < a.off[b + 1] - o;
}
__device__ __forceinline__ float4 seq_keep4(const SeqLnArgs& a, uint64_t seed, int64_t r, int q) {
  if (a.seed == nullptr) return make_float4(1.f, 1.f, 1.f, 1.f);
  uint32_t wd[4];
  philox4x32_10(seed, (uint64_t)r, a.site * (uint32_t)(a.d / 4) + (uint32_t)q, wd);
  return make_float4(wd[0] < a.thresh ? 0.f : a.scale, wd[1] < a.thresh ? 0.f : a.scale, wd[2] < a.thresh ? 0.f : a.scale,
                     wd[3] < a.thresh ? 0.f : a.scale);
}

// NV float4 per lane: d <= 256 * NV
template <int NV>
__global__ __launch_bounds__(kBlock) void seq_ln_fwd_kernel(SeqLnArgs a) {
  const int lane = threadIdx.x & 63;
  const int dq = a.d / 4;
  const uint64_t seed = a.seed ? *a.seed : 0;
  const int64_t n_waves = (int64_t)gridDim.x * (kBlock / 64);
  for (int64_t row = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6); row < a.rows; row += n_waves) {
    int64_t r;
    const bool valid = seq_row(a, row, &r);
    float4 z[NV];
    float sum = 0.f;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int q = v * 64 + lane;
      z[v] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (valid && q < dq) {
        const float4 x = reinterpret_cast<const float4*>(a.A)[row * dq + q];
        const float4 kp = seq_keep4(a, seed, r, q);
        z[v] = make_float4(x.x * kp.x, x.y * kp.y, x.z * kp.z, x.w * kp.w);
        if (a.R) {
          const float4 rr = reinterpret_cast<const float4*>(a.R)[row * dq + q];
          z[v].x += rr.x; z[v].y += rr.y; z[v].z += rr.z; z[v].w += rr.w;
        }
      }
      sum += (z[v].x + z[v].y) + (z[v].z + z[v].w);
    }
    const float mu = wave_allreduce_sum(sum) / a.d;
    float var = 0.f;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int q = v * 64 + lane;
      if (q < dq) {
        const float c0 = z[v].x - mu, c1 = z[v].y - mu, c2 = z[v].z - mu, c3 = z[v].w - mu;
        var = fmaf(c0, c0, var); var = fmaf(c1, c1, var); var = fmaf(c2, c2, var); var = fmaf(c3, c3, var);
      }
    }
    const float rs = valid ? 1.0f / sqrtf(wave_allreduce_sum(var) / a.d + kSeqLnEps) : 0.f;
    if (lane == 0) a.rstd[row] = rs;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int q = v * 64 + lane;
      if (q < dq) {
        float4 xh = make_float4(0.f, 0.f, 0.f, 0.f), y = xh;
        if (valid) {
          const float4 wv = reinterpret_cast<const float4*>(a.w)[q], bv = reinterpret_cast<const float4*>(a.b)[q];
          xh = make_float4((z[v].x - mu) * rs, (z[v].y - mu) * rs, (z[v].z - mu) * rs, (z[v].w - mu) * rs);
          y = make_float4(fmaf(xh.x, wv.x, bv.x), fmaf(xh.y, wv.y, bv.y), fmaf(xh.z, wv.z, bv.z), fmaf(xh.w, wv.w, bv.w));
        }
        reinterpret_cast<float4*>(a.xhat)[row * dq + q] = xh;
        reinterpret_cast<float4*>(a.Y)[row * dq + q] = y;
      }
    }
  }
}

template <int NV>
__global__ __launch_bounds__(kBlock) void seq_ln_bwd_kernel(SeqLnArgs a) {
  __shared__ float4 comb[(kBlock / 64) * 2 * 64 * NV];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int dq = a.d / 4;
  const uint64_t seed = a.seed ? *a.seed : 0;
  const int64_t n_waves = (int64_t)gridDim.x * (kBlock / 64);
  float4 gw[NV], gb[NV];
#pragma unroll
  for (int v = 0; v < NV; ++v) gw[v] = gb[v] = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int64_t row = (int64_t)blockIdx.x * (kBlock / 64) + wave; row < a.rows; row += n_waves) {
    int64_t r;
    const bool valid = seq_row(a, row, &r);
    const float rs = a.rstd[row];
    float4 dxh[NV], xh[NV];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int q = v * 64 + lane;
      dxh[v] = xh[v] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (valid && q < dq) {
        const float4 dy = reinterpret_cast<const float4*>(a.dY)[row * dq + q];
        const float4 wv = reinterpret_cast<const float4*>(a.w)[q];
        xh[v] = reinterpret_cast<const float4*>(a.xhat)[row * dq + q];
        dxh[v] = make_float4(dy.x * wv.x, dy.y * wv.y, dy.z * wv.z, dy.w * wv.w);
        gw[v].x = fmaf(dy.x, xh[v].x, gw[v].x); gw[v].y = fmaf(dy.y, xh[v].y, gw[v].y);
        gw[v].z = fmaf(dy.z, xh[v].z, gw[v].z); gw[v].w = fmaf(dy.w, xh[v].w, gw[v].w);
        gb[v].x += dy.x; gb[v].y += dy.y; gb[v].z += dy.z; gb[v].w += dy.w;
      }
      s1 += (dxh[v].x + dxh[v].y) + (dxh[v].z + dxh[v].w);
      s2 = fmaf(dxh[v].x, xh[v].x, s2); s2 = fmaf(dxh[v].y, xh[v].y, s2); s2 = fmaf(dxh[v].z, xh[v].z, s2); s2 = fmaf(dxh[v].w, xh[v].w, s2);
    }
    const float m1 = wave_allreduce_sum(s1) / a.d, m2 = wave_allreduce_sum(s2) / a.d;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int q = v * 64 + lane;
      if (q < dq) {
        const float4 dz = make_float4(rs * (dxh[v].x - m1 - xh[v].x * m2), rs * (dxh[v].y - m1 - xh[v].y * m2),
                                      rs * (dxh[v].z - m1 - xh[v].z * m2), rs * (dxh[v].w - m1 - xh[v].w * m2));
        if (a.dR) reinterpret_cast<float4*>(a.dR)[row * dq + q] = dz;
        if (a.dA) {
          const float4 kp = valid ? seq_keep4(a, seed, r, q) : make_float4(0.f, 0.f, 0.f, 0.f);
          reinterpret_cast<float4*>(a.dA)[row * dq + q] = make_float4(dz.x * kp.x, dz.y * kp.y, dz.z * kp.z, dz.w * kp.w);
        }
      }
    }
  }
  // this workgroup's share of d weight / d bias: the four waves in wave order
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    comb[((wave * 2 + 0) * NV + v) * 64 + lane] = gw[v];
    comb[((wave * 2 + 1) * NV + v) * 64 + lane] = gb[v];
  }
  __syncthreads();
  if (wave == 0) {
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int q = v * 64 + lane;
      if (q < dq) {
#pragma unroll
        for (int which = 0; which < 2; ++which) {
          float4 t = comb[((0 * 2 + which) * NV + v) * 64 + lane];
          for (int wv = 1; wv < kBlock / 64; ++wv) {
            const float4 u = comb[((wv * 2 + which) * NV + v) * 64 + lane];
            t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
          }
          reinterpret_cast<float4*>(a.part)[((size_t)blockIdx.x * 2 + which) * dq + q] = t;
        }
      }
    }
  }
}

// d weight [d], d bias [d] from the workgroups' shares, in workgroup order
__global__ __launch_bounds__(kBlock) void seq_ln_reduce_kernel(const float* __restrict__ part, int blocks, int d,
                                                               float* __restrict__ dw, float* __restrict__ db) {
  const int c = blockIdx.x * kBlock + threadIdx.x;
  if (c >= 2 * d) return;
  const int which = c / d, k = c - which * d;
  float s = 0.f;
  for (int q = 0; q < blocks; ++q) s += part[((size_t)q * 2 + which) * d + k];
  (which == 0 ? dw : db)[k] = s;
}

constexpr int kSeqLnBwdBlocks = 512;

}  // namespace rc

using namespace rc;

extern "C" int rc_seq_offsets(const int64_t* lengths, int64_t B, int L, int32_t* off, rc_stream_t stream) {
  RC_REQUIRE(off && (B == 0 || lengths), "rc_seq_offsets: null pointer");
  RC_REQUIRE(B >= 0 && L >= 1 && B * (int64_t)L < ((int64_t)1 << 31), "rc_seq_offsets: bad shape B=%lld L=%d", (long long)B, L);
  hipLaunchKernelGGL(seq_offsets_kernel, dim3(1), dim3(1024), 0, as_stream(stream), lengths, B, L, off);
  RC_LAUNCH_CHECK();
  return RC_OK;
}

extern "C" int rc_seq_embed_fwd(const float* item_emb, const float* pos_emb, const int64_t* hist, const int64_t* lengths, int64_t B,
                                int L, int d, float* X, rc_stream_t stream) {
  if (B == 0) return RC_OK;
  RC_REQUIRE(item_emb && pos_emb && hist && lengths && X, "rc_seq_embed_fwd: null pointer");
  RC_REQUIRE(B > 0 && L >= 1 && d >= 1, "rc_seq_embed_fwd: bad shape B=%lld L=%d d=%d", (long long)B, L, d);
  const bool vec = d % 4 == 0 && reinterpret_cast<uintptr_t>(item_emb) % 16 == 0 && reinterpret_cast<uintptr_t>(pos_emb) % 16 == 0 &&
                   reinterpret_cast<uintptr_t>(X) % 16 == 0;
  const int64_t total = B * L * (vec ? d / 4 : d);
  int64_t blocks = (total + kBlock - 1) / kBlock;
  if (blocks > 256 * 32) blocks = 256 * 32;
  if (vec)
    hipLaunchKernelGGL((seq_embed_kernel<4>), dim3((unsigned)blocks), dim3(kBlock), 0, as_stream(stream), item_emb, pos_emb, hist, lengths, B, L, d, X);
  else
    hipLaunchKernelGGL((seq_embed_kernel<1>), dim3((unsigned)blocks), dim3(kBlock), 0, as_stream(stream), item_emb, pos_emb, hist, lengths, B, L, d, X);
  RC_LAUNCH_CHECK();
  return RC_OK;
}

extern "C" int rc_seq_pick_last_fwd(const float* X, const int64_t* lengths, int64_t B, int L, int d, float* hv, rc_stream_t stream) {
  if (B == 0) return RC_OK;
  RC_REQUIRE(X && lengths && hv, "rc_seq_pick_last_fwd: null pointer");
  RC_REQUIRE(B > 0 && L >= 1 && d >= 1, "rc_seq_pick_last_fwd: bad shape B=%lld L=%d d=%d", (long long)B, L, d);
  int64_t blocks = (B * d + kBlock - 1) / kBlock;
  if (blocks > 256 * 16) blocks = 256 * 16;
  hipLaunchKernelGGL(seq_pick_last_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, as_stream(stream), X, lengths, B, L, d, hv);
  RC_LAUNCH_CHECK();
  return RC_OK;
}

extern "C" int rc_seq_pick_last_bwd(const float* dhv, const int64_t* lengths, int64_t B, int L, int d, float* dX, rc_stream_t stream) {
  if (B == 0) return RC_OK;
  RC_REQUIRE(dhv && lengths && dX, "rc_seq_pick_last_bwd: null pointer");
  RC_REQUIRE(B > 0 && L >= 1 && d >= 1, "rc_seq_pick_last_bwd: bad shape B=%lld L=%d d=%d", (long long)B, L, d);
  int64_t blocks = (B * L * d + kBlock - 1) / kBlock;
  if (blocks > 256 * 32) blocks = 256 * 32;
  hipLaunchKernelGGL(seq_pick_last_bwd_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, as_stream(stream), dhv, lengths, B, L, d, dX);
  RC_LAUNCH_CHECK();
  return RC_OK;
}

extern "C" int rc_seq_pos_grad(const float* dX, const int64_t* lengths, int64_t B, int L, int d, int n_pos, float* grad_pos, rc_stream_t stream) {
  RC_REQUIRE(grad_pos && n_pos >= 1 && L >= 1 && d >= 1 && B >= 0, "rc_seq_pos_grad: bad arguments (n_pos=%d, L=%d, d=%d)", n_pos, L, d);
  RC_REQUIRE(B == 0 || (dX && lengths), "rc_seq_pos_grad: null pointer");
  const bool vec = d % 4 == 0 && reinterpret_cast<uintptr_t>(dX) % 16 == 0;
  if (vec)
    hipLaunchKernelGGL((seq_pos_grad_kernel<4>), dim3((unsigned)n_pos), dim3(kBlock), 0, as_stream(stream), dX, lengths, B, L, d, n_pos, grad_pos);
  else
    hipLaunchKernelGGL((seq_pos_grad_kernel<1>), dim3((unsigned)n_pos), dim3(kBlock), 0, as_stream(stream), dX, lengths, B, L, d, n_pos, grad_pos);
  RC_LAUNCH_CHECK();
  return RC_OK;
}

extern "C" int rc_seq_attention_supported(int L, int dk) { return (L >= 1 && L <= kSeqMaxL && dk >= 1 && dk <= kSeqMaxDk) ? 1 : 0; }

static int seq_attn_args(const char* who, SeqAttnArgs* a, const float* Q, const float* K, const float* V, const int32_t* off,
                         const uint8_t* mask, int mask_batch, int causal, int64_t B, int L, int H, int dk) {
  RC_REQUIRE(Q && K && V, "%s: null pointer", who);
  RC_REQUIRE(B > 0 && H >= 1, "%s: bad shape B=%lld H=%d", who, (long long)B, H);
  if (!rc_seq_attention_supported(L, dk))
    return fail(RC_ERR_UNSUPPORTED, "%s: L=%d (<= %d), head width %d (<= %d) not covered", who, L, kSeqMaxL, dk, kSeqMaxDk);
  RC_REQUIRE(B * (int64_t)H * L < ((int64_t)1 << 31) * (kBlock / 64), "%s: too many rows", who);
  memset(a, 0, sizeof(*a));
  a->Q = Q; a->K = K; a->V = V; a->off = off; a->mask = mask; a->mask_batch = mask_batch ? 1 : 0; a->causal = causal ? 1 : 0;
  a->B = B; a->L = L; a->H = H; a->dk = dk;
  a->inv_sqrt = 1.0f / sqrtf((float)dk);
  return RC_OK;
}

#define RC_SEQ_LAUNCH(KERN, a, s)                                                                                    \
  do {                                                                                                               \
    const unsigned blocks_ = (unsigned)(((a).B * (a).H * (a).L + kBlock / 64 - 1) / (kBlock / 64));                   \
    const int t_ = ((a).L + 63) / 64;                                                                                \
    if (t_ <= 1) hipLaunchKernelGGL((KERN<1>), dim3(blocks_), dim3(kBlock), 0, s, a);                                \
    else if (t_ <= 2) hipLaunchKernelGGL((KERN<2>), dim3(blocks_), dim3(kBlock), 0, s, a);                           \
    else if (t_ <= 4) hipLaunchKernelGGL((KERN<4>), dim3(blocks_), dim3(kBlock), 0, s, a);                           \
    else if (t_ <= 8) hipLaunchKernelGGL((KERN<8>), dim3(blocks_), dim3(kBlock), 0, s, a);                           \
    else hipLaunchKernelGGL((KERN<16>), dim3(blocks_), dim3(kBlock), 0, s, a);                                       \
    RC_LAUNCH_CHECK();                                                                                               \
  } while (0)

extern "C" int rc_seq_attention_fwd(const float* Q, const float* K, const float* V, const int32_t* off, const uint8_t* mask,
                                    int mask_batch, int causal, int64_t B, int L, int H, int dk, float* ctx, float* lse,
                                    rc_stream_t stream) {
  if (B == 0) return RC_OK;
  SeqAttnArgs a;
  RC_TRY(seq_attn_args("rc_seq_attention_fwd", &a, Q, K, V, off, mask, mask_batch, causal, B, L, H, dk));
  RC_REQUIRE(ctx && lse, "rc_seq_attention_fwd: null pointer");
  a.ctx = ctx; a.lse = lse;
  hipStream_t s = as_stream(stream);
  RC_SEQ_LAUNCH(seq_attn_fwd_kernel, a, s);
  return RC_OK;
}

extern "C" int rc_seq_attention_bwd(const float* Q, const float* K, const float* V, const int32_t* off, const uint8_t* mask,
                                    int mask_batch, int causal, int64_t B, int L, int H, int dk, const float* lse,
                                    const float* dctx, float* Dv, float* dQ, float* dK, float* dV, rc_stream_t stream) {
  if (B == 0) return RC_OK;
  SeqAttnArgs a;
  RC_TRY(seq_attn_args("rc_seq_attention_bwd", &a, Q, K, V, off, mask, mask_batch, causal, B, L, H, dk));
  RC_REQUIRE(lse && dctx && Dv && dQ && dK && dV, "rc_seq_attention_bwd: null pointer");
  a.lse = const_cast<float*>(lse); a.dctx = dctx; a.Dv = Dv; a.dQ = dQ; a.dK = dK; a.dV = dV;
  hipStream_t s = as_stream(stream);
  RC_SEQ_LAUNCH(seq_attn_bwd_q_kernel, a, s);       // writes Dv, which the key-row pass reads
  RC_SEQ_LAUNCH(seq_attn_bwd_kv_kernel, a, s);
  return RC_OK;
}

static int seq_ln_args(const char* who, SeqLnArgs* a, const float* w, const float* b, const int32_t* off, int64_t rows, int L, int d,
                       float drop_p, const uint64_t* seed_dev, uint32_t site) {
  RC_REQUIRE(w && b, "%s: null pointer", who);
  RC_REQUIRE(rows > 0 && L >= 1 && (off == nullptr || rows % L == 0), "%s: bad shape rows=%lld L=%d", who, (long long)rows, L);
  if (d < 4 || d % 4 != 0 || d > kSeqMaxD)
    return fail(RC_ERR_UNSUPPORTED, "%s: row width %d (a multiple of 4 up to %d)", who, d, kSeqMaxD);
  RC_REQUIRE(drop_p >= 0.f && drop_p < 1.f && (drop_p == 0.f || seed_dev), "%s: dropout p=%g needs p in [0, 1) and a device seed", who, (double)drop_p);
  memset(a, 0, sizeof(*a));
  a->w = w; a->b = b; a->off = off; a->rows = rows; a->L = L; a->d = d;
  if (drop_p > 0.f) {
    a->seed = seed_dev;
    a->thresh = (uint32_t)((double)drop_p * 4294967296.0);
    a->scale = 1.0f / (1.0f - drop_p);
    a->site = site;
  }
  return RC_OK;
}

extern "C" int rc_seq_add_layernorm_fwd(const float* A, const float* R, const float* w, const float* b, const int32_t* off, int64_t rows,
                                        int L, int d, float drop_p, const uint64_t* seed_dev, uint32_t site, float* Y, float* xhat,
                                        float* rstd, rc_stream_t stream) {
  if (rows == 0) return RC_OK;
  SeqLnArgs a;
  RC_TRY(seq_ln_args("rc_seq_add_layernorm_fwd", &a, w, b, off, rows, L, d, drop_p, seed_dev, site));
  RC_REQUIRE(A && Y && xhat && rstd, "rc_seq_add_layernorm_fwd: null pointer");
  RC_REQUIRE(reinterpret_cast<uintptr_t>(A) % 16 == 0 && reinterpret_cast<uintptr_t>(R) % 16 == 0 && reinterpret_cast<uintptr_t>(Y) % 16 == 0 &&
             reinterpret_cast<uintptr_t>(xhat) % 16 == 0 && reinterpret_cast<uintptr_t>(w) % 16 == 0 && reinterpret_cast<uintptr_t>(b) % 16 == 0,
             "rc_seq_add_layernorm_fwd: buffers must be 16-byte aligned");
  a.A = A; a.R = R; a.Y = Y; a.xhat = xhat; a.rstd = rstd;
  int64_t blocks = (rows + kBlock / 64 - 1) / (kBlock / 64);
  if (blocks > 256 * 16) blocks = 256 * 16;
  hipStream_t s = as_stream(stream);
  if (d <= 256) hipLaunchKernelGGL((seq_ln_fwd_kernel<1>), dim3((unsigned)blocks), dim3(kBlock), 0, s, a);
  else if (d <= 512) hipLaunchKernelGGL((seq_ln_fwd_kernel<2>), dim3((unsigned)blocks), dim3(kBlock), 0, s, a);
  else hipLaunchKernelGGL((seq_ln_fwd_kernel<4>), dim3((unsigned)blocks), dim3(kBlock), 0, s, a);
  RC_LAUNCH_CHECK();
  return RC_OK;
}

extern "C" size_t rc_seq_add_layernorm_bwd_workspace_bytes(int d) {
  if (d < 1) d = 1;
  return align_up((size_t)kSeqLnBwdBlocks * 2 * (size_t)d * sizeof(float), 256);
}

extern "C" int rc_seq_add_layernorm_bwd(const float* dY, const float* xhat, const float* rstd, const float* w, const int32_t* off,
                                        int64_t rows, int L, int d, float drop_p, const uint64_t* seed_dev, uint32_t site, float* dA,
                                        float* dR, float* dw, float* db, void* ws, size_t ws_bytes, rc_stream_t stream) {
  RC_REQUIRE(dw && db, "rc_seq_add_layernorm_bwd: null pointer");
  hipStream_t s = as_stream(stream);
  if (rows == 0) {
    RC_HIP(hipMemsetAsync(dw, 0, (size_t)d * sizeof(float), s));
    RC_HIP(hipMemsetAsync(db, 0, (size_t)d * sizeof(float), s));
    return RC_OK;
  }
  SeqLnArgs a;
  RC_TRY(seq_ln_args("rc_seq_add_layernorm_bwd", &a, w, w, off, rows, L, d, drop_p, seed_dev, site));
  RC_REQUIRE(dY && xhat && rstd && ws, "rc_seq_add_layernorm_bwd: null pointer");
  RC_REQUIRE(ws_bytes >= rc_seq_add_layernorm_bwd_workspace_bytes(d), "rc_seq_add_layernorm_bwd: workspace %zu < %zu", ws_bytes,
             rc_seq_add_layernorm_bwd_workspace_bytes(d));
  RC_REQUIRE(reinterpret_cast<uintptr_t>(dY) % 16 == 0 && reinterpret_cast<uintptr_t>(xhat) % 16 == 0 && reinterpret_cast<uintptr_t>(dA) % 16 == 0 &&
             reinterpret_cast<uintptr_t>(dR) % 16 == 0 && reinterpret_cast<uintptr_t>(w) % 16 == 0, "rc_seq_add_layernorm_bwd: buffers must be 16-byte aligned");
  a.dY = dY; a.xhat = const_cast<float*>(xhat); a.rstd = const_cast<float*>(rstd); a.dA = dA; a.dR = dR;
  a.part = static_cast<float*>(ws);
  int64_t blocks = (rows + kBlock / 64 - 1) / (kBlock / 64);
  if (blocks > kSeqLnBwdBlocks) blocks = kSeqLnBwdBlocks;
  if (d <= 256) hipLaunchKernelGGL((seq_ln_bwd_kernel<1>), dim3((unsigned)blocks), dim3(kBlock), 0, s, a);
  else if (d <= 512) hipLaunchKernelGGL((seq_ln_bwd_kernel<2>), dim3((unsigned)blocks), dim3(kBlock), 0, s, a);
  else hipLaunchKernelGGL((seq_ln_bwd_kernel<4>), dim3((unsigned)blocks), dim3(kBlock), 0, s, a);
  RC_LAUNCH_CHECK();
  hipLaunchKernelGGL(seq_ln_reduce_kernel, dim3((unsigned)((2 * d + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, a.part, (int)blocks, d, dw, db);
  RC_LAUNCH_CHECK();
  return RC_OK;
}
