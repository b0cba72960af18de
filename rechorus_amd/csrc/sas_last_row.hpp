// sas_last_row.hpp -- the LAST encoder block when only one query row per sequence is consumed
// (models/sequential/SASRec.py:76: his_vector = his[arange(B), lengths - 1]; causal mask, utils/layers.py:92-118), WITHOUT
// materialising keys and values.  Included by sasrec_batch.hip (uses its sb_len / kBlock / wave reductions).
//
// With ONE query row q (= Wq x_last + bq) per sequence and head h (rows hc .. hc + dk of the projections):
//   score_j = q_h . k_hj = q_h . (Wk_h x_j + bk_h) = (Wk_h^T q_h) . x_j + q_h . bk_h  =  qt_h . x_j + cq_h
//   ctx_h   = sum_j p_j v_hj = Wv_h (sum_j p_j x_j) + bv_h sum_j p_j                  =  Wv_h xbar_h + bv_h
// so the block needs the layer input rows x_j ONCE (27 MB at config 3) instead of K, V projections of every row plus their
// re-read by the attention.  The backward has the same shape (g = d ctx, gt_h = Wv_h^T g_h, cg_h = g_h . bv_h):
//   dp_j = g_h . v_hj = gt_h . x_j + cg_h;   ds_j = p_j (dp_j - sum_j p_j dp_j) / sqrt(dk)
//   dX_j = sum_h ds_hj qt_h + p_hj gt_h                       (= dK_j Wk + dV_j Wv of the reference's graph)
//   dWk_h = sum_b q_h (x) ybar_h,  ybar_h = sum_j ds_hj x_j;   dWv_h = sum_b g_h (x) xbar_h
//   dbk_h = sum_b q_h sum_j ds_hj (zero up to rounding, as in the reference);  dbv = sum_b g
//   dq_h  = Wk_h ybar_h + bk_h sum_j ds_hj
// Everything per-row is a streaming pass over x; everything else lives on B rows.  Same mathematics as the all-rows path;
// the association of the products differs, so results agree to fp32 rounding (tests/test_gpu_sasrec.py compares both paths).
#pragma once

namespace rc {

__device__ __forceinline__ void wave_lds_sync() {   // LDS written by some lanes of this wave, read by others
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

constexpr int kLrMaxHeads = 4;
constexpr int kLrMaxL = kSasLongLP;   // history_max the one-row path covers (a 128-row tile, two keys per lane)
constexpr int kLrBlock = 512;   // the per-sequence head products: eight waves per workgroup, one sequence per wave and pass

struct SbLrRows {          // where the rows of the block's input live
  const float* X;          // [*, D]: row j of sequence b at (off ? off[b] : b * L) + j; nullptr: gathered from the tables
  const int32_t* off;      // compact row space, or nullptr: padded [B, L] rows
  const float* item_emb;   // gather (the block is the encoder's first): item_emb[hist[b, j]] + pos_emb[len - j], SASRec.py:62-66
  const float* pos_emb;
  const int64_t* hist;
  const int64_t* lengths;
  int B, L;
};
__device__ __forceinline__ size_t sb_lr_rowbase(const SbLrRows& r, int b) { return r.off ? (size_t)r.off[b] : (size_t)b * r.L; }

// ---- per-sequence head products against a weight matrix resident in LDS ------------------------------------------------------------
// outT[b, h, i] = sum_c in[b, hc + c] W[hc + c, i]   (qt from q and Wk; gt from g and Wv),  cs[b, h] = sum_c in[b, hc + c] bias[hc + c].
// (The products in the other direction -- ctx_h = Wv_h xbar_h + bv_h and dq_h = Wk_h ybar_h + bk_h sum ds -- are formed where they are
// consumed: in sb_block16_fwd_kernel's prologue and in sb_lr_tail_kernel.)
// MODE 0: `in` is given.  MODE 1 / 2: in = q = Wq x_last + bq, x_last read from X (1) or gathered from the tables (2);
// x_last and q are stored too.
struct SbLrHeadTArgs {
  SbLrRows rows;
  const float* in;         // MODE 0: [B, D]
  const float *Wq, *bq;    // MODE 1 / 2
  const float *W, *bias;   // [D, D] nn.Linear layout [out, in], [D]
  float *xl, *q;           // MODE 1 / 2 out: [B, D]
  float* outT;             // [B, H, D]
  float* cs;               // [B, 4]
  int32_t* off_seq;        // MODE 1 / 2: off_seq[B] = B (the row space of one row per sequence; sb_offsets_kernel writes it otherwise)
  int H;
};

template <int D, int MODE>
__global__ __launch_bounds__(kLrBlock) void sb_lr_headT_kernel(SbLrHeadTArgs a) {
  constexpr int SW = D + 4, LPR = D / 4;
  extern __shared__ float lds[];
  float* Ws = lds;                                   // [D][SW]
  float* Wqs = Ws + D * SW;                          // MODE != 0: [D][SW]
  float* wave_s = Wqs + (MODE ? D * SW : 0) + (threadIdx.x >> 6) * 2 * D;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int B = a.rows.B, L = a.rows.L, H = a.H, DK = D / H;
  if (MODE && blockIdx.x == 0 && threadIdx.x == 0 && a.off_seq) a.off_seq[B] = B;
  // the wave's input: x_last (MODE 1 / 2; 0 for an empty history) or in[b] (MODE 0)
  auto load_input = [&](int b) -> float {
    if (lane >= D) return 0.f;
    if (MODE == 0) return a.in[(size_t)b * D + lane];
    const int n = sb_len(a.rows.lengths, b, L);
    if (n == 0) return 0.f;
    if (MODE == 1) return a.rows.X[(sb_lr_rowbase(a.rows, b) + n - 1) * D + lane];
    // position id of the last row: len - (len - 1) = 1
    return a.rows.item_emb[a.rows.hist[(size_t)b * L + n - 1] * D + lane] + a.rows.pos_emb[D + lane];
  };
  int b = (int)blockIdx.x * (kLrBlock / 64) + wave;
  float pre = b < B ? load_input(b) : 0.f;   // the first sequence's input travels while the weights are staged
  for (int idx = threadIdx.x; idx < D * LPR; idx += kLrBlock) {
    const int o = idx / LPR, c = idx % LPR;
    *reinterpret_cast<float4*>(Ws + o * SW + 4 * c) = reinterpret_cast<const float4*>(a.W)[idx];
    if (MODE) *reinterpret_cast<float4*>(Wqs + o * SW + 4 * c) = reinterpret_cast<const float4*>(a.Wq)[idx];
  }
  __syncthreads();
  float* sx = wave_s;
  float* sq = wave_s + D;
  bool first = true;
  for (; b < B; b += (int)gridDim.x * (kLrBlock / 64)) {
    const float inv = first ? pre : load_input(b);
    first = false;
    float qv = inv;
    if (MODE) {
      if (lane < D) {
        sx[lane] = inv;
        a.xl[(size_t)b * D + lane] = inv;
      }
      wave_lds_sync();
      if (lane < D) {
        float acc = a.bq[lane];
#pragma unroll
        for (int c = 0; c < LPR; ++c) {
          const float4 w4 = *reinterpret_cast<const float4*>(Wqs + lane * SW + 4 * c);
          const float4 x4 = *reinterpret_cast<const float4*>(sx + 4 * c);
          acc = fmaf(w4.x, x4.x, acc); acc = fmaf(w4.y, x4.y, acc); acc = fmaf(w4.z, x4.z, acc); acc = fmaf(w4.w, x4.w, acc);
        }
        qv = acc;
        a.q[(size_t)b * D + lane] = acc;
      }
    }
    if (lane < D) sq[lane] = qv;
    wave_lds_sync();
    if (lane < D) {
      for (int h = 0; h < H; ++h) {
        float acc = 0.f;
        const float* wr = Ws + (h * DK) * SW + lane;
        const float* qr = sq + h * DK;
#pragma unroll 4
        for (int c = 0; c < DK; ++c) acc = fmaf(qr[c], wr[c * SW], acc);
        a.outT[((size_t)b * H + h) * D + lane] = acc;
      }
    }
    if (lane < kLrMaxHeads) {
      float t = 0.f;
      if (lane < H)
        for (int c = 0; c < DK; ++c) t = fmaf(sq[lane * DK + c], a.bias[lane * DK + c], t);
      a.cs[(size_t)b * kLrMaxHeads + lane] = t;
    }
    wave_lds_sync();
  }
}

// The tail of the block's backward on one row per sequence, in one pass: dq = Wk_h ybar_h + bk_h sum_j ds_hj (stored: the query
// projection's weight gradient needs it), d x_last = dq Wq + dZ1, added to the last row's dX (each row of G is touched by one wave).
// (Three launches before -- head product, 16-row-tile product, row add: 22 us of mostly launch latency at config 3.)
struct SbLrTailArgs {
  const float* ybar;        // [B, H, D]
  const float* sds;         // [B, 4]
  const float* gl;          // [B, D] dZ1
  const float *Wk, *bk, *Wq;
  const int64_t* lengths;
  const int32_t* g_off;     // G row of (b, j): (g_off ? g_off[b] : b * L) + j
  float* dq;                // [B, D] out
  float* G;
  int B, L, H;
};

template <int D>
__global__ __launch_bounds__(kLrBlock) void sb_lr_tail_kernel(SbLrTailArgs a) {
  constexpr int SW = D + 4, LPR = D / 4;
  extern __shared__ float lds[];
  float* Wks = lds;                // [D][SW]
  float* Wqs = Wks + D * SW;       // [D][SW]
  float* sin = Wqs + D * SW + (threadIdx.x >> 6) * (kLrMaxHeads * SW + D);   // this wave's [H][SW] head sums, then [D] dq
  float* sdq = sin + kLrMaxHeads * SW;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int H = a.H, DK = D / H;
  constexpr int NPRE = kLrMaxHeads * D / 64;
  int b = (int)blockIdx.x * (kLrBlock / 64) + wave;
  float pre[NPRE];   // the first sequence's input travels while the weights are staged
#pragma unroll
  for (int k = 0; k < NPRE; ++k) pre[k] = (b < a.B && lane + 64 * k < H * D) ? a.ybar[(size_t)b * H * D + lane + 64 * k] : 0.f;
  for (int idx = threadIdx.x; idx < D * LPR; idx += kLrBlock) {
    const int o = idx / LPR, c = idx % LPR;
    *reinterpret_cast<float4*>(Wks + o * SW + 4 * c) = reinterpret_cast<const float4*>(a.Wk)[idx];
    *reinterpret_cast<float4*>(Wqs + o * SW + 4 * c) = reinterpret_cast<const float4*>(a.Wq)[idx];
  }
  __syncthreads();
  bool first = true;
  for (; b < a.B; b += (int)gridDim.x * (kLrBlock / 64)) {
#pragma unroll
    for (int k = 0; k < NPRE; ++k) {
      const int e = lane + 64 * k;
      if (e < H * D) sin[(e / D) * SW + e % D] = first ? pre[k] : a.ybar[(size_t)b * H * D + e];
    }
    first = false;
    const int64_t len = a.lengths[b];
    const int n = (int)(len < 0 ? 0 : (len > a.L ? a.L : len));
    const float gv = lane < D ? a.gl[(size_t)b * D + lane] : 0.f;
    wave_lds_sync();
    if (lane < D) {
      const int h = lane / DK;
      float acc = 0.f;
#pragma unroll
      for (int c = 0; c < LPR; ++c) {
        const float4 w4 = *reinterpret_cast<const float4*>(Wks + lane * SW + 4 * c);
        const float4 v4 = *reinterpret_cast<const float4*>(sin + h * SW + 4 * c);
        acc = fmaf(w4.x, v4.x, acc); acc = fmaf(w4.y, v4.y, acc); acc = fmaf(w4.z, v4.z, acc); acc = fmaf(w4.w, v4.w, acc);
      }
      acc = fmaf(a.bk[lane], a.sds[(size_t)b * kLrMaxHeads + h], acc);
      a.dq[(size_t)b * D + lane] = acc;
      sdq[lane] = acc;
    }
    wave_lds_sync();
    if (lane < D && n > 0) {
      float acc = gv;
      const float* wc = Wqs + lane;
#pragma unroll 8
      for (int o = 0; o < D; ++o) acc = fmaf(sdq[o], wc[o * SW], acc);
      float* g = a.G + ((a.g_off ? (size_t)a.g_off[b] : (size_t)b * a.L) + n - 1) * D + lane;
      *g += acc;
    }
    wave_lds_sync();
  }
}

// ---- the streaming passes: one 4-wave workgroup per sequence ---------------------------------------------------------------------------
// The sequence's rows (history_max <= 128) are staged once in LDS (coalesced: D / 4 lanes per row).  Wave h owns head h: with lane = key it takes the
// scores and the softmax of its head, with lane = feature the weighted row sum.  (First version: one wave per sequence doing
// the four heads in turn -- eight waves per CU, each a chain of dependent global loads: 33 / 49 us forward / backward at
// config 3.  Four waves per sequence put 28-32 waves on a CU and cut each chain to a quarter.)
struct SbLrAttnArgs {
  SbLrRows rows;
  float* Xsave;             // fwd, gathering: the gathered rows are stored at the row base (the backward reads them back)
  const float* qt;          // [B, H, D]
  const float* cq;          // [B, 4]
  float* p;                 // [B, H, L] probabilities (fwd out, bwd in)
  float* xbar;              // fwd out [B, H, D]
  const float* gt;          // bwd [B, H, D]
  const float* cg;          // bwd [B, 4]
  float* G;                 // bwd out: dX rows at (g_off ? g_off[b] : b * L) + j; without g_off all L rows (zero past the length)
  const int32_t* g_off;
  float* ybar;              // bwd out [B, H, D]
  float* sds;               // bwd out [B, 4]
};

__device__ __forceinline__ float lr_wave_sum(float x) { return row_allreduce_sum<64>(x); }
__device__ __forceinline__ float lr_wave_max(float x) {
  x = fmaxf(x, dpp_mov<0xB1>(x));
  x = fmaxf(x, dpp_mov<0x4E>(x));
  x = fmaxf(x, dpp_mov<0x141>(x));
  x = fmaxf(x, dpp_mov<0x140>(x));
  x = fmaxf(x, __shfl_xor(x, 16, 64));
  x = fmaxf(x, __shfl_xor(x, 32, 64));
  return x;
}

// sum_i row[i] vec[i]: the lane's row of the tile against one head vector (broadcast reads)
template <int D>
__device__ __forceinline__ float sb_lr_row_dot(const float* row, const float* vec, float s) {
#pragma unroll
  for (int c = 0; c < D / 4; ++c) {
    const float4 x4 = *reinterpret_cast<const float4*>(row + 4 * c);
    const float4 q4 = *reinterpret_cast<const float4*>(vec + 4 * c);
    s = fmaf(x4.x, q4.x, s); s = fmaf(x4.y, q4.y, s); s = fmaf(x4.z, q4.z, s); s = fmaf(x4.w, q4.w, s);
  }
  return s;
}

// sum_{j < n} w[4 j + h] tile[j][4 f .. 4 f + 3]: lane = (feature quad f = lane % (D / 4), key group g = lane / (D / 4)); group g takes
// the keys j = g, g + NG, ...; the groups are added through xor-shuffles, every lane returns the total of its feature quad.
// (First version: lane = feature, one 4-byte LDS read per key and lane -- 25 dependent read latencies per sequence, a quarter of
// the forward kernel's time; one 16-byte read per 4 keys here.)
template <int D>
__device__ __forceinline__ float4 sb_lr_weighted_rows(const float* tile, const float* w, int h, int n, int lane) {
  constexpr int ST = D + 4, LPR = D / 4, NG = 64 / LPR;
  const int f = lane % LPR, g = lane / LPR;
  float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0;
  const float* col = tile + 4 * f;
  int j = g;
  for (; j + NG < n; j += 2 * NG) {
    const float4 x0 = *reinterpret_cast<const float4*>(col + j * ST), x1 = *reinterpret_cast<const float4*>(col + (j + NG) * ST);
    const float w0 = w[4 * j + h], w1 = w[4 * (j + NG) + h];
    a0.x = fmaf(w0, x0.x, a0.x); a0.y = fmaf(w0, x0.y, a0.y); a0.z = fmaf(w0, x0.z, a0.z); a0.w = fmaf(w0, x0.w, a0.w);
    a1.x = fmaf(w1, x1.x, a1.x); a1.y = fmaf(w1, x1.y, a1.y); a1.z = fmaf(w1, x1.z, a1.z); a1.w = fmaf(w1, x1.w, a1.w);
  }
  if (j < n) {
    const float4 x0 = *reinterpret_cast<const float4*>(col + j * ST);
    const float w0 = w[4 * j + h];
    a0.x = fmaf(w0, x0.x, a0.x); a0.y = fmaf(w0, x0.y, a0.y); a0.z = fmaf(w0, x0.z, a0.z); a0.w = fmaf(w0, x0.w, a0.w);
  }
  a0.x += a1.x; a0.y += a1.y; a0.z += a1.z; a0.w += a1.w;
#pragma unroll
  for (int off = LPR; off < 64; off <<= 1) {
    a0.x += __shfl_xor(a0.x, off, 64); a0.y += __shfl_xor(a0.y, off, 64);
    a0.z += __shfl_xor(a0.z, off, 64); a0.w += __shfl_xor(a0.w, off, 64);
  }
  return a0;
}

template <int D, int NH, bool GATHER>
__global__ __launch_bounds__(kBlock) void sb_lr_attn_fwd_kernel(SbLrAttnArgs a) {
  constexpr int ST = D + 4, LPR = D / 4, RPP = kBlock / LPR, NPASS = 64 / RPP;
  __shared__ __attribute__((aligned(16))) float tile[64 * ST];
  __shared__ __attribute__((aligned(16))) float sqt[NH * ST];
  __shared__ __attribute__((aligned(16))) float sp[64 * kLrMaxHeads];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int B = a.rows.B, L = a.rows.L;
  const float sqrt_dk = sqrtf((float)(D / NH));
  const int jr = t / LPR, cc = t % LPR;
  for (int b = blockIdx.x; b < B; b += gridDim.x) {
    const int n = sb_len(a.rows.lengths, b, L);   // workgroup-uniform
    if (n == 0) {   // empty history: defined zeros for the kernels that follow
      for (int e = t; e < NH * D; e += kBlock) a.xbar[(size_t)b * NH * D + e] = 0.f;
      for (int e = t; e < NH * L; e += kBlock) a.p[(size_t)b * NH * L + e] = 0.f;
      continue;
    }
    for (int e = t; e < NH * D; e += kBlock) sqt[(e / D) * ST + e % D] = a.qt[(size_t)b * NH * D + e];
    const size_t base = sb_lr_rowbase(a.rows, b);
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
      const int j = ps * RPP + jr;
      if (j < n) {
        float4 v;
        if (GATHER) {
          const float4 it = reinterpret_cast<const float4*>(a.rows.item_emb)[a.rows.hist[(size_t)b * L + j] * LPR + cc];
          const float4 ps4 = reinterpret_cast<const float4*>(a.rows.pos_emb)[(size_t)(n - j) * LPR + cc];
          v = make_float4(it.x + ps4.x, it.y + ps4.y, it.z + ps4.z, it.w + ps4.w);
          reinterpret_cast<float4*>(a.Xsave)[(base + j) * LPR + cc] = v;
        } else {
          v = reinterpret_cast<const float4*>(a.rows.X)[(base + j) * LPR + cc];
        }
        *reinterpret_cast<float4*>(tile + j * ST + 4 * cc) = v;
      }
    }
    __syncthreads();
    if (wave < NH) {   // head `wave`, lane = key
      const bool on = lane < n;
      const float s = sb_lr_row_dot<D>(tile + (on ? lane : 0) * ST, sqt + wave * ST, a.cq[(size_t)b * kLrMaxHeads + wave]);
      const float sc = sas_div_scale(s, sqrt_dk);
      const float m = lr_wave_max(on ? sc : -INFINITY);
      const float e = on ? expf(sc - m) : 0.f;
      const float pv = e * (1.0f / lr_wave_sum(e));
      if (lane < L) a.p[((size_t)b * NH + wave) * L + lane] = pv;
      sp[4 * lane + wave] = pv;
    }
    __syncthreads();
    if (wave < NH) {   // lane = feature
      const float4 acc = sb_lr_weighted_rows<D>(tile, sp, wave, n, lane);
      if (lane < D / 4) reinterpret_cast<float4*>(a.xbar + ((size_t)b * NH + wave) * D)[lane] = acc;
    }
    __syncthreads();   // the tile is rewritten by the next sequence
  }
}

template <int D, int NH>
__global__ __launch_bounds__(kBlock) void sb_lr_attn_bwd_kernel(SbLrAttnArgs a) {
  constexpr int ST = D + 4, LPR = D / 4, RPP = kBlock / LPR, NPASS = 64 / RPP;
  __shared__ __attribute__((aligned(16))) float tile[64 * ST];
  __shared__ __attribute__((aligned(16))) float sqt[NH * ST];
  __shared__ __attribute__((aligned(16))) float sgt[NH * ST];
  __shared__ __attribute__((aligned(16))) float sw[64 * kLrMaxHeads];   // ds by (key, head)
  __shared__ __attribute__((aligned(16))) float sp[64 * kLrMaxHeads];   // p by (key, head)
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int B = a.rows.B, L = a.rows.L;
  const float sqrt_dk = sqrtf((float)(D / NH));
  const int jr = t / LPR, cc = t % LPR;
  for (int b = blockIdx.x; b < B; b += gridDim.x) {
    const int n = sb_len(a.rows.lengths, b, L);
    const size_t gbase = a.g_off ? (size_t)a.g_off[b] : (size_t)b * L;
    const int rows_out = a.g_off ? n : L;
    if (n == 0) {
      for (int e = t; e < NH * D; e += kBlock) a.ybar[(size_t)b * NH * D + e] = 0.f;
      if (t < kLrMaxHeads) a.sds[(size_t)b * kLrMaxHeads + t] = 0.f;
#pragma unroll
      for (int ps = 0; ps < NPASS; ++ps)
        if (ps * RPP + jr < rows_out) reinterpret_cast<float4*>(a.G)[(gbase + ps * RPP + jr) * LPR + cc] = make_float4(0.f, 0.f, 0.f, 0.f);
      continue;
    }
    for (int e = t; e < NH * D; e += kBlock) {
      sqt[(e / D) * ST + e % D] = a.qt[(size_t)b * NH * D + e];
      sgt[(e / D) * ST + e % D] = a.gt[(size_t)b * NH * D + e];
    }
    const size_t base = sb_lr_rowbase(a.rows, b);
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
      const int j = ps * RPP + jr;
      if (j < n) *reinterpret_cast<float4*>(tile + j * ST + 4 * cc) = reinterpret_cast<const float4*>(a.rows.X)[(base + j) * LPR + cc];
    }
    __syncthreads();
    if (wave < NH) {   // head `wave`, lane = key
      const bool on = lane < n;
      const float pv = on ? a.p[((size_t)b * NH + wave) * L + lane] : 0.f;
      const float dp = sb_lr_row_dot<D>(tile + (on ? lane : 0) * ST, sgt + wave * ST, a.cg[(size_t)b * kLrMaxHeads + wave]);
      const float dot = lr_wave_sum(pv * dp);
      const float ds = sas_div_scale(pv * (dp - dot), sqrt_dk);
      const float tot = lr_wave_sum(ds);
      if (lane == 0) a.sds[(size_t)b * kLrMaxHeads + wave] = tot;
      sw[4 * lane + wave] = ds;
      sp[4 * lane + wave] = pv;
    }
    __syncthreads();
    if (wave < NH) {   // lane = feature
      const float4 acc = sb_lr_weighted_rows<D>(tile, sw, wave, n, lane);
      if (lane < D / 4) reinterpret_cast<float4*>(a.ybar + ((size_t)b * NH + wave) * D)[lane] = acc;
    }
    __syncthreads();   // every wave is done with the x rows: the tile now takes the dX rows
    {   // row `lane`, columns [wave * D / 4, (wave + 1) * D / 4); rows past the length have ds = p = 0
      float dsv[NH], pvv[NH];
#pragma unroll
      for (int h = 0; h < NH; ++h) {
        dsv[h] = sw[4 * lane + h];
        pvv[h] = sp[4 * lane + h];
      }
      float* row = tile + lane * ST + wave * (D / 4);
#pragma unroll
      for (int c = 0; c < D / 16; ++c) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int h = 0; h < NH; ++h) {
          const float4 q4 = *reinterpret_cast<const float4*>(sqt + h * ST + wave * (D / 4) + 4 * c);
          const float4 g4 = *reinterpret_cast<const float4*>(sgt + h * ST + wave * (D / 4) + 4 * c);
          v.x = fmaf(dsv[h], q4.x, v.x); v.y = fmaf(dsv[h], q4.y, v.y); v.z = fmaf(dsv[h], q4.z, v.z); v.w = fmaf(dsv[h], q4.w, v.w);
          v.x = fmaf(pvv[h], g4.x, v.x); v.y = fmaf(pvv[h], g4.y, v.y); v.z = fmaf(pvv[h], g4.z, v.z); v.w = fmaf(pvv[h], g4.w, v.w);
        }
        *reinterpret_cast<float4*>(row + 4 * c) = v;
      }
    }
    __syncthreads();
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
      const int j = ps * RPP + jr;
      if (j < rows_out) reinterpret_cast<float4*>(a.G)[(gbase + j) * LPR + cc] = *reinterpret_cast<const float4*>(tile + j * ST + 4 * cc);
    }
    __syncthreads();
  }
}

// The same two passes for 64 < history_max <= 128: KPL = 2 keys per lane, a tile of 64 * KPL rows.  (Kept apart from the kernels above:
// written with KPL = 1 the backward compiles to 146 instead of 108 registers -- three instead of four workgroups per CU on the path
// every benchmark configuration takes.)
template <int D, int NH, bool GATHER, int KPL>
__global__ __launch_bounds__(kBlock) void sb_lr_attn_fwd_long_kernel(SbLrAttnArgs a) {
  constexpr int ST = D + 4, LPR = D / 4, RPP = kBlock / LPR, TR = 64 * KPL, NPASS = TR / RPP;
  __shared__ __attribute__((aligned(16))) float tile[TR * ST];
  __shared__ __attribute__((aligned(16))) float sqt[NH * ST];
  __shared__ __attribute__((aligned(16))) float sp[TR * kLrMaxHeads];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int B = a.rows.B, L = a.rows.L;
  const float sqrt_dk = sqrtf((float)(D / NH));
  const int jr = t / LPR, cc = t % LPR;
  for (int b = blockIdx.x; b < B; b += gridDim.x) {
    const int n = sb_len(a.rows.lengths, b, L);   // workgroup-uniform
    if (n == 0) {   // empty history: defined zeros for the kernels that follow
      for (int e = t; e < NH * D; e += kBlock) a.xbar[(size_t)b * NH * D + e] = 0.f;
      for (int e = t; e < NH * L; e += kBlock) a.p[(size_t)b * NH * L + e] = 0.f;
      continue;
    }
    for (int e = t; e < NH * D; e += kBlock) sqt[(e / D) * ST + e % D] = a.qt[(size_t)b * NH * D + e];
    const size_t base = sb_lr_rowbase(a.rows, b);
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
      const int j = ps * RPP + jr;
      if (j < n) {
        float4 v;
        if (GATHER) {
          const float4 it = reinterpret_cast<const float4*>(a.rows.item_emb)[a.rows.hist[(size_t)b * L + j] * LPR + cc];
          const float4 ps4 = reinterpret_cast<const float4*>(a.rows.pos_emb)[(size_t)(n - j) * LPR + cc];
          v = make_float4(it.x + ps4.x, it.y + ps4.y, it.z + ps4.z, it.w + ps4.w);
          reinterpret_cast<float4*>(a.Xsave)[(base + j) * LPR + cc] = v;
        } else {
          v = reinterpret_cast<const float4*>(a.rows.X)[(base + j) * LPR + cc];
        }
        *reinterpret_cast<float4*>(tile + j * ST + 4 * cc) = v;
      }
    }
    __syncthreads();
    if (wave < NH) {   // head `wave`, lane = key (+ 64 k)
      const float cq = a.cq[(size_t)b * kLrMaxHeads + wave];
      float sc[KPL], mx = -INFINITY;
#pragma unroll
      for (int k = 0; k < KPL; ++k) {
        const int j = lane + 64 * k;
        const bool on = j < n;
        sc[k] = sas_div_scale(sb_lr_row_dot<D>(tile + (on ? j : 0) * ST, sqt + wave * ST, cq), sqrt_dk);
        if (on) mx = fmaxf(mx, sc[k]);
      }
      const float m = lr_wave_max(mx);
      float e[KPL], es = 0.f;
#pragma unroll
      for (int k = 0; k < KPL; ++k) {
        e[k] = lane + 64 * k < n ? expf(sc[k] - m) : 0.f;
        es += e[k];
      }
      const float rz = 1.0f / lr_wave_sum(es);
#pragma unroll
      for (int k = 0; k < KPL; ++k) {
        const int j = lane + 64 * k;
        const float pv = e[k] * rz;
        if (j < L) a.p[((size_t)b * NH + wave) * L + j] = pv;
        sp[4 * j + wave] = pv;
      }
    }
    __syncthreads();
    if (wave < NH) {   // lane = (feature quad, key group)
      const float4 acc = sb_lr_weighted_rows<D>(tile, sp, wave, n, lane);
      if (lane < D / 4) reinterpret_cast<float4*>(a.xbar + ((size_t)b * NH + wave) * D)[lane] = acc;
    }
    __syncthreads();   // the tile is rewritten by the next sequence
  }
}

template <int D, int NH, int KPL>
__global__ __launch_bounds__(kBlock) void sb_lr_attn_bwd_long_kernel(SbLrAttnArgs a) {
  constexpr int ST = D + 4, LPR = D / 4, RPP = kBlock / LPR, TR = 64 * KPL, NPASS = TR / RPP;
  __shared__ __attribute__((aligned(16))) float tile[TR * ST];
  __shared__ __attribute__((aligned(16))) float sqt[NH * ST];
  __shared__ __attribute__((aligned(16))) float sgt[NH * ST];
  __shared__ __attribute__((aligned(16))) float sw[TR * kLrMaxHeads];   // ds by (key, head)
  __shared__ __attribute__((aligned(16))) float sp[TR * kLrMaxHeads];   // p by (key, head)
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int B = a.rows.B, L = a.rows.L;
  const float sqrt_dk = sqrtf((float)(D / NH));
  const int jr = t / LPR, cc = t % LPR;
  for (int b = blockIdx.x; b < B; b += gridDim.x) {
    const int n = sb_len(a.rows.lengths, b, L);
    const size_t gbase = a.g_off ? (size_t)a.g_off[b] : (size_t)b * L;
    const int rows_out = a.g_off ? n : L;
    if (n == 0) {
      for (int e = t; e < NH * D; e += kBlock) a.ybar[(size_t)b * NH * D + e] = 0.f;
      if (t < kLrMaxHeads) a.sds[(size_t)b * kLrMaxHeads + t] = 0.f;
#pragma unroll
      for (int ps = 0; ps < NPASS; ++ps)
        if (ps * RPP + jr < rows_out) reinterpret_cast<float4*>(a.G)[(gbase + ps * RPP + jr) * LPR + cc] = make_float4(0.f, 0.f, 0.f, 0.f);
      continue;
    }
    for (int e = t; e < NH * D; e += kBlock) {
      sqt[(e / D) * ST + e % D] = a.qt[(size_t)b * NH * D + e];
      sgt[(e / D) * ST + e % D] = a.gt[(size_t)b * NH * D + e];
    }
    const size_t base = sb_lr_rowbase(a.rows, b);
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
      const int j = ps * RPP + jr;
      if (j < n) *reinterpret_cast<float4*>(tile + j * ST + 4 * cc) = reinterpret_cast<const float4*>(a.rows.X)[(base + j) * LPR + cc];
    }
    __syncthreads();
    if (wave < NH) {   // head `wave`, lane = key (+ 64 k)
      const float cg = a.cg[(size_t)b * kLrMaxHeads + wave];
      float pv[KPL], dp[KPL], pd = 0.f;
#pragma unroll
      for (int k = 0; k < KPL; ++k) {
        const int j = lane + 64 * k;
        const bool on = j < n;
        pv[k] = on ? a.p[((size_t)b * NH + wave) * L + j] : 0.f;
        dp[k] = sb_lr_row_dot<D>(tile + (on ? j : 0) * ST, sgt + wave * ST, cg);
        pd += pv[k] * dp[k];
      }
      const float dot = lr_wave_sum(pd);
      float ds[KPL], dsum = 0.f;
#pragma unroll
      for (int k = 0; k < KPL; ++k) {
        ds[k] = sas_div_scale(pv[k] * (dp[k] - dot), sqrt_dk);
        dsum += ds[k];
      }
      const float tot = lr_wave_sum(dsum);
      if (lane == 0) a.sds[(size_t)b * kLrMaxHeads + wave] = tot;
#pragma unroll
      for (int k = 0; k < KPL; ++k) {
        sw[4 * (lane + 64 * k) + wave] = ds[k];
        sp[4 * (lane + 64 * k) + wave] = pv[k];
      }
    }
    __syncthreads();
    if (wave < NH) {   // lane = (feature quad, key group)
      const float4 acc = sb_lr_weighted_rows<D>(tile, sw, wave, n, lane);
      if (lane < D / 4) reinterpret_cast<float4*>(a.ybar + ((size_t)b * NH + wave) * D)[lane] = acc;
    }
    __syncthreads();   // every wave is done with the x rows: the tile now takes the dX rows
#pragma unroll
    for (int k = 0; k < KPL; ++k) {   // row lane + 64 k, columns [wave * D / 4, (wave + 1) * D / 4); rows past the length have ds = p = 0
      const int j = lane + 64 * k;
      float dsv[NH], pvv[NH];
#pragma unroll
      for (int h = 0; h < NH; ++h) {
        dsv[h] = sw[4 * j + h];
        pvv[h] = sp[4 * j + h];
      }
      float* row = tile + j * ST + wave * (D / 4);
#pragma unroll
      for (int c = 0; c < D / 16; ++c) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int h = 0; h < NH; ++h) {
          const float4 q4 = *reinterpret_cast<const float4*>(sqt + h * ST + wave * (D / 4) + 4 * c);
          const float4 g4 = *reinterpret_cast<const float4*>(sgt + h * ST + wave * (D / 4) + 4 * c);
          v.x = fmaf(dsv[h], q4.x, v.x); v.y = fmaf(dsv[h], q4.y, v.y); v.z = fmaf(dsv[h], q4.z, v.z); v.w = fmaf(dsv[h], q4.w, v.w);
          v.x = fmaf(pvv[h], g4.x, v.x); v.y = fmaf(pvv[h], g4.y, v.y); v.z = fmaf(pvv[h], g4.z, v.z); v.w = fmaf(pvv[h], g4.w, v.w);
        }
        *reinterpret_cast<float4*>(row + 4 * c) = v;
      }
    }
    __syncthreads();
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
      const int j = ps * RPP + jr;
      if (j < rows_out) reinterpret_cast<float4*>(a.G)[(gbase + j) * LPR + cc] = *reinterpret_cast<const float4*>(tile + j * ST + 4 * cc);
    }
    __syncthreads();
  }
}

// ---- parameter gradients of the key / value projections from the per-sequence sums ---------------------------------------------------
// dWk[o, i] = sum_b q[b, o] ybar[b, h(o), i], dbk[o] = sum_b q[b, o] sds[b, h(o)], dWv[o, i] = sum_b g[b, o] xbar[b, h(o), i],
// dbv[o] = sum_b g[b, o]; workgroup w sums the sequences [w * chunk, (w + 1) * chunk) in ascending order into its partial slice.
struct SbLrWgradArgs {
  const float *q, *ybar, *sds, *g, *xbar;
  float *gWk, *gbk, *gWv, *gbv;    // this workgroup's slice at + blockIdx.x * part_stride
  size_t part_stride;
  int B, H, chunk;
};

constexpr int kLrWgradSeqs = 16;   // sequences per pass of a workgroup

template <int D>
__global__ __launch_bounds__(kBlock) void sb_lr_wgrad_kernel(SbLrWgradArgs a) {
  constexpr int OPT = D * D / kBlock;   // outputs per thread (all in one head: OPT divides dk)
  constexpr int CH = kLrWgradSeqs;
  __shared__ __attribute__((aligned(16))) float sq[CH * D];
  __shared__ __attribute__((aligned(16))) float sg[CH * D];
  __shared__ float ssd[CH * kLrMaxHeads];
  const int i = threadIdx.x % D, o0 = (threadIdx.x / D) * OPT;
  const int H = a.H, DK = D / H, h = o0 / DK;
  float ak[OPT], av[OPT], bk[OPT], bv[OPT];
#pragma unroll
  for (int t = 0; t < OPT; ++t) ak[t] = av[t] = bk[t] = bv[t] = 0.f;
  const int b0 = (int)blockIdx.x * a.chunk, b1 = min(a.B, b0 + a.chunk);
  for (int c0 = b0; c0 < b1; c0 += CH) {
    const int m = min(CH, b1 - c0);
    // every operand of the pass is requested before the first is used: q, g, sum ds through LDS (each thread needs the
    // 2 * OPT values of its outputs for every sequence), the per-head sums straight into registers
    float yv[CH], xv[CH];
#pragma unroll
    for (int k = 0; k < CH; ++k) {
      yv[k] = k < m ? a.ybar[((size_t)(c0 + k) * H + h) * D + i] : 0.f;
      xv[k] = k < m ? a.xbar[((size_t)(c0 + k) * H + h) * D + i] : 0.f;
    }
    __syncthreads();   // the previous pass's readers are done
    for (int e = threadIdx.x; e < CH * D / 4; e += kBlock) {
      const bool on = e / (D / 4) < m;
      float4 vq = make_float4(0.f, 0.f, 0.f, 0.f), vg = vq;     // (no named zero in a ternary: it was kept in a stack slot)
      if (on) {
        vq = reinterpret_cast<const float4*>(a.q + (size_t)c0 * D)[e];
        vg = reinterpret_cast<const float4*>(a.g + (size_t)c0 * D)[e];
      }
      reinterpret_cast<float4*>(sq)[e] = vq;
      reinterpret_cast<float4*>(sg)[e] = vg;
    }
    if ((int)threadIdx.x < CH * kLrMaxHeads)
      ssd[threadIdx.x] = (int)threadIdx.x / kLrMaxHeads < m ? a.sds[(size_t)c0 * kLrMaxHeads + threadIdx.x] : 0.f;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < CH; ++k) {   // sequences in ascending order
      const float sd = ssd[k * kLrMaxHeads + h];
#pragma unroll
      for (int t4 = 0; t4 < OPT / 4; ++t4) {
        const float4 q4 = *reinterpret_cast<const float4*>(sq + k * D + o0 + 4 * t4);
        const float4 g4 = *reinterpret_cast<const float4*>(sg + k * D + o0 + 4 * t4);
        const float qq[4] = {q4.x, q4.y, q4.z, q4.w}, gg[4] = {g4.x, g4.y, g4.z, g4.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int t = 4 * t4 + e;
          ak[t] = fmaf(qq[e], yv[k], ak[t]);
          av[t] = fmaf(gg[e], xv[k], av[t]);
          bk[t] = fmaf(qq[e], sd, bk[t]);
          bv[t] += gg[e];
        }
      }
    }
  }
  const size_t sl = (size_t)blockIdx.x * a.part_stride;
#pragma unroll
  for (int t = 0; t < OPT; ++t) {
    a.gWk[sl + (size_t)(o0 + t) * D + i] = ak[t];
    a.gWv[sl + (size_t)(o0 + t) * D + i] = av[t];
    if (i == 0) {
      a.gbk[sl + o0 + t] = bk[t];
      a.gbv[sl + o0 + t] = bv[t];
    }
  }
}

}  // namespace rc
