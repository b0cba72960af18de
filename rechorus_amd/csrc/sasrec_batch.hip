// sasrec_batch.hip -- the SASRec encoder (models/sequential/SASRec.py:51-86, utils/layers.py:9-63,92-118)
// decomposed at BATCH level: the valid history rows of the whole batch form one compact row space
// [R = sum len, d]; everything that is row-wise (the five d x d projections, both LayerNorms, the FFN) is a
// streaming kernel over that space with the weights resident in LDS, and only the attention itself runs per
// sequence.  sasrec.hip keeps a sequence's whole layer in LDS and pays ~30 barrier-separated phases per
// sequence on one workgroup per CU (matrix cores busy 7 % of the time); here every phase is one launch over
// ~100 K rows, bound by HBM streaming of [R, d] activations (26 MB each at B = 4096, mean length 25).
//
// Same arithmetic as sasrec.hip: fp32 v_mfma_f32_32x32x2_f32 for every contraction, LayerNorm with biased
// variance and eps 1e-5, causal mask only, position id = length - index, no attention output projection,
// dropout 0.  Activations needed by the backward pass (layer input, q, k, v, xhat1, y1, relu hidden, xhat2,
// the two rstd vectors) are SAVED by the forward pass -- they are its natural intermediates -- instead of being
// recomputed.  Dense-parameter gradients: per-workgroup partials in private slices, summed in fixed order
// (sas_reduce_partials_kernel); no float atomics anywhere.
#include "common.hpp"
#include "philox.hpp"
#include "sas_mma.hpp"

namespace rc {

constexpr int kSbTile = 64;  // rows per tile of the row-space kernels

// ---- row space ---------------------------------------------------------------------------------------

// off[b] = sum_{b' < b} min(len[b'], L) (exclusive), off[B] = R.  One workgroup; B <= 2^24.
__global__ __launch_bounds__(kBlock) void sb_offsets_kernel(const int64_t* __restrict__ lengths, int B, int L,
                                                            int32_t* __restrict__ off, int32_t* __restrict__ off_seq) {
  // off_seq[B] = B: the "row space" of one row per sequence (the last layer's last-row path runs the same kernels on it)
  if (threadIdx.x == 0 && off_seq) off_seq[B] = B;
  __shared__ int s_sum[kBlock];
  const int per = (B + kBlock - 1) / kBlock;
  const int lo = min((int)threadIdx.x * per, B), hi = min(lo + per, B);
  int s = 0;
  for (int b = lo; b < hi; ++b) {
    const int64_t n = lengths[b];
    s += (int)(n < 0 ? 0 : (n > L ? L : n));
  }
  s_sum[threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    int run = 0;
    for (int t = 0; t < kBlock; ++t) {
      const int c = s_sum[t];
      s_sum[t] = run;
      run += c;
    }
    off[B] = run;
  }
  __syncthreads();
  int run = s_sum[threadIdx.x];
  for (int b = lo; b < hi; ++b) {
    off[b] = run;
    const int64_t n = lengths[b];
    run += (int)(n < 0 ? 0 : (n > L ? L : n));
  }
}

__device__ __forceinline__ int sb_len(const int64_t* lengths, int b, int L) {
  const int64_t n = lengths[b];
  return (int)(n < 0 ? 0 : (n > L ? L : n));
}

// X[off[b] + i] = item_emb[hist[b, i]] + pos_emb[len - i]  (SASRec.py:62-66)
template <int D>
__global__ __launch_bounds__(kBlock) void sb_embed_kernel(const float* __restrict__ item_emb,
                                                          const float* __restrict__ pos_emb,
                                                          const int64_t* __restrict__ hist,
                                                          const int64_t* __restrict__ lengths, int B, int L,
                                                          const int32_t* __restrict__ off, float* __restrict__ X) {
  constexpr int LPR = D / 4;
  const int l = threadIdx.x % LPR;
  const int64_t total = (int64_t)B * L;
  for (int64_t e = (int64_t)blockIdx.x * (kBlock / LPR) + threadIdx.x / LPR; e < total;
       e += (int64_t)gridDim.x * (kBlock / LPR)) {
    const int b = (int)(e / L), i = (int)(e - (int64_t)b * L);
    const int n = sb_len(lengths, b, L);
    if (i >= n) continue;
    const int64_t r = off[b] + i;
    const float4 a = reinterpret_cast<const float4*>(item_emb)[hist[e] * LPR + l];
    const float4 p = reinterpret_cast<const float4*>(pos_emb)[(int64_t)(n - i) * LPR + l];
    reinterpret_cast<float4*>(X)[r * LPR + l] = make_float4(a.x + p.x, a.y + p.y, a.z + p.z, a.w + p.w);
  }
}

// ---- Y_w = epilogue(X . W_w^T)  or  X . W_w  over the row space, W_w resident in LDS ---------------------

struct SbLinArgs {
  const float* X;       // [R, D]
  const float* W[3];    // nn.Linear weights [out, in]
  const float* bias[3]; // may be null
  float* Y[3];          // [R, D]
  const float* res;     // optional: Y += res            (same row space)
  const float* mask;    // optional: Y = mask > 0 ? Y : 0 (ReLU backward)
  int relu;
  const int32_t* off;
  int B;
};

template <int D, int NW, bool TRANS>
__global__ __launch_bounds__(kBlock) void sb_linear_kernel(SbLinArgs a) {
  constexpr int SD = D + 1;
  extern __shared__ float lds[];
  float* Ws = lds;                    // [NW][D][SD]
  float* Xs = lds + NW * D * SD;      // [kSbTile][SD]
  // biases live in LDS too: a global load behind the epilogue's global stores waits for those stores (gfx950 counts
  // loads and stores in one in-order vmcnt) -- 16 serialised load / store round trips per lane and block otherwise
  float* Bs = Xs + kSbTile * SD;      // [NW][D]
  for (int w = 0; w < NW; ++w) {
    for (int idx = threadIdx.x; idx < D * D; idx += kBlock) Ws[w * D * SD + (idx / D) * SD + idx % D] = a.W[w][idx];
    for (int idx = threadIdx.x; idx < D; idx += kBlock) Bs[w * D + idx] = a.bias[w] ? a.bias[w][idx] : 0.f;
  }
  const int R = a.off[a.B];
  const int tiles = (R + kSbTile - 1) / kSbTile;
  // the next tile's rows are requested while the current tile is multiplied: each thread holds its
  // kSbTile * D / 4 / 256 float4 of the tile in registers between the two barriers
  constexpr int PF = kSbTile * (D / 4) / kBlock;
  float4 pf[PF];
  auto fetch = [&](int tile) {
    const int r0 = tile * kSbTile;
    const int m = min(kSbTile, R - r0);
#pragma unroll
    for (int q = 0; q < PF; ++q) {
      const int idx = threadIdx.x + q * kBlock, i = idx / (D / 4), c = idx % (D / 4);
      pf[q] = i < m ? reinterpret_cast<const float4*>(a.X)[(size_t)(r0 + i) * (D / 4) + c] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  if ((int)blockIdx.x < tiles) fetch(blockIdx.x);
  for (int tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    const int r0 = tile * kSbTile;
    const int m = min(kSbTile, R - r0);
    __syncthreads();  // previous tile's readers are done (and the weights are in place)
#pragma unroll
    for (int q = 0; q < PF; ++q) {
      const int idx = threadIdx.x + q * kBlock, i = idx / (D / 4), c = idx % (D / 4);
      float* d = Xs + i * SD + 4 * c;
      d[0] = pf[q].x; d[1] = pf[q].y; d[2] = pf[q].z; d[3] = pf[q].w;
    }
    __syncthreads();
    if (tile + (int)gridDim.x < tiles) fetch(tile + gridDim.x);
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      const float* Wl = Ws + w * D * SD;
      const MatB mb = TRANS ? MatB{Wl, SD, 1} : MatB{Wl, 1, SD};  // dx = dy . W   |   y = x . W^T
      const float* bias = Bs + w * D;
      float* Y = a.Y[w];
      sas_mm(MatA{Xs, SD, 1}, mb, m, D, D, false, [&](int i, int j, float v) {
        const size_t e = (size_t)(r0 + i) * D + j;
        v += bias[j];
        if (a.relu) v = fmaxf(v, 0.f);
        if (a.mask) v = a.mask[e] > 0.f ? v : 0.f;
        if (a.res) v += a.res[e];
        Y[e] = v;
      });
    }
  }
}

// Y = res + X_0 . W_0 + X_1 . W_1 + X_2 . W_2  (dX of the three projections: dZ1 + dQ Wq + dK Wk + dV Wv) in one
// pass: the accumulators of a wave's 32x32 output block stay in registers across the three inputs, whose tiles
// are staged one after the other (next one prefetched) -- instead of three launches that each re-read and
// re-write the [R, D] running sum.
struct SbSum3Args {
  const float* X[3];
  const float* W[3];  // [out, in]: the product uses W as is (dx = dy . W)
  const float* res;
  float* Y;
  const int32_t* off;
  int B;
};

template <int D>
__global__ __launch_bounds__(kBlock) void sb_linear3_sum_kernel(SbSum3Args a) {
  constexpr int SD = D + 1, NRB = kSbTile / 32, NCB = D / 32, NQ = NRB * NCB;  // NQ <= 4: one block per wave
  extern __shared__ float lds[];
  float* Ws = lds;                // [3][D][SD]
  float* Xs = lds + 3 * D * SD;   // [kSbTile][SD]
  for (int w = 0; w < 3; ++w)
    for (int idx = threadIdx.x; idx < D * D; idx += kBlock) Ws[w * D * SD + (idx / D) * SD + idx % D] = a.W[w][idx];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, kh = lane >> 5;
  const int R = a.off[a.B];
  const int tiles = (R + kSbTile - 1) / kSbTile;
  const int my_tiles = (int)blockIdx.x < tiles ? (tiles - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
  constexpr int PF = kSbTile * (D / 4) / kBlock;
  float4 pf[PF];
  auto fetch = [&](int step) {  // step = (tile ordinal, input w)
    const int tile = blockIdx.x + (step / 3) * gridDim.x, w = step % 3;
    const int r0 = tile * kSbTile;
    const int m = min(kSbTile, R - r0);
#pragma unroll
    for (int q = 0; q < PF; ++q) {
      const int idx = threadIdx.x + q * kBlock, i = idx / (D / 4), c = idx % (D / 4);
      pf[q] = i < m ? reinterpret_cast<const float4*>(a.X[w])[(size_t)(r0 + i) * (D / 4) + c] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  if (my_tiles > 0) fetch(0);
  const int rb = wave % NRB, cb = wave / NRB;
  sas_f32x16 acc;
  for (int step = 0; step < 3 * my_tiles; ++step) {
    const int w = step % 3;
    if (w == 0)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    __syncthreads();
#pragma unroll
    for (int q = 0; q < PF; ++q) {
      const int idx = threadIdx.x + q * kBlock, i = idx / (D / 4), c = idx % (D / 4);
      float* d = Xs + i * SD + 4 * c;
      d[0] = pf[q].x; d[1] = pf[q].y; d[2] = pf[q].z; d[3] = pf[q].w;
    }
    __syncthreads();
    if (step + 1 < 3 * my_tiles) fetch(step + 1);
    if (wave < NQ) {  // wave-uniform
      const float* ap = Xs + (rb * 32 + (lane & 31)) * SD + kh;                 // a(i, k) = X[i][k]
      const float* bp = Ws + w * D * SD + kh * SD + cb * 32 + (lane & 31);      // b(k, j) = W[k][j]
      // the residual values of this block are requested BEFORE the last product (and all of them before the first
      // store): res may alias Y, so a load behind a store of the epilogue would wait for that store -- 16 serialised
      // round trips per lane and tile in round 2's `Y[e] = acc + res[e]` loop
      float rv[16];
      if (w == 2) {
        const int tile = blockIdx.x + (step / 3) * gridDim.x;
        const int r0 = tile * kSbTile;
        const int m = min(kSbTile, R - r0);
        const int j = cb * 32 + (lane & 31);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int i = rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
          rv[r] = i < m ? a.res[(size_t)(r0 + i) * D + j] : 0.f;
        }
      }
#pragma unroll
      for (int k0 = 0; k0 < D; k0 += 32) {
        float av[16], bv[16];
#pragma unroll
        for (int t = 0; t < 16; ++t) {
          av[t] = ap[k0 + 2 * t];
          bv[t] = bp[(k0 + 2 * t) * SD];
        }
#pragma unroll
        for (int t = 0; t < 16; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t], bv[t], acc, 0, 0, 0);
      }
      if (w == 2) {
        const int tile = blockIdx.x + (step / 3) * gridDim.x;
        const int r0 = tile * kSbTile;
        const int m = min(kSbTile, R - r0);
        const int j = cb * 32 + (lane & 31);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int i = rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
          if (i < m) {
            const size_t e = (size_t)(r0 + i) * D + j;
            a.Y[e] = acc[r] + rv[r];
          }
        }
      }
    }
  }
}

// ---- the two projection kernels on 16 x 16 x 4 tiles, operands straight from global memory ----------------------
// Y^T = W X^T: a lane (i, g) = (lane & 15, lane >> 4) holds its row's columns 16 c + 4 g .. + 3 as ONE float4 per chunk --
// the B operand of every MFMA of the tile -- and receives Y[row i][16 n + 4 g .. + 3] as the accumulator: one float4
// store.  The weights wait in LDS in a layout whose ds_read_b128 is the A operand of four MFMAs (row stride D + 4: the
// sixteen lanes of a group hit sixteen different 4-bank groups).  No row tile in LDS, no barrier after the weights are
// staged: a wave streams its 16-row tiles on its own, the other waves of its SIMD cover its loads.  Workgroups of up
// to 16 waves stage the weights once per CU.  (The 32 x 32 x 2 kernels above staged every 64-row tile through LDS with
// scalar ds_write / ds_read and two barriers: 51 us for the QKV projection of 104 K rows whose MFMA floor is 16 us.)
constexpr int kSb16MaxWaves = 16;

template <int D>
__device__ __forceinline__ void sb16_load_rows(const float* __restrict__ X, int row, int R, int g, float (&x)[D / 16][4]) {
#pragma unroll
  for (int c = 0; c < D / 16; ++c) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row < R) v = *reinterpret_cast<const float4*>(X + (size_t)row * D + 16 * c + 4 * g);
    x[c][0] = v.x; x[c][1] = v.y; x[c][2] = v.z; x[c][3] = v.w;
  }
}

// acc[n] += W[16 n + i][.] . x over the D input columns, for the output tiles n = N0, N0 + 1 (two independent chains)
template <int D, int N0>
__device__ __forceinline__ void sb16_product_pair(const float* Wl, int i, int g, const float (&x)[D / 16][4], sas_f32x4& a0,
                                                  sas_f32x4& a1) {
  constexpr int S = D + 4;
#pragma unroll
  for (int c = 0; c < D / 16; ++c) {
    const float4 w0 = *reinterpret_cast<const float4*>(Wl + (16 * N0 + i) * S + 16 * c + 4 * g);
    const float4 w1 = *reinterpret_cast<const float4*>(Wl + (16 * (N0 + 1) + i) * S + 16 * c + 4 * g);
    a0 = sas_mfma16(w0.x, x[c][0], a0); a1 = sas_mfma16(w1.x, x[c][0], a1);
    a0 = sas_mfma16(w0.y, x[c][1], a0); a1 = sas_mfma16(w1.y, x[c][1], a1);
    a0 = sas_mfma16(w0.z, x[c][2], a0); a1 = sas_mfma16(w1.z, x[c][2], a1);
    a0 = sas_mfma16(w0.w, x[c][3], a0); a1 = sas_mfma16(w1.w, x[c][3], a1);
  }
}

// acc += keep * W[16 n + i][.] . x for ONE output tile n (runtime); keep = 0 drops this lane's output row from the product
template <int D>
__device__ __forceinline__ void sb16_product_one(const float* Wl, int n, int i, int g, const float (&x)[D / 16][4], float keep,
                                                 sas_f32x4& a0) {
  constexpr int S = D + 4;
#pragma unroll
  for (int c = 0; c < D / 16; ++c) {
    const float4 w0 = *reinterpret_cast<const float4*>(Wl + (16 * n + i) * S + 16 * c + 4 * g);
    a0 = sas_mfma16(w0.x * keep, x[c][0], a0);
    a0 = sas_mfma16(w0.y * keep, x[c][1], a0);
    a0 = sas_mfma16(w0.z * keep, x[c][2], a0);
    a0 = sas_mfma16(w0.w * keep, x[c][3], a0);
  }
}

// Y_w = X W_w^T + b_w for NW projections of one pass over X (3: q, k, v; 2: k, v of a last layer; 1: q of the last rows)
template <int D, int NW>
__global__ __launch_bounds__(64 * kSb16MaxWaves) void sb_qkv16_kernel(SbLinArgs a) {
  constexpr int S = D + 4;
  extern __shared__ float lds[];
  float* Ws = lds;                // [NW][D][S]
  float* Bs = lds + NW * D * S;   // [NW][D]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 15, g = lane >> 4;
  const int nw = blockDim.x >> 6;
  const int R = a.off[a.B];
  const int tiles = (R + 15) >> 4;
  const int stride = (int)gridDim.x * nw;
  int t = (int)blockIdx.x * nw + wave;
  float x[D / 16][4], xn[D / 16][4];
  if (t < tiles) sb16_load_rows<D>(a.X, 16 * t + i, R, g, x);   // the first tile's rows travel while the weights are staged
  for (int idx = threadIdx.x; idx < NW * D * (D / 4); idx += blockDim.x) {
    const int w = idx / (D * (D / 4)), rem = idx % (D * (D / 4)), o = rem / (D / 4), c4 = rem % (D / 4);
    *reinterpret_cast<float4*>(Ws + (w * D + o) * S + 4 * c4) = reinterpret_cast<const float4*>(a.W[w])[o * (D / 4) + c4];
  }
  for (int idx = threadIdx.x; idx < NW * D; idx += blockDim.x) Bs[idx] = a.bias[idx / D] ? a.bias[idx / D][idx % D] : 0.f;
  __syncthreads();
  for (; t < tiles; t += stride) {
    asm volatile("" ::: "memory");   // the weights are re-read from LDS per tile: hoisted out of this loop they are 192 registers (spills)
    const int row = 16 * t + i;
    if (t + stride < tiles) sb16_load_rows<D>(a.X, 16 * (t + stride) + i, R, g, xn);   // requested before this tile's stores
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      const float* Wl = Ws + w * D * S;
      float* Y = a.Y[w];
#pragma unroll
      for (int n = 0; n < D / 16; n += 2) {
        sas_f32x4 a0 = sas_zero4(), a1 = sas_zero4();
        if (n == 0) sb16_product_pair<D, 0>(Wl, i, g, x, a0, a1);
        else if (n == 2) sb16_product_pair<D, (D >= 64 ? 2 : 0)>(Wl, i, g, x, a0, a1);
        const float4 b0 = *reinterpret_cast<const float4*>(Bs + w * D + 16 * n + 4 * g);
        const float4 b1 = *reinterpret_cast<const float4*>(Bs + w * D + 16 * (n + 1) + 4 * g);
        if (row < R) {
          *reinterpret_cast<float4*>(Y + (size_t)row * D + 16 * n + 4 * g) = make_float4(a0[0] + b0.x, a0[1] + b0.y, a0[2] + b0.z, a0[3] + b0.w);
          *reinterpret_cast<float4*>(Y + (size_t)row * D + 16 * (n + 1) + 4 * g) = make_float4(a1[0] + b1.x, a1[1] + b1.y, a1[2] + b1.z, a1[3] + b1.w);
        }
      }
    }
#pragma unroll
    for (int c = 0; c < D / 16; ++c)
#pragma unroll
      for (int e = 0; e < 4; ++e) x[c][e] = xn[c][e];
  }
}

// Y = res + X_0 W_0 + X_1 W_1 + X_2 W_2 (dx = dy . W): the same tiles with the weights transposed while they are staged
// (NIN = 3: dq, dk, dv with the residual; 2: dk, dv of a last layer whose queries are the last rows only; 1: those rows' dq)
template <int D, int NIN>
__global__ __launch_bounds__(64 * kSb16MaxWaves) void sb_sum3_16_kernel(SbSum3Args a) {
  constexpr int S = D + 4;
  extern __shared__ float lds[];
  float* Ws = lds;               // [NIN][D (in)][S (out)]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 15, g = lane >> 4;
  const int nw = blockDim.x >> 6;
  const int R = a.off[a.B];
  const int tiles = (R + 15) >> 4;
  const int stride = (int)gridDim.x * nw;
  int t = (int)blockIdx.x * nw + wave;
  float x0[D / 16][4], x1[D / 16][4], x2[D / 16][4], rs[D / 16][4];
  auto fetch = [&](int tile) {
    const int row = 16 * tile + i;
    sb16_load_rows<D>(a.X[0], row, R, g, x0);
    if (NIN > 1) sb16_load_rows<D>(a.X[1], row, R, g, x1);
    if (NIN > 2) sb16_load_rows<D>(a.X[2], row, R, g, x2);
    // (res may be Y: a lane reads exactly the float4s it writes, all before its first store; null: no residual)
    sb16_load_rows<D>(a.res, row, a.res ? R : 0, g, rs);
  };
  if (t < tiles) fetch(t);   // the first tile's rows travel while the weights are staged
  for (int idx = threadIdx.x; idx < NIN * D * D; idx += blockDim.x) {
    const int w = idx / (D * D), rem = idx % (D * D), o = rem / D, in = rem % D;
    Ws[(w * D + in) * S + o] = a.W[w][rem];
  }
  __syncthreads();
  bool first = true;
  for (; t < tiles; t += stride) {
    asm volatile("" ::: "memory");   // (as above)
    const int row = 16 * t + i;
    if (!first) fetch(t);
    first = false;
#pragma unroll
    for (int n = 0; n < D / 16; n += 2) {
      sas_f32x4 a0 = sas_zero4(), a1 = sas_zero4();
      if (n == 0) {
        sb16_product_pair<D, 0>(Ws, i, g, x0, a0, a1);
        if (NIN > 1) sb16_product_pair<D, 0>(Ws + D * S, i, g, x1, a0, a1);
        if (NIN > 2) sb16_product_pair<D, 0>(Ws + 2 * D * S, i, g, x2, a0, a1);
      } else if (n == 2) {
        sb16_product_pair<D, (D >= 64 ? 2 : 0)>(Ws, i, g, x0, a0, a1);
        if (NIN > 1) sb16_product_pair<D, (D >= 64 ? 2 : 0)>(Ws + D * S, i, g, x1, a0, a1);
        if (NIN > 2) sb16_product_pair<D, (D >= 64 ? 2 : 0)>(Ws + 2 * D * S, i, g, x2, a0, a1);
      }
      if (row < R) {
        *reinterpret_cast<float4*>(a.Y + (size_t)row * D + 16 * n + 4 * g) =
            make_float4(a0[0] + rs[n][0], a0[1] + rs[n][1], a0[2] + rs[n][2], a0[3] + rs[n][3]);
        *reinterpret_cast<float4*>(a.Y + (size_t)row * D + 16 * (n + 1) + 4 * g) =
            make_float4(a1[0] + rs[n + 1][0], a1[1] + rs[n + 1][1], a1[2] + rs[n + 1][2], a1[3] + rs[n + 1][3]);
      }
    }
  }
}

// RC_SAS_ROWS16=0: the 32 x 32 x 2 projection kernels with LDS row tiles (rounds 1-3), for A/B timing and the equivalence test
static bool sb_rows16() {
  const char* v = getenv("RC_SAS_ROWS16");
  return !(v && v[0] == '0');
}
// launch geometry of the 16-row-tile kernels: 16 waves per workgroup, one workgroup per CU once there are tiles for all of them
static void sb_rows16_geometry(int64_t rmax, int* grid, int* block) {
  const int64_t tiles = (rmax + 15) / 16;
  int nw = tiles >= 256 * kSb16MaxWaves ? kSb16MaxWaves : 4;
  if (const char* v = getenv("RC_SB16_WAVES")) {   // experiment switch: 4 / 8 / 16 waves per workgroup
    const int q = atoi(v);
    if (q == 4 || q == 8 || q == 16) nw = q;
  }
  int64_t gr = (tiles + nw - 1) / nw;
  const int64_t cap = 256 * (int64_t)(kSb16MaxWaves / nw < 3 ? kSb16MaxWaves / nw : 3);
  if (gr > cap) gr = cap;
  *grid = (int)(gr < 1 ? 1 : gr);
  *block = 64 * nw;
}

// ---- LayerNorm over rows: z = A (+ Bv) -> xhat, rstd, y = w * xhat + b ----------------------------------

// Training-mode dropout of the two residual branches of a TransformerLayer (utils/layers.py:104,110 dropout1 on the
// attention context, :114,117 dropout2 on the FFN output; both feed `layer_norm(drop(branch) + residual)`): the mask
// is never stored.  Element (compact row r, feature f) of site s = 2 * layer + {0: dropout1, 1: dropout2} is dropped
// iff word (f & 3) of Philox4x32-10(key = seed, counter = (r, s * D/4 + (f >> 2))) < thresh; kept values are scaled by
// 1 / (1 - p).  The LayerNorm kernels regenerate it: forward masks the branch before the residual add, backward
// emits the masked gradient for the branch next to the unmasked one for the residual path.  seed == nullptr: off.
struct SbDrop {
  const uint64_t* seed;   // device memory: a captured step replays with a fresh mask once the host bumps it
  uint32_t thresh;
  float scale;
  uint32_t site;
};

template <int D>
__device__ __forceinline__ float4 sb_drop_keep4(const SbDrop& dr, uint64_t seed, int64_t r, int l) {
  uint32_t wd[4];
  philox4x32_10(seed, (uint64_t)r, dr.site * (uint32_t)(D / 4) + (uint32_t)l, wd);
  return make_float4(wd[0] < dr.thresh ? 0.f : dr.scale, wd[1] < dr.thresh ? 0.f : dr.scale,
                     wd[2] < dr.thresh ? 0.f : dr.scale, wd[3] < dr.thresh ? 0.f : dr.scale);
}

template <int D>
__global__ __launch_bounds__(kBlock) void sb_ln_fwd_kernel(const float* __restrict__ A, const float* __restrict__ Bv,
                                                           const float* __restrict__ w, const float* __restrict__ bb,
                                                           const int32_t* __restrict__ off, int B,
                                                           float* __restrict__ xhat, float* __restrict__ rstd,
                                                           float* __restrict__ y, SbDrop dr) {
  constexpr int LPR = D / 4;
  const int l = threadIdx.x % LPR;
  const int R = off[B];
  const float4 wv = reinterpret_cast<const float4*>(w)[l], bv = reinterpret_cast<const float4*>(bb)[l];
  const uint64_t seed = dr.seed ? *dr.seed : 0;
  for (int64_t r = (int64_t)blockIdx.x * (kBlock / LPR) + threadIdx.x / LPR; r < R; r += (int64_t)gridDim.x * (kBlock / LPR)) {
    float4 z = reinterpret_cast<const float4*>(A)[r * LPR + l];
    if (dr.seed) {
      const float4 kp = sb_drop_keep4<D>(dr, seed, r, l);
      z.x *= kp.x; z.y *= kp.y; z.z *= kp.z; z.w *= kp.w;
    }
    if (Bv) {
      const float4 t = reinterpret_cast<const float4*>(Bv)[r * LPR + l];
      z.x += t.x; z.y += t.y; z.z += t.z; z.w += t.w;
    }
    const float mu = row_allreduce_sum<LPR>((z.x + z.y) + (z.z + z.w)) / D;
    const float cx = z.x - mu, cy = z.y - mu, cz = z.z - mu, cw = z.w - mu;
    const float var = row_allreduce_sum<LPR>(fmaf(cx, cx, fmaf(cy, cy, fmaf(cz, cz, cw * cw)))) / D;
    const float rs = 1.0f / sqrtf(var + kLnEps);
    const float4 xh = make_float4(cx * rs, cy * rs, cz * rs, cw * rs);
    reinterpret_cast<float4*>(xhat)[r * LPR + l] = xh;
    if (l == 0) rstd[r] = rs;
    reinterpret_cast<float4*>(y)[r * LPR + l] =
        make_float4(fmaf(xh.x, wv.x, bv.x), fmaf(xh.y, wv.y, bv.y), fmaf(xh.z, wv.z, bv.z), fmaf(xh.w, wv.w, bv.w));
  }
}

// LayerNorm backward in place: G holds dY on entry, dZ on exit (with dropout: Gdrop = mask * dZ as well);
// per-workgroup partial d(weight), d(bias) go to
// gw[blockIdx][:D], gb[blockIdx][:D] (stride = part_stride floats between workgroups)
template <int D>
__global__ __launch_bounds__(kBlock) void sb_ln_bwd_kernel(float* __restrict__ G, const float* __restrict__ xhat,
                                                           const float* __restrict__ rstd, const float* __restrict__ w,
                                                           const int32_t* __restrict__ off, int B,
                                                           float* __restrict__ gw, float* __restrict__ gb,
                                                           size_t part_stride, SbDrop dr, float* __restrict__ Gdrop) {
  constexpr int LPR = D / 4, GPB = kBlock / LPR;
  __shared__ float s_red[2][GPB][D + 1];
  const int l = threadIdx.x % LPR, grp = threadIdx.x / LPR;
  const int R = off[B];
  const float4 wv = reinterpret_cast<const float4*>(w)[l];
  const uint64_t seed = dr.seed ? *dr.seed : 0;
  float4 aw = make_float4(0.f, 0.f, 0.f, 0.f), ab = aw;
  for (int64_t r = (int64_t)blockIdx.x * GPB + grp; r < R; r += (int64_t)gridDim.x * GPB) {
    const float4 g = reinterpret_cast<const float4*>(G)[r * LPR + l];
    const float4 xh = reinterpret_cast<const float4*>(xhat)[r * LPR + l];
    aw.x = fmaf(g.x, xh.x, aw.x); aw.y = fmaf(g.y, xh.y, aw.y); aw.z = fmaf(g.z, xh.z, aw.z); aw.w = fmaf(g.w, xh.w, aw.w);
    ab.x += g.x; ab.y += g.y; ab.z += g.z; ab.w += g.w;
    const float4 dx = make_float4(g.x * wv.x, g.y * wv.y, g.z * wv.z, g.w * wv.w);
    const float m1 = row_allreduce_sum<LPR>((dx.x + dx.y) + (dx.z + dx.w)) / D;
    const float m2 = row_allreduce_sum<LPR>(fmaf(dx.x, xh.x, fmaf(dx.y, xh.y, fmaf(dx.z, xh.z, dx.w * xh.w)))) / D;
    const float rs = rstd[r];
    const float4 dz = make_float4(rs * (dx.x - m1 - xh.x * m2), rs * (dx.y - m1 - xh.y * m2), rs * (dx.z - m1 - xh.z * m2),
                                  rs * (dx.w - m1 - xh.w * m2));
    reinterpret_cast<float4*>(G)[r * LPR + l] = dz;
    if (dr.seed) {  // gradient of the dropped branch: the same mask as the forward pass
      const float4 kp = sb_drop_keep4<D>(dr, seed, r, l);
      reinterpret_cast<float4*>(Gdrop)[r * LPR + l] = make_float4(dz.x * kp.x, dz.y * kp.y, dz.z * kp.z, dz.w * kp.w);
    }
  }
  float* sw = &s_red[0][grp][4 * l];
  float* sb = &s_red[1][grp][4 * l];
  sw[0] = aw.x; sw[1] = aw.y; sw[2] = aw.z; sw[3] = aw.w;
  sb[0] = ab.x; sb[1] = ab.y; sb[2] = ab.z; sb[3] = ab.w;
  __syncthreads();
  for (int k = threadIdx.x; k < 2 * D; k += kBlock) {  // fixed order over the groups
    const int which = k / D, c = k % D;
    float t = 0.f;
    for (int g2 = 0; g2 < GPB; ++g2) t += s_red[which][g2][c];
    (which == 0 ? gw : gb)[(size_t)blockIdx.x * part_stride + c] = t;
  }
}

// ---- the post-attention half of a TransformerLayer in ONE row-space kernel (utils/layers.py:110-118) -------------------
//   y1 = LayerNorm1(drop1(ctx) + x);  h = relu(y1 W1^T + b1);  t = h W2^T + b2;  xnext = LayerNorm2(drop2(t) + y1)
// Round 2 ran this as four launches (LayerNorm, two sb_linear, LayerNorm) that handed [R, D] intermediates to each other
// through HBM: 11 row passes and four fill / drain phases for 16 + 29 + 29 + 16 us at config 3.  Here a 64-row tile stays in
// LDS from the first LayerNorm to the second: both weight matrices are resident, the row statistics run one lane-group
// per row on the staged tile, the two products are the same 32x32x2 MFMA tile loops (identical summation order), and
// only what the backward pass reads later leaves the CU (xhat1, rstd1, y1, h, xhat2, rstd2) beside the layer output.
// The next tile's ctx / x rows travel while the current tile is multiplied.
struct SbBlockArgs {
  const float* ctx;   // [R, D] attention output
  const float* x;     // [R, D] layer input (residual)
  const float *ln1w, *ln1b, *W1, *b1, *W2, *b2, *ln2w, *ln2b;
  float *xh1, *rstd1, *y1, *h, *xh2, *rstd2, *xnext;
  const int32_t* off;
  int B;
  SbDrop dr;          // dr.site = 2 * layer (dropout1); dropout2 uses site + 1
  const int64_t* row_len = nullptr;   // sb_block16_fwd_kernel, one row per sequence: xnext[row] = 0 where row_len[row] <= 0 (empty history)
  // sb_block16_fwd_kernel on the K / V-free last-row path: the attention output is formed here, ctx[row, o] = Wv[o, :] . xbar[row, h(o), :]
  // + bv[o] (zero for an empty history), instead of being read from `ctx` (sas_last_row.hpp: ctx_h = Wv_h xbar_h + bv_h)
  const float* xbar = nullptr;        // [R, H, D]
  const float *Wv = nullptr, *bv = nullptr;
  int H = 0;
};

template <int D>
__global__ __launch_bounds__(kBlock) void sb_block_fwd_kernel(SbBlockArgs a) {
  constexpr int SD = D + 1, LPR = D / 4, GPB = kBlock / LPR, NPASS = kSbTile / GPB;
  extern __shared__ float lds[];
  float* W1s = lds;                       // [D][SD]
  float* W2s = W1s + D * SD;              // [D][SD]
  float* Ys = W2s + D * SD;               // [kSbTile][SD]: y1, then (in place) the FFN output t
  float* Hs = Ys + kSbTile * SD;          // [kSbTile][SD]
  float* Bs = Hs + kSbTile * SD;          // [2][D]: b1, b2 (a global load behind the epilogues' global stores would wait for them)
  for (int idx = threadIdx.x; idx < D * D; idx += kBlock) {
    W1s[(idx / D) * SD + idx % D] = a.W1[idx];
    W2s[(idx / D) * SD + idx % D] = a.W2[idx];
  }
  for (int idx = threadIdx.x; idx < D; idx += kBlock) {
    Bs[idx] = a.b1[idx];
    Bs[D + idx] = a.b2[idx];
  }
  const int l = threadIdx.x % LPR, grp = threadIdx.x / LPR;
  const int R = a.off[a.B];
  const int tiles = (R + kSbTile - 1) / kSbTile;
  const bool drop = a.dr.seed != nullptr;
  const uint64_t seed = drop ? *a.dr.seed : 0;
  SbDrop dr1 = a.dr, dr2 = a.dr;
  dr2.site = a.dr.site + 1u;
  const float4 w1v = reinterpret_cast<const float4*>(a.ln1w)[l], b1v = reinterpret_cast<const float4*>(a.ln1b)[l];
  const float4 w2v = reinterpret_cast<const float4*>(a.ln2w)[l], b2v = reinterpret_cast<const float4*>(a.ln2b)[l];
  float4 pc[NPASS], px[NPASS];
  auto fetch = [&](int tile) {
    const int r0 = tile * kSbTile;
#pragma unroll
    for (int q = 0; q < NPASS; ++q) {
      const int r = r0 + q * GPB + grp;
      if (r < R) {
        pc[q] = reinterpret_cast<const float4*>(a.ctx)[(size_t)r * LPR + l];
        px[q] = reinterpret_cast<const float4*>(a.x)[(size_t)r * LPR + l];
      }
    }
  };
#ifdef RC_X_TIMING
  uint64_t tstamp[6] = {0, 0, 0, 0, 0, 0};
  uint64_t t_prev = wall_clock64();
  const uint64_t t_begin = t_prev;
  int n_my_tiles = 0;
#define RC_STAMP(k) do { const uint64_t t_now = wall_clock64(); tstamp[k] += t_now - t_prev; t_prev = t_now; } while (0)
#else
#define RC_STAMP(k)
#endif
  if ((int)blockIdx.x < tiles) fetch(blockIdx.x);
  for (int tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    const int r0 = tile * kSbTile;
    const int m = min(kSbTile, R - r0);
    __syncthreads();  // the previous tile's readers of Ys / Hs are done (and the weights are in place)
    RC_STAMP(0);
    // ---- LayerNorm1 on the staged rows: one lane-group per row
#pragma unroll
    for (int q = 0; q < NPASS; ++q) {
      const int i = q * GPB + grp;
      const int64_t r = (int64_t)r0 + i;
      float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
      if (i < m) {
        z = pc[q];
        if (drop) {
          const float4 kp = sb_drop_keep4<D>(dr1, seed, r, l);
          z.x *= kp.x; z.y *= kp.y; z.z *= kp.z; z.w *= kp.w;
        }
        z.x += px[q].x; z.y += px[q].y; z.z += px[q].z; z.w += px[q].w;
      }
      const float mu = row_allreduce_sum<LPR>((z.x + z.y) + (z.z + z.w)) / D;
      const float cx = z.x - mu, cy = z.y - mu, cz = z.z - mu, cw = z.w - mu;
      const float var = row_allreduce_sum<LPR>(fmaf(cx, cx, fmaf(cy, cy, fmaf(cz, cz, cw * cw)))) / D;
      const float rs = 1.0f / sqrtf(var + kLnEps);
      const float4 xh = make_float4(cx * rs, cy * rs, cz * rs, cw * rs);
      const float4 y = make_float4(fmaf(xh.x, w1v.x, b1v.x), fmaf(xh.y, w1v.y, b1v.y), fmaf(xh.z, w1v.z, b1v.z), fmaf(xh.w, w1v.w, b1v.w));
      float* d = Ys + i * SD + 4 * l;
      d[0] = y.x; d[1] = y.y; d[2] = y.z; d[3] = y.w;   // rows past m: finite filler, their outputs are discarded
      if (i < m) {
        reinterpret_cast<float4*>(a.xh1)[(size_t)r * LPR + l] = xh;
        if (l == 0) a.rstd1[r] = rs;
        reinterpret_cast<float4*>(a.y1)[(size_t)r * LPR + l] = y;
      }
    }
    __syncthreads();
    RC_STAMP(1);
    if (tile + (int)gridDim.x < tiles) fetch(tile + gridDim.x);   // travels during the two products
    // ---- h = relu(y1 W1^T + b1)
    sas_mm(MatA{Ys, SD, 1}, MatB{W1s, 1, SD}, m, D, D, false, [&](int i, int j, float v) {
      v = fmaxf(v + Bs[j], 0.f);
      Hs[i * SD + j] = v;
      a.h[(size_t)(r0 + i) * D + j] = v;
    });
    __syncthreads();
    RC_STAMP(2);
    // ---- t = h W2^T + b2 (+ y1: without dropout the residual is added right here), in place over y1
    sas_mm(MatA{Hs, SD, 1}, MatB{W2s, 1, SD}, m, D, D, false, [&](int i, int j, float v) {
      v += Bs[D + j];
      if (!drop) v += Ys[i * SD + j];
      Ys[i * SD + j] = v;
    });
    __syncthreads();
    RC_STAMP(3);
    // ---- LayerNorm2
#pragma unroll
    for (int q = 0; q < NPASS; ++q) {
      const int i = q * GPB + grp;
      const int64_t r = (int64_t)r0 + i;
      const float* sp = Ys + i * SD + 4 * l;
      float4 z = make_float4(sp[0], sp[1], sp[2], sp[3]);
      if (drop && i < m) {  // drop2(t) + y1 (y1 was overwritten in LDS: read it back, L2-hot)
        const float4 kp = sb_drop_keep4<D>(dr2, seed, r, l);
        const float4 y = reinterpret_cast<const float4*>(a.y1)[(size_t)r * LPR + l];
        z.x *= kp.x; z.y *= kp.y; z.z *= kp.z; z.w *= kp.w;     // (the same two roundings as the unfused kernels)
        z.x += y.x; z.y += y.y; z.z += y.z; z.w += y.w;
      }
      const float mu = row_allreduce_sum<LPR>((z.x + z.y) + (z.z + z.w)) / D;
      const float cx = z.x - mu, cy = z.y - mu, cz = z.z - mu, cw = z.w - mu;
      const float var = row_allreduce_sum<LPR>(fmaf(cx, cx, fmaf(cy, cy, fmaf(cz, cz, cw * cw)))) / D;
      const float rs = 1.0f / sqrtf(var + kLnEps);
      const float4 xh = make_float4(cx * rs, cy * rs, cz * rs, cw * rs);
      if (i < m) {
        reinterpret_cast<float4*>(a.xh2)[(size_t)r * LPR + l] = xh;
        if (l == 0) a.rstd2[r] = rs;
        reinterpret_cast<float4*>(a.xnext)[(size_t)r * LPR + l] =
            make_float4(fmaf(xh.x, w2v.x, b2v.x), fmaf(xh.y, w2v.y, b2v.y), fmaf(xh.z, w2v.z, b2v.z), fmaf(xh.w, w2v.w, b2v.w));
      }
    }
    RC_STAMP(4);
#ifdef RC_X_TIMING
    ++n_my_tiles;
#endif
  }
#ifdef RC_X_TIMING
  if ((blockIdx.x == 0 || blockIdx.x == 300) && threadIdx.x == 0)
    printf("sb_block_fwd wg %d: %d tiles, ticks (100 MHz): wait0 %llu ln1 %llu gemm1 %llu gemm2 %llu ln2 %llu total %llu\n", (int)blockIdx.x,
           n_my_tiles, (unsigned long long)tstamp[0], (unsigned long long)tstamp[1], (unsigned long long)tstamp[2],
           (unsigned long long)tstamp[3], (unsigned long long)tstamp[4], (unsigned long long)(wall_clock64() - t_begin));
#endif
#undef RC_STAMP
}

// ---- attention per sequence: ctx = softmax(causal(Q K^T / sqrt(dk))) V, head by head ---------------------

struct SbAttnArgs {
  const float *q, *k, *v;   // [R, D]
  float* ctx;               // fwd out [R, D]
  const float* dctx;        // bwd in  [R, D]
  float *dq, *dk, *dv;      // bwd out [R, D]
  const int64_t* lengths;
  const int32_t* off;
  int B, L, n_heads, lp;
  const int32_t* seq_list;  // sequences of this launch (nullptr: all), see sas_bucket_kernel
  const int32_t* seq_count;
};

}  // namespace rc
#include "sas_attn_reg.hpp"   // register-resident attention (needs SbAttnArgs, sb_len)
#include "sas_last_row.hpp"   // the last block on one query row per sequence, without keys / values
namespace rc {

__host__ __device__ inline int sb_buf_floats(int D, int lp) { return lp * ((D + 1) > (lp + 1) ? (D + 1) : (lp + 1)); }

template <int D>
__device__ __forceinline__ void sb_load_rows(float* dst, const float* __restrict__ src, int64_t r0, int n) {
  constexpr int SD = D + 1;
  for (int idx = threadIdx.x; idx < n * (D / 4); idx += kBlock) {
    const int i = idx / (D / 4), c = idx % (D / 4);
    const float4 v = reinterpret_cast<const float4*>(src)[(size_t)(r0 + i) * (D / 4) + c];
    float* d = dst + i * SD + 4 * c;
    d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
  }
}
template <int D>
__device__ __forceinline__ void sb_store_rows(float* __restrict__ dst, const float* src, int64_t r0, int n) {
  constexpr int SD = D + 1;
  for (int idx = threadIdx.x; idx < n * (D / 4); idx += kBlock) {
    const int i = idx / (D / 4), c = idx % (D / 4);
    const float* s = src + i * SD + 4 * c;
    reinterpret_cast<float4*>(dst)[(size_t)(r0 + i) * (D / 4) + c] = make_float4(s[0], s[1], s[2], s[3]);
  }
}

template <int D>
__global__ __launch_bounds__(kBlock) void sb_attn_fwd_kernel(SbAttnArgs a) {
  constexpr int SD = D + 1;
  const int LP = a.lp, SA = LP + 1, BUF = sb_buf_floats(D, LP);
  extern __shared__ float lds[];
  float *Q = lds, *K = Q + BUF, *V = K + BUF, *A = V + BUF, *C = A + BUF;
  const int dk = D / a.n_heads;
  const float sqrt_dk = sqrtf((float)dk);
  const int todo = a.seq_count ? *a.seq_count : a.B;
  for (int w = blockIdx.x; w < todo; w += gridDim.x) {
    const int b = a.seq_list ? a.seq_list[w] : w;
    const int n = sb_len(a.lengths, b, a.L);
    if (n == 0) continue;  // workgroup-uniform
    const int64_t r0 = a.off[b];
    sb_load_rows<D>(Q, a.q, r0, n);
    sb_load_rows<D>(K, a.k, r0, n);
    sb_load_rows<D>(V, a.v, r0, n);
    __syncthreads();
    for (int hh = 0; hh < a.n_heads; ++hh) {
      sas_attn_probs<D>(A, Q, K, n, hh, dk, sqrt_dk, SA);
      const int hc = hh * dk;
      sas_mm(MatA{A, SA, 1}, MatB{V + hc, SD, 1}, n, dk, n, false, [&](int i, int c, float v) { C[i * SD + hc + c] = v; });
      __syncthreads();
    }
    sb_store_rows<D>(a.ctx, C, r0, n);
    __syncthreads();
  }
}

// backward of the above: dq, dk, dv from dctx (probabilities are recomputed from q, k)
template <int D>
__global__ __launch_bounds__(kBlock) void sb_attn_bwd_kernel(SbAttnArgs a) {
  constexpr int SD = D + 1;
  const int LP = a.lp, SA = LP + 1, BUF = sb_buf_floats(D, LP);
  extern __shared__ float lds[];
  float *Q = lds, *K = Q + BUF, *V = K + BUF, *G = V + BUF, *C = G + BUF, *A = C + BUF, *T = A + BUF;
  const int dk = D / a.n_heads;
  const float sqrt_dk = sqrtf((float)dk);
  const int todo = a.seq_count ? *a.seq_count : a.B;
  for (int w = blockIdx.x; w < todo; w += gridDim.x) {
    const int b = a.seq_list ? a.seq_list[w] : w;
    const int n = sb_len(a.lengths, b, a.L);
    if (n == 0) continue;
    const int64_t r0 = a.off[b];
    sb_load_rows<D>(Q, a.q, r0, n);
    sb_load_rows<D>(K, a.k, r0, n);
    sb_load_rows<D>(V, a.v, r0, n);
    sb_load_rows<D>(G, a.dctx, r0, n);
    __syncthreads();
    for (int hh = 0; hh < a.n_heads; ++hh) {  // dV, dK overwrite V, K in place; dQ goes to C
      const int hc = hh * dk;
      sas_attn_probs<D>(A, Q, K, n, hh, dk, sqrt_dk, SA);
      sas_mm(MatA{G + hc, SD, 1}, MatB{V + hc, 1, SD}, n, n, dk, true, [&](int i, int j, float v) { T[i * SA + j] = v; });
      __syncthreads();
      sas_mm(MatA{A, 1, SA}, MatB{G + hc, SD, 1}, n, dk, n, false, [&](int j, int c, float v) { V[j * SD + hc + c] = v; });
      {  // dS = A * (dA - rowsum(dA*A)) / sqrt(dk), in place in T, 0 above the diagonal
        const int rows_here = LP <= 32 ? 8 : 16;
        sas_softmax_bwd_rows(T, A, n, SA, sqrt_dk, (int)(threadIdx.x >> 6) * rows_here, rows_here);
      }
      __syncthreads();
      sas_mm(MatA{T, SA, 1}, MatB{K + hc, SD, 1}, n, dk, n, false, [&](int i, int c, float v) { C[i * SD + hc + c] = v; });
      __syncthreads();
      sas_mm(MatA{T, 1, SA}, MatB{Q + hc, SD, 1}, n, dk, n, false, [&](int j, int c, float v) { K[j * SD + hc + c] = v; });
      __syncthreads();
    }
    sb_store_rows<D>(a.dq, C, r0, n);
    sb_store_rows<D>(a.dk, K, r0, n);
    sb_store_rows<D>(a.dv, V, r0, n);
    __syncthreads();
  }
}

// ---- one WAVE per head: no workgroup barrier between the phases of a head -------------------------------------
// blockDim = 64 * n_heads (n_heads <= 4).  Q, K, V (and dctx) of the sequence are staged once for all heads; each
// wave owns its head's probability (and dA / dS) scratch and writes its column slice of the outputs straight to
// global memory.  Two barriers per sequence instead of three (five in the backward) per head.

template <int D>
__device__ __forceinline__ void sb_load_rows_n(float* dst, const float* __restrict__ src, int64_t r0, int n) {
  constexpr int SD = D + 1;
  for (int idx = threadIdx.x; idx < n * (D / 4); idx += blockDim.x) {
    const int i = idx / (D / 4), c = idx % (D / 4);
    const float4 v = reinterpret_cast<const float4*>(src)[(size_t)(r0 + i) * (D / 4) + c];
    float* d = dst + i * SD + 4 * c;
    d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
  }
}

// WPH waves per head.  WPH = 1: a head's phases need no workgroup barrier at all (the wave is alone on its scratch).
// WPH = 2 (the long length class, whose LDS footprint allows one or two workgroups per CU only): two waves share a
// head -- its 32x32 output blocks alternate between them, each takes 32 of the rows in the softmax phases -- and
// the phases are separated by barriers again, with twice the waves in flight.
template <typename Epi>
__device__ __forceinline__ void sb_mm_head(MatA A, MatB B, int M, int N, int K, bool causal, Epi epi, int sub, int wph) {
  sas_mm_part(A, B, M, N, K, causal, epi, sub, wph);
}

// Rows of the NEXT sequence travel (global -> registers) while the current one is multiplied: a workgroup holds one
// sequence in LDS and one or two workgroups fit a CU, so nothing else covers the load latency -- without this every
// sequence paid a full memory round trip between its two barriers.  Needs the full complement of 64 * 4 * WPH threads
// (n_heads = 4): PF float4 per thread and buffer cover lp <= 64 rows.
template <int D, int NBUF, int PF>
struct SbRowPrefetch {
  float4 v[NBUF][PF];
  int n;
  int64_t r0;
  __device__ __forceinline__ void fetch(const float* const* src, int n_, int64_t r0_) {
    n = n_;
    r0 = r0_;
#pragma unroll
    for (int q = 0; q < PF; ++q) {
      const int idx = threadIdx.x + q * blockDim.x, i = idx / (D / 4), c = idx % (D / 4);
      if (i < n) {
#pragma unroll
        for (int bq = 0; bq < NBUF; ++bq) v[bq][q] = reinterpret_cast<const float4*>(src[bq])[(size_t)(r0 + i) * (D / 4) + c];
      }
    }
  }
  __device__ __forceinline__ void stage(float* const* dst) const {
    constexpr int SD = D + 1;
#pragma unroll
    for (int q = 0; q < PF; ++q) {
      const int idx = threadIdx.x + q * blockDim.x, i = idx / (D / 4), c = idx % (D / 4);
      if (i < n) {
#pragma unroll
        for (int bq = 0; bq < NBUF; ++bq) {
          float* d = dst[bq] + i * SD + 4 * c;
          d[0] = v[bq][q].x; d[1] = v[bq][q].y; d[2] = v[bq][q].z; d[3] = v[bq][q].w;
        }
      }
    }
  }
};

// PF: float4 per thread and buffer of the prefetch (0 = none): ceil(lp * D / 4 / threads), chosen by the length class
template <int D, int WPH, int PF>
__global__ __launch_bounds__(256 * WPH) void sb_attn_fwd_wave_kernel(SbAttnArgs a) {
  constexpr int SD = D + 1;
  const int LP = a.lp, SA = LP + 1, BUF = sb_buf_floats(D, LP);
  extern __shared__ float lds[];
  float *Q = lds, *K = Q + BUF, *V = K + BUF;
  const int wave = threadIdx.x >> 6, hh = wave / WPH, sub = wave % WPH;
  float* A = V + BUF + hh * LP * SA;
  const int dk = D / a.n_heads, hc = hh * dk;
  const float sqrt_dk = sqrtf((float)dk);
  const int todo = a.seq_count ? *a.seq_count : a.B;
  const bool prefetch = PF > 0;   // (the launcher picks PF > 0 only with 256 * WPH threads and lp * D / 4 <= PF * threads)
  SbRowPrefetch<D, 3, (PF > 0 ? PF : 1)> pf;
  const float* const srcs[3] = {a.q, a.k, a.v};
  float* const dsts[3] = {Q, K, V};
  auto seq_of = [&](int w, int* n_out) -> int64_t {
    const int b = a.seq_list ? a.seq_list[w] : w;
    *n_out = sb_len(a.lengths, b, a.L);
    return a.off[b];
  };
  if (prefetch && (int)blockIdx.x < todo) {
    int n0;
    const int64_t r = seq_of(blockIdx.x, &n0);
    pf.fetch(srcs, n0, r);
  }
  for (int w = blockIdx.x; w < todo; w += gridDim.x) {
    int n;
    int64_t r0;
    if (prefetch) {
      n = pf.n;
      r0 = pf.r0;
      __syncthreads();  // the previous sequence's readers are done
      pf.stage(dsts);
      __syncthreads();
      if (w + (int)gridDim.x < todo) {  // the next sequence's rows travel during this one's products
        int nn;
        const int64_t r = seq_of(w + gridDim.x, &nn);
        pf.fetch(srcs, nn, r);
      }
      if (n == 0) continue;  // workgroup-uniform
    } else {
      r0 = seq_of(w, &n);
      if (n == 0) continue;  // workgroup-uniform
      __syncthreads();  // the previous sequence's readers are done
      sb_load_rows_n<D>(Q, a.q, r0, n);
      sb_load_rows_n<D>(K, a.k, r0, n);
      sb_load_rows_n<D>(V, a.v, r0, n);
      __syncthreads();
    }
    sb_mm_head(MatA{Q + hc, SD, 1}, MatB{K + hc, 1, SD}, n, n, dk, true, [&](int i, int j, float v) { A[i * SA + j] = sas_div_scale(v, sqrt_dk); },
               sub, WPH);
    if (WPH > 1) __syncthreads();
    if (32 * sub < n) sas_softmax_causal_rows(A, n, SA, 32 * sub, 32);  // two lanes per row
    if (WPH == 1 && n > 32) sas_softmax_causal_rows(A, n, SA, 32, 32);
    if (WPH > 1) __syncthreads();
    sb_mm_head(MatA{A, SA, 1}, MatB{V + hc, SD, 1}, n, dk, n, false,
               [&](int i, int c, float v) { a.ctx[(size_t)(r0 + i) * D + hc + c] = v; }, sub, WPH);
  }
}

template <int D, int WPH, int PF>
__global__ __launch_bounds__(256 * WPH) void sb_attn_bwd_wave_kernel(SbAttnArgs a) {
  constexpr int SD = D + 1;
  const int LP = a.lp, SA = LP + 1, BUF = sb_buf_floats(D, LP);
  extern __shared__ float lds[];
  float *Q = lds, *K = Q + BUF, *V = K + BUF, *G = V + BUF;
  const int wave = threadIdx.x >> 6, hh = wave / WPH, sub = wave % WPH;
  float* A = G + BUF + (2 * hh) * LP * SA;
  float* T = A + LP * SA;
  const int dk = D / a.n_heads, hc = hh * dk;
  const float sqrt_dk = sqrtf((float)dk);
  const int todo = a.seq_count ? *a.seq_count : a.B;
  const bool prefetch = PF > 0;
  SbRowPrefetch<D, 4, (PF > 0 ? PF : 1)> pf;
  const float* const srcs[4] = {a.q, a.k, a.v, a.dctx};
  float* const dsts[4] = {Q, K, V, G};
  auto seq_of = [&](int w, int* n_out) -> int64_t {
    const int b = a.seq_list ? a.seq_list[w] : w;
    *n_out = sb_len(a.lengths, b, a.L);
    return a.off[b];
  };
#ifdef RC_X_TIMING
  uint64_t ts[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  uint64_t tp = wall_clock64();
  int n_seq = 0, n_rows = 0;
#define RC_ST(k) do { const uint64_t tn = wall_clock64(); ts[k] += tn - tp; tp = tn; } while (0)
#else
#define RC_ST(k)
#endif
  if (prefetch && (int)blockIdx.x < todo) {
    int n0;
    const int64_t r = seq_of(blockIdx.x, &n0);
    pf.fetch(srcs, n0, r);
  }
  for (int w = blockIdx.x; w < todo; w += gridDim.x) {
    int n;
    int64_t r0;
    if (prefetch) {
      n = pf.n;
      r0 = pf.r0;
      __syncthreads();
      RC_ST(0);
      pf.stage(dsts);
      __syncthreads();
      RC_ST(1);
      if (w + (int)gridDim.x < todo) {
        int nn;
        const int64_t r = seq_of(w + gridDim.x, &nn);
        pf.fetch(srcs, nn, r);
      }
      if (n == 0) continue;
#ifdef RC_X_TIMING
      ++n_seq; n_rows += n;
#endif
    } else {
      r0 = seq_of(w, &n);
      if (n == 0) continue;
      __syncthreads();
      sb_load_rows_n<D>(Q, a.q, r0, n);
      sb_load_rows_n<D>(K, a.k, r0, n);
      sb_load_rows_n<D>(V, a.v, r0, n);
      sb_load_rows_n<D>(G, a.dctx, r0, n);
      __syncthreads();
    }
    sb_mm_head(MatA{Q + hc, SD, 1}, MatB{K + hc, 1, SD}, n, n, dk, true, [&](int i, int j, float v) { A[i * SA + j] = sas_div_scale(v, sqrt_dk); },
               sub, WPH);
    // dA = dCtx_h . V_h^T (lower triangle)
    sb_mm_head(MatA{G + hc, SD, 1}, MatB{V + hc, 1, SD}, n, n, dk, true, [&](int i, int j, float v) { T[i * SA + j] = v; }, sub, WPH);
    if (WPH > 1) __syncthreads();
    RC_ST(2);
    if (32 * sub < n) sas_softmax_causal_rows(A, n, SA, 32 * sub, 32);
    if (WPH == 1 && n > 32) sas_softmax_causal_rows(A, n, SA, 32, 32);
    if (WPH > 1) __syncthreads();
    RC_ST(3);
    // dV_h = A^T . dCtx_h
    sb_mm_head(MatA{A, 1, SA}, MatB{G + hc, SD, 1}, n, dk, n, false,
               [&](int j, int c, float v) { a.dv[(size_t)(r0 + j) * D + hc + c] = v; }, sub, WPH);
    RC_ST(4);
    // (no barrier: dV reads A and G only, dS below rewrites T and reads A)
    if (32 * sub < n) sas_softmax_bwd_rows(T, A, n, SA, sqrt_dk, 32 * sub, 32);  // dS in place in T
    if (WPH == 1 && n > 32) sas_softmax_bwd_rows(T, A, n, SA, sqrt_dk, 32, 32);
    if (WPH > 1) __syncthreads();
    RC_ST(5);
    // dQ_h = dS . K_h,  dK_h = dS^T . Q_h
    sb_mm_head(MatA{T, SA, 1}, MatB{K + hc, SD, 1}, n, dk, n, false,
               [&](int i, int c, float v) { a.dq[(size_t)(r0 + i) * D + hc + c] = v; }, sub, WPH);
    sb_mm_head(MatA{T, 1, SA}, MatB{Q + hc, SD, 1}, n, dk, n, false,
               [&](int j, int c, float v) { a.dk[(size_t)(r0 + j) * D + hc + c] = v; }, sub, WPH);
    RC_ST(6);
  }
#ifdef RC_X_TIMING
  if ((blockIdx.x == 0 || blockIdx.x == 100) && threadIdx.x == 0)
    printf("attn_bwd<WPH %d PF %d> lp %d wg %d: %d seqs %d rows; ticks(100MHz): wait %llu stage %llu QK+dA %llu softmax %llu dV %llu dS %llu dQ+dK %llu\n",
           WPH, PF, a.lp, (int)blockIdx.x, n_seq, n_rows, (unsigned long long)ts[0], (unsigned long long)ts[1], (unsigned long long)ts[2],
           (unsigned long long)ts[3], (unsigned long long)ts[4], (unsigned long long)ts[5], (unsigned long long)ts[6]);
#endif
#undef RC_ST
}

// ---- the same block on 16-row tiles without an LDS row tile (the pattern of sb_qkv16_kernel) ---------------------------------
// In the Y^T = W X^T form a product's accumulator comes out in the very layout the next product wants as its B operand
// (lane (i, g): row i, columns 16 n + 4 g .. + 3), so LayerNorm1 -> FFN1 -> FFN2 -> LayerNorm2 chain through REGISTERS: a row's
// statistics are the lane's sixteen values plus two cross-lane steps, the activations never touch LDS, and a wave walks its
// tiles without any barrier.  LDS holds the two weight matrices (row stride D + 4: one ds_read_b128 = the A operand of four
// MFMAs) and the six parameter vectors.
template <int D>
__device__ __forceinline__ void sb16_layernorm(float (&z)[D / 16][4], const float* lnw, const float* lnb, int g, float (&xh)[D / 16][4],
                                               float (&y)[D / 16][4], float* rs_out) {
  float sum = 0.f;
#pragma unroll
  for (int c = 0; c < D / 16; ++c) sum += (z[c][0] + z[c][1]) + (z[c][2] + z[c][3]);
  const float mu = sas_groups_sum(sum) / D;
  float sq = 0.f;
#pragma unroll
  for (int c = 0; c < D / 16; ++c)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      z[c][e] -= mu;
      sq = fmaf(z[c][e], z[c][e], sq);
    }
  const float rs = 1.0f / sqrtf(sas_groups_sum(sq) / D + kLnEps);
#pragma unroll
  for (int c = 0; c < D / 16; ++c) {
    const float4 w = *reinterpret_cast<const float4*>(lnw + 16 * c + 4 * g), b = *reinterpret_cast<const float4*>(lnb + 16 * c + 4 * g);
    const float wv[4] = {w.x, w.y, w.z, w.w}, bv[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      xh[c][e] = z[c][e] * rs;
      y[c][e] = fmaf(xh[c][e], wv[e], bv[e]);
    }
  }
  *rs_out = rs;
}

template <int D>
__device__ __forceinline__ void sb16_store_rows(float* __restrict__ Y, int row, int g, const float (&v)[D / 16][4]) {
#pragma unroll
  for (int c = 0; c < D / 16; ++c)
    *reinterpret_cast<float4*>(Y + (size_t)row * D + 16 * c + 4 * g) = make_float4(v[c][0], v[c][1], v[c][2], v[c][3]);
}

constexpr int kSb16BlockWaves = 8;   // (the block kernel holds five row sets: 256 registers per lane at 8 waves per workgroup)

template <int D>
__global__ __launch_bounds__(64 * kSb16BlockWaves) void sb_block16_fwd_kernel(SbBlockArgs a) {
  constexpr int NC = D / 16, S = D + 4;
  extern __shared__ float lds[];
  float* W1s = lds;               // [D][S]
  float* W2s = W1s + D * S;       // [D][S]
  float* Wvs = W2s + D * S;       // [D][S], only with a.xbar
  float* Ps = Wvs + (a.xbar ? D * S : 0);   // [7][D]: b1, b2, ln1w, ln1b, ln2w, ln2b, bv
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 15, g = lane >> 4;
  const int nw = blockDim.x >> 6;
  const int R = a.off[a.B];
  const int tiles = (R + 15) >> 4;
  const int stride = (int)gridDim.x * nw;
  int t = (int)blockIdx.x * nw + wave;
  float zc[NC][4], zx[NC][4];
  if (t < tiles) {   // the first tile's rows travel while the weights are staged
    if (!a.xbar) sb16_load_rows<D>(a.ctx, 16 * t + i, R, g, zc);
    sb16_load_rows<D>(a.x, 16 * t + i, R, g, zx);
  }
  for (int idx = threadIdx.x; idx < D * (D / 4); idx += blockDim.x) {
    const int o = idx / (D / 4), c4 = idx % (D / 4);
    *reinterpret_cast<float4*>(W1s + o * S + 4 * c4) = reinterpret_cast<const float4*>(a.W1)[idx];
    *reinterpret_cast<float4*>(W2s + o * S + 4 * c4) = reinterpret_cast<const float4*>(a.W2)[idx];
    if (a.xbar) *reinterpret_cast<float4*>(Wvs + o * S + 4 * c4) = reinterpret_cast<const float4*>(a.Wv)[idx];
  }
  for (int idx = threadIdx.x; idx < D; idx += blockDim.x) {
    Ps[idx] = a.b1[idx]; Ps[D + idx] = a.b2[idx];
    Ps[2 * D + idx] = a.ln1w[idx]; Ps[3 * D + idx] = a.ln1b[idx];
    Ps[4 * D + idx] = a.ln2w[idx]; Ps[5 * D + idx] = a.ln2b[idx];
    if (a.xbar) Ps[6 * D + idx] = a.bv[idx];
  }
  __syncthreads();
  const bool drop = a.dr.seed != nullptr;
  const uint64_t seed = drop ? *a.dr.seed : 0;
  SbDrop dr1 = a.dr, dr2 = a.dr;
  dr2.site = a.dr.site + 1u;
  bool first = true;
  for (; t < tiles; t += stride) {
    asm volatile("" ::: "memory");   // weights and parameters are re-read from LDS per tile (see sb_qkv16_kernel)
    const int row = 16 * t + i;
    const bool valid = row < R;
    if (!first) {
      if (!a.xbar) sb16_load_rows<D>(a.ctx, row, R, g, zc);
      sb16_load_rows<D>(a.x, row, R, g, zx);
    }
    first = false;
    if (a.xbar) {   // ctx tile n = Wv[16 n .., :] . xbar[row, head of those outputs, :] + bv (0 for an empty history)
      const int DK = D / a.H;
      const int hpt = DK >= 16 ? 1 : 16 / DK;   // heads per 16-output tile (dk = 8: two, each product masked to its output rows)
      const float live = (valid && !(a.row_len && a.row_len[row] <= 0)) ? 1.f : 0.f;
      float xb[NC][4];
      int cur = -1;
#pragma unroll
      for (int n = 0; n < NC; ++n) {
        sas_f32x4 acc = sas_zero4();
        for (int hl = 0; hl < hpt; ++hl) {
          const int h = (16 * n) / DK + hl;   // wave-uniform
          if (h != cur) {
            cur = h;
            const float* src = a.xbar + ((size_t)(valid ? row : 0) * a.H + h) * D;
#pragma unroll
            for (int c = 0; c < NC; ++c) {
              const float4 v = valid ? *reinterpret_cast<const float4*>(src + 16 * c + 4 * g) : make_float4(0.f, 0.f, 0.f, 0.f);
              xb[c][0] = v.x; xb[c][1] = v.y; xb[c][2] = v.z; xb[c][3] = v.w;
            }
          }
          sb16_product_one<D>(Wvs, n, i, g, xb, (hpt == 1 || i / DK == hl) ? 1.f : 0.f, acc);
        }
        const float4 b4 = *reinterpret_cast<const float4*>(Ps + 6 * D + 16 * n + 4 * g);
        zc[n][0] = fmaf(b4.x, live, acc[0]); zc[n][1] = fmaf(b4.y, live, acc[1]);
        zc[n][2] = fmaf(b4.z, live, acc[2]); zc[n][3] = fmaf(b4.w, live, acc[3]);
      }
    }
    // ---- y1 = LayerNorm1(drop1(ctx) + x)
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      if (drop) {
        const float4 kp = sb_drop_keep4<D>(dr1, seed, row, 4 * c + g);
        zc[c][0] *= kp.x; zc[c][1] *= kp.y; zc[c][2] *= kp.z; zc[c][3] *= kp.w;
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) zc[c][e] += zx[c][e];
    }
    float xh[NC][4], y1[NC][4], rs;
    sb16_layernorm<D>(zc, Ps + 2 * D, Ps + 3 * D, g, xh, y1, &rs);
    if (valid) {
      sb16_store_rows<D>(a.xh1, row, g, xh);
      sb16_store_rows<D>(a.y1, row, g, y1);
      if (g == 0) a.rstd1[row] = rs;
    }
    // ---- h = relu(y1 W1^T + b1): the accumulator of output tile n is the lane's float4 of h[row][16 n + 4 g ..]
    float h[NC][4];
#pragma unroll
    for (int n = 0; n < NC; n += 2) {
      sas_f32x4 a0 = sas_zero4(), a1 = sas_zero4();
      if (n == 0) sb16_product_pair<D, 0>(W1s, i, g, y1, a0, a1);
      else if (n == 2) sb16_product_pair<D, (D >= 64 ? 2 : 0)>(W1s, i, g, y1, a0, a1);
      const float4 b0 = *reinterpret_cast<const float4*>(Ps + 16 * n + 4 * g), b1 = *reinterpret_cast<const float4*>(Ps + 16 * (n + 1) + 4 * g);
      h[n][0] = fmaxf(a0[0] + b0.x, 0.f); h[n][1] = fmaxf(a0[1] + b0.y, 0.f); h[n][2] = fmaxf(a0[2] + b0.z, 0.f); h[n][3] = fmaxf(a0[3] + b0.w, 0.f);
      h[n + 1][0] = fmaxf(a1[0] + b1.x, 0.f); h[n + 1][1] = fmaxf(a1[1] + b1.y, 0.f); h[n + 1][2] = fmaxf(a1[2] + b1.z, 0.f); h[n + 1][3] = fmaxf(a1[3] + b1.w, 0.f);
    }
    if (valid) sb16_store_rows<D>(a.h, row, g, h);
    // ---- t = h W2^T + b2; z2 = drop2(t) + y1  (into zc)
#pragma unroll
    for (int n = 0; n < NC; n += 2) {
      sas_f32x4 a0 = sas_zero4(), a1 = sas_zero4();
      if (n == 0) sb16_product_pair<D, 0>(W2s, i, g, h, a0, a1);
      else if (n == 2) sb16_product_pair<D, (D >= 64 ? 2 : 0)>(W2s, i, g, h, a0, a1);
      const float4 b0 = *reinterpret_cast<const float4*>(Ps + D + 16 * n + 4 * g), b1 = *reinterpret_cast<const float4*>(Ps + D + 16 * (n + 1) + 4 * g);
      const float bb0[4] = {b0.x, b0.y, b0.z, b0.w}, bb1[4] = {b1.x, b1.y, b1.z, b1.w};
      float4 k0 = make_float4(1.f, 1.f, 1.f, 1.f), k1 = k0;
      if (drop) {
        k0 = sb_drop_keep4<D>(dr2, seed, row, 4 * n + g);
        k1 = sb_drop_keep4<D>(dr2, seed, row, 4 * (n + 1) + g);
      }
      const float kk0[4] = {k0.x, k0.y, k0.z, k0.w}, kk1[4] = {k1.x, k1.y, k1.z, k1.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float v0 = a0[e] + bb0[e], v1 = a1[e] + bb1[e];
        if (drop) { v0 *= kk0[e]; v1 *= kk1[e]; }
        zc[n][e] = v0 + y1[n][e];
        zc[n + 1][e] = v1 + y1[n + 1][e];
      }
    }
    // ---- xnext = LayerNorm2(z2)
    float xh2[NC][4], yo[NC][4], rs2;
    sb16_layernorm<D>(zc, Ps + 4 * D, Ps + 5 * D, g, xh2, yo, &rs2);
    if (valid) {
      if (a.row_len && a.row_len[row] <= 0) {
#pragma unroll
        for (int c = 0; c < NC; ++c) yo[c][0] = yo[c][1] = yo[c][2] = yo[c][3] = 0.f;
      }
      sb16_store_rows<D>(a.xh2, row, g, xh2);
      sb16_store_rows<D>(a.xnext, row, g, yo);
      if (g == 0) a.rstd2[row] = rs2;
    }
  }
}

// ---- weight gradients: dW[o][k] = sum_r dY[r][o] X[r][k], db[o] = sum_r dY[r][o] --------------------------
// NP (dY, X) pairs that share X are handled by one launch (dq, dk, dv against the layer input).  Each workgroup
// keeps its accumulators in registers over all of its row tiles and writes ONE partial block per pair.

struct SbWgradArgs {
  const float* dY[3];
  const float* X;
  float* gW[3];  // this workgroup's slice = gW[p] + blockIdx.x * part_stride
  float* gb[3];
  size_t part_stride;
  const int32_t* off;
  int B;
};

template <int D, int NP>
__global__ __launch_bounds__(kBlock) void sb_wgrad_kernel(SbWgradArgs a) {
  constexpr int SD = D + 1, NB = D / 32;  // NB x NB output blocks of 32 x 32
  extern __shared__ float lds[];
  float* Xs = lds;                 // [kSbTile][SD]
  float* Ys = lds + kSbTile * SD;  // [NP][kSbTile][SD]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int R = a.off[a.B];
  const int tiles = (R + kSbTile - 1) / kSbTile;
  // output blocks q = (pair, rb, cb) are dealt round-robin to the waves; NP * NB * NB <= 12 -> at most 3 per wave
  constexpr int NQ = NP * NB * NB, QPW = (NQ + 3) / 4;
  sas_f32x16 acc[QPW];
#pragma unroll
  for (int s = 0; s < QPW; ++s)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[s][r] = 0.f;
  float bsum = 0.f;   // thread t < NP * D: db of pair t / D, column t % D (one wave per pair: the column sums of a tile do not pile up on wave 0)

  // the next tile's rows (X and the NP dY's) are requested while the current tile is multiplied
  constexpr int PF = kSbTile * (D / 4) / kBlock;
  float4 pfx[PF], pfy[NP][PF];
  auto fetch = [&](int tile) {
    const int r0 = tile * kSbTile;
    const int m = min(kSbTile, R - r0);
#pragma unroll
    for (int q = 0; q < PF; ++q) {
      const int idx = threadIdx.x + q * kBlock, i = idx / (D / 4), c = idx % (D / 4);
      // rows past m are zero-filled (assigned, not selected from a named zero: a const float4 in the ternary was kept in a stack slot --
      // 32 bytes of scratch per lane for one dead store)
      pfx[q] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (i < m) pfx[q] = reinterpret_cast<const float4*>(a.X)[(size_t)(r0 + i) * (D / 4) + c];
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        pfy[p][q] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i < m) pfy[p][q] = reinterpret_cast<const float4*>(a.dY[p])[(size_t)(r0 + i) * (D / 4) + c];
      }
    }
  };
  if ((int)blockIdx.x < tiles) fetch(blockIdx.x);
  for (int tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    const int r0 = tile * kSbTile;
    const int m = min(kSbTile, R - r0);
    __syncthreads();
#pragma unroll
    for (int q = 0; q < PF; ++q) {
      const int idx = threadIdx.x + q * kBlock, i = idx / (D / 4), c = idx % (D / 4);
      float* d = Xs + i * SD + 4 * c;
      d[0] = pfx[q].x; d[1] = pfx[q].y; d[2] = pfx[q].z; d[3] = pfx[q].w;
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        float* e = Ys + (p * kSbTile + i) * SD + 4 * c;
        e[0] = pfy[p][q].x; e[1] = pfy[p][q].y; e[2] = pfy[p][q].z; e[3] = pfy[p][q].w;
      }
    }
    __syncthreads();
    if (tile + (int)gridDim.x < tiles) fetch(tile + gridDim.x);
#pragma unroll
    for (int s = 0; s < QPW; ++s) {
      const int q = wave + 4 * s;
      if (q < NQ) {  // wave-uniform
        const int p = q / (NB * NB), rb = (q % (NB * NB)) % NB, cb = (q % (NB * NB)) / NB;
        const int kh = lane >> 5;
        // a(o, r) = dY[r][o]: lanes run along o;  b(r, k) = X[r][k]: lanes run along k
        const float* ap = Ys + (p * kSbTile) * SD + rb * 32 + (lane & 31) + kh * SD;
        const float* bp = Xs + cb * 32 + (lane & 31) + kh * SD;
#pragma unroll
        for (int k0 = 0; k0 < kSbTile; k0 += 32) {
          float av[16], bv[16];
#pragma unroll
          for (int t = 0; t < 16; ++t) {
            av[t] = ap[(k0 + 2 * t) * SD];
            bv[t] = bp[(k0 + 2 * t) * SD];
          }
#pragma unroll
          for (int t = 0; t < 16; ++t) acc[s] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t], bv[t], acc[s], 0, 0, 0);
        }
      }
    }
    if ((int)threadIdx.x < NP * D) {  // bias: column sums of dY, rows in ascending order
      const int p = threadIdx.x / D, c = threadIdx.x % D;
      float t = 0.f;
      for (int i = 0; i < m; ++i) t += Ys[(p * kSbTile + i) * SD + c];
      bsum += t;
    }
  }
#pragma unroll
  for (int s = 0; s < QPW; ++s) {
    const int q = wave + 4 * s;
    if (q < NQ) {
      const int p = q / (NB * NB), rb = (q % (NB * NB)) % NB, cb = (q % (NB * NB)) / NB;
      float* out = a.gW[p] + (size_t)blockIdx.x * a.part_stride;
      const int k = cb * 32 + (lane & 31);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int o = rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        out[o * D + k] = acc[s][r];
      }
    }
  }
  if ((int)threadIdx.x < NP * D) (a.gb[threadIdx.x / D] + (size_t)blockIdx.x * a.part_stride)[threadIdx.x % D] = bsum;
}

// ---- backward of the post-attention half in ONE row-space kernel -------------------------------------------------------
// In: G = d(layer output).  Out: G = dZ1 = d(ctx + x) (Gb = mask1 * dZ1 with dropout), and this workgroup's partial sums of
// d(ln2 w, b), dW2, db2, dW1, db1, d(ln1 w, b).  Round 2: LayerNorm2 backward, weight gradient, dX product (+ ReLU mask),
// weight gradient, dX product (+ residual), LayerNorm1 backward -- six launches, each streaming [R, D] operands through
// HBM (15 + 21 + 28 + 21 + 28 + 15 us at config 3).  Here a 64-row tile goes through all six steps in LDS; the weight-
// gradient accumulators (2 of the 8 32x32 blocks per wave) live in registers across the tiles, as in sb_wgrad_kernel.
struct SbBlockBwdArgs {
  float* G;
  float* Gb;
  const float *xh2, *rstd2, *h, *y1, *xh1, *rstd1;
  const float *ln2w, *W2, *W1, *ln1w;
  float* part;          // this layer's block of the partial-gradient buffer; slice of workgroup w at + w * part_stride
  size_t part_stride;
  const int32_t* off;
  int B;
  SbDrop dr;            // dr.site = 2 * layer (dropout1); dropout2 = site + 1
  const float* Gin = nullptr;         // one row per sequence: the incoming gradient is read from here (G only receives dZ1) ...
  const int64_t* row_len = nullptr;   // ... and is zero for the rows with row_len <= 0 (empty histories)
};

template <int D>
__global__ __launch_bounds__(kBlock) void sb_block_bwd_kernel(SbBlockBwdArgs a) {
  using Cfg = SasCfg<D>;
  constexpr int SD = D + 1, LPR = D / 4, GPB = kBlock / LPR, NPASS = kSbTile / GPB, NB = D / 32;
  constexpr int NQ = 2 * NB * NB, QPW = (NQ + 3) / 4;   // weight-gradient blocks: (which, rb, cb), which 0 = dW2, 1 = dW1
  extern __shared__ float lds[];
  float* W1s = lds;                       // [D][SD]
  float* W2s = W1s + D * SD;              // [D][SD]
  float* Gs = W2s + D * SD;               // [kSbTile][SD]: mask2 * dZ2, then dY1
  float* Hs = Gs + kSbTile * SD;          // [kSbTile][SD]: h, then dHpre = (Gs . W2) * (h > 0)
  float* Ys = Hs + kSbTile * SD;          // [kSbTile][SD]: y1
  float* Us = Ys + kSbTile * SD;          // [kSbTile][SD]: unmasked dZ2 (dropout only; else == Gs)
  for (int idx = threadIdx.x; idx < D * D; idx += kBlock) {
    W1s[(idx / D) * SD + idx % D] = a.W1[idx];
    W2s[(idx / D) * SD + idx % D] = a.W2[idx];
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l = threadIdx.x % LPR, grp = threadIdx.x / LPR;
  const int R = a.off[a.B];
  const int tiles = (R + kSbTile - 1) / kSbTile;
  const bool drop = a.dr.seed != nullptr;
  const uint64_t seed = drop ? *a.dr.seed : 0;
  SbDrop dr1 = a.dr, dr2 = a.dr;
  dr2.site = a.dr.site + 1u;
  if (!drop) Us = Gs;
  const float4 w2v = reinterpret_cast<const float4*>(a.ln2w)[l], w1v = reinterpret_cast<const float4*>(a.ln1w)[l];

  sas_f32x16 acc[QPW];
#pragma unroll
  for (int q = 0; q < QPW; ++q)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
  float bsum2 = 0.f, bsum1 = 0.f;   // threads < D: db2[t], db1[t]
  float4 aw2 = make_float4(0.f, 0.f, 0.f, 0.f), ab2 = aw2, aw1 = aw2, ab1 = aw2;

  float4 pg[NPASS], pxh[NPASS], ph[NPASS], py[NPASS];
  float prs[NPASS];
  auto fetch = [&](int tile) {
    const int r0 = tile * kSbTile;
#pragma unroll
    for (int q = 0; q < NPASS; ++q) {
      const int r = r0 + q * GPB + grp;
      if (r < R) {
        pg[q] = reinterpret_cast<const float4*>(a.Gin ? a.Gin : a.G)[(size_t)r * LPR + l];
        if (a.row_len && a.row_len[r] <= 0) pg[q] = make_float4(0.f, 0.f, 0.f, 0.f);
        pxh[q] = reinterpret_cast<const float4*>(a.xh2)[(size_t)r * LPR + l];
        ph[q] = reinterpret_cast<const float4*>(a.h)[(size_t)r * LPR + l];
        py[q] = reinterpret_cast<const float4*>(a.y1)[(size_t)r * LPR + l];
        prs[q] = a.rstd2[r];
      }
    }
  };
  if ((int)blockIdx.x < tiles) fetch(blockIdx.x);
  for (int tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    const int r0 = tile * kSbTile;
    const int m = min(kSbTile, R - r0);
    __syncthreads();  // the previous tile's readers are done (and the weights are in place)
    // ---- 1. LayerNorm2 backward, one lane-group per row; stage mask2 * dZ2, h, y1 (rows past m: zeros)
#pragma unroll
    for (int q = 0; q < NPASS; ++q) {
      const int i = q * GPB + grp;
      const int64_t r = (int64_t)r0 + i;
      const float4 z0 = make_float4(0.f, 0.f, 0.f, 0.f);
      float4 dz = z0, dzm = z0, hh = z0, yy = z0;
      const bool on = i < m;
      float4 g = z0, xh = z0;
      if (on) {
        g = pg[q];
        xh = pxh[q];
        hh = ph[q];
        yy = py[q];
        aw2.x = fmaf(g.x, xh.x, aw2.x); aw2.y = fmaf(g.y, xh.y, aw2.y); aw2.z = fmaf(g.z, xh.z, aw2.z); aw2.w = fmaf(g.w, xh.w, aw2.w);
        ab2.x += g.x; ab2.y += g.y; ab2.z += g.z; ab2.w += g.w;
      }
      const float4 dx = make_float4(g.x * w2v.x, g.y * w2v.y, g.z * w2v.z, g.w * w2v.w);
      const float m1 = row_allreduce_sum<LPR>((dx.x + dx.y) + (dx.z + dx.w)) / D;
      const float m2 = row_allreduce_sum<LPR>(fmaf(dx.x, xh.x, fmaf(dx.y, xh.y, fmaf(dx.z, xh.z, dx.w * xh.w)))) / D;
      if (on) {
        const float rs = prs[q];
        dz = make_float4(rs * (dx.x - m1 - xh.x * m2), rs * (dx.y - m1 - xh.y * m2), rs * (dx.z - m1 - xh.z * m2),
                         rs * (dx.w - m1 - xh.w * m2));
        dzm = dz;
        if (drop) {
          const float4 kp = sb_drop_keep4<D>(dr2, seed, r, l);
          dzm = make_float4(dz.x * kp.x, dz.y * kp.y, dz.z * kp.z, dz.w * kp.w);
        }
      }
      float* gd = Gs + i * SD + 4 * l;
      gd[0] = dzm.x; gd[1] = dzm.y; gd[2] = dzm.z; gd[3] = dzm.w;
      if (drop) {
        float* ud = Us + i * SD + 4 * l;
        ud[0] = dz.x; ud[1] = dz.y; ud[2] = dz.z; ud[3] = dz.w;
      }
      float* hd = Hs + i * SD + 4 * l;
      hd[0] = hh.x; hd[1] = hh.y; hd[2] = hh.z; hd[3] = hh.w;
      float* yd = Ys + i * SD + 4 * l;
      yd[0] = yy.x; yd[1] = yy.y; yd[2] = yy.z; yd[3] = yy.w;
    }
    __syncthreads();
    // this tile's LayerNorm1 operands and the next tile's rows travel during the products
    float4 cxh[NPASS];
    float crs[NPASS];
#pragma unroll
    for (int q = 0; q < NPASS; ++q) {
      const int r = r0 + q * GPB + grp;
      if (r < R) {
        cxh[q] = reinterpret_cast<const float4*>(a.xh1)[(size_t)r * LPR + l];
        crs[q] = a.rstd1[r];
      }
    }
    if (tile + (int)gridDim.x < tiles) fetch(tile + gridDim.x);
    // ---- 2. dW2 += (mask2 dZ2)^T . h ;  db2 += column sums
#pragma unroll
    for (int sq = 0; sq < QPW; ++sq) {
      const int q = wave + 4 * sq;
      if (q < NQ && q < NB * NB) {  // wave-uniform
        const int rb = q % NB, cb = q / NB, kh = lane >> 5;
        const float* ap = Gs + rb * 32 + (lane & 31) + kh * SD;
        const float* bp = Hs + cb * 32 + (lane & 31) + kh * SD;
#pragma unroll
        for (int k0 = 0; k0 < kSbTile; k0 += 32) {
          float av[16], bv[16];
#pragma unroll
          for (int t = 0; t < 16; ++t) {
            av[t] = ap[(k0 + 2 * t) * SD];
            bv[t] = bp[(k0 + 2 * t) * SD];
          }
#pragma unroll
          for (int t = 0; t < 16; ++t) acc[sq] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t], bv[t], acc[sq], 0, 0, 0);
        }
      }
    }
    if ((int)threadIdx.x < D) {
      float t = 0.f;
      for (int i = 0; i < m; ++i) t += Gs[i * SD + threadIdx.x];
      bsum2 += t;
    }
    __syncthreads();  // every reader of h is done: step 3 overwrites it
    // ---- 3. dHpre = (mask2 dZ2 . W2) * (h > 0), in place over h
    sas_mm(MatA{Gs, SD, 1}, MatB{W2s, SD, 1}, m, D, D, false, [&](int i, int j, float v) {
      Hs[i * SD + j] = Hs[i * SD + j] > 0.f ? v : 0.f;
    });
    __syncthreads();
    // ---- 4. dW1 += dHpre^T . y1 ;  db1 += column sums
#pragma unroll
    for (int sq = 0; sq < QPW; ++sq) {
      const int q = wave + 4 * sq;
      if (q < NQ && q >= NB * NB) {
        const int qq = q - NB * NB, rb = qq % NB, cb = qq / NB, kh = lane >> 5;
        const float* ap = Hs + rb * 32 + (lane & 31) + kh * SD;
        const float* bp = Ys + cb * 32 + (lane & 31) + kh * SD;
#pragma unroll
        for (int k0 = 0; k0 < kSbTile; k0 += 32) {
          float av[16], bv[16];
#pragma unroll
          for (int t = 0; t < 16; ++t) {
            av[t] = ap[(k0 + 2 * t) * SD];
            bv[t] = bp[(k0 + 2 * t) * SD];
          }
#pragma unroll
          for (int t = 0; t < 16; ++t) acc[sq] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t], bv[t], acc[sq], 0, 0, 0);
        }
      }
    }
    if ((int)threadIdx.x >= D && (int)threadIdx.x < 2 * D) {   // (the second wave: db2's sums are the first wave's)
      float t = 0.f;
      for (int i = 0; i < m; ++i) t += Hs[i * SD + threadIdx.x - D];
      bsum1 += t;
    }
    // ---- 5. dY1 = dHpre . W1 + dZ2 (the residual path), into Gs (step 4 reads Hs / Ys only: no barrier needed before)
    sas_mm(MatA{Hs, SD, 1}, MatB{W1s, SD, 1}, m, D, D, false, [&](int i, int j, float v) { Gs[i * SD + j] = v + Us[i * SD + j]; });
    __syncthreads();
    // ---- 6. LayerNorm1 backward -> dZ1 (global), mask1 * dZ1 for the attention branch
#pragma unroll
    for (int q = 0; q < NPASS; ++q) {
      const int i = q * GPB + grp;
      const int64_t r = (int64_t)r0 + i;
      const bool on = i < m;
      const float* sp = Gs + i * SD + 4 * l;
      const float4 z0 = make_float4(0.f, 0.f, 0.f, 0.f);
      float4 g = z0, xh = z0;
      if (on) {
        g = make_float4(sp[0], sp[1], sp[2], sp[3]);
        xh = cxh[q];
        aw1.x = fmaf(g.x, xh.x, aw1.x); aw1.y = fmaf(g.y, xh.y, aw1.y); aw1.z = fmaf(g.z, xh.z, aw1.z); aw1.w = fmaf(g.w, xh.w, aw1.w);
        ab1.x += g.x; ab1.y += g.y; ab1.z += g.z; ab1.w += g.w;
      }
      const float4 dx = make_float4(g.x * w1v.x, g.y * w1v.y, g.z * w1v.z, g.w * w1v.w);
      const float m1 = row_allreduce_sum<LPR>((dx.x + dx.y) + (dx.z + dx.w)) / D;
      const float m2 = row_allreduce_sum<LPR>(fmaf(dx.x, xh.x, fmaf(dx.y, xh.y, fmaf(dx.z, xh.z, dx.w * xh.w)))) / D;
      if (on) {
        const float rs = crs[q];
        const float4 dz = make_float4(rs * (dx.x - m1 - xh.x * m2), rs * (dx.y - m1 - xh.y * m2), rs * (dx.z - m1 - xh.z * m2),
                                      rs * (dx.w - m1 - xh.w * m2));
        reinterpret_cast<float4*>(a.G)[(size_t)r * LPR + l] = dz;
        if (drop) {
          const float4 kp = sb_drop_keep4<D>(dr1, seed, r, l);
          reinterpret_cast<float4*>(a.Gb)[(size_t)r * LPR + l] = make_float4(dz.x * kp.x, dz.y * kp.y, dz.z * kp.z, dz.w * kp.w);
        }
      }
    }
  }
  // ---- this workgroup's partial sums
  float* out = a.part + (size_t)blockIdx.x * a.part_stride;
#pragma unroll
  for (int sq = 0; sq < QPW; ++sq) {
    const int q = wave + 4 * sq;
    if (q < NQ) {
      const bool second = q >= NB * NB;
      const int qq = second ? q - NB * NB : q, rb = qq % NB, cb = qq / NB;
      float* o = out + (second ? Cfg::oW1 : Cfg::oW2);
      const int k = cb * 32 + (lane & 31);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int oo = rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        o[oo * D + k] = acc[sq][r];
      }
    }
  }
  if ((int)threadIdx.x < D) out[Cfg::ob2 + threadIdx.x] = bsum2;
  else if ((int)threadIdx.x < 2 * D) out[Cfg::ob1 + threadIdx.x - D] = bsum1;
  // LayerNorm weight / bias partials: lane-group accumulators combined over the groups in a fixed order through LDS
  __syncthreads();
  float* red = Gs;  // [4][GPB][SD] floats: 4 * 16 * 65 <= 2 tiles
  {
    const float4 v4[4] = {aw2, ab2, aw1, ab1};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float* d = red + (k * GPB + grp) * SD + 4 * l;
      d[0] = v4[k].x; d[1] = v4[k].y; d[2] = v4[k].z; d[3] = v4[k].w;
    }
  }
  __syncthreads();
  for (int k = threadIdx.x; k < 4 * D; k += kBlock) {
    const int which = k / D, c = k % D;
    float t = 0.f;
    for (int g2 = 0; g2 < GPB; ++g2) t += red[(which * GPB + g2) * SD + c];
    const int o = which == 0 ? Cfg::oln2w : (which == 1 ? Cfg::oln2b : (which == 2 ? Cfg::oln1w : Cfg::oln1b));
    out[o + c] = t;
  }
}

// ---- small row kernels -------------------------------------------------------------------------------

// hv[b] = X[off[b] + len - 1] (0 for an empty history)   (SASRec.py:76)
template <int D>
__global__ __launch_bounds__(kBlock) void sb_last_rows_kernel(const float* __restrict__ X, const int64_t* __restrict__ lengths,
                                                              const int32_t* __restrict__ off, int B, int L,
                                                              float* __restrict__ hv) {
  constexpr int LPR = D / 4;
  const int l = threadIdx.x % LPR;
  for (int b = blockIdx.x * (kBlock / LPR) + threadIdx.x / LPR; b < B; b += gridDim.x * (kBlock / LPR)) {
    const int n = sb_len(lengths, b, L);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (n > 0) v = reinterpret_cast<const float4*>(X)[(size_t)(off[b] + n - 1) * LPR + l];
    reinterpret_cast<float4*>(hv)[(size_t)b * LPR + l] = v;
  }
}

// G[off[b] + i] = dhv[b] on the last row (i = len - 1) of sequence b, 0 on its other rows
template <int D>
__global__ __launch_bounds__(kBlock) void sb_seed_kernel(const float* __restrict__ dhv, const int64_t* __restrict__ lengths,
                                                         const int32_t* __restrict__ off, int B, int L,
                                                         float* __restrict__ G) {
  constexpr int LPR = D / 4;
  const int l = threadIdx.x % LPR;
  const int64_t total = (int64_t)B * L;
  for (int64_t e = (int64_t)blockIdx.x * (kBlock / LPR) + threadIdx.x / LPR; e < total;
       e += (int64_t)gridDim.x * (kBlock / LPR)) {
    const int b = (int)(e / L), i = (int)(e - (int64_t)b * L);
    const int n = sb_len(lengths, b, L);
    if (i >= n) continue;
    reinterpret_cast<float4*>(G)[(size_t)(off[b] + i) * LPR + l] =
        i == n - 1 ? reinterpret_cast<const float4*>(dhv)[(size_t)b * LPR + l] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
}

// g_hist[b, i] = i < len ? G[off[b] + i] : 0   (padded layout the table update consumes)
template <int D>
__global__ __launch_bounds__(kBlock) void sb_unpack_kernel(const float* __restrict__ G, const int64_t* __restrict__ lengths,
                                                           const int32_t* __restrict__ off, int B, int L,
                                                           float* __restrict__ g_hist) {
  constexpr int LPR = D / 4;
  const int l = threadIdx.x % LPR;
  const int64_t total = (int64_t)B * L;
  for (int64_t e = (int64_t)blockIdx.x * (kBlock / LPR) + threadIdx.x / LPR; e < total;
       e += (int64_t)gridDim.x * (kBlock / LPR)) {
    const int b = (int)(e / L), i = (int)(e - (int64_t)b * L);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < sb_len(lengths, b, L)) v = reinterpret_cast<const float4*>(G)[(size_t)(off[b] + i) * LPR + l];
    reinterpret_cast<float4*>(g_hist)[e * LPR + l] = v;
  }
}

// ---- position-table gradient --------------------------------------------------------------------------------
// dP[p] = sum over sequences b with len_b >= p of g_hist[b, len_b - p]  (position id = len - index, SASRec.py:64;
// id 0 = padding, whose slots carry zero gradient).  The key range is tiny (history_max + 1 rows), each row
// collects up to B occurrences: instead of the generic sort + segmented sum, one workgroup per (position, chunk of
// kPosChunk sequences) adds its rows in fixed order, a second pass adds the chunks.
constexpr int kPosChunk = 256;   // (1024 before: 4 x 51 workgroups at B = 4096 whose lane-groups walked 64 sequences one dependent
                                 //  load pair at a time -- 41 us for 52 MB; now 16 x 51 workgroups, eight sequences in flight per group)

template <int D>
__global__ __launch_bounds__(kBlock) void sb_pos_grad_kernel(const float* __restrict__ g_hist, const int64_t* __restrict__ lengths,
                                                             int B, int L, float* __restrict__ part /*[chunks][L+1][D]*/) {
  constexpr int LPR = D / 4, GPB = kBlock / LPR;
  __shared__ float s_red[GPB][D + 4];
  const int p = blockIdx.x, chunk = blockIdx.y;
  const int l = threadIdx.x % LPR, grp = threadIdx.x / LPR;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  const int b_end = min(B, (chunk + 1) * kPosChunk);
  if (p >= 1) {
    constexpr int U = 8;
    for (int b0 = chunk * kPosChunk + grp; b0 < b_end; b0 += U * GPB) {
      int n[U];
      float4 v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) n[u] = b0 + u * GPB < b_end ? sb_len(lengths, b0 + u * GPB, L) : 0;
#pragma unroll
      for (int u = 0; u < U; ++u)
        v[u] = n[u] >= p ? reinterpret_cast<const float4*>(g_hist)[((size_t)(b0 + u * GPB) * L + (n[u] - p)) * LPR + l]
                         : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int u = 0; u < U; ++u) {   // (ascending b, as before)
        acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w;
      }
    }
  }
  float* sr = &s_red[grp][4 * l];
  sr[0] = acc.x; sr[1] = acc.y; sr[2] = acc.z; sr[3] = acc.w;
  __syncthreads();
  for (int k = threadIdx.x; k < D; k += kBlock) {
    float t = 0.f;
    for (int g = 0; g < GPB; ++g) t += s_red[g][k];
    part[((size_t)chunk * (L + 1) + p) * D + k] = t;
  }
}

// out[i] = sum of the chunks for i < count, 0 for count <= i < total (the table's rows past history_max: the gradient buffer needs
// no separate fill)
__global__ __launch_bounds__(kBlock) void sb_pos_reduce_kernel(const float* __restrict__ part, int chunks, int count, int total,
                                                               float* __restrict__ out) {
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < total; i += gridDim.x * kBlock) {
    float t = 0.f;
    if (i < count)
      for (int c = 0; c < chunks; ++c) t += part[(size_t)c * count + i];
    out[i] = t;
  }
}

// ---- host side: buffer layout and launch sequences ---------------------------------------------------------

// ---- the LAST layer needs one row per sequence -----------------------------------------------------------------------------
// Only position len - 1 of the last block's output is consumed (models/sequential/SASRec.py:76) and the mask is causal, so
// the last block needs ONE query row per sequence: its keys / values for all rows (two of the three projections), the last
// row's query, one row of attention per head, and LayerNorm - FFN - LayerNorm on B rows instead of sum(len).  The backward
// mirrors it: dK, dV for all rows (rank-one per head), dQ and the block's backward on B rows.  With --num_layers 1 (the
// reference's default) that is the whole encoder.  Same arithmetic as the full path for that row; results differ only in
// summation order.  (Off with dropout: the masks are keyed by the row index of the full row space.)
struct SbLastAttnArgs {
  const float* q;      // [B, D] query of each sequence's last row
  const float *k, *v;  // [R, D]
  float* ctx;          // fwd out [B, D]
  const float* dctx;   // bwd in [B, D]
  float* dq;           // bwd out [B, D]
  float *dk, *dv;      // bwd out [R, D]
  const int64_t* lengths;
  const int32_t* off;
  int B, L, n_heads;
};

// one wave per (sequence, head); lane j = key j (history_max <= 64)
template <int D, int DK, bool BWD>
__global__ __launch_bounds__(kBlock) void sb_attn_last_kernel(SbLastAttnArgs a) {
  constexpr int NV = DK / 4;
  const int lane = threadIdx.x & 63;
  const int item = (int)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
  if (item >= a.B * a.n_heads) return;   // wave-uniform
  const int b = item / a.n_heads, hh = item % a.n_heads, hc = hh * DK;
  const int n = sb_len(a.lengths, b, a.L);
  if (n == 0) {   // an empty history: defined zeros for the kernels that follow (its output row is masked out at the end)
    if (lane < DK) (BWD ? a.dq : a.ctx)[(size_t)b * D + hc + lane] = 0.f;
    return;
  }
  const float sqrt_dk = sqrtf((float)DK);
  const bool on = lane < n;
  const size_t row = (size_t)a.off[b] + (on ? lane : 0);
  float qv[DK], kv[DK], vv[DK];
#pragma unroll
  for (int c = 0; c < NV; ++c) {
    const float4 q4 = reinterpret_cast<const float4*>(a.q + (size_t)b * D + hc)[c];
    const float4 k4 = reinterpret_cast<const float4*>(a.k + row * D + hc)[c];
    const float4 v4 = reinterpret_cast<const float4*>(a.v + row * D + hc)[c];
    qv[4 * c] = q4.x; qv[4 * c + 1] = q4.y; qv[4 * c + 2] = q4.z; qv[4 * c + 3] = q4.w;
    kv[4 * c] = k4.x; kv[4 * c + 1] = k4.y; kv[4 * c + 2] = k4.z; kv[4 * c + 3] = k4.w;
    vv[4 * c] = v4.x; vv[4 * c + 1] = v4.y; vv[4 * c + 2] = v4.z; vv[4 * c + 3] = v4.w;
  }
  float sc = 0.f;
#pragma unroll
  for (int c = 0; c < DK; ++c) sc = fmaf(qv[c], kv[c], sc);
  sc = sas_div_scale(sc, sqrt_dk);
  const float m = wave_allreduce_max(on ? sc : -INFINITY);
  const float e = on ? expf(sc - m) : 0.f;
  const float rz = 1.0f / wave_allreduce_sum(e);
  const float p = e * rz;
  if (!BWD) {
    float mine = 0.f;
#pragma unroll
    for (int c = 0; c < DK; ++c) {
      const float t = wave_allreduce_sum(p * vv[c]);
      if (lane == c) mine = t;
    }
    if (lane < DK) a.ctx[(size_t)b * D + hc + lane] = mine;
    return;
  }
  float gv[DK];
#pragma unroll
  for (int c = 0; c < NV; ++c) {
    const float4 g4 = reinterpret_cast<const float4*>(a.dctx + (size_t)b * D + hc)[c];
    gv[4 * c] = g4.x; gv[4 * c + 1] = g4.y; gv[4 * c + 2] = g4.z; gv[4 * c + 3] = g4.w;
  }
  float dp = 0.f;
#pragma unroll
  for (int c = 0; c < DK; ++c) dp = fmaf(gv[c], vv[c], dp);
  const float dot = wave_allreduce_sum(p * dp);
  const float ds = sas_div_scale(p * (dp - dot), sqrt_dk);
  if (on) {
#pragma unroll
    for (int c = 0; c < NV; ++c) {
      reinterpret_cast<float4*>(a.dk + row * D + hc)[c] = make_float4(ds * qv[4 * c], ds * qv[4 * c + 1], ds * qv[4 * c + 2], ds * qv[4 * c + 3]);
      reinterpret_cast<float4*>(a.dv + row * D + hc)[c] = make_float4(p * gv[4 * c], p * gv[4 * c + 1], p * gv[4 * c + 2], p * gv[4 * c + 3]);
    }
  }
  float mine = 0.f;
#pragma unroll
  for (int c = 0; c < DK; ++c) {
    const float t = wave_allreduce_sum(ds * kv[c]);
    if (lane == c) mine = t;
  }
  if (lane < DK) a.dq[(size_t)b * D + hc + lane] = mine;
}

// out[b] = len[b] > 0 ? in[b] : 0  (in may be out)
template <int D>
__global__ __launch_bounds__(kBlock) void sb_mask_empty_kernel(const float* __restrict__ in, const int64_t* __restrict__ lengths, int B,
                                                               float* __restrict__ out) {
  constexpr int LPR = D / 4;
  const int l = threadIdx.x % LPR;
  for (int b = blockIdx.x * (kBlock / LPR) + threadIdx.x / LPR; b < B; b += gridDim.x * (kBlock / LPR)) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (lengths[b] > 0) v = reinterpret_cast<const float4*>(in)[(size_t)b * LPR + l];
    reinterpret_cast<float4*>(out)[(size_t)b * LPR + l] = v;
  }
}

// G[off[b] + len - 1] += T[b] for every non-empty sequence (each row of G is touched by one lane-group: no atomics);
// off == nullptr: G is the padded [B, L] layout
template <int D>
__global__ __launch_bounds__(kBlock) void sb_last_add_kernel(const float* __restrict__ T, const int64_t* __restrict__ lengths,
                                                             const int32_t* __restrict__ off, int B, int L, float* __restrict__ G) {
  constexpr int LPR = D / 4;
  const int l = threadIdx.x % LPR;
  for (int b = blockIdx.x * (kBlock / LPR) + threadIdx.x / LPR; b < B; b += gridDim.x * (kBlock / LPR)) {
    const int n = sb_len(lengths, b, L);
    if (n == 0) continue;
    float4* g = reinterpret_cast<float4*>(G) + ((off ? (size_t)off[b] : (size_t)b * L) + n - 1) * LPR + l;
    const float4 t = reinterpret_cast<const float4*>(T)[(size_t)b * LPR + l];
    float4 x = *g;
    x.x += t.x; x.y += t.y; x.z += t.z; x.w += t.w;
    *g = x;
  }
}

static bool sb_fused_block();
// RC_SAS_LAST_ROW: 0 = the last layer on all rows like the others; 1 = one query row per sequence over materialised keys /
// values (sb_attn_last_kernel); unset / 2 = the same without keys and values (sas_last_row.hpp).  (A/B timing, equivalence tests)
static int sb_last_row_request() {
  const char* v = getenv("RC_SAS_LAST_ROW");
  return v ? atoi(v) : 2;
}
// -> 0: all rows, 1: one query row over K / V, 2: one query row, no K / V
template <int D>
static int sb_last_row_mode(int n_heads, int B, int L, bool drop) {
  // (small row spaces are launch-bound: the first version's six extra launches cost more than its rows save -- B = 256, L = 50:
  //  0.281 against 0.266 ms per step; RC_SAS_LAST_ROW_MIN overrides its threshold on B * history_max.  The K / V-free version
  //  has FEWER launches than the all-rows path (4 + 7 against 7 + 8 with one block) and is taken at every size: B = 256, L = 50
  //  0.249 -> 0.190 ms per replayed step)
  static const int64_t min_rows = [] {
    const char* v = getenv("RC_SAS_LAST_ROW_MIN");
    return v ? (int64_t)atoll(v) : (int64_t)32768;
  }();
  const int want = sb_last_row_request();
  if (want <= 0 || drop || n_heads < 1 || D % n_heads != 0 || L > kLrMaxL || !sb_fused_block() || !sb_rows16()) return 0;
  const int dk = D / n_heads;
  // the buffers of the K / V-free version live in the saved state's per-layer arrays: H * D + 4 <= L * D and 2 D + H L <= L D
  const bool v2 = (n_heads == 1 || n_heads == 2 || n_heads == 4) && L >= 3 && L >= n_heads + 1 && dk % (D * D / kBlock) == 0;
  const bool v1 = L >= 2 && L <= kSasLP && (dk == 16 || dk == 32 || dk == 64) && (int64_t)B * L >= min_rows;
  if (want >= 2 && v2) return 2;
  return v1 ? 1 : 0;
}

static int sb_fill_layers(SasLayer* layer, const float* const* layer_params, int n_layers) {
  RC_REQUIRE(n_layers >= 1 && n_layers <= kSasMaxLayers, "SASRec: num_layers must be in [1, %d]", kSasMaxLayers);
  RC_REQUIRE(layer_params != nullptr, "SASRec: layer parameter table missing");
  for (int l = 0; l < n_layers; ++l) {
    const float* const* q = layer_params + 14 * l;
    for (int k = 0; k < 14; ++k) RC_REQUIRE(q[k] != nullptr, "SASRec: layer %d parameter %d is null", l, k);
    SasLayer& s = layer[l];
    memset(&s, 0, sizeof(s));
    s.Wq = q[0]; s.bq = q[1]; s.Wk = q[2]; s.bk = q[3]; s.Wv = q[4]; s.bv = q[5]; s.ln1w = q[6]; s.ln1b = q[7];
    s.W1 = q[8]; s.b1 = q[9]; s.W2 = q[10]; s.b2 = q[11]; s.ln2w = q[12]; s.ln2b = q[13];
  }
  return RC_OK;
}

// saved activations of one layer, [Rmax, D] each (+ two [Rmax] vectors); Rmax = B * L
struct SbSaved {
  float *x, *q, *k, *v, *xh1, *y1, *h, *xh2, *rstd1, *rstd2;
};
static size_t sb_layer_floats(size_t rmax, int d) { return 8 * rmax * d + 2 * rmax; }
static SbSaved sb_saved(float* state, int l, size_t rmax, int d) {
  float* p = state + (size_t)l * sb_layer_floats(rmax, d);
  SbSaved s;
  s.x = p; s.q = s.x + rmax * d; s.k = s.q + rmax * d; s.v = s.k + rmax * d; s.xh1 = s.v + rmax * d;
  s.y1 = s.xh1 + rmax * d; s.h = s.y1 + rmax * d; s.xh2 = s.h + rmax * d; s.rstd1 = s.xh2 + rmax * d;
  s.rstd2 = s.rstd1 + rmax;
  return s;
}

constexpr int kSbPartWg = 512;  // workgroups that own a partial-gradient slice

// the row-space offsets and the length buckets are computed by the forward pass and kept in the tail of the saved
// state: the backward pass reads them back instead of recomputing them
static size_t sb_state_act_floats(size_t rmax, int d, int n_layers) { return (size_t)n_layers * sb_layer_floats(rmax, d) + rmax * d; }
static size_t sb_state_int_floats(int B) { return 6 * (size_t)B + 80; }

struct SbWs {
  int32_t *off, *bucket;  // live in the state buffer
  int32_t* off_seq;       // [B + 1], only off_seq[B] = B is defined: the row space of one row per sequence
  float *t0, *t1, *t2, *t3, *t4;  // [Rmax, D] scratch (t4: the masked branch gradient when dropout is on)
  float* part;               // [kSbPartWg][n_layers * PL]
  size_t total;
};
static SbWs sb_carve(void* base, float* state, int B, int L, int d, int n_layers, bool bwd) {
  Carver cv(base);
  SbWs w;
  const size_t rmax = (size_t)B * L;
  w.off = state ? reinterpret_cast<int32_t*>(state + sb_state_act_floats(rmax, d, n_layers)) : nullptr;
  w.bucket = state ? w.off + (size_t)B + 4 : nullptr;
  w.off_seq = state ? w.bucket + 4 * (size_t)B + 68 : nullptr;
  w.t0 = cv.take<float>(rmax * d);
  w.t1 = bwd ? cv.take<float>(rmax * d) : nullptr;
  w.t2 = bwd ? cv.take<float>(rmax * d) : nullptr;
  w.t3 = bwd ? cv.take<float>(rmax * d) : nullptr;
  w.t4 = bwd ? cv.take<float>(rmax * d) : nullptr;
  w.part = bwd ? cv.take<float>((size_t)kSbPartWg * n_layers * (5 * (size_t)d * d + 9 * d)) : nullptr;
  w.total = cv.off;
  return w;
}

static int sb_row_grid(int64_t rows, int lpr) {
  int64_t g = (rows + (kBlock / lpr) - 1) / (kBlock / lpr);
  if (g > 256 * 8) g = 256 * 8;
  return (int)(g < 1 ? 1 : g);
}

template <int D, int NW, bool TRANS>
static int sb_linear(const SbLinArgs& a, int64_t rmax, hipStream_t s) {
  const size_t lds = ((size_t)(NW * D + kSbTile) * (D + 1) + (size_t)NW * D) * sizeof(float);
  auto kern = sb_linear_kernel<D, NW, TRANS>;
  RC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  int64_t tiles = (rmax + kSbTile - 1) / kSbTile;
  const int per_cu = (int)((160 * 1024) / lds);
  const int64_t cap = 256 * (int64_t)(per_cu < 1 ? 1 : (per_cu > 4 ? 4 : per_cu));
  if (tiles > cap) tiles = cap;
  hipLaunchKernelGGL(kern, dim3((unsigned)(tiles < 1 ? 1 : tiles)), dim3(kBlock), lds, s, a);
  RC_LAUNCH_CHECK();
  return RC_OK;
}

// attention launches.  Sequences are split on the device into length classes <= 16, <= 32, <= history_max
// (sas_bucket_n_kernel); a class runs with as many LDS rows as its longest member, so the short histories that
// dominate real data get many resident workgroups per CU (17 KB forward, 25 KB backward at 16 rows, d = 64).
// Up to 4 heads run one wave per head (sb_attn_*_wave_kernel) when the per-head scratch fits the LDS; otherwise
// the four waves of the workgroup share each head's blocks.
static SasBuckets sb_buckets(int L) {
  SasBuckets bk;
  bk.n = 0;
  for (int t : {16, 32})
    if (t < L) bk.thr[bk.n++] = t;
  bk.thr[bk.n++] = L;
  return bk;
}

// RC_SAS_REG_ATTN=0: the LDS-tile attention kernels (rounds 1-3) for every shape, for A/B timing and the equivalence test
static bool sb_reg_attention() {
  const char* v = getenv("RC_SAS_REG_ATTN");
  return !(v && v[0] == '0');
}

template <int D, bool BWD>
static int sb_attention(SbAttnArgs a, int32_t* bucket, bool make_buckets, hipStream_t s) {
  if (sb_reg_attention() && sas_reg_attn_fits(D, a.n_heads, a.L)) {   // every (sequence, head) by one wave, operands in registers
    const int dk = D / a.n_heads;
    void (*kern)(SbAttnArgs) = nullptr;
    if (dk == 16) kern = sb_attn_reg_kernel<D, 16, BWD>;
    else if (dk == 32) kern = sb_attn_reg_kernel<D, 32, BWD>;
    else if constexpr (D >= 64) kern = sb_attn_reg_kernel<D, 64, BWD>;
    if (kern != nullptr) {
      a.seq_list = nullptr;
      a.seq_count = nullptr;
      const int64_t items = (int64_t)a.B * a.n_heads;
      hipLaunchKernelGGL(kern, dim3((unsigned)((items + 3) / 4)), dim3(256), 0, s, a);
      RC_LAUNCH_CHECK();
      return RC_OK;
    }
  }
  const SasBuckets bk = sb_buckets(a.L);
  int32_t* count = bucket + 4 * (size_t)a.B;
  if (make_buckets) {
    hipLaunchKernelGGL(sas_bucket_n_kernel, dim3(1), dim3(kBlock), 0, s, a.lengths, a.B, bk, bucket, count);
    RC_LAUNCH_CHECK();
  }
  for (int k = 0; k < bk.n; ++k) {
    a.lp = (bk.thr[k] + 1) / 2 * 2;  // even: odd LDS row strides
    a.seq_list = bucket + (size_t)k * a.B;
    a.seq_count = count + k;
    const size_t buf = (size_t)sb_buf_floats(D, a.lp) * sizeof(float);
    const size_t head = (size_t)a.lp * (a.lp + 1) * sizeof(float);
    const size_t lds_wave = (BWD ? 4 : 3) * buf + (BWD ? 2 : 1) * a.n_heads * head;
    const bool per_wave = a.n_heads <= 4 && lds_wave <= 160 * 1024;
    const bool block_ok = a.lp == 32 || a.lp == 64;  // the shared-block kernels split 32 / 64 rows over 4 waves
    if (!per_wave && !block_ok) a.lp = a.lp <= 32 ? 32 : 64;
    const size_t buf2 = (size_t)sb_buf_floats(D, a.lp) * sizeof(float);
    const size_t lds = per_wave ? lds_wave : (size_t)(BWD ? 7 : 5) * buf2;
    const int per_cu = (int)((160 * 1024) / lds);
    // two waves per head where the LDS footprint leaves the CU short of waves (the long length class)
    const int wph = per_wave && a.lp > 32 && per_cu <= 2 ? 2 : 1;
    // prefetch of the next sequence's rows: needs the full 64 * 4 * wph threads; PF float4 per thread and buffer
    const int threads_w = 64 * a.n_heads * wph;
    int pfn = 0;
    if (per_wave && a.n_heads == 4 && getenv("RC_SAS_NO_PREFETCH") == nullptr) {
      pfn = (a.lp * (D / 4) + threads_w - 1) / threads_w;
      pfn = pfn <= 1 ? 1 : (pfn <= 2 ? 2 : (pfn <= 4 ? 4 : 0));
    }
    void (*kern)(SbAttnArgs) = nullptr;
    if (!per_wave) kern = BWD ? sb_attn_bwd_kernel<D> : sb_attn_fwd_kernel<D>;
#define RC_SB_PICK(W_, P_) kern = BWD ? sb_attn_bwd_wave_kernel<D, W_, P_> : sb_attn_fwd_wave_kernel<D, W_, P_>
    else if (wph == 2) { if (pfn == 1) RC_SB_PICK(2, 1); else if (pfn == 2) RC_SB_PICK(2, 2); else if (pfn == 4) RC_SB_PICK(2, 4); else RC_SB_PICK(2, 0); }
    else { if (pfn == 1) RC_SB_PICK(1, 1); else if (pfn == 2) RC_SB_PICK(1, 2); else if (pfn == 4) RC_SB_PICK(1, 4); else RC_SB_PICK(1, 0); }
#undef RC_SB_PICK
    RC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int grid = 256 * (per_cu < 1 ? 1 : (per_cu > 8 ? 8 : per_cu));
    if (grid > a.B) grid = a.B;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(per_wave ? 64 * a.n_heads * wph : kBlock), lds, s, a);
    RC_LAUNCH_CHECK();
  }
  return RC_OK;
}

// RC_SAS_FUSED_BLOCK=0: the post-attention half as four launches (round 2), for A/B timing and the equivalence test
static bool sb_fused_block() {
  const char* v = getenv("RC_SAS_FUSED_BLOCK");
  return !(v && v[0] == '0');
}


// ---- the K / V-free last block (sas_last_row.hpp): launch sequences ----------------------------------------------------------------------
struct SbLastBufs {   // where its per-sequence arrays live inside the layer's saved state
  float *q, *xl, *p, *qt, *cq, *xbar;
};
static SbLastBufs sb_last_bufs(const SbSaved& sv, int B, int L, int d, int n_heads) {
  SbLastBufs u;
  u.q = sv.q; u.xl = sv.q + (size_t)B * d; u.p = sv.q + 2 * (size_t)B * d;          // [B, D], [B, D], [B, H, L]
  u.qt = sv.k; u.cq = sv.k + (size_t)B * n_heads * d;                                // [B, H, D], [B, 4]
  u.xbar = sv.v;                                                                      // [B, H, D]
  return u;
}
// partial-gradient slices the last block's backward writes when it runs on one row per sequence: every kernel of that path
// launches exactly this many workgroups, so with one block no slice is left unwritten (no zero fill, a short reduction)
static int sb_last_slots(int B) {
  const int64_t tiles = ((int64_t)B + kSbTile - 1) / kSbTile;
  return (int)(tiles < 1 ? 1 : (tiles > kSbPartWg ? kSbPartWg : tiles));
}
// ... except the key / value projections' gradients: sb_lr_wgrad_kernel sums 16 sequences per workgroup (64 would leave most
// of the chip idle: 84 us at config 3), the reduction reads that index range with its own slice count
static int sb_last_kv_slots(int B) {
  const int64_t g = ((int64_t)B + kLrWgradSeqs - 1) / kLrWgradSeqs;
  return (int)(g < 1 ? 1 : (g > kSbPartWg ? kSbPartWg : g));
}
static int sb_lr_grid8(int B) {   // workgroups of 8 waves, one sequence per wave and pass; the weights are staged once per workgroup
  const int g = (B + 7) / 8;
  return g < 1 ? 1 : (g > 1024 ? 1024 : g);
}

template <int D>
static int sb_last_block_fwd(const float* item_emb, const float* pos_emb, const SasLayer& p, int n_heads, const int64_t* hist,
                             const int64_t* lengths, int B, int L, bool gather, const SbSaved& sv, float* hv, const SbWs& w,
                             SbDrop dr, hipStream_t s) {
  const SbLastBufs u = sb_last_bufs(sv, B, L, D, n_heads);
  SbLrRows rows;
  memset(&rows, 0, sizeof(rows));
  rows.lengths = lengths; rows.B = B; rows.L = L;
  if (gather) { rows.item_emb = item_emb; rows.pos_emb = pos_emb; rows.hist = hist; }
  else { rows.X = sv.x; rows.off = w.off; }
  {   // x_last, q = Wq x_last + bq, qt_h = Wk_h^T q_h, cq_h = q_h . bk_h
    SbLrHeadTArgs a;
    memset(&a, 0, sizeof(a));
    a.rows = rows; a.Wq = p.Wq; a.bq = p.bq; a.W = p.Wk; a.bias = p.bk; a.xl = u.xl; a.q = u.q; a.outT = u.qt; a.cs = u.cq;
    a.off_seq = w.off_seq; a.H = n_heads;
    const size_t lds = ((size_t)2 * D * (D + 4) + (kLrBlock / 64) * 2 * D) * sizeof(float);
    if (gather) hipLaunchKernelGGL((sb_lr_headT_kernel<D, 2>), dim3(sb_lr_grid8(B)), dim3(kLrBlock), lds, s, a);
    else hipLaunchKernelGGL((sb_lr_headT_kernel<D, 1>), dim3(sb_lr_grid8(B)), dim3(kLrBlock), lds, s, a);
    RC_LAUNCH_CHECK();
  }
  {   // probabilities and xbar_h = sum_j p_hj x_j: the one pass over the rows
    SbLrAttnArgs a;
    memset(&a, 0, sizeof(a));
    a.rows = rows; a.Xsave = gather ? sv.x : nullptr; a.qt = u.qt; a.cq = u.cq; a.p = u.p; a.xbar = u.xbar;
    const dim3 grid((unsigned)(B < 2048 ? (B < 1 ? 1 : B) : 2048)), block(kBlock);   // 19.5 KB of LDS per workgroup: eight per CU
#define RC_LR_FWD(NH)                                                                                        \
  do {                                                                                                       \
    if (L > 64) {                                                                                            \
      if (gather) hipLaunchKernelGGL((sb_lr_attn_fwd_long_kernel<D, NH, true, 2>), grid, block, 0, s, a);    \
      else hipLaunchKernelGGL((sb_lr_attn_fwd_long_kernel<D, NH, false, 2>), grid, block, 0, s, a);          \
    } else {                                                                                                 \
      if (gather) hipLaunchKernelGGL((sb_lr_attn_fwd_kernel<D, NH, true>), grid, block, 0, s, a);            \
      else hipLaunchKernelGGL((sb_lr_attn_fwd_kernel<D, NH, false>), grid, block, 0, s, a);                  \
    }                                                                                                        \
  } while (0)
    if (n_heads == 1) RC_LR_FWD(1);
    else if (n_heads == 2) RC_LR_FWD(2);
    else RC_LR_FWD(4);
#undef RC_LR_FWD
    RC_LAUNCH_CHECK();
  }
  SbBlockArgs bk;
  bk.ctx = nullptr; bk.x = u.xl; bk.ln1w = p.ln1w; bk.ln1b = p.ln1b; bk.W1 = p.W1; bk.b1 = p.b1; bk.W2 = p.W2; bk.b2 = p.b2;
  bk.ln2w = p.ln2w; bk.ln2b = p.ln2b; bk.xh1 = sv.xh1; bk.rstd1 = sv.rstd1; bk.y1 = sv.y1; bk.h = sv.h; bk.xh2 = sv.xh2;
  bk.rstd2 = sv.rstd2; bk.xnext = hv; bk.off = w.off_seq; bk.B = B; bk.dr = dr;
  bk.row_len = lengths;   // an empty history's output row is zero (SASRec.py:76 reads his[b, -1] of an all-padding row)
  bk.xbar = u.xbar; bk.Wv = p.Wv; bk.bv = p.bv; bk.H = n_heads;   // ctx_h = Wv_h xbar_h + bv_h inside the block kernel
  const size_t lds16 = ((size_t)3 * D * (D + 4) + 7 * (size_t)D) * sizeof(float);
  auto kern16 = sb_block16_fwd_kernel<D>;
  RC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern16), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds16));
  int64_t gr = (((int64_t)B + 15) / 16 + 3) / 4;
  if (gr > 768) gr = 768;
  hipLaunchKernelGGL(kern16, dim3((unsigned)(gr < 1 ? 1 : gr)), dim3(256), lds16, s, bk);
  RC_LAUNCH_CHECK();
  return RC_OK;
}

template <int D, int NP>
static int sb_wgrad(const SbWgradArgs& a, int64_t rmax, hipStream_t s);

// Gout: the compact gradient rows of the block's input (padded == false), or the caller's padded [B, L, D] array (one block)
template <int D>
static int sb_last_block_bwd(const SasLayer& p, int n_heads, const int64_t* lengths, int B, int L, bool padded, const SbSaved& sv,
                             const float* dhv, float* Gout, float* gp, size_t stride, const SbWs& w, SbDrop dr, hipStream_t s, int part = 0) {
  // part (rc_sasrec_batch_bwd_part): 0 everything; 1 the four launches that end with the input gradient rows complete (Gout); 2 the
  // two parameter-gradient launches behind them, which nothing downstream of Gout waits for
  using Cfg = SasCfg<D>;
  const SbLastBufs u = sb_last_bufs(sv, B, L, D, n_heads);
  float* gl = w.t1;                                  // [B, D]: dhv (empty histories: 0), then dZ1 = d ctx
  float* dql = w.t1 + (size_t)B * D;                 // [B, D]
  float* gt = w.t2;                                  // [B, H, D]
  float* cg = w.t2 + (size_t)B * n_heads * D;        // [B, 4]
  float* ybar = w.t3;                                // [B, H, D]
  float* sds = w.t3 + (size_t)B * n_heads * D;       // [B, 4]
  if (part != 2) {
  {
    SbBlockBwdArgs bb;
    bb.G = gl; bb.Gb = gl; bb.xh2 = sv.xh2; bb.rstd2 = sv.rstd2; bb.h = sv.h; bb.y1 = sv.y1; bb.xh1 = sv.xh1; bb.rstd1 = sv.rstd1;
    bb.ln2w = p.ln2w; bb.W2 = p.W2; bb.W1 = p.W1; bb.ln1w = p.ln1w; bb.part = gp; bb.part_stride = stride; bb.off = w.off_seq; bb.B = B;
    bb.dr = dr;
    bb.Gin = dhv; bb.row_len = lengths;   // d hv of an empty history is not propagated (its output row was zeroed)
    const size_t lds = (size_t)(2 * D + 3 * kSbTile) * (D + 1) * sizeof(float);
    auto kern = sb_block_bwd_kernel<D>;
    RC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int64_t tiles = ((int64_t)B + kSbTile - 1) / kSbTile;
    if (tiles > kSbPartWg) tiles = kSbPartWg;
    hipLaunchKernelGGL(kern, dim3((unsigned)(tiles < 1 ? 1 : tiles)), dim3(kBlock), lds, s, bb);
    RC_LAUNCH_CHECK();
  }
  {   // gt_h = Wv_h^T g_h, cg_h = g_h . bv_h
    SbLrHeadTArgs a;
    memset(&a, 0, sizeof(a));
    a.rows.B = B; a.rows.L = L; a.rows.lengths = lengths;
    a.in = gl; a.W = p.Wv; a.bias = p.bv; a.outT = gt; a.cs = cg; a.H = n_heads;
    const size_t lds = ((size_t)D * (D + 4) + (kLrBlock / 64) * 2 * D) * sizeof(float);
    hipLaunchKernelGGL((sb_lr_headT_kernel<D, 0>), dim3(sb_lr_grid8(B)), dim3(kLrBlock), lds, s, a);
    RC_LAUNCH_CHECK();
  }
  {   // dX rows, ybar_h = sum_j ds_hj x_j, sum_j ds_hj
    SbLrAttnArgs a;
    memset(&a, 0, sizeof(a));
    a.rows.X = sv.x; a.rows.off = padded ? nullptr : w.off; a.rows.lengths = lengths; a.rows.B = B; a.rows.L = L;
    a.qt = u.qt; a.cq = u.cq; a.p = u.p; a.gt = gt; a.cg = cg; a.G = Gout; a.g_off = padded ? nullptr : w.off; a.ybar = ybar;
    a.sds = sds;
    const dim3 grid((unsigned)(B < 1792 ? (B < 1 ? 1 : B) : 1792)), block(kBlock);   // 21.6 KB of LDS per workgroup: seven per CU
#define RC_LR_BWD(NH)                                                                                  \
  do {                                                                                                 \
    if (L > 64) hipLaunchKernelGGL((sb_lr_attn_bwd_long_kernel<D, NH, 2>), grid, block, 0, s, a);      \
    else hipLaunchKernelGGL((sb_lr_attn_bwd_kernel<D, NH>), grid, block, 0, s, a);                     \
  } while (0)
    if (n_heads == 1) RC_LR_BWD(1);
    else if (n_heads == 2) RC_LR_BWD(2);
    else RC_LR_BWD(4);
#undef RC_LR_BWD
    RC_LAUNCH_CHECK();
  }
  {   // dq_h = Wk_h ybar_h + bk_h sum_j ds_hj;  d x_last = dq Wq + dZ1, added to the last row's dX
    SbLrTailArgs a;
    memset(&a, 0, sizeof(a));
    a.ybar = ybar; a.sds = sds; a.gl = gl; a.Wk = p.Wk; a.bk = p.bk; a.Wq = p.Wq; a.lengths = lengths;
    a.g_off = padded ? nullptr : w.off; a.dq = dql; a.G = Gout; a.B = B; a.L = L; a.H = n_heads;
    const size_t lds = ((size_t)2 * D * (D + 4) + (kLrBlock / 64) * (kLrMaxHeads * (D + 4) + D)) * sizeof(float);
    hipLaunchKernelGGL((sb_lr_tail_kernel<D>), dim3(sb_lr_grid8(B)), dim3(kLrBlock), lds, s, a);
    RC_LAUNCH_CHECK();
  }
  }
  if (part == 1) return RC_OK;
  {   // dWk, dbk, dWv, dbv from the per-sequence sums
    SbLrWgradArgs a;
    memset(&a, 0, sizeof(a));
    a.q = u.q; a.ybar = ybar; a.sds = sds; a.g = gl; a.xbar = u.xbar;
    a.gWk = gp + Cfg::oWk; a.gbk = gp + Cfg::obk; a.gWv = gp + Cfg::oWv; a.gbv = gp + Cfg::obv; a.part_stride = stride;
    a.B = B; a.H = n_heads;
    const int grid = sb_last_kv_slots(B);
    a.chunk = (B + grid - 1) / grid;   // (workgroups past the last sequence write zeros)
    hipLaunchKernelGGL((sb_lr_wgrad_kernel<D>), dim3((unsigned)grid), dim3(kBlock), 0, s, a);
    RC_LAUNCH_CHECK();
  }
  SbWgradArgs g;
  memset(&g, 0, sizeof(g));
  g.off = w.off_seq; g.B = B; g.part_stride = stride;
  g.dY[0] = dql; g.X = u.xl; g.gW[0] = gp + Cfg::oWq; g.gb[0] = gp + Cfg::obq;
  RC_TRY((sb_wgrad<D, 1>(g, (int64_t)B, s)));
  return RC_OK;
}

template <int D>
static int sb_forward(const float* item_emb, const float* pos_emb, const SasLayer* layer, int n_layers, int n_heads,
                      const int64_t* hist, const int64_t* lengths, int B, int L, float* hv, float* state,
                      const SbWs& w, SbDrop dr, hipStream_t s) {
  constexpr int LPR = D / 4;
  const bool drop = dr.seed != nullptr;
  const size_t rmax = (size_t)B * L;
  const int last_mode = sb_last_row_mode<D>(n_heads, B, L, drop);
  const bool last_row = last_mode != 0;
  // one block, K / V-free last-row path: the block reads the table rows itself (and stores them padded, [B, L, D], for the
  // backward) -- no compact row space, no embedding pass
  const bool lr_gather = last_mode == 2 && n_layers == 1;
  if (L > kSasLP && !lr_gather)
    return fail(RC_ERR_UNSUPPORTED, "rc_sasrec_batch_fwd: history_max %d > %d is covered by the one-row path only (one block, no "
                "dropout, 1 / 2 / 4 heads, RC_SAS_LAST_ROW unset)", L, kSasLP);
  SbSaved sv = sb_saved(state, 0, rmax, D);
  if (!lr_gather) {
    hipLaunchKernelGGL(sb_offsets_kernel, dim3(1), dim3(kBlock), 0, s, lengths, B, L, w.off, w.off_seq);
    RC_LAUNCH_CHECK();
    hipLaunchKernelGGL((sb_embed_kernel<D>), dim3(sb_row_grid((int64_t)rmax, LPR)), dim3(kBlock), 0, s, item_emb, pos_emb, hist,
                       lengths, B, L, w.off, sv.x);
    RC_LAUNCH_CHECK();
  }
  for (int l = 0; l < n_layers; ++l) {
    const SasLayer& p = layer[l];
    sv = sb_saved(state, l, rmax, D);
    float* xnext = l + 1 < n_layers ? sb_saved(state, l + 1, rmax, D).x : state + (size_t)n_layers * sb_layer_floats(rmax, D);
    SbLinArgs a;
    memset(&a, 0, sizeof(a));
    if (last_mode == 2 && l == n_layers - 1) {
      RC_TRY((sb_last_block_fwd<D>(item_emb, pos_emb, p, n_heads, hist, lengths, B, L, lr_gather, sv, hv, w, dr, s)));
      return RC_OK;
    }
    if (last_row && l == n_layers - 1) {
      // k, v for all rows; the last row's x and q; one attention row per (sequence, head); the block on B rows -> hv
      float* xl = sv.q + (size_t)B * D;   // [B, D] the sequences' last rows (kept for the backward); sv.q[0, B): their queries
      float* ctxl = w.t0;                 // [B, D]
      a.off = w.off; a.B = B;
      a.X = sv.x; a.W[0] = p.Wk; a.W[1] = p.Wv; a.bias[0] = p.bk; a.bias[1] = p.bv; a.Y[0] = sv.k; a.Y[1] = sv.v;
      {
        const size_t lds = (size_t)(2 * D * (D + 4) + 2 * D) * sizeof(float);
        auto kern = sb_qkv16_kernel<D, 2>;
        RC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        int grid, block;
        sb_rows16_geometry((int64_t)rmax, &grid, &block);
        hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3((unsigned)block), lds, s, a);
        RC_LAUNCH_CHECK();
      }
      hipLaunchKernelGGL((sb_last_rows_kernel<D>), dim3(sb_row_grid(B, LPR)), dim3(kBlock), 0, s, sv.x, lengths, w.off, B, L, xl);
      RC_LAUNCH_CHECK();
      memset(&a, 0, sizeof(a));
      a.off = w.off_seq; a.B = B;
      a.X = xl; a.W[0] = p.Wq; a.bias[0] = p.bq; a.Y[0] = sv.q;
      {
        const size_t lds = (size_t)(D * (D + 4) + D) * sizeof(float);
        auto kern = sb_qkv16_kernel<D, 1>;
        RC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        int grid, block;
        sb_rows16_geometry((int64_t)B, &grid, &block);
        hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3((unsigned)block), lds, s, a);
        RC_LAUNCH_CHECK();
      }
      SbLastAttnArgs la;
      memset(&la, 0, sizeof(la));
      la.q = sv.q; la.k = sv.k; la.v = sv.v; la.ctx = ctxl; la.lengths = lengths; la.off = w.off; la.B = B; la.L = L; la.n_heads = n_heads;
      {
        const int dk = D / n_heads;
        const unsigned blocks = (unsigned)(((int64_t)B * n_heads + kBlock / 64 - 1) / (kBlock / 64));
        if (dk == 16) hipLaunchKernelGGL((sb_attn_last_kernel<D, 16, false>), dim3(blocks), dim3(kBlock), 0, s, la);
        else if (dk == 32) hipLaunchKernelGGL((sb_attn_last_kernel<D, 32, false>), dim3(blocks), dim3(kBlock), 0, s, la);
        else hipLaunchKernelGGL((sb_attn_last_kernel<D, (D >= 64 ? 64 : 32), false>), dim3(blocks), dim3(kBlock), 0, s, la);
        RC_LAUNCH_CHECK();
      }
      SbBlockArgs bk;
      bk.ctx = ctxl; bk.x = xl; bk.ln1w = p.ln1w; bk.ln1b = p.ln1b; bk.W1 = p.W1; bk.b1 = p.b1; bk.W2 = p.W2; bk.b2 = p.b2;
      bk.ln2w = p.ln2w; bk.ln2b = p.ln2b; bk.xh1 = sv.xh1; bk.rstd1 = sv.rstd1; bk.y1 = sv.y1; bk.h = sv.h; bk.xh2 = sv.xh2;
      bk.rstd2 = sv.rstd2; bk.xnext = hv; bk.off = w.off_seq; bk.B = B; bk.dr = dr;
      {
        const size_t lds16 = ((size_t)2 * D * (D + 4) + 6 * (size_t)D) * sizeof(float);
        auto kern16 = sb_block16_fwd_kernel<D>;
        RC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern16), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds16));
        int64_t gr = (((int64_t)B + 15) / 16 + 3) / 4;
        if (gr > 768) gr = 768;
        hipLaunchKernelGGL(kern16, dim3((unsigned)(gr < 1 ? 1 : gr)), dim3(256), lds16, s, bk);
        RC_LAUNCH_CHECK();
      }
      hipLaunchKernelGGL((sb_mask_empty_kernel<D>), dim3(sb_row_grid(B, LPR)), dim3(kBlock), 0, s, hv, lengths, B, hv);
      RC_LAUNCH_CHECK();
      return RC_OK;
    }
    a.off = w.off; a.B = B;
    a.X = sv.x; a.W[0] = p.Wq; a.W[1] = p.Wk; a.W[2] = p.Wv; a.bias[0] = p.bq; a.bias[1] = p.bk; a.bias[2] = p.bv;
    a.Y[0] = sv.q; a.Y[1] = sv.k; a.Y[2] = sv.v;
    if (sb_rows16()) {
      const size_t lds = (size_t)(3 * D * (D + 4) + 3 * D) * sizeof(float);
      auto kern = sb_qkv16_kernel<D, 3>;
      RC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      int grid, block;
      sb_rows16_geometry((int64_t)rmax, &grid, &block);
      hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3((unsigned)block), lds, s, a);
      RC_LAUNCH_CHECK();
    } else {
      RC_TRY((sb_linear<D, 3, false>(a, (int64_t)rmax, s)));
    }
    SbAttnArgs at;
    memset(&at, 0, sizeof(at));
    at.q = sv.q; at.k = sv.k; at.v = sv.v; at.ctx = w.t0; at.lengths = lengths; at.off = w.off; at.B = B; at.L = L;
    at.n_heads = n_heads;
    RC_TRY((sb_attention<D, false>(at, w.bucket, l == 0, s)));
    dr.site = 2u * (uint32_t)l;      // dropout1 on the attention context (utils/layers.py:110)
    if (sb_fused_block()) {          // LayerNorm1 -> FFN -> LayerNorm2 in one launch
      SbBlockArgs bk;
      bk.ctx = w.t0; bk.x = sv.x; bk.ln1w = p.ln1w; bk.ln1b = p.ln1b; bk.W1 = p.W1; bk.b1 = p.b1; bk.W2 = p.W2; bk.b2 = p.b2;
      bk.ln2w = p.ln2w; bk.ln2b = p.ln2b; bk.xh1 = sv.xh1; bk.rstd1 = sv.rstd1; bk.y1 = sv.y1; bk.h = sv.h; bk.xh2 = sv.xh2;
      bk.rstd2 = sv.rstd2; bk.xnext = xnext; bk.off = w.off; bk.B = B; bk.dr = dr;
      if (sb_rows16()) {   // 16-row tiles, activations chained through registers
        const size_t lds16 = ((size_t)2 * D * (D + 4) + 6 * (size_t)D) * sizeof(float);
        auto kern16 = sb_block16_fwd_kernel<D>;
        RC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern16), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds16));
        const int64_t tiles16 = ((int64_t)rmax + 15) / 16;
        // 140 registers per lane: three waves per SIMD.  Four-wave workgroups place one wave on each SIMD, so three of them
        // share a CU (36 KB of LDS each); an eight-wave workgroup would be alone on its CU with two waves per SIMD
        int nw = 4;
        if (const char* v = getenv("RC_SB16_BLOCK_WAVES")) nw = atoi(v) == 8 ? 8 : 4;   // experiment switch
        int64_t gr = (tiles16 + nw - 1) / nw;
        const int64_t cap16 = nw == 8 ? 256 : 768;
        if (gr > cap16) gr = cap16;
        hipLaunchKernelGGL(kern16, dim3((unsigned)(gr < 1 ? 1 : gr)), dim3(64 * nw), lds16, s, bk);
        RC_LAUNCH_CHECK();
        continue;
      }
      const size_t lds = ((size_t)(2 * D + 2 * kSbTile) * (D + 1) + 2 * (size_t)D) * sizeof(float);
      auto kern = sb_block_fwd_kernel<D>;
      RC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      int64_t tiles = ((int64_t)rmax + kSbTile - 1) / kSbTile;
      const int per_cu = (int)((160 * 1024) / lds);
      const int64_t cap = 256 * (int64_t)(per_cu < 1 ? 1 : (per_cu > 4 ? 4 : per_cu));
      if (tiles > cap) tiles = cap;
      hipLaunchKernelGGL(kern, dim3((unsigned)(tiles < 1 ? 1 : tiles)), dim3(kBlock), lds, s, bk);
      RC_LAUNCH_CHECK();
      continue;
    }
    hipLaunchKernelGGL((sb_ln_fwd_kernel<D>), dim3(sb_row_grid((int64_t)rmax, LPR)), dim3(kBlock), 0, s, w.t0, sv.x, p.ln1w,
                       p.ln1b, w.off, B, sv.xh1, sv.rstd1, sv.y1, dr);
    RC_LAUNCH_CHECK();
    memset(&a, 0, sizeof(a));
    a.off = w.off; a.B = B;
    a.X = sv.y1; a.W[0] = p.W1; a.bias[0] = p.b1; a.Y[0] = sv.h; a.relu = 1;
    RC_TRY((sb_linear<D, 1, false>(a, (int64_t)rmax, s)));
    // without dropout the residual y1 is added by the projection's epilogue; with dropout2 (utils/layers.py:117) the
    // LayerNorm kernel masks the FFN output first and adds the residual itself
    a.X = sv.h; a.W[0] = p.W2; a.bias[0] = p.b2; a.Y[0] = w.t0; a.relu = 0; a.res = drop ? nullptr : sv.y1;
    RC_TRY((sb_linear<D, 1, false>(a, (int64_t)rmax, s)));
    dr.site = 2u * (uint32_t)l + 1u;
    hipLaunchKernelGGL((sb_ln_fwd_kernel<D>), dim3(sb_row_grid((int64_t)rmax, LPR)), dim3(kBlock), 0, s, w.t0,
                       drop ? static_cast<const float*>(sv.y1) : static_cast<const float*>(nullptr), p.ln2w, p.ln2b, w.off, B,
                       sv.xh2, sv.rstd2, xnext, dr);
    RC_LAUNCH_CHECK();
  }
  const float* xout = state + (size_t)n_layers * sb_layer_floats(rmax, D);
  hipLaunchKernelGGL((sb_last_rows_kernel<D>), dim3(sb_row_grid(B, LPR)), dim3(kBlock), 0, s, xout, lengths, w.off, B, L, hv);
  RC_LAUNCH_CHECK();
  return RC_OK;
}

template <int D, int NP>
static int sb_wgrad(const SbWgradArgs& a, int64_t rmax, hipStream_t s) {
  const size_t lds = (size_t)(1 + NP) * kSbTile * (D + 1) * sizeof(float);
  auto kern = sb_wgrad_kernel<D, NP>;
  RC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  int64_t tiles = (rmax + kSbTile - 1) / kSbTile;
  if (tiles > kSbPartWg) tiles = kSbPartWg;
  hipLaunchKernelGGL(kern, dim3((unsigned)(tiles < 1 ? 1 : tiles)), dim3(kBlock), lds, s, a);
  RC_LAUNCH_CHECK();
  return RC_OK;
}

template <int D>
static int sb_backward(const SasLayer* layer, int n_layers, int n_heads, const int64_t* lengths, int B, int L,
                       const float* state, const float* dhv, float* g_hist, float* dense_out, const SbWs& w,
                       SbDrop dr, hipStream_t s, int part = 0) {
  using Cfg = SasCfg<D>;
  const bool drop = dr.seed != nullptr;
  constexpr int LPR = D / 4, PL = Cfg::PL;
  const size_t rmax = (size_t)B * L;
  const size_t stride = (size_t)n_layers * PL;  // floats between the slices of consecutive workgroups
  const int last_mode = sb_last_row_mode<D>(n_heads, B, L, drop);
  const bool last_row = last_mode != 0;
  const bool lean = last_mode == 2 && n_layers == 1;   // every launch writes the same sb_last_slots(B) slices
  if (L > kSasLP && !lean)
    return fail(RC_ERR_UNSUPPORTED, "rc_sasrec_batch_bwd: history_max %d > %d is covered by the one-row path only (one block, no "
                "dropout, 1 / 2 / 4 heads, RC_SAS_LAST_ROW unset)", L, kSasLP);
  const int slots = lean ? sb_last_slots(B) : kSbPartWg;
  if (!lean) {      // only the one-block last-row path splits: here part 1 is the whole backward pass, part 2 nothing
    if (part == 2) return RC_OK;
    part = 0;
  }
  if (!lean) RC_HIP(hipMemsetAsync(w.part, 0, (size_t)kSbPartWg * stride * sizeof(float), s));
  float* G = w.t0;
  if (!last_row) {
    hipLaunchKernelGGL((sb_seed_kernel<D>), dim3(sb_row_grid((int64_t)rmax, LPR)), dim3(kBlock), 0, s, dhv, lengths, w.off, B, L, G);
    RC_LAUNCH_CHECK();
  }
  const int ln_grid = sb_row_grid((int64_t)rmax, LPR) < kSbPartWg ? sb_row_grid((int64_t)rmax, LPR) : kSbPartWg;
  for (int l = n_layers - 1; l >= 0; --l) {
    const SasLayer& p = layer[l];
    const SbSaved sv = sb_saved(const_cast<float*>(state), l, rmax, D);
    float* gp = w.part + (size_t)l * PL;
    if (last_mode == 2 && l == n_layers - 1) {
      RC_TRY((sb_last_block_bwd<D>(p, n_heads, lengths, B, L, n_layers == 1, sv, dhv, n_layers == 1 ? g_hist : G, gp, stride, w, dr, s, part)));
      continue;
    }
    if (last_row && l == n_layers - 1) {
      // the last block saw one row per sequence (sb_forward): block backward on B rows, one attention row per (sequence, head)
      // backward -> dQ [B], dK / dV [R]; dX = dK Wk + dV Wv on all rows, + (dQ Wq + dZ1) on the last rows
      const float* xl = sv.q + (size_t)B * D;
      float* gl = w.t1;                        // [B, D]: dhv (empty histories: 0), then dZ1
      float* dql = w.t1 + (size_t)B * D;       // [B, D]
      float* tl = w.t4;                        // [B, D]
      hipLaunchKernelGGL((sb_mask_empty_kernel<D>), dim3(sb_row_grid(B, LPR)), dim3(kBlock), 0, s, dhv, lengths, B, gl);
      RC_LAUNCH_CHECK();
      {
        SbBlockBwdArgs bb;
        bb.G = gl; bb.Gb = gl; bb.xh2 = sv.xh2; bb.rstd2 = sv.rstd2; bb.h = sv.h; bb.y1 = sv.y1; bb.xh1 = sv.xh1; bb.rstd1 = sv.rstd1;
        bb.ln2w = p.ln2w; bb.W2 = p.W2; bb.W1 = p.W1; bb.ln1w = p.ln1w; bb.part = gp; bb.part_stride = stride; bb.off = w.off_seq; bb.B = B;
        bb.dr = dr;
        bb.dr.site = 2u * (uint32_t)l;
        const size_t lds = (size_t)(2 * D + 3 * kSbTile) * (D + 1) * sizeof(float);
        auto kern = sb_block_bwd_kernel<D>;
        RC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        int64_t tiles = ((int64_t)B + kSbTile - 1) / kSbTile;
        if (tiles > kSbPartWg) tiles = kSbPartWg;
        hipLaunchKernelGGL(kern, dim3((unsigned)(tiles < 1 ? 1 : tiles)), dim3(kBlock), lds, s, bb);
        RC_LAUNCH_CHECK();
      }
      SbLastAttnArgs la;
      memset(&la, 0, sizeof(la));
      la.q = sv.q; la.k = sv.k; la.v = sv.v; la.dctx = gl; la.dq = dql; la.dk = w.t2; la.dv = w.t3;
      la.lengths = lengths; la.off = w.off; la.B = B; la.L = L; la.n_heads = n_heads;
      {
        const int dk = D / n_heads;
        const unsigned blocks = (unsigned)(((int64_t)B * n_heads + kBlock / 64 - 1) / (kBlock / 64));
        if (dk == 16) hipLaunchKernelGGL((sb_attn_last_kernel<D, 16, true>), dim3(blocks), dim3(kBlock), 0, s, la);
        else if (dk == 32) hipLaunchKernelGGL((sb_attn_last_kernel<D, 32, true>), dim3(blocks), dim3(kBlock), 0, s, la);
        else hipLaunchKernelGGL((sb_attn_last_kernel<D, (D >= 64 ? 64 : 32), true>), dim3(blocks), dim3(kBlock), 0, s, la);
        RC_LAUNCH_CHECK();
      }
      SbWgradArgs g;
      memset(&g, 0, sizeof(g));
      g.off = w.off; g.B = B; g.part_stride = stride;
      g.dY[0] = w.t2; g.dY[1] = w.t3; g.X = sv.x;
      g.gW[0] = gp + Cfg::oWk; g.gb[0] = gp + Cfg::obk; g.gW[1] = gp + Cfg::oWv; g.gb[1] = gp + Cfg::obv;
      RC_TRY((sb_wgrad<D, 2>(g, (int64_t)rmax, s)));
      memset(&g, 0, sizeof(g));
      g.off = w.off_seq; g.B = B; g.part_stride = stride;
      g.dY[0] = dql; g.X = xl; g.gW[0] = gp + Cfg::oWq; g.gb[0] = gp + Cfg::obq;
      RC_TRY((sb_wgrad<D, 1>(g, (int64_t)B, s)));
      {
        SbSum3Args q;
        memset(&q, 0, sizeof(q));
        q.X[0] = w.t2; q.X[1] = w.t3; q.W[0] = p.Wk; q.W[1] = p.Wv; q.res = nullptr; q.Y = G; q.off = w.off; q.B = B;
        const size_t lds16 = (size_t)(2 * D * (D + 4)) * sizeof(float);
        auto kern16 = sb_sum3_16_kernel<D, 2>;
        RC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern16), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds16));
        int grid, block;
        sb_rows16_geometry((int64_t)rmax, &grid, &block);
        hipLaunchKernelGGL(kern16, dim3((unsigned)grid), dim3((unsigned)block), lds16, s, q);
        RC_LAUNCH_CHECK();
        memset(&q, 0, sizeof(q));
        q.X[0] = dql; q.W[0] = p.Wq; q.res = gl; q.Y = tl; q.off = w.off_seq; q.B = B;
        const size_t lds1 = (size_t)(D * (D + 4)) * sizeof(float);
        auto kern1 = sb_sum3_16_kernel<D, 1>;
        RC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern1), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1));
        sb_rows16_geometry((int64_t)B, &grid, &block);
        hipLaunchKernelGGL(kern1, dim3((unsigned)grid), dim3((unsigned)block), lds1, s, q);
        RC_LAUNCH_CHECK();
      }
      hipLaunchKernelGGL((sb_last_add_kernel<D>), dim3(sb_row_grid(B, LPR)), dim3(kBlock), 0, s, tl, lengths, w.off, B, L, G);
      RC_LAUNCH_CHECK();
      continue;
    }
    // LayerNorm2
    // (dropout: G = dZ2 feeds the residual path, Gb = mask2 * dZ2 the FFN branch)
    float* Gb = drop ? w.t4 : G;
    SbWgradArgs g;
    SbLinArgs a;
    if (sb_fused_block()) {   // LayerNorm2 backward -> FFN backward (both weight gradients) -> LayerNorm1 backward: one launch
      SbBlockBwdArgs bb;
      bb.G = G; bb.Gb = Gb; bb.xh2 = sv.xh2; bb.rstd2 = sv.rstd2; bb.h = sv.h; bb.y1 = sv.y1; bb.xh1 = sv.xh1; bb.rstd1 = sv.rstd1;
      bb.ln2w = p.ln2w; bb.W2 = p.W2; bb.W1 = p.W1; bb.ln1w = p.ln1w; bb.part = gp; bb.part_stride = stride; bb.off = w.off; bb.B = B;
      bb.dr = dr;
      bb.dr.site = 2u * (uint32_t)l;
      const size_t lds = (size_t)(2 * D + (drop ? 4 : 3) * kSbTile) * (D + 1) * sizeof(float);
      auto kern = sb_block_bwd_kernel<D>;
      RC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      int64_t tiles = ((int64_t)rmax + kSbTile - 1) / kSbTile;
      const int per_cu = (int)((160 * 1024) / lds);
      int64_t cap = 256 * (int64_t)(per_cu < 1 ? 1 : (per_cu > 2 ? 2 : per_cu));
      if (cap > kSbPartWg) cap = kSbPartWg;   // one partial-gradient slice per workgroup
      if (tiles > cap) tiles = cap;
      hipLaunchKernelGGL(kern, dim3((unsigned)(tiles < 1 ? 1 : tiles)), dim3(kBlock), lds, s, bb);
      RC_LAUNCH_CHECK();
    } else {
    dr.site = 2u * (uint32_t)l + 1u;
    hipLaunchKernelGGL((sb_ln_bwd_kernel<D>), dim3(ln_grid), dim3(kBlock), 0, s, G, sv.xh2, sv.rstd2, p.ln2w, w.off, B,
                       gp + Cfg::oln2w, gp + Cfg::oln2b, stride, dr, Gb);
    RC_LAUNCH_CHECK();
    // FFN: dW2, db2; dHpre = (dZ2 . W2) * relu'(h); dW1, db1; dY1 = dZ2 + dHpre . W1
    memset(&g, 0, sizeof(g));
    g.off = w.off; g.B = B; g.part_stride = stride;
    g.dY[0] = Gb; g.X = sv.h; g.gW[0] = gp + Cfg::oW2; g.gb[0] = gp + Cfg::ob2;
    RC_TRY((sb_wgrad<D, 1>(g, (int64_t)rmax, s)));
    memset(&a, 0, sizeof(a));
    a.off = w.off; a.B = B;
    a.X = Gb; a.W[0] = p.W2; a.Y[0] = w.t1; a.mask = sv.h;
    RC_TRY((sb_linear<D, 1, true>(a, (int64_t)rmax, s)));
    g.dY[0] = w.t1; g.X = sv.y1; g.gW[0] = gp + Cfg::oW1; g.gb[0] = gp + Cfg::ob1;
    RC_TRY((sb_wgrad<D, 1>(g, (int64_t)rmax, s)));
    a.X = w.t1; a.W[0] = p.W1; a.Y[0] = G; a.mask = nullptr; a.res = G;
    RC_TRY((sb_linear<D, 1, true>(a, (int64_t)rmax, s)));
    // LayerNorm1: G = dZ1 = dCtx = the residual branch of dX
    dr.site = 2u * (uint32_t)l;
    hipLaunchKernelGGL((sb_ln_bwd_kernel<D>), dim3(ln_grid), dim3(kBlock), 0, s, G, sv.xh1, sv.rstd1, p.ln1w, w.off, B,
                       gp + Cfg::oln1w, gp + Cfg::oln1b, stride, dr, Gb);
    RC_LAUNCH_CHECK();
    }
    memset(&g, 0, sizeof(g));
    g.off = w.off; g.B = B; g.part_stride = stride;
    // attention
    SbAttnArgs at;
    memset(&at, 0, sizeof(at));
    at.q = sv.q; at.k = sv.k; at.v = sv.v; at.dctx = Gb; at.dq = w.t1; at.dk = w.t2; at.dv = w.t3;
    at.lengths = lengths; at.off = w.off; at.B = B; at.L = L; at.n_heads = n_heads;
    RC_TRY((sb_attention<D, true>(at, w.bucket, false, s)));
    // projections: parameter gradients against the layer input, dX = dZ1 + dQ Wq + dK Wk + dV Wv
    g.dY[0] = w.t1; g.dY[1] = w.t2; g.dY[2] = w.t3; g.X = sv.x;
    g.gW[0] = gp + Cfg::oWq; g.gb[0] = gp + Cfg::obq; g.gW[1] = gp + Cfg::oWk; g.gb[1] = gp + Cfg::obk;
    g.gW[2] = gp + Cfg::oWv; g.gb[2] = gp + Cfg::obv;
    RC_TRY((sb_wgrad<D, 3>(g, (int64_t)rmax, s)));
    {
      SbSum3Args q;
      q.X[0] = w.t1; q.X[1] = w.t2; q.X[2] = w.t3; q.W[0] = p.Wq; q.W[1] = p.Wk; q.W[2] = p.Wv;
      q.res = G; q.Y = G; q.off = w.off; q.B = B;
      if (sb_rows16()) {
        const size_t lds16 = (size_t)(3 * D * (D + 4)) * sizeof(float);
        auto kern16 = sb_sum3_16_kernel<D, 3>;
        RC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern16), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds16));
        int grid, block;
        sb_rows16_geometry((int64_t)rmax, &grid, &block);
        hipLaunchKernelGGL(kern16, dim3((unsigned)grid), dim3((unsigned)block), lds16, s, q);
        RC_LAUNCH_CHECK();
        continue;
      }
      const size_t lds = (size_t)(3 * D + kSbTile) * (D + 1) * sizeof(float);
      auto kern = sb_linear3_sum_kernel<D>;
      RC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      int64_t tiles = ((int64_t)rmax + kSbTile - 1) / kSbTile;
      const int per_cu = (int)((160 * 1024) / lds);
      const int64_t cap = 256 * (int64_t)(per_cu < 1 ? 1 : per_cu);
      if (tiles > cap) tiles = cap;
      hipLaunchKernelGGL(kern, dim3((unsigned)(tiles < 1 ? 1 : tiles)), dim3(kBlock), lds, s, q);
      RC_LAUNCH_CHECK();
    }
  }
  if (!(last_mode == 2 && n_layers == 1)) {   // (that path writes the padded layout itself)
    hipLaunchKernelGGL((sb_unpack_kernel<D>), dim3(sb_row_grid((int64_t)rmax, LPR)), dim3(kBlock), 0, s, G, lengths, w.off, B, L,
                       g_hist);
    RC_LAUNCH_CHECK();
  }
  if (part == 1) return RC_OK;     // (lean: the unpack above does not run; the parameter gradients and their reduction are part 2)
  const int count = n_layers * PL;
  if (lean)   // (one block: the index range Wk .. bv of its parameter block has its own slice count)
    hipLaunchKernelGGL(sas_reduce_partials_kernel, dim3((count + 63) / 64), dim3(kBlock), 0, s, w.part, slots, count,
                       dense_out, (int)Cfg::oWk, (int)Cfg::oln1w, sb_last_kv_slots(B));
  else
    hipLaunchKernelGGL(sas_reduce_partials_kernel, dim3((count + 63) / 64), dim3(kBlock), 0, s, w.part, slots, count,
                       dense_out, 0, 0, 0);
  RC_LAUNCH_CHECK();
  return RC_OK;
}

}  // namespace rc

using namespace rc;

extern "C" size_t rc_sasrec_batch_state_floats(int B, int L, int d, int n_layers) {
  if (B < 1 || L < 1 || d < 1 || n_layers < 1) return 0;
  return sb_state_act_floats((size_t)B * L, d, n_layers) + sb_state_int_floats(B);
}

extern "C" size_t rc_sasrec_batch_workspace_bytes(int B, int L, int d, int n_layers) {
  if (B < 1 || L < 1 || d < 1 || n_layers < 1) return 0;
  return sb_carve(nullptr, nullptr, B, L, d, n_layers, true).total + 256;
}

// dropout arguments -> kernel fields; p == 0 switches the mask off
static int sb_set_dropout(SbDrop& dr, float drop_p, const uint64_t* seed_dev, const char* who) {
  memset(&dr, 0, sizeof(dr));
  if (!(drop_p >= 0.f && drop_p < 1.f)) return fail(RC_ERR_INVALID_ARG, "%s: dropout p=%g outside [0, 1)", who, (double)drop_p);
  if (drop_p > 0.f && seed_dev == nullptr) return fail(RC_ERR_INVALID_ARG, "%s: dropout p=%g needs a device seed", who, (double)drop_p);
  if (drop_p > 0.f) {
    dr.seed = seed_dev;
    dr.thresh = (uint32_t)((double)drop_p * 4294967296.0);
    dr.scale = 1.0f / (1.0f - drop_p);
  }
  return RC_OK;
}

extern "C" int rc_sasrec_batch_fwd_dropout(const float* item_emb, const float* pos_emb, const float* const* layer_params,
                                           int n_layers, int n_heads, const int64_t* hist, const int64_t* lengths, int B,
                                           int L, int d, float drop_p, const uint64_t* seed_dev, float* hv, float* state,
                                           void* ws, size_t ws_bytes, rc_stream_t stream) {
  if (B == 0) return RC_OK;
  RC_REQUIRE(item_emb && pos_emb && hist && lengths && hv && state && ws, "rc_sasrec_batch_fwd: null pointer");
  if (!rc_sasrec_supported(d, n_layers, n_heads, L))
    return fail(RC_ERR_UNSUPPORTED, "rc_sasrec_batch_fwd: d=%d layers=%d heads=%d L=%d not supported", d, n_layers, n_heads, L);
  RC_REQUIRE((int64_t)B * L < ((int64_t)1 << 31), "rc_sasrec_batch_fwd: B * L too large");
  SasLayer layer[kSasMaxLayers];
  RC_TRY(sb_fill_layers(layer, layer_params, n_layers));
  SbDrop dr;
  RC_TRY(sb_set_dropout(dr, drop_p, seed_dev, "rc_sasrec_batch_fwd"));
  const SbWs w = sb_carve(ws, state, B, L, d, n_layers, false);
  if (ws_bytes < w.total) return fail(RC_ERR_WORKSPACE, "rc_sasrec_batch_fwd: workspace %zu < %zu", ws_bytes, w.total);
  hipStream_t s = as_stream(stream);
  return d == 64 ? sb_forward<64>(item_emb, pos_emb, layer, n_layers, n_heads, hist, lengths, B, L, hv, state, w, dr, s)
                 : sb_forward<32>(item_emb, pos_emb, layer, n_layers, n_heads, hist, lengths, B, L, hv, state, w, dr, s);
}

extern "C" int rc_sasrec_batch_fwd(const float* item_emb, const float* pos_emb, const float* const* layer_params,
                                   int n_layers, int n_heads, const int64_t* hist, const int64_t* lengths, int B, int L,
                                   int d, float* hv, float* state, void* ws, size_t ws_bytes, rc_stream_t stream) {
  return rc_sasrec_batch_fwd_dropout(item_emb, pos_emb, layer_params, n_layers, n_heads, hist, lengths, B, L, d, 0.f, nullptr,
                                     hv, state, ws, ws_bytes, stream);
}

extern "C" int rc_sasrec_batch_bwd_dropout(const float* const* layer_params, int n_layers, int n_heads,
                                           const int64_t* lengths, int B, int L, int d, float drop_p,
                                           const uint64_t* seed_dev, const float* state, const float* dhv, float* g_hist,
                                           float* dense_grads, void* ws, size_t ws_bytes, rc_stream_t stream) {
  if (B == 0) return RC_OK;
  RC_REQUIRE(lengths && state && dhv && g_hist && dense_grads && ws, "rc_sasrec_batch_bwd: null pointer");
  if (!rc_sasrec_supported(d, n_layers, n_heads, L))
    return fail(RC_ERR_UNSUPPORTED, "rc_sasrec_batch_bwd: d=%d layers=%d heads=%d L=%d not supported", d, n_layers, n_heads, L);
  SasLayer layer[kSasMaxLayers];
  RC_TRY(sb_fill_layers(layer, layer_params, n_layers));
  SbDrop dr;
  RC_TRY(sb_set_dropout(dr, drop_p, seed_dev, "rc_sasrec_batch_bwd"));
  const SbWs w = sb_carve(ws, const_cast<float*>(state), B, L, d, n_layers, true);
  if (ws_bytes < w.total) return fail(RC_ERR_WORKSPACE, "rc_sasrec_batch_bwd: workspace %zu < %zu", ws_bytes, w.total);
  hipStream_t s = as_stream(stream);
  return d == 64 ? sb_backward<64>(layer, n_layers, n_heads, lengths, B, L, state, dhv, g_hist, dense_grads, w, dr, s)
                 : sb_backward<32>(layer, n_layers, n_heads, lengths, B, L, state, dhv, g_hist, dense_grads, w, dr, s);
}

/* The backward pass in two calls, for a caller that lets other work start as soon as the history rows' gradient g_hist is complete
 * (the item-table update on another stream) while the encoder's parameter gradients are still being formed: part 1 = every launch up
 * to and including the one that completes g_hist, part 2 = the rest (dense_grads is complete after part 2).  Only the one-block
 * last-row path has such a tail (rc_sasrec_batch_bwd_splits: 1); elsewhere part 1 is the whole pass and part 2 nothing. */
extern "C" int rc_sasrec_batch_bwd_splits(int d, int n_layers, int n_heads, int B, int L, float drop_p) {
  if (!rc_sasrec_supported(d, n_layers, n_heads, L) || n_layers != 1) return 0;
  const int mode = d == 64 ? sb_last_row_mode<64>(n_heads, B, L, drop_p > 0.f) : sb_last_row_mode<32>(n_heads, B, L, drop_p > 0.f);
  return mode == 2 ? 1 : 0;
}

extern "C" int rc_sasrec_batch_bwd_part(const float* const* layer_params, int n_layers, int n_heads, const int64_t* lengths, int B,
                                        int L, int d, float drop_p, const uint64_t* seed_dev, const float* state, const float* dhv,
                                        float* g_hist, float* dense_grads, void* ws, size_t ws_bytes, int part, rc_stream_t stream) {
  if (B == 0) return RC_OK;
  RC_REQUIRE(part == 1 || part == 2, "rc_sasrec_batch_bwd_part: part %d (1 or 2)", part);
  RC_REQUIRE(lengths && state && dhv && g_hist && dense_grads && ws, "rc_sasrec_batch_bwd_part: null pointer");
  if (!rc_sasrec_supported(d, n_layers, n_heads, L))
    return fail(RC_ERR_UNSUPPORTED, "rc_sasrec_batch_bwd_part: d=%d layers=%d heads=%d L=%d not supported", d, n_layers, n_heads, L);
  SasLayer layer[kSasMaxLayers];
  RC_TRY(sb_fill_layers(layer, layer_params, n_layers));
  SbDrop dr;
  RC_TRY(sb_set_dropout(dr, drop_p, seed_dev, "rc_sasrec_batch_bwd_part"));
  const SbWs w = sb_carve(ws, const_cast<float*>(state), B, L, d, n_layers, true);
  if (ws_bytes < w.total) return fail(RC_ERR_WORKSPACE, "rc_sasrec_batch_bwd_part: workspace %zu < %zu", ws_bytes, w.total);
  hipStream_t s = as_stream(stream);
  return d == 64 ? sb_backward<64>(layer, n_layers, n_heads, lengths, B, L, state, dhv, g_hist, dense_grads, w, dr, s, part)
                 : sb_backward<32>(layer, n_layers, n_heads, lengths, B, L, state, dhv, g_hist, dense_grads, w, dr, s, part);
}

extern "C" int rc_sasrec_batch_bwd(const float* const* layer_params, int n_layers, int n_heads, const int64_t* lengths,
                                   int B, int L, int d, const float* state, const float* dhv, float* g_hist,
                                   float* dense_grads, void* ws, size_t ws_bytes, rc_stream_t stream) {
  return rc_sasrec_batch_bwd_dropout(layer_params, n_layers, n_heads, lengths, B, L, d, 0.f, nullptr, state, dhv, g_hist,
                                     dense_grads, ws, ws_bytes, stream);
}

extern "C" size_t rc_sasrec_pos_grad_workspace_bytes(int B, int L, int d) {
  if (B < 1 || L < 1 || d < 1) return 0;
  const size_t chunks = ((size_t)B + kPosChunk - 1) / kPosChunk;
  return align_up(chunks * (size_t)(L + 1) * d * sizeof(float), 256);
}

extern "C" int rc_sasrec_pos_grad(const float* g_hist, const int64_t* lengths, int B, int L, int d, int n_pos,
                                  float* grad_pos, void* ws, size_t ws_bytes, rc_stream_t stream) {
  RC_REQUIRE(grad_pos != nullptr && n_pos >= L + 1 && L >= 1 && (d == 32 || d == 64),
             "rc_sasrec_pos_grad: bad arguments (n_pos=%d, L=%d, d=%d)", n_pos, L, d);
  hipStream_t s = as_stream(stream);
  if (B == 0) {
    RC_HIP(hipMemsetAsync(grad_pos, 0, (size_t)n_pos * d * sizeof(float), s));
    return RC_OK;
  }
  RC_REQUIRE(g_hist && lengths && ws, "rc_sasrec_pos_grad: null pointer");
  if (ws_bytes < rc_sasrec_pos_grad_workspace_bytes(B, L, d))
    return fail(RC_ERR_WORKSPACE, "rc_sasrec_pos_grad: workspace %zu < %zu", ws_bytes, rc_sasrec_pos_grad_workspace_bytes(B, L, d));
  const int chunks = (B + kPosChunk - 1) / kPosChunk;
  float* part = (chunks == 1 && n_pos == L + 1) ? grad_pos : static_cast<float*>(ws);   // (one chunk, no rows to zero: written in place)
  if (d == 64)
    hipLaunchKernelGGL((sb_pos_grad_kernel<64>), dim3(L + 1, chunks), dim3(kBlock), 0, s, g_hist, lengths, B, L, part);
  else
    hipLaunchKernelGGL((sb_pos_grad_kernel<32>), dim3(L + 1, chunks), dim3(kBlock), 0, s, g_hist, lengths, B, L, part);
  RC_LAUNCH_CHECK();
  if (part != grad_pos) {
    const int count = (L + 1) * d, total = n_pos * d;
    hipLaunchKernelGGL(sb_pos_reduce_kernel, dim3((total + kBlock - 1) / kBlock), dim3(kBlock), 0, s, part, chunks, count, total, grad_pos);
    RC_LAUNCH_CHECK();
  }
  return RC_OK;
}
