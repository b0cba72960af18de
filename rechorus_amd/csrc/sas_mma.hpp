// sas_mma.hpp -- pieces shared by the two SASRec encoder implementations (sasrec.hip: one workgroup per
// sequence, everything in LDS; sasrec_batch.hip: batch-level kernels over the compact [sum len, d] row space):
// parameter block layout, the 32x32x2 fp32 MFMA tile product on strided operands, one head's causal
// attention probabilities, the fixed-order reduction of per-workgroup partial gradients.
#pragma once
#include "common.hpp"

namespace rc {

typedef float sas_f32x16 __attribute__((ext_vector_type(16)));

// v_mfma_f32_16x16x4_f32: lane l supplies A[i = l & 15][k = l >> 4] and B[k = l >> 4][j = l & 15], receives C[4 (l >> 4) + r][l & 15]
typedef float sas_f32x4 __attribute__((ext_vector_type(4)));
#if defined(__HIPCC__)
__device__ __forceinline__ sas_f32x4 sas_mfma16(float a, float b, sas_f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ sas_f32x4 sas_zero4() {
  sas_f32x4 z = {0.f, 0.f, 0.f, 0.f};
  return z;
}
#endif

constexpr int kSasLP = 64;        // max rows (history length) per sequence of the all-rows kernels
constexpr int kSasLongLP = 128;   // ... of the batch encoder's one-row path (one block, no dropout; csrc/sas_last_row.hpp)
constexpr int kSasMaxLayers = 4;

// shapes every path covers (per-sequence kernels, all-rows batch kernels, dropout)
inline bool sas_core_supported(int d, int n_layers, int n_heads, int L) {
  return (d == 32 || d == 64) && n_layers >= 1 && n_layers <= kSasMaxLayers && n_heads >= 1 && d % n_heads == 0 && L >= 1 &&
         L <= kSasLP;
}
// 64 < history_max <= 128: the batch encoder when its K / V-free one-row path is the whole encoder
inline bool sas_long_supported(int d, int n_layers, int n_heads, int L) {
  return (d == 32 || d == 64) && n_layers == 1 && (n_heads == 1 || n_heads == 2 || n_heads == 4) && L > kSasLP && L <= kSasLongLP &&
         (d / n_heads) % (d * d / 256) == 0;
}
constexpr float kLnEps = 1e-5f;   // nn.LayerNorm default

struct SasLayer {  // device pointers, nn.Linear layout [out, in]
  const float *Wq, *bq, *Wk, *bk, *Wv, *bv, *ln1w, *ln1b, *W1, *b1, *W2, *b2, *ln2w, *ln2b;
  const float *WqT, *WkT, *WvT, *W1T, *W2T;  // [in, out] copies (sas_transpose_kernel)
};

template <int D>
struct SasCfg {
  static constexpr int SD = D + 1;                               // row stride of [rows][D] buffers
  static constexpr int PL = 5 * D * D + 9 * D;                   // dense-parameter floats per layer
  // offsets inside one layer's parameter-gradient block (canonical order)
  static constexpr int oWq = 0, obq = oWq + D * D, oWk = obq + D, obk = oWk + D * D, oWv = obk + D,
                       obv = oWv + D * D, oln1w = obv + D, oln1b = oln1w + D, oW1 = oln1b + D,
                       ob1 = oW1 + D * D, oW2 = ob1 + D, ob2 = oW2 + D * D, oln2w = ob2 + D,
                       oln2b = oln2w + D;
};

// ---- C[M x N] = A[M x K] . B[K x N] on v_mfma_f32_32x32x2_f32 -------------------------------------
struct MatA { const float* p; int si, sk; };  // a(i,k) = p[i*si + k*sk]
struct MatB { const float* p; int sk, sj; };  // b(k,j) = p[k*sk + j*sj]

// Every wave of the workgroup calls this; 32x32 output blocks are dealt round-robin to the 4 waves.
// epi(i, j, value) runs once per valid output element.  causal: skip blocks entirely above the
// diagonal (their elements are never read).  Out-of-range rows / columns only ever influence
// out-of-range outputs (discarded), so only the K range needs exact masking.
template <typename Epi>
__device__ __forceinline__ void sas_mm_part(MatA A, MatB B, int M, int N, int K, bool causal, Epi epi, int first,
                                            int step) {
  const int lane = threadIdx.x & 63;
  const int nrb = (M + 31) >> 5, ncb = (N + 31) >> 5;
  for (int q = first; q < nrb * ncb; q += step) {
    const int rb = q % nrb, cb = q / nrb;
    if (causal && cb > rb) continue;  // wave-uniform
    const int i = rb * 32 + (lane & 31), j = cb * 32 + (lane & 31), kh = lane >> 5;
    const float* ap = A.p + (i < M ? i : M - 1) * A.si + kh * A.sk;
    const float* bp = B.p + (j < N ? j : N - 1) * B.sj + kh * B.sk;
    sas_f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    int k0 = 0;
    // a workgroup is alone on its CU (LDS), so nothing else hides operand latency: fetch the
    // operands of 16 (then 4) MFMAs before issuing them -- 32 independent loads in flight
    for (; k0 + 32 <= K; k0 += 32) {
      float av[16], bv[16];
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        av[t] = ap[(k0 + 2 * t) * A.sk];
        bv[t] = bp[(k0 + 2 * t) * B.sk];
      }
#pragma unroll
      for (int t = 0; t < 16; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t], bv[t], acc, 0, 0, 0);
    }
    for (; k0 + 8 <= K; k0 += 8) {
      float av[4], bv[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        av[t] = ap[(k0 + 2 * t) * A.sk];
        bv[t] = bp[(k0 + 2 * t) * B.sk];
      }
#pragma unroll
      for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t], bv[t], acc, 0, 0, 0);
    }
    for (; k0 + 2 <= K; k0 += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[k0 * A.sk], bp[k0 * B.sk], acc, 0, 0, 0);
    if (k0 < K) {  // odd K: the kh = 1 half has no column left and must feed zeros
      const float av = kh == 0 ? ap[k0 * A.sk] : 0.f;
      const float bv = kh == 0 ? bp[k0 * B.sk] : 0.f;
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int ii = rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
      if (ii < M && j < N) epi(ii, j, acc[r]);
    }
  }
}

// the 4 waves of a workgroup share the output blocks
template <typename Epi>
__device__ __forceinline__ void sas_mm(MatA A, MatB B, int M, int N, int K, bool causal, Epi epi) {
  sas_mm_part(A, B, M, N, K, causal, epi, (int)(threadIdx.x >> 6), kBlock / 64);
}
// the calling wave computes every output block on its own (no other wave touches the operands or the result)
template <typename Epi>
__device__ __forceinline__ void sas_mm_wave(MatA A, MatB B, int M, int N, int K, bool causal, Epi epi) {
  sas_mm_part(A, B, M, N, K, causal, epi, 0, 1);
}

// x / sqrt(dk): for dk in {1, 4, 16, 64} the divisor is a power of two and the product with its reciprocal is the same
// float, bit for bit -- one multiply instead of the ~10-instruction IEEE division sequence per score
__device__ __forceinline__ float sas_div_scale(float v, float s) {
  return (s == 4.f || s == 2.f || s == 8.f || s == 1.f) ? v * (1.0f / s) : v / s;
}

// ---- row-wise phases of the attention: one lane per (row, part), all rows of a wave at once ---------------------
// The calling wave owns rows [row_base, row_base + rows_here) (rows_here a power of two <= 64); the 64 / rows_here
// lanes of a row take the columns j = part, part + parts, ... and combine through xor-shuffles.  (One row per
// wave iteration with full-wave reductions, the first version, spent ~700 dependent cycles per row.)
__device__ __forceinline__ float sas_parts_sum(float x, int rows_here) {
  for (int off = rows_here; off < 64; off <<= 1) x += __shfl_xor(x, off, 64);
  return x;
}
__device__ __forceinline__ float sas_parts_max(float x, int rows_here) {
  for (int off = rows_here; off < 64; off <<= 1) x = fmaxf(x, __shfl_xor(x, off, 64));
  return x;
}
// A[i][j] <- softmax_j (A[i][j], j <= i), 0 for i < j < n
__device__ __forceinline__ void sas_softmax_causal_rows(float* A, int n, int SA, int row_base, int rows_here) {
  const int lane = threadIdx.x & 63, parts = 64 / rows_here;
  const int i = row_base + lane % rows_here, p = lane / rows_here;
  const bool on = i < n;
  float m = -INFINITY;
  if (on)
#pragma unroll 4
    for (int j = p; j <= i; j += parts) m = fmaxf(m, A[i * SA + j]);
  m = sas_parts_max(m, rows_here);
  float z = 0.f;
  if (on)
#pragma unroll 4
    for (int j = p; j <= i; j += parts) {
      const float e = expf(A[i * SA + j] - m);
      A[i * SA + j] = e;
      z += e;
    }
  z = sas_parts_sum(z, rows_here);
  const float rz = 1.0f / z;   // one division per row; the probabilities are e * (1 / z) (within 1 ulp of e / z)
  if (on)
#pragma unroll 4
    for (int j = p; j < n; j += parts) A[i * SA + j] = j <= i ? A[i * SA + j] * rz : 0.f;
}
// softmax backward in place: T[i][j] <- A[i][j] * (T[i][j] - sum_j' A[i][j'] T[i][j']) / sqrt_dk for j <= i, else 0
__device__ __forceinline__ void sas_softmax_bwd_rows(float* T, const float* A, int n, int SA, float sqrt_dk, int row_base,
                                                     int rows_here) {
  const int lane = threadIdx.x & 63, parts = 64 / rows_here;
  const int i = row_base + lane % rows_here, p = lane / rows_here;
  const bool on = i < n;
  float dot = 0.f;
  if (on)
#pragma unroll 4
    for (int j = p; j <= i; j += parts) dot = fmaf(A[i * SA + j], T[i * SA + j], dot);
  dot = sas_parts_sum(dot, rows_here);
  if (on)
#pragma unroll 4
    for (int j = p; j < n; j += parts) T[i * SA + j] = j <= i ? sas_div_scale(A[i * SA + j] * (T[i * SA + j] - dot), sqrt_dk) : 0.f;
}

// attention probabilities of head hh into A[i][j] (0 for j > i), rows [0, n)
template <int D>
__device__ __forceinline__ void sas_attn_probs(float* A, const float* Q, const float* K, int n, int hh,
                                               int dk, float sqrt_dk, int SA) {
  constexpr int SD = SasCfg<D>::SD;
  sas_mm(MatA{Q + hh * dk, SD, 1}, MatB{K + hh * dk, 1, SD}, n, n, dk, true,
         [&](int i, int j, float v) { A[i * SA + j] = sas_div_scale(v, sqrt_dk); });
  __syncthreads();
  const int rows_here = SA - 1 <= 32 ? 8 : 16;  // rows per wave: 4 waves cover 32 rows, or up to 64
  sas_softmax_causal_rows(A, n, SA, (int)(threadIdx.x >> 6) * rows_here, rows_here);
  __syncthreads();
}

// the same by ONE wave for its own head (A is private to the wave: LDS operations of a wave execute in order, so
// the rows it wrote are visible to its later reads without a workgroup barrier)
template <int D>
__device__ __forceinline__ void sas_attn_probs_wave(float* A, const float* Q, const float* K, int n, int hh, int dk,
                                                    float sqrt_dk, int SA) {
  constexpr int SD = SasCfg<D>::SD;
  sas_mm_wave(MatA{Q + hh * dk, SD, 1}, MatB{K + hh * dk, 1, SD}, n, n, dk, true,
              [&](int i, int j, float v) { A[i * SA + j] = sas_div_scale(v, sqrt_dk); });
  sas_softmax_causal_rows(A, n, SA, 0, 32);  // two lanes per row
  if (n > 32) sas_softmax_causal_rows(A, n, SA, 32, 32);
}

// out[i] = sum_w p[w][i].  A workgroup owns 64 consecutive outputs; wave v sums the partials w = v, v+4, ...
// with four interleaved accumulators (16 independent chains of n_wg/16 adds instead of ONE chain of n_wg,
// which was pure load/add latency), then the 16 chain sums are combined in a fixed order through LDS:
// deterministic, no float atomics.
// (alt_lo <= i < alt_hi: that index range was written by n_wg_alt workgroups instead of n_wg)
static __global__ __launch_bounds__(kBlock) void sas_reduce_partials_kernel(const float* __restrict__ p, int n_wg,
                                                                     int count, float* __restrict__ out, int alt_lo = 0,
                                                                     int alt_hi = 0, int n_wg_alt = 0) {
  __shared__ float sm[4][4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = (int)blockIdx.x * 64 + lane;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  if (i >= alt_lo && i < alt_hi) n_wg = n_wg_alt;
  if (i < count) {
    int w = wave;
    for (; w + 12 < n_wg; w += 16) {
      acc[0] += p[(size_t)w * count + i];
      acc[1] += p[(size_t)(w + 4) * count + i];
      acc[2] += p[(size_t)(w + 8) * count + i];
      acc[3] += p[(size_t)(w + 12) * count + i];
    }
    for (int k = 0; w < n_wg; w += 4, ++k) acc[k] += p[(size_t)w * count + i];
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) sm[wave][k][lane] = acc[k];
  __syncthreads();
  if (wave == 0 && i < count) {
    float total = 0.f;
#pragma unroll
    for (int v = 0; v < 4; ++v)
#pragma unroll
      for (int k = 0; k < 4; ++k) total += sm[v][k][lane];
    out[i] = total;
  }
}

// Split the batch by length: list[0 .. count[0]) = sequences of <= 32 items in batch order, list[B ..
// B + count[1]) = the longer ones.  One workgroup, ballot compaction, order-preserving: the assignment of
// sequences to workgroups (and with it the summation order of the dense-gradient partials) is fixed.
static __global__ __launch_bounds__(kBlock) void sas_bucket_kernel(const int64_t* __restrict__ lengths, int B,
                                                            int32_t* __restrict__ list, int32_t* __restrict__ count) {
  __shared__ int s_wave[2][kBlock / 64];
  __shared__ int s_base[2];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x < 2) s_base[threadIdx.x] = 0;
  __syncthreads();
  for (int b0 = 0; b0 < B; b0 += kBlock) {
    const int b = b0 + threadIdx.x;
    const bool in = b < B;
    const bool is_short = in && lengths[b] <= 32;
    const unsigned long long ms = __ballot(is_short), ml = __ballot(in && !is_short);
    if (lane == 0) {
      s_wave[0][wave] = __popcll(ms);
      s_wave[1][wave] = __popcll(ml);
    }
    __syncthreads();
    if (in) {
      const int k = is_short ? 0 : 1;
      int off = s_base[k] + __popcll((k == 0 ? ms : ml) & ((1ull << lane) - 1ull));
      for (int v = 0; v < wave; ++v) off += s_wave[k][v];
      list[(size_t)k * B + off] = b;
    }
    __syncthreads();
    if (threadIdx.x < 2) {
      int t = 0;
      for (int v = 0; v < kBlock / 64; ++v) t += s_wave[threadIdx.x][v];
      s_base[threadIdx.x] += t;
    }
    __syncthreads();
  }
  if (threadIdx.x < 2) count[threadIdx.x] = s_base[threadIdx.x];
}

static size_t sas_bucket_bytes(int B) { return align_up((size_t)(2 * (size_t)B + 64) * sizeof(int32_t), 256); }


// The same for up to 4 length classes: class k holds the sequences with thr[k-1] < len <= thr[k] (thr ascending, the
// last one >= the longest history); list[k * B ...], count[k].
struct SasBuckets { int n; int thr[4]; };
static __global__ __launch_bounds__(kBlock) void sas_bucket_n_kernel(const int64_t* __restrict__ lengths, int B, SasBuckets bk,
                                                                     int32_t* __restrict__ list, int32_t* __restrict__ count) {
  __shared__ int s_wave[4][kBlock / 64];
  __shared__ int s_base[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x < 4) s_base[threadIdx.x] = 0;
  __syncthreads();
  for (int b0 = 0; b0 < B; b0 += kBlock) {
    const int b = b0 + threadIdx.x;
    int cls = -1;
    if (b < B) {
      const int64_t n = lengths[b];
      cls = bk.n - 1;
      for (int k = bk.n - 2; k >= 0; --k)
        if (n <= bk.thr[k]) cls = k;
    }
    unsigned long long mine = 0;
    for (int k = 0; k < bk.n; ++k) {
      const unsigned long long m = __ballot(cls == k);
      if (cls == k) mine = m;
      if (lane == 0) s_wave[k][wave] = __popcll(m);
    }
    __syncthreads();
    if (cls >= 0) {
      int off = s_base[cls] + __popcll(mine & ((1ull << lane) - 1ull));
      for (int v = 0; v < wave; ++v) off += s_wave[cls][v];
      list[(size_t)cls * B + off] = b;
    }
    __syncthreads();
    if ((int)threadIdx.x < bk.n) {
      int t = 0;
      for (int v = 0; v < kBlock / 64; ++v) t += s_wave[threadIdx.x][v];
      s_base[threadIdx.x] += t;
    }
    __syncthreads();
  }
  if ((int)threadIdx.x < bk.n) count[threadIdx.x] = s_base[threadIdx.x];
}

}  // namespace rc
