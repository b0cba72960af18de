// eval_rank.hip -- the ranking-metric half of BaseRunner.evaluate_method on the device.
//
// Reference (helpers/BaseRunner.py:52-78): gt_rank = (predictions >= predictions[:, 0]).sum(-1), i.e.
// the rank of the ground-truth item (column 0) among its candidates with ties counted AGAINST it;
// then HR@k = mean(gt_rank <= k), NDCG@k = mean((gt_rank <= k) / log2(gt_rank + 1)).
// rc_target_rank produces gt_rank (one wave per prediction row, ballot + popcount); the means over
// a handful of k are host work on the [N] int32 vector (or rc_reduce_sum on device).
//
// rc_full_catalogue_rank is the --test_all path (models/BaseModel.py:194-195 + BaseRunner.py:243-250):
// every item in [1, n_items) is a candidate and items the user already clicked are masked to -inf.
// It never materialises the [N, n_items] score matrix: a workgroup owns 32 users, keeps their vectors
// in LDS, streams the item table once through fp32 MFMA tiles and counts scores >= the target's score
// on the fly; clicked items are subtracted afterwards from the user's (sorted) clicked list.
#include "common.hpp"

#include <mutex>

namespace rc {

__global__ __launch_bounds__(kBlock) void target_rank_kernel(const float* __restrict__ pred, int64_t n, int C,
                                                             int32_t* __restrict__ rank) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
  if (row >= n) return;  // wave-uniform
  const float* p = pred + row * C;
  const float t = p[0];
  int cnt = 0;
  for (int c = lane; c < C; c += 64) cnt += (p[c] >= t) ? 1 : 0;
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) cnt += __shfl_xor(cnt, off, 64);
  if (lane == 0) rank[row] = cnt;
}

// ---- full catalogue ---------------------------------------------------------------------------------
// Workgroup = 4 waves and UT*32 users (MFMA M = 32 per user tile).  A wave takes item tiles of 32
// (MFMA N = 32): tiles w, w+4, ... of the workgroup's catalogue slice; blockIdx.y slices the catalogue
// so that the grid has >> 256 workgroups even for few users; partial counts are combined with integer
// atomics (order-independent, exact).  K assignment: lanes 0-31 feed k in [0, D/2), lanes 32-63 feed
// k in [D/2, D) -- each lane holds half of "its" user rows (registers, loaded once) and half of "its"
// item row (8 float4 loads for D = 64).  score_chain() restates the resulting summation order for the
// scalar kernels, so target scores, catalogue scores and clicked-item scores compare consistently.
template <int D>
__device__ __forceinline__ float score_chain(const float* __restrict__ a, const float* __restrict__ b) {
  float acc = 0.f;
  for (int s = 0; s < D / 2; ++s) {
    acc = fmaf(a[s], b[s], acc);
    acc = fmaf(a[D / 2 + s], b[D / 2 + s], acc);
  }
  return acc;
}

template <int D>
__global__ __launch_bounds__(kBlock) void target_score_kernel(const float* __restrict__ Uvec,
                                                              const float* __restrict__ I,
                                                              const int64_t* __restrict__ targets, int64_t N,
                                                              float* __restrict__ tscore) {
  const int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (r < N) tscore[r] = score_chain<D>(Uvec + r * D, I + targets[r] * D);
}

template <int D, int UT>
__global__ __launch_bounds__(kBlock) void full_rank_kernel(
    const float* __restrict__ Uvec /*[N, D] user (or sequence) vectors*/, const float* __restrict__ I,
    const float* __restrict__ tscore, const int64_t* __restrict__ targets, int64_t N, int64_t n_items,
    int64_t items_per_slice, int32_t* __restrict__ rank /*[N], pre-zeroed*/) {
  using f32x16 = __attribute__((__vector_size__(16 * sizeof(float)))) float;
  constexpr int H = D / 2;  // k values per lane half
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col = lane & 31, half = lane >> 5;
  const int64_t u0 = (int64_t)blockIdx.x * (32 * UT);
  // A operand: user row (u0 + 32*t + col), k in [half*H, half*H + H), resident in registers
  float a[UT][H];
#pragma unroll
  for (int t = 0; t < UT; ++t) {
    const int64_t u = u0 + 32 * t + col;
    const float4* src = reinterpret_cast<const float4*>(Uvec + (u < N ? u : 0) * D + half * H);
#pragma unroll
    for (int q = 0; q < H / 4; ++q) {
      const float4 v = (u < N) ? src[q] : make_float4(0.f, 0.f, 0.f, 0.f);
      a[t][4 * q] = v.x; a[t][4 * q + 1] = v.y; a[t][4 * q + 2] = v.z; a[t][4 * q + 3] = v.w;
    }
  }
  // accumulator slot r of tile t is user row (r&3) + 8*(r>>2) + 4*half, item column col
  float tgt[UT][16];
  int tid[UT][16];
  int cnt[UT][16];
#pragma unroll
  for (int t = 0; t < UT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int64_t u = u0 + 32 * t + (r & 3) + 8 * (r >> 2) + 4 * half;
      tgt[t][r] = (u < N) ? tscore[u] : __builtin_inff();
      tid[t][r] = (u < N) ? (int)targets[u] : -1;
      cnt[t][r] = 0;
    }
  const int64_t i_begin = 1 + (int64_t)blockIdx.y * items_per_slice;
  int64_t i_end = i_begin + items_per_slice;
  if (i_end > n_items) i_end = n_items;
  for (int64_t t0 = i_begin + (int64_t)wave * 32; t0 < i_end; t0 += 4 * 32) {
    const int64_t item = t0 + col;
    const bool valid = item < i_end;
    const float4* irow = reinterpret_cast<const float4*>(I + (valid ? item : (int64_t)0) * D + half * H);
    float b[H];
#pragma unroll
    for (int q = 0; q < H / 4; ++q) {
      const float4 v = irow[q];
      b[4 * q] = v.x; b[4 * q + 1] = v.y; b[4 * q + 2] = v.z; b[4 * q + 3] = v.w;
    }
#pragma unroll
    for (int t = 0; t < UT; ++t) {
      f32x16 acc = {0};
#pragma unroll
      for (int s = 0; s < H; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t][s], b[s], acc, 0, 0, 0);
      if (valid) {
        const int it = (int)item;
#pragma unroll
        for (int r = 0; r < 16; ++r) cnt[t][r] += (acc[r] >= tgt[t][r] && it != tid[t][r]) ? 1 : 0;
      }
    }
  }
  // sum over the 32 item columns (lanes of one half), then one integer atomic per user per wave
#pragma unroll
  for (int t = 0; t < UT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      int c = cnt[t][r];
#pragma unroll
      for (int off = 16; off >= 1; off >>= 1) c += __shfl_xor(c, off, 64);
      const int64_t u = u0 + 32 * t + (r & 3) + 8 * (r >> 2) + 4 * half;
      if (col == 0 && c != 0 && u < N) atomicAdd(&rank[u], c);
    }
}

// rank = 1 (the target, column 0 of the reference's candidate list) + catalogue count - clicked items
// that scored >= target (masked to -inf by BaseRunner.predict :243-250); the target's own catalogue
// column was already skipped by full_rank_kernel
template <int D>
__global__ __launch_bounds__(kBlock) void full_rank_finish_kernel(
    const float* __restrict__ Uvec, const float* __restrict__ I, const float* __restrict__ tscore,
    const int64_t* __restrict__ users, const int64_t* __restrict__ targets, int64_t N, int64_t n_items,
    const int64_t* __restrict__ clicked_ptr, const int64_t* __restrict__ clicked_items,
    int32_t* __restrict__ rank) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
  if (row >= N) return;  // wave-uniform
  int sub = 0;
  if (clicked_ptr) {
    const int64_t u = users[row];
    const float t = tscore[row];
    const int64_t tgt_item = targets[row];
    int64_t prev = -1;
    for (int64_t j = clicked_ptr[u] + lane; j < clicked_ptr[u + 1]; j += 64) {
      const int64_t it = clicked_items[j];
      // the list is sorted: a repeated id is masked once
      prev = (j > clicked_ptr[u]) ? clicked_items[j - 1] : -1;
      if (it == tgt_item || it < 1 || it >= n_items || it == prev) continue;
      sub += (score_chain<D>(Uvec + row * D, I + it * D) >= t) ? 1 : 0;
    }
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) sub += __shfl_xor(sub, off, 64);
  if (lane == 0) rank[row] = 1 + rank[row] - sub;
}

}  // namespace rc

// ---- impression lists: HR / NDCG / MAP @k with a variable number of positives and negatives per row ----------------------
// Reference: ImpressionRunner.evaluate + evaluate_method + HR_at_k / NDCG_at_k / AP_at_k (helpers/ImpressionRunner.py:18-66,
// 74-133,135-168).  A row holds its positives in columns [0, pos_num) and its negatives in [max_pos, max_pos + neg_num); the
// reference masks everything else to -inf, subtracts 1e-6 from the positive columns (so a positive that ties with a negative
// ranks below it, :89-96), argsorts -score with a stable merge sort, truncates the ranked 0/1 labels to pos + neg entries and
// forms the metrics in float64.  Only the positions of the POSITIVES in that order matter:
//   r_i   = #{valid j : key_j > key_i} + #{valid j < i : key_j == key_i}          (key = (double) score - 1e-6 [column < max_pos])
//   HR@k  = [min_i r_i < k];   DCG@k = sum_{r_i < k} 1 / log2(r_i + 2);   IDCG@k = sum_{t < min(p, k)} 1 / log2(t + 2)
//   AP@k  = sum_{r_i < k} #{i' : r_i' <= r_i} / (r_i + 1)  /  clip(p, 1, k)           (p = number of valid positives)
// One wave per row: keys of the row in LDS, a positive's rank by a strided count + wave sum, the per-k sums in float64.
namespace rc {

constexpr int kListMetricsMaxK = 16;
constexpr int kListMetricsMaxN = 2048;   // columns per row the LDS staging holds (4 waves x 2048 doubles + ranks = 80 KB)

struct ListMetricArgs {
  const float* pred;        // [N, n]
  const int64_t* pos_num;   // [N] | null (one positive per row)
  const int64_t* neg_num;   // [N]
  int64_t N;
  int n, max_pos, n_k;
  int topk[kListMetricsMaxK];
  double* out;              // [N, 3, n_k]: NDCG, MAP, HR
};

__device__ __forceinline__ double wave_sum_f64(double x) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) x += __shfl_xor(x, off, 64);
  return x;
}
__device__ __forceinline__ int wave_sum_i32(int x) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) x += __shfl_xor(x, off, 64);
  return x;
}

__global__ __launch_bounds__(kBlock) void list_metrics_kernel(ListMetricArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char list_metrics_smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double* key = reinterpret_cast<double*>(list_metrics_smem) + (size_t)wave * a.n;
  int* rnk = reinterpret_cast<int*>(reinterpret_cast<double*>(list_metrics_smem) + (size_t)(kBlock / 64) * a.n) + (size_t)wave * 2 * a.max_pos;
  int* cnt = rnk + a.max_pos;
  const int64_t row = (int64_t)blockIdx.x * (kBlock / 64) + wave;
  if (row >= a.N) return;   // wave-uniform; no workgroup barrier below
  const int64_t pn = a.pos_num ? a.pos_num[row] : 1;
  const int64_t nn = a.neg_num[row];
  const int p = (int)(pn < 0 ? 0 : (pn > a.max_pos ? a.max_pos : pn));                      // (:99-102)
  const int q = (int)(nn < 0 ? 0 : (nn > a.n - a.max_pos ? a.n - a.max_pos : nn));
  const float* x = a.pred + row * a.n;
  for (int c = lane; c < a.n; c += 64) {
    const bool valid = c < p || (c >= a.max_pos && c < a.max_pos + q);
    key[c] = valid ? (double)x[c] - (c < a.max_pos ? 1e-6 : 0.0) : -__builtin_inf();
  }
  __builtin_amdgcn_wave_barrier();   // (ds operations of one wave complete in order)
  // ranks of the positives among the valid entries
  for (int i = 0; i < p; ++i) {
    const double ki = key[i];
    int c = 0;
    for (int j = lane; j < a.n; j += 64) {
      const bool valid = j < p || (j >= a.max_pos && j < a.max_pos + q);
      const double kj = key[j];
      c += (valid && (kj > ki || (kj == ki && j < i))) ? 1 : 0;
    }
    c = wave_sum_i32(c);
    if (lane == 0) rnk[i] = c;
  }
  __builtin_amdgcn_wave_barrier();
  for (int i = lane; i < p; i += 64) {     // positives ranked at or before positive i (itself included)
    int c = 0;
    const int ri = rnk[i];
    for (int t = 0; t < p; ++t) c += rnk[t] <= ri ? 1 : 0;
    cnt[i] = c;
  }
  __builtin_amdgcn_wave_barrier();
  double* out = a.out + row * 3 * a.n_k;
  for (int kk = 0; kk < a.n_k; ++kk) {
    const int k = a.topk[kk];
    double dcg = 0.0, ap = 0.0, idcg = 0.0;
    int hit = 0;
    for (int i = lane; i < p; i += 64) {
      const int ri = rnk[i];
      if (ri < k) {
        dcg += 1.0 / log2((double)(ri + 2));
        ap += (double)cnt[i] / (double)(ri + 1);
        hit = 1;
      }
    }
    const int ideal = p < k ? p : k;
    for (int t = lane; t < ideal; t += 64) idcg += 1.0 / log2((double)(t + 2));
    dcg = wave_sum_f64(dcg); ap = wave_sum_f64(ap); idcg = wave_sum_f64(idcg);
    hit = wave_sum_i32(hit);
    if (lane == 0) {
      const int cap = p < 1 ? 1 : (p > k ? k : p);
      out[0 * a.n_k + kk] = dcg / (idcg == 0.0 ? 1.0 : idcg);
      out[1 * a.n_k + kk] = ap / (double)cap;
      out[2 * a.n_k + kk] = hit > 0 ? 1.0 : 0.0;
    }
  }
}

// column means of per-row metrics [N, w] -> [w]: one workgroup per column, a fixed tree over fixed strides
__global__ __launch_bounds__(kBlock) void list_metrics_mean_kernel(const double* __restrict__ per_row, int64_t N, int w, double* __restrict__ mean) {
  __shared__ double part[kBlock / 64];
  const int col = blockIdx.x;
  double s = 0.0;
  for (int64_t r = threadIdx.x; r < N; r += kBlock) s += per_row[r * w + col];
  s = wave_sum_f64(s);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int q = 0; q < kBlock / 64; ++q) t += part[q];
    mean[col] = N > 0 ? t / (double)N : 0.0;
  }
}

}  // namespace rc

using namespace rc;

extern "C" int rc_target_rank(const float* pred, int64_t n, int C, int32_t* rank, rc_stream_t stream) {
  if (n == 0) return RC_OK;
  RC_REQUIRE(pred && rank, "rc_target_rank: null pointer");
  RC_REQUIRE(n > 0 && C >= 1, "rc_target_rank: bad shape n=%lld C=%d", (long long)n, C);
  const int64_t blocks = (n + (kBlock / 64) - 1) / (kBlock / 64);
  RC_REQUIRE(blocks <= kMaxGridX, "rc_target_rank: too many rows");
  hipLaunchKernelGGL(target_rank_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, as_stream(stream), pred, n, C, rank);
  RC_LAUNCH_CHECK();
  return RC_OK;
}

template <int D>
static int launch_full_rank(const float* Uvec, const float* I, const int64_t* users, const int64_t* targets, int64_t N,
                            int64_t n_items, const int64_t* clicked_ptr, const int64_t* clicked_items, float* tscore,
                            int32_t* rank, hipStream_t s) {
  constexpr int UT = 2;
  RC_HIP(hipMemsetAsync(rank, 0, (size_t)N * sizeof(int32_t), s));
  hipLaunchKernelGGL((target_score_kernel<D>), dim3((unsigned)((N + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, Uvec, I,
                     targets, N, tscore);
  RC_LAUNCH_CHECK();
  const int64_t user_blocks = (N + 32 * UT - 1) / (32 * UT);
  // slice the catalogue until the grid has ~2048 workgroups; a slice is a multiple of the 128-item stride
  int64_t slices = (2048 + user_blocks - 1) / user_blocks;
  const int64_t max_slices = (n_items + 511) / 512;
  if (slices > max_slices) slices = max_slices;
  if (slices < 1) slices = 1;
  if (slices > 65535) slices = 65535;
  int64_t per = (n_items - 1 + slices - 1) / slices;
  per = (per + 127) / 128 * 128;
  slices = (n_items - 1 + per - 1) / per;
  if (slices < 1) slices = 1;
  RC_REQUIRE(user_blocks <= kMaxGridX, "rc_full_catalogue_rank: too many rows");
  hipLaunchKernelGGL((full_rank_kernel<D, UT>), dim3((unsigned)user_blocks, (unsigned)slices), dim3(kBlock), 0, s, Uvec,
                     I, tscore, targets, N, n_items, per, rank);
  RC_LAUNCH_CHECK();
  hipLaunchKernelGGL((full_rank_finish_kernel<D>), dim3((unsigned)((N + 3) / 4)), dim3(kBlock), 0, s, Uvec, I, tscore,
                     users, targets, N, n_items, clicked_ptr, clicked_items, rank);
  RC_LAUNCH_CHECK();
  return RC_OK;
}

extern "C" int rc_full_catalogue_rank_supported(int d) { return d == 32 || d == 64 || d == 128; }

extern "C" int rc_full_catalogue_rank(const float* Uvec, const float* I, const int64_t* users, const int64_t* targets,
                                      int64_t N, int64_t n_items, int d, const int64_t* clicked_ptr,
                                      const int64_t* clicked_items, float* target_score, int32_t* rank,
                                      rc_stream_t stream) {
  if (N == 0) return RC_OK;
  RC_REQUIRE(Uvec && I && targets && target_score && rank, "rc_full_catalogue_rank: null pointer");
  RC_REQUIRE((clicked_ptr == nullptr) == (clicked_items == nullptr) && (clicked_ptr == nullptr || users),
             "rc_full_catalogue_rank: clicked_ptr, clicked_items and users go together");
  RC_REQUIRE(N > 0 && n_items >= 2 && n_items < ((int64_t)1 << 31), "rc_full_catalogue_rank: bad shape N=%lld n_items=%lld",
             (long long)N, (long long)n_items);
  RC_REQUIRE(reinterpret_cast<uintptr_t>(Uvec) % 16 == 0 && reinterpret_cast<uintptr_t>(I) % 16 == 0,
             "rc_full_catalogue_rank: tables must be 16-byte aligned");
  hipStream_t s = as_stream(stream);
  switch (d) {
    case 32: return launch_full_rank<32>(Uvec, I, users, targets, N, n_items, clicked_ptr, clicked_items, target_score, rank, s);
    case 64: return launch_full_rank<64>(Uvec, I, users, targets, N, n_items, clicked_ptr, clicked_items, target_score, rank, s);
    case 128: return launch_full_rank<128>(Uvec, I, users, targets, N, n_items, clicked_ptr, clicked_items, target_score, rank, s);
    default: return fail(RC_ERR_UNSUPPORTED, "rc_full_catalogue_rank: emb_size must be 32, 64 or 128, got %d", d);
  }
}

extern "C" int rc_list_metrics_supported(int n, int max_pos, int n_k) {
  return (n >= 1 && n <= kListMetricsMaxN && max_pos >= 0 && max_pos <= n && n_k >= 1 && n_k <= kListMetricsMaxK) ? 1 : 0;
}

extern "C" int rc_list_metrics(const float* pred, const int64_t* pos_num, const int64_t* neg_num, int64_t N, int n, int max_pos,
                               const int* topk, int n_k, double* per_row, double* mean, rc_stream_t stream) {
  RC_REQUIRE(topk && (N == 0 || (pred && neg_num && per_row)), "rc_list_metrics: null pointer");
  if (!rc_list_metrics_supported(n, max_pos, n_k))
    return fail(RC_ERR_UNSUPPORTED, "rc_list_metrics: n=%d (<= %d), max_pos=%d, %d values of k (<= %d) not covered", n, kListMetricsMaxN, max_pos,
                n_k, kListMetricsMaxK);
  RC_REQUIRE(N >= 0, "rc_list_metrics: N=%lld", (long long)N);
  hipStream_t s = as_stream(stream);
  ListMetricArgs a;
  memset(&a, 0, sizeof(a));
  a.pred = pred; a.pos_num = pos_num; a.neg_num = neg_num; a.N = N; a.n = n; a.max_pos = max_pos; a.n_k = n_k; a.out = per_row;
  for (int i = 0; i < n_k; ++i) {
    RC_REQUIRE(topk[i] >= 1, "rc_list_metrics: k = %d", topk[i]);
    a.topk[i] = topk[i];
  }
  if (N > 0) {
    const size_t lds = (size_t)(kBlock / 64) * ((size_t)n * sizeof(double) + 2 * (size_t)max_pos * sizeof(int));
    {   // more than 64 KB of dynamic LDS needs the attribute, once per device of the process
      static std::mutex mu;
      static bool done[64] = {};
      int dev = 0;
      RC_HIP(hipGetDevice(&dev));
      std::lock_guard<std::mutex> lock(mu);
      if (dev < 0 || dev >= 64 || !done[dev]) {
        RC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(list_metrics_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)((kBlock / 64) * ((size_t)kListMetricsMaxN * (sizeof(double) + 2 * sizeof(int))))));
        if (dev >= 0 && dev < 64) done[dev] = true;
      }
    }
    const int64_t blocks = (N + kBlock / 64 - 1) / (kBlock / 64);
    RC_REQUIRE(blocks <= kMaxGridX, "rc_list_metrics: too many rows");
    hipLaunchKernelGGL(list_metrics_kernel, dim3((unsigned)blocks), dim3(kBlock), lds, s, a);
    RC_LAUNCH_CHECK();
  }
  if (mean) {
    hipLaunchKernelGGL(list_metrics_mean_kernel, dim3((unsigned)(3 * n_k)), dim3(kBlock), 0, s, per_row, N, 3 * n_k, mean);
    RC_LAUNCH_CHECK();
  }
  return RC_OK;
}
