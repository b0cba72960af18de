// seg_update.hip -- atomic-free segmented scatter of embedding-row gradients, fused with
// the optimizer's row update.
//
// Reference semantics being replaced: autograd's EmbeddingBackward ->
// aten::embedding_dense_backward (zero-fill a dense [n_rows,d] grad, index_add every
// occurrence) followed by torch.optim's step over the table
// (helpers/BaseRunner.py:193,205-206; optimizer built at :110-114).
//
// Input is the stably sorted id list of the batch (sort_ids.hip).  The lane-group at
// sorted position j owns row keys[j] iff j is the head of its segment; it sums the
// segment's per-occurrence gradient rows in ascending j (fixed order -> bit-reproducible,
// no float atomics), then reads the table row once, applies the update, writes it once.
// The per-occurrence gradient row is rebuilt on the fly as coef[o] * Src[srow(o)]
// (for BPRMF items: g[b,c] * U[uid[b]]), so it never exists in HBM.
// Segments longer than kLongSeg (hot Zipf rows) are deferred to a workgroup-per-row
// kernel that strides 256/LPR lane-groups over the segment and combines their partials in
// a fixed LDS tree.  The only atomics are integer appends to the deferred-row list.
#include "common.hpp"
#include "opt_math.hpp"

namespace rc {

constexpr int kLongSeg = 32;  // occurrences handled sequentially by one lane-group

struct SegArgs {
  float* W;
  float* M;
  float* V;
  const uint32_t* keys;
  const uint32_t* perm;
  int64_t n_occ;
  const float* coef;
  const float* src;
  const int64_t* src_index;
  int div;
  int d;  // generic kernels only
  float* dense_grad;
  uint32_t* long_list;
  uint32_t* n_long;
  uint32_t long_cap;
  int skip_single;  // 1: single-occurrence rows were already updated by the fused kernel
  OptScalars o;
};

// apply the reduced gradient g to row `key` (w = current row slice, already loaded)
template <int D, int MODE>
__device__ __forceinline__ void apply_row4(const SegArgs& a, uint32_t key, int l, float4 w,
                                           const float4& g) {
  constexpr int LPR = D / 4;
  const size_t idx = (size_t)key * LPR + l;
  if (MODE == MODE_DENSE_GRAD) {
    reinterpret_cast<float4*>(a.dense_grad)[idx] = g;
    return;
  }
  opt_row4<MODE>(a.o, a.W, a.M, a.V, idx, w, g);
}

template <int D, int MODE>
__device__ __forceinline__ float4 load_row4(const SegArgs& a, uint32_t key, int l) {
  constexpr int LPR = D / 4;
  if (MODE == MODE_DENSE_GRAD) return make_float4(0, 0, 0, 0);
  return reinterpret_cast<const float4*>(a.W)[(size_t)key * LPR + l];
}

// gradient row of occurrence o (= perm[j]), lane's float4
template <int D>
__device__ __forceinline__ float4 occ_grad4_o(const SegArgs& a, uint32_t o, int l) {
  constexpr int LPR = D / 4;
  const float c = a.coef ? a.coef[o] : 1.0f;
  int64_t sr = (a.div == 1) ? (int64_t)o : (int64_t)(o / (uint32_t)a.div);
  if (a.src_index) sr = a.src_index[sr];
  float4 s = reinterpret_cast<const float4*>(a.src)[(size_t)sr * LPR + l];
  s.x *= c; s.y *= c; s.z *= c; s.w *= c;
  return s;
}
template <int D>
__device__ __forceinline__ float4 occ_grad4(const SegArgs& a, int64_t jj, int l) {
  return occ_grad4_o<D>(a, a.perm[jj], l);
}

__device__ __forceinline__ void add4(float4& x, const float4& y) {
  x.x += y.x; x.y += y.y; x.z += y.z; x.w += y.w;
}

// SKIP = true: single-occurrence rows were updated upstream (fused BPRMF kernel), every
// surviving head has >= 2 occurrences, so the first two gradient rows are fetched together.
// Dependent-load depth per row: {keys[j-1..j+1], perm[j], perm[j+1]} -> {W row, coef, index}
// -> {src rows} -> store.
template <int D, int MODE, bool SKIP>
__global__ __launch_bounds__(kBlock) void seg_update_kernel(SegArgs a) {
  constexpr int LPR = D / 4;
  constexpr int GPB = kBlock / LPR;
  const int l = threadIdx.x % LPR;
  const int64_t n = a.n_occ;
  const int64_t j = (int64_t)blockIdx.x * GPB + threadIdx.x / LPR;
  if (j >= n) return;  // no cross-lane ops in this kernel
  const bool has_next = j + 1 < n;
  const uint32_t key = a.keys[j];
  const uint32_t kprev = j > 0 ? a.keys[j - 1] : ~key;
  const uint32_t knext = has_next ? a.keys[j + 1] : ~key;
  const uint32_t o0 = a.perm[j];
  const uint32_t o1 = a.perm[has_next ? j + 1 : j];
  if (kprev == key) return;  // not a segment head
  const bool multi = knext == key;
  if (SKIP && !multi) return;  // singleton: already updated by the fused kernel
  const float4 w = load_row4<D, MODE>(a, key, l);
  float4 acc = occ_grad4_o<D>(a, o0, l);
  if (SKIP || multi) {
    const float4 s1 = occ_grad4_o<D>(a, o1, l);
    add4(acc, s1);
    int64_t jj = j + 2;
    while (jj < n && a.keys[jj] == key) {
      if (jj - j >= kLongSeg) {  // hot row: hand over to the workgroup-per-row kernel
        if (l == 0) {
          const uint32_t slot = atomicAdd(a.n_long, 1u);
          if (slot < a.long_cap) a.long_list[slot] = (uint32_t)j;
        }
        return;
      }
      const float4 s = occ_grad4<D>(a, jj, l);
      add4(acc, s);
      ++jj;
    }
  }
  apply_row4<D, MODE>(a, key, l, w, acc);
}

// end of the segment that starts at j0 (first index with a different key)
__device__ __forceinline__ int64_t segment_end(const uint32_t* __restrict__ keys, int64_t n,
                                               int64_t j0) {
  const uint32_t key = keys[j0];
  int64_t lo = j0;  // keys[lo] == key
  int64_t step = kLongSeg;
  int64_t hi = j0 + step;
  while (hi < n && keys[hi] == key) {
    lo = hi;
    step <<= 1;
    hi = lo + step;
  }
  if (hi > n) hi = n;  // keys[hi] != key or hi == n
  while (hi - lo > 1) {
    const int64_t mid = lo + (hi - lo) / 2;
    if (keys[mid] == key) lo = mid; else hi = mid;
  }
  return hi;
}

template <int D, int MODE>
__global__ __launch_bounds__(kBlock) void seg_update_long_kernel(SegArgs a) {
  constexpr int LPR = D / 4;
  constexpr int GPB = kBlock / LPR;
  __shared__ float4 part[kBlock];
  __shared__ int64_t s_end;
  const int l = threadIdx.x % LPR;
  const int g = threadIdx.x / LPR;
  uint32_t n_long = *a.n_long;
  if (n_long > a.long_cap) n_long = a.long_cap;
  for (uint32_t i = blockIdx.x; i < n_long; i += gridDim.x) {
    const int64_t j0 = a.long_list[i];
    if (threadIdx.x == 0) s_end = segment_end(a.keys, a.n_occ, j0);
    __syncthreads();
    const int64_t end = s_end;
    // four independent occurrences per lane-group per trip (fixed pattern -> fixed order)
    float4 acc = make_float4(0, 0, 0, 0);
    for (int64_t jj = j0 + g; jj < end; jj += 4 * GPB) {
      const float4 z = make_float4(0, 0, 0, 0);
      const float4 s0 = occ_grad4<D>(a, jj, l);
      const float4 s1 = (jj + GPB < end) ? occ_grad4<D>(a, jj + GPB, l) : z;
      const float4 s2 = (jj + 2 * GPB < end) ? occ_grad4<D>(a, jj + 2 * GPB, l) : z;
      const float4 s3 = (jj + 3 * GPB < end) ? occ_grad4<D>(a, jj + 3 * GPB, l) : z;
      add4(acc, s0); add4(acc, s1); add4(acc, s2); add4(acc, s3);
    }
    part[threadIdx.x] = acc;
    __syncthreads();
    for (int off = GPB / 2; off >= 1; off >>= 1) {
      if (g < off) {
        float4 x = part[threadIdx.x];
        const float4 y = part[threadIdx.x + off * LPR];
        x.x += y.x; x.y += y.y; x.z += y.z; x.w += y.w;
        part[threadIdx.x] = x;
      }
      __syncthreads();
    }
    if (g == 0) {
      const uint32_t key = a.keys[j0];
      apply_row4<D, MODE>(a, key, l, load_row4<D, MODE>(a, key, l), part[threadIdx.x]);
    }
    __syncthreads();  // part[] and s_end are reused by the next row
  }
}

// ---- any d (<= 512): one wave per sorted position, lanes stride over the row ----------
constexpr int kGenChunks = 8;

template <int MODE>
__device__ __forceinline__ void apply_row_generic(const SegArgs& a, uint32_t key, int lane,
                                                  const float* acc) {
#pragma unroll
  for (int q = 0; q < kGenChunks; ++q) {
    const int k = lane + 64 * q;
    if (k >= a.d) continue;
    const size_t idx = (size_t)key * a.d + k;
    if (MODE == MODE_DENSE_GRAD) {
      a.dense_grad[idx] = acc[q];
      continue;
    }
    float w = a.W[idx], m = 0.f, v = 0.f;
    if (MODE == MODE_ADAM || MODE == MODE_ADAGRAD) m = a.M[idx];
    if (MODE == MODE_ADAM) v = a.V[idx];
    opt_elem<MODE>(a.o, acc[q], w, m, v);
    a.W[idx] = w;
    if (MODE == MODE_ADAM || MODE == MODE_ADAGRAD) a.M[idx] = m;
    if (MODE == MODE_ADAM) a.V[idx] = v;
  }
}

__device__ __forceinline__ void occ_grad_generic(const SegArgs& a, int64_t jj, int lane,
                                                 float* acc) {
  const uint32_t o = a.perm[jj];
  const float c = a.coef ? a.coef[o] : 1.0f;
  int64_t sr = (a.div == 1) ? (int64_t)o : (int64_t)(o / (uint32_t)a.div);
  if (a.src_index) sr = a.src_index[sr];
  const float* s = a.src + (size_t)sr * a.d;
#pragma unroll
  for (int q = 0; q < kGenChunks; ++q) {
    const int k = lane + 64 * q;
    if (k < a.d) acc[q] += c * s[k];
  }
}

// generic path handles segments of any length sequentially (correctness fall-back)
template <int MODE>
__global__ __launch_bounds__(kBlock) void seg_update_generic_kernel(SegArgs a) {
  const int lane = threadIdx.x & 63;
  const int64_t j = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
  if (j >= a.n_occ) return;
  const uint32_t key = a.keys[j];
  if (j > 0 && a.keys[j - 1] == key) return;
  if (a.skip_single && !(j + 1 < a.n_occ && a.keys[j + 1] == key)) return;
  float acc[kGenChunks];
#pragma unroll
  for (int q = 0; q < kGenChunks; ++q) acc[q] = 0.f;
  for (int64_t jj = j; jj < a.n_occ && a.keys[jj] == key; ++jj) occ_grad_generic(a, jj, lane, acc);
  apply_row_generic<MODE>(a, key, lane, acc);
}

template <int D, int MODE>
static int launch_seg(const SegArgs& a, hipStream_t s) {
  constexpr int GPB = kBlock / (D / 4);
  const int64_t blocks = (a.n_occ + GPB - 1) / GPB;
  if (blocks > kMaxGridX) return fail(RC_ERR_UNSUPPORTED, "seg_update: grid too large");
  RC_HIP(hipMemsetAsync(a.n_long, 0, sizeof(uint32_t), s));
  if (a.skip_single)
    hipLaunchKernelGGL((seg_update_kernel<D, MODE, true>), dim3((unsigned)blocks), dim3(kBlock), 0, s, a);
  else
    hipLaunchKernelGGL((seg_update_kernel<D, MODE, false>), dim3((unsigned)blocks), dim3(kBlock), 0, s, a);
  RC_LAUNCH_CHECK();
  unsigned lblocks = a.long_cap < 1024u ? a.long_cap : 1024u;
  if (lblocks == 0) lblocks = 1;
  hipLaunchKernelGGL((seg_update_long_kernel<D, MODE>), dim3(lblocks), dim3(kBlock), 0, s, a);
  RC_LAUNCH_CHECK();
  return RC_OK;
}

template <int MODE>
static int launch_seg_mode(const SegArgs& a, bool aligned, hipStream_t s) {
  if (aligned) {
    switch (a.d) {
      case 16: return launch_seg<16, MODE>(a, s);
      case 32: return launch_seg<32, MODE>(a, s);
      case 64: return launch_seg<64, MODE>(a, s);
      case 128: return launch_seg<128, MODE>(a, s);
      case 256: return launch_seg<256, MODE>(a, s);
      default: break;
    }
  }
  if (a.d > 64 * kGenChunks)
    return fail(RC_ERR_UNSUPPORTED, "rc_segmented_update: d=%d > %d", a.d, 64 * kGenChunks);
  const int64_t blocks = (a.n_occ + (kBlock / 64) - 1) / (kBlock / 64);
  if (blocks > kMaxGridX) return fail(RC_ERR_UNSUPPORTED, "seg_update: grid too large");
  hipLaunchKernelGGL((seg_update_generic_kernel<MODE>), dim3((unsigned)blocks), dim3(kBlock), 0, s,
                     a);
  RC_LAUNCH_CHECK();
  return RC_OK;
}

// ---- dense (exact torch semantics) optimizer step over a whole tensor -------------------
template <int MODE>
__global__ __launch_bounds__(kBlock) void dense_update_kernel(SegArgs a, const float* __restrict__ G,
                                                              int64_t n) {
  const int64_t n4 = n / 4;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n4;
       i += (int64_t)gridDim.x * kBlock) {
    const float4 g = reinterpret_cast<const float4*>(G)[i];
    float4 w = reinterpret_cast<const float4*>(a.W)[i];
    float4 m = make_float4(0, 0, 0, 0), v = make_float4(0, 0, 0, 0);
    if (MODE == MODE_ADAM || MODE == MODE_ADAGRAD) m = reinterpret_cast<const float4*>(a.M)[i];
    if (MODE == MODE_ADAM) v = reinterpret_cast<const float4*>(a.V)[i];
    opt_elem<MODE>(a.o, g.x, w.x, m.x, v.x);
    opt_elem<MODE>(a.o, g.y, w.y, m.y, v.y);
    opt_elem<MODE>(a.o, g.z, w.z, m.z, v.z);
    opt_elem<MODE>(a.o, g.w, w.w, m.w, v.w);
    reinterpret_cast<float4*>(a.W)[i] = w;
    if (MODE == MODE_ADAM || MODE == MODE_ADAGRAD) reinterpret_cast<float4*>(a.M)[i] = m;
    if (MODE == MODE_ADAM) reinterpret_cast<float4*>(a.V)[i] = v;
  }
  // tail (n % 4 elements)
  const int64_t i = n4 * 4 + (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i < n) {
    float w = a.W[i], m = 0.f, v = 0.f;
    if (MODE == MODE_ADAM || MODE == MODE_ADAGRAD) m = a.M[i];
    if (MODE == MODE_ADAM) v = a.V[i];
    opt_elem<MODE>(a.o, G[i], w, m, v);
    a.W[i] = w;
    if (MODE == MODE_ADAM || MODE == MODE_ADAGRAD) a.M[i] = m;
    if (MODE == MODE_ADAM) a.V[i] = v;
  }
}

template <int MODE>
__global__ __launch_bounds__(kBlock) void dense_update_scalar_kernel(SegArgs a,
                                                                     const float* __restrict__ G,
                                                                     int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * kBlock) {
    float w = a.W[i], m = 0.f, v = 0.f;
    if (MODE == MODE_ADAM || MODE == MODE_ADAGRAD) m = a.M[i];
    if (MODE == MODE_ADAM) v = a.V[i];
    opt_elem<MODE>(a.o, G[i], w, m, v);
    a.W[i] = w;
    if (MODE == MODE_ADAM || MODE == MODE_ADAGRAD) a.M[i] = m;
    if (MODE == MODE_ADAM) a.V[i] = v;
  }
}

template <int MODE>
static int launch_dense(const SegArgs& a, const float* G, int64_t n, bool aligned, hipStream_t s) {
  int64_t work = aligned ? (n / 4 > 0 ? n / 4 : 1) : n;
  int64_t blocks = (work + kBlock - 1) / kBlock;
  if (blocks > 256 * 16) blocks = 256 * 16;
  if (blocks < 1) blocks = 1;
  if (aligned)
    hipLaunchKernelGGL((dense_update_kernel<MODE>), dim3((unsigned)blocks), dim3(kBlock), 0, s, a, G, n);
  else
    hipLaunchKernelGGL((dense_update_scalar_kernel<MODE>), dim3((unsigned)blocks), dim3(kBlock), 0, s, a, G, n);
  RC_LAUNCH_CHECK();
  return RC_OK;
}

}  // namespace rc

using namespace rc;

// flag[o] = 1 iff occurrence o is the only one of its row in the batch
__global__ __launch_bounds__(rc::kBlock) void mark_singletons_kernel(
    const uint32_t* __restrict__ keys, const uint32_t* __restrict__ perm, int64_t n,
    uint8_t* __restrict__ flag) {
  for (int64_t j = (int64_t)blockIdx.x * rc::kBlock + threadIdx.x; j < n;
       j += (int64_t)gridDim.x * rc::kBlock) {
    const uint32_t k = keys[j];
    const bool single = (j == 0 || keys[j - 1] != k) && (j + 1 >= n || keys[j + 1] != k);
    flag[perm[j]] = single ? 1 : 0;
  }
}

extern "C" int rc_mark_singletons(const uint32_t* keys, const uint32_t* perm, int64_t n_occ,
                                  uint8_t* flag, rc_stream_t stream) {
  if (n_occ == 0) return RC_OK;
  RC_REQUIRE(keys && perm && flag, "rc_mark_singletons: null pointer");
  RC_REQUIRE(n_occ > 0 && n_occ < ((int64_t)1 << 31), "rc_mark_singletons: bad n_occ");
  int64_t blocks = (n_occ + kBlock - 1) / kBlock;
  if (blocks > 256 * 16) blocks = 256 * 16;
  hipLaunchKernelGGL(mark_singletons_kernel, dim3((unsigned)blocks), dim3(kBlock), 0,
                     as_stream(stream), keys, perm, n_occ, flag);
  RC_LAUNCH_CHECK();
  return RC_OK;
}

extern "C" size_t rc_segmented_workspace_bytes(int64_t n_occ) {
  if (n_occ < 1) n_occ = 1;
  const size_t cap = (size_t)(n_occ / kLongSeg) + 1;
  return align_up(cap * sizeof(uint32_t), 256) + 256;
}

extern "C" int rc_segmented_update(float* W, float* m, float* v, int d, const uint32_t* keys,
                                   const uint32_t* perm, int64_t n_occ, const float* coef,
                                   const float* src, const int64_t* src_index, int div,
                                   const rc_opt_hyper* h, float* dense_grad, int flags, void* ws,
                                   size_t ws_bytes, rc_stream_t stream) {
  if (n_occ == 0) return RC_OK;
  RC_REQUIRE(keys && perm && src && ws, "rc_segmented_update: null pointer");
  RC_REQUIRE(d >= 1 && div >= 1 && n_occ >= 0 && n_occ < ((int64_t)1 << 31),
             "rc_segmented_update: bad shape d=%d div=%d n_occ=%lld", d, div, (long long)n_occ);
  RC_REQUIRE(dense_grad != nullptr || W != nullptr, "rc_segmented_update: no output (W or dense_grad)");
  if (n_occ == 0) return RC_OK;
  if (ws_bytes < rc_segmented_workspace_bytes(n_occ))
    return fail(RC_ERR_WORKSPACE, "rc_segmented_update: workspace %zu < %zu", ws_bytes,
                rc_segmented_workspace_bytes(n_occ));
  SegArgs a;
  memset(&a, 0, sizeof(a));
  a.W = W; a.M = m; a.V = v;
  a.keys = keys; a.perm = perm; a.n_occ = n_occ;
  a.coef = coef; a.src = src; a.src_index = src_index; a.div = div; a.d = d;
  a.dense_grad = dense_grad;
  a.skip_single = (flags & RC_SEG_SKIP_SINGLETONS) ? 1 : 0;
  Carver cv(ws);
  a.n_long = cv.take<uint32_t>(1);
  a.long_cap = (uint32_t)(n_occ / kLongSeg) + 1;
  a.long_list = cv.take<uint32_t>(a.long_cap);
  hipStream_t s = as_stream(stream);
  bool aligned = (d % 4 == 0) && (reinterpret_cast<uintptr_t>(src) % 16 == 0);
  if (dense_grad) {
    aligned = aligned && (reinterpret_cast<uintptr_t>(dense_grad) % 16 == 0);
    return launch_seg_mode<MODE_DENSE_GRAD>(a, aligned, s);
  }
  RC_TRY(fill_opt_scalars(h, &a.o));
  aligned = aligned && (reinterpret_cast<uintptr_t>(W) % 16 == 0);
  switch (h->opt) {
    case RC_OPT_SGD:
      return launch_seg_mode<MODE_SGD>(a, aligned, s);
    case RC_OPT_ADAM:
      RC_REQUIRE(m && v, "rc_segmented_update: Adam needs m and v");
      aligned = aligned && (reinterpret_cast<uintptr_t>(m) % 16 == 0) &&
                (reinterpret_cast<uintptr_t>(v) % 16 == 0);
      return launch_seg_mode<MODE_ADAM>(a, aligned, s);
    case RC_OPT_ADAGRAD:
      RC_REQUIRE(m, "rc_segmented_update: Adagrad needs m (state_sum)");
      aligned = aligned && (reinterpret_cast<uintptr_t>(m) % 16 == 0);
      return launch_seg_mode<MODE_ADAGRAD>(a, aligned, s);
    default:
      return fail(RC_ERR_INVALID_ARG, "rc_segmented_update: unknown optimizer %d", h->opt);
  }
}

extern "C" int rc_dense_update(float* W, const float* G, float* m, float* v, int64_t n,
                               const rc_opt_hyper* h, rc_stream_t stream) {
  if (n == 0) return RC_OK;
  RC_REQUIRE(W && G, "rc_dense_update: null pointer");
  RC_REQUIRE(n > 0, "rc_dense_update: n < 0");
  SegArgs a;
  memset(&a, 0, sizeof(a));
  a.W = W; a.M = m; a.V = v;
  RC_TRY(fill_opt_scalars(h, &a.o));
  hipStream_t s = as_stream(stream);
  bool aligned = (reinterpret_cast<uintptr_t>(W) % 16 == 0) && (reinterpret_cast<uintptr_t>(G) % 16 == 0);
  switch (h->opt) {
    case RC_OPT_SGD:
      return launch_dense<MODE_SGD>(a, G, n, aligned, s);
    case RC_OPT_ADAM:
      RC_REQUIRE(m && v, "rc_dense_update: Adam needs m and v");
      aligned = aligned && (reinterpret_cast<uintptr_t>(m) % 16 == 0) &&
                (reinterpret_cast<uintptr_t>(v) % 16 == 0);
      return launch_dense<MODE_ADAM>(a, G, n, aligned, s);
    case RC_OPT_ADAGRAD:
      RC_REQUIRE(m, "rc_dense_update: Adagrad needs m (state_sum)");
      aligned = aligned && (reinterpret_cast<uintptr_t>(m) % 16 == 0);
      return launch_dense<MODE_ADAGRAD>(a, G, n, aligned, s);
    default:
      return fail(RC_ERR_INVALID_ARG, "rc_dense_update: unknown optimizer %d", h->opt);
  }
}
