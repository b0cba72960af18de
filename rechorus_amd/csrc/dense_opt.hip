// dense_opt.hip -- exact dense optimizer step over a whole tensor: torch.optim semantics
// including weight decay on EVERY element (reference: helpers/BaseRunner.py:110-114 builds
// torch.optim.<name>(..., weight_decay=l2); :206 steps it).  Used for small tables, where
// reproducing the reference's dense Adam/SGD/Adagrad bit-for-bit-in-semantics is affordable,
// and for the dense (MLP / attention) parameters.  Streaming, HBM-bound: W, G (+m, v) read
// once, W (+m, v) written once, float4 per lane.
#include "common.hpp"
#include "opt_math.hpp"

namespace rc {

template <int MODE>
__global__ __launch_bounds__(kBlock) void dense_update_kernel(OptScalars o, float* __restrict__ W,
                                                              const float* __restrict__ G,
                                                              float* __restrict__ M,
                                                              float* __restrict__ V, int64_t n) {
  const int64_t n4 = n / 4;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n4;
       i += (int64_t)gridDim.x * kBlock) {
    const float4 g = reinterpret_cast<const float4*>(G)[i];
    const float4 w = reinterpret_cast<const float4*>(W)[i];
    opt_row4<MODE>(o, W, M, V, (size_t)i, w, g);
  }
  const int64_t i = n4 * 4 + (int64_t)blockIdx.x * kBlock + threadIdx.x;  // n % 4 tail
  if (i < n) {
    float w = W[i], m = 0.f, v = 0.f;
    if (mode_has_m(MODE)) m = M[i];
    if (mode_has_v(MODE)) v = V[i];
    opt_elem<MODE>(o, G[i], w, m, v);
    W[i] = w;
    if (mode_has_m(MODE)) M[i] = m;
    if (mode_has_v(MODE)) V[i] = v;
  }
}

template <int MODE>
__global__ __launch_bounds__(kBlock) void dense_update_scalar_kernel(
    OptScalars o, float* __restrict__ W, const float* __restrict__ G, float* __restrict__ M,
    float* __restrict__ V, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * kBlock) {
    float w = W[i], m = 0.f, v = 0.f;
    if (mode_has_m(MODE)) m = M[i];
    if (mode_has_v(MODE)) v = V[i];
    opt_elem<MODE>(o, G[i], w, m, v);
    W[i] = w;
    if (mode_has_m(MODE)) M[i] = m;
    if (mode_has_v(MODE)) V[i] = v;
  }
}

template <int MODE>
static int launch_dense(const OptScalars& o, float* W, const float* G, float* M, float* V,
                        int64_t n, bool aligned, hipStream_t s) {
  const int64_t work = aligned ? (n / 4 > 0 ? n / 4 : 1) : n;
  int64_t blocks = (work + kBlock - 1) / kBlock;
  if (blocks > 256 * 16) blocks = 256 * 16;  // grid-stride beyond 16 blocks per CU
  if (aligned)
    hipLaunchKernelGGL((dense_update_kernel<MODE>), dim3((unsigned)blocks), dim3(kBlock), 0, s, o,
                       W, G, M, V, n);
  else
    hipLaunchKernelGGL((dense_update_scalar_kernel<MODE>), dim3((unsigned)blocks), dim3(kBlock), 0,
                       s, o, W, G, M, V, n);
  RC_LAUNCH_CHECK();
  return RC_OK;
}

// ---- many tensors, one launch ---------------------------------------------------------------------------
// A model's optimizer.step() touches every parameter tensor; one launch per tensor is ~10 us each of pure
// latency for the small ones (biases, LayerNorms, [vocab,1] tables).  Here up to kMultiMax tensors share a
// launch: a workgroup owns a 4096-element chunk of one tensor (found by a scan of <= kMultiMax block
// offsets), the arithmetic is opt_elem, exactly as in the single-tensor kernels.
constexpr int kMultiMax = 36;
constexpr int kMultiChunk = kBlock * 16;

struct MultiArgs {
  float* W[kMultiMax];
  const float* G[kMultiMax];
  float* M[kMultiMax];
  float* V[kMultiMax];
  int64_t n[kMultiMax];
  OptScalars o[kMultiMax];
  uint32_t blk0[kMultiMax + 1];
  uint8_t aligned[kMultiMax];
  int T;
  // capturable mode (hipGraph replay): Adam's step count lives in device memory and the two
  // bias-correction scalars are derived from it in the kernel instead of on the host
  const int64_t* step_dev;
  double beta1, beta2;
  float lr[kMultiMax];
};

template <int MODE>
__global__ __launch_bounds__(kBlock) void dense_update_multi_kernel(MultiArgs a) {
  int t = 0;
  while (t + 1 < a.T && blockIdx.x >= a.blk0[t + 1]) ++t;  // block-uniform
  const int64_t base = (int64_t)(blockIdx.x - a.blk0[t]) * kMultiChunk;
  const int64_t n = a.n[t];
  const int64_t end = base + kMultiChunk < n ? base + kMultiChunk : n;
  float* W = a.W[t];
  const float* G = a.G[t];
  float* M = a.M[t];
  float* V = a.V[t];
  OptScalars o = a.o[t];
  if (MODE == MODE_ADAM && a.step_dev != nullptr) {
    // same expressions as fill_opt_scalars (double, then narrowed like torch does)
    const double step = (double)*a.step_dev;
    const double bc1 = 1.0 - pow(a.beta1, step), bc2 = 1.0 - pow(a.beta2, step);
    o.neg_step = (float)(-((double)a.lr[t] / bc1));
    o.bc2_sqrt = (float)sqrt(bc2);
  }
  if (a.aligned[t]) {
    const int64_t end4 = base + ((end - base) / 4) * 4;
    for (int64_t i = base + 4 * (int64_t)threadIdx.x; i < end4; i += 4 * kBlock) {
      const float4 g = *reinterpret_cast<const float4*>(G + i);
      const float4 w = *reinterpret_cast<const float4*>(W + i);
      opt_row4<MODE>(o, W, M, V, (size_t)(i / 4), w, g);
    }
    for (int64_t i = end4 + threadIdx.x; i < end; i += kBlock) {
      float w = W[i], m = 0.f, v = 0.f;
      if (mode_has_m(MODE)) m = M[i];
      if (mode_has_v(MODE)) v = V[i];
      opt_elem<MODE>(o, G[i], w, m, v);
      W[i] = w;
      if (mode_has_m(MODE)) M[i] = m;
      if (mode_has_v(MODE)) V[i] = v;
    }
  } else {
    for (int64_t i = base + threadIdx.x; i < end; i += kBlock) {
      float w = W[i], m = 0.f, v = 0.f;
      if (mode_has_m(MODE)) m = M[i];
      if (mode_has_v(MODE)) v = V[i];
      opt_elem<MODE>(o, G[i], w, m, v);
      W[i] = w;
      if (mode_has_m(MODE)) M[i] = m;
      if (mode_has_v(MODE)) V[i] = v;
    }
  }
}

// ---- row-flagged dense update: the [vocab, d] field tables of a step that touches few rows -------------------------------
// torch.optim.Adam over a dense gradient moves EVERY row every step (g = 0 still decays m, v and steps along m), so the pass
// over the tables cannot be skipped -- but it only depends on the gradient for the rows the batch looked up.  Those rows are
// flagged by the gather (flags[row] == the step's number); `touched = 0` updates every OTHER row with g = 0 (no gradient buffer
// is read, so none is zero-filled, and the pass can run beside forward / backward on a second stream), `touched = 1` the
// flagged rows with their row sums.  Items with flags == nullptr are plain tensors (every element, from G).  Element
// arithmetic: opt_elem, as in dense_update_multi_kernel -- the two passes together are bit-identical to the dense step.
constexpr int kRowsMax = 24;

struct RowsArgs {
  float* W[kRowsMax];
  const float* G[kRowsMax];
  float* M[kRowsMax];
  float* V[kRowsMax];
  const int32_t* flags[kRowsMax];
  int64_t n[kRowsMax];
  OptScalars o[kRowsMax];
  uint32_t blk0[kRowsMax + 1];
  int row_w[kRowsMax];
  float lr[kRowsMax];
  uint8_t aligned[kRowsMax];
  int T;
  int touched;
  const int64_t* step_dev;
  double beta1, beta2;
};

template <int MODE>
__device__ __forceinline__ void rows_elem(const OptScalars& o, float* W, const float* G, float* M, float* V, const int32_t* flags,
                                          int64_t row0, uint32_t rem0, uint32_t row_w, int32_t gen, int mode, int i) {
  bool has_g = true;
  if (flags) {
    const bool hit = flags[row0 + (rem0 + (uint32_t)i) / row_w] == gen;
    if (mode != 2 && hit != (mode == 1)) return;
    has_g = hit && mode != 0;
  }
  float w1 = W[i], m1 = M[i], v1 = V[i];
  opt_elem<MODE>(o, has_g ? G[i] : 0.f, w1, m1, v1);
  W[i] = w1; M[i] = m1; V[i] = v1;
}

template <int MODE>
__global__ __launch_bounds__(kBlock) void dense_update_rows_kernel(RowsArgs a) {
  const int64_t step64 = *a.step_dev;
  const int32_t gen = (int32_t)step64;
  const int mode = a.touched;     // 0: the rows without this step's stamp (g = 0); 1: the stamped rows (from G); 2: every row
  const uint32_t n_chunks = a.blk0[a.T];
  int t = 0;
  // (the grid may be smaller than the number of chunks: the pass over the untouched rows shares the chip with the step's other
  // kernels -- a few resident workgroups per CU that stream, instead of a grid that fills every wave slot)
  for (uint32_t chunk = blockIdx.x; chunk < n_chunks; chunk += gridDim.x) {
    while (t + 1 < a.T && chunk >= a.blk0[t + 1]) ++t;  // block-uniform, chunks ascend
    const int64_t base = (int64_t)(chunk - a.blk0[t]) * kMultiChunk;
    const int64_t n = a.n[t];
    const int len = (int)(base + kMultiChunk < n ? kMultiChunk : n - base);
    float* W = a.W[t] + base;
    const float* G = a.G[t] ? a.G[t] + base : nullptr;
    float* M = a.M[t] + base;
    float* V = a.V[t] + base;
    const int32_t* flags = a.flags[t];
    const uint32_t row_w = (uint32_t)a.row_w[t];
    OptScalars o = a.o[t];
    auto bias_corrections = [&]() {
      if (MODE == MODE_ADAM) {
        const double step = (double)step64;
        const double bc1 = 1.0 - pow(a.beta1, step), bc2 = 1.0 - pow(a.beta2, step);
        o.neg_step = (float)(-((double)a.lr[t] / bc1));
        o.bc2_sqrt = (float)sqrt(bc2);
      }
    };
    const int64_t row0 = flags ? base / row_w : 0;
    const uint32_t rem0 = flags ? (uint32_t)(base - row0 * row_w) : 0u;
    if (a.aligned[t]) {   // 16-byte aligned pointers and row_w % 4 == 0: a float4 never straddles two rows
      const int len4 = (len / 4) * 4;
      bias_corrections();
      // (the loop of dense_update_multi_kernel plus the flag test: holding a chunk's four float4 per thread in registers to batch
      // the loads cost occupancy -- 138 VGPRs, 5.5 TB/s against this loop's 7)
      for (int i = 4 * (int)threadIdx.x; i < len4; i += 4 * kBlock) {
        bool has_g = true;
        if (flags) {
          const bool hit = flags[row0 + (rem0 + (uint32_t)i) / row_w] == gen;
          if (mode != 2 && hit != (mode == 1)) continue;
          has_g = hit && mode != 0;
        }
        float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
        if (has_g) g = *reinterpret_cast<const float4*>(G + i);
        const float4 w = *reinterpret_cast<const float4*>(W + i);
        opt_row4<MODE>(o, W, M, V, (size_t)(i / 4), w, g);
      }
      for (int i = len4 + (int)threadIdx.x; i < len; i += kBlock) rows_elem<MODE>(o, W, G, M, V, flags, row0, rem0, row_w, gen, mode, i);
    } else {
      bias_corrections();
      for (int i = (int)threadIdx.x; i < len; i += kBlock) rows_elem<MODE>(o, W, G, M, V, flags, row0, rem0, row_w, gen, mode, i);
    }
  }
}

}  // namespace rc

using namespace rc;

extern "C" int rc_dense_update(float* W, const float* G, float* m, float* v, int64_t n,
                               const rc_opt_hyper* h, rc_stream_t stream) {
  if (n == 0) return RC_OK;
  RC_REQUIRE(W && G, "rc_dense_update: null pointer");
  RC_REQUIRE(n > 0, "rc_dense_update: n < 0");
  OptScalars o;
  RC_TRY(fill_opt_scalars(h, &o, /*dense=*/true));
  hipStream_t s = as_stream(stream);
  auto al = [](const void* p) { return reinterpret_cast<uintptr_t>(p) % 16 == 0; };
  bool aligned = al(W) && al(G);
  switch (h->opt) {
    case RC_OPT_SGD:
      return launch_dense<MODE_SGD>(o, W, G, m, v, n, aligned, s);
    case RC_OPT_ADAM:
      RC_REQUIRE(m && v, "rc_dense_update: Adam needs m and v");
      return launch_dense<MODE_ADAM>(o, W, G, m, v, n, aligned && al(m) && al(v), s);
    case RC_OPT_ADAGRAD:
      RC_REQUIRE(m, "rc_dense_update: Adagrad needs m (state_sum)");
      return launch_dense<MODE_ADAGRAD>(o, W, G, m, v, n, aligned && al(m), s);
    case RC_OPT_ADADELTA:
      RC_REQUIRE(m && v, "rc_dense_update: Adadelta needs m (square_avg) and v (acc_delta)");
      return launch_dense<MODE_ADADELTA>(o, W, G, m, v, n, aligned && al(m) && al(v), s);
    default:
      return fail(RC_ERR_INVALID_ARG, "rc_dense_update: unknown optimizer %d", h->opt);
  }
}

__global__ void step_increment_kernel(int64_t* step) { *step += 1; }

extern "C" int rc_step_increment(int64_t* step_dev, rc_stream_t stream) {
  RC_REQUIRE(step_dev != nullptr, "rc_step_increment: null pointer");
  hipLaunchKernelGGL(step_increment_kernel, dim3(1), dim3(1), 0, as_stream(stream), step_dev);
  RC_LAUNCH_CHECK();
  return RC_OK;
}

__global__ void step_increment2_kernel(int64_t* a, int64_t* b) { *a += 1; *b += 1; }

// two device counters of one training step in one launch (the dropout seed of a tower and Adam's step count: in a replayed graph a
// launch costs ~4.6 us whatever it does)
extern "C" int rc_step_increment2(int64_t* a_dev, int64_t* b_dev, rc_stream_t stream) {
  RC_REQUIRE(a_dev != nullptr && b_dev != nullptr && a_dev != b_dev, "rc_step_increment2: two distinct counters");
  hipLaunchKernelGGL(step_increment2_kernel, dim3(1), dim3(1), 0, as_stream(stream), a_dev, b_dev);
  RC_LAUNCH_CHECK();
  return RC_OK;
}

extern "C" int rc_dense_update_multi(float* const* W, const float* const* G, float* const* m, float* const* v,
                                     const int64_t* n, const rc_opt_hyper* h, int n_tensors, rc_stream_t stream) {
  return rc_dense_update_multi_dev(W, G, m, v, n, h, n_tensors, nullptr, stream);
}

extern "C" int rc_dense_update_multi_dev(float* const* W, const float* const* G, float* const* m, float* const* v,
                                         const int64_t* n, const rc_opt_hyper* h, int n_tensors,
                                         const int64_t* step_dev, rc_stream_t stream) {
  if (n_tensors == 0) return RC_OK;
  RC_REQUIRE(W && G && n && h && n_tensors > 0, "rc_dense_update_multi: bad arguments");
  hipStream_t s = as_stream(stream);
  auto al = [](const void* p) { return reinterpret_cast<uintptr_t>(p) % 16 == 0; };
  const int opt = h[0].opt;
  for (int t0 = 0; t0 < n_tensors;) {
    MultiArgs a;
    memset(&a, 0, sizeof(a));
    uint32_t blocks = 0;
    int T = 0, t = t0;
    for (; t < n_tensors && T < kMultiMax; ++t) {
      RC_REQUIRE(h[t].opt == opt, "rc_dense_update_multi: one optimizer kind per call");
      RC_REQUIRE(n[t] >= 0, "rc_dense_update_multi: n < 0");
      if (n[t] == 0) continue;
      RC_REQUIRE(W[t] && G[t], "rc_dense_update_multi: null pointer (tensor %d)", t);
      float* mt = m ? m[t] : nullptr;
      float* vt = v ? v[t] : nullptr;
      RC_REQUIRE((opt != RC_OPT_ADAM && opt != RC_OPT_ADADELTA) || (mt && vt), "rc_dense_update_multi: Adam / Adadelta need m and v (tensor %d)", t);
      RC_REQUIRE(opt != RC_OPT_ADAGRAD || mt, "rc_dense_update_multi: Adagrad needs m (tensor %d)", t);
      a.W[T] = W[t]; a.G[T] = G[t]; a.M[T] = mt; a.V[T] = vt; a.n[T] = n[t];
      RC_TRY(fill_opt_scalars(&h[t], &a.o[T], /*dense=*/true));
      a.lr[T] = (float)h[t].lr;
      RC_REQUIRE(step_dev == nullptr || (h[t].beta1 == h[0].beta1 && h[t].beta2 == h[0].beta2),
                 "rc_dense_update_multi_dev: one (beta1, beta2) per call");
      a.aligned[T] = al(W[t]) && al(G[t]) && al(mt) && al(vt);
      a.blk0[T] = blocks;
      const int64_t nb = (n[t] + kMultiChunk - 1) / kMultiChunk;
      RC_REQUIRE(nb < ((int64_t)1 << 30) - blocks, "rc_dense_update_multi: grid too large");
      blocks += (uint32_t)nb;
      ++T;
    }
    t0 = t;   // (empty tensors do not take a slot: the next launch starts where this one stopped)
    if (T == 0) continue;
    a.blk0[T] = blocks;
    a.T = T;
    a.step_dev = step_dev;
    a.beta1 = h[0].beta1;
    a.beta2 = h[0].beta2;
    switch (opt) {
      case RC_OPT_SGD:
        hipLaunchKernelGGL((dense_update_multi_kernel<MODE_SGD>), dim3(blocks), dim3(kBlock), 0, s, a);
        break;
      case RC_OPT_ADAM:
        hipLaunchKernelGGL((dense_update_multi_kernel<MODE_ADAM>), dim3(blocks), dim3(kBlock), 0, s, a);
        break;
      case RC_OPT_ADAGRAD:
        hipLaunchKernelGGL((dense_update_multi_kernel<MODE_ADAGRAD>), dim3(blocks), dim3(kBlock), 0, s, a);
        break;
      case RC_OPT_ADADELTA:
        hipLaunchKernelGGL((dense_update_multi_kernel<MODE_ADADELTA>), dim3(blocks), dim3(kBlock), 0, s, a);
        break;
      default:
        return fail(RC_ERR_INVALID_ARG, "rc_dense_update_multi: unknown optimizer %d", opt);
    }
    RC_LAUNCH_CHECK();
  }
  return RC_OK;
}

extern "C" int rc_dense_update_rows_dev(float* const* W, const float* const* G, float* const* m, float* const* v,
                                        const int64_t* n, const int32_t* const* flags, const int* row_w, const rc_opt_hyper* h,
                                        int n_tensors, int touched, int max_blocks, const int64_t* step_dev, rc_stream_t stream) {
  if (n_tensors == 0) return RC_OK;
  RC_REQUIRE(W && n && h && flags && row_w && n_tensors > 0, "rc_dense_update_rows_dev: bad arguments");
  RC_REQUIRE(step_dev != nullptr, "rc_dense_update_rows_dev: the step count lives in device memory (step_dev)");
  RC_REQUIRE(touched >= 0 && touched <= 2, "rc_dense_update_rows_dev: touched is 0, 1 or 2");
  hipStream_t s = as_stream(stream);
  auto al = [](const void* p) { return reinterpret_cast<uintptr_t>(p) % 16 == 0; };
  for (int t0 = 0; t0 < n_tensors;) {
    RowsArgs a;
    memset(&a, 0, sizeof(a));
    uint32_t blocks = 0;
    int T = 0, t = t0;
    for (; t < n_tensors && T < kRowsMax; ++t) {
      RC_REQUIRE(h[t].opt == RC_OPT_ADAM, "rc_dense_update_rows_dev: Adam only (the other optimizers leave rows with g = 0 and "
                 "no weight decay alone -- their sparse-aware step is rc_segmented_update)");
      RC_REQUIRE(n[t] >= 0, "rc_dense_update_rows_dev: n < 0");
      if (n[t] == 0) continue;
      const bool needs_g = flags[t] == nullptr || touched != 0;
      RC_REQUIRE(W[t] && m && v && m[t] && v[t], "rc_dense_update_rows_dev: null pointer (tensor %d)", t);
      RC_REQUIRE(!needs_g || (G && G[t]), "rc_dense_update_rows_dev: null gradient (tensor %d)", t);
      RC_REQUIRE(flags[t] == nullptr || (row_w[t] >= 1 && n[t] % row_w[t] == 0), "rc_dense_update_rows_dev: n is not rows x row_w (tensor %d)", t);
      RC_REQUIRE(h[t].beta1 == h[0].beta1 && h[t].beta2 == h[0].beta2, "rc_dense_update_rows_dev: one (beta1, beta2) per call");
      a.W[T] = W[t]; a.G[T] = needs_g ? G[t] : nullptr; a.M[T] = m[t]; a.V[T] = v[t]; a.n[T] = n[t];
      a.flags[T] = flags[t];
      a.row_w[T] = flags[t] ? row_w[t] : 4;
      RC_TRY(fill_opt_scalars(&h[t], &a.o[T], /*dense=*/true));
      a.lr[T] = (float)h[t].lr;
      a.aligned[T] = al(W[t]) && (!needs_g || al(G[t])) && al(m[t]) && al(v[t]) && a.row_w[T] % 4 == 0;
      a.blk0[T] = blocks;
      const int64_t nb = (n[t] + kMultiChunk - 1) / kMultiChunk;
      RC_REQUIRE(nb < ((int64_t)1 << 30) - blocks, "rc_dense_update_rows_dev: grid too large");
      blocks += (uint32_t)nb;
      ++T;
    }
    t0 = t;
    if (T == 0) continue;
    a.blk0[T] = blocks;
    a.T = T;
    a.touched = touched;
    a.step_dev = step_dev;
    a.beta1 = h[0].beta1;
    a.beta2 = h[0].beta2;
    const uint32_t grid = max_blocks > 0 && blocks > (uint32_t)max_blocks ? (uint32_t)max_blocks : blocks;
    hipLaunchKernelGGL((dense_update_rows_kernel<MODE_ADAM>), dim3(grid), dim3(kBlock), 0, s, a);
    RC_LAUNCH_CHECK();
  }
  return RC_OK;
}
