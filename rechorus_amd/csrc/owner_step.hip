// owner_step.hip -- local kernels of the row-sharded ("owner computes") training step
// (rechorus_amd/sharded.py; SURVEY.md §8e: tables sharded by row, owner(id) = id mod W).
//
// The reference is single-device; these replace nothing in it one-to-one.  They are the pieces of
// rc_bprmf_train_step re-cut at the points where the sharded step has to exchange data over xGMI:
//
//   rc_route_by_owner    stable counting sort of a batch's ids by owner rank -> send order, packed
//                        (tuple, local row) messages, per-destination counts.  Deterministic (no
//                        atomics in the placement), so the owner sees occurrences in a fixed order.
//   rc_owner_unpack      received messages -> tuple index / local row arrays for the owner kernels.
//   rc_owner_backward    on the item rows' owner, after dL/dscore came back: one lane-group per
//                        (source rank, tuple) run of occurrences -- they arrive grouped by tuple because
//                        the source sends them in batch order -- accumulates the tuple's partial user
//                        gradient  pug[t] = sum g*I[row]  and, while the row is in registers, applies
//                        the optimizer to rows that occur once on this owner (same singleton fast
//                        path as the fused single-GPU kernel).  Multi-occurrence rows are left to
//                        rc_segmented_update.
#include "common.hpp"
#include "opt_math.hpp"

namespace rc {

constexpr int kRouteIters = 16;                       // tile = 16 * 256 = 4096 ids per workgroup
constexpr int kRouteTile = kRouteIters * kBlock;
constexpr int kRouteSlots = kRouteIters * (kBlock / 64);  // (iteration, wave) slots of a tile
constexpr int kMaxWorld = 64;

// per (slot, owner) counts of one tile in LDS; returns this thread's owner per iteration in own[]
// and its rank among the same-owner lanes of its wave in below[]
__device__ __forceinline__ void route_count_tile(const int64_t* __restrict__ ids, int64_t n, int world,
                                                 int64_t tile0, uint32_t* s_cnt /*[slots][world]*/,
                                                 int* own, int* below) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int e = threadIdx.x; e < kRouteSlots * world; e += kBlock) s_cnt[e] = 0;
  __syncthreads();
#pragma unroll
  for (int it = 0; it < kRouteIters; ++it) {
    const int64_t p = tile0 + (int64_t)it * kBlock + threadIdx.x;
    const int w = p < n ? (int)(ids[p] % world) : -1;
    own[it] = w;
    below[it] = 0;
    unsigned long long todo = __ballot(w >= 0);
    while (todo) {  // one trip per distinct owner present in the wave
      const int first = __ffsll((long long)todo) - 1;
      const int w0 = __shfl(w, first, 64);
      const unsigned long long same = __ballot(w == w0);
      if (w == w0) below[it] = __popcll(same & ((1ull << lane) - 1ull));
      if (lane == first) s_cnt[(it * (kBlock / 64) + wave) * world + w0] = (uint32_t)__popcll(same);
      todo &= ~same;
    }
  }
  __syncthreads();
}

__global__ __launch_bounds__(kBlock) void route_hist_kernel(const int64_t* __restrict__ ids, int64_t n, int world,
                                                            uint32_t* __restrict__ hist /*[world][n_tiles]*/,
                                                            int n_tiles) {
  extern __shared__ uint32_t s_cnt[];
  int own[kRouteIters], below[kRouteIters];
  route_count_tile(ids, n, world, (int64_t)blockIdx.x * kRouteTile, s_cnt, own, below);
  if ((int)threadIdx.x < world) {
    uint32_t total = 0;
    for (int s = 0; s < kRouteSlots; ++s) total += s_cnt[s * world + threadIdx.x];
    hist[(size_t)threadIdx.x * n_tiles + blockIdx.x] = total;
  }
}

// exclusive scan of hist in (owner, tile) order, in place; counts[w] = ids owned by rank w
__global__ __launch_bounds__(kBlock) void route_scan_kernel(uint32_t* __restrict__ hist, int world, int n_tiles,
                                                            int64_t* __restrict__ counts) {
  __shared__ uint32_t s_part[kBlock];
  // counts[w]: integer sums, order-free -- the whole workgroup per owner (one thread per owner walking all tiles one dependent load
  // after the other took 51 us for the 1,600 tiles of a 6.5 M-id batch)
  for (int w = 0; w < world; ++w) {
    uint32_t c = 0;
    for (int i = threadIdx.x; i < n_tiles; i += kBlock) c += hist[(size_t)w * n_tiles + i];
    s_part[threadIdx.x] = c;
    __syncthreads();
    for (int off = kBlock / 2; off >= 1; off >>= 1) {
      if ((int)threadIdx.x < off) s_part[threadIdx.x] += s_part[threadIdx.x + off];
      __syncthreads();
    }
    if (threadIdx.x == 0) counts[w] = (int64_t)s_part[0];
    __syncthreads();
  }
  const int64_t total = (int64_t)world * n_tiles;
  const int64_t per = (total + kBlock - 1) / kBlock;
  const int64_t lo = (int64_t)threadIdx.x * per < total ? (int64_t)threadIdx.x * per : total;
  const int64_t hi = lo + per < total ? lo + per : total;
  uint32_t sum = 0;
  for (int64_t e = lo; e < hi; ++e) sum += hist[e];
  s_part[threadIdx.x] = sum;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t run = 0;
    for (int i = 0; i < kBlock; ++i) {
      const uint32_t c = s_part[i];
      s_part[i] = run;
      run += c;
    }
  }
  __syncthreads();
  uint32_t run = s_part[threadIdx.x];
  for (int64_t e = lo; e < hi; ++e) {
    const uint32_t c = hist[e];
    hist[e] = run;
    run += c;
  }
}

__global__ __launch_bounds__(kBlock) void route_scatter_kernel(
    const int64_t* __restrict__ ids, int64_t n, int world, const uint32_t* __restrict__ base /*[world][n_tiles]*/,
    int n_tiles, int64_t tuple_base, int div, uint32_t* __restrict__ order, int64_t* __restrict__ packed,
    int64_t* __restrict__ local_row) {
  extern __shared__ uint32_t s_cnt[];
  int own[kRouteIters], below[kRouteIters];
  const int64_t tile0 = (int64_t)blockIdx.x * kRouteTile;
  route_count_tile(ids, n, world, tile0, s_cnt, own, below);
  // exclusive scan over the tile's slots, per owner, offset by the tile's global base
  if ((int)threadIdx.x < world) {
    uint32_t run = base[(size_t)threadIdx.x * n_tiles + blockIdx.x];
    for (int s = 0; s < kRouteSlots; ++s) {
      const uint32_t c = s_cnt[s * world + threadIdx.x];
      s_cnt[s * world + threadIdx.x] = run;
      run += c;
    }
  }
  __syncthreads();
  const int wave = threadIdx.x >> 6;
#pragma unroll
  for (int it = 0; it < kRouteIters; ++it) {
    const int w = own[it];
    if (w < 0) continue;
    const int64_t p = tile0 + (int64_t)it * kBlock + threadIdx.x;
    const uint32_t dst = s_cnt[(it * (kBlock / 64) + wave) * world + w] + (uint32_t)below[it];
    const int64_t row = ids[p] / world;
    order[dst] = (uint32_t)p;
    if (packed) packed[dst] = ((tuple_base + p / div) << 32) | row;
    if (local_row) local_row[dst] = row;
  }
}

__global__ __launch_bounds__(kBlock) void owner_unpack_kernel(const int64_t* __restrict__ packed, int64_t n,
                                                              int64_t* __restrict__ t_idx, int64_t* __restrict__ rows,
                                                              uint32_t* __restrict__ t32) {
  for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < n; e += (int64_t)gridDim.x * kBlock) {
    const int64_t m = packed[e];
    const int64_t t = m >> 32;
    t_idx[e] = t;
    rows[e] = m & 0xFFFFFFFFll;
    t32[e] = (uint32_t)t;
  }
}

struct OwnerArgs {
  float* I;
  float* M;
  float* V;
  const float* Uall;
  const uint32_t* t32;
  const int64_t* rows;
  const float* g;
  const uint8_t* single;
  const uint32_t* heads;
  const uint32_t* n_heads;
  int64_t n;
  float* pug;
  OptScalars o;
};

// one lane-group per run of occurrences of one tuple (ascending position = fixed summation order)
template <int D, int MODE>
__global__ __launch_bounds__(kBlock) void owner_backward_kernel(OwnerArgs a) {
  constexpr int LPR = D / 4;
  constexpr int GPB = kBlock / LPR;
  const int l = threadIdx.x % LPR;
  const int64_t grp = (int64_t)blockIdx.x * GPB + threadIdx.x / LPR;
  if (grp >= (int64_t)*a.n_heads) return;  // no cross-lane ops in this kernel
  const int64_t j0 = a.heads[grp];
  const uint32_t t = a.t32[j0];
  const float4 u = reinterpret_cast<const float4*>(a.Uall)[(size_t)t * LPR + l];
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  // four occurrences per trip: their row ids, coefficients and rows are requested together (the loop is
  // bound by the latency of the dependent loads rows[j] -> I[row]); accumulation stays in ascending order
  constexpr int U4 = 4;
  int64_t j = j0;
  while (j < a.n && a.t32[j] == t) {
    size_t idx[U4];
    float gj[U4];
    float4 x[U4];
    bool on[U4];
#pragma unroll
    for (int q = 0; q < U4; ++q) {
      on[q] = j + q < a.n && a.t32[j + q] == t;
      idx[q] = on[q] ? (size_t)a.rows[j + q] * LPR + l : 0;
      gj[q] = on[q] ? a.g[j + q] : 0.f;
    }
#pragma unroll
    for (int q = 0; q < U4; ++q)
      x[q] = on[q] ? load_stream4(reinterpret_cast<const float4*>(a.I) + idx[q]) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int q = 0; q < U4; ++q) {
      if (!on[q]) break;  // a run is contiguous: once off, the rest is off
      float4 s = x[q];
      s.x *= gj[q]; s.y *= gj[q]; s.z *= gj[q]; s.w *= gj[q];
      acc.x += s.x; acc.y += s.y; acc.z += s.z; acc.w += s.w;
      if (MODE != MODE_NONE && a.single[j + q]) {  // nobody else reads or writes this row in this step
        const float4 grad = make_float4(u.x * gj[q], u.y * gj[q], u.z * gj[q], u.w * gj[q]);
        opt_row4<MODE>(a.o, a.I, a.M, a.V, idx[q], x[q], grad);
      }
    }
    j += U4;
  }
  reinterpret_cast<float4*>(a.pug)[(size_t)t * LPR + l] = acc;
}

template <int MODE>
static int launch_owner_backward(const OwnerArgs& a, int d, int64_t n, hipStream_t s) {
#define RC_OWNER_CASE(DD)                                                                                   \
  case DD: {                                                                                                \
    const int64_t blocks = (n + (kBlock / (DD / 4)) - 1) / (kBlock / (DD / 4));                             \
    if (blocks > kMaxGridX) return fail(RC_ERR_UNSUPPORTED, "rc_owner_backward: grid too large");           \
    hipLaunchKernelGGL((owner_backward_kernel<DD, MODE>), dim3((unsigned)blocks), dim3(kBlock), 0, s, a);   \
    break;                                                                                                  \
  }
  switch (d) {
    RC_OWNER_CASE(16)
    RC_OWNER_CASE(32)
    RC_OWNER_CASE(64)
    RC_OWNER_CASE(128)
    RC_OWNER_CASE(256)
    default: return fail(RC_ERR_UNSUPPORTED, "rc_owner_backward: emb_size must be 16/32/64/128/256, got %d", d);
  }
#undef RC_OWNER_CASE
  RC_LAUNCH_CHECK();
  return RC_OK;
}

struct RouteWs {
  uint32_t* hist;
  size_t total;
};
static RouteWs carve_route_ws(void* base, int64_t n, int world) {
  Carver cv(base);
  RouteWs w;
  const int64_t n_tiles = (n + kRouteTile - 1) / kRouteTile;
  w.hist = cv.take<uint32_t>((size_t)world * (size_t)(n_tiles > 0 ? n_tiles : 1) + 1);
  w.total = cv.off;
  return w;
}

}  // namespace rc

using namespace rc;

extern "C" size_t rc_route_workspace_bytes(int64_t n, int world) {
  if (n < 0 || world < 1) return 0;
  return carve_route_ws(nullptr, n, world).total;
}

extern "C" int rc_route_by_owner(const int64_t* ids, int64_t n, int world, int64_t tuple_base, int div,
                                 uint32_t* order, int64_t* packed, int64_t* local_row, int64_t* counts, void* ws,
                                 size_t ws_bytes, rc_stream_t stream) {
  RC_REQUIRE(world >= 1 && world <= kMaxWorld, "rc_route_by_owner: world must be in [1, %d], got %d", kMaxWorld, world);
  RC_REQUIRE(counts != nullptr, "rc_route_by_owner: null pointer");
  hipStream_t s = as_stream(stream);
  if (n == 0) {
    RC_HIP(hipMemsetAsync(counts, 0, (size_t)world * sizeof(int64_t), s));
    return RC_OK;
  }
  RC_REQUIRE(ids && order && ws, "rc_route_by_owner: null pointer");
  RC_REQUIRE(n > 0 && n < ((int64_t)1 << 31) && div >= 1, "rc_route_by_owner: bad shape n=%lld div=%d", (long long)n, div);
  const RouteWs w = carve_route_ws(ws, n, world);
  if (ws_bytes < w.total) return fail(RC_ERR_WORKSPACE, "rc_route_by_owner: workspace %zu < %zu", ws_bytes, w.total);
  const int n_tiles = (int)((n + kRouteTile - 1) / kRouteTile);
  const size_t lds = (size_t)kRouteSlots * world * sizeof(uint32_t);
  hipLaunchKernelGGL(route_hist_kernel, dim3((unsigned)n_tiles), dim3(kBlock), lds, s, ids, n, world, w.hist, n_tiles);
  RC_LAUNCH_CHECK();
  hipLaunchKernelGGL(route_scan_kernel, dim3(1), dim3(kBlock), 0, s, w.hist, world, n_tiles, counts);
  RC_LAUNCH_CHECK();
  hipLaunchKernelGGL(route_scatter_kernel, dim3((unsigned)n_tiles), dim3(kBlock), lds, s, ids, n, world, w.hist,
                     n_tiles, tuple_base, div, order, packed, local_row);
  RC_LAUNCH_CHECK();
  return RC_OK;
}

extern "C" int rc_owner_unpack(const int64_t* packed, int64_t n, int64_t* t_idx, int64_t* rows, uint32_t* t32,
                               rc_stream_t stream) {
  if (n == 0) return RC_OK;
  RC_REQUIRE(packed && t_idx && rows && t32, "rc_owner_unpack: null pointer");
  RC_REQUIRE(n > 0, "rc_owner_unpack: n < 0");
  int64_t blocks = (n + kBlock - 1) / kBlock;
  if (blocks > 256 * 16) blocks = 256 * 16;
  hipLaunchKernelGGL(owner_unpack_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, as_stream(stream), packed, n, t_idx,
                     rows, t32);
  RC_LAUNCH_CHECK();
  return RC_OK;
}

extern "C" size_t rc_owner_backward_workspace_bytes(int64_t n) {
  if (n < 1) n = 1;
  return ((size_t)n + 64) * sizeof(uint32_t);
}

extern "C" int rc_owner_backward(float* I, float* mI, float* vI, int d, const float* Uall, const uint32_t* t32,
                                 const int64_t* rows, const float* g, const uint8_t* single, int64_t n,
                                 int64_t n_tuples, const rc_opt_hyper* h, float* pug, void* ws, size_t ws_bytes,
                                 rc_stream_t stream) {
  RC_REQUIRE(pug != nullptr && n_tuples >= 0 && d >= 1, "rc_owner_backward: bad arguments");
  hipStream_t s = as_stream(stream);
  RC_HIP(hipMemsetAsync(pug, 0, (size_t)n_tuples * d * sizeof(float), s));  // tuples with no row here
  if (n == 0) return RC_OK;
  RC_REQUIRE(I && Uall && t32 && rows && g && ws, "rc_owner_backward: null pointer");
  RC_REQUIRE(n > 0 && n < ((int64_t)1 << 31), "rc_owner_backward: bad n=%lld", (long long)n);
  RC_REQUIRE(ws_bytes >= rc_owner_backward_workspace_bytes(n), "rc_owner_backward: workspace too small");
  auto al = [](const void* p) { return reinterpret_cast<uintptr_t>(p) % 16 == 0; };
  RC_REQUIRE(al(I) && al(mI) && al(vI) && al(Uall) && al(pug), "rc_owner_backward: tables must be 16-byte aligned");
  uint32_t* n_heads = static_cast<uint32_t*>(ws);
  uint32_t* heads = n_heads + 64;
  // run starts: positions where the tuple index changes (t32 is non-decreasing by construction)
  RC_TRY(rc_segment_heads(t32, t32, n, 0, nullptr, heads, n_heads, stream));
  OwnerArgs a;
  memset(&a, 0, sizeof(a));
  a.I = I; a.M = mI; a.V = vI; a.Uall = Uall; a.t32 = t32; a.rows = rows; a.g = g; a.single = single;
  a.heads = heads; a.n_heads = n_heads; a.n = n; a.pug = pug;
  if (single == nullptr) return launch_owner_backward<MODE_NONE>(a, d, n, s);
  RC_TRY(fill_opt_scalars(h, &a.o));
  const int mode = mode_of(h);
  RC_REQUIRE(mode != MODE_ADAM || (mI && vI), "rc_owner_backward: Adam needs m and v");
  RC_REQUIRE(mode != MODE_ADAGRAD || mI, "rc_owner_backward: Adagrad needs m (state_sum)");
  switch (mode) {
    case MODE_SGD: return launch_owner_backward<MODE_SGD>(a, d, n, s);
    case MODE_ADAM: return launch_owner_backward<MODE_ADAM>(a, d, n, s);
    default: return launch_owner_backward<MODE_ADAGRAD>(a, d, n, s);
  }
}
