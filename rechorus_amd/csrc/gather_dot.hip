// gather_dot.hip -- forward-only kernels: embedding row gather and the BPRMF/GMF score
//   pred[b,c] = <U[uid[b]], I[iid[b,c]]>      (reference: models/general/BPRMF.py:39-42)
//
// HBM-bound.  A table row of d fp32 is read by LPR = d/4 consecutive lanes as one float4
// each (d = 64: 16 lanes x 16 B = one fully used 256-B segment), so one wave instruction
// fetches 64/LPR rows; the dot is finished with DPP row reductions (no LDS traffic).
#include "common.hpp"

namespace rc {

// ---- rc_gather_rows --------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void gather_rows_vec4_kernel(
    const float4* __restrict__ W, int dq, const int64_t* __restrict__ ids, int64_t n,
    float4* __restrict__ out) {
  const int64_t total = n * dq;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * kBlock) {
    const int64_t r = i / dq;
    const int q = (int)(i - r * dq);
    out[i] = W[ids[r] * dq + q];
  }
}

// two tables that share the ids, rows side by side: out[i, :] = (Wa[ids[i], :] | Wb[ids[i], :]) -- the block a sharded NeuMF rank
// serves for the ids it owns (mf | mlp), written once instead of two gathers and a concatenation
__global__ __launch_bounds__(kBlock) void gather_rows_pair_vec4_kernel(const float4* __restrict__ Wa, const float4* __restrict__ Wb, int dq,
                                                                      const int64_t* __restrict__ ids, int64_t n, float4* __restrict__ out) {
  const int64_t total = n * 2 * dq;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (int64_t)gridDim.x * kBlock) {
    const int64_t r = i / (2 * dq);
    const int q = (int)(i - r * 2 * dq);
    out[i] = q < dq ? Wa[ids[r] * dq + q] : Wb[ids[r] * dq + (q - dq)];
  }
}

__global__ __launch_bounds__(kBlock) void gather_rows_scalar_kernel(
    const float* __restrict__ W, int d, const int64_t* __restrict__ ids, int64_t n,
    float* __restrict__ out) {
  const int64_t total = n * d;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * kBlock) {
    const int64_t r = i / d;
    const int q = (int)(i - r * d);
    out[i] = W[ids[r] * d + q];
  }
}

// ---- rc_gather_dot_fwd -------------------------------------------------------------
// One lane-group (LPR lanes) per candidate, PER candidates per group so that PER row
// loads are in flight per lane before the first dot.
template <int D, int PER>
__global__ __launch_bounds__(kBlock) void gather_dot_fwd_kernel(
    const float* __restrict__ U, const float* __restrict__ I,
    const int64_t* __restrict__ uid, const int64_t* __restrict__ iid, int64_t n_pairs,
    int C, float* __restrict__ pred) {
  constexpr int LPR = D / 4;
  constexpr int GPB = kBlock / LPR;  // lane-groups per block
  const int l = threadIdx.x % LPR;
  const int64_t g = (int64_t)blockIdx.x * GPB + threadIdx.x / LPR;
  const int64_t o0 = g * PER;
  // no early return: the DPP reductions need whole rows of lanes, which they have
  // because LPR divides the wave; groups past the end clamp their loads and skip stores.
  float4 r[PER];
  float4 u[PER];
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    int64_t o = o0 + k;
    if (o >= n_pairs) o = n_pairs - 1;
    const int64_t b = o / C;
    const int64_t ui = uid[b];
    const int64_t ii = iid[o];
    u[k] = reinterpret_cast<const float4*>(U + ui * D)[l];
    r[k] = reinterpret_cast<const float4*>(I + ii * D)[l];
  }
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    float p = row_allreduce_sum<LPR>(dot4(u[k], r[k]));
    const int64_t o = o0 + k;
    if (l == 0 && o < n_pairs) pred[o] = p;
  }
}

// any d: one wave per candidate, lanes stride over the row
__global__ __launch_bounds__(kBlock) void gather_dot_fwd_generic_kernel(
    const float* __restrict__ U, const float* __restrict__ I,
    const int64_t* __restrict__ uid, const int64_t* __restrict__ iid, int64_t n_pairs,
    int C, int d, float* __restrict__ pred) {
  const int lane = threadIdx.x & 63;
  const int64_t wave0 = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
  const int64_t nw = (int64_t)gridDim.x * (kBlock / 64);
  for (int64_t o = wave0; o < n_pairs; o += nw) {
    const float* ur = U + uid[o / C] * d;
    const float* ir = I + iid[o] * d;
    float acc = 0.f;
    for (int k = lane; k < d; k += 64) acc = fmaf(ur[k], ir[k], acc);
    acc = wave_allreduce_sum(acc);
    if (lane == 0) pred[o] = acc;
  }
}

// ---- rc_weighted_row_sum: out[b,:] = sum_c coef[b,c] * W[ids[b,c],:] ---------------------
// (the user-side half of MulBackward/SumBackward of BPRMF.py:42).  One wave per tuple, its
// 64/LPR lane-groups stride over the candidates, partial sums combined across groups.
template <int D>
__global__ __launch_bounds__(kBlock) void weighted_row_sum_kernel(
    const float* __restrict__ W, const int64_t* __restrict__ ids, const float* __restrict__ coef,
    int B, int C, float* __restrict__ out) {
  constexpr int LPR = D / 4;
  constexpr int G = 64 / LPR;
  const int lane = threadIdx.x & 63;
  const int grp = lane / LPR;
  const int l = lane % LPR;
  const int64_t t_raw = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
  const bool tv = t_raw < B;  // wave-uniform
  const int64_t t = tv ? t_raw : (int64_t)B - 1;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int c = grp; c < C; c += G) {
    const float g = coef[t * C + c];
    const float4 r = reinterpret_cast<const float4*>(W + ids[t * C + c] * D)[l];
    acc.x = fmaf(g, r.x, acc.x);
    acc.y = fmaf(g, r.y, acc.y);
    acc.z = fmaf(g, r.z, acc.z);
    acc.w = fmaf(g, r.w, acc.w);
  }
  acc.x = groups_allreduce_sum<LPR, 64>(acc.x);
  acc.y = groups_allreduce_sum<LPR, 64>(acc.y);
  acc.z = groups_allreduce_sum<LPR, 64>(acc.z);
  acc.w = groups_allreduce_sum<LPR, 64>(acc.w);
  if (tv && grp == 0) reinterpret_cast<float4*>(out + t * D)[l] = acc;
}

__global__ __launch_bounds__(kBlock) void weighted_row_sum_generic_kernel(
    const float* __restrict__ W, const int64_t* __restrict__ ids, const float* __restrict__ coef,
    int B, int C, int d, float* __restrict__ out) {
  // one thread per output element; candidates summed sequentially
  const int64_t total = (int64_t)B * d;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * kBlock) {
    const int64_t t = i / d;
    const int k = (int)(i - t * d);
    float acc = 0.f;
    for (int c = 0; c < C; ++c) acc = fmaf(coef[t * C + c], W[ids[t * C + c] * d + k], acc);
    out[i] = acc;
  }
}

template <int D>
static int launch_gather_dot(const float* U, const float* I, const int64_t* uid,
                             const int64_t* iid, int64_t n_pairs, int C, float* pred,
                             hipStream_t s) {
  constexpr int PER = 4;
  constexpr int GPB = kBlock / (D / 4);
  const int64_t groups = (n_pairs + PER - 1) / PER;
  const int64_t blocks = (groups + GPB - 1) / GPB;
  if (blocks > kMaxGridX) return fail(RC_ERR_UNSUPPORTED, "gather_dot: grid too large");
  hipLaunchKernelGGL((gather_dot_fwd_kernel<D, PER>), dim3((unsigned)blocks), dim3(kBlock), 0,
                     s, U, I, uid, iid, n_pairs, C, pred);
  RC_LAUNCH_CHECK();
  return RC_OK;
}

}  // namespace rc

using namespace rc;

extern "C" int rc_gather_rows(const float* W, int d, const int64_t* ids, int64_t n,
                              float* out, rc_stream_t stream) {
  if (n == 0) return RC_OK;  // empty batch: nothing to check, nothing to do
  RC_REQUIRE(W && ids && out, "rc_gather_rows: null pointer");
  RC_REQUIRE(d >= 1 && n > 0, "rc_gather_rows: bad shape d=%d n=%lld", d, (long long)n);
  hipStream_t s = as_stream(stream);
  const bool vec = (d % 4 == 0) && (reinterpret_cast<uintptr_t>(W) % 16 == 0) &&
                   (reinterpret_cast<uintptr_t>(out) % 16 == 0);
  const int64_t total = vec ? n * (d / 4) : n * (int64_t)d;
  int64_t blocks = (total + kBlock - 1) / kBlock;
  if (blocks > 256 * 32) blocks = 256 * 32;  // grid-stride above 32 blocks/CU
  if (vec) {
    hipLaunchKernelGGL(gather_rows_vec4_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, s,
                       reinterpret_cast<const float4*>(W), d / 4, ids, n,
                       reinterpret_cast<float4*>(out));
  } else {
    hipLaunchKernelGGL(gather_rows_scalar_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, s,
                       W, d, ids, n, out);
  }
  RC_LAUNCH_CHECK();
  return RC_OK;
}

extern "C" int rc_gather_rows_pair(const float* Wa, const float* Wb, int d, const int64_t* ids, int64_t n, float* out, rc_stream_t stream) {
  if (n == 0) return RC_OK;
  RC_REQUIRE(Wa && Wb && ids && out, "rc_gather_rows_pair: null pointer");
  RC_REQUIRE(d >= 4 && d % 4 == 0 && n > 0, "rc_gather_rows_pair: bad shape d=%d (a multiple of 4) n=%lld", d, (long long)n);
  RC_REQUIRE(reinterpret_cast<uintptr_t>(Wa) % 16 == 0 && reinterpret_cast<uintptr_t>(Wb) % 16 == 0 && reinterpret_cast<uintptr_t>(out) % 16 == 0,
             "rc_gather_rows_pair: tables and output must be 16-byte aligned");
  const int64_t total = n * (d / 2);
  int64_t blocks = (total + kBlock - 1) / kBlock;
  if (blocks > 256 * 32) blocks = 256 * 32;
  hipLaunchKernelGGL(gather_rows_pair_vec4_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, as_stream(stream), reinterpret_cast<const float4*>(Wa),
                     reinterpret_cast<const float4*>(Wb), d / 4, ids, n, reinterpret_cast<float4*>(out));
  RC_LAUNCH_CHECK();
  return RC_OK;
}

extern "C" int rc_gather_dot_fwd(const float* U, const float* I, const int64_t* uid,
                                 const int64_t* iid, int B, int C, int d, float* pred,
                                 rc_stream_t stream) {
  if (B == 0) return RC_OK;
  RC_REQUIRE(U && I && uid && iid && pred, "rc_gather_dot_fwd: null pointer");
  RC_REQUIRE(B > 0 && C >= 1 && d >= 1, "rc_gather_dot_fwd: bad shape B=%d C=%d d=%d", B, C, d);
  hipStream_t s = as_stream(stream);
  const int64_t n_pairs = (int64_t)B * C;
  const bool aligned = (reinterpret_cast<uintptr_t>(U) % 16 == 0) &&
                       (reinterpret_cast<uintptr_t>(I) % 16 == 0);
  if (aligned) {
    switch (d) {
      case 16: return launch_gather_dot<16>(U, I, uid, iid, n_pairs, C, pred, s);
      case 32: return launch_gather_dot<32>(U, I, uid, iid, n_pairs, C, pred, s);
      case 64: return launch_gather_dot<64>(U, I, uid, iid, n_pairs, C, pred, s);
      case 128: return launch_gather_dot<128>(U, I, uid, iid, n_pairs, C, pred, s);
      case 256: return launch_gather_dot<256>(U, I, uid, iid, n_pairs, C, pred, s);
      default: break;
    }
  }
  int64_t blocks = (n_pairs + 3) / 4;
  if (blocks > 256 * 32) blocks = 256 * 32;
  hipLaunchKernelGGL(gather_dot_fwd_generic_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, s,
                     U, I, uid, iid, n_pairs, C, d, pred);
  RC_LAUNCH_CHECK();
  return RC_OK;
}

extern "C" int rc_weighted_row_sum(const float* W, const int64_t* ids, const float* coef, int B,
                                   int C, int d, float* out, rc_stream_t stream) {
  if (B == 0) return RC_OK;
  RC_REQUIRE(W && ids && coef && out, "rc_weighted_row_sum: null pointer");
  RC_REQUIRE(B > 0 && C >= 1 && d >= 1, "rc_weighted_row_sum: bad shape B=%d C=%d d=%d", B, C, d);
  hipStream_t s = as_stream(stream);
  const bool aligned = (reinterpret_cast<uintptr_t>(W) % 16 == 0) &&
                       (reinterpret_cast<uintptr_t>(out) % 16 == 0);
  const int blocks = (B + (kBlock / 64) - 1) / (kBlock / 64);
#define RC_WRS(D_)                                                                           \
  hipLaunchKernelGGL((weighted_row_sum_kernel<D_>), dim3(blocks), dim3(kBlock), 0, s, W, ids, \
                     coef, B, C, out);                                                       \
  RC_LAUNCH_CHECK();                                                                         \
  return RC_OK
  if (aligned) {
    switch (d) {
      case 16: RC_WRS(16);
      case 32: RC_WRS(32);
      case 64: RC_WRS(64);
      case 128: RC_WRS(128);
      case 256: RC_WRS(256);
      default: break;
    }
  }
#undef RC_WRS
  int64_t gblocks = ((int64_t)B * d + kBlock - 1) / kBlock;
  if (gblocks > 256 * 32) gblocks = 256 * 32;
  hipLaunchKernelGGL(weighted_row_sum_generic_kernel, dim3((unsigned)gblocks), dim3(kBlock), 0, s, W,
                     ids, coef, B, C, d, out);
  RC_LAUNCH_CHECK();
  return RC_OK;
}
