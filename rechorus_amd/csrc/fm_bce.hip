// fm_bce.hip -- the factorization-machine second-order term over stacked field vectors and the
// binary cross-entropy of the CTR task, forward and closed-form backward.
//
// Reference: FMBase.forward / DeepFMBase.forward (models/context/FM.py:59-63, DeepFM.py:19-23):
//     fm[n] = sum_k 0.5 * ((sum_f v[n,f,k])^2 - sum_f v[n,f,k]^2)
//   d fm[n] / d v[n,f,k] = (sum_f' v[n,f',k]) - v[n,f,k]
// and CTRModel.loss with nn.BCELoss (models/BaseModel.py:259-267) on probabilities p = sigmoid(.):
//     L = mean_i -( y_i * max(log p_i, -100) + (1-y_i) * max(log1p(-p_i), -100) )     (torch clamps the logs)
//   dL/dp_i = (p_i - y_i) / max(p_i (1-p_i), 1e-12) / n                               (torch's backward)
//
// Both are HBM-bound streaming kernels: one lane-group (d/4 lanes, a float4 each) per instance walks
// the F field vectors once (forward) or twice (backward: the field sum, then the per-field gradient).
#include <mutex>

#include "common.hpp"
#include "numeric_grads.hpp"
#include "small_plan.hpp"

namespace rc {

template <int D>
__global__ __launch_bounds__(kBlock) void fm2_fwd_kernel(const float* __restrict__ V, int64_t n, int F,
                                                         float* __restrict__ out) {
  constexpr int LPR = D / 4;
  constexpr int GPB = kBlock / LPR;
  const int l = threadIdx.x % LPR;
  const int64_t i_raw = (int64_t)blockIdx.x * GPB + threadIdx.x / LPR;
  const int64_t i = i_raw < n ? i_raw : n - 1;  // keep every lane in the DPP reduction
  const float4* v = reinterpret_cast<const float4*>(V + i * F * D) + l;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f), q = s;
  for (int f = 0; f < F; ++f) {
    const float4 x = v[f * LPR];
    s.x += x.x; s.y += x.y; s.z += x.z; s.w += x.w;
    q.x = fmaf(x.x, x.x, q.x); q.y = fmaf(x.y, x.y, q.y); q.z = fmaf(x.z, x.z, q.z); q.w = fmaf(x.w, x.w, q.w);
  }
  float part = 0.5f * (s.x * s.x - q.x) + 0.5f * (s.y * s.y - q.y) + 0.5f * (s.z * s.z - q.z) +
               0.5f * (s.w * s.w - q.w);
  part = row_allreduce_sum<LPR>(part);
  if (l == 0 && i_raw < n) out[i] = part;
}

template <int D>
__global__ __launch_bounds__(kBlock) void fm2_bwd_kernel(const float* __restrict__ V, const float* __restrict__ g,
                                                         int64_t n, int F, const float* add, float* dV) {
  constexpr int LPR = D / 4;
  constexpr int GPB = kBlock / LPR;
  const int l = threadIdx.x % LPR;
  const int64_t i = (int64_t)blockIdx.x * GPB + threadIdx.x / LPR;
  if (i >= n) return;  // no cross-lane ops here
  const float4* v = reinterpret_cast<const float4*>(V + i * F * D) + l;
  float4* dv = reinterpret_cast<float4*>(dV + i * F * D) + l;
  if (add) {   // dV = add + d fm2 / dV (the gradient of the same field vectors through another consumer; add may be dV itself)
    const float4* ad = reinterpret_cast<const float4*>(add + i * F * D) + l;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int f = 0; f < F; ++f) {
      const float4 x = v[f * LPR];
      s.x += x.x; s.y += x.y; s.z += x.z; s.w += x.w;
    }
    const float gi = g[i];
    for (int f = 0; f < F; ++f) {
      const float4 x = v[f * LPR];
      const float4 a = ad[f * LPR];
      dv[f * LPR] = make_float4(a.x + gi * (s.x - x.x), a.y + gi * (s.y - x.y), a.z + gi * (s.z - x.z), a.w + gi * (s.w - x.w));
    }
    return;
  }
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int f = 0; f < F; ++f) {
    const float4 x = v[f * LPR];
    s.x += x.x; s.y += x.y; s.z += x.z; s.w += x.w;
  }
  const float gi = g[i];
  for (int f = 0; f < F; ++f) {
    const float4 x = v[f * LPR];
    dv[f * LPR] = make_float4(gi * (s.x - x.x), gi * (s.y - x.y), gi * (s.z - x.z), gi * (s.w - x.w));
  }
}

// any emb_size: one wave per instance, lanes stride over the d components (correctness path for widths
// without a float4 lane-group tiling)
__global__ __launch_bounds__(kBlock) void fm2_fwd_generic_kernel(const float* __restrict__ V, int64_t n, int F, int d,
                                                                 float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int64_t i = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
  if (i >= n) return;  // wave-uniform
  const float* v = V + i * F * d;
  float part = 0.f;
  for (int k = lane; k < d; k += 64) {
    float s = 0.f, q = 0.f;
    for (int f = 0; f < F; ++f) {
      const float x = v[f * d + k];
      s += x;
      q = fmaf(x, x, q);
    }
    part += 0.5f * (s * s - q);
  }
  part = wave_allreduce_sum(part);
  if (lane == 0) out[i] = part;
}

__global__ __launch_bounds__(kBlock) void fm2_bwd_generic_kernel(const float* __restrict__ V, const float* __restrict__ g,
                                                                 int64_t n, int F, int d, const float* add, float* dV) {
  const int lane = threadIdx.x & 63;
  const int64_t i = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
  if (i >= n) return;
  const float* v = V + i * F * d;
  const float* ad = add ? add + i * F * d : nullptr;
  float* dv = dV + i * F * d;
  const float gi = g[i];
  for (int k = lane; k < d; k += 64) {
    float s = 0.f;
    for (int f = 0; f < F; ++f) s += v[f * d + k];
    for (int f = 0; f < F; ++f) dv[f * d + k] = (ad ? ad[f * d + k] : 0.f) + gi * (s - v[f * d + k]);
  }
}

__global__ __launch_bounds__(kBlock) void bce_prob_kernel(const float* __restrict__ p, const float* __restrict__ y,
                                                          int64_t n, float inv_n, float* __restrict__ loss_vec,
                                                          float* __restrict__ gp) {
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
    const float pi = p[i], yi = y[i];
    const float lp = fmaxf(logf(pi), -100.f), lq = fmaxf(log1pf(-pi), -100.f);
    loss_vec[i] = -(yi * lp + (1.0f - yi) * lq);
    if (gp) gp[i] = (pi - yi) / fmaxf((1.0f - pi) * pi, 1e-12f) * inv_n;
  }
}

// The CTR head in one pass (models/context/FM.py:59-60 / DeepFM.py:27 / WideDeep.py:46: prediction = overall_bias + sum_f
// first-order value (+ pairwise term) (+ MLP output); BaseContextModel.py:74-78: sigmoid; BaseModel.py:259-267: nn.BCELoss
// with torch's -100 clamps of the logs): z, p, the per-row loss term and d loss / d z.  Seven elementwise / reduction launches
// of a few microseconds each otherwise -- a tenth of the replayed DeepFM step at B = 1,024.
__global__ __launch_bounds__(kBlock) void ctr_head_kernel(const float* __restrict__ bias, const float* __restrict__ lin, int F,
                                                          const float* __restrict__ t1, const float* __restrict__ t2,
                                                          const int64_t* __restrict__ y, int64_t n, float inv_n,
                                                          float* __restrict__ p_out, float* __restrict__ loss_vec,
                                                          float* __restrict__ gz) {
  const float b0 = bias[0];
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
    float sl = 0.f;
    for (int f = 0; f < F; ++f) sl += lin[i * F + f];
    float z = b0 + sl;
    if (t1) z += t1[i];
    if (t2) z += t2[i];
    const float pi = 1.0f / (1.0f + expf(-z));
    const float yi = (float)y[i];
    const float lp = fmaxf(logf(pi), -100.f), lq = fmaxf(log1pf(-pi), -100.f);
    p_out[i] = pi;
    loss_vec[i] = -(yi * lp + (1.0f - yi) * lq);
    const float gp = (pi - yi) / fmaxf((1.0f - pi) * pi, 1e-12f) * inv_n;   // d loss / d p, as rc_bce_prob_fwd_bwd
    gz[i] = gp * (1.0f - pi) * pi;                                          // sigmoid backward: grad * (1 - out) * out
  }
}

// The same with the two sums the step needs next to it: loss = mean of the loss terms (nn.BCELoss's reduction) and sum gz (the
// gradient of overall_bias), in ONE workgroup for a batch of the reference's size -- a thread walks its rows in ascending order,
// the threads' partials meet in a fixed LDS tree.  (reduce_sum for the loss and torch's sum for the bias were two more launches of
// ~4.7 us in a replayed step whose arithmetic takes 1 us.)
constexpr int kCtrOneWg = 1024;
__global__ __launch_bounds__(kCtrOneWg) void ctr_head_sums_kernel(const float* __restrict__ bias, const float* __restrict__ lin, int F,
                                                                const float* __restrict__ t1, const float* __restrict__ t2,
                                                                const int64_t* __restrict__ y, int64_t n, float inv_n,
                                                                float* __restrict__ p_out, float* __restrict__ loss_vec,
                                                                float* __restrict__ gz, float* __restrict__ sums /* loss mean, sum gz */,
                                                                float* __restrict__ g_lin, float* __restrict__ g_bias, int64_t* __restrict__ bump) {
  __shared__ float red[2][kCtrOneWg];
  const float b0 = bias[0];
  float sl_loss = 0.f, sl_gz = 0.f;
  for (int64_t i = threadIdx.x; i < n; i += kCtrOneWg) {
    float sl = 0.f;
    for (int f = 0; f < F; ++f) sl += lin[i * F + f];
    float z = b0 + sl;
    if (t1) z += t1[i];
    if (t2) z += t2[i];
    const float pi = 1.0f / (1.0f + expf(-z));
    const float yi = (float)y[i];
    const float lp = fmaxf(logf(pi), -100.f), lq = fmaxf(log1pf(-pi), -100.f);
    const float li = -(yi * lp + (1.0f - yi) * lq);
    p_out[i] = pi;
    loss_vec[i] = li;
    const float gp = (pi - yi) / fmaxf((1.0f - pi) * pi, 1e-12f) * inv_n;
    const float gi = gp * (1.0f - pi) * pi;
    gz[i] = gi;
    if (g_lin)      // rc_ctr_head_fwd_full: the backward fan-out for a seed gradient of exactly 1 (gz * 1.0f is gz)
      for (int f = 0; f < F; ++f) g_lin[i * F + f] = gi;
    sl_loss += li;
    sl_gz += gi;
  }
  red[0][threadIdx.x] = sl_loss;
  red[1][threadIdx.x] = sl_gz;
  __syncthreads();
  for (int s = kCtrOneWg / 2; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) {
      red[0][threadIdx.x] += red[0][threadIdx.x + s];
      red[1][threadIdx.x] += red[1][threadIdx.x + s];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    sums[0] = red[0][0] * inv_n;
    sums[1] = red[1][0];
    if (g_bias) g_bias[0] = red[1][0];
    if (bump) bump[0] += 1;     // a device counter nobody reads in this launch (Adam's step count: its reader is the update kernel)
  }
}

// backward fan-out of the head: g = gz * g_loss[0] to the [n] terms, the same value to each of the F first-order weights of a row
// (written out as the contiguous [n, F] block the field-gradient kernels read), d bias = (sum gz) * g_loss[0]
__global__ __launch_bounds__(kBlock) void ctr_head_bwd_kernel(const float* __restrict__ gz, const float* __restrict__ sums,
                                                              const float* __restrict__ g_loss, int64_t n, int F, float* __restrict__ g,
                                                              float* __restrict__ g_lin, float* __restrict__ g_bias) {
  const float gl = g_loss[0];
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
    const float v = gz[i] * gl;
    g[i] = v;
    for (int f = 0; f < F; ++f) g_lin[i * F + f] = v;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) g_bias[0] = sums[1] * gl;
}

}  // namespace rc

using namespace rc;

extern "C" int rc_ctr_head_fwd_bwd_sums(const float* bias, const float* lin, int F, const float* term1, const float* term2,
                                        const int64_t* label, int64_t n, float* p, float* loss_vec, float* gz, float* sums,
                                        rc_stream_t stream) {
  RC_REQUIRE(bias && lin && label && p && loss_vec && gz && sums, "rc_ctr_head_fwd_bwd_sums: null pointer");
  RC_REQUIRE(n > 0 && n <= 65536 && F >= 1, "rc_ctr_head_fwd_bwd_sums: n=%lld (1 .. 65,536 rows: one workgroup) F=%d", (long long)n, F);
  hipLaunchKernelGGL(ctr_head_sums_kernel, dim3(1), dim3(kCtrOneWg), 0, as_stream(stream), bias, lin, F, term1, term2, label, n,
                     1.0f / (float)n, p, loss_vec, gz, sums, (float*)nullptr, (float*)nullptr, (int64_t*)nullptr);
  RC_LAUNCH_CHECK();
  return RC_OK;
}

/* rc_ctr_head_fwd_bwd_sums that also leaves the backward fan-out of a seed gradient of exactly one -- g_lin [n, F] = gz broadcast
 * over a row's first-order weights, g_bias [1] = sum gz (rc_ctr_head_bwd's outputs for g_loss = 1, bit for bit; g itself is gz) --
 * and, optionally, increments a device counter that nothing in this launch reads (bump, may be NULL): a whole training step that
 * seeds loss.backward() with 1 needs no rc_ctr_head_bwd launch and no launch for Adam's step count. */
extern "C" int rc_ctr_head_fwd_full(const float* bias, const float* lin, int F, const float* term1, const float* term2,
                                    const int64_t* label, int64_t n, float* p, float* loss_vec, float* gz, float* sums, float* g_lin,
                                    float* g_bias, int64_t* bump, rc_stream_t stream) {
  RC_REQUIRE(bias && lin && label && p && loss_vec && gz && sums && g_lin && g_bias, "rc_ctr_head_fwd_full: null pointer");
  RC_REQUIRE(n > 0 && n <= 65536 && F >= 1, "rc_ctr_head_fwd_full: n=%lld (1 .. 65,536 rows: one workgroup) F=%d", (long long)n, F);
  hipLaunchKernelGGL(ctr_head_sums_kernel, dim3(1), dim3(kCtrOneWg), 0, as_stream(stream), bias, lin, F, term1, term2, label, n,
                     1.0f / (float)n, p, loss_vec, gz, sums, g_lin, g_bias, bump);
  RC_LAUNCH_CHECK();
  return RC_OK;
}

extern "C" int rc_ctr_head_bwd(const float* gz, const float* sums, const float* g_loss, int64_t n, int F, float* g, float* g_lin,
                               float* g_bias, rc_stream_t stream) {
  RC_REQUIRE(gz && sums && g_loss && g && g_lin && g_bias, "rc_ctr_head_bwd: null pointer");
  RC_REQUIRE(n > 0 && F >= 1, "rc_ctr_head_bwd: bad shape n=%lld F=%d", (long long)n, F);
  int64_t blocks = (n + kBlock - 1) / kBlock;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(ctr_head_bwd_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, as_stream(stream), gz, sums, g_loss, n, F, g, g_lin, g_bias);
  RC_LAUNCH_CHECK();
  return RC_OK;
}

extern "C" int rc_ctr_head_fwd_bwd(const float* bias, const float* lin, int F, const float* term1, const float* term2,
                                   const int64_t* label, int64_t n, float* p, float* loss_vec, float* gz, rc_stream_t stream) {
  if (n == 0) return RC_OK;
  RC_REQUIRE(bias && lin && label && p && loss_vec && gz, "rc_ctr_head_fwd_bwd: null pointer");
  RC_REQUIRE(n > 0 && F >= 1, "rc_ctr_head_fwd_bwd: bad shape n=%lld F=%d", (long long)n, F);
  int64_t blocks = (n + kBlock - 1) / kBlock;
  if (blocks > 256 * 16) blocks = 256 * 16;
  hipLaunchKernelGGL(ctr_head_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, as_stream(stream), bias, lin, F, term1, term2, label, n,
                     1.0f / (float)n, p, loss_vec, gz);
  RC_LAUNCH_CHECK();
  return RC_OK;
}

#define RC_FM_DISPATCH(KERN, ...)                                                                   \
  switch (d) {                                                                                      \
    case 16: hipLaunchKernelGGL((KERN<16>), dim3(blocks(16)), dim3(kBlock), 0, s, __VA_ARGS__); break; \
    case 32: hipLaunchKernelGGL((KERN<32>), dim3(blocks(32)), dim3(kBlock), 0, s, __VA_ARGS__); break; \
    case 64: hipLaunchKernelGGL((KERN<64>), dim3(blocks(64)), dim3(kBlock), 0, s, __VA_ARGS__); break; \
    case 128: hipLaunchKernelGGL((KERN<128>), dim3(blocks(128)), dim3(kBlock), 0, s, __VA_ARGS__); break; \
    default: generic = true; break;                                                                 \
  }

extern "C" int rc_fm_second_order_fwd(const float* V, int64_t n, int F, int d, float* out, rc_stream_t stream) {
  if (n == 0) return RC_OK;
  RC_REQUIRE(V && out, "rc_fm_second_order_fwd: null pointer");
  RC_REQUIRE(n > 0 && F >= 1, "rc_fm_second_order_fwd: bad shape n=%lld F=%d", (long long)n, F);
  RC_REQUIRE(reinterpret_cast<uintptr_t>(V) % 16 == 0 || d % 4 != 0, "rc_fm_second_order_fwd: V must be 16-byte aligned");
  hipStream_t s = as_stream(stream);
  auto blocks = [&](int dd) { return (unsigned)((n + (kBlock / (dd / 4)) - 1) / (kBlock / (dd / 4))); };
  bool generic = false;
  RC_FM_DISPATCH(fm2_fwd_kernel, V, n, F, out);
  if (generic) {
    RC_REQUIRE(d >= 1, "rc_fm_second_order_fwd: bad emb_size %d", d);
    hipLaunchKernelGGL(fm2_fwd_generic_kernel, dim3((unsigned)((n + 3) / 4)), dim3(kBlock), 0, s, V, n, F, d, out);
  }
  RC_LAUNCH_CHECK();
  return RC_OK;
}

static int fm_second_order_bwd_impl(const float* V, const float* gout, int64_t n, int F, int d, const float* add, float* dV,
                                    rc_stream_t stream);

extern "C" int rc_fm_second_order_bwd(const float* V, const float* gout, int64_t n, int F, int d, float* dV,
                                      rc_stream_t stream) {
  return fm_second_order_bwd_impl(V, gout, n, F, d, nullptr, dV, stream);
}

// dV = add + d fm2 / dV: the field vectors' gradient through the FM term on top of their gradient through another consumer
// (the deep tower of DeepFM, models/context/DeepFM.py:19-28) in one pass -- autograd would form the two and add them
extern "C" int rc_fm_second_order_bwd_add(const float* V, const float* gout, int64_t n, int F, int d, const float* add, float* dV,
                                          rc_stream_t stream) {
  RC_REQUIRE(add != nullptr && reinterpret_cast<uintptr_t>(add) % 16 == 0, "rc_fm_second_order_bwd_add: add is null / not 16-byte aligned");
  return fm_second_order_bwd_impl(V, gout, n, F, d, add, dV, stream);
}

static int fm_second_order_bwd_impl(const float* V, const float* gout, int64_t n, int F, int d, const float* add, float* dV,
                                    rc_stream_t stream) {
  if (n == 0) return RC_OK;
  RC_REQUIRE(V && gout && dV, "rc_fm_second_order_bwd: null pointer");
  RC_REQUIRE(n > 0 && F >= 1, "rc_fm_second_order_bwd: bad shape n=%lld F=%d", (long long)n, F);
  RC_REQUIRE(reinterpret_cast<uintptr_t>(V) % 16 == 0 && reinterpret_cast<uintptr_t>(dV) % 16 == 0,
             "rc_fm_second_order_bwd: V and dV must be 16-byte aligned");
  hipStream_t s = as_stream(stream);
  auto blocks = [&](int dd) { return (unsigned)((n + (kBlock / (dd / 4)) - 1) / (kBlock / (dd / 4))); };
  bool generic = false;
  RC_FM_DISPATCH(fm2_bwd_kernel, V, gout, n, F, add, dV);
  if (generic) {
    RC_REQUIRE(d >= 1, "rc_fm_second_order_bwd: bad emb_size %d", d);
    hipLaunchKernelGGL(fm2_bwd_generic_kernel, dim3((unsigned)((n + 3) / 4)), dim3(kBlock), 0, s, V, gout, n, F, d, add, dV);
  }
  RC_LAUNCH_CHECK();
  return RC_OK;
}

extern "C" int rc_bce_prob_fwd_bwd(const float* p, const float* y, int64_t n, float inv_n, float* loss_vec,
                                   float* gp, rc_stream_t stream) {
  if (n == 0) return RC_OK;
  RC_REQUIRE(p && y && loss_vec, "rc_bce_prob_fwd_bwd: null pointer");
  RC_REQUIRE(n > 0, "rc_bce_prob_fwd_bwd: n < 0");
  int64_t blocks = (n + kBlock - 1) / kBlock;
  if (blocks > 256 * 16) blocks = 256 * 16;
  hipLaunchKernelGGL(bce_prob_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, as_stream(stream), p, y, n, inv_n,
                     loss_vec, gp);
  RC_LAUNCH_CHECK();
  return RC_OK;
}

// ---- all F field lookups of a context model in one launch ------------------------------------------------
// FMBase._get_embeddings_FM (models/context/FM.py:44-57) gathers field by field and stacks; here one
// kernel walks a table-pointer array and writes the stacked block out[B, C, F, d] directly, plus the
// composite row id cid[B, C, F] = row_offset[f] + id (the sort key of the backward pass, which then needs
// ONE sort + ONE segmented sum for all F dense gradients instead of F of each).
namespace rc {

struct FieldArgs {
  const float* table[kMaxFields];
  const int64_t* ids[kMaxFields];
  int64_t row_offset[kMaxFields];
  int per_row[kMaxFields];  // 1: ids [B] (user / situation field, broadcast over candidates); 0: ids [B, C]
  const float* table1[kMaxFields];   // rc_gather_fields_pair: the [vocab, 1] tables gathered with the same ids (FM.py:44-57), else unused
  int kind[kMaxFields];     // rc_gather_fields_mixed: RC_FIELD_IDS, or the value type of a numeric field (ids[f] then points at the values)
  int64_t numeric_key;      // what a numeric field's occurrences carry in cid (they are no rows of the virtual table)
  int F;
  int C;
  int d;
  int64_t n;  // B * C
};

// One thread per (row, float4 of the vector) -- or per float where d % 4 != 0 --, walking the F fields: the field index is uniform
// over the wave, so the table / id pointers are scalar loads from the kernel arguments and the lookups of four fields are in flight
// together.  (One thread per output element with the field decoded from the element index indexed the pointer arrays per lane --
// the compiler keeps such an array in scratch -- and paid three 64-bit divisions per element: 93 us for 268 MB out at B = 131,072.)
// MIXED: some field is numeric -- nn.Linear(1, d, bias=False) on the feature's value (FM.py:38-41): its "row" is x * W[:, 0]
// (table[f] = the d weights), its first-order value x * w1; cid carries numeric_key, no row flag is stamped.  The field index
// is uniform over the wave, so the kind test is a scalar branch.
// FMQ > 0 (VEC = 4, d = 4 FMQ in {16, 32, 64, 128}): the thread also keeps the field sum and the sum of squares of its four columns
// and the row's lane-group forms the FM pairwise term 0.5 sum_k ((sum_f v)^2 - sum_f v^2) (FM.py:61) -- the arithmetic of
// fm2_fwd_kernel, order for order, without the second pass over the stacked block -- and writes the field sum S[r, :] out for the
// backward pass (d fm / d v[r, f, :] = S[r, :] - v[r, f, :]).
// block / n_blocks / n_threads: this workgroup's place among the launch's gather workgroups (the launch may hold others).
template <int VEC, bool MIXED, int FMQ>
__device__ __forceinline__ void gather_fields_body(const FieldArgs& a, float* __restrict__ out, int64_t* __restrict__ cid,
                                                   float* __restrict__ out1, int32_t* __restrict__ row_flags,
                                                   const int64_t* __restrict__ step_dev, int step_add, float* __restrict__ fm_out,
                                                   float* __restrict__ fm_sum, unsigned block, unsigned n_blocks, int n_threads) {
  static_assert(FMQ == 0 || VEC == 4, "the FM term rides with the float4 tiling only");
  const int dq = FMQ > 0 ? FMQ : a.d / VEC;
  // rc_gather_fields_pair_mark: every looked-up composite row is stamped with the step's number (the row-flagged dense update,
  // dense_opt.hip, tells the rows of this batch from the rest by it; equal stamps from duplicate ids race benignly)
  const int32_t gen = row_flags ? (int32_t)(*step_dev + step_add) : 0;
  const int64_t total = a.n * dq;
  for (int64_t e = (int64_t)block * n_threads + threadIdx.x; e < total; e += (int64_t)n_blocks * n_threads) {
    const int64_t r = e / dq;            // b * C + c
    const int q = (int)(e - r * dq);
    const int64_t rb = r / a.C;          // b (fields given per row)
    constexpr int U = 4;
    float4 fs = make_float4(0.f, 0.f, 0.f, 0.f), fq = fs;
    for (int f0 = 0; f0 < a.F; f0 += U) {
      int64_t id[U];
      float xv[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int f = f0 + u < a.F ? f0 + u : a.F - 1;   // (uniform)
        const int64_t at = a.per_row[f] ? rb : r;
        if (MIXED && a.kind[f] != RC_FIELD_IDS) {
          xv[u] = field_value(a.kind[f], a.ids[f], at);
          id[u] = 0;            // the d weights are row 0 of the field's "table"
        } else {
          xv[u] = 1.0f;
          id[u] = a.ids[f][at];
        }
      }
      if (VEC == 4) {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int f = f0 + u < a.F ? f0 + u : a.F - 1;
          v[u] = reinterpret_cast<const float4*>(a.table[f])[id[u] * dq + q];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          if (MIXED && a.kind[f0 + u < a.F ? f0 + u : a.F - 1] != RC_FIELD_IDS) {
            v[u].x *= xv[u]; v[u].y *= xv[u]; v[u].z *= xv[u]; v[u].w *= xv[u];
          }
          if (f0 + u < a.F) {
            reinterpret_cast<float4*>(out)[(r * a.F + f0 + u) * dq + q] = v[u];
            if (FMQ > 0) {
              const float4 x = v[u];
              fs.x += x.x; fs.y += x.y; fs.z += x.z; fs.w += x.w;
              fq.x = fmaf(x.x, x.x, fq.x); fq.y = fmaf(x.y, x.y, fq.y); fq.z = fmaf(x.z, x.z, fq.z); fq.w = fmaf(x.w, x.w, fq.w);
            }
          }
        }
      } else {
        float v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int f = f0 + u < a.F ? f0 + u : a.F - 1;
          v[u] = a.table[f][id[u] * dq + q];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          if (MIXED && a.kind[f0 + u < a.F ? f0 + u : a.F - 1] != RC_FIELD_IDS) v[u] *= xv[u];
          if (f0 + u < a.F) out[(r * a.F + f0 + u) * dq + q] = v[u];
        }
      }
      if (q == 0 && cid) {
#pragma unroll
        for (int u = 0; u < U; ++u)
          if (f0 + u < a.F)
            cid[r * a.F + f0 + u] = (MIXED && a.kind[f0 + u] != RC_FIELD_IDS) ? a.numeric_key : a.row_offset[f0 + u] + id[u];
      }
      if (q == 0 && row_flags) {
#pragma unroll
        for (int u = 0; u < U; ++u)
          if (f0 + u < a.F && !(MIXED && a.kind[f0 + u] != RC_FIELD_IDS)) row_flags[a.row_offset[f0 + u] + id[u]] = gen;
      }
      if (q == 0 && out1) {   // the first-order weights of the same ids: [n, F]
        float w1[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int f = f0 + u < a.F ? f0 + u : a.F - 1;
          w1[u] = a.table1[f][id[u]];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          if (MIXED && a.kind[f0 + u < a.F ? f0 + u : a.F - 1] != RC_FIELD_IDS) w1[u] *= xv[u];
          if (f0 + u < a.F) out1[r * a.F + f0 + u] = w1[u];
        }
      }
    }
    if (FMQ > 0) {
      float part = 0.5f * (fs.x * fs.x - fq.x) + 0.5f * (fs.y * fs.y - fq.y) + 0.5f * (fs.z * fs.z - fq.z) + 0.5f * (fs.w * fs.w - fq.w);
      part = row_allreduce_sum<(FMQ > 0 ? FMQ : 1)>(part);   // (a row's lanes are all in the loop together: total = n * FMQ)
      if (q == 0) fm_out[r] = part;
      reinterpret_cast<float4*>(fm_sum)[r * dq + q] = fs;
    }
  }
}

template <int VEC, bool MIXED>
__global__ __launch_bounds__(kBlock) void gather_fields_kernel(FieldArgs a, float* __restrict__ out,
                                                               int64_t* __restrict__ cid, float* __restrict__ out1,
                                                               int32_t* __restrict__ row_flags, const int64_t* __restrict__ step_dev,
                                                               int step_add) {
  gather_fields_body<VEC, MIXED, 0>(a, out, cid, out1, row_flags, step_dev, step_add, nullptr, nullptr, blockIdx.x, gridDim.x, kBlock);
}

// ---- the gather of a SMALL batch with the backward pass's grouping in the same launch -------------------------------------
// At the reference's own batch size (B = 1,024, CTR_MIND.sh:8) a replayed step is a chain of launches of ~5-10 us each, so work that
// depends on the batch's ids alone should not be a link of its own: the 128 plan workgroups of the small route (small_plan.hpp:
// group the composite (field, id) keys by row, what embedding_dense_backward's sort does) run BESIDE the gather workgroups here and
// leave the plan in the workspace the backward pass's row sums read (rc_small_row_sums_planned).  The plan workgroups form the keys
// from the fields' id tensors themselves -- position p = r F + f, key = row_offset[f] + ids[f][r or r / C], kSmallSkipKey for a
// numeric field --, a chunk of 64 rows of one field at a time, so that the field's descriptors are scalar loads from the kernel
// arguments (a per-lane index into those arrays would put them in scratch).
// chunk c = 64 consecutive rows of ONE field (f = c / chunks-per-field, wave-uniform: the field's id pointer, row offset and kind are
// scalar loads from the kernel arguments, the id loads coalesced), recorded position r * F + f (the occurrence's row in the
// [n, F, .] gradient blocks); a key belongs to one field, so its positions ascend with the scan
struct FieldPlanKey {
  const FieldArgs* fa;
  uint32_t cpf;      // chunks per field = ceil(n / 64)
  uint32_t magic_c;  // r / C by one v_mul_hi_u32
  struct Ctx {       // one field's descriptors (scalar registers)
    const int64_t* ids;
    uint32_t row_offset, f, first_chunk;
    bool per_row, numeric;
  };
  __device__ __forceinline__ uint32_t chunks(const SmallPlanArgs&) const { return cpf * (uint32_t)fa->F; }
  __device__ __forceinline__ uint32_t group_end(const SmallPlanArgs&, uint32_t c) const {
    const uint32_t cu = (uint32_t)__builtin_amdgcn_readfirstlane((int)c);
    return (cu / cpf + 1u) * cpf;
  }
  __device__ __forceinline__ Ctx open(const SmallPlanArgs&, uint32_t c) const {
    const uint32_t cu = (uint32_t)__builtin_amdgcn_readfirstlane((int)c);
    Ctx x;
    x.f = cu / cpf;
    x.first_chunk = x.f * cpf;
    x.ids = fa->ids[x.f];
    x.row_offset = (uint32_t)fa->row_offset[x.f];
    x.per_row = fa->per_row[x.f] != 0;
    x.numeric = fa->kind[x.f] != RC_FIELD_IDS;
    return x;
  }
  __device__ __forceinline__ uint32_t key(const SmallPlanArgs&, const Ctx& x, uint32_t c, uint32_t lane) const {
    const uint32_t r = (c - x.first_chunk) * 64u + lane;
    if (r >= (uint32_t)fa->n || x.numeric) return kSmallSkipKey;
    return x.row_offset + (uint32_t)x.ids[x.per_row ? small_div(r, magic_c) : r];
  }
  __device__ __forceinline__ uint32_t pos(const SmallPlanArgs&, const Ctx& x, uint32_t c, uint32_t lane) const {
    return ((c - x.first_chunk) * 64u + lane) * (uint32_t)fa->F + x.f;
  }
};

template <bool MIXED, int FMQ>
__global__ __launch_bounds__(kSmallThreads) void gather_fields_plan_kernel(FieldArgs a, float* __restrict__ out, int64_t* __restrict__ cid,
                                                                           float* __restrict__ out1, int32_t* __restrict__ row_flags,
                                                                           const int64_t* __restrict__ step_dev, int step_add,
                                                                           float* __restrict__ fm_out, float* __restrict__ fm_sum,
                                                                           unsigned gather_blocks, SmallPlanArgs plan, int64_t* __restrict__ bump) {
  // a device counter nothing in this launch reads (the deep tower's dropout seed: its first reader is the tower's first kernel)
  if (bump != nullptr && blockIdx.x == 0 && threadIdx.x == 0) bump[0] += 1;
  if (blockIdx.x < gather_blocks) {   // (workgroup-uniform)
    gather_fields_body<4, MIXED, FMQ>(a, out, cid, out1, row_flags, step_dev, step_add, fm_out, fm_sum, blockIdx.x, gather_blocks, kSmallThreads);
    return;
  }
  extern __shared__ __attribute__((aligned(16))) unsigned char gather_plan_smem[];
  FieldPlanKey key;
  key.fa = &a;
  key.cpf = ((uint32_t)a.n + 63u) >> 6;
  key.magic_c = small_div_magic((uint32_t)a.C);      // r < n <= 32,768 rows, C <= n: r * C < 2^32
  small_plan_block<kSmallCapBig, kSmallWaveCapBig, FieldPlanKey>(plan, blockIdx.x - gather_blocks, gather_plan_smem, key);
}

}  // namespace rc

// the FM term and / or the small route's plan inside the gather launch (rc_gather_fields_fused)
template <bool MIXED, int FMQ>
static int gather_fused_launch_t(const FieldArgs& a, float* out, int64_t* cid, float* out1, int32_t* row_flags, const int64_t* step_dev,
                                 int step_add, float* fm_out, float* fm_sum, unsigned gather_blocks, const SmallPlanArgs* plan, int64_t* bump,
                                 hipStream_t s) {
  auto kern = gather_fields_plan_kernel<MIXED, FMQ>;
  SmallPlanArgs p;
  memset(&p, 0, sizeof(p));
  size_t lds = 0;
  unsigned grid = gather_blocks;
  if (plan != nullptr) {
    p = *plan;
    lds = kSmallLdsBytesBig;
    grid += kSmallPlanWgs;
    static std::mutex mu;       // function attributes are per device: once per device of the process
    static bool done[64] = {};
    int dev = 0;
    RC_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(mu);
    if (dev < 0 || dev >= 64 || !done[dev]) {
      RC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kSmallLdsBytesBig));
      if (dev >= 0 && dev < 64) done[dev] = true;
    }
  }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(kSmallThreads), lds, s, a, out, cid, out1, row_flags, step_dev, step_add, fm_out, fm_sum, gather_blocks, p, bump);
  RC_LAUNCH_CHECK();
  return RC_OK;
}

static int gather_fields_fused_launch(const FieldArgs& a, bool vec, bool mixed, float* out, int64_t* cid, float* out1, int32_t* row_flags,
                                      const int64_t* step_dev, int step_add, float* fm_out, float* fm_sum, void* plan_ws,
                                      size_t plan_ws_bytes, int64_t* bump, rc_stream_t stream) {
  const int d = a.d;
  RC_REQUIRE(bump == nullptr || bump != step_dev, "rc_gather_fields_fused: the gather reads step_dev, it cannot be the counter to increment");
  RC_REQUIRE(vec && (d == 16 || d == 32 || d == 64 || d == 128), "rc_gather_fields_fused: d = %d (16 / 32 / 64 / 128, 16-byte aligned tables and output)", d);
  RC_REQUIRE((fm_out == nullptr) == (fm_sum == nullptr), "rc_gather_fields_fused: the FM term and the field sums come together");
  RC_REQUIRE(fm_sum == nullptr || reinterpret_cast<uintptr_t>(fm_sum) % 16 == 0, "rc_gather_fields_fused: fm_sum must be 16-byte aligned");
  const int64_t n_keys = a.n * a.F;
  SmallPlanArgs p;
  memset(&p, 0, sizeof(p));
  if (plan_ws != nullptr) {
    int64_t n_rows = 0;
    for (int f = 0; f < a.F; ++f) n_rows = a.row_offset[f] > n_rows ? a.row_offset[f] : n_rows;
    RC_REQUIRE(n_rows < ((int64_t)1 << 32) - 1, "rc_gather_fields_fused: the concatenated table has too many rows for 32-bit keys");
    if (!rc_small_row_sums_supported(n_keys, 1, d))
      return fail(RC_ERR_UNSUPPORTED, "rc_gather_fields_fused: %lld keys are not a small batch (<= %d) or the LDS of this device is too small", (long long)n_keys, kSmallMaxKeys);
    if (plan_ws_bytes < rc_small_row_sums_workspace_bytes(n_keys))
      return fail(RC_ERR_WORKSPACE, "rc_gather_fields_fused: plan workspace %zu < %zu", plan_ws_bytes, rc_small_row_sums_workspace_bytes(n_keys));
    Carver cv(plan_ws);     // the layout rc_small_row_sums_planned reads (small_step.hip)
    p.rows = cv.take<rc_plan_row>((size_t)kSmallPlanWgs * (size_t)n_keys);
    p.occ = cv.take<uint32_t>((size_t)kSmallPlanWgs * (size_t)n_keys);
    p.cnt = cv.take<SmallCnt>(kSmallPlanWgs);
    p.n_a = (uint32_t)n_keys; p.n = (uint32_t)n_keys; p.base_b = 0xFFFFFFFFu;     // one list: every key is a row of it
  }
  const int64_t total = a.n * (d / 4);
  int64_t blocks = (total + kSmallThreads - 1) / kSmallThreads;
  if (blocks > 256 * 16) blocks = 256 * 16;
  hipStream_t s = as_stream(stream);
  const SmallPlanArgs* pp = plan_ws != nullptr ? &p : nullptr;
  const bool fm = fm_out != nullptr;
#define RC_GF(M_, Q_) return gather_fused_launch_t<M_, Q_>(a, out, cid, out1, row_flags, step_dev, step_add, fm_out, fm_sum, (unsigned)blocks, pp, bump, s)
  if (!fm) { if (mixed) RC_GF(true, 0); RC_GF(false, 0); }
  switch (d) {
    case 16: if (mixed) RC_GF(true, 4); RC_GF(false, 4);
    case 32: if (mixed) RC_GF(true, 8); RC_GF(false, 8);
    case 64: if (mixed) RC_GF(true, 16); RC_GF(false, 16);
    default: if (mixed) RC_GF(true, 32); RC_GF(false, 32);
  }
#undef RC_GF
}

static int gather_fields_impl(const float* const* tables, const float* const* tables1, const void* const* ids, const int* per_row,
                              const int* kind, int64_t numeric_key, const int64_t* row_offset, int F, int64_t B, int C, int d, float* out,
                              float* out1, int64_t* cid, int32_t* row_flags, const int64_t* step_dev, int step_add, rc_stream_t stream,
                              float* fm_out = nullptr, float* fm_sum = nullptr, void* plan_ws = nullptr, size_t plan_ws_bytes = 0,
                              int64_t* bump = nullptr) {
  if (B == 0) return RC_OK;
  RC_REQUIRE((row_flags == nullptr) == (step_dev == nullptr), "rc_gather_fields_pair_mark: the row flags and the step count come together");
  RC_REQUIRE(tables && ids && per_row && row_offset && out, "rc_gather_fields: null pointer");
  RC_REQUIRE((tables1 == nullptr) == (out1 == nullptr), "rc_gather_fields_pair: the second table family and its output come together");
  RC_REQUIRE(F >= 1 && F <= kMaxFields, "rc_gather_fields: F must be in [1, %d], got %d", kMaxFields, F);
  RC_REQUIRE(B > 0 && C >= 1 && d >= 1, "rc_gather_fields: bad shape B=%lld C=%d d=%d", (long long)B, C, d);
  FieldArgs a;
  memset(&a, 0, sizeof(a));
  bool vec = d % 4 == 0 && reinterpret_cast<uintptr_t>(out) % 16 == 0;
  bool mixed = false;
  for (int f = 0; f < F; ++f) {
    RC_REQUIRE(tables[f] && ids[f] && (!tables1 || tables1[f]), "rc_gather_fields: null table / ids for field %d", f);
    a.table[f] = tables[f];
    a.table1[f] = tables1 ? tables1[f] : nullptr;
    a.ids[f] = static_cast<const int64_t*>(ids[f]);
    a.row_offset[f] = row_offset[f];
    a.per_row[f] = per_row[f];
    a.kind[f] = kind ? kind[f] : RC_FIELD_IDS;
    RC_REQUIRE(a.kind[f] >= RC_FIELD_IDS && a.kind[f] <= RC_FIELD_I64, "rc_gather_fields_mixed: kind[%d] = %d is no rc_field_kind", f, a.kind[f]);
    mixed = mixed || a.kind[f] != RC_FIELD_IDS;
    vec = vec && reinterpret_cast<uintptr_t>(tables[f]) % 16 == 0;
  }
  a.numeric_key = numeric_key;
  a.F = F; a.C = C; a.d = d; a.n = B * C;
  if (fm_out != nullptr || plan_ws != nullptr) return gather_fields_fused_launch(a, vec, mixed, out, cid, out1, row_flags, step_dev, step_add, fm_out, fm_sum, plan_ws, plan_ws_bytes, bump, stream);
  const int64_t total = a.n * (vec ? d / 4 : d);   // one thread per (row, float4 | float), walking the fields
  int64_t blocks = (total + kBlock - 1) / kBlock;
  if (blocks > 256 * 32) blocks = 256 * 32;
  void (*kern)(FieldArgs, float*, int64_t*, float*, int32_t*, const int64_t*, int) =
      vec ? (mixed ? gather_fields_kernel<4, true> : gather_fields_kernel<4, false>)
          : (mixed ? gather_fields_kernel<1, true> : gather_fields_kernel<1, false>);
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(kBlock), 0, as_stream(stream), a, out, cid, out1, row_flags, step_dev, step_add);
  RC_LAUNCH_CHECK();
  return RC_OK;
}

extern "C" int rc_gather_fields(const float* const* tables, const int64_t* const* ids, const int* per_row,
                                const int64_t* row_offset, int F, int64_t B, int C, int d, float* out, int64_t* cid,
                                rc_stream_t stream) {
  return gather_fields_impl(tables, nullptr, reinterpret_cast<const void* const*>(ids), per_row, nullptr, -1, row_offset, F, B, C, d, out, nullptr, cid, nullptr, nullptr, 0, stream);
}

extern "C" int rc_gather_fields_pair(const float* const* tables, const float* const* tables1, const int64_t* const* ids, const int* per_row,
                                     const int64_t* row_offset, int F, int64_t B, int C, int d, float* out, float* out1, int64_t* cid,
                                     rc_stream_t stream) {
  RC_REQUIRE(tables1 && out1, "rc_gather_fields_pair: null pointer");
  return gather_fields_impl(tables, tables1, reinterpret_cast<const void* const*>(ids), per_row, nullptr, -1, row_offset, F, B, C, d, out, out1, cid, nullptr, nullptr, 0, stream);
}

extern "C" int rc_gather_fields_pair_mark(const float* const* tables, const float* const* tables1, const int64_t* const* ids,
                                          const int* per_row, const int64_t* row_offset, int F, int64_t B, int C, int d, float* out,
                                          float* out1, int64_t* cid, int32_t* row_flags, const int64_t* step_dev, int step_add,
                                          rc_stream_t stream) {
  RC_REQUIRE(tables1 && out1 && row_flags && step_dev, "rc_gather_fields_pair_mark: null pointer");
  return gather_fields_impl(tables, tables1, reinterpret_cast<const void* const*>(ids), per_row, nullptr, -1, row_offset, F, B, C, d, out, out1, cid, row_flags, step_dev, step_add, stream);
}

extern "C" int rc_gather_fields_mixed(const float* const* tables, const float* const* tables1, const void* const* ids,
                                      const int* per_row, const int* kind, int64_t numeric_key, const int64_t* row_offset, int F,
                                      int64_t B, int C, int d, float* out, float* out1, int64_t* cid, int32_t* row_flags,
                                      const int64_t* step_dev, int step_add, rc_stream_t stream) {
  RC_REQUIRE(kind, "rc_gather_fields_mixed: null pointer");
  return gather_fields_impl(tables, tables1, ids, per_row, kind, numeric_key, row_offset, F, B, C, d, out, out1, cid, row_flags, step_dev,
                            step_add, stream);
}

/* rc_gather_fields_mixed with what the rest of a small step needs from the same pass (each part optional):
 *   fm_out [B * C], fm_sum [B * C, d]   the FM pairwise term of every row (rc_fm_second_order_fwd's value, bit for bit) and the field
 *                                       sum its backward needs -- no second pass over the stacked block;
 *   plan_ws                             the grouping of the composite keys (rc_small_row_sums' first launch) by 128 workgroups
 *                                       beside the gather's, left where rc_small_row_sums_planned reads it.
 *   bump                                a device counter incremented by one thread of the launch (nothing in it may read it).
 * d in {16, 32, 64, 128}; the plan: B * C * F <= 32,768 keys.  kind may be null (every field a table). */
extern "C" int rc_gather_fields_fused(const float* const* tables, const float* const* tables1, const void* const* ids,
                                      const int* per_row, const int* kind, int64_t numeric_key, const int64_t* row_offset, int F,
                                      int64_t B, int C, int d, float* out, float* out1, int64_t* cid, int32_t* row_flags,
                                      const int64_t* step_dev, int step_add, float* fm_out, float* fm_sum, void* plan_ws,
                                      size_t plan_ws_bytes, int64_t* bump, rc_stream_t stream) {
  RC_REQUIRE(fm_out != nullptr || plan_ws != nullptr, "rc_gather_fields_fused: neither the FM term nor the plan was asked for");
  return gather_fields_impl(tables, tables1, ids, per_row, kind, numeric_key, row_offset, F, B, C, d, out, out1, cid, row_flags, step_dev,
                            step_add, stream, fm_out, fm_sum, plan_ws, plan_ws_bytes, bump);
}

// ---- weight gradients of the numeric fields ---------------------------------------------------------------------------
// A numeric field is nn.Linear(1, d, bias=False) applied to the feature's value (models/context/FM.py:38-41,47-48); autograd's
// Linear backward gives dW[:, 0] = sum_n x[n] * gV[n, f, :] and, for the first-order Linear(1, 1), dw1 = sum_n x[n] * gL[n, f].
// Weighted column sums over a strided slice of the per-occurrence gradient blocks: chunks of kNumericChunk rows per workgroup,
// every lane-group its rows in ascending order, the groups combined in a fixed order through LDS, the chunks by a second
// launch in ascending order (a batch of up to kNumericChunk rows: one launch writes the gradients themselves).  No atomics.
namespace rc {

struct NumericGradArgs {
  NumericCommon c;
  const void* values[kMaxFields];   // per numeric slot j
  float* dW[kMaxFields];            // [d]
  float* dw1[kMaxFields];           // [1]
  int kind[kMaxFields];
  int per_row[kMaxFields];
  int field[kMaxFields];            // the slot's field index in [0, F)
};

constexpr int kNumericThreads = 1024;   // sixteen waves: 64 rows of a 64-float field vector per step, a 1,024-row chunk in two trips of eight

template <int VEC>
__global__ __launch_bounds__(kNumericThreads) void numeric_field_grads_kernel(NumericGradArgs a) {
  const int j = blockIdx.y;
  NumericSlot sl;
  sl.values = a.values[j]; sl.dW = a.dW[j]; sl.dw1 = a.dw1[j]; sl.kind = a.kind[j]; sl.per_row = a.per_row[j]; sl.field = a.field[j];
  numeric_slot_grads<VEC, kNumericThreads, 8>(sl, a.c, j, blockIdx.x, gridDim.x == 1 && a.c.n <= kNumericChunk);
}

// chunk partials -> gradients, ascending chunk order; one thread per (slot, column)
__global__ __launch_bounds__(kBlock) void numeric_field_reduce_kernel(NumericGradArgs a, int chunks) {
  const int w = a.c.d + 1;
  const int t = blockIdx.x * kBlock + threadIdx.x;
  if (t >= a.c.n_numeric * w) return;
  const int j = t / w, c = t - j * w;
  if (c == a.c.d ? a.c.gL == nullptr : a.c.gV == nullptr) return;
  float s = 0.f;
  for (int k = 0; k < chunks; ++k) s += a.c.part[((size_t)k * a.c.n_numeric + j) * w + c];
  if (c == a.c.d) a.dw1[j][0] = s;
  else a.dW[j][c] = s;
}

}  // namespace rc

extern "C" size_t rc_numeric_field_grads_workspace_bytes(int64_t n, int n_numeric, int d) {
  if (n < 1 || n_numeric < 1 || d < 1) return 256;
  const int64_t chunks = (n + kNumericChunk - 1) / kNumericChunk;
  return align_up((size_t)chunks * (size_t)n_numeric * (size_t)(d + 1) * sizeof(float), 256);
}

extern "C" int rc_numeric_field_grads(const float* gV, const float* gL, const void* const* values, const int* per_row, const int* kind,
                                      const int* field, int n_numeric, int F, int64_t B, int C, int d, float* const* dW,
                                      float* const* dw1, void* ws, size_t ws_bytes, rc_stream_t stream) {
  if (n_numeric == 0) return RC_OK;
  RC_REQUIRE(values && per_row && kind && field && (gV || gL), "rc_numeric_field_grads: null pointer");
  RC_REQUIRE((gV == nullptr || dW) && (gL == nullptr || dw1), "rc_numeric_field_grads: a gradient block without its outputs");
  RC_REQUIRE(n_numeric >= 1 && n_numeric <= F && F <= kMaxFields, "rc_numeric_field_grads: n_numeric=%d of F=%d fields (<= %d)", n_numeric, F, kMaxFields);
  RC_REQUIRE(B >= 0 && C >= 1 && d >= 1, "rc_numeric_field_grads: bad shape B=%lld C=%d d=%d", (long long)B, C, d);
  NumericGradArgs a;
  memset(&a, 0, sizeof(a));
  hipStream_t s = as_stream(stream);
  for (int j = 0; j < n_numeric; ++j) {
    RC_REQUIRE(values[j] && (!gV || dW[j]) && (!gL || dw1[j]), "rc_numeric_field_grads: null pointer for numeric field %d", j);
    RC_REQUIRE(field[j] >= 0 && field[j] < F, "rc_numeric_field_grads: field[%d] = %d outside [0, %d)", j, field[j], F);
    RC_REQUIRE(kind[j] >= RC_FIELD_F32 && kind[j] <= RC_FIELD_I64, "rc_numeric_field_grads: kind[%d] = %d is no numeric rc_field_kind", j, kind[j]);
    a.values[j] = values[j]; a.kind[j] = kind[j]; a.per_row[j] = per_row[j]; a.field[j] = field[j];
    a.dW[j] = dW ? dW[j] : nullptr; a.dw1[j] = dw1 ? dw1[j] : nullptr;
  }
  a.c.gV = gV; a.c.gL = gL; a.c.n = B * C; a.c.n_numeric = n_numeric; a.c.F = F; a.c.C = C; a.c.d = d;
  if (a.c.n == 0) {   // an empty batch: zero gradients
    for (int j = 0; j < n_numeric; ++j) {
      if (gV) RC_HIP(hipMemsetAsync(a.dW[j], 0, (size_t)d * sizeof(float), s));
      if (gL) RC_HIP(hipMemsetAsync(a.dw1[j], 0, sizeof(float), s));
    }
    return RC_OK;
  }
  const int64_t chunks = (a.c.n + kNumericChunk - 1) / kNumericChunk;
  RC_REQUIRE(a.c.n < ((int64_t)1 << 31), "rc_numeric_field_grads: too many rows");
  if (chunks > 1) {
    RC_REQUIRE(ws != nullptr && ws_bytes >= rc_numeric_field_grads_workspace_bytes(a.c.n, n_numeric, d), "rc_numeric_field_grads: workspace %zu < %zu",
               ws_bytes, rc_numeric_field_grads_workspace_bytes(a.c.n, n_numeric, d));
    a.c.part = static_cast<float*>(ws);
  }
  const bool vec = d % 4 == 0 && (gV == nullptr || reinterpret_cast<uintptr_t>(gV) % 16 == 0);
  if (vec)
    hipLaunchKernelGGL((numeric_field_grads_kernel<4>), dim3((unsigned)chunks, (unsigned)n_numeric), dim3(kNumericThreads), 0, s, a);
  else
    hipLaunchKernelGGL((numeric_field_grads_kernel<1>), dim3((unsigned)chunks, (unsigned)n_numeric), dim3(kNumericThreads), 0, s, a);
  RC_LAUNCH_CHECK();
  if (chunks > 1) {
    const int total = n_numeric * (d + 1);
    hipLaunchKernelGGL(numeric_field_reduce_kernel, dim3((unsigned)((total + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, a, (int)chunks);
    RC_LAUNCH_CHECK();
  }
  return RC_OK;
}

// ---- point-wise BCE over a ranking list (ContextModel.loss, loss_n == 'BCE': models/BaseContextModel.py:53-56)
//   p = sigmoid(prediction);  loss = -mean_b [ log p_b0 + sum_{k>=1} log(1 - p_bk) ]
// computed through the sigmoid like the reference (not the softplus form), gradient as autograd derives it:
//   d/dx_b0 = -(1 - p_b0) / B,   d/dx_bk = p_bk / B.   One wave per row.
namespace rc {
__global__ __launch_bounds__(kBlock) void bce_ranking_kernel(const float* __restrict__ pred, int64_t B, int C,
                                                             float inv_b, float* __restrict__ loss_vec,
                                                             float* __restrict__ gpred) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
  if (row >= B) return;  // wave-uniform
  const float* x = pred + row * C;
  float acc = 0.f;
  for (int c = lane; c < C; c += 64) {
    const float p = sigmoidf_(x[c]);
    acc += c == 0 ? logf(p) : logf(1.0f - p);
    if (gpred) gpred[row * C + c] = (c == 0 ? -(1.0f - p) : p) * inv_b;
  }
  acc = wave_allreduce_sum(acc);
  if (lane == 0) loss_vec[row] = -acc;
}
}  // namespace rc

extern "C" int rc_bce_ranking_fwd_bwd(const float* pred, int64_t B, int C, float inv_b, float* loss_vec, float* gpred,
                                      rc_stream_t stream) {
  if (B == 0) return RC_OK;
  RC_REQUIRE(pred && loss_vec, "rc_bce_ranking_fwd_bwd: null pointer");
  RC_REQUIRE(B > 0 && C >= 1, "rc_bce_ranking_fwd_bwd: bad shape B=%lld C=%d", (long long)B, C);
  const int64_t blocks = (B + (kBlock / 64) - 1) / (kBlock / 64);
  RC_REQUIRE(blocks <= kMaxGridX, "rc_bce_ranking_fwd_bwd: too many rows");
  hipLaunchKernelGGL(bce_ranking_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, as_stream(stream), pred, B, C, inv_b,
                     loss_vec, gpred);
  RC_LAUNCH_CHECK();
  return RC_OK;
}
