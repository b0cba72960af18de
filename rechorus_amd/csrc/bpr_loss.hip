// bpr_loss.hip -- GeneralModel.loss (reference: models/BaseModel.py:182-185) forward and
// its closed-form gradient, plus the deterministic batch reduction.
//
//   pos = p[:,0]; neg = p[:,1:];  w = softmax(neg - max(neg));  s_k = sigmoid(pos - neg_k)
//   P = sum_k w_k s_k;  L_b = -log(clamp(P, 1e-8, 1-1e-8))
//   dL/dP   = -inv_b / P   (0 outside the clamp range)
//   dP/dpos = sum_k w_k s_k (1 - s_k)
//   dP/dneg_j = -w_j s_j (1 - s_j) + w_j (s_j - P)
// The reference subtracts the *global* max before softmax (BaseModel.py:184); softmax is
// shift invariant so the per-row max used here gives the same value to fp32 round-off.
#include "bpr_math.hpp"
#include "common.hpp"

namespace rc {

// one wave per row; lanes stride over the negatives (any C >= 2)
__global__ __launch_bounds__(kBlock) void bpr_loss_kernel(const float* __restrict__ pred,
                                                          int B, int C, float inv_b,
                                                          float* __restrict__ loss_vec,
                                                          float* __restrict__ gpred) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
  if (row >= B) return;  // whole wave exits together
  const float* p = pred + row * C;
  const float pos = p[0];
  float mx = -INFINITY;
  for (int c = 1 + lane; c < C; c += 64) mx = fmaxf(mx, p[c]);
  mx = wave_allreduce_max(mx);
  float se = 0.f;
  for (int c = 1 + lane; c < C; c += 64) se += expf(p[c] - mx);
  se = wave_allreduce_sum(se);
  const float inv_se = 1.0f / se;
  float P = 0.f, A = 0.f;
  for (int c = 1 + lane; c < C; c += 64) {
    const float w = expf(p[c] - mx) * inv_se;
    const float s = sigmoidf_(pos - p[c]);
    P = fmaf(w, s, P);
    A = fmaf(w, s * (1.0f - s), A);
  }
  P = wave_allreduce_sum(P);
  A = wave_allreduce_sum(A);
  const BprRow br = bpr_row(P, inv_b);
  if (lane == 0) loss_vec[row] = br.loss;
  if (gpred) {
    float* g = gpred + row * C;
    if (lane == 0) g[0] = br.dLdP * A;
    for (int c = 1 + lane; c < C; c += 64) {
      const float w = expf(p[c] - mx) * inv_se;
      const float s = sigmoidf_(pos - p[c]);
      g[c] = br.dLdP * bpr_dP_dneg(w, s, P);
    }
  }
}

// Few negatives (C - 1 <= 16: the reference's default --num_neg 1, NeuMF's K = 4): sixteen lanes per row, four rows per wave --
// a wave per row left 59 of 64 lanes idle (32 us for the [65536, 5] predictions of a NeuMF step).  The reductions are the xor
// butterfly over the low four lane bits, i.e. the last four steps of the wave version, whose first two steps add zeros here:
// same results bit for bit.
__global__ __launch_bounds__(kBlock) void bpr_loss_small_kernel(const float* __restrict__ pred, int B, int C, float inv_b,
                                                                float* __restrict__ loss_vec, float* __restrict__ gpred) {
  const int l = threadIdx.x & 15;
  const int64_t row = (int64_t)blockIdx.x * (kBlock / 16) + (threadIdx.x >> 4);
  const bool live = row < B;            // (the shuffles below need every lane of the wave)
  const float* p = pred + (live ? row : 0) * C;
  const bool on = live && 1 + l < C;
  const float pos = p[0];
  const float x = on ? p[1 + l] : 0.f;
  float mx = on ? x : -INFINITY;
#pragma unroll
  for (int off = 8; off >= 1; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
  const float e = on ? expf(x - mx) : 0.f;
  float se = e;
#pragma unroll
  for (int off = 8; off >= 1; off >>= 1) se += __shfl_xor(se, off, 64);
  const float inv_se = 1.0f / se;
  const float w = e * inv_se;
  const float s = sigmoidf_(pos - x);
  float P = on ? fmaf(w, s, 0.f) : 0.f, A = on ? fmaf(w, s * (1.0f - s), 0.f) : 0.f;
#pragma unroll
  for (int off = 8; off >= 1; off >>= 1) {
    P += __shfl_xor(P, off, 64);
    A += __shfl_xor(A, off, 64);
  }
  if (!live) return;
  const BprRow br = bpr_row(P, inv_b);
  if (l == 0) loss_vec[row] = br.loss;
  if (gpred) {
    float* g = gpred + row * C;
    if (l == 0) g[0] = br.dLdP * A;
    if (on) g[1 + l] = br.dLdP * bpr_dP_dneg(w, s, P);
  }
}

// single workgroup, fixed order: thread t sums elements t, t+256, ... (float4-wide when
// the buffer allows), then an LDS tree.  n is a batch size here, not a table size.
__global__ __launch_bounds__(kBlock) void reduce_sum_kernel(const float* __restrict__ x,
                                                            int64_t n, float scale,
                                                            float* __restrict__ out) {
  __shared__ float sm[kBlock];
  const float acc = fixed_order_partial<kBlock, true>(x, n, (int)threadIdx.x);
  sm[threadIdx.x] = acc;
  __syncthreads();
  for (int off = kBlock / 2; off >= 1; off >>= 1) {
    if ((int)threadIdx.x < off) sm[threadIdx.x] += sm[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = sm[0] * scale;
}

}  // namespace rc

using namespace rc;

extern "C" int rc_bpr_loss_fwd_bwd(const float* pred, int B, int C, float inv_b,
                                   float* loss_vec, float* gpred, rc_stream_t stream) {
  RC_REQUIRE(pred && loss_vec, "rc_bpr_loss_fwd_bwd: null pointer");
  RC_REQUIRE(B >= 0 && C >= 2, "rc_bpr_loss_fwd_bwd: need C >= 2 (one negative), got B=%d C=%d",
             B, C);
  if (B == 0) return RC_OK;
  if (C - 1 <= 16) {
    const int blocks = (B + (kBlock / 16) - 1) / (kBlock / 16);
    hipLaunchKernelGGL(bpr_loss_small_kernel, dim3(blocks), dim3(kBlock), 0, as_stream(stream), pred, B, C, inv_b, loss_vec, gpred);
    RC_LAUNCH_CHECK();
    return RC_OK;
  }
  const int blocks = (B + (kBlock / 64) - 1) / (kBlock / 64);
  hipLaunchKernelGGL(bpr_loss_kernel, dim3(blocks), dim3(kBlock), 0, as_stream(stream), pred, B,
                     C, inv_b, loss_vec, gpred);
  RC_LAUNCH_CHECK();
  return RC_OK;
}

extern "C" int rc_reduce_sum(const float* x, int64_t n, float scale, float* out,
                             rc_stream_t stream) {
  RC_REQUIRE(x && out, "rc_reduce_sum: null pointer");
  RC_REQUIRE(n >= 0, "rc_reduce_sum: n < 0");
  hipLaunchKernelGGL(reduce_sum_kernel, dim3(1), dim3(kBlock), 0, as_stream(stream), x, n, scale,
                     out);
  RC_LAUNCH_CHECK();
  return RC_OK;
}
