// listwise_loss.hip -- list-wise softmax cross-entropy over an impression list, forward and the
// closed-form gradient (reference: ImpressionModel.loss, loss_n == 'softmaxCE',
// models/BaseImpressionModel.py:44-48,96-107; labels built in helpers/ImpressionRunner.py:187-190).
//
//   mask = target != -1;  have_neg_b = mask[b, P]            (P = train_max_pos_item)
//   p = softmax over the valid columns of row b (row max subtracted)
//   row_b = -(sum_{i < P, mask} log p_i) / #(target == 1)
//   loss = sum_b row_b * have_neg_b / sum_b have_neg_b
//   dloss/dx_bj = -(have_neg_b / H / n_b) * (1[j in S_b] - |S_b| p_bj)  on valid columns, 0 on padding
//
// One wave per row (lists are 40 ... a few hundred entries), shuffle reductions; the normaliser
// H = sum_b have_neg_b is produced by a first single-workgroup kernel (deterministic).
#include "common.hpp"

namespace rc {

__global__ __launch_bounds__(kBlock) void listwise_count_kernel(const int64_t* __restrict__ target, int B,
                                                                int n, int P, float* __restrict__ h_out) {
  __shared__ float sm[kBlock];
  float acc = 0.f;
  for (int b = threadIdx.x; b < B; b += kBlock) acc += target[(size_t)b * n + P] != -1 ? 1.f : 0.f;
  sm[threadIdx.x] = acc;
  __syncthreads();
  for (int off = kBlock / 2; off >= 1; off >>= 1) {
    if ((int)threadIdx.x < off) sm[threadIdx.x] += sm[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) h_out[0] = sm[0];
}

__global__ __launch_bounds__(kBlock) void listwise_softmax_ce_kernel(
    const float* __restrict__ pred, const int64_t* __restrict__ target, int B, int n, int P,
    const float* __restrict__ h_sum, float* __restrict__ loss_vec, float* __restrict__ gpred) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
  if (row >= B) return;  // wave-uniform
  const float* x = pred + row * n;
  const int64_t* tg = target + row * n;
  float mx = -INFINITY;
  for (int j = lane; j < n; j += 64)
    if (tg[j] != -1) mx = fmaxf(mx, x[j]);
  mx = wave_allreduce_max(mx);
  float se = 0.f, n_pos = 0.f, s_cnt = 0.f;
  for (int j = lane; j < n; j += 64) {
    const int64_t t = tg[j];
    if (t != -1) se += expf(x[j] - mx);
    if (t == 1) n_pos += 1.f;
    if (j < P && t != -1) s_cnt += 1.f;
  }
  se = wave_allreduce_sum(se);
  n_pos = wave_allreduce_sum(n_pos);
  s_cnt = wave_allreduce_sum(s_cnt);
  const float log_se = logf(se);
  float lsum = 0.f;  // sum over valid positive slots of log p_i = (x_i - mx) - log(se)
  for (int j = lane; j < P && j < n; j += 64)
    if (tg[j] != -1) lsum += (x[j] - mx) - log_se;
  lsum = wave_allreduce_sum(lsum);
  const float have_neg = tg[P] != -1 ? 1.f : 0.f;
  const float H = h_sum[0];
  const float row_loss = -lsum / n_pos;
  if (lane == 0) loss_vec[row] = row_loss * have_neg / H;  // summing loss_vec gives the loss
  if (gpred) {
    const float coef = -(have_neg / H) / n_pos;
    for (int j = lane; j < n; j += 64) {
      float g = 0.f;
      if (tg[j] != -1) {
        const float p = expf(x[j] - mx) / se;
        g = coef * ((j < P ? 1.f : 0.f) - s_cnt * p);
      }
      gpred[row * n + j] = g;
    }
  }
}

// ---- list-level BPR (ImpressionModel.loss, loss_n 'BPR' / 'BPRhard': models/BaseImpressionModel.py:50-89) ----
//   valid positives i (columns < P, target != -1), valid negatives j (columns >= P, target != -1)
//   a = softmax over the positives of s (of -s for 'hard'),  b = softmax over the negatives of s
//   Q = sum_i a_i sum_j b_j sigmoid(s_i - s_j);   row loss = -log Q;   loss = mean over rows
// Both softmax weights are differentiated through, as autograd does in the reference:
//   dQ/ds_i =  a_i sum_j b_j sig'_ij  +/- a_i (Q_i - Q)       Q_i = sum_j b_j sig_ij   (- for 'hard')
//   dQ/ds_j = -b_j sum_i a_i sig'_ij  +   b_j (R_j - Q)       R_j = sum_i a_i sig_ij
// One wave per row; lanes stride over the positives (then the negatives) and loop over the other side,
// O(P * N) sigmoids per row (400 at the default 20 + 20).
__global__ __launch_bounds__(kBlock) void list_bpr_kernel(const float* __restrict__ pred,
                                                          const int64_t* __restrict__ target, int B, int n, int P,
                                                          int hard, float inv_b, float* __restrict__ loss_vec,
                                                          float* __restrict__ gpred) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
  if (row >= B) return;  // wave-uniform
  const float* x = pred + row * n;
  const int64_t* tg = target + row * n;
  const float sgn = hard ? -1.f : 1.f;
  float ma = -INFINITY, mb = -INFINITY;
  for (int j = lane; j < n; j += 64) {
    if (tg[j] == -1) continue;
    if (j < P) ma = fmaxf(ma, sgn * x[j]); else mb = fmaxf(mb, x[j]);
  }
  ma = wave_allreduce_max(ma);
  mb = wave_allreduce_max(mb);
  float sa = 0.f, sb = 0.f;
  for (int j = lane; j < n; j += 64) {
    if (tg[j] == -1) continue;
    if (j < P) sa += expf(sgn * x[j] - ma); else sb += expf(x[j] - mb);
  }
  sa = wave_allreduce_sum(sa);
  sb = wave_allreduce_sum(sb);
  // Q = sum_i a_i Q_i
  float q_part = 0.f;
  for (int i = lane; i < P && i < n; i += 64) {
    if (tg[i] == -1) continue;
    float qi = 0.f;
    for (int j = P; j < n; ++j)
      if (tg[j] != -1) qi += (expf(x[j] - mb) / sb) * sigmoidf_(x[i] - x[j]);
    q_part += (expf(sgn * x[i] - ma) / sa) * qi;
  }
  const float Q = wave_allreduce_sum(q_part);
  if (lane == 0) loss_vec[row] = -logf(Q);
  if (!gpred) return;
  const float dLdQ = -inv_b / Q;
  for (int c = lane; c < n; c += 64) {
    float g = 0.f;
    if (tg[c] != -1) {
      if (c < P) {  // positive i = c
        const float a = expf(sgn * x[c] - ma) / sa;
        float qi = 0.f, di = 0.f;
        for (int j = P; j < n; ++j)
          if (tg[j] != -1) {
            const float b = expf(x[j] - mb) / sb, sg = sigmoidf_(x[c] - x[j]);
            qi += b * sg;
            di += b * sg * (1.f - sg);
          }
        g = dLdQ * (a * di + sgn * a * (qi - Q));
      } else {  // negative j = c
        const float b = expf(x[c] - mb) / sb;
        float rj = 0.f, ej = 0.f;
        for (int i = 0; i < P; ++i)
          if (tg[i] != -1) {
            const float a = expf(sgn * x[i] - ma) / sa, sg = sigmoidf_(x[i] - x[c]);
            rj += a * sg;
            ej += a * sg * (1.f - sg);
          }
        g = dLdQ * (b * (rj - Q) - b * ej);
      }
    }
    gpred[row * n + c] = g;
  }
}

// torch.nn.functional.softplus (beta 1, threshold 20)
__device__ __forceinline__ float softplusf_(float x) { return x > 20.f ? x : log1pf(expf(x)); }

// ---- list-level BPR, re-weighting AFTER / BEFORE the log-sigmoid (models/BaseImpressionModel.py:76-81) -----------
//   a, b as in list_bpr_kernel (softmax weights of the valid positives / negatives), d_ij = s_i - s_j
//   after :  row = sum_i a_i T_i,  T_i = sum_j b_j softplus(-d_ij)
//            d/ds_i = -a_i sum_j b_j r_ij +/- a_i (T_i - row)      r_ij = sigmoid(-d_ij)   (- for 'hard')
//            d/ds_j =  b_j sum_i a_i r_ij +   b_j (R_j - row)      R_j = sum_i a_i softplus(-d_ij)
//   before:  row = sum_{i valid pos} softplus(x_i) + (n - #valid pos) log 2,   x_i = -a_i (s_i - m),  m = sum_j b_j s_j
//            (the reference sums softplus over ALL n columns of the masked matrix: every other column adds softplus(0))
//            d/ds_k (pos) = -/+ (y_k - a_k Y) - sigmoid(x_k) a_k     y_i = sigmoid(x_i) a_i (s_i - m),  Y = sum y
//            d/ds_k (neg) = Z b_k (1 + s_k - m)                      Z = sum_i sigmoid(x_i) a_i
// loss = mean over rows (inv_b).  One wave per row.
template <bool BEFORE>
__global__ __launch_bounds__(kBlock) void list_bpr_reweight_kernel(const float* __restrict__ pred,
                                                                   const int64_t* __restrict__ target, int B, int n, int P,
                                                                   int hard, float inv_b, float* __restrict__ loss_vec,
                                                                   float* __restrict__ gpred) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
  if (row >= B) return;  // wave-uniform
  const float* x = pred + row * n;
  const int64_t* tg = target + row * n;
  const float sgn = hard ? -1.f : 1.f;
  float ma = -INFINITY, mb = -INFINITY;
  for (int j = lane; j < n; j += 64) {
    if (tg[j] == -1) continue;
    if (j < P) ma = fmaxf(ma, sgn * x[j]); else mb = fmaxf(mb, x[j]);
  }
  ma = wave_allreduce_max(ma);
  mb = wave_allreduce_max(mb);
  float sa = 0.f, sb = 0.f, n_pos = 0.f;
  for (int j = lane; j < n; j += 64) {
    if (tg[j] == -1) continue;
    if (j < P) {
      sa += expf(sgn * x[j] - ma);
      n_pos += 1.f;
    } else {
      sb += expf(x[j] - mb);
    }
  }
  sa = wave_allreduce_sum(sa);
  sb = wave_allreduce_sum(sb);
  n_pos = wave_allreduce_sum(n_pos);
  if (BEFORE) {
    float m = 0.f;
    for (int j = P + lane; j < n; j += 64)
      if (tg[j] != -1) m += (expf(x[j] - mb) / sb) * x[j];
    m = wave_allreduce_sum(m);
    float l_part = 0.f, Y = 0.f, Z = 0.f;
    for (int i = lane; i < P && i < n; i += 64) {
      if (tg[i] == -1) continue;
      const float a = expf(sgn * x[i] - ma) / sa;
      const float xi = -a * (x[i] - m);
      const float sg = sigmoidf_(xi);
      l_part += softplusf_(xi);
      Y += sg * a * (x[i] - m);
      Z += sg * a;
    }
    const float rowl = wave_allreduce_sum(l_part) + ((float)n - n_pos) * 0.69314718055994531f;
    Y = wave_allreduce_sum(Y);
    Z = wave_allreduce_sum(Z);
    if (lane == 0) loss_vec[row] = rowl;
    if (!gpred) return;
    for (int c = lane; c < n; c += 64) {
      float g = 0.f;
      if (tg[c] != -1) {
        if (c < P) {
          const float a = expf(sgn * x[c] - ma) / sa;
          const float xc = -a * (x[c] - m);
          const float sg = sigmoidf_(xc);
          g = -sgn * (sg * a * (x[c] - m) - a * Y) - sg * a;
        } else {
          g = Z * (expf(x[c] - mb) / sb) * (1.f + x[c] - m);
        }
      }
      gpred[row * n + c] = inv_b * g;
    }
    return;
  }
  // after
  float l_part = 0.f;
  for (int i = lane; i < P && i < n; i += 64) {
    if (tg[i] == -1) continue;
    float ti = 0.f;
    for (int j = P; j < n; ++j)
      if (tg[j] != -1) ti += (expf(x[j] - mb) / sb) * softplusf_(-(x[i] - x[j]));
    l_part += (expf(sgn * x[i] - ma) / sa) * ti;
  }
  const float rowl = wave_allreduce_sum(l_part);
  if (lane == 0) loss_vec[row] = rowl;
  if (!gpred) return;
  for (int c = lane; c < n; c += 64) {
    float g = 0.f;
    if (tg[c] != -1) {
      if (c < P) {
        const float a = expf(sgn * x[c] - ma) / sa;
        float tc = 0.f, rc_ = 0.f;
        for (int j = P; j < n; ++j)
          if (tg[j] != -1) {
            const float b = expf(x[j] - mb) / sb, d = x[c] - x[j];
            tc += b * softplusf_(-d);
            rc_ += b * sigmoidf_(-d);
          }
        g = -a * rc_ + sgn * a * (tc - rowl);
      } else {
        const float b = expf(x[c] - mb) / sb;
        float rj = 0.f, ej = 0.f;
        for (int i = 0; i < P; ++i)
          if (tg[i] != -1) {
            const float a = expf(sgn * x[i] - ma) / sa, d = x[i] - x[c];
            rj += a * softplusf_(-d);
            ej += a * sigmoidf_(-d);
          }
        g = b * ej + b * (rj - rowl);
      }
    }
    gpred[row * n + c] = inv_b * g;
  }
}

// ---- list-level BPR without re-weighting, 'BPR...simple' (models/BaseImpressionModel.py:82-83) ---------------------
//   row = sum over the valid (positive i, negative j) pairs of softplus(-(s_i - s_j)); the reference returns the rows
//   UNREDUCED ([B]): loss_vec is the result, gpred[b, c] = scale * d row_b / d s_c  (the caller multiplies by the
//   incoming gradient of row b):   d/ds_i = -sum_j sigmoid(-d_ij),   d/ds_j = sum_i sigmoid(-d_ij).  One wave per row.
__global__ __launch_bounds__(kBlock) void list_bpr_simple_kernel(const float* __restrict__ pred, const int64_t* __restrict__ target,
                                                                int B, int n, int P, float scale, float* __restrict__ loss_vec,
                                                                float* __restrict__ gpred) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
  if (row >= B) return;  // wave-uniform
  const float* x = pred + row * n;
  const int64_t* tg = target + row * n;
  float l_part = 0.f;
  for (int c = lane; c < n; c += 64) {
    float g = 0.f;
    if (tg[c] != -1) {
      if (c < P) {
        for (int j = P; j < n; ++j)
          if (tg[j] != -1) {
            const float d = x[c] - x[j];
            l_part += softplusf_(-d);
            g -= sigmoidf_(-d);
          }
      } else {
        for (int i = 0; i < P; ++i)
          if (tg[i] != -1) g += sigmoidf_(-(x[i] - x[c]));
      }
    }
    if (gpred) gpred[row * n + c] = scale * g;
  }
  l_part = wave_allreduce_sum(l_part);
  if (lane == 0) loss_vec[row] = l_part;
}

// ---- listnet / attention_rank (models/BaseImpressionModel.py:84-94, 109-126) -------------------------------------
//   t = softmax over the valid columns of the labels (1 / 0);  have_neg, H as for softmaxCE
//   listnet:        p = softmax of the scores over ALL n columns (the reference does not mask them; padding only drops
//                   out of the sum),  row = -sum_{valid} t_k log p_k,        d/ds_c = p_c - t_c [c valid]   (every column)
//   attention_rank: p = softmax over the valid columns,  row = -sum t_k log p_k - sum_{p_k != 1} (1 - t_k) log(1 - p_k)
//                   d/ds_c = (p_c - t_c) + p_c (u_c [p_c != 1] - S),  u_k = (1 - t_k) / (1 - p_k),  S = sum_{p_k != 1} u_k p_k
//   loss = sum_b row_b have_neg_b / H
template <bool ATTENTION>
__global__ __launch_bounds__(kBlock) void listnet_kernel(const float* __restrict__ pred, const int64_t* __restrict__ target,
                                                        int B, int n, int P, const float* __restrict__ h_sum,
                                                        float* __restrict__ loss_vec, float* __restrict__ gpred) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
  if (row >= B) return;  // wave-uniform
  const float* x = pred + row * n;
  const int64_t* tg = target + row * n;
  float mx = -INFINITY, mt = -INFINITY;
  for (int j = lane; j < n; j += 64) {
    const bool valid = tg[j] != -1;
    if (valid || !ATTENTION) mx = fmaxf(mx, x[j]);
    if (valid) mt = fmaxf(mt, (float)tg[j]);
  }
  mx = wave_allreduce_max(mx);
  mt = wave_allreduce_max(mt);
  float se = 0.f, st = 0.f;
  for (int j = lane; j < n; j += 64) {
    const bool valid = tg[j] != -1;
    if (valid || !ATTENTION) se += expf(x[j] - mx);
    if (valid) st += expf((float)tg[j] - mt);
  }
  se = wave_allreduce_sum(se);
  st = wave_allreduce_sum(st);
  const float log_se = logf(se);
  float l1 = 0.f, l2 = 0.f, S = 0.f;
  for (int j = lane; j < n; j += 64) {
    if (tg[j] == -1) continue;
    const float t = expf((float)tg[j] - mt) / st;
    l1 -= t * ((x[j] - mx) - log_se);
    if (ATTENTION) {
      const float p = expf(x[j] - mx) / se;
      if (p != 1.f) {
        l2 -= (1.f - t) * logf(1.f - p);
        S += (1.f - t) / (1.f - p) * p;
      }
    }
  }
  l1 = wave_allreduce_sum(l1);
  l2 = wave_allreduce_sum(l2);
  S = wave_allreduce_sum(S);
  const float have_neg = tg[P] != -1 ? 1.f : 0.f;
  const float scale = have_neg / h_sum[0];
  if (lane == 0) loss_vec[row] = (l1 + l2) * scale;  // summing loss_vec gives the loss
  if (!gpred) return;
  for (int c = lane; c < n; c += 64) {
    const bool valid = tg[c] != -1;
    float g = 0.f;
    if (valid || !ATTENTION) {
      const float p = expf(x[c] - mx) / se;
      const float t = valid ? expf((float)tg[c] - mt) / st : 0.f;
      g = p - t;
      if (ATTENTION) g += p * ((p != 1.f ? (1.f - t) / (1.f - p) : 0.f) - S);
    }
    gpred[row * n + c] = scale * g;
  }
}

}  // namespace rc

using namespace rc;

extern "C" int rc_softmax_ce_fwd_bwd(const float* pred, const int64_t* target, int B, int n, int max_pos,
                                     float* loss_vec, float* h_sum, float* gpred, rc_stream_t stream) {
  if (B == 0) return RC_OK;
  RC_REQUIRE(pred && target && loss_vec && h_sum, "rc_softmax_ce_fwd_bwd: null pointer");
  RC_REQUIRE(B > 0 && n >= 2 && max_pos >= 1 && max_pos < n,
             "rc_softmax_ce_fwd_bwd: need 1 <= max_pos < n (got B=%d n=%d max_pos=%d)", B, n, max_pos);
  hipStream_t s = as_stream(stream);
  hipLaunchKernelGGL(listwise_count_kernel, dim3(1), dim3(kBlock), 0, s, target, B, n, max_pos, h_sum);
  RC_LAUNCH_CHECK();
  const int blocks = (B + (kBlock / 64) - 1) / (kBlock / 64);
  hipLaunchKernelGGL(listwise_softmax_ce_kernel, dim3(blocks), dim3(kBlock), 0, s, pred, target, B, n, max_pos,
                     h_sum, loss_vec, gpred);
  RC_LAUNCH_CHECK();
  return RC_OK;
}

extern "C" int rc_list_bpr_fwd_bwd(const float* pred, const int64_t* target, int B, int n, int max_pos, int hard,
                                   float inv_b, float* loss_vec, float* gpred, rc_stream_t stream) {
  if (B == 0) return RC_OK;
  RC_REQUIRE(pred && target && loss_vec, "rc_list_bpr_fwd_bwd: null pointer");
  RC_REQUIRE(B > 0 && n >= 2 && max_pos >= 1 && max_pos < n,
             "rc_list_bpr_fwd_bwd: need 1 <= max_pos < n (got B=%d n=%d max_pos=%d)", B, n, max_pos);
  const int blocks = (B + (kBlock / 64) - 1) / (kBlock / 64);
  hipLaunchKernelGGL(list_bpr_kernel, dim3(blocks), dim3(kBlock), 0, as_stream(stream), pred, target, B, n, max_pos,
                     hard, inv_b, loss_vec, gpred);
  RC_LAUNCH_CHECK();
  return RC_OK;
}

// Every list-wise loss name of ImpressionModel.loss (models/BaseImpressionModel.py:44-129): kind = RC_LIST_*.  loss_vec [B]
// holds per-row terms whose fixed-order sum (rc_reduce_sum, scale 1/B for the BPR kinds, 1 for the others) is the loss
// -- except RC_LIST_BPR_SIMPLE, whose rows ARE the result (the reference leaves them unreduced); h_sum [1] is scratch for the kinds normalised by the number
// of rows that have a negative (softmaxCE, listnet, attention_rank); gpred (optional) = dloss/dpred.
extern "C" int rc_list_loss_fwd_bwd(const float* pred, const int64_t* target, int B, int n, int max_pos, int kind,
                                    float inv_b, float* loss_vec, float* h_sum, float* gpred, rc_stream_t stream) {
  if (B == 0) return RC_OK;
  RC_REQUIRE(pred && target && loss_vec, "rc_list_loss_fwd_bwd: null pointer");
  RC_REQUIRE(B > 0 && n >= 2 && max_pos >= 1 && max_pos < n,
             "rc_list_loss_fwd_bwd: need 1 <= max_pos < n (got B=%d n=%d max_pos=%d)", B, n, max_pos);
  hipStream_t s = as_stream(stream);
  const int blocks = (B + (kBlock / 64) - 1) / (kBlock / 64);
  switch (kind) {
    case RC_LIST_BPR:
    case RC_LIST_BPR_HARD:
      return rc_list_bpr_fwd_bwd(pred, target, B, n, max_pos, kind == RC_LIST_BPR_HARD, inv_b, loss_vec, gpred, stream);
    case RC_LIST_SOFTMAX_CE:
      RC_REQUIRE(h_sum, "rc_list_loss_fwd_bwd: softmaxCE needs h_sum");
      return rc_softmax_ce_fwd_bwd(pred, target, B, n, max_pos, loss_vec, h_sum, gpred, stream);
    case RC_LIST_BPR_AFTER:
    case RC_LIST_BPR_HARD_AFTER:
      hipLaunchKernelGGL((list_bpr_reweight_kernel<false>), dim3(blocks), dim3(kBlock), 0, s, pred, target, B, n, max_pos,
                         kind == RC_LIST_BPR_HARD_AFTER, inv_b, loss_vec, gpred);
      break;
    case RC_LIST_BPR_BEFORE:
    case RC_LIST_BPR_HARD_BEFORE:
      hipLaunchKernelGGL((list_bpr_reweight_kernel<true>), dim3(blocks), dim3(kBlock), 0, s, pred, target, B, n, max_pos,
                         kind == RC_LIST_BPR_HARD_BEFORE, inv_b, loss_vec, gpred);
      break;
    case RC_LIST_BPR_SIMPLE:
      hipLaunchKernelGGL(list_bpr_simple_kernel, dim3(blocks), dim3(kBlock), 0, s, pred, target, B, n, max_pos, inv_b, loss_vec, gpred);
      break;
    case RC_LIST_LISTNET:
    case RC_LIST_ATTENTION_RANK:
      RC_REQUIRE(h_sum, "rc_list_loss_fwd_bwd: listnet / attention_rank need h_sum");
      hipLaunchKernelGGL(listwise_count_kernel, dim3(1), dim3(kBlock), 0, s, target, B, n, max_pos, h_sum);
      RC_LAUNCH_CHECK();
      if (kind == RC_LIST_LISTNET)
        hipLaunchKernelGGL((listnet_kernel<false>), dim3(blocks), dim3(kBlock), 0, s, pred, target, B, n, max_pos, h_sum, loss_vec, gpred);
      else
        hipLaunchKernelGGL((listnet_kernel<true>), dim3(blocks), dim3(kBlock), 0, s, pred, target, B, n, max_pos, h_sum, loss_vec, gpred);
      break;
    default:
      return fail(RC_ERR_INVALID_ARG, "rc_list_loss_fwd_bwd: unknown kind %d", kind);
  }
  RC_LAUNCH_CHECK();
  return RC_OK;
}
