// sasrec.hip -- the SASRec sequence encoder (reference: models/sequential/SASRec.py:51-86 with
// utils/layers.py TransformerLayer :92-118 and MultiHeadAttention :9-63), forward and backward,
// fp32, dropout 0.  One workgroup per sequence; the whole [len, d] working set of a layer
// (X, Q, K, V, attention probabilities of one head, LayerNorm-normalised activations, FFN hidden)
// lives in LDS (8 buffers x 16.6 KB at d = 64).
//
// Every contraction -- the five d x d projections, QK^T, AV and their backward counterparts -- runs
// on the fp32 matrix cores: v_mfma_f32_32x32x2_f32, i.e. exact f32 FMA chains (1e-5 parity with
// the reference's fp32 matmuls; there is no reduced-precision path).  Operands are fetched
// straight from LDS (row stride d+1 / len+1: conflict-free for lane <-> row) or, for the weights,
// from L2 with the lanes running along the contiguous dimension ([in,out] transposed copies are
// made once per call for the forward direction).  A 4-wave workgroup owns the 32x32 output blocks
// round-robin; blocks above the causal diagonal are skipped.
//
// Reference quirks kept: causal mask only (padding is on the right, so valid rows never see it);
// position id = length - index; no attention output projection; softmax after subtracting a
// maximum (the reference's global max is a mathematical no-op; the row max is used); rows past
// the sequence length are never computed because the reference zeroes them (SASRec.py:74) and
// nothing valid attends to them.
//
// The backward kernel re-runs each layer's forward from the saved layer input (written by the
// forward kernel in training mode) instead of storing activations, then walks the layer backwards.
// Dense-parameter gradients are accumulated per workgroup in a private slice of a partial buffer
// and summed over workgroups in fixed order afterwards (no float atomics, bit-reproducible).
#include "common.hpp"
#include "sas_mma.hpp"

namespace rc {

struct SasArgs {
  const float* item_emb;   // [n_items, D]
  const float* pos_emb;    // [max_his+1, D]
  SasLayer layer[kSasMaxLayers];
  int n_layers, n_heads;
  const int64_t* hist;     // [B, L] right padded with 0
  const int64_t* lengths;  // [B]
  int B, L;
  float* hv;               // fwd out: [B, D] encoder output at position length-1
  float* xsave;            // fwd out (training) / bwd in: layer inputs [B][n_layers][L][D]
  const float* dhv;        // bwd in: [B, D]
  float* g_hist;           // bwd out: [B, L, D] gradient of the layer-0 input rows (0 past length)
  float* part;             // bwd: per-workgroup partial dense grads [n_wg][n_layers * PL]
  // one launch covers the sequences seq_list[0 .. *seq_count) (nullptr: all B) with `lp` padded rows
  int lp;
  const int32_t* seq_list;
  const int32_t* seq_count;
};

// LDS geometry depends on the padded row count: 32 rows (8 buffers = 66 KB, two workgroups per CU) or 64
// rows (133 KB, one per CU).  sa = row stride of the [rows][rows] buffer.  With history_max > 32 the batch
// is split by length on the device (sas_bucket_kernel) and the sequences of <= 32 items -- most of them in
// practice -- run the 32-row geometry in their own launch.
__host__ __device__ inline int sas_buf_floats(int D, int lp) { return lp * ((D + 1) > (lp + 1) ? (D + 1) : (lp + 1)); }
__host__ __device__ inline int sas_lds_floats(int D, int lp) { return 8 * sas_buf_floats(D, lp) + 2 * lp; }

// ---- building blocks (all threads of the workgroup call them; n = valid rows) ----------------

// LayerNorm of rows z[i][:] -> xhat (in place) and rstd[i]; optionally y = w*xhat + b into yout
template <int D>
__device__ __forceinline__ void sas_layernorm(float* z, float* rstd, float* yout, const float* __restrict__ w,
                                              const float* __restrict__ b, int n) {
  constexpr int SD = SasCfg<D>::SD;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = wave; i < n; i += kBlock / 64) {
    float s = 0.f;
    for (int k = lane; k < D; k += 64) s += z[i * SD + k];
    const float mu = wave_allreduce_sum(s) / D;
    float v = 0.f;
    for (int k = lane; k < D; k += 64) {
      const float c = z[i * SD + k] - mu;
      v = fmaf(c, c, v);
    }
    const float rs = 1.0f / sqrtf(wave_allreduce_sum(v) / D + kLnEps);
    if (lane == 0) rstd[i] = rs;
    for (int k = lane; k < D; k += 64) {
      const float xh = (z[i * SD + k] - mu) * rs;
      z[i * SD + k] = xh;
      if (yout) yout[i * SD + k] = fmaf(xh, w[k], b[k]);
    }
  }
}

// one transformer block forward on LDS buffers.
//   in : X (layer input)            out: Y <- layer output X' (if KEEP: Y <- xhat2 instead)
//   C <- xhat1, H <- relu hidden, rstd1/rstd2; Q, K, V <- projections
//   A: scratch for one head's attention probabilities; may alias H (H is written after attention)
template <int D, bool KEEP>
__device__ void sas_layer_forward(const SasLayer& p, float* X, float* Q, float* K, float* V, float* A,
                                  float* C, float* H, float* Y, float* rstd1, float* rstd2, int n,
                                  int n_heads, int SA) {
  constexpr int SD = SasCfg<D>::SD;
  const int dk = D / n_heads;
  const float sqrt_dk = sqrtf((float)dk);
  const MatA aX{X, SD, 1};
  sas_mm(aX, MatB{p.WqT, D, 1}, n, D, D, false, [&](int i, int o, float v) { Q[i * SD + o] = v + p.bq[o]; });
  sas_mm(aX, MatB{p.WkT, D, 1}, n, D, D, false, [&](int i, int o, float v) { K[i * SD + o] = v + p.bk[o]; });
  sas_mm(aX, MatB{p.WvT, D, 1}, n, D, D, false, [&](int i, int o, float v) { V[i * SD + o] = v + p.bv[o]; });
  __syncthreads();
  for (int hh = 0; hh < n_heads; ++hh) {
    sas_attn_probs<D>(A, Q, K, n, hh, dk, sqrt_dk, SA);
    const int hc = hh * dk;  // ctx_h = A . V_h, + residual
    sas_mm(MatA{A, SA, 1}, MatB{V + hc, SD, 1}, n, dk, n, false,
           [&](int i, int c, float v) { C[i * SD + hc + c] = v + X[i * SD + hc + c]; });
    __syncthreads();
  }
  sas_layernorm<D>(C, rstd1, Y, p.ln1w, p.ln1b, n);  // C <- xhat1, Y <- y1
  __syncthreads();
  sas_mm(MatA{Y, SD, 1}, MatB{p.W1T, D, 1}, n, D, D, false,
         [&](int i, int o, float v) { H[i * SD + o] = fmaxf(v + p.b1[o], 0.f); });
  __syncthreads();
  // z2 = H W2^T + b2 + y1: Y is only touched by the epilogue (own element), H is the operand
  sas_mm(MatA{H, SD, 1}, MatB{p.W2T, D, 1}, n, D, D, false,
         [&](int i, int o, float v) { Y[i * SD + o] += v + p.b2[o]; });
  __syncthreads();
  // KEEP: Y <- xhat2 (what LayerNorm2's backward needs); else Y <- layer output
  sas_layernorm<D>(Y, rstd2, KEEP ? nullptr : Y, p.ln2w, p.ln2b, n);
  __syncthreads();
}

template <int D>
__device__ __forceinline__ void sas_load_input(const SasArgs& a, int64_t b, int n, float* X) {
  constexpr int SD = SasCfg<D>::SD;
  for (int idx = threadIdx.x; idx < n * D; idx += kBlock) {
    const int i = idx / D, k = idx % D;
    const int64_t item = a.hist[b * a.L + i];
    X[i * SD + k] = a.item_emb[item * D + k] + a.pos_emb[(int64_t)(n - i) * D + k];  // position = len - i
  }
}

template <int D, bool SAVE>
__global__ __launch_bounds__(kBlock) void sasrec_fwd_kernel(SasArgs a) {
  using Cfg = SasCfg<D>;
  constexpr int SD = Cfg::SD;
  const int LP = a.lp, SA = LP + 1, BUF = sas_buf_floats(D, LP);
  extern __shared__ float lds[];
  float *X = lds, *Q = X + BUF, *K = Q + BUF, *V = K + BUF, *A = V + BUF, *C = A + BUF, *H = C + BUF,
        *Y = H + BUF, *rstd1 = Y + BUF, *rstd2 = rstd1 + LP;
  const int todo = a.seq_count ? *a.seq_count : a.B;
  for (int w = blockIdx.x; w < todo; w += gridDim.x) {
    const int64_t b = a.seq_list ? a.seq_list[w] : w;
    int n = (int)a.lengths[b];
    if (n > a.L) n = a.L;
    sas_load_input<D>(a, b, n, X);
    __syncthreads();
    for (int l = 0; l < a.n_layers; ++l) {
      if (SAVE) {
        float* xs = a.xsave + ((size_t)b * a.n_layers + l) * a.L * D;
        for (int idx = threadIdx.x; idx < n * D; idx += kBlock) xs[idx] = X[(idx / D) * SD + idx % D];
      }
      sas_layer_forward<D, false>(a.layer[l], X, Q, K, V, A, C, H, Y, rstd1, rstd2, n, a.n_heads, SA);
      for (int idx = threadIdx.x; idx < n * D; idx += kBlock) X[(idx / D) * SD + idx % D] = Y[(idx / D) * SD + idx % D];
      __syncthreads();
    }
    for (int k = threadIdx.x; k < D; k += kBlock) a.hv[b * D + k] = n > 0 ? X[(n - 1) * SD + k] : 0.f;
    __syncthreads();
  }
}

// ---- backward helpers ---------------------------------------------------------------------------

// gb[k] += sum_i da[i][k]
template <int D>
__device__ __forceinline__ void sas_accum_colsum(float* gb, const float* da, int n) {
  constexpr int SD = SasCfg<D>::SD;
  for (int k = threadIdx.x; k < D; k += kBlock) {
    float acc = 0.f;
    for (int i = 0; i < n; ++i) acc += da[i * SD + k];
    gb[k] += acc;
  }
}
// gW[o][k] += sum_i da[i][o] * xb[i][k]   (gW: this workgroup's private slice in global memory)
template <int D>
__device__ __forceinline__ void sas_accum_outer(float* gW, const float* da, const float* xb, int n) {
  constexpr int SD = SasCfg<D>::SD;
  sas_mm(MatA{da, 1, SD}, MatB{xb, SD, 1}, D, D, n, false, [&](int o, int k, float v) { gW[o * D + k] += v; });
}
// G[i][k] += sum_o da[i][o] * W[o][k]   (W: nn.Linear weight [out, in] in global memory)
template <int D>
__device__ __forceinline__ void sas_backprop_linear(float* G, const float* da, const float* __restrict__ W, int n) {
  constexpr int SD = SasCfg<D>::SD;
  sas_mm(MatA{da, SD, 1}, MatB{W, D, 1}, n, D, D, false, [&](int i, int k, float v) { G[i * SD + k] += v; });
}
// LayerNorm backward in place: G holds dY on entry, dZ on exit; gw/gb accumulate d(weight)/d(bias)
template <int D>
__device__ __forceinline__ void sas_layernorm_bwd(float* G, const float* xhat, const float* rstd,
                                                  const float* __restrict__ w, float* gw, float* gb, int n) {
  constexpr int SD = SasCfg<D>::SD;
  for (int k = threadIdx.x; k < D; k += kBlock) {
    float aw = 0.f, ab = 0.f;
    for (int i = 0; i < n; ++i) {
      aw = fmaf(G[i * SD + k], xhat[i * SD + k], aw);
      ab += G[i * SD + k];
    }
    gw[k] += aw;
    gb[k] += ab;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = wave; i < n; i += kBlock / 64) {
    float s1 = 0.f, s2 = 0.f;
    for (int k = lane; k < D; k += 64) {
      const float dx = G[i * SD + k] * w[k];
      s1 += dx;
      s2 = fmaf(dx, xhat[i * SD + k], s2);
    }
    const float m1 = wave_allreduce_sum(s1) / D, m2 = wave_allreduce_sum(s2) / D;
    const float rs = rstd[i];
    for (int k = lane; k < D; k += 64) {
      const float dx = G[i * SD + k] * w[k];
      G[i * SD + k] = rs * (dx - m1 - xhat[i * SD + k] * m2);
    }
  }
  __syncthreads();
}

template <int D>
__global__ __launch_bounds__(kBlock) void sasrec_bwd_kernel(SasArgs a) {
  using Cfg = SasCfg<D>;
  constexpr int SD = Cfg::SD, PL = Cfg::PL;
  const int LP = a.lp, SA = LP + 1, BUF = sas_buf_floats(D, LP);
  extern __shared__ float lds[];
  float *X = lds, *Q = X + BUF, *K = Q + BUF, *V = K + BUF, *G = V + BUF, *C = G + BUF, *H = C + BUF,
        *Y = H + BUF, *rstd1 = Y + BUF, *rstd2 = rstd1 + LP;
  float* A = Y;  // attention probabilities reuse Y once the FFN backward no longer needs y1
  float* T = H;  // dA / dS reuse H once the FFN backward is done
  float* part = a.part + (size_t)blockIdx.x * a.n_layers * PL;
  const int todo = a.seq_count ? *a.seq_count : a.B;
  for (int w = blockIdx.x; w < todo; w += gridDim.x) {
    const int64_t b = a.seq_list ? a.seq_list[w] : w;
    int n = (int)a.lengths[b];
    if (n > a.L) n = a.L;
    // dL/d(output of the last layer): only row n-1 (SASRec.py:76), everything else 0
    for (int idx = threadIdx.x; idx < n * D; idx += kBlock) {
      const int i = idx / D, k = idx % D;
      G[i * SD + k] = (i == n - 1) ? a.dhv[b * D + k] : 0.f;
    }
    __syncthreads();
    for (int l = a.n_layers - 1; l >= 0; --l) {
      const SasLayer& p = a.layer[l];
      float* gp = part + (size_t)l * PL;
      const float* xs = a.xsave + ((size_t)b * a.n_layers + l) * a.L * D;
      for (int idx = threadIdx.x; idx < n * D; idx += kBlock) X[(idx / D) * SD + idx % D] = xs[idx];
      __syncthreads();
      // forward of this layer again: C = xhat1, H = relu hidden, Y = xhat2, Q/K/V, rstd1/2
      // (G, the incoming gradient, is untouched; H doubles as the attention scratch)
      sas_layer_forward<D, true>(p, X, Q, K, V, /*A scratch = */ H, C, H, Y, rstd1, rstd2, n, a.n_heads, SA);
      // ---- LayerNorm2, FFN ------------------------------------------------------------------
      sas_layernorm_bwd<D>(G, Y, rstd2, p.ln2w, gp + Cfg::oln2w, gp + Cfg::oln2b, n);  // G = dZ2
      sas_accum_outer<D>(gp + Cfg::oW2, G, H, n);
      sas_accum_colsum<D>(gp + Cfg::ob2, G, n);
      for (int idx = threadIdx.x; idx < n * D; idx += kBlock) {  // Y <- y1 = ln1w*xhat1 + ln1b
        const int i = idx / D, k = idx % D;
        Y[i * SD + k] = fmaf(C[i * SD + k], p.ln1w[k], p.ln1b[k]);
      }
      __syncthreads();
      // H <- dHpre = (dZ2 . W2) * relu'(H): the epilogue owns its element of H, operands are G, W2
      sas_mm(MatA{G, SD, 1}, MatB{p.W2, D, 1}, n, D, D, false,
             [&](int i, int k, float v) { H[i * SD + k] = H[i * SD + k] > 0.f ? v : 0.f; });
      __syncthreads();
      sas_accum_outer<D>(gp + Cfg::oW1, H, Y, n);
      sas_accum_colsum<D>(gp + Cfg::ob1, H, n);
      sas_backprop_linear<D>(G, H, p.W1, n);  // G = dY1 (residual dZ2 + FFN path)
      __syncthreads();
      // ---- LayerNorm1 ----------------------------------------------------------------------------
      sas_layernorm_bwd<D>(G, C, rstd1, p.ln1w, gp + Cfg::oln1w, gp + Cfg::oln1b, n);  // G = dZ1 = dCtx = dX(residual)
      // ---- attention, head by head.  dV, dK overwrite V, K in place; dQ goes to C (xhat1 is dead) ---
      const int dk = D / a.n_heads;
      const float sqrt_dk = sqrtf((float)dk);
      for (int hh = 0; hh < a.n_heads; ++hh) {
        const int hc = hh * dk;
        sas_attn_probs<D>(A, Q, K, n, hh, dk, sqrt_dk, SA);
        // dA = dCtx_h . V_h^T (lower triangle)
        sas_mm(MatA{G + hc, SD, 1}, MatB{V + hc, 1, SD}, n, n, dk, true,
               [&](int i, int j, float v) { T[i * SA + j] = v; });
        __syncthreads();
        // dV_h = A^T . dCtx_h, in place (V_h is not read again)
        sas_mm(MatA{A, 1, SA}, MatB{G + hc, SD, 1}, n, dk, n, false,
               [&](int j, int c, float v) { V[j * SD + hc + c] = v; });
        {  // dS = A * (dA - rowsum(dA*A)) / sqrt(dk), in place in T, 0 above the diagonal
          const int rows_here = LP <= 32 ? 8 : 16;
          sas_softmax_bwd_rows(T, A, n, SA, sqrt_dk, (int)(threadIdx.x >> 6) * rows_here, rows_here);
        }
        __syncthreads();
        // dQ_h = dS . K_h  -> C[:, head columns]
        sas_mm(MatA{T, SA, 1}, MatB{K + hc, SD, 1}, n, dk, n, false,
               [&](int i, int c, float v) { C[i * SD + hc + c] = v; });
        __syncthreads();
        // dK_h = dS^T . Q_h, in place over K_h (dQ_h above was its last reader)
        sas_mm(MatA{T, 1, SA}, MatB{Q + hc, SD, 1}, n, dk, n, false,
               [&](int j, int c, float v) { K[j * SD + hc + c] = v; });
        __syncthreads();
      }
      // ---- projections: parameter grads and dX = dZ1 + dQ Wq + dK Wk + dV Wv  (dQ lives in C) ---------
      sas_accum_outer<D>(gp + Cfg::oWq, C, X, n);
      sas_accum_colsum<D>(gp + Cfg::obq, C, n);
      sas_accum_outer<D>(gp + Cfg::oWk, K, X, n);
      sas_accum_colsum<D>(gp + Cfg::obk, K, n);
      sas_accum_outer<D>(gp + Cfg::oWv, V, X, n);
      sas_accum_colsum<D>(gp + Cfg::obv, V, n);
      sas_backprop_linear<D>(G, C, p.Wq, n);
      __syncthreads();
      sas_backprop_linear<D>(G, K, p.Wk, n);
      __syncthreads();
      sas_backprop_linear<D>(G, V, p.Wv, n);
      __syncthreads();
    }
    // gradient of the layer-0 input rows = item-row + position-row gradients of the history
    float* gh = a.g_hist + (size_t)b * a.L * D;
    for (int idx = threadIdx.x; idx < a.L * D; idx += kBlock) {
      const int i = idx / D;
      gh[idx] = i < n ? G[i * SD + idx % D] : 0.f;
    }
    __syncthreads();
  }
}

// dst[l][m][k][o] = W_m[o][k] for the five square weights of every layer
__global__ __launch_bounds__(kBlock) void sas_transpose_kernel(SasArgs a, float* dst, int D) {
  const int per = 5 * D * D;
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < a.n_layers * per; i += gridDim.x * kBlock) {
    const int l = i / per, r = i % per, m = r / (D * D), e = r % (D * D), k = e / D, o = e % D;
    const SasLayer& p = a.layer[l];
    const float* W = m == 0 ? p.Wq : m == 1 ? p.Wk : m == 2 ? p.Wv : m == 3 ? p.W1 : p.W2;
    dst[i] = W[o * D + k];
  }
}

// transposed copies live at the start of the workspace: [n_layers][5][D][D]
static int sas_make_transposes(SasArgs* a, int D, float* dst, hipStream_t s) {
  const int total = a->n_layers * 5 * D * D;
  hipLaunchKernelGGL(sas_transpose_kernel, dim3((total + kBlock - 1) / kBlock), dim3(kBlock), 0, s, *a, dst, D);
  RC_LAUNCH_CHECK();
  for (int l = 0; l < a->n_layers; ++l) {
    float* base = dst + (size_t)l * 5 * D * D;
    a->layer[l].WqT = base;
    a->layer[l].WkT = base + D * D;
    a->layer[l].WvT = base + 2 * D * D;
    a->layer[l].W1T = base + 3 * D * D;
    a->layer[l].W2T = base + 4 * D * D;
  }
  return RC_OK;
}

static size_t sas_transpose_floats(int d, int n_layers) { return (size_t)n_layers * 5 * d * d; }

static int sas_grid(int B, int lp) {
  const int cap = lp == 32 ? 1024 : 512;  // two resident workgroups per CU at 66 KB of LDS
  return B < cap ? (B < 1 ? 1 : B) : cap;
}

// the launches of one pass: everything at 32 rows when history_max <= 32, else short and long buckets
struct SasPlan {
  int n;
  int lp[2];
  const int32_t* list[2];
  const int32_t* count[2];
};
static int sas_plan(const SasArgs& a, void* bucket_ws, hipStream_t s, SasPlan* plan) {
  if (a.L <= 32) {
    plan->n = 1;
    plan->lp[0] = 32; plan->list[0] = nullptr; plan->count[0] = nullptr;
    return RC_OK;
  }
  int32_t* list = static_cast<int32_t*>(bucket_ws);
  int32_t* count = list + 2 * (size_t)a.B;
  hipLaunchKernelGGL(sas_bucket_kernel, dim3(1), dim3(kBlock), 0, s, a.lengths, a.B, list, count);
  RC_LAUNCH_CHECK();
  plan->n = 2;
  plan->lp[0] = 32; plan->list[0] = list; plan->count[0] = count;
  plan->lp[1] = kSasLP; plan->list[1] = list + a.B; plan->count[1] = count + 1;
  return RC_OK;
}

static int sas_fill_layers(SasArgs* a, const float* const* layer_params, int n_layers) {
  RC_REQUIRE(n_layers >= 1 && n_layers <= kSasMaxLayers, "SASRec: num_layers must be in [1, %d]", kSasMaxLayers);
  RC_REQUIRE(layer_params != nullptr, "SASRec: layer parameter table missing");
  for (int l = 0; l < n_layers; ++l) {
    const float* const* q = layer_params + 14 * l;
    for (int k = 0; k < 14; ++k) RC_REQUIRE(q[k] != nullptr, "SASRec: layer %d parameter %d is null", l, k);
    SasLayer& s = a->layer[l];
    s.Wq = q[0]; s.bq = q[1]; s.Wk = q[2]; s.bk = q[3]; s.Wv = q[4]; s.bv = q[5]; s.ln1w = q[6]; s.ln1b = q[7];
    s.W1 = q[8]; s.b1 = q[9]; s.W2 = q[10]; s.b2 = q[11]; s.ln2w = q[12]; s.ln2b = q[13];
  }
  a->n_layers = n_layers;
  return RC_OK;
}

template <int D>
static int sas_launch_fwd(SasArgs a, bool save, void* bucket_ws, hipStream_t s) {
  SasPlan plan;
  RC_TRY(sas_plan(a, bucket_ws, s, &plan));
  auto kern = save ? sasrec_fwd_kernel<D, true> : sasrec_fwd_kernel<D, false>;
  const size_t lds_max = (size_t)sas_lds_floats(D, plan.lp[plan.n - 1]) * sizeof(float);
  RC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max));
  for (int k = 0; k < plan.n; ++k) {
    a.lp = plan.lp[k]; a.seq_list = plan.list[k]; a.seq_count = plan.count[k];
    const size_t lds_bytes = (size_t)sas_lds_floats(D, a.lp) * sizeof(float);
    hipLaunchKernelGGL(kern, dim3(sas_grid(a.B, a.lp)), dim3(kBlock), lds_bytes, s, a);
    RC_LAUNCH_CHECK();
  }
  return RC_OK;
}

template <int D>
static int sas_launch_bwd(SasArgs a, float* dense_out, void* bucket_ws, hipStream_t s) {
  SasPlan plan;
  RC_TRY(sas_plan(a, bucket_ws, s, &plan));
  int n_wg = 0;
  for (int k = 0; k < plan.n; ++k) n_wg = sas_grid(a.B, plan.lp[k]) > n_wg ? sas_grid(a.B, plan.lp[k]) : n_wg;
  const int count = a.n_layers * SasCfg<D>::PL;
  RC_HIP(hipMemsetAsync(a.part, 0, (size_t)n_wg * count * sizeof(float), s));
  auto kern = sasrec_bwd_kernel<D>;
  const size_t lds_max = (size_t)sas_lds_floats(D, plan.lp[plan.n - 1]) * sizeof(float);
  RC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max));
  for (int k = 0; k < plan.n; ++k) {  // same stream: the second launch adds to the partial slots of the first
    a.lp = plan.lp[k]; a.seq_list = plan.list[k]; a.seq_count = plan.count[k];
    const size_t lds_bytes = (size_t)sas_lds_floats(D, a.lp) * sizeof(float);
    hipLaunchKernelGGL(kern, dim3(sas_grid(a.B, a.lp)), dim3(kBlock), lds_bytes, s, a);
    RC_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(sas_reduce_partials_kernel, dim3((count + 63) / 64), dim3(kBlock), 0, s, a.part,
                     n_wg, count, dense_out);
  RC_LAUNCH_CHECK();
  return RC_OK;
}

}  // namespace rc

using namespace rc;

extern "C" int rc_sasrec_supported(int d, int n_layers, int n_heads, int L) {
  return (sas_core_supported(d, n_layers, n_heads, L) || sas_long_supported(d, n_layers, n_heads, L)) ? 1 : 0;
}

extern "C" int rc_sasrec_dense_param_count(int d) { return 5 * d * d + 9 * d; }

extern "C" size_t rc_sasrec_workspace_bytes(int B, int d, int n_layers) {
  if (B < 1 || d < 1 || n_layers < 1) return 0;
  return align_up(sas_transpose_floats(d, n_layers) * sizeof(float), 256) + sas_bucket_bytes(B) +
         align_up((size_t)1024 * n_layers * (5 * (size_t)d * d + 9 * d) * sizeof(float), 256) + 256;
}

extern "C" int rc_sasrec_fwd(const float* item_emb, const float* pos_emb, const float* const* layer_params,
                             int n_layers, int n_heads, const int64_t* hist, const int64_t* lengths, int B,
                             int L, int d, float* hv, float* xsave, void* ws, size_t ws_bytes,
                             rc_stream_t stream) {
  if (B == 0) return RC_OK;
  RC_REQUIRE(item_emb && pos_emb && hist && lengths && hv && ws, "rc_sasrec_fwd: null pointer");
  if (!sas_core_supported(d, n_layers, n_heads, L))
    return fail(RC_ERR_UNSUPPORTED, "rc_sasrec_fwd: d=%d layers=%d heads=%d L=%d not supported (d in {32,64}, L<=%d)",
                d, n_layers, n_heads, L, kSasLP);
  SasArgs a;
  memset(&a, 0, sizeof(a));
  RC_TRY(sas_fill_layers(&a, layer_params, n_layers));
  a.item_emb = item_emb; a.pos_emb = pos_emb; a.n_heads = n_heads; a.hist = hist; a.lengths = lengths;
  a.B = B; a.L = L; a.hv = hv; a.xsave = xsave;
  hipStream_t s = as_stream(stream);
  const size_t t_bytes = align_up(sas_transpose_floats(d, n_layers) * sizeof(float), 256);
  if (ws_bytes < t_bytes + sas_bucket_bytes(B))
    return fail(RC_ERR_WORKSPACE, "rc_sasrec_fwd: workspace %zu too small (rc_sasrec_workspace_bytes)", ws_bytes);
  RC_TRY(sas_make_transposes(&a, d, reinterpret_cast<float*>(ws), s));
  void* bucket_ws = reinterpret_cast<char*>(ws) + t_bytes;
  return d == 64 ? sas_launch_fwd<64>(a, xsave != nullptr, bucket_ws, s) : sas_launch_fwd<32>(a, xsave != nullptr, bucket_ws, s);
}

extern "C" int rc_sasrec_bwd(const float* const* layer_params, int n_layers, int n_heads, const int64_t* lengths,
                             int B, int L, int d, const float* xsave, const float* dhv, float* g_hist,
                             float* dense_grads, void* ws, size_t ws_bytes, rc_stream_t stream) {
  if (B == 0) return RC_OK;
  RC_REQUIRE(lengths && xsave && dhv && g_hist && dense_grads && ws, "rc_sasrec_bwd: null pointer");
  if (!sas_core_supported(d, n_layers, n_heads, L))
    return fail(RC_ERR_UNSUPPORTED, "rc_sasrec_bwd: d=%d layers=%d heads=%d L=%d not supported", d, n_layers, n_heads, L);
  if (ws_bytes < rc_sasrec_workspace_bytes(B, d, n_layers))
    return fail(RC_ERR_WORKSPACE, "rc_sasrec_bwd: workspace %zu < %zu", ws_bytes, rc_sasrec_workspace_bytes(B, d, n_layers));
  SasArgs a;
  memset(&a, 0, sizeof(a));
  RC_TRY(sas_fill_layers(&a, layer_params, n_layers));
  a.n_heads = n_heads; a.lengths = lengths; a.B = B; a.L = L;
  a.xsave = const_cast<float*>(xsave); a.dhv = dhv; a.g_hist = g_hist;
  hipStream_t s = as_stream(stream);
  RC_TRY(sas_make_transposes(&a, d, reinterpret_cast<float*>(ws), s));
  const size_t t_bytes = align_up(sas_transpose_floats(d, n_layers) * sizeof(float), 256);
  void* bucket_ws = reinterpret_cast<char*>(ws) + t_bytes;
  a.part = reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + t_bytes + sas_bucket_bytes(B));
  return d == 64 ? sas_launch_bwd<64>(a, dense_grads, bucket_ws, s) : sas_launch_bwd<32>(a, dense_grads, bucket_ws, s);
}
