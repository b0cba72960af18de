// sasrec.hip -- the SASRec sequence encoder (reference: models/sequential/SASRec.py:51-86 with
// utils/layers.py TransformerLayer :92-118 and MultiHeadAttention :9-63), forward and backward,
// fp32, dropout 0.  One workgroup per sequence; the whole [len, d] working set of a layer
// (X, Q, K, V, attention probabilities of one head, LayerNorm-normalised activations, FFN hidden)
// lives in LDS (8 buffers x 16.6 KB at d = 64), weights are read from L2.
//
// Reference quirks kept: causal mask only (padding is on the right, so valid rows never see it);
// position id = length - index; no attention output projection; softmax after subtracting a
// maximum (the reference's global max is a mathematical no-op; the row max is used); rows past
// the sequence length are never computed because the reference zeroes them (SASRec.py:74) and
// nothing valid attends to them.  Only rows < length are touched here.
//
// The backward kernel re-runs each layer's forward from the saved layer input (written by the
// forward kernel in training mode) instead of storing activations, then walks the layer backwards.
// Dense-parameter gradients are accumulated per workgroup in a private slice of a partial buffer
// and summed over workgroups in fixed order afterwards (no float atomics, bit-reproducible).
//
// This first version is VALU-only (per-thread dot products over LDS operands); the QK^T / AV
// contractions are the candidates for the fp32 MFMA path.
#include "common.hpp"

namespace rc {

constexpr int kSasLP = 64;        // max rows (history length) per sequence
constexpr int kSasMaxLayers = 4;
constexpr float kLnEps = 1e-5f;   // nn.LayerNorm default

struct SasLayer {  // device pointers, nn.Linear layout [out, in]
  const float *Wq, *bq, *Wk, *bk, *Wv, *bv, *ln1w, *ln1b, *W1, *b1, *W2, *b2, *ln2w, *ln2b;
  // [in, out] copies made once per call (sas_transpose_kernel): in x W^T the lanes run over the
  // output feature, so W^T[k][o] is the coalesced operand (W[o][k] puts every lane on its own line)
  const float *WqT, *WkT, *WvT, *W1T, *W2T;
};

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
};

template <int D>
struct SasCfg {
  static constexpr int SD = D + 1;                               // row stride of [rows][D] buffers
  static constexpr int SA = kSasLP + 1;                          // row stride of the [rows][rows] buffer
  static constexpr int BUF = kSasLP * (SD > SA ? SD : SA);       // floats per LDS buffer
  static constexpr int PL = 5 * D * D + 9 * D;                   // dense-parameter floats per layer
  // offsets inside one layer's parameter-gradient block (canonical order)
  static constexpr int oWq = 0, obq = oWq + D * D, oWk = obq + D, obk = oWk + D * D, oWv = obk + D,
                       obv = oWv + D * D, oln1w = obv + D, oln1b = oln1w + D, oW1 = oln1b + D,
                       ob1 = oW1 + D * D, oW2 = ob1 + D, ob2 = oW2 + D * D, oln2w = ob2 + D,
                       oln2b = oln2w + D;
  static constexpr int kLdsFloats = 8 * BUF + 2 * kSasLP;
};

// ---- building blocks (all threads of the workgroup call them; n = valid rows) ----------------

// out[i][o] = (RELU) b[o] + sum_k in[i][k] * W[o][k], with WT = W^T ([in, out]) as the operand
template <int D, bool RELU>
__device__ __forceinline__ void sas_linear(float* out, const float* in, const float* __restrict__ WT,
                                           const float* __restrict__ b, int n) {
  constexpr int SD = SasCfg<D>::SD;
  for (int idx = threadIdx.x; idx < n * D; idx += kBlock) {
    const int i = idx / D, o = idx % D;
    const float* x = in + i * SD;
    float acc = b[o];
#pragma unroll 8
    for (int k = 0; k < D; ++k) acc = fmaf(x[k], WT[k * D + o], acc);
    out[i * SD + o] = RELU ? fmaxf(acc, 0.f) : acc;
  }
}

// attention probabilities of head hh into A[i][j] (j <= i), rows [0, n)
template <int D>
__device__ __forceinline__ void sas_attn_probs(float* A, const float* Q, const float* K, int n, int hh,
                                               int dk, float sqrt_dk) {
  constexpr int SD = SasCfg<D>::SD, SA = SasCfg<D>::SA;
  for (int idx = threadIdx.x; idx < n * n; idx += kBlock) {
    const int i = idx / n, j = idx % n;
    if (j <= i) {
      const float* q = Q + i * SD + hh * dk;
      const float* k = K + j * SD + hh * dk;
      float acc = 0.f;
      for (int c = 0; c < dk; ++c) acc = fmaf(q[c], k[c], acc);
      A[i * SA + j] = acc / sqrt_dk;
    }
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = wave; i < n; i += kBlock / 64) {  // one wave per row; n <= 64 = wave width
    const bool on = lane <= i;
    const float s = on ? A[i * SA + lane] : -INFINITY;
    const float m = wave_allreduce_max(s);
    const float e = on ? expf(s - m) : 0.f;
    const float z = wave_allreduce_sum(e);
    if (lane < n) A[i * SA + lane] = e / z;
  }
  __syncthreads();
}

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
                                  int n_heads) {
  constexpr int SD = SasCfg<D>::SD, SA = SasCfg<D>::SA;
  const int dk = D / n_heads;
  const float sqrt_dk = sqrtf((float)dk);
  sas_linear<D, false>(Q, X, p.WqT, p.bq, n);
  sas_linear<D, false>(K, X, p.WkT, p.bk, n);
  sas_linear<D, false>(V, X, p.WvT, p.bv, n);
  __syncthreads();
  for (int hh = 0; hh < n_heads; ++hh) {
    sas_attn_probs<D>(A, Q, K, n, hh, dk, sqrt_dk);
    for (int idx = threadIdx.x; idx < n * dk; idx += kBlock) {  // ctx_h = A . V_h, + residual
      const int i = idx / dk, c = hh * dk + idx % dk;
      float acc = 0.f;
      for (int j = 0; j <= i; ++j) acc = fmaf(A[i * SA + j], V[j * SD + c], acc);
      C[i * SD + c] = acc + X[i * SD + c];
    }
    __syncthreads();
  }
  sas_layernorm<D>(C, rstd1, Y, p.ln1w, p.ln1b, n);  // C <- xhat1, Y <- y1
  __syncthreads();
  sas_linear<D, true>(H, Y, p.W1T, p.b1, n);
  __syncthreads();
  for (int idx = threadIdx.x; idx < n * D; idx += kBlock) {  // z2 = H W2^T + b2 + y1, in place in Y
    const int i = idx / D, o = idx % D;
    const float* h = H + i * SD;
    float acc = p.b2[o];
#pragma unroll 8
    for (int k = 0; k < D; ++k) acc = fmaf(h[k], p.W2T[k * D + o], acc);
    Y[i * SD + o] += acc;  // each thread owns its element of Y
  }
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
  constexpr int SD = Cfg::SD, BUF = Cfg::BUF;
  extern __shared__ float lds[];
  float *X = lds, *Q = X + BUF, *K = Q + BUF, *V = K + BUF, *A = V + BUF, *C = A + BUF, *H = C + BUF,
        *Y = H + BUF, *rstd1 = Y + BUF, *rstd2 = rstd1 + kSasLP;
  for (int64_t b = blockIdx.x; b < a.B; b += gridDim.x) {
    int n = (int)a.lengths[b];
    if (n > a.L) n = a.L;
    sas_load_input<D>(a, b, n, X);
    __syncthreads();
    for (int l = 0; l < a.n_layers; ++l) {
      if (SAVE) {
        float* xs = a.xsave + ((size_t)b * a.n_layers + l) * a.L * D;
        for (int idx = threadIdx.x; idx < n * D; idx += kBlock) xs[idx] = X[(idx / D) * SD + idx % D];
      }
      sas_layer_forward<D, false>(a.layer[l], X, Q, K, V, A, C, H, Y, rstd1, rstd2, n, a.n_heads);
      for (int idx = threadIdx.x; idx < n * D; idx += kBlock) X[(idx / D) * SD + idx % D] = Y[(idx / D) * SD + idx % D];
      __syncthreads();
    }
    for (int k = threadIdx.x; k < D; k += kBlock) a.hv[b * D + k] = n > 0 ? X[(n - 1) * SD + k] : 0.f;
    __syncthreads();
  }
}

// ---- backward helpers ---------------------------------------------------------------------------

// gW[o][k] += sum_i da[i][o] * xb[i][k]   (gW: this workgroup's private slice in global memory)
template <int D>
__device__ __forceinline__ void sas_accum_outer(float* gW, const float* da, const float* xb, int n) {
  constexpr int SD = SasCfg<D>::SD;
  for (int e = threadIdx.x; e < D * D; e += kBlock) {
    const int o = e / D, k = e % D;
    float acc = 0.f;
    for (int i = 0; i < n; ++i) acc = fmaf(da[i * SD + o], xb[i * SD + k], acc);
    gW[e] += acc;
  }
}
template <int D>
__device__ __forceinline__ void sas_accum_colsum(float* gb, const float* da, int n) {
  constexpr int SD = SasCfg<D>::SD;
  for (int k = threadIdx.x; k < D; k += kBlock) {
    float acc = 0.f;
    for (int i = 0; i < n; ++i) acc += da[i * SD + k];
    gb[k] += acc;
  }
}
// G[i][k] += sum_o da[i][o] * W[o][k]
template <int D>
__device__ __forceinline__ void sas_backprop_linear(float* G, const float* da, const float* __restrict__ W, int n) {
  constexpr int SD = SasCfg<D>::SD;
  for (int idx = threadIdx.x; idx < n * D; idx += kBlock) {
    const int i = idx / D, k = idx % D;
    const float* d = da + i * SD;
    float acc = 0.f;
#pragma unroll 8
    for (int o = 0; o < D; ++o) acc = fmaf(d[o], W[o * D + k], acc);
    G[i * SD + k] += acc;
  }
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
  constexpr int SD = Cfg::SD, SA = Cfg::SA, BUF = Cfg::BUF, PL = Cfg::PL;
  constexpr int RMAX = (kSasLP * D + kBlock - 1) / kBlock;  // per-thread elements of an [n][dk<=D] tile
  extern __shared__ float lds[];
  float *X = lds, *Q = X + BUF, *K = Q + BUF, *V = K + BUF, *G = V + BUF, *C = G + BUF, *H = C + BUF,
        *Y = H + BUF, *rstd1 = Y + BUF, *rstd2 = rstd1 + kSasLP;
  float* A = Y;  // attention probabilities reuse Y (xhat2) once LayerNorm2's backward is done
  float* T = H;  // dA / dS reuse H once the FFN backward is done
  float* part = a.part + (size_t)blockIdx.x * a.n_layers * PL;
  for (int64_t b = blockIdx.x; b < a.B; b += gridDim.x) {
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
      // (G, the incoming gradient, is untouched)
      sas_layer_forward<D, true>(p, X, Q, K, V, /*A scratch = */ H, C, H, Y, rstd1, rstd2, n, a.n_heads);
      // ---- LayerNorm2, FFN ------------------------------------------------------------------
      sas_layernorm_bwd<D>(G, Y, rstd2, p.ln2w, gp + Cfg::oln2w, gp + Cfg::oln2b, n);  // G = dZ2
      sas_accum_outer<D>(gp + Cfg::oW2, G, H, n);
      sas_accum_colsum<D>(gp + Cfg::ob2, G, n);
      for (int idx = threadIdx.x; idx < n * D; idx += kBlock) {  // Y <- y1 = ln1w*xhat1 + ln1b
        const int i = idx / D, k = idx % D;
        Y[i * SD + k] = fmaf(C[i * SD + k], p.ln1w[k], p.ln1b[k]);
      }
      __syncthreads();
      for (int idx = threadIdx.x; idx < n * D; idx += kBlock) {  // H <- dHpre = (dZ2 W2) * relu'
        const int i = idx / D, k = idx % D;
        const float* d = G + i * SD;
        float acc = 0.f;
#pragma unroll 8
        for (int o = 0; o < D; ++o) acc = fmaf(d[o], p.W2[o * D + k], acc);
        H[i * SD + k] = H[i * SD + k] > 0.f ? acc : 0.f;
      }
      __syncthreads();
      sas_accum_outer<D>(gp + Cfg::oW1, H, Y, n);
      sas_accum_colsum<D>(gp + Cfg::ob1, H, n);
      sas_backprop_linear<D>(G, H, p.W1, n);  // G = dY1 (residual dZ2 + FFN path)
      __syncthreads();
      // ---- LayerNorm1 ----------------------------------------------------------------------------
      sas_layernorm_bwd<D>(G, C, rstd1, p.ln1w, gp + Cfg::oln1w, gp + Cfg::oln1b, n);  // G = dZ1 = dCtx = dX(residual)
      // ---- attention, head by head; dV, dQ, dK overwrite V, Q, K in place ---------------------------
      const int dk = D / a.n_heads;
      const float sqrt_dk = sqrtf((float)dk);
      for (int hh = 0; hh < a.n_heads; ++hh) {
        sas_attn_probs<D>(A, Q, K, n, hh, dk, sqrt_dk);
        for (int idx = threadIdx.x; idx < n * n; idx += kBlock) {  // dA = dCtx_h V_h^T
          const int i = idx / n, j = idx % n;
          float acc = 0.f;
          if (j <= i) {
            const float* g = G + i * SD + hh * dk;
            const float* v = V + j * SD + hh * dk;
            for (int c = 0; c < dk; ++c) acc = fmaf(g[c], v[c], acc);
          }
          T[i * SA + j] = acc;
        }
        __syncthreads();
        float rv[RMAX], rq[RMAX];
#pragma unroll
        for (int r = 0; r < RMAX; ++r) {  // dV_h[j][c] = sum_{i>=j} A[i][j] dCtx[i][c]
          const int idx = threadIdx.x + r * kBlock;
          rv[r] = 0.f;
          if (idx < n * dk) {
            const int j = idx / dk, c = hh * dk + idx % dk;
            float acc = 0.f;
            for (int i = j; i < n; ++i) acc = fmaf(A[i * SA + j], G[i * SD + c], acc);
            rv[r] = acc;
          }
        }
        {  // dS = A * (dA - rowsum(dA*A)) / sqrt(dk), in place in T, one wave per row
          const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
          for (int i = wave; i < n; i += kBlock / 64) {
            const bool on = lane <= i;
            const float pa = on ? A[i * SA + lane] : 0.f;
            const float da = on ? T[i * SA + lane] : 0.f;
            const float dot = wave_allreduce_sum(pa * da);
            if (lane < n) T[i * SA + lane] = on ? pa * (da - dot) / sqrt_dk : 0.f;
          }
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < RMAX; ++r) {  // dQ_h = dS K_h ; dK_h = dS^T Q_h  (into registers)
          const int idx = threadIdx.x + r * kBlock;
          rq[r] = 0.f;
          if (idx < n * dk) {
            const int i = idx / dk, c = hh * dk + idx % dk;
            float aq = 0.f;
            for (int j = 0; j <= i; ++j) aq = fmaf(T[i * SA + j], K[j * SD + c], aq);
            rq[r] = aq;
          }
        }
        float rk[RMAX];
#pragma unroll
        for (int r = 0; r < RMAX; ++r) {
          const int idx = threadIdx.x + r * kBlock;
          rk[r] = 0.f;
          if (idx < n * dk) {
            const int j = idx / dk, c = hh * dk + idx % dk;
            float ak = 0.f;
            for (int i = j; i < n; ++i) ak = fmaf(T[i * SA + j], Q[i * SD + c], ak);
            rk[r] = ak;
          }
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < RMAX; ++r) {
          const int idx = threadIdx.x + r * kBlock;
          if (idx < n * dk) {
            const int i = idx / dk, c = hh * dk + idx % dk;
            V[i * SD + c] = rv[r];
            Q[i * SD + c] = rq[r];
            K[i * SD + c] = rk[r];
          }
        }
        __syncthreads();
      }
      // ---- projections: parameter grads and dX = dZ1 + dQ Wq + dK Wk + dV Wv ------------------------
      sas_accum_outer<D>(gp + Cfg::oWq, Q, X, n);
      sas_accum_colsum<D>(gp + Cfg::obq, Q, n);
      sas_accum_outer<D>(gp + Cfg::oWk, K, X, n);
      sas_accum_colsum<D>(gp + Cfg::obk, K, n);
      sas_accum_outer<D>(gp + Cfg::oWv, V, X, n);
      sas_accum_colsum<D>(gp + Cfg::obv, V, n);
      sas_backprop_linear<D>(G, Q, p.Wq, n);
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

__global__ __launch_bounds__(kBlock) void sas_reduce_partials_kernel(const float* __restrict__ p, int n_wg,
                                                                     int count, float* __restrict__ out) {
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < count; i += gridDim.x * kBlock) {
    float acc = 0.f;
    for (int w = 0; w < n_wg; ++w) acc += p[(size_t)w * count + i];
    out[i] = acc;
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

static int sas_grid(int B) { return B < 512 ? (B < 1 ? 1 : B) : 512; }

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
static int sas_launch_fwd(const SasArgs& a, bool save, hipStream_t s) {
  const size_t lds_bytes = (size_t)SasCfg<D>::kLdsFloats * sizeof(float);
  if (save) {
    auto kern = sasrec_fwd_kernel<D, true>;
    RC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    hipLaunchKernelGGL(kern, dim3(sas_grid(a.B)), dim3(kBlock), lds_bytes, s, a);
  } else {
    auto kern = sasrec_fwd_kernel<D, false>;
    RC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    hipLaunchKernelGGL(kern, dim3(sas_grid(a.B)), dim3(kBlock), lds_bytes, s, a);
  }
  RC_LAUNCH_CHECK();
  return RC_OK;
}

template <int D>
static int sas_launch_bwd(const SasArgs& a, float* dense_out, hipStream_t s) {
  const size_t lds_bytes = (size_t)SasCfg<D>::kLdsFloats * sizeof(float);
  const int n_wg = sas_grid(a.B);
  const int count = a.n_layers * SasCfg<D>::PL;
  RC_HIP(hipMemsetAsync(a.part, 0, (size_t)n_wg * count * sizeof(float), s));
  auto kern = sasrec_bwd_kernel<D>;
  RC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
  hipLaunchKernelGGL(kern, dim3(n_wg), dim3(kBlock), lds_bytes, s, a);
  RC_LAUNCH_CHECK();
  hipLaunchKernelGGL(sas_reduce_partials_kernel, dim3((count + kBlock - 1) / kBlock), dim3(kBlock), 0, s, a.part,
                     n_wg, count, dense_out);
  RC_LAUNCH_CHECK();
  return RC_OK;
}

}  // namespace rc

using namespace rc;

extern "C" int rc_sasrec_supported(int d, int n_layers, int n_heads, int L) {
  return ((d == 32 || d == 64) && n_layers >= 1 && n_layers <= kSasMaxLayers && n_heads >= 1 && d % n_heads == 0 &&
          L >= 1 && L <= kSasLP) ? 1 : 0;
}

extern "C" int rc_sasrec_dense_param_count(int d) { return 5 * d * d + 9 * d; }

extern "C" size_t rc_sasrec_workspace_bytes(int B, int d, int n_layers) {
  if (B < 1 || d < 1 || n_layers < 1) return 0;
  return align_up(sas_transpose_floats(d, n_layers) * sizeof(float), 256) +
         align_up((size_t)sas_grid(B) * n_layers * (5 * (size_t)d * d + 9 * d) * sizeof(float), 256) + 256;
}

extern "C" int rc_sasrec_fwd(const float* item_emb, const float* pos_emb, const float* const* layer_params,
                             int n_layers, int n_heads, const int64_t* hist, const int64_t* lengths, int B,
                             int L, int d, float* hv, float* xsave, void* ws, size_t ws_bytes,
                             rc_stream_t stream) {
  if (B == 0) return RC_OK;
  RC_REQUIRE(item_emb && pos_emb && hist && lengths && hv && ws, "rc_sasrec_fwd: null pointer");
  if (!rc_sasrec_supported(d, n_layers, n_heads, L))
    return fail(RC_ERR_UNSUPPORTED, "rc_sasrec_fwd: d=%d layers=%d heads=%d L=%d not supported (d in {32,64}, L<=%d)",
                d, n_layers, n_heads, L, kSasLP);
  SasArgs a;
  memset(&a, 0, sizeof(a));
  RC_TRY(sas_fill_layers(&a, layer_params, n_layers));
  a.item_emb = item_emb; a.pos_emb = pos_emb; a.n_heads = n_heads; a.hist = hist; a.lengths = lengths;
  a.B = B; a.L = L; a.hv = hv; a.xsave = xsave;
  hipStream_t s = as_stream(stream);
  if (ws_bytes < sas_transpose_floats(d, n_layers) * sizeof(float))
    return fail(RC_ERR_WORKSPACE, "rc_sasrec_fwd: workspace %zu too small (rc_sasrec_workspace_bytes)", ws_bytes);
  RC_TRY(sas_make_transposes(&a, d, reinterpret_cast<float*>(ws), s));
  return d == 64 ? sas_launch_fwd<64>(a, xsave != nullptr, s) : sas_launch_fwd<32>(a, xsave != nullptr, s);
}

extern "C" int rc_sasrec_bwd(const float* const* layer_params, int n_layers, int n_heads, const int64_t* lengths,
                             int B, int L, int d, const float* xsave, const float* dhv, float* g_hist,
                             float* dense_grads, void* ws, size_t ws_bytes, rc_stream_t stream) {
  if (B == 0) return RC_OK;
  RC_REQUIRE(lengths && xsave && dhv && g_hist && dense_grads && ws, "rc_sasrec_bwd: null pointer");
  if (!rc_sasrec_supported(d, n_layers, n_heads, L))
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
  a.part = reinterpret_cast<float*>(reinterpret_cast<char*>(ws) +
                                    align_up(sas_transpose_floats(d, n_layers) * sizeof(float), 256));
  return d == 64 ? sas_launch_bwd<64>(a, dense_grads, s) : sas_launch_bwd<32>(a, dense_grads, s);
}
