// tower_tail.hip -- the TAIL of an MLP tower at a small batch: the last hidden layer (width N2 <= 64) and the output Linear
// (N2 -> 1) of utils/layers.py:201-243 (MLP_Block of models/context/DeepFM.py:25, WideDeep.py:42-47 at --layers [512,64] or
// [64]; models/general/NeuMF.py:47-52 for a multi-layer tower) as ONE forward and ONE backward kernel.
//
// Why: at the reference's own CTR batch (B = 1,024, docs/demo_scripts_results/CTR_MIND.sh:8) these two layers hold 12 % of the
// tower's FLOPs and took 116 of its 222 us: a 64 x 64-tile GEMM has 16 output tiles for a [1024, 64] product, so each one went
// through split-K (product + epilogue kernel), the width -> 1 layer was a GEMM with 63 of 64 tile columns empty, the K = 1 and
// K = 64 backward products waited a full operand round trip per K step -- eleven launches of 5-15 us.  Here
//   rc_tower_tail_fwd:  H2 = drop(relu(X W2^T + b2)) [M, N2] and z = H2 w3 + b3 [M]
//   rc_tower_tail_bwd:  dZ2 = dz w3^T * mask(H2) (never stored), dX = (dZ2 W2) * mask(X), dW2 = dZ2^T X, db2, dw3 = H2^T dz, db3
// A workgroup owns 16 batch rows (one v_mfma_f32_16x16x4_f32 row block; exact fp32 FMA chains like every product of the engine).
// Forward: eight waves -- wave (w, kh) owns output columns 16 w .. 16 w + 15 and half kh of the reduction; both operands stream
// from L2 as one float4 per lane and K step of 16, all of a wave's steps requested before its first MFMA (K <= 512); the halves
// meet in LDS; bias / ReLU / dropout in registers (the mask of
// rc_linear_fwd: element (m, n) dropped iff word (m & 3) of Philox4x32-10(seed, (m >> 2, site 65536 + n)) < p 2^32 -- the four
// rows a lane holds share one Philox block), the output layer as a 16-lane DPP sum + a sum over the waves in LDS.
// Backward: the row block's dZ2 and X go to LDS once; product 1 (reduction over N2) walks 64-column groups, its epilogue
// applies the mask of the layer below (X is that layer's saved output) -- what rc_linear_bwd_chain does in its dX product;
// product 2 (reduction over the 16 rows) accumulates dW2 in registers across the row blocks of the workgroup; per-workgroup
// partials are summed in fixed order by tower_tail_reduce_kernel.  No float atomics.
#include "common.hpp"
#include "philox.hpp"

namespace rc {

typedef float tt4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ tt4 tt_mma(float a, float b, tt4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

constexpr int kTailRows = 16;       // batch rows of a workgroup's block
constexpr int kTailU = 8;           // K steps (of 16) per register set of the forward product
constexpr int kTailMaxWgs = 64;     // workgroups (= partials of the weight gradients) of the backward kernel

struct TailFwdArgs {
  const float* X;     // [M, K]
  const float* W2;    // [N2, K]
  const float* b2;    // [N2] or null
  const float* w3;    // [N2]
  const float* b3;    // [1] or null
  int64_t M;
  int K, N2;
  const uint64_t* seed;
  uint32_t drop_thresh;
  float keep_scale;
  uint32_t site;
  float* H2;          // [M, N2]
  float* z;           // [M]
};

__device__ __forceinline__ float4 tt_ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }

constexpr int kTailFwdThreads = 512;   // 8 waves: column slab w = wave & 3, half kh = wave >> 2 of the reduction

__global__ __launch_bounds__(kTailFwdThreads) void tower_tail_fwd_kernel(TailFwdArgs a) {
  __shared__ float zs[4][kTailRows];
  __shared__ tt4 half1[4][64];     // the accumulators of the second half of the reduction
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, i = lane & 15, g = lane >> 4;
  const int w = wave & 3, kh = wave >> 2;
  const int64_t r0 = (int64_t)blockIdx.x * kTailRows;
  const bool active = 16 * w < a.N2;
  tt4 acc = {0.f, 0.f, 0.f, 0.f};
  if (active) {
    int64_t row = r0 + i;
    if (row >= a.M) row = a.M - 1;                 // a tail row multiplies the last row; nothing of it is stored
    const float* xa = a.X + row * a.K + 4 * g;     // A(row i, k): X[r0 + i][k]
    const float* wb = a.W2 + (int64_t)(16 * w + i) * a.K + 4 * g;   // B(k, col i): W2[16 w + i][k]
    // MFMA e of K step s contracts k = 16 s + 4 g' + e for the four lane groups g' (both operands agree: any order of the
    // reduction index is a valid product).  K = 512: 16 steps per half, i.e. both register sets are requested before the first
    // MFMA -- ONE operand round trip per workgroup (four waves walking all 32 steps paid four: 11.4 us per launch).
    const int all_steps = a.K / 16, per = (all_steps + 1) / 2;
    const int s_lo = kh * per, n_steps = (s_lo + per < all_steps ? s_lo + per : all_steps);
    float4 xa0[kTailU], wb0[kTailU], xa1[kTailU], wb1[kTailU];
#pragma unroll
    for (int u = 0; u < kTailU; ++u)
      if (s_lo + u < n_steps) { xa0[u] = tt_ld4(xa + 16 * (s_lo + u)); wb0[u] = tt_ld4(wb + 16 * (s_lo + u)); }
    for (int s0 = s_lo; s0 < n_steps; s0 += 2 * kTailU) {
#pragma unroll
      for (int u = 0; u < kTailU; ++u)
        if (s0 + kTailU + u < n_steps) { xa1[u] = tt_ld4(xa + 16 * (s0 + kTailU + u)); wb1[u] = tt_ld4(wb + 16 * (s0 + kTailU + u)); }
#pragma unroll
      for (int u = 0; u < kTailU; ++u)
        if (s0 + u < n_steps) {
          acc = tt_mma(xa0[u].x, wb0[u].x, acc); acc = tt_mma(xa0[u].y, wb0[u].y, acc);
          acc = tt_mma(xa0[u].z, wb0[u].z, acc); acc = tt_mma(xa0[u].w, wb0[u].w, acc);
        }
#pragma unroll
      for (int u = 0; u < kTailU; ++u)
        if (s0 + 2 * kTailU + u < n_steps) { xa0[u] = tt_ld4(xa + 16 * (s0 + 2 * kTailU + u)); wb0[u] = tt_ld4(wb + 16 * (s0 + 2 * kTailU + u)); }
#pragma unroll
      for (int u = 0; u < kTailU; ++u)
        if (s0 + kTailU + u < n_steps) {
          acc = tt_mma(xa1[u].x, wb1[u].x, acc); acc = tt_mma(xa1[u].y, wb1[u].y, acc);
          acc = tt_mma(xa1[u].z, wb1[u].z, acc); acc = tt_mma(xa1[u].w, wb1[u].w, acc);
        }
    }
  }
  if (kh == 1) half1[w][lane] = acc;
  __syncthreads();
  // epilogue (first four waves): acc[r] is (row r0 + 4 g + r, column n = 16 w + i)
  float zp[4] = {0.f, 0.f, 0.f, 0.f};
  if (active && kh == 0) {
    const tt4 h = half1[w][lane];
    acc[0] += h[0]; acc[1] += h[1]; acc[2] += h[2]; acc[3] += h[3];
    const int n = 16 * w + i;
    const float bn = a.b2 ? a.b2[n] : 0.f, wn = a.w3[n];
    float keep[4] = {1.f, 1.f, 1.f, 1.f};
    if (a.seed) {
      uint32_t wd[4];
      philox4x32_10(*a.seed, (uint64_t)((r0 + 4 * g) >> 2), a.site * 65536u + (uint32_t)n, wd);
#pragma unroll
      for (int e = 0; e < 4; ++e) keep[e] = wd[e] < a.drop_thresh ? 0.f : a.keep_scale;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int64_t m = r0 + 4 * g + r;
      float v = fmaxf(acc[r] + bn, 0.f);
      if (a.seed) v *= keep[r];
      if (m < a.M) a.H2[m * a.N2 + n] = v;
      zp[r] = v * wn;
    }
  }
  if (kh == 0) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float t = row_allreduce_sum<16>(zp[r]);
      if (i == 0) zs[w][4 * g + r] = t;
    }
  }
  __syncthreads();
  if (threadIdx.x < kTailRows) {
    const int64_t m = r0 + threadIdx.x;
    float t = zs[0][threadIdx.x];
    for (int q = 1; 16 * q < a.N2; ++q) t += zs[q][threadIdx.x];
    if (a.b3) t += a.b3[0];
    if (m < a.M) a.z[m] = t;
  }
}

struct TailBwdArgs {
  const float* X;     // [M, K]  input of the hidden layer (the saved output of the layer below when x_act)
  const float* W2;    // [N2, K]
  const float* w3;    // [N2]
  const float* H2;    // [M, N2] saved output of the hidden layer: its own ReLU / dropout mask
  const float* dz;    // [M]     gradient of the tower's output
  int64_t M;
  int K, N2;
  float scale2;       // 1 / (1 - p) of the hidden layer's dropout
  int x_act;          // dX is multiplied by (X > 0 ? x_scale : 0)
  float x_scale;
  float* dX;          // [M, K] or null
  float* pW2;         // [grid][N2][K]
  float* pvec;        // [grid][2 N2 + 1]: db2 | dw3 | db3
};

template <int NA, int KS>   // N2 = 16 NA, K = 64 KS
__global__ __launch_bounds__(kBlock) void tower_tail_bwd_kernel(TailBwdArgs a) {
  constexpr int N2 = 16 * NA, K = 64 * KS, LZ = N2 + 4, LX = K + 4;
  extern __shared__ float lds[];
  float* Zs = lds;                    // [16][LZ]   dZ2 of the row block
  float* Xs = Zs + kTailRows * LZ;    // [16][LX]   X of the row block
  float* red = Xs + kTailRows * LX;   // [16][2 N2 + 1] (vector partials of the workgroup)
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, i = lane & 15, g = lane >> 4;
  const int64_t n_blocks = (a.M + kTailRows - 1) / kTailRows;
  tt4 acc2[KS][NA];   // dW2 tile (rows 16 a' + 4 g + r, columns 16 (wave + 4 j) + i)
#pragma unroll
  for (int j = 0; j < KS; ++j)
#pragma unroll
    for (int q = 0; q < NA; ++q) acc2[j][q] = tt4{0.f, 0.f, 0.f, 0.f};
  // the staging role of a thread: row tm, four columns 4 tn .. of dZ2 (fixed over the blocks: its sums stay in registers)
  const int tm = threadIdx.x >> 4, tn = threadIdx.x & 15;
  const bool zrole = 4 * tn < N2;
  float4 w3v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (zrole) w3v = tt_ld4(a.w3 + 4 * tn);
  float sb2[4] = {0.f, 0.f, 0.f, 0.f}, sw3[4] = {0.f, 0.f, 0.f, 0.f}, sb3 = 0.f;
  // Product 1's weights do not depend on the row block: requested ONCE, before anything is staged, one float4 per (n, 64-column
  // group) -- the lane's four columns 64 J + 4 i + c.  MFMA column i of "slab" c then stands for column 64 J + 4 i + c (any
  // assignment of columns to MFMA lanes is a valid product), so a lane ends up with four ADJACENT columns of its rows: float4
  // mask reads and float4 stores.  (First version: one scalar load per MFMA, requested slab by slab behind the barrier -- eight
  // dependent round trips, 24.5 us per launch.)
  constexpr int NJ = (KS + 3) / 4;    // 64-column groups J = wave + 4 jj of this wave
  float4 wv[NJ][NA][4];
  if (a.dX != nullptr) {
#pragma unroll
    for (int jj = 0; jj < NJ; ++jj)
      if (wave + 4 * jj < KS) {
#pragma unroll
        for (int q = 0; q < NA; ++q)
#pragma unroll
          for (int e = 0; e < 4; ++e) wv[jj][q][e] = tt_ld4(a.W2 + (int64_t)(16 * q + 4 * g + e) * K + 64 * (wave + 4 * jj) + 4 * i);
      }
  }

  for (int64_t blk = blockIdx.x; blk < n_blocks; blk += gridDim.x) {
    const int64_t r0 = blk * kTailRows;
    __syncthreads();     // the previous block's operand reads are done
    {   // stage X (zeros past the batch) and dZ2
      for (int t = threadIdx.x; t < kTailRows * (K / 4); t += kBlock) {
        const int m = t / (K / 4), c4 = t % (K / 4);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r0 + m < a.M) v = tt_ld4(a.X + (r0 + m) * K + 4 * c4);
        *reinterpret_cast<float4*>(Xs + m * LX + 4 * c4) = v;
      }
      if (zrole) {
        float4 zv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r0 + tm < a.M) {
          const float d = a.dz[r0 + tm];
          const float4 h = tt_ld4(a.H2 + (r0 + tm) * N2 + 4 * tn);
          zv.x = h.x > 0.f ? d * w3v.x * a.scale2 : 0.f;
          zv.y = h.y > 0.f ? d * w3v.y * a.scale2 : 0.f;
          zv.z = h.z > 0.f ? d * w3v.z * a.scale2 : 0.f;
          zv.w = h.w > 0.f ? d * w3v.w * a.scale2 : 0.f;
          sw3[0] = fmaf(d, h.x, sw3[0]); sw3[1] = fmaf(d, h.y, sw3[1]); sw3[2] = fmaf(d, h.z, sw3[2]); sw3[3] = fmaf(d, h.w, sw3[3]);
          sb2[0] += zv.x; sb2[1] += zv.y; sb2[2] += zv.z; sb2[3] += zv.w;
          if (tn == 0) sb3 += d;
        }
        *reinterpret_cast<float4*>(Zs + tm * LZ + 4 * tn) = zv;
      }
    }
    __syncthreads();
    // ---- product 1: dX[m, k] = sum_n dZ2[m, n] W2[n, k], masked by the layer below -----------------------------------------
    if (a.dX != nullptr) {
      float za[NA][4];
#pragma unroll
      for (int q = 0; q < NA; ++q) {
        const float4 v = *reinterpret_cast<const float4*>(Zs + i * LZ + 16 * q + 4 * g);
        za[q][0] = v.x; za[q][1] = v.y; za[q][2] = v.z; za[q][3] = v.w;
      }
#pragma unroll
      for (int jj = 0; jj < NJ; ++jj) {
        const int J = wave + 4 * jj;
        if (J < KS) {
          tt4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = c0, c2 = c0, c3 = c0;
#pragma unroll
          for (int q = 0; q < NA; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              c0 = tt_mma(za[q][e], wv[jj][q][e].x, c0);
              c1 = tt_mma(za[q][e], wv[jj][q][e].y, c1);
              c2 = tt_mma(za[q][e], wv[jj][q][e].z, c2);
              c3 = tt_mma(za[q][e], wv[jj][q][e].w, c3);
            }
          const int col = 64 * J + 4 * i;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int64_t m = r0 + 4 * g + r;
            float4 v = make_float4(c0[r], c1[r], c2[r], c3[r]);
            if (a.x_act) {
              const float4 x = *reinterpret_cast<const float4*>(Xs + (4 * g + r) * LX + col);
              v.x = x.x > 0.f ? v.x * a.x_scale : 0.f;
              v.y = x.y > 0.f ? v.y * a.x_scale : 0.f;
              v.z = x.z > 0.f ? v.z * a.x_scale : 0.f;
              v.w = x.w > 0.f ? v.w * a.x_scale : 0.f;
            }
            if (m < a.M) *reinterpret_cast<float4*>(a.dX + m * K + col) = v;
          }
        }
      }
    }
    // ---- product 2: dW2[n, k] += sum over the block's rows of dZ2[m, n] X[m, k] ----------------------------------------------
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      float av[NA], bv[KS];
#pragma unroll
      for (int q = 0; q < NA; ++q) av[q] = Zs[(4 * s + g) * LZ + 16 * q + i];
#pragma unroll
      for (int j = 0; j < KS; ++j) bv[j] = Xs[(4 * s + g) * LX + 16 * (wave + 4 * j) + i];
#pragma unroll
      for (int j = 0; j < KS; ++j)
#pragma unroll
        for (int q = 0; q < NA; ++q) acc2[j][q] = tt_mma(av[q], bv[j], acc2[j][q]);
    }
  }
  // ---- per-workgroup partials ------------------------------------------------------------------------------------------
  float* pw = a.pW2 + (size_t)blockIdx.x * N2 * K;
#pragma unroll
  for (int j = 0; j < KS; ++j)
#pragma unroll
    for (int q = 0; q < NA; ++q)
#pragma unroll
      for (int r = 0; r < 4; ++r) pw[(size_t)(16 * q + 4 * g + r) * K + 16 * (wave + 4 * j) + i] = acc2[j][q][r];
  __syncthreads();
  if (zrole) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      red[tm * (2 * N2 + 1) + 4 * tn + e] = sb2[e];
      red[tm * (2 * N2 + 1) + N2 + 4 * tn + e] = sw3[e];
    }
    if (tn == 0) red[tm * (2 * N2 + 1) + 2 * N2] = sb3;
  }
  __syncthreads();
  for (int k = threadIdx.x; k < 2 * N2 + 1; k += kBlock) {
    float t = 0.f;
    for (int m = 0; m < kTailRows; ++m) t += red[m * (2 * N2 + 1) + k];
    a.pvec[(size_t)blockIdx.x * (2 * N2 + 1) + k] = t;
  }
}

// out = sum over the workgroups' partials in a FIXED order: wave q of a workgroup adds the q-th quarter of the partials in
// ascending order (independent loads, sixteen in flight), the four quarter sums are added in order q = 0..3.  (One thread per
// element walking all 64 partials one load after the other took 16.8 us for the 64 x 512 weight gradient.)
__global__ __launch_bounds__(kBlock) void tower_tail_reduce_kernel(const float* __restrict__ pW2, const float* __restrict__ pvec, int parts,
                                                                  int N2, int K, float* __restrict__ dW2, float* __restrict__ db2,
                                                                  float* __restrict__ dw3, float* __restrict__ db3) {
  __shared__ float4 sm[4][64];
  const int q = threadIdx.x >> 6, o = threadIdx.x & 63;
  const int64_t nw4 = (int64_t)N2 * K / 4;
  const int per = (parts + 3) / 4, p0 = q * per, p1 = (p0 + per < parts ? p0 + per : parts);
  const float4* src = reinterpret_cast<const float4*>(pW2);
  for (int64_t base = (int64_t)blockIdx.x * 64; base < nw4; base += (int64_t)gridDim.x * 64) {
    const int64_t idx = base + o;
    float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
    if (idx < nw4) {
#pragma unroll 16
      for (int p = p0; p < p1; ++p) {
        const float4 v = src[(size_t)p * nw4 + idx];
        t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
      }
    }
    sm[q][o] = t;
    __syncthreads();
    if (q == 0 && idx < nw4) {
      const float4 a = sm[0][o], b = sm[1][o], c = sm[2][o], d = sm[3][o];
      reinterpret_cast<float4*>(dW2)[idx] = make_float4(((a.x + b.x) + c.x) + d.x, ((a.y + b.y) + c.y) + d.y, ((a.z + b.z) + c.z) + d.z,
                                                        ((a.w + b.w) + c.w) + d.w);
    }
    __syncthreads();
  }
  if (blockIdx.x == gridDim.x - 1) {   // the three small vectors, same scheme (a thread walking all partials one dependent load
                                      // after the other made this block the kernel's critical path: 18 us)
    __shared__ float sv[4][64];
    const int nv = 2 * N2 + 1;
    for (int base = 0; base < nv; base += 64) {
      const int v = base + o;
      float t = 0.f;
      if (v < nv) {
        for (int pb = p0; pb < p1; pb += 16) {
          float tmp[16];
#pragma unroll
          for (int u = 0; u < 16; ++u) tmp[u] = pb + u < p1 ? pvec[(size_t)(pb + u) * nv + v] : 0.f;
#pragma unroll
          for (int u = 0; u < 16; ++u) t += tmp[u];
        }
      }
      sv[q][o] = t;
      __syncthreads();
      if (q == 0 && v < nv) {
        const float r = ((sv[0][o] + sv[1][o]) + sv[2][o]) + sv[3][o];
        if (v < N2) { if (db2) db2[v] = r; }
        else if (v < 2 * N2) dw3[v - N2] = r;
        else if (db3) db3[0] = r;
      }
      __syncthreads();
    }
  }
}

static int tail_parts(int64_t M) {
  const int64_t blocks = (M + kTailRows - 1) / kTailRows;
  return (int)(blocks < kTailMaxWgs ? blocks : kTailMaxWgs);
}

template <int NA, int KS>
static int launch_tail_bwd(const TailBwdArgs& a, int grid, hipStream_t s) {
  const size_t lds = sizeof(float) * (size_t)(kTailRows * (16 * NA + 4) + kTailRows * (64 * KS + 4) + kTailRows * (2 * 16 * NA + 1));
  auto kern = tower_tail_bwd_kernel<NA, KS>;
  RC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(kern, dim3(grid), dim3(kBlock), lds, s, a);
  RC_LAUNCH_CHECK();
  return RC_OK;
}

}  // namespace rc

using namespace rc;

extern "C" int rc_tower_tail_supported(int64_t M, int K, int N2) {
  const bool n_ok = N2 == 16 || N2 == 32 || N2 == 64;
  const bool k_ok = K == 64 || K == 128 || K == 256 || K == 512;
  return (M >= 1 && n_ok && k_ok) ? 1 : 0;
}

extern "C" size_t rc_tower_tail_workspace_bytes(int64_t M, int K, int N2) {
  if (!rc_tower_tail_supported(M, K, N2)) return 0;
  const size_t parts = (size_t)tail_parts(M);
  return align_up(parts * (size_t)N2 * K * sizeof(float), 256) + align_up(parts * (size_t)(2 * N2 + 1) * sizeof(float), 256);
}

extern "C" int rc_tower_tail_fwd(const float* X, const float* W2, const float* b2, const float* w3, const float* b3, int64_t M, int K,
                                 int N2, float drop_p, const uint64_t* seed_dev, uint32_t site, float* H2, float* z, rc_stream_t stream) {
  if (M == 0) return RC_OK;
  RC_REQUIRE(X && W2 && w3 && H2 && z, "rc_tower_tail_fwd: null pointer");
  if (!rc_tower_tail_supported(M, K, N2))
    return fail(RC_ERR_UNSUPPORTED, "rc_tower_tail_fwd: M=%lld K=%d N2=%d not supported (N2 in {16,32,64}, K in {64,128,256,512})", (long long)M, K, N2);
  RC_REQUIRE(reinterpret_cast<uintptr_t>(X) % 16 == 0 && reinterpret_cast<uintptr_t>(W2) % 16 == 0, "rc_tower_tail_fwd: X / W2 must be 16-byte aligned");
  RC_REQUIRE(drop_p >= 0.f && drop_p < 1.f && (drop_p == 0.f || seed_dev), "rc_tower_tail_fwd: dropout p=%g needs p in [0, 1) and a device seed", (double)drop_p);
  TailFwdArgs a;
  memset(&a, 0, sizeof(a));
  a.X = X; a.W2 = W2; a.b2 = b2; a.w3 = w3; a.b3 = b3; a.M = M; a.K = K; a.N2 = N2; a.site = site; a.H2 = H2; a.z = z;
  if (drop_p > 0.f) {
    a.seed = seed_dev;
    a.drop_thresh = (uint32_t)((double)drop_p * 4294967296.0);
    a.keep_scale = 1.0f / (1.0f - drop_p);
  }
  const int64_t blocks = (M + kTailRows - 1) / kTailRows;
  RC_REQUIRE(blocks < kMaxGridX, "rc_tower_tail_fwd: batch too large");
  hipLaunchKernelGGL(tower_tail_fwd_kernel, dim3((unsigned)blocks), dim3(kTailFwdThreads), 0, as_stream(stream), a);
  RC_LAUNCH_CHECK();
  return RC_OK;
}

extern "C" int rc_tower_tail_bwd(const float* X, const float* W2, const float* w3, const float* H2, const float* dz, int64_t M, int K,
                                 int N2, float drop_p, int x_act, float x_drop_p, float* dX, float* dW2, float* db2, float* dw3,
                                 float* db3, void* ws, size_t ws_bytes, rc_stream_t stream) {
  RC_REQUIRE(X && W2 && w3 && H2 && dz && dW2 && dw3, "rc_tower_tail_bwd: null pointer");
  if (!rc_tower_tail_supported(M > 0 ? M : 1, K, N2))
    return fail(RC_ERR_UNSUPPORTED, "rc_tower_tail_bwd: M=%lld K=%d N2=%d not supported (N2 in {16,32,64}, K in {64,128,256,512})", (long long)M, K, N2);
  hipStream_t s = as_stream(stream);
  if (M == 0) {
    RC_HIP(hipMemsetAsync(dW2, 0, (size_t)N2 * K * sizeof(float), s));
    RC_HIP(hipMemsetAsync(dw3, 0, (size_t)N2 * sizeof(float), s));
    if (db2) RC_HIP(hipMemsetAsync(db2, 0, (size_t)N2 * sizeof(float), s));
    if (db3) RC_HIP(hipMemsetAsync(db3, 0, sizeof(float), s));
    return RC_OK;
  }
  RC_REQUIRE(ws != nullptr && ws_bytes >= rc_tower_tail_workspace_bytes(M, K, N2), "rc_tower_tail_bwd: workspace %zu < %zu", ws_bytes,
             rc_tower_tail_workspace_bytes(M, K, N2));
  RC_REQUIRE(drop_p >= 0.f && drop_p < 1.f && x_drop_p >= 0.f && x_drop_p < 1.f, "rc_tower_tail_bwd: dropout p outside [0, 1)");
  RC_REQUIRE(reinterpret_cast<uintptr_t>(X) % 16 == 0 && reinterpret_cast<uintptr_t>(H2) % 16 == 0 && reinterpret_cast<uintptr_t>(w3) % 16 == 0 &&
                 reinterpret_cast<uintptr_t>(W2) % 16 == 0 && reinterpret_cast<uintptr_t>(dW2) % 16 == 0 && (dX == nullptr || reinterpret_cast<uintptr_t>(dX) % 16 == 0),
             "rc_tower_tail_bwd: X / W2 / H2 / w3 / dX must be 16-byte aligned");
  const int parts = tail_parts(M);
  TailBwdArgs a;
  memset(&a, 0, sizeof(a));
  a.X = X; a.W2 = W2; a.w3 = w3; a.H2 = H2; a.dz = dz; a.M = M; a.K = K; a.N2 = N2;
  a.scale2 = 1.0f / (1.0f - drop_p);
  a.x_act = x_act ? 1 : 0;
  a.x_scale = 1.0f / (1.0f - x_drop_p);
  a.dX = dX;
  a.pW2 = static_cast<float*>(ws);
  a.pvec = reinterpret_cast<float*>(static_cast<char*>(ws) + align_up((size_t)parts * (size_t)N2 * K * sizeof(float), 256));
  int rc = RC_ERR_UNSUPPORTED;
#define RC_TT(NA_, KS_) \
  if (N2 == 16 * NA_ && K == 64 * KS_) rc = launch_tail_bwd<NA_, KS_>(a, parts, s)
  RC_TT(1, 1); RC_TT(1, 2); RC_TT(1, 4); RC_TT(1, 8);
  RC_TT(2, 1); RC_TT(2, 2); RC_TT(2, 4); RC_TT(2, 8);
  RC_TT(4, 1); RC_TT(4, 2); RC_TT(4, 4); RC_TT(4, 8);
#undef RC_TT
  RC_TRY(rc);
  int64_t blocks = ((int64_t)N2 * K / 4 + 63) / 64;
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(tower_tail_reduce_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, s, a.pW2, a.pvec, parts, N2, K, dW2, db2, dw3, db3);
  RC_LAUNCH_CHECK();
  return RC_OK;
}
