// mlp.hip -- the dense layers of the ranking heads as hand-written fp32 MFMA GEMMs (v_mfma_f32_32x32x2_f32: exact
// f32 FMA chains -- gfx950 has no TF32, and the 1e-5 parity bar rules out bf16).
//
// Reference: utils/layers.py:201-243 (MLP_Block: Linear -> ReLU -> Dropout per hidden layer, Linear output layer; the
// deep part of models/context/DeepFM.py:25, WideDeep.py:42-47 at --layers [512,64]) and models/general/NeuMF.py:47-52,
// 69-72 (the MLP tower `for layer in self.mlp: h = dropout(relu(layer(h)))` for any --layers) with the autograd of
// nn.Linear.  Replaces aten::addmm / mm (rocBLAS) + relu + dropout + their backward kernels.
//
//   rc_linear_fwd   Y = drop(relu(X W^T + b))                    one GEMM, bias / ReLU / dropout in the epilogue
//   rc_linear_bwd   dZ = dY * (Y > 0 ? 1/(1-p) : 0)               (elementwise; the saved output IS the mask)
//                   dX = dZ W                                     GEMM
//                   [dW | db] = dZ^T [X | 1]                      GEMM over the batch, split into row ranges whose
//                                                                 partial sums are combined in a fixed order
//
// One kernel serves the three products: C[M x N] = A[M x K] B[K x N] with either operand stored reduction-major
// or reduction-minor.  Tile 64 x 64 per workgroup (2 x 2 waves, one 32 x 32 MFMA block each), K step 32: the tiles are
// staged in LDS reduction-major ([k][i], row stride 68), K step 32 (two 16-row loader passes), so an MFMA operand fetch is one conflict-free ds_read_b32
// across the lanes; the next K step's global loads are in flight while the current one is multiplied.
#include <mutex>

#include "common.hpp"
#include "philox.hpp"

namespace rc {

typedef float mlp_f32x16 __attribute__((ext_vector_type(16)));

constexpr int kMlpBM = 64, kMlpBN = 64, kMlpBK = 32, kMlpLD = 68;   // (K step 16: 22.8 us per GEMM at B = 1024 -- one L2 round trip per 8 MFMAs)
constexpr int kMlpSub = 16;   // k rows one pass of the loaders covers; a K step is kMlpBK / kMlpSub passes
constexpr int kMlpPD = 3;     // K steps in flight between global memory and LDS (register ring)
// (tried and dropped, round 3: reduction-major operands transposed while they are staged so that every operand fetch is a
//  ds_read_b128: 5.14 instead of 4.82 ms per DeepFM step at B = 131,072 -- the four strided ds_write_b32 cost more than the reads save;
//  the operands of step s + 1 read from LDS into a second register set before the MFMAs of step s --
//  2,090 instead of 1,890 cycles per K step, the compiler drains the reads before the first MFMA either way; two accumulator
//  chains instead of one: no change, the dependent-issue gap is not what the step waits for)

struct MlpOperand {
  const float* p;
  int64_t ld;       // elements between consecutive rows of the STORED matrix
  int k_major;      // 1: stored [outer][k] (k contiguous);  0: stored [k][outer] (outer contiguous)
};

struct MlpGemm {
  MlpOperand A, B;          // A(i, k), B(k, j)
  int64_t M;
  int N, K;                 // K: reduction length of ONE split
  int ones_col;             // >= 0: B(k, ones_col) = 1 for every k (bias gradient rides along), columns beyond it 0
  float* C;                 // [splits][M][ldc]
  int64_t ldc;
  int64_t split_stride_k;   // reduction offset between splits (blockIdx.z)
  int64_t k_total;          // full reduction length (the last split may be short)
  // epilogue
  const float* bias;        // [N] or null
  int relu;
  const uint64_t* seed;     // dropout: device seed, null = off
  uint32_t drop_thresh;
  float keep_scale;
  uint32_t site;
  // dX products: the output is the gradient w.r.t. a drop(relu(.)) activation whose saved value is `mask` ([M][ldc], same layout
  // as C): C = acc * (mask > 0 ? mask_scale : 0) -- the separate masking pass of the layer below, fused (null: off)
  const float* mask;
  float mask_scale;
};

// dropout mask of element (m, n): dropped iff word (m & 3) of Philox4x32-10(key = seed, counter = (m >> 2, site * 65536 + n))
// < p * 2^32 -- four consecutive rows share one Philox call, which is how an MFMA lane holds its accumulator rows
__device__ __forceinline__ void mlp_keep4(const MlpGemm& g, uint64_t seed, int64_t m4, int n, float keep[4]) {
  uint32_t w[4];
  philox4x32_10(seed, (uint64_t)m4, g.site * 65536u + (uint32_t)n, w);
#pragma unroll
  for (int e = 0; e < 4; ++e) keep[e] = w[e] < g.drop_thresh ? 0.f : g.keep_scale;
}

// global -> registers: this thread's float4 of the tile starting at (outer0, k0) of an operand
__device__ __forceinline__ float4 mlp_load4(const MlpOperand& o, int64_t outer0, int64_t outer_n, int64_t k0, int64_t k_end,
                                            int ones_col, bool vec_ok) {
  const int t = threadIdx.x;
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (o.k_major) {  // thread -> (outer = t / 4, k = 4 (t % 4) ..)
    const int64_t i = outer0 + (t >> 2), k = k0 + 4 * (t & 3);
    if (i < outer_n) {
      const float* src = o.p + i * o.ld + k;
      if (vec_ok && k + 3 < k_end) v = *reinterpret_cast<const float4*>(src);
      else {
        if (k < k_end) v.x = src[0];
        if (k + 1 < k_end) v.y = src[1];
        if (k + 2 < k_end) v.z = src[2];
        if (k + 3 < k_end) v.w = src[3];
      }
    }
  } else {          // thread -> (k = t / 16, outer = 4 (t % 16) ..)
    const int64_t k = k0 + (t >> 4), i = outer0 + 4 * (t & 15);
    if (k < k_end) {
      const float* src = o.p + k * o.ld + i;
      if (vec_ok && i + 3 < outer_n) v = *reinterpret_cast<const float4*>(src);
      else {
        if (i < outer_n) v.x = src[0];
        if (i + 1 < outer_n) v.y = src[1];
        if (i + 2 < outer_n) v.z = src[2];
        if (i + 3 < outer_n) v.w = src[3];
      }
      if (ones_col >= 0) {  // the column of ones (and nothing beyond it)
        if (i == ones_col) v.x = 1.f;
        if (i + 1 == ones_col) v.y = 1.f;
        if (i + 2 == ones_col) v.z = 1.f;
        if (i + 3 == ones_col) v.w = 1.f;
      }
    }
  }
  return v;
}

// registers -> LDS tile [k][outer] (row stride kMlpLD)
__device__ __forceinline__ void mlp_stage(float* tile, const MlpOperand& o, const float4& v) {
  const int t = threadIdx.x;
  if (o.k_major) {
    const int i = t >> 2, k = 4 * (t & 3);
    tile[(k + 0) * kMlpLD + i] = v.x;
    tile[(k + 1) * kMlpLD + i] = v.y;
    tile[(k + 2) * kMlpLD + i] = v.z;
    tile[(k + 3) * kMlpLD + i] = v.w;
  } else {
    const int k = t >> 4, i = 4 * (t & 15);
    *reinterpret_cast<float4*>(tile + k * kMlpLD + i) = v;
  }
}

// The generic loaders above cost ~60 VALU instructions per float4 (64-bit index products, range checks); with one wave per
// SIMD -- a 1,024-row product is at most one workgroup per CU -- that arithmetic is fully exposed: device timestamps showed
// 3,500 shader cycles per K step against 1,024 of MFMA work.  A thread's float4 of consecutive K steps lies a constant stride
// apart, so its address and its range class are fixed before the loop: 0 = outside the matrix (always zero), 1 = one aligned
// float4 whenever the K step is complete, 2 = edge / unaligned (the generic loader).
// 3 = four columns past the matrix: a constant (zeros, or the 1 of the column of ones that carries the bias gradient -- the tile
// that holds nothing but that column ran the generic loader for every float4 and set the duration of every weight-gradient
// product: 20-25 us instead of ~9 at B = 1,024).
struct MlpFastSrc {
  const float* p;
  int mode;
  float c0, c1, c2, c3;   // mode 3: the constant float4 (four scalars: a float4 member was kept in a stack slot -- 80 bytes of scratch)
};
__device__ __forceinline__ float4 mlp_cst(const MlpFastSrc& f) { return make_float4(f.c0, f.c1, f.c2, f.c3); }
__device__ __forceinline__ MlpFastSrc mlp_fast_src(const MlpOperand& o, int64_t outer, int64_t outer_n, int64_t k, bool vec_ok,
                                                   int ones_col = -1) {
  MlpFastSrc f;
  f.c0 = f.c1 = f.c2 = f.c3 = 0.f;
  if (o.k_major) {   // thread's float4 runs along k
    f.p = o.p + outer * o.ld + k;
    f.mode = outer >= outer_n ? 0 : (vec_ok ? 1 : 2);
  } else {           // along outer, four columns outer .. outer + 3 of row k
    f.p = o.p + k * o.ld + outer;
    if (outer >= outer_n) {
      f.mode = 3;
      f.c0 = outer == ones_col ? 1.f : 0.f;
      f.c1 = outer + 1 == ones_col ? 1.f : 0.f;
      f.c2 = outer + 2 == ones_col ? 1.f : 0.f;
      f.c3 = outer + 3 == ones_col ? 1.f : 0.f;
    } else {
      f.mode = (vec_ok && outer + 3 < outer_n) ? 1 : 2;
    }
  }
  return f;
}

// LDS tile of one operand, two layouts.  Stored reduction-major ([k][outer] in memory): staged as it comes, [k][outer] with row
// stride 68, one ds_write_b128 per thread, the MFMA operand of reduction index k is a conflict-free ds_read_b32 of row k.
// Stored reduction-minor ([outer][k], k contiguous -- X and W of the forward product): staged as it comes as well, [outer][k]
// with row stride 36 (round 2 transposed it with four scalar ds_write_b32 per float4), and a lane fetches FOUR reduction
// indices of its row with one ds_read_b128 (the sixteen lanes of a pass start in sixteen different 4-bank groups).  The k-th
// MFMA of a K step may contract any reduction index as long as both operands agree: MFMA (u, e), u = 0..7, e = 0..3 of a
// 32-wide step takes k = 8 u + 4 kh + e (kh = lane >> 5).
constexpr int kMlpLDK = kMlpBK + 4;                        // row stride of the [outer][k] layout
constexpr int kMlpTile = kMlpBM * kMlpLDK > kMlpBK * kMlpLD ? kMlpBM * kMlpLDK : kMlpBK * kMlpLD;   // floats per buffer

template <bool KM>
__device__ __forceinline__ void mlp_stage_t(float* tile, const float4& v, int q) {
  const int t = threadIdx.x;
  if (KM) *reinterpret_cast<float4*>(tile + (t >> 2) * kMlpLDK + 4 * (t & 3) + q * kMlpSub) = v;
  else *reinterpret_cast<float4*>(tile + ((t >> 4) + q * kMlpSub) * kMlpLD + 4 * (t & 15)) = v;
}
// the four operand values of MFMAs (u, 0..3) for this lane: row / column `o` of the tile
template <bool KM>
__device__ __forceinline__ void mlp_operand4(const float* tile, int o, int kh, int u, float (&x)[4]) {
  if (KM) {
    const float4 v = *reinterpret_cast<const float4*>(tile + o * kMlpLDK + 8 * u + 4 * kh);
    x[0] = v.x; x[1] = v.y; x[2] = v.z; x[3] = v.w;
  } else {
    const float* p = tile + (8 * u + 4 * kh) * kMlpLD + o;
    x[0] = p[0]; x[1] = p[kMlpLD]; x[2] = p[2 * kMlpLD]; x[3] = p[3 * kMlpLD];
  }
}

template <bool AKM, bool BKM>
__global__ __launch_bounds__(kBlock) void mlp_gemm_kernel(MlpGemm g, int vec_a, int vec_b) {
  __shared__ __attribute__((aligned(16))) float As[2][kMlpTile];
  __shared__ __attribute__((aligned(16))) float Bs[2][kMlpTile];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wr = wave >> 1, wc = wave & 1;                     // this wave's 32 x 32 block of the 64 x 64 tile
  const int64_t m0 = (int64_t)blockIdx.x * kMlpBM;
  const int64_t n0 = (int64_t)blockIdx.y * kMlpBN;
  const int64_t kb = (int64_t)blockIdx.z * g.split_stride_k;
  const int64_t ke = (kb + g.K < g.k_total) ? kb + g.K : g.k_total;
  // the B operand's "outer" extent: N real columns, + 1 when the ones column rides along
  const int64_t bn = g.ones_col >= 0 ? (int64_t)g.ones_col : (int64_t)g.N;
  mlp_f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;

  constexpr int NP = kMlpBK / kMlpSub;   // loader passes per K step
  // Register ring of kMlpPD K steps: at M = 1,024 a product is 16..128 workgroups, at most one per CU, so nothing but the
  // workgroup's own prefetch covers the L2 / HBM round trip (~1-2 us against 0.43 us of MFMA work per K step).  One step
  // ahead (round 2) left every K step waiting for its operands: 2.2 us per step, 36 us for a 1,024 x 512 x 512 product.
  float4 ra[kMlpPD][NP], rb[kMlpPD][NP];
  MlpFastSrc fa[NP], fb[NP];
  {
    const int t = threadIdx.x;
#pragma unroll
    for (int q = 0; q < NP; ++q) {
      fa[q] = g.A.k_major ? mlp_fast_src(g.A, m0 + (t >> 2), g.M, kb + 4 * (t & 3) + q * kMlpSub, vec_a != 0)
                          : mlp_fast_src(g.A, m0 + 4 * (t & 15), g.M, kb + (t >> 4) + q * kMlpSub, vec_a != 0);
      fb[q] = g.B.k_major ? mlp_fast_src(g.B, n0 + (t >> 2), bn, kb + 4 * (t & 3) + q * kMlpSub, vec_b != 0)
                          : mlp_fast_src(g.B, n0 + 4 * (t & 15), bn, kb + (t >> 4) + q * kMlpSub, vec_b != 0, g.ones_col);
    }
  }
  const int64_t stride_a = g.A.k_major ? (int64_t)kMlpBK : (int64_t)kMlpBK * g.A.ld;
  const int64_t stride_b = g.B.k_major ? (int64_t)kMlpBK : (int64_t)kMlpBK * g.B.ld;
  // tile of K step `step` (its first reduction index k0, float offsets off_a / off_b from the step-0 addresses) -> registers
  // interior tile: every thread's float4s are plain aligned loads -- ONE uniform branch per K step instead of per-lane classes
  bool mine_fast = true;
#pragma unroll
  for (int q = 0; q < NP; ++q) mine_fast = mine_fast && fa[q].mode != 2 && fb[q].mode != 2;
  const bool all_fast = __syncthreads_and(mine_fast ? 1 : 0) != 0;
  auto fetch = [&](float4 (&va)[NP], float4 (&vb)[NP], int64_t k0, int64_t off_a, int64_t off_b) {
    const bool full = k0 + kMlpBK <= ke;   // workgroup-uniform
    if (all_fast && full) {
#pragma unroll
      for (int q = 0; q < NP; ++q) {
        va[q] = fa[q].mode == 1 ? *reinterpret_cast<const float4*>(fa[q].p + off_a) : mlp_cst(fa[q]);
        vb[q] = fb[q].mode == 1 ? *reinterpret_cast<const float4*>(fb[q].p + off_b) : mlp_cst(fb[q]);
      }
      return;
    }
#pragma unroll
    for (int q = 0; q < NP; ++q) {
      if (fa[q].mode == 0 || k0 >= ke) va[q] = make_float4(0.f, 0.f, 0.f, 0.f);
      else if (full && fa[q].mode == 1) va[q] = *reinterpret_cast<const float4*>(fa[q].p + off_a);
      else va[q] = mlp_load4(g.A, m0, g.M, k0 + q * kMlpSub, ke, -1, vec_a != 0);
      if (fb[q].mode == 0 || k0 >= ke) vb[q] = make_float4(0.f, 0.f, 0.f, 0.f);
      else if (full && fb[q].mode == 1) vb[q] = *reinterpret_cast<const float4*>(fb[q].p + off_b);
      else vb[q] = mlp_load4(g.B, n0, bn, k0 + q * kMlpSub, ke, g.ones_col, vec_b != 0);
    }
  };
  int64_t off_a = 0, off_b = 0;   // offsets of the NEXT tile to request
#pragma unroll
  for (int sl = 0; sl < kMlpPD; ++sl) {
    fetch(ra[sl], rb[sl], kb + (int64_t)sl * kMlpBK, off_a, off_b);
    off_a += stride_a;
    off_b += stride_b;
  }
#ifdef RC_X_TIMING
  const uint64_t t_w0 = wall_clock64(), t_c0 = clock64();
#endif
  int buf = 0;
#pragma unroll
  for (int q = 0; q < NP; ++q) {
    mlp_stage_t<AKM>(As[0], ra[0][q], q);
    mlp_stage_t<BKM>(Bs[0], rb[0][q], q);
  }
  __syncthreads();
  const int ai = wr * 32 + (lane & 31), bj = wc * 32 + (lane & 31), kh = lane >> 5;
  for (int64_t kbase = kb; kbase < ke; kbase += (int64_t)kMlpPD * kMlpBK) {
#pragma unroll
    for (int sl = 0; sl < kMlpPD; ++sl) {     // ring slot sl holds the tile of K step (kbase - kb) / kMlpBK + sl
      const int64_t k0 = kbase + (int64_t)sl * kMlpBK;
      if (k0 < ke) {                           // workgroup-uniform
        const bool more = k0 + kMlpBK < ke;
        // slot sl was staged before this step's barrier: refill it with the tile kMlpPD steps ahead
        fetch(ra[sl], rb[sl], k0 + (int64_t)kMlpPD * kMlpBK, off_a, off_b);
        off_a += stride_a;
        off_b += stride_b;
        float av[kMlpBK / 8][4], bv[kMlpBK / 8][4];
#pragma unroll
        for (int u = 0; u < kMlpBK / 8; ++u) {
          mlp_operand4<AKM>(As[buf], ai, kh, u, av[u]);
          mlp_operand4<BKM>(Bs[buf], bj, kh, u, bv[u]);
        }
#pragma unroll
        for (int u = 0; u < kMlpBK / 8; ++u)
#pragma unroll
          for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u][e], bv[u][e], acc, 0, 0, 0);
        if (more) {
#pragma unroll
          for (int q = 0; q < NP; ++q) {
            mlp_stage_t<AKM>(As[buf ^ 1], ra[(sl + 1) % kMlpPD][q], q);
            mlp_stage_t<BKM>(Bs[buf ^ 1], rb[(sl + 1) % kMlpPD][q], q);
          }
        }
        __syncthreads();
        buf ^= 1;
      }
    }
  }

#ifdef RC_X_TIMING
  if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0)
    printf("mlp_gemm M %lld N %d K %lld grid %u x %u x %u: K loop %llu ticks (100 MHz), %llu shader cycles, %lld steps\n", (long long)g.M, g.N,
           (long long)(ke - kb), gridDim.x, gridDim.y, gridDim.z, (unsigned long long)(wall_clock64() - t_w0),
           (unsigned long long)(clock64() - t_c0), (long long)((ke - kb + kMlpBK - 1) / kMlpBK));
#endif
  // ---- epilogue: acc[r] is C(m0 + wr*32 + (r & 3) + 8 (r >> 2) + 4 kh, n0 + wc*32 + (lane & 31))
  const int64_t j = n0 + wc * 32 + (lane & 31);
  const int64_t ncols = g.ones_col >= 0 ? (int64_t)g.ones_col + 1 : (int64_t)g.N;
  if (j >= ncols) return;
  const float bj_ = (g.bias != nullptr && j < g.N) ? g.bias[j] : 0.f;
  const uint64_t seed = g.seed ? *g.seed : 0;
  float* C = g.C + (size_t)blockIdx.z * (size_t)g.M * (size_t)g.ldc;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int64_t mb = m0 + wr * 32 + 8 * q + 4 * kh;    // rows mb .. mb + 3 (mb is a multiple of 4)
    float keep[4] = {1.f, 1.f, 1.f, 1.f};
    if (g.seed && mb < g.M) mlp_keep4(g, seed, mb >> 2, (int)j, keep);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int64_t m = mb + e;
      if (m >= g.M) continue;
      float v = acc[4 * q + e] + bj_;
      if (g.relu) v = fmaxf(v, 0.f);
      if (g.seed) v *= keep[e];
      if (g.mask) v = g.mask[m * g.ldc + j] > 0.f ? v * g.mask_scale : 0.f;
      C[m * g.ldc + j] = v;
    }
  }
}

// ---- 32 x 64 tiles for the forward / dX products of a SMALL batch ---------------------------------------------------------------
// A 1,024 x 512 x 512 product is 128 tiles of 64 x 64: half the CUs idle, each wave walking 256 MFMAs (6.8 us of matrix pipe) --
// which is why round 2 cut the reduction into splits (512 workgroups, short K loops) and paid a second launch that sums the
// partial planes and applies the epilogue: 15 + 7 us per product at the reference's CTR batch, where a launch costs 4.5 us whatever it
// does.  Here a workgroup owns 32 x 64 outputs (256 workgroups for that product: one per CU, every SIMD busy), a wave 16 x 32 as two
// v_mfma_f32_16x16x4_f32 blocks over the FULL reduction (3.4 us of matrix pipe), epilogue in the kernel: one launch, no partial planes.
// A is stored [row][reduction] (X of the forward product, dZ of the dX product); B either way (W[n][k] forward, W[red][k] for dX).
// LDS: A [32][36], B [64][36] or [32][68] per buffer, two buffers; K step 32 = two groups of 16 (MFMA e of group u contracts
// k = 16 u + 4 g + e: reduction-contiguous operands are one ds_read_b128 per group); register ring of three K steps like the 64 x 64 kernel.
constexpr int kSmBM = 32, kSmBN = 64, kSmBK = 32, kSmLDK = kSmBK + 4, kSmLDN = kSmBN + 4, kSmPD = 3;
typedef float mlp_f32x4 __attribute__((ext_vector_type(4)));

template <bool BKM>
__global__ __launch_bounds__(kBlock) void mlp_gemm_small_kernel(MlpGemm g) {
  __shared__ __attribute__((aligned(16))) float As[2][kSmBM * kSmLDK];
  __shared__ __attribute__((aligned(16))) float Bs[2][BKM ? kSmBN * kSmLDK : kSmBK * kSmLDN];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, i = lane & 15, gq = lane >> 4;
  const int wr = wave >> 1, wc = wave & 1;
  const int64_t m0 = (int64_t)blockIdx.x * kSmBM;
  const int64_t n0 = (int64_t)blockIdx.y * kSmBN;
  // this thread's float4s of a K step: A (row t / 8, k 4 (t % 8)); B reduction-contiguous: rows t / 8 and t / 8 + 32, same k;
  // B outer-contiguous: reduction rows t / 16 and t / 16 + 16, columns 4 (t % 16)
  int64_t arow = m0 + (t >> 3);
  if (arow >= g.M) arow = g.M - 1;                       // (a row past the batch repeats the last one; nothing of it is stored)
  const float* pa = g.A.p + arow * g.A.ld + 4 * (t & 7);
  const float* pb0;
  const float* pb1;
  bool b_on = true;
  int64_t bstep;
  if (BKM) {
    int64_t r0 = n0 + (t >> 3), r1 = r0 + 32;
    if (r0 >= g.N) r0 = g.N - 1;
    if (r1 >= g.N) r1 = g.N - 1;
    pb0 = g.B.p + r0 * g.B.ld + 4 * (t & 7);
    pb1 = g.B.p + r1 * g.B.ld + 4 * (t & 7);
    bstep = kSmBK;
  } else {
    const int64_t col = n0 + 4 * (t & 15);
    b_on = col < g.N;                                     // (N % 4 == 0: a float4 is inside or outside)
    pb0 = g.B.p + (int64_t)(t >> 4) * g.B.ld + (b_on ? col : 0);
    pb1 = pb0 + 16 * g.B.ld;
    bstep = (int64_t)kSmBK * g.B.ld;
  }
  const int n_steps = g.K / kSmBK;
  // the register ring as NAMED slots (an indexed array behind a lambda stayed in a stack slot: 160 bytes of scratch per lane)
  float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0, a2 = a0, p0 = a0, p1 = a0, p2 = a0, q0 = a0, q1 = a0, q2 = a0;
#define RC_SM_LOAD(A_, P_, Q_, STEP_)                                                      \
  do {                                                                                     \
    A_ = *reinterpret_cast<const float4*>(pa + (int64_t)(STEP_) * kSmBK);                  \
    P_ = *reinterpret_cast<const float4*>(pb0 + (STEP_) * bstep);                          \
    Q_ = *reinterpret_cast<const float4*>(pb1 + (STEP_) * bstep);                          \
    if (!b_on) P_ = Q_ = make_float4(0.f, 0.f, 0.f, 0.f);                                  \
  } while (0)
#define RC_SM_STAGE(BUF_, A_, P_, Q_)                                                                          \
  do {                                                                                                         \
    *reinterpret_cast<float4*>(As[BUF_] + (t >> 3) * kSmLDK + 4 * (t & 7)) = A_;                               \
    if (BKM) {                                                                                                 \
      *reinterpret_cast<float4*>(Bs[BUF_] + (t >> 3) * kSmLDK + 4 * (t & 7)) = P_;                             \
      *reinterpret_cast<float4*>(Bs[BUF_] + ((t >> 3) + 32) * kSmLDK + 4 * (t & 7)) = Q_;                      \
    } else {                                                                                                   \
      *reinterpret_cast<float4*>(Bs[BUF_] + (t >> 4) * kSmLDN + 4 * (t & 15)) = P_;                            \
      *reinterpret_cast<float4*>(Bs[BUF_] + ((t >> 4) + 16) * kSmLDN + 4 * (t & 15)) = Q_;                     \
    }                                                                                                          \
  } while (0)
  if (0 < n_steps) RC_SM_LOAD(a0, p0, q0, 0);
  if (1 < n_steps) RC_SM_LOAD(a1, p1, q1, 1);
  if (2 < n_steps) RC_SM_LOAD(a2, p2, q2, 2);
  mlp_f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  RC_SM_STAGE(0, a0, p0, q0);
  __syncthreads();
  int buf = 0;
  // one K step: refill the slot that went to LDS before this step's barrier, multiply the staged tile, stage the next slot
#define RC_SM_STEP(S_, A_, P_, Q_, NA_, NP_, NQ_)                                                                        \
  if ((S_) < n_steps) {                                                                                                  \
    if ((S_) + kSmPD < n_steps) RC_SM_LOAD(A_, P_, Q_, (S_) + kSmPD);                                                    \
    const float* as = As[buf] + (16 * wr + i) * kSmLDK + 4 * gq;                                                         \
    float4 av[2], bv[2][2];                                                                                              \
    _Pragma("unroll") for (int u = 0; u < 2; ++u) {                                                                      \
      av[u] = *reinterpret_cast<const float4*>(as + 16 * u);                                                             \
      _Pragma("unroll") for (int cb = 0; cb < 2; ++cb) {                                                                 \
        if (BKM) {                                                                                                       \
          bv[u][cb] = *reinterpret_cast<const float4*>(Bs[buf] + (32 * wc + 16 * cb + i) * kSmLDK + 16 * u + 4 * gq);   \
        } else {                                                                                                         \
          const float* bp = Bs[buf] + (16 * u + 4 * gq) * kSmLDN + 32 * wc + 16 * cb + i;                                \
          bv[u][cb] = make_float4(bp[0], bp[kSmLDN], bp[2 * kSmLDN], bp[3 * kSmLDN]);                                   \
        }                                                                                                                \
      }                                                                                                                  \
    }                                                                                                                    \
    _Pragma("unroll") for (int u = 0; u < 2; ++u) {                                                                      \
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u].x, bv[u][0].x, acc0, 0, 0, 0);                                   \
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u].x, bv[u][1].x, acc1, 0, 0, 0);                                   \
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u].y, bv[u][0].y, acc0, 0, 0, 0);                                   \
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u].y, bv[u][1].y, acc1, 0, 0, 0);                                   \
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u].z, bv[u][0].z, acc0, 0, 0, 0);                                   \
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u].z, bv[u][1].z, acc1, 0, 0, 0);                                   \
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u].w, bv[u][0].w, acc0, 0, 0, 0);                                   \
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u].w, bv[u][1].w, acc1, 0, 0, 0);                                   \
    }                                                                                                                    \
    if ((S_) + 1 < n_steps) RC_SM_STAGE(buf ^ 1, NA_, NP_, NQ_);                                                         \
    __syncthreads();                                                                                                     \
    buf ^= 1;                                                                                                            \
  }
  for (int s0 = 0; s0 < n_steps; s0 += kSmPD) {
    RC_SM_STEP(s0, a0, p0, q0, a1, p1, q1)
    RC_SM_STEP(s0 + 1, a1, p1, q1, a2, p2, q2)
    RC_SM_STEP(s0 + 2, a2, p2, q2, a0, p0, q0)
  }
#undef RC_SM_STEP
#undef RC_SM_STAGE
#undef RC_SM_LOAD
  // epilogue: acc{cb}[r] is C(m0 + 16 wr + 4 gq + r, n0 + 32 wc + 16 cb + i)
  const uint64_t seed = g.seed ? *g.seed : 0;
  const int64_t mb = m0 + 16 * wr + 4 * gq;
#pragma unroll
  for (int cb = 0; cb < 2; ++cb) {
    const int64_t j = n0 + 32 * wc + 16 * cb + i;
    if (j >= g.N) continue;
    const float bj = g.bias ? g.bias[j] : 0.f;
    float keep[4] = {1.f, 1.f, 1.f, 1.f};
    if (g.seed && mb < g.M) mlp_keep4(g, seed, mb >> 2, (int)j, keep);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int64_t m = mb + r;
      if (m >= g.M) continue;
      float v = (cb == 0 ? acc0[r] : acc1[r]) + bj;
      if (g.relu) v = fmaxf(v, 0.f);
      if (g.seed) v *= keep[r];
      if (g.mask) v = g.mask[m * g.ldc + j] > 0.f ? v * g.mask_scale : 0.f;
      g.C[m * g.ldc + j] = v;
    }
  }
}

// ---- 128 x 128 tiles for the large products -------------------------------------------------------------------------------
// A 64 x 64 tile re-reads each operand element once per 64 columns of the other one and issues two ds_read_b32 per MFMA; at
// B = 131,072 the three products of a layer ran at 20-34 % of the matrix peak.  On 128 x 128 tiles a wave owns a 64 x 64 quarter as
// 2 x 2 MFMA blocks: every operand register feeds two MFMAs and every element fetched from L2 is used against 128 columns.  The
// workgroup -> tile map is XCD-aware: workgroup L runs on XCD L % 8, and the tiles of one XCD walk all column tiles of a row tile
// before the next row tile, so the A rows of a tile are fetched from HBM once and found in that XCD's L2 by the other column tiles.
// (The first cut of this kernel -- K step 16, operands read right before their MFMAs, 80 bytes of scratch per lane -- is gone:
//  mlp_gemm_big2_kernel below replaced it in round 4; shapes it does not take run on the 64 x 64 tiles.)
constexpr int kBigBM = 128, kBigBN = 128;

// ---- 128 x 128 tiles, second cut: K step 32, operand registers double-buffered against the MFMAs ---------------------------------
// Device timing of the kernel above at 131,072 x 512 x 512 (tools/gemm_probe.py with the pieces switched off one at a time):
// 824-900 us in all, 604 us with nothing but its LDS reads and MFMAs (72 % of the matrix peak), 409 us with everything BUT the
// MFMAs -- the two barely overlap: a K step is 32 LDS reads, a wait, 32 MFMAs, the staging writes, a barrier, and with two
// workgroups per CU the other one is usually in the same phase.  Here a K step is 32 wide (64 MFMAs per wave, half the barriers
// per flop) and cut into four groups of 16 MFMAs; the operands of group u + 1 are read from LDS while group u runs (two register
// sets), the next K step's tile is staged under group 2 and the first operands of the next buffer are read behind the barrier
// under group 3.  Operand layouts and the k = 8 u + 4 kh + e contraction order as in the 64 x 64 kernel (a reduction-minor
// operand gives four reduction indices per ds_read_b128).  Fast path only -- 16-byte aligned operands, K steps complete, a
// column count of B that is a multiple of four (the column of ones of the weight-gradient products included) -- everything else
// takes the kernel above.  Workgroup -> (tile, split) map: XCD-aware for many row tiles as above; for the weight-gradient
// products (few tiles, many splits of the batch) the tiles of ONE split run side by side on ONE XCD, so that the split's slices
// of dZ and X are fetched from HBM once and found in that XCD's L2 by the other tiles.
constexpr int kB2K = 32;
constexpr int kB2LDK = kB2K + 4;     // [outer][k] layout (reduction-minor operands)
constexpr int kB2LD = 132;           // [k][outer] layout (reduction-major operands)
constexpr int kB2Tile = 128 * kB2LDK > kB2K * kB2LD ? 128 * kB2LDK : kB2K * kB2LD;   // floats per buffer

struct B2Src {   // one float4 of the tile per thread and K step: its address and class (0 constant, 1 load)
  const float* p;
  int load;
  float4 cst;
};

template <bool KM>
__device__ __forceinline__ B2Src b2_src(const MlpOperand& o, int64_t outer0, int64_t outer_n, int64_t k0, int ones_col, int q) {
  const int t = threadIdx.x;
  B2Src f;
  f.cst = make_float4(0.f, 0.f, 0.f, 0.f);
  f.load = 1;
  if (KM) {   // row t / 8 + 32 q of the tile, k = 4 (t % 8) ..: rows past the matrix read its last row (never stored)
    int64_t i = outer0 + (t >> 3) + 32 * q;
    if (i >= outer_n) i = outer_n - 1;
    f.p = o.p + i * o.ld + k0 + 4 * (t & 7);
  } else {    // k row t / 32 + 8 q, columns 4 (t % 32) ..: columns past the matrix are constants
    const int64_t i = outer0 + 4 * (t & 31);
    f.p = o.p + (k0 + (t >> 5) + 8 * q) * o.ld + i;
    if (i >= outer_n) {
      f.load = 0;
      f.p = o.p;
      f.cst = make_float4(i == ones_col ? 1.f : 0.f, i + 1 == ones_col ? 1.f : 0.f, i + 2 == ones_col ? 1.f : 0.f,
                          i + 3 == ones_col ? 1.f : 0.f);
    }
  }
  return f;
}

template <bool KM>
__device__ __forceinline__ void b2_stage(float* tile, const float4& v, int q) {
  const int t = threadIdx.x;
  if (KM) *reinterpret_cast<float4*>(tile + ((t >> 3) + 32 * q) * kB2LDK + 4 * (t & 7)) = v;
  else *reinterpret_cast<float4*>(tile + ((t >> 5) + 8 * q) * kB2LD + 4 * (t & 31)) = v;
}

// the operand values of MFMAs (u, 0..3) for this lane: rows / columns o and o + 32 of the tile
template <bool KM>
__device__ __forceinline__ void b2_operands(const float* tile, int o, int kh, int u, float (&x)[2][4]) {
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    if (KM) {
      const float4 v = *reinterpret_cast<const float4*>(tile + (o + 32 * h) * kB2LDK + 8 * u + 4 * kh);
      x[h][0] = v.x; x[h][1] = v.y; x[h][2] = v.z; x[h][3] = v.w;
    } else {
      const float* p = tile + (8 * u + 4 * kh) * kB2LD + o + 32 * h;
      x[h][0] = p[0]; x[h][1] = p[kB2LD]; x[h][2] = p[2 * kB2LD]; x[h][3] = p[3 * kB2LD];
    }
  }
}

template <bool AKM, bool BKM>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(2, 2))) void mlp_gemm_big2_kernel(MlpGemm g, int m_tiles, int n_tiles, int splits) {
  extern __shared__ __attribute__((aligned(16))) float b2_lds[];   // As[2] | Bs[2]
  float* As = b2_lds;
  float* Bs = b2_lds + 2 * kB2Tile;
  // ---- workgroup -> (row tile, column tile, split)
  const int L = blockIdx.x, xcd = L & 7, slot = L >> 3;
  int mt, nt, sp;
  if (splits > 1) {            // the tiles of a split side by side on one XCD
    const int tiles = m_tiles * n_tiles;
    sp = (slot / tiles) * 8 + xcd;
    const int tile = slot % tiles;
    mt = tile / n_tiles;
    nt = tile % n_tiles;
    if (sp >= splits) return;
  } else if (m_tiles >= 16) {  // a row tile's column tiles side by side on one XCD
    sp = 0;
    mt = (slot / n_tiles) * 8 + xcd;
    nt = slot % n_tiles;
    if (mt >= m_tiles) return;
  } else {
    sp = 0;
    mt = L / n_tiles;
    nt = L % n_tiles;
    if (mt >= m_tiles) return;
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wr = wave >> 1, wc = wave & 1;   // this wave's 64 x 64 quarter
  const int64_t m0 = (int64_t)mt * 128, n0 = (int64_t)nt * 128;
  const int64_t kb = (int64_t)sp * g.split_stride_k;
  const int64_t ke = (kb + g.K < g.k_total) ? kb + g.K : g.k_total;
  const int64_t bn = g.ones_col >= 0 ? (int64_t)g.ones_col : (int64_t)g.N;          // columns of B in memory
  const int64_t ncols = g.ones_col >= 0 ? (int64_t)g.ones_col + 1 : (int64_t)g.N;   // columns of the product
  // 32-column blocks of this wave that hold a column of the product (the tile of the column of ones: one block of one wave)
  const int nb_live = n0 + wc * 64 + 32 < ncols ? 2 : (n0 + wc * 64 < ncols ? 1 : 0);
  mlp_f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  // this thread's four float4 per operand and K step: address, class (load / constant), the constant
  const float* pa[4];
  const float* pb[4];
  bool lb[4];
  float4 cb[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const B2Src sa = b2_src<AKM>(g.A, m0, g.M, kb, -1, q);
    const B2Src sb = b2_src<BKM>(g.B, n0, bn, kb, g.ones_col, q);
    pa[q] = sa.p;   // (A: rows past the matrix read its last row, always a load)
    pb[q] = sb.p; lb[q] = sb.load != 0; cb[q] = sb.cst;
  }
  const int64_t stride_a = AKM ? (int64_t)kB2K : (int64_t)kB2K * g.A.ld;
  const int64_t stride_b = BKM ? (int64_t)kB2K : (int64_t)kB2K * g.B.ld;
  float4 ra[4], rb[4];
#define B2_FETCH(OA, OB)                                                              \
  _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                     \
    ra[q] = *reinterpret_cast<const float4*>(pa[q] + (OA));                           \
    rb[q] = *reinterpret_cast<const float4*>(pb[q] + (lb[q] ? (OB) : 0));             \
  }
#define B2_STAGE(BUF)                                                                 \
  _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                     \
    /* (component selects: a select between two array elements makes both arrays stack objects) */ \
    b2_stage<AKM>(As + (BUF) * kB2Tile, make_float4(ra[q].x, ra[q].y, ra[q].z, ra[q].w), q); \
    b2_stage<BKM>(Bs + (BUF) * kB2Tile, make_float4(lb[q] ? rb[q].x : cb[q].x, lb[q] ? rb[q].y : cb[q].y, \
                                                    lb[q] ? rb[q].z : cb[q].z, lb[q] ? rb[q].w : cb[q].w), q); \
  }
  int64_t off_a = 0, off_b = 0;
  B2_FETCH(off_a, off_b)
  B2_STAGE(0)
  __syncthreads();
  const int ai = wr * 64 + (lane & 31), bj = wc * 64 + (lane & 31), kh = lane >> 5;
  float xa[2][2][4], xb[2][2][4];   // [register set][32-row / 32-column block][e]
  b2_operands<AKM>(As, ai, kh, 0, xa[0]);
  b2_operands<BKM>(Bs, bj, kh, 0, xb[0]);
  int buf = 0;
  for (int64_t k0 = kb; k0 < ke; k0 += kB2K) {
    const bool more = k0 + kB2K < ke;   // workgroup-uniform
    if (more) {
      off_a += stride_a;
      off_b += stride_b;
      B2_FETCH(off_a, off_b)
    }
    const float* at = As + buf * kB2Tile;
    const float* bt = Bs + buf * kB2Tile;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int cur = u & 1, nxt = cur ^ 1;
      if (u < 3) {
        b2_operands<AKM>(at, ai, kh, u + 1, xa[nxt]);
        b2_operands<BKM>(bt, bj, kh, u + 1, xb[nxt]);
      }
      __builtin_amdgcn_sched_barrier(0);
#define B2_MFMAS(E0, E1)                                                                                          \
  if (nb_live == 2) {                                                                                             \
    _Pragma("unroll") for (int e = (E0); e < (E1); ++e) {                                                         \
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[cur][0][e], xb[cur][0][e], acc[0][0], 0, 0, 0);         \
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[cur][0][e], xb[cur][1][e], acc[0][1], 0, 0, 0);         \
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[cur][1][e], xb[cur][0][e], acc[1][0], 0, 0, 0);         \
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[cur][1][e], xb[cur][1][e], acc[1][1], 0, 0, 0);         \
    }                                                                                                             \
  } else if (nb_live == 1) {                                                                                      \
    _Pragma("unroll") for (int e = (E0); e < (E1); ++e) {                                                         \
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[cur][0][e], xb[cur][0][e], acc[0][0], 0, 0, 0);         \
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[cur][1][e], xb[cur][0][e], acc[1][0], 0, 0, 0);         \
    }                                                                                                             \
  }
      if (u < 3) {
        B2_MFMAS(0, 4)
        __builtin_amdgcn_sched_barrier(0);
        if (u == 2 && more) { B2_STAGE(buf ^ 1) }
      } else {
        // the barrier and the first operands of the next buffer go between the two halves of the last group: the matrix pipe
        // still holds eight MFMAs while the waves meet and the reads travel
        B2_MFMAS(0, 2)
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        if (more) {   // (u == 3: cur = 1, the set the next K step starts from is 0)
          b2_operands<AKM>(As + (buf ^ 1) * kB2Tile, ai, kh, 0, xa[0]);
          b2_operands<BKM>(Bs + (buf ^ 1) * kB2Tile, bj, kh, 0, xb[0]);
        }
        __builtin_amdgcn_sched_barrier(0);
        B2_MFMAS(2, 4)
      }
#undef B2_MFMAS
    }
    buf ^= 1;
  }
#undef B2_FETCH
#undef B2_STAGE
  // ---- epilogue: acc[a][b][r] is C(m0 + wr*64 + a*32 + (r & 3) + 8 (r >> 2) + 4 kh, n0 + wc*64 + b*32 + (lane & 31))
  const uint64_t seed = g.seed ? *g.seed : 0;
  float* C = g.C + (size_t)sp * (size_t)g.M * (size_t)g.ldc;
  if (m0 + 128 <= g.M && n0 + 128 <= ncols && g.seed == nullptr) {   // a tile inside the product, no dropout: no range checks
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int64_t j = n0 + wc * 64 + b * 32 + (lane & 31);
      const float bj_ = (g.bias != nullptr && j < g.N) ? g.bias[j] : 0.f;
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        float* row = C + (m0 + wr * 64 + a * 32 + 4 * kh) * g.ldc + j;
        const float* mrow = g.mask ? g.mask + (m0 + wr * 64 + a * 32 + 4 * kh) * g.ldc + j : nullptr;
        float mk[16];
        if (g.mask) {
#pragma unroll
          for (int r = 0; r < 16; ++r) mk[r] = mrow[((r & 3) + 8 * (r >> 2)) * g.ldc];
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float v = acc[a][b][r] + bj_;
          if (g.relu) v = fmaxf(v, 0.f);
          if (g.mask) v = mk[r] > 0.f ? v * g.mask_scale : 0.f;
          row[((r & 3) + 8 * (r >> 2)) * g.ldc] = v;
        }
      }
    }
    return;
  }
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    const int64_t j = n0 + wc * 64 + b * 32 + (lane & 31);
    if (j >= ncols) continue;
    const float bj_ = (g.bias != nullptr && j < g.N) ? g.bias[j] : 0.f;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int64_t mb = m0 + wr * 64 + a * 32 + 8 * q + 4 * kh;
        float keep[4] = {1.f, 1.f, 1.f, 1.f};
        if (g.seed && mb < g.M) mlp_keep4(g, seed, mb >> 2, (int)j, keep);
        float mk[4] = {1.f, 1.f, 1.f, 1.f};
        if (g.mask) {
#pragma unroll
          for (int e = 0; e < 4; ++e) mk[e] = (mb + e < g.M && g.mask[(mb + e) * g.ldc + j] > 0.f) ? g.mask_scale : 0.f;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int64_t m = mb + e;
          if (m >= g.M) continue;
          float v = acc[a][b][4 * q + e] + bj_;
          if (g.relu) v = fmaxf(v, 0.f);
          if (g.seed) v *= keep[e];
          if (g.mask) v *= mk[e];
          C[m * g.ldc + j] = v;
        }
      }
  }
}

// dZ = dY * (Y > 0 ? scale : 0): the saved output of drop(relu(.)) is its own mask
__global__ __launch_bounds__(kBlock) void mlp_mask_kernel(const float* __restrict__ dY, const float* __restrict__ Y, int64_t n,
                                                          float scale, float* __restrict__ dZ) {
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) dZ[i] = Y[i] > 0.f ? dY[i] * scale : 0.f;
}

// out[i] = sum_s part[s][i] in split order; rows of width ld_in = K + 1 -> dW [N, K] and db [N]
__global__ __launch_bounds__(kBlock) void mlp_reduce_kernel(const float* __restrict__ part, int splits, int N, int K,
                                                            float* __restrict__ dW, float* __restrict__ db) {
  const int64_t total = (int64_t)N * (K + 1);
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (int64_t)gridDim.x * kBlock) {
    float s = 0.f;
    for (int q = 0; q < splits; ++q) s += part[(size_t)q * total + i];
    const int n = (int)(i / (K + 1)), k = (int)(i % (K + 1));
    if (k < K) dW[(size_t)n * K + k] = s;
    else if (db != nullptr) db[n] = s;
  }
}

// ---- weight gradient of a layer with one to four outputs (the last Linear of every tower: width -> 1) ------------------------------
// [dW | db][n, k] = sum_m dZ[m, n] [X | 1][m, k] for N <= 4 is a column reduction of X weighted by dZ, not a matrix product: a
// 64 x 64 tile spent 147 us on the 131,072 x 64 -> 1 layer with 63 of its 64 rows empty.  Here a lane-group of K / 4 lanes owns a
// row of X per trip (float4 per lane, four rows in flight), the groups of a workgroup are summed in a fixed tree, and the
// partial sums of the workgroups go through mlp_reduce_kernel like the splits of the tiled product (fixed order, no atomics).
constexpr int kNarrowWgs = 512;

template <int NOUT>
__global__ __launch_bounds__(kBlock) void mlp_dw_narrow_kernel(const float* __restrict__ dZ, const float* __restrict__ X, int64_t M,
                                                               int K, float* __restrict__ part /* [gridDim.x][NOUT][K + 1] */) {
  const int lpr = K / 4;                 // lanes per row
  const int gpb = kBlock / lpr;          // rows per trip of the workgroup
  const int l = threadIdx.x % lpr, grp = threadIdx.x / lpr;
  float4 acc[NOUT];
  float accb[NOUT];
#pragma unroll
  for (int n = 0; n < NOUT; ++n) {
    acc[n] = make_float4(0.f, 0.f, 0.f, 0.f);
    accb[n] = 0.f;
  }
  const bool live = grp < gpb;           // (K / 4 need not divide 256)
  constexpr int U = 4;
  for (int64_t m0 = (int64_t)blockIdx.x * gpb + grp; live && m0 < M; m0 += (int64_t)U * gridDim.x * gpb) {
    float4 x[U];
    float z[U][NOUT];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t m = m0 + (int64_t)u * gridDim.x * gpb;
      const int64_t mc = m < M ? m : M - 1;
      x[u] = reinterpret_cast<const float4*>(X + mc * K)[l];
#pragma unroll
      for (int n = 0; n < NOUT; ++n) z[u][n] = m < M ? dZ[mc * NOUT + n] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int n = 0; n < NOUT; ++n) {
        acc[n].x = fmaf(z[u][n], x[u].x, acc[n].x);
        acc[n].y = fmaf(z[u][n], x[u].y, acc[n].y);
        acc[n].z = fmaf(z[u][n], x[u].z, acc[n].z);
        acc[n].w = fmaf(z[u][n], x[u].w, acc[n].w);
        accb[n] += z[u][n];
      }
  }
  // sum over the lane-groups of the workgroup, fixed order, through LDS (one column quad at a time: K / 4 passes of the small buffer)
  extern __shared__ float nr_lds[];      // [gpb][NOUT][K + 1]
  float* mine = nr_lds + (size_t)grp * NOUT * (K + 1);
  if (live) {
#pragma unroll
    for (int n = 0; n < NOUT; ++n) {
      mine[n * (K + 1) + 4 * l + 0] = acc[n].x;
      mine[n * (K + 1) + 4 * l + 1] = acc[n].y;
      mine[n * (K + 1) + 4 * l + 2] = acc[n].z;
      mine[n * (K + 1) + 4 * l + 3] = acc[n].w;
      if (l == 0) mine[n * (K + 1) + K] = accb[n];
    }
  }
  __syncthreads();
  const int total = NOUT * (K + 1);
  for (int i = threadIdx.x; i < total; i += kBlock) {
    float t = 0.f;
    for (int q = 0; q < gpb; ++q) t += nr_lds[(size_t)q * total + i];
    part[(size_t)blockIdx.x * total + i] = t;
  }
}

// mlp_reduce_kernel for many planes of few elements (the narrow route: up to 512 workgroup partials of N (K + 1) <= 4,100 sums):
// one wave per element, the lanes take planes q = lane, lane + 64, ..., the lane sums meet in a fixed xor tree
__global__ __launch_bounds__(kBlock) void mlp_reduce_wave_kernel(const float* __restrict__ part, int splits, int N, int K,
                                                                 float* __restrict__ dW, float* __restrict__ db) {
  const int64_t total = (int64_t)N * (K + 1);
  const int lane = threadIdx.x & 63;
  const int64_t i = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
  if (i >= total) return;   // wave-uniform
  float s = 0.f;
  for (int q = lane; q < splits; q += 64) s += part[(size_t)q * total + i];
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
  if (lane == 0) {
    const int n = (int)(i / (K + 1)), k = (int)(i % (K + 1));
    if (k < K) dW[(size_t)n * K + k] = s;
    else if (db != nullptr) db[n] = s;
  }
}

static bool vec4_ok(const MlpOperand& o) { return reinterpret_cast<uintptr_t>(o.p) % 16 == 0 && o.ld % 4 == 0; }

// RC_MLP_BIG=0: 64 x 64 tiles for every product (round 2), for A/B timing
static bool mlp_big_enabled() {
  const char* v = getenv("RC_MLP_BIG");
  return !(v && v[0] == '0');
}

// RC_MLP_BIG2=0: the first 128 x 128 kernel for every large product (A/B timing)
static bool mlp_big2_enabled() {
  const char* v = getenv("RC_MLP_BIG2");
  return !(v && v[0] == '0');
}

// dynamic LDS beyond 64 KB needs the attribute, once per (device, kernel variant)
static int b2_set_lds_limit(const void* kern, int variant, size_t lds) {
  static std::mutex mu;
  static bool done[64][4] = {};
  int dev = 0;
  RC_HIP(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lock(mu);
  if (dev >= 0 && dev < 64 && !done[dev][variant]) {
    RC_HIP(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    done[dev][variant] = true;
  }
  return RC_OK;
}

static int mlp_launch(const MlpGemm& g, int splits, hipStream_t s, bool want_big2 = false) {
  const int64_t ncols = g.ones_col >= 0 ? (int64_t)g.ones_col + 1 : (int64_t)g.N;
  // 128 x 128 tiles once they fill the chip (>= 2048 of them, 8 per CU: below that the 64 x 64 tiles' finer grain wins -- B = 16,384:
  // 1.05 against 1.14 ms per step) and the product is at least a tile wide
  const int64_t mt = (g.M + kBigBM - 1) / kBigBM, nt = (ncols + kBigBN - 1) / kBigBN;
  if (mlp_big_enabled() && g.M >= kBigBM && ncols >= kBigBN && (mt * nt * splits >= 2048 || want_big2) && mt < (1 << 24)) {
    const int64_t bn = g.ones_col >= 0 ? (int64_t)g.ones_col : (int64_t)g.N;
    const bool steps_complete = g.K % kB2K == 0 && (splits == 1 ? g.k_total == g.K : (g.split_stride_k == g.K && g.k_total % kB2K == 0));
    if (mlp_big2_enabled() && vec4_ok(g.A) && vec4_ok(g.B) && steps_complete && (g.B.k_major || bn % 4 == 0) && g.K >= kB2K &&
        mt * nt * (int64_t)((splits + 7) / 8 * 8) < (1 << 28)) {
      const size_t lds = 4 * (size_t)kB2Tile * sizeof(float);
      void (*kern)(MlpGemm, int, int, int) =
          g.A.k_major ? (g.B.k_major ? mlp_gemm_big2_kernel<true, true> : mlp_gemm_big2_kernel<true, false>)
                      : (g.B.k_major ? mlp_gemm_big2_kernel<false, true> : mlp_gemm_big2_kernel<false, false>);
      RC_TRY(b2_set_lds_limit(reinterpret_cast<const void*>(kern), (g.A.k_major ? 2 : 0) + (g.B.k_major ? 1 : 0), lds));
      int64_t blocks;
      if (splits > 1) blocks = (int64_t)((splits + 7) / 8) * 8 * mt * nt;
      else blocks = mt >= 16 ? ((mt + 7) / 8) * nt * 8 : mt * nt;
      hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(kBlock), lds, s, g, (int)mt, (int)nt, splits);
      RC_LAUNCH_CHECK();
      return RC_OK;
    }
  }
  dim3 grid((unsigned)((g.M + kMlpBM - 1) / kMlpBM), (unsigned)((ncols + kMlpBN - 1) / kMlpBN), (unsigned)splits);
  void (*kern)(MlpGemm, int, int) = g.A.k_major ? (g.B.k_major ? mlp_gemm_kernel<true, true> : mlp_gemm_kernel<true, false>)
                                                : (g.B.k_major ? mlp_gemm_kernel<false, true> : mlp_gemm_kernel<false, false>);
  hipLaunchKernelGGL(kern, grid, dim3(kBlock), 0, s, g, vec4_ok(g.A) ? 1 : 0, vec4_ok(g.B) ? 1 : 0);
  RC_LAUNCH_CHECK();
  return RC_OK;
}

// The weight-gradient product of a large batch on the 128 x 128 tiles (mlp_gemm_big2_kernel, the tiles of a split side by side
// on one XCD): at most two workgroups per CU -- one round --, a multiple of eight splits.  0 = stays on the 64 x 64 tiles.
static int mlp_dw_big_splits(int64_t M, int N, int K) {
  if (!mlp_big_enabled() || !mlp_big2_enabled() || M < 16384 || M % kB2K != 0 || N < 128 || K < 127 || K % 4 != 0 || N % 4 != 0) return 0;
  const int64_t tiles = ((int64_t)(N + 127) / 128) * ((K + 1 + 127) / 128);
  int64_t s = 512 / tiles / 8 * 8;   // one round of two workgroups per CU, a multiple of eight splits (one XCD each)
  const int64_t max_s = M / 1024;   // at least 32 K steps per split
  if (s > max_s) s = max_s / 8 * 8;
  if (s > 64) s = 64;
  return s >= 8 ? (int)s : 0;
}

static int mlp_splits(int64_t M, int N, int K) {
  // (the weight-gradient products stay on 64 x 64 tiles: [N, K + 1] is a few dozen tiles however large the batch, and at
  //  20 tiles x 39 splits the 128 x 128 kernel measured 1.87 ms against ~1.5 ms at B = 131,072)
  const int64_t tiles = ((int64_t)(N + kMlpBM - 1) / kMlpBM) * ((K + 1 + kMlpBN - 1) / kMlpBN);
  int64_t s = (1024 + tiles - 1) / tiles;              // ~4 workgroups per CU
  const int64_t max_s = (M + 127) / 128;               // at least 128 batch rows (four K steps) per split
  if (s > max_s) s = max_s;
  if (s < 1) s = 1;
  if (s > 64) s = 64;
  return (int)s;
}

// Split-K for the forward / dX products of a small batch: a 1,024 x 512 x 512 product is 128 tiles of 64 x 64 -- half the CUs
// idle, each workgroup alone on its CU walking 16 dependent K steps.  Cut into `s` reduction ranges it is s x 128 workgroups of
// 16 / s steps; the partial products [s][M][N] are summed in split order (no float atomics) by a second launch that also applies
// bias, ReLU and dropout.  1 = no split (enough tiles, or a short reduction).
static int mlp_k_splits(int64_t M, int N, int K) {
  static const int mode = [] {   // RC_MLP_SPLITK=0: no split-K of the forward / dX products (A/B)
    const char* v = getenv("RC_MLP_SPLITK");
    return (v && v[0] == '0') ? 0 : 1;
  }();
  if (mode == 0) return 1;
  const int64_t tiles = ((M + kMlpBM - 1) / kMlpBM) * ((N + kMlpBN - 1) / kMlpBN);
  if (tiles >= 256 || K < 128) return 1;
  int64_t s = (512 + tiles - 1) / tiles;
  if (s > K / 64) s = K / 64;
  if (s > 8) s = 8;
  return (int)(s < 1 ? 1 : s);
}

// Y[m, n] = drop(relu(sum_s part[s][m][n] + b[n])): thread (m4, n) owns rows 4 m4 .. 4 m4 + 3 of column n (the dropout mask's
// four words of one Philox call, as in the GEMM epilogue)
__global__ __launch_bounds__(kBlock) void mlp_split_epilogue_kernel(MlpGemm g, const float* __restrict__ part, int splits, float* __restrict__ Y) {
  const int64_t m4n = (g.M + 3) / 4;
  const int64_t total = m4n * g.N;
  const uint64_t seed = g.seed ? *g.seed : 0;
  const size_t plane = (size_t)g.M * (size_t)g.N;
  for (int64_t idx = (int64_t)blockIdx.x * kBlock + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * kBlock) {
    const int64_t m4 = idx / g.N;
    const int n = (int)(idx % g.N);
    const float bn = g.bias ? g.bias[n] : 0.f;
    float keep[4] = {1.f, 1.f, 1.f, 1.f};
    if (g.seed) mlp_keep4(g, seed, m4, n, keep);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int64_t m = 4 * m4 + e;
      if (m >= g.M) continue;
      float v = 0.f;
      for (int q = 0; q < splits; ++q) v += part[(size_t)q * plane + (size_t)m * g.N + n];
      v += bn;
      if (g.relu) v = fmaxf(v, 0.f);
      if (g.seed) v *= keep[e];
      if (g.mask) v = g.mask[(size_t)m * g.N + n] > 0.f ? v * g.mask_scale : 0.f;   // (ldc = N for every product that takes this path)
      Y[(size_t)m * g.N + n] = v;
    }
  }
}

// one product C = A . B with its epilogue in `g`, through split-K when `part` (room for mlp_k_splits planes) is given
// RC_MLP_SMALL=0: the 64 x 64 tiles (+ split-K) for every small product (A/B)
static bool mlp_small_ok(const MlpGemm& g) {
  static const int mode = [] {
    const char* v = getenv("RC_MLP_SMALL");
    return (v && v[0] == '0') ? 0 : 1;
  }();
  if (mode == 0 || g.ones_col >= 0 || !g.A.k_major || g.k_total != g.K || g.K % kSmBK != 0 || g.K < 2 * kSmBK) return false;
  if (!vec4_ok(g.A) || !vec4_ok(g.B) || (!g.B.k_major && g.N % 4 != 0)) return false;
  const int64_t tiles64 = ((g.M + kMlpBM - 1) / kMlpBM) * ((g.N + kMlpBN - 1) / kMlpBN);
  return tiles64 < 512 && g.N >= 16;      // (beyond that the 64 x 64 tiles fill the chip by themselves)
}

static int mlp_product(MlpGemm g, float* part, hipStream_t s) {
  if (mlp_small_ok(g)) {
    dim3 grid((unsigned)((g.M + kSmBM - 1) / kSmBM), (unsigned)((g.N + kSmBN - 1) / kSmBN), 1);
    if (g.B.k_major) hipLaunchKernelGGL(mlp_gemm_small_kernel<true>, grid, dim3(kBlock), 0, s, g);
    else hipLaunchKernelGGL(mlp_gemm_small_kernel<false>, grid, dim3(kBlock), 0, s, g);
    RC_LAUNCH_CHECK();
    return RC_OK;
  }
  const int splits = part ? mlp_k_splits(g.M, g.N, g.K) : 1;
  if (splits <= 1) return mlp_launch(g, 1, s);
  MlpGemm p = g;   // the partial products: no epilogue
  int64_t per = (g.K + splits - 1) / splits;
  per = (per + kMlpBK - 1) / kMlpBK * kMlpBK;
  const int used = (int)((g.K + per - 1) / per);
  p.C = part; p.ldc = g.N; p.K = (int)per; p.split_stride_k = per; p.k_total = g.K;
  p.bias = nullptr; p.relu = 0; p.seed = nullptr; p.mask = nullptr;
  RC_TRY(mlp_launch(p, used, s));
  const int64_t total = ((g.M + 3) / 4) * g.N;
  int64_t blocks = (total + kBlock - 1) / kBlock;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(mlp_split_epilogue_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, s, g, part, used, g.C);
  RC_LAUNCH_CHECK();
  return RC_OK;
}
// the weighted column sum for one to four outputs (mlp_dw_narrow_kernel): K a multiple of 4 up to 1,024 whose lane-groups fit the
// workgroup's LDS buffer, a batch large enough to fill the workgroups
static bool mlp_dw_narrow_ok(int64_t M, int N, int K) {
  if (N < 1 || N > 4 || K % 4 != 0 || K < 4 || K > 1024 || M < 4096) return false;
  const int gpb = kBlock / (K / 4);
  return gpb >= 1 && (size_t)gpb * N * (K + 1) * sizeof(float) <= 48 * 1024;
}

// planes of partial weight gradients the workspace holds (any route)
static int mlp_dw_parts(int64_t M, int N, int K) {
  const int a = mlp_splits(M, N, K), b = mlp_dw_big_splits(M, N, K), c = mlp_dw_narrow_ok(M, N, K) ? kNarrowWgs : 0;
  return a > b ? (a > c ? a : c) : (b > c ? b : c);
}
static size_t mlp_split_bytes(int64_t M, int N, int K) {
  const int sp = mlp_k_splits(M, N, K);
  return sp > 1 ? align_up((size_t)sp * (size_t)M * (size_t)N * sizeof(float), 256) : 0;
}

}  // namespace rc

using namespace rc;

extern "C" size_t rc_linear_fwd_workspace_bytes(int64_t M, int N, int K) {
  if (M < 1 || N < 1 || K < 1) return 0;
  return mlp_split_bytes(M, N, K) + 256;
}

extern "C" size_t rc_linear_bwd_workspace_bytes(int64_t M, int N, int K) {
  if (M < 1 || N < 1 || K < 1) return 0;
  const size_t dz = align_up((size_t)M * N * sizeof(float), 256);
  const size_t part = align_up((size_t)mlp_dw_parts(M, N, K) * (size_t)N * (size_t)(K + 1) * sizeof(float), 256);
  return dz + part + mlp_split_bytes(M, K, N);   // (the dX product: [M, K] out, reduction over N)
}

static int linear_fwd_impl(const float* X, const float* W, const float* b, int64_t M, int N, int K, int relu, float drop_p,
                           const uint64_t* seed_dev, uint32_t site, float* Y, void* ws, size_t ws_bytes, rc_stream_t stream);

extern "C" int rc_linear_fwd(const float* X, const float* W, const float* b, int64_t M, int N, int K, int relu, float drop_p,
                             const uint64_t* seed_dev, uint32_t site, float* Y, rc_stream_t stream) {
  return linear_fwd_impl(X, W, b, M, N, K, relu, drop_p, seed_dev, site, Y, nullptr, 0, stream);
}

// the same with a workspace (rc_linear_fwd_workspace_bytes): small batches are cut along the reduction (split-K)
extern "C" int rc_linear_fwd_ws(const float* X, const float* W, const float* b, int64_t M, int N, int K, int relu, float drop_p,
                                const uint64_t* seed_dev, uint32_t site, float* Y, void* ws, size_t ws_bytes, rc_stream_t stream) {
  RC_REQUIRE(ws == nullptr || ws_bytes >= rc_linear_fwd_workspace_bytes(M, N, K), "rc_linear_fwd_ws: workspace %zu < %zu", ws_bytes,
             rc_linear_fwd_workspace_bytes(M, N, K));
  return linear_fwd_impl(X, W, b, M, N, K, relu, drop_p, seed_dev, site, Y, ws, ws_bytes, stream);
}

static int linear_fwd_impl(const float* X, const float* W, const float* b, int64_t M, int N, int K, int relu, float drop_p,
                           const uint64_t* seed_dev, uint32_t site, float* Y, void* ws, size_t ws_bytes, rc_stream_t stream) {
  (void)ws_bytes;
  if (M == 0) return RC_OK;
  RC_REQUIRE(X && W && Y, "rc_linear_fwd: null pointer");
  RC_REQUIRE(M > 0 && N >= 1 && K >= 1 && N < 65536, "rc_linear_fwd: bad shape M=%lld N=%d K=%d", (long long)M, N, K);
  RC_REQUIRE(drop_p >= 0.f && drop_p < 1.f && (drop_p == 0.f || seed_dev), "rc_linear_fwd: dropout p=%g needs p in [0, 1) and a device seed", (double)drop_p);
  MlpGemm g;
  memset(&g, 0, sizeof(g));
  g.A = MlpOperand{X, K, 1};
  g.B = MlpOperand{W, K, 1};
  g.M = M; g.N = N; g.K = K; g.ones_col = -1; g.C = Y; g.ldc = N; g.split_stride_k = K; g.k_total = K;
  g.bias = b; g.relu = relu ? 1 : 0; g.site = site;
  if (drop_p > 0.f) {
    g.seed = seed_dev;
    g.drop_thresh = (uint32_t)((double)drop_p * 4294967296.0);
    g.keep_scale = 1.0f / (1.0f - drop_p);
  }
  return mlp_product(g, static_cast<float*>(ws), as_stream(stream));
}

static int linear_bwd_impl(const float* X, const float* W, const float* Y, const float* dY, int64_t M, int N, int K, float drop_p,
                           int x_act, float x_drop_p, float* dX, float* dW, float* db, void* ws, size_t ws_bytes, rc_stream_t stream);

extern "C" int rc_linear_bwd(const float* X, const float* W, const float* Y, const float* dY, int64_t M, int N, int K,
                             float drop_p, float* dX, float* dW, float* db, void* ws, size_t ws_bytes, rc_stream_t stream) {
  return linear_bwd_impl(X, W, Y, dY, M, N, K, drop_p, 0, 0.f, dX, dW, db, ws, ws_bytes, stream);
}

// rc_linear_bwd inside a chain of layers: with x_act != 0 the input X of this layer is the drop(relu(.)) output of the layer
// below, and dX comes out already multiplied by that layer's mask (X > 0 ? 1 / (1 - x_drop_p) : 0) in the product's epilogue --
// the layer below is then called with Y = NULL (its dY is its dZ): one pass over [M, K] less per layer boundary.
extern "C" int rc_linear_bwd_chain(const float* X, const float* W, const float* Y, const float* dY, int64_t M, int N, int K,
                                   float drop_p, int x_act, float x_drop_p, float* dX, float* dW, float* db, void* ws,
                                   size_t ws_bytes, rc_stream_t stream) {
  RC_REQUIRE(x_drop_p >= 0.f && x_drop_p < 1.f, "rc_linear_bwd_chain: dropout p=%g of the layer below outside [0, 1)", (double)x_drop_p);
  return linear_bwd_impl(X, W, Y, dY, M, N, K, drop_p, x_act, x_drop_p, dX, dW, db, ws, ws_bytes, stream);
}

static int linear_bwd_impl(const float* X, const float* W, const float* Y, const float* dY, int64_t M, int N, int K, float drop_p,
                           int x_act, float x_drop_p, float* dX, float* dW, float* db, void* ws, size_t ws_bytes, rc_stream_t stream) {
  RC_REQUIRE(X && W && dY && (dW || dX), "rc_linear_bwd: null pointer");
  RC_REQUIRE(M >= 0 && N >= 1 && K >= 1, "rc_linear_bwd: bad shape M=%lld N=%d K=%d", (long long)M, N, K);
  hipStream_t s = as_stream(stream);
  if (M == 0) {
    if (dW) RC_HIP(hipMemsetAsync(dW, 0, (size_t)N * K * sizeof(float), s));
    if (db) RC_HIP(hipMemsetAsync(db, 0, (size_t)N * sizeof(float), s));
    return RC_OK;
  }
  RC_REQUIRE(ws != nullptr && ws_bytes >= rc_linear_bwd_workspace_bytes(M, N, K), "rc_linear_bwd: workspace %zu < %zu", ws_bytes,
             rc_linear_bwd_workspace_bytes(M, N, K));
  RC_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "rc_linear_bwd: dropout p=%g outside [0, 1)", (double)drop_p);
  float* dz_buf = static_cast<float*>(ws);
  float* part = reinterpret_cast<float*>(static_cast<char*>(ws) + align_up((size_t)M * N * sizeof(float), 256));
  const float* dZ = dY;
  if (Y != nullptr) {  // through drop(relu(.)): the saved output is the mask
    const int64_t n = M * (int64_t)N;
    int64_t blocks = (n + kBlock - 1) / kBlock;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(mlp_mask_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, s, dY, Y, n, 1.0f / (1.0f - drop_p), dz_buf);
    RC_LAUNCH_CHECK();
    dZ = dz_buf;
  }
  if (dX != nullptr) {  // dX[m, k] = sum_n dZ[m, n] W[n, k]
    MlpGemm g;
    memset(&g, 0, sizeof(g));
    g.A = MlpOperand{dZ, N, 1};
    g.B = MlpOperand{W, K, 0};
    g.M = M; g.N = K; g.K = N; g.ones_col = -1; g.C = dX; g.ldc = K; g.split_stride_k = N; g.k_total = N;
    if (x_act) {
      g.mask = X;
      g.mask_scale = 1.0f / (1.0f - x_drop_p);
    }
    float* dx_part = mlp_k_splits(M, K, N) > 1 ? reinterpret_cast<float*>(reinterpret_cast<char*>(part) + align_up((size_t)mlp_dw_parts(M, N, K) * (size_t)N * (size_t)(K + 1) * sizeof(float), 256)) : nullptr;
    RC_TRY(mlp_product(g, dx_part, s));
  }
  if (dW == nullptr) return RC_OK;   // (rc_linear_bwd_chain: the caller forms the weight gradient in another call, e.g. on another stream)
  if (mlp_dw_narrow_ok(M, N, K) && reinterpret_cast<uintptr_t>(X) % 16 == 0) {   // one to four outputs: a weighted column sum
    const int gpb = kBlock / (K / 4);
    int64_t wgs = (M + gpb - 1) / gpb;
    if (wgs > kNarrowWgs) wgs = kNarrowWgs;
    const size_t lds = (size_t)gpb * N * (K + 1) * sizeof(float);
    void (*kern)(const float*, const float*, int64_t, int, float*) =
        N == 1 ? mlp_dw_narrow_kernel<1> : (N == 2 ? mlp_dw_narrow_kernel<2> : (N == 3 ? mlp_dw_narrow_kernel<3> : mlp_dw_narrow_kernel<4>));
    hipLaunchKernelGGL(kern, dim3((unsigned)wgs), dim3(kBlock), lds, s, dZ, X, M, K, part);
    RC_LAUNCH_CHECK();
    const int64_t total = (int64_t)N * (K + 1);
    hipLaunchKernelGGL(mlp_reduce_wave_kernel, dim3((unsigned)((total + kBlock / 64 - 1) / (kBlock / 64))), dim3(kBlock), 0, s, part,
                       (int)wgs, N, K, dW, db);
    RC_LAUNCH_CHECK();
  } else {  // [dW | db][n, k] = sum_m dZ[m, n] [X | 1][m, k], the batch cut into splits
    const int big = (reinterpret_cast<uintptr_t>(dZ) % 16 == 0 && reinterpret_cast<uintptr_t>(X) % 16 == 0) ? mlp_dw_big_splits(M, N, K) : 0;
    const int splits = big ? big : mlp_splits(M, N, K);
    int64_t per = (M + splits - 1) / splits;
    per = (per + kMlpBK - 1) / kMlpBK * kMlpBK;
    MlpGemm g;
    memset(&g, 0, sizeof(g));
    g.A = MlpOperand{dZ, N, 0};
    g.B = MlpOperand{X, K, 0};
    g.M = N; g.N = K; g.K = (int)per; g.ones_col = K; g.C = part; g.ldc = K + 1; g.split_stride_k = per; g.k_total = M;
    const int used = (int)((M + per - 1) / per);
    RC_TRY(mlp_launch(g, used, s, big != 0));
    const int64_t total = (int64_t)N * (K + 1);
    int64_t blocks = (total + kBlock - 1) / kBlock;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(mlp_reduce_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, s, part, used, N, K, dW, db);
    RC_LAUNCH_CHECK();
  }
  return RC_OK;
}
