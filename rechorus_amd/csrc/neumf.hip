// neumf.hip -- the NeuMF interaction head (reference: models/general/NeuMF.py:56-76), one
// hidden MLP layer (the reference's default `--layers [64]`), forward and backward, fp32.
//
//   mf   = mf_u[u] * mf_i[i]                          (GMF branch, d)
//   h0   = [mlp_u[u] ; mlp_i[i]]                      (2d)
//   h1   = drop(relu(W1 h0 + b1))                     (L1; training-mode dropout p in the DROP instantiations:
//                                                      counter-based mask keyed by a device-side seed, never stored)
//   pred = w_out[:d] . mf + w_out[d:] . h1            (Linear(d+L1, 1, bias=False))
//
// 33 kFLOP (fwd) per candidate against 1 KB of gathered rows at d=128: the MLP sits on the
// fp32 ridge of the chip, so it runs on the fp32 MATRIX cores: v_mfma_f32_32x32x2_f32 (exact
// f32 FMA chains -- same rounding class as the reference's addmm) with both operands in LDS.
//
// A persistent workgroup keeps W1 in LDS (padded stride 2d+1: conflict-free ds_read_b32 for the
// MFMA operand fetch, lane <-> row) and walks 64-candidate tiles:
//   gather   lane-group (d/4 lanes) per candidate: 4 table rows; mf dot by DPP row reduction;
//            h0 tile -> LDS As[64][2d+1]
//   fwd      Z1^T[L1 x 64] = W1 . h0^T        (A = W1 block, B = h0^T block, K = 2d)
//   (bwd)    dz1 = g * w_h * relu'            -> LDS Zs[L1][65]
//            dh0^T[2d x 64] = W1^T . dz1      (K = L1)   -> per-occurrence row gradients
//            dW1[L1 x 2d] += dz1 . h0         (K = 64)   accumulators persist across tiles
// The backward kernel recomputes the forward GEMM instead of storing activations.  Dense
// parameter gradients leave the kernel as per-workgroup partials summed in fixed order by
// neumf_reduce_partials_kernel (deterministic, no float atomics).  Table-row gradients are
// written per occurrence ([B*C, d] x 4) and consumed by rc_segmented_update_pair (mf / mlp table of a side in one pass).
#include "common.hpp"
#include "philox.hpp"

namespace rc {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kTileM = 64;

struct NeumfArgs {
  const float* mf_u;
  const float* mf_i;
  const float* mlp_u;
  const float* mlp_i;
  const float* W1;     // [L1, 2d] (nn.Linear.weight)
  const float* b1;     // [L1]
  const float* w_out;  // [d + L1]  (prediction.weight[0])
  const int64_t* uid;
  const int64_t* iid;
  int B, C;
  int64_t n;           // B*C
  float* pred;         // fwd
  const float* gpred;  // bwd
  float* g_mf_u;       // bwd: [n, d] per-occurrence row gradients
  float* g_mf_i;
  float* g_mlp_u;
  float* g_mlp_i;
  float* pW1;          // bwd: per-workgroup partials [n_wg][L1*2d], [n_wg][L1], [n_wg][d+L1]
  float* pb1;
  float* pwout;
  // dropout on the hidden layer (NeuMF.py:70, nn.Dropout after the ReLU): element (candidate n, feature f) is
  // dropped iff word (f & 3) of Philox(seed, n, f >> 2) < drop_thresh; kept values are scaled by keep_scale.
  // seed_dev == nullptr: no dropout.  The seed is read from device memory so that a captured step replays
  // with a new mask once the host (or the graph) has bumped it.
  const uint64_t* seed_dev;
  uint32_t drop_thresh;
  float keep_scale;
};

template <int D, int L1>
struct NeumfCfg {
  static constexpr int K0 = 2 * D;
  // LDS row stride of Ws / As (floats): a multiple of 4 whose quarter is odd -- rows are 16-byte aligned (float4 staging, one
  // ds_read_b128 = four reduction indices of the forward product's operands) and the sixteen lanes of a ds_read_b128 pass start
  // in sixteen different 4-bank groups.  (K0 + 1 in rounds 1-2: scalar staging, one ds_read_b32 per operand and MFMA.)
  static constexpr int SW = K0 + 4;
  static constexpr int SZ = kTileM + 1;  // LDS row stride of Zs
  static constexpr int NRB = L1 / 32;    // 32-row blocks of hidden features
  static constexpr int NKB = K0 / 32;    // 32-row blocks of h0 features
  static constexpr int LPR = D / 4;      // lanes per table row
  static constexpr int NG = kBlock / LPR;
  static constexpr int kLdsFloats = L1 * SW + kTileM * SW + L1 * SZ + L1 + (D + L1) + 2 * kTileM +
                                    NRB * kTileM + kBlock * 4;
  static_assert(D % 32 == 0 && L1 % 32 == 0, "NeuMF MFMA path needs d and L1 multiples of 32");
};

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// accumulator register r of lane -> row inside a 32x32 block (column = lane & 31)
__device__ __forceinline__ int acc_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

template <int D, int L1, bool BWD, bool DROP>
__global__ __launch_bounds__(kBlock) void neumf_kernel(NeumfArgs a) {
  using Cfg = NeumfCfg<D, L1>;
  constexpr int K0 = Cfg::K0, SW = Cfg::SW, SZ = Cfg::SZ, NRB = Cfg::NRB, NKB = Cfg::NKB;
  constexpr int LPR = Cfg::LPR, NG = Cfg::NG, M = kTileM;
  constexpr int NB_F = NRB * 2, QF = (NB_F + 3) / 4;    // fwd blocks (rb, cb)
  constexpr int NB_H = NKB * 2, QH = (NB_H + 3) / 4;    // dh0 blocks (kb, cb)
  constexpr int NB_W = NRB * NKB, QW = (NB_W + 3) / 4;  // dW1 blocks (rb, kb)

  extern __shared__ float lds[];
  float* Ws = lds;
  float* As = Ws + L1 * SW;
  float* Zs = As + M * SW;
  float* sb1 = Zs + L1 * SZ;
  float* swo = sb1 + L1;
  float* sg = swo + (D + L1);
  float* smf = sg + M;
  float* spp = smf + M;          // [NRB][M]
  float* sred = spp + NRB * M;   // [kBlock*4] scratch for the final reductions

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int grp = tid / LPR;
  const int l = tid % LPR;

  for (int i = tid; i < L1 * K0 / 4; i += kBlock)
    *reinterpret_cast<float4*>(Ws + (i / (K0 / 4)) * SW + 4 * (i % (K0 / 4))) = reinterpret_cast<const float4*>(a.W1)[i];
  for (int i = tid; i < L1; i += kBlock) sb1[i] = a.b1[i];
  for (int i = tid; i < D + L1; i += kBlock) swo[i] = a.w_out[i];
  __syncthreads();
  const float4 wm = make_float4(swo[4 * l], swo[4 * l + 1], swo[4 * l + 2], swo[4 * l + 3]);

  f32x16 accW[QW];
  float db1acc[QF][16], dwhacc[QF][16];
  float4 accwm = make_float4(0.f, 0.f, 0.f, 0.f);
  if (BWD) {
#pragma unroll
    for (int s = 0; s < QW; ++s)
#pragma unroll
      for (int r = 0; r < 16; ++r) accW[s][r] = 0.f;
#pragma unroll
    for (int s = 0; s < QF; ++s)
#pragma unroll
      for (int r = 0; r < 16; ++r) db1acc[s][r] = dwhacc[s][r] = 0.f;
  }

  const int64_t n_tiles = (a.n + M - 1) / M;
  // Software pipeline over tiles: the table rows of tile t+1 are requested (into registers) right after
  // tile t's rows have been staged in LDS, so their latency is covered by tile t's GEMMs.  A workgroup
  // is alone on its CU (LDS), nothing else would hide it: SQ counters showed the waves parked on
  // s_waitcnt for 65 % of their cycles before (forward 0.305 -> 0.251 ms).  NIT iterations x 4 float4 per thread, registers are free
  // at one wave per SIMD.
  constexpr int NIT = M / NG;
  float4 pmu[NIT], pmi[NIT], phu[NIT], phi[NIT];
  float pg[NIT];
  auto fetch = [&](int64_t tile) {
    const int64_t n0 = tile * M;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int64_t n = n0 + grp + it * NG;
      const bool valid = n < a.n;
      const int64_t nn = valid ? n : a.n - 1;
      const int64_t u = a.uid[nn / a.C];
      const int64_t item = a.iid[nn];
      pmu[it] = reinterpret_cast<const float4*>(a.mf_u + u * D)[l];
      pmi[it] = reinterpret_cast<const float4*>(a.mf_i + item * D)[l];
      phu[it] = reinterpret_cast<const float4*>(a.mlp_u + u * D)[l];
      phi[it] = reinterpret_cast<const float4*>(a.mlp_i + item * D)[l];
      pg[it] = (BWD && valid) ? a.gpred[nn] : 0.f;
    }
  };
  // (forward only: in the backward kernel the prefetch competes with the per-occurrence gradient stores of the
  // current tile and measured slower wherever it is placed -- 0.70 ms right after staging, 0.69 ms under the dW1 GEMM,
  // against 0.54 ms with the rows requested together at the top of their tile; round 3: also 0.71 ms with the next
  // tile's rows requested at the TOP of the tile, before any of its stores, into a second register set -- 512
  // registers, 8-14 spilled)
  if (!BWD && (int64_t)blockIdx.x < n_tiles) fetch(blockIdx.x);
  for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int64_t n0 = tile * M;
    if (BWD) fetch(tile);
    // ---- stage the prefetched rows: one lane-group per candidate (uniform trip count: NG divides M) ---
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int cc = grp + it * NG;
      const int64_t n = n0 + cc;
      const bool valid = n < a.n;
      const float4 mu = pmu[it], mi = pmi[it];
      float4 hu = phu[it], hi = phi[it];
      if (!valid) hu = hi = make_float4(0.f, 0.f, 0.f, 0.f);
      const float4 mm = make_float4(mu.x * mi.x, mu.y * mi.y, mu.z * mi.z, mu.w * mi.w);
      const float dot = row_allreduce_sum<LPR>(dot4(wm, mm));
      if (l == 0) smf[cc] = dot;
      float* arow = As + cc * SW;
      *reinterpret_cast<float4*>(arow + 4 * l) = hu;
      *reinterpret_cast<float4*>(arow + D + 4 * l) = hi;
      if (BWD) {
        const float g = pg[it];
        if (l == 0) sg[cc] = g;
        if (valid) {  // d pred / d mf rows: g * w_mf * (other row)
          const float4 gw = make_float4(g * wm.x, g * wm.y, g * wm.z, g * wm.w);
          reinterpret_cast<float4*>(a.g_mf_u + n * D)[l] = make_float4(gw.x * mi.x, gw.y * mi.y, gw.z * mi.z, gw.w * mi.w);
          reinterpret_cast<float4*>(a.g_mf_i + n * D)[l] = make_float4(gw.x * mu.x, gw.y * mu.y, gw.z * mu.z, gw.w * mu.w);
        }
        accwm.x = fmaf(g, mm.x, accwm.x); accwm.y = fmaf(g, mm.y, accwm.y);
        accwm.z = fmaf(g, mm.z, accwm.z); accwm.w = fmaf(g, mm.w, accwm.w);
      }
    }
    __syncthreads();
    if (!BWD && tile + gridDim.x < n_tiles) fetch(tile + gridDim.x);  // in flight during this tile's GEMMs

    // ---- forward GEMM  Z1^T = W1 . h0^T, epilogue ---------------------------------------------
#pragma unroll
    for (int s = 0; s < QF; ++s) {
      const int q = wave + 4 * s;
      if (q < NB_F) {  // wave-uniform
        const int rb = q % NRB, cb = q / NRB;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        // both operands run along the reduction index in LDS: one ds_read_b128 each feeds four MFMAs -- MFMA (u, e) contracts
        // k = 8 u + 4 (lane >> 5) + e (any order of the reduction is a valid product as long as both operands agree)
        const float* wa = Ws + (rb * 32 + (lane & 31)) * SW + 4 * (lane >> 5);
        const float* hb = As + (cb * 32 + (lane & 31)) * SW + 4 * (lane >> 5);
#pragma unroll 4
        for (int k0 = 0; k0 < K0; k0 += 8) {
          const float4 w4 = *reinterpret_cast<const float4*>(wa + k0);
          const float4 h4 = *reinterpret_cast<const float4*>(hb + k0);
          acc = mfma32(w4.x, h4.x, acc);
          acc = mfma32(w4.y, h4.y, acc);
          acc = mfma32(w4.z, h4.z, acc);
          acc = mfma32(w4.w, h4.w, acc);
        }
        const int cand = cb * 32 + (lane & 31);
        const float gj = BWD ? sg[cand] : 0.f;
        float pp = 0.f;
        float keep[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) keep[r] = 1.0f;
        if (DROP) {  // 4 Philox blocks per lane, one per run of 4 consecutive features
          const uint64_t seed = *a.seed_dev;
#pragma unroll
          for (int q4 = 0; q4 < 4; ++q4) {
            uint32_t w[4];
            philox4x32_10(seed, (uint64_t)(n0 + cand), (uint32_t)((rb * 32 + acc_row(4 * q4, lane)) >> 2), w);
#pragma unroll
            for (int e = 0; e < 4; ++e) keep[4 * q4 + e] = w[e] < a.drop_thresh ? 0.f : a.keep_scale;
          }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int f = rb * 32 + acc_row(r, lane);
          const float z = acc[r] + sb1[f];
          const float h = fmaxf(z, 0.f) * keep[r];
          const float wh = swo[D + f];
          pp = fmaf(wh, h, pp);
          if (BWD) {
            const float dz = z > 0.f ? gj * wh * keep[r] : 0.f;
            Zs[f * SZ + cand] = dz;
            db1acc[s][r] += dz;
            dwhacc[s][r] = fmaf(gj, h, dwhacc[s][r]);
          }
        }
        pp += __shfl_xor(pp, 32, 64);
        if (lane < 32) spp[rb * M + cand] = pp;
      }
    }
    __syncthreads();

    if (!BWD) {
      for (int cc = tid; cc < M; cc += kBlock) {
        const int64_t n = n0 + cc;
        if (n < a.n) {
          float p = smf[cc];
#pragma unroll
          for (int rb = 0; rb < NRB; ++rb) p += spp[rb * M + cc];
          a.pred[n] = p;
        }
      }
    } else {
      // ---- dh0^T = W1^T . dz1  ->  per-occurrence gradients of the mlp_u / mlp_i rows ----------
#pragma unroll
      for (int s = 0; s < QH; ++s) {
        const int q = wave + 4 * s;
        if (q < NB_H) {
          const int kb = q % NKB, cb = q / NKB;
          f32x16 acc;
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[r] = 0.f;
          const float* wa = Ws + (lane >> 5) * SW + kb * 32 + (lane & 31);
          const float* zb = Zs + (lane >> 5) * SZ + cb * 32 + (lane & 31);
#pragma unroll 8
          for (int f0 = 0; f0 < L1; f0 += 2) acc = mfma32(wa[f0 * SW], zb[f0 * SZ], acc);
          const int64_t n = n0 + cb * 32 + (lane & 31);
          if (n < a.n) {
            float* base = (kb * 32 < D) ? (a.g_mlp_u + n * D + kb * 32) : (a.g_mlp_i + n * D + (kb * 32 - D));
#pragma unroll
            for (int rq = 0; rq < 4; ++rq)
              reinterpret_cast<float4*>(base + 8 * rq + 4 * (lane >> 5))[0] =
                  make_float4(acc[4 * rq], acc[4 * rq + 1], acc[4 * rq + 2], acc[4 * rq + 3]);
          }
        }
      }
      // ---- dW1 += dz1 . h0 (accumulators live across tiles) -------------------------------------
#pragma unroll
      for (int s = 0; s < QW; ++s) {
        const int q = wave + 4 * s;
        if (q < NB_W) {
          const int rb = q % NRB, kb = q / NRB;
          const float* za = Zs + (rb * 32 + (lane & 31)) * SZ + (lane >> 5);
          const float* hb = As + (lane >> 5) * SW + kb * 32 + (lane & 31);
#pragma unroll 8
          for (int c0 = 0; c0 < M; c0 += 2) accW[s] = mfma32(za[c0], hb[c0 * SW], accW[s]);
        }
      }
    }
    __syncthreads();  // As / Zs / sg / spp are rewritten by the next tile
  }

  if (BWD) {
    const size_t wg = blockIdx.x;
    // dW1 partial: block (rb, kb): row = hidden feature, column = h0 feature
#pragma unroll
    for (int s = 0; s < QW; ++s) {
      const int q = wave + 4 * s;
      if (q < NB_W) {
        const int rb = q % NRB, kb = q / NRB;
#pragma unroll
        for (int r = 0; r < 16; ++r)
          a.pW1[wg * (L1 * K0) + (size_t)(rb * 32 + acc_row(r, lane)) * K0 + kb * 32 + (lane & 31)] = accW[s][r];
      }
    }
    // db1 / dw_h: sum the per-lane partials over the 32 candidate columns, then over the
    // column blocks (different waves) through LDS, fixed order
    for (int i = tid; i < 2 * L1 * 2; i += kBlock) sred[i] = 0.f;
    __syncthreads();
#pragma unroll
    for (int s = 0; s < QF; ++s) {
      const int q = wave + 4 * s;
      if (q < NB_F) {
        const int rb = q % NRB, cb = q / NRB;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float x = db1acc[s][r], y = dwhacc[s][r];
#pragma unroll
          for (int off = 16; off >= 1; off >>= 1) {
            x += __shfl_xor(x, off, 64);
            y += __shfl_xor(y, off, 64);
          }
          if ((lane & 31) == 0) {
            const int f = rb * 32 + acc_row(r, lane);
            sred[(cb * 2 + 0) * L1 + f] = x;   // [cb][0: db1, 1: dwh][f]
            sred[(cb * 2 + 1) * L1 + f] = y;
          }
        }
      }
    }
    __syncthreads();
    for (int f = tid; f < L1; f += kBlock) {
      a.pb1[wg * L1 + f] = sred[0 * L1 + f] + sred[2 * L1 + f];
      a.pwout[wg * (D + L1) + D + f] = sred[1 * L1 + f] + sred[3 * L1 + f];
    }
    __syncthreads();
    // dw_mf: per-lane float4 partials -> sum over the NG lane-groups in group order
    reinterpret_cast<float4*>(sred)[tid] = accwm;
    __syncthreads();
    for (int k = tid; k < D; k += kBlock) {
      float acc = 0.f;
      for (int g2 = 0; g2 < NG; ++g2) acc += sred[(g2 * LPR + k / 4) * 4 + (k & 3)];
      a.pwout[wg * (D + L1) + k] = acc;
    }
  }
}

// out[i] = sum_w p[w][i] for the three partial arrays of one backward call, in ONE launch.
// A workgroup owns 64 consecutive outputs: wave v sums the partials w = v, v+4, v+8, ... with four
// interleaved accumulators (16 independent chains of n_wg/16 adds instead of one chain of n_wg -- the
// serial version cost 60 us per array, pure latency), then the 16 chain sums are combined in a FIXED
// order through LDS: deterministic, no float atomics.
struct ReduceArgs {
  const float* p[3];
  float* out[3];
  int count[3];
  int first_block[4];  // block ranges of the three arrays
  int n_wg;
};

__global__ __launch_bounds__(kBlock) void neumf_reduce_partials_kernel(ReduceArgs r) {
  __shared__ float sm[4][4][64];
  int which = 0;
  while (which < 2 && (int)blockIdx.x >= r.first_block[which + 1]) ++which;  // block-uniform
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int count = r.count[which];
  const int i = ((int)blockIdx.x - r.first_block[which]) * 64 + lane;
  const float* __restrict__ p = r.p[which];
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  if (i < count) {
    int w = wave;
    for (; w + 12 < r.n_wg; w += 16) {
      acc[0] += p[(size_t)w * count + i];
      acc[1] += p[(size_t)(w + 4) * count + i];
      acc[2] += p[(size_t)(w + 8) * count + i];
      acc[3] += p[(size_t)(w + 12) * count + i];
    }
    for (int k = 0; w < r.n_wg; w += 4, ++k) acc[k] += p[(size_t)w * count + i];
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) sm[wave][k][lane] = acc[k];
  __syncthreads();
  if (wave == 0 && i < count) {
    float total = 0.f;
#pragma unroll
    for (int v = 0; v < 4; ++v)
#pragma unroll
      for (int k = 0; k < 4; ++k) total += sm[v][k][lane];
    r.out[which][i] = total;
  }
}

// the three partial arrays of one backward call (also neumf_step.hip's) -> dW1, db1, dw_out
int neumf_reduce_partials(const float* pW1, const float* pb1, const float* pwout, float* dW1, float* db1, float* dw_out, int cW, int cb,
                          int co, int n_wg, hipStream_t s) {
  ReduceArgs r;
  r.p[0] = pW1; r.p[1] = pb1; r.p[2] = pwout;
  r.out[0] = dW1; r.out[1] = db1; r.out[2] = dw_out;
  r.count[0] = cW; r.count[1] = cb; r.count[2] = co;
  r.first_block[0] = 0;
  for (int k = 0; k < 3; ++k) r.first_block[k + 1] = r.first_block[k] + (r.count[k] + 63) / 64;
  r.n_wg = n_wg;
  hipLaunchKernelGGL(neumf_reduce_partials_kernel, dim3(r.first_block[3]), dim3(kBlock), 0, s, r);
  RC_LAUNCH_CHECK();
  return RC_OK;
}

// ---- forward on 16-candidate tiles (no dropout) -----------------------------------------------------------------------------------------
// The forward pass alone is bound by its gathers (4 table rows of d floats per candidate: 671 MB at the config-4 shape), not by
// the 11 GFLOP of the hidden layer: neumf_kernel<.., false, ..> stages every 64-candidate tile through LDS behind barriers with ONE
// wave per SIMD, so gather latency, staging and the MFMA chain take turns (0.25 ms: 2.7 TB/s of gathers).  Here a wave owns a
// 16-candidate tile on its own: the mlp rows go from global memory straight into the B operand registers of v_mfma_f32_16x16x4_f32
// (Z1^T = W1 h0^T: lane (i, g) holds h0[row i][16 c + 4 g ..], the layout a float4 load delivers), W1 waits in LDS (row stride
// 2 d + 4: one ds_read_b128 feeds four MFMAs), the GMF product is formed from the two mf rows as they arrive, the epilogue
// (bias, ReLU, output dot) runs on the accumulators and one xor-shuffle pair completes the row's prediction.  No LDS tile, no
// barrier after the weights are staged, eight waves per workgroup whose gathers and MFMA chains overlap freely.
typedef float f32x4m __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4m mfma16(float a, float b, f32x4m c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
constexpr int kFwd16Waves = 8;

template <int D, int L1>
__global__ __launch_bounds__(64 * kFwd16Waves) void neumf_fwd16_kernel(NeumfArgs a) {
  constexpr int K0 = 2 * D, SW = K0 + 4, NCU = D / 16, NT = L1 / 16;
  static_assert(NT % 2 == 0, "output tiles are processed in pairs");
  extern __shared__ float lds[];
  float* Ws = lds;                 // [L1][SW]
  float* sb1 = Ws + L1 * SW;       // [L1]
  float* swo = sb1 + L1;           // [D + L1]
  for (int i = threadIdx.x; i < L1 * K0 / 4; i += blockDim.x)
    *reinterpret_cast<float4*>(Ws + (i / (K0 / 4)) * SW + 4 * (i % (K0 / 4))) = reinterpret_cast<const float4*>(a.W1)[i];
  for (int i = threadIdx.x; i < L1; i += blockDim.x) sb1[i] = a.b1[i];
  for (int i = threadIdx.x; i < D + L1; i += blockDim.x) swo[i] = a.w_out[i];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 15, g = lane >> 4;
  const int nw = blockDim.x >> 6;
  const int64_t tiles = (a.n + 15) / 16;
  for (int64_t t = (int64_t)blockIdx.x * nw + wave; t < tiles; t += (int64_t)gridDim.x * nw) {
    asm volatile("" ::: "memory");   // the weights are re-read from LDS per tile (the compiler would hoist them into registers)
    const int64_t n = 16 * t + i;
    const bool valid = n < a.n;
    const int64_t nn = valid ? n : a.n - 1;
    const int64_t u = a.uid[nn / a.C], item = a.iid[nn];
    float x[2 * NCU][4];
    const float* hu = a.mlp_u + u * D + 4 * g;
    const float* hi = a.mlp_i + item * D + 4 * g;
#pragma unroll
    for (int c = 0; c < NCU; ++c) {
      const float4 v = *reinterpret_cast<const float4*>(hu + 16 * c), w = *reinterpret_cast<const float4*>(hi + 16 * c);
      x[c][0] = v.x; x[c][1] = v.y; x[c][2] = v.z; x[c][3] = v.w;
      x[NCU + c][0] = w.x; x[NCU + c][1] = w.y; x[NCU + c][2] = w.z; x[NCU + c][3] = w.w;
    }
    // GMF branch: the lane's share of w_mf . (mf_u * mf_i)
    float pp = 0.f;
    const float* mu = a.mf_u + u * D + 4 * g;
    const float* mi = a.mf_i + item * D + 4 * g;
#pragma unroll
    for (int c = 0; c < NCU; ++c) {
      const float4 p = *reinterpret_cast<const float4*>(mu + 16 * c), q = *reinterpret_cast<const float4*>(mi + 16 * c);
      const float4 w = *reinterpret_cast<const float4*>(swo + 16 * c + 4 * g);
      pp = fmaf(w.x, p.x * q.x, pp); pp = fmaf(w.y, p.y * q.y, pp); pp = fmaf(w.z, p.z * q.z, pp); pp = fmaf(w.w, p.w * q.w, pp);
    }
    // hidden layer: accumulator r of output tile nt = Z1[row i][16 nt + 4 g + r]
#pragma unroll
    for (int nt = 0; nt < NT; nt += 2) {
      f32x4m a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
      const float* w0p = Ws + (16 * nt + i) * SW + 4 * g;
      const float* w1p = Ws + (16 * (nt + 1) + i) * SW + 4 * g;
#pragma unroll
      for (int c = 0; c < 2 * NCU; ++c) {
        const float4 w0 = *reinterpret_cast<const float4*>(w0p + 16 * c), w1 = *reinterpret_cast<const float4*>(w1p + 16 * c);
        a0 = mfma16(w0.x, x[c][0], a0); a1 = mfma16(w1.x, x[c][0], a1);
        a0 = mfma16(w0.y, x[c][1], a0); a1 = mfma16(w1.y, x[c][1], a1);
        a0 = mfma16(w0.z, x[c][2], a0); a1 = mfma16(w1.z, x[c][2], a1);
        a0 = mfma16(w0.w, x[c][3], a0); a1 = mfma16(w1.w, x[c][3], a1);
      }
      const float4 b0 = *reinterpret_cast<const float4*>(sb1 + 16 * nt + 4 * g), b1v = *reinterpret_cast<const float4*>(sb1 + 16 * (nt + 1) + 4 * g);
      const float4 o0 = *reinterpret_cast<const float4*>(swo + D + 16 * nt + 4 * g), o1 = *reinterpret_cast<const float4*>(swo + D + 16 * (nt + 1) + 4 * g);
      pp = fmaf(o0.x, fmaxf(a0[0] + b0.x, 0.f), pp); pp = fmaf(o0.y, fmaxf(a0[1] + b0.y, 0.f), pp);
      pp = fmaf(o0.z, fmaxf(a0[2] + b0.z, 0.f), pp); pp = fmaf(o0.w, fmaxf(a0[3] + b0.w, 0.f), pp);
      pp = fmaf(o1.x, fmaxf(a1[0] + b1v.x, 0.f), pp); pp = fmaf(o1.y, fmaxf(a1[1] + b1v.y, 0.f), pp);
      pp = fmaf(o1.z, fmaxf(a1[2] + b1v.z, 0.f), pp); pp = fmaf(o1.w, fmaxf(a1[3] + b1v.w, 0.f), pp);
    }
    pp += __shfl_xor(pp, 16, 64);
    pp += __shfl_xor(pp, 32, 64);
    if (g == 0 && valid) a.pred[n] = pp;
  }
}

// RC_NEUMF_FWD16=0: the 64-candidate-tile kernel for the forward pass as well (A/B timing)
static bool neumf_fwd16_enabled() {
  static const bool on = [] {
    const char* v = getenv("RC_NEUMF_FWD16");
    return !(v && v[0] == '0');
  }();
  return on;
}

template <int D, int L1>
static int launch_neumf_fwd16(const NeumfArgs& a, hipStream_t s) {
  const size_t lds_bytes = ((size_t)L1 * (2 * D + 4) + L1 + D + L1) * sizeof(float);
  auto kern = neumf_fwd16_kernel<D, L1>;
  RC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
  const int64_t tiles = (a.n + 15) / 16;
  const int per_cu = (int)((160 * 1024) / lds_bytes) >= 2 ? 2 : 1;   // (128 registers per lane: two 8-wave workgroups share a CU)
  int64_t grid = (tiles + kFwd16Waves - 1) / kFwd16Waves;
  if (grid > 256 * per_cu) grid = 256 * per_cu;
  hipLaunchKernelGGL(kern, dim3((unsigned)(grid < 1 ? 1 : grid)), dim3(64 * kFwd16Waves), lds_bytes, s, a);
  RC_LAUNCH_CHECK();
  return RC_OK;
}

template <int D, int L1, bool BWD, bool DROP>
static int launch_neumf(const NeumfArgs& a, int n_wg, hipStream_t s) {
  using Cfg = NeumfCfg<D, L1>;
  const size_t lds_bytes = (size_t)Cfg::kLdsFloats * sizeof(float);
  auto kern = neumf_kernel<D, L1, BWD, DROP>;
  RC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                             (int)lds_bytes));
  hipLaunchKernelGGL(kern, dim3(n_wg), dim3(kBlock), lds_bytes, s, a);
  RC_LAUNCH_CHECK();
  return RC_OK;
}

static size_t neumf_lds_bytes(int d, int l1) {
  const size_t k0 = 2 * (size_t)d;
  return sizeof(float) * ((size_t)l1 * (k0 + 4) + kTileM * (k0 + 4) + (size_t)l1 * (kTileM + 1) + l1 + (d + l1) +
                          2 * kTileM + (l1 / 32) * kTileM + kBlock * 4);
}

static bool neumf_supported(int d, int l1) {
  const bool shape = (d == 32 || d == 64 || d == 128) && (l1 == 32 || l1 == 64 || l1 == 128);
  return shape && neumf_lds_bytes(d, l1) <= 160 * 1024;
}

static int neumf_grid(int64_t n, int d, int l1) {
  const int64_t tiles = (n + kTileM - 1) / kTileM;
  const int per_cu = (int)((160 * 1024) / neumf_lds_bytes(d, l1));
  int64_t g = 256 * (int64_t)(per_cu < 1 ? 1 : (per_cu > 2 ? 2 : per_cu));
  if (g > tiles) g = tiles;
  return (int)(g < 1 ? 1 : g);
}

template <bool BWD>
static int dispatch_neumf(const NeumfArgs& a, int d, int l1, int n_wg, hipStream_t s) {
  if (!BWD && !a.seed_dev && neumf_fwd16_enabled()) {   // forward without dropout: 16-candidate tiles
#define RC_NF(D_, L_) \
  if (d == D_ && l1 == L_) return launch_neumf_fwd16<D_, L_>(a, s)
    RC_NF(32, 32); RC_NF(32, 64); RC_NF(32, 128);
    RC_NF(64, 32); RC_NF(64, 64); RC_NF(64, 128);
    RC_NF(128, 32); RC_NF(128, 64);
#undef RC_NF
  }
#define RC_NM(D_, L_) \
  if (d == D_ && l1 == L_)                                                                 \
    return a.seed_dev ? launch_neumf<D_, L_, BWD, true>(a, n_wg, s) : launch_neumf<D_, L_, BWD, false>(a, n_wg, s)
  RC_NM(32, 32); RC_NM(32, 64); RC_NM(32, 128);
  RC_NM(64, 32); RC_NM(64, 64); RC_NM(64, 128);
  RC_NM(128, 32); RC_NM(128, 64);
#undef RC_NM
  return fail(RC_ERR_UNSUPPORTED, "NeuMF: no kernel for d=%d, hidden=%d", d, l1);
}

}  // namespace rc

using namespace rc;

extern "C" int rc_neumf_supported(int d, int l1) {
  return (neumf_supported(d, l1) && !(d == 128 && l1 == 128)) ? 1 : 0;
}

extern "C" size_t rc_neumf_workspace_bytes(int B, int C, int d, int l1) {
  if (!rc_neumf_supported(d, l1) || B < 1 || C < 1) return 0;
  const int n_wg = neumf_grid((int64_t)B * C, d, l1);
  const size_t per = (size_t)l1 * 2 * d + l1 + (d + l1);
  return align_up((size_t)n_wg * per * sizeof(float), 256) + 256;
}

// dropout arguments -> kernel fields; p == 0 or seed_dev == nullptr switches the mask off
static int set_dropout(NeumfArgs& a, float drop_p, const uint64_t* seed_dev, const char* who) {
  if (!(drop_p >= 0.f && drop_p < 1.f)) return fail(RC_ERR_INVALID_ARG, "%s: dropout p=%g outside [0, 1)", who, (double)drop_p);
  if (drop_p > 0.f && seed_dev == nullptr) return fail(RC_ERR_INVALID_ARG, "%s: dropout p=%g needs a device seed", who, (double)drop_p);
  if (drop_p > 0.f) {
    a.seed_dev = seed_dev;
    a.drop_thresh = (uint32_t)((double)drop_p * 4294967296.0);
    a.keep_scale = 1.0f / (1.0f - drop_p);
  }
  return RC_OK;
}

extern "C" int rc_neumf_fwd(const float* mf_u, const float* mf_i, const float* mlp_u,
                            const float* mlp_i, const float* W1, const float* b1,
                            const float* w_out, const int64_t* uid, const int64_t* iid, int B,
                            int C, int d, int l1, float* pred, rc_stream_t stream) {
  return rc_neumf_fwd_dropout(mf_u, mf_i, mlp_u, mlp_i, W1, b1, w_out, uid, iid, B, C, d, l1, 0.f, nullptr, pred, stream);
}

extern "C" int rc_neumf_fwd_dropout(const float* mf_u, const float* mf_i, const float* mlp_u,
                                    const float* mlp_i, const float* W1, const float* b1,
                                    const float* w_out, const int64_t* uid, const int64_t* iid, int B,
                                    int C, int d, int l1, float drop_p, const uint64_t* seed_dev,
                                    float* pred, rc_stream_t stream) {
  if (B == 0) return RC_OK;
  RC_REQUIRE(mf_u && mf_i && mlp_u && mlp_i && W1 && b1 && w_out && uid && iid && pred,
             "rc_neumf_fwd: null pointer");
  RC_REQUIRE(B > 0 && C >= 1, "rc_neumf_fwd: bad shape B=%d C=%d", B, C);
  if (!rc_neumf_supported(d, l1))
    return fail(RC_ERR_UNSUPPORTED, "rc_neumf_fwd: d=%d hidden=%d not supported (d, hidden in {32,64,128}, LDS <= 160 KB)", d, l1);
  NeumfArgs a;
  memset(&a, 0, sizeof(a));
  a.mf_u = mf_u; a.mf_i = mf_i; a.mlp_u = mlp_u; a.mlp_i = mlp_i; a.W1 = W1; a.b1 = b1; a.w_out = w_out;
  a.uid = uid; a.iid = iid; a.B = B; a.C = C; a.n = (int64_t)B * C; a.pred = pred;
  RC_TRY(set_dropout(a, drop_p, seed_dev, "rc_neumf_fwd"));
  return dispatch_neumf<false>(a, d, l1, neumf_grid(a.n, d, l1), as_stream(stream));
}

extern "C" int rc_neumf_bwd(const float* mf_u, const float* mf_i, const float* mlp_u,
                            const float* mlp_i, const float* W1, const float* b1,
                            const float* w_out, const int64_t* uid, const int64_t* iid,
                            const float* gpred, int B, int C, int d, int l1, float* g_mf_u,
                            float* g_mf_i, float* g_mlp_u, float* g_mlp_i, float* dW1, float* db1,
                            float* dw_out, void* ws, size_t ws_bytes, rc_stream_t stream) {
  return rc_neumf_bwd_dropout(mf_u, mf_i, mlp_u, mlp_i, W1, b1, w_out, uid, iid, gpred, B, C, d, l1, 0.f, nullptr,
                              g_mf_u, g_mf_i, g_mlp_u, g_mlp_i, dW1, db1, dw_out, ws, ws_bytes, stream);
}

extern "C" int rc_neumf_bwd_dropout(const float* mf_u, const float* mf_i, const float* mlp_u,
                                    const float* mlp_i, const float* W1, const float* b1,
                                    const float* w_out, const int64_t* uid, const int64_t* iid,
                                    const float* gpred, int B, int C, int d, int l1, float drop_p,
                                    const uint64_t* seed_dev, float* g_mf_u, float* g_mf_i, float* g_mlp_u,
                                    float* g_mlp_i, float* dW1, float* db1, float* dw_out, void* ws,
                                    size_t ws_bytes, rc_stream_t stream) {
  if (B == 0) return RC_OK;
  RC_REQUIRE(mf_u && mf_i && mlp_u && mlp_i && W1 && b1 && w_out && uid && iid && gpred && g_mf_u &&
                 g_mf_i && g_mlp_u && g_mlp_i && dW1 && db1 && dw_out && ws,
             "rc_neumf_bwd: null pointer");
  RC_REQUIRE(B > 0 && C >= 1, "rc_neumf_bwd: bad shape B=%d C=%d", B, C);
  if (!rc_neumf_supported(d, l1))
    return fail(RC_ERR_UNSUPPORTED, "rc_neumf_bwd: d=%d hidden=%d not supported", d, l1);
  if (ws_bytes < rc_neumf_workspace_bytes(B, C, d, l1))
    return fail(RC_ERR_WORKSPACE, "rc_neumf_bwd: workspace %zu < %zu", ws_bytes,
                rc_neumf_workspace_bytes(B, C, d, l1));
  hipStream_t s = as_stream(stream);
  NeumfArgs a;
  memset(&a, 0, sizeof(a));
  a.mf_u = mf_u; a.mf_i = mf_i; a.mlp_u = mlp_u; a.mlp_i = mlp_i; a.W1 = W1; a.b1 = b1; a.w_out = w_out;
  a.uid = uid; a.iid = iid; a.B = B; a.C = C; a.n = (int64_t)B * C; a.gpred = gpred;
  a.g_mf_u = g_mf_u; a.g_mf_i = g_mf_i; a.g_mlp_u = g_mlp_u; a.g_mlp_i = g_mlp_i;
  RC_TRY(set_dropout(a, drop_p, seed_dev, "rc_neumf_bwd"));
  const int n_wg = neumf_grid(a.n, d, l1);
  const int cW = l1 * 2 * d, cb = l1, co = d + l1;
  float* p = reinterpret_cast<float*>(ws);
  a.pW1 = p;
  a.pb1 = p + (size_t)n_wg * cW;
  a.pwout = a.pb1 + (size_t)n_wg * cb;
  RC_TRY(dispatch_neumf<true>(a, d, l1, n_wg, s));
  return neumf_reduce_partials(a.pW1, a.pb1, a.pwout, dW1, db1, dw_out, cW, cb, co, n_wg, s);
}
