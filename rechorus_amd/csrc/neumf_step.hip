// neumf_step.hip -- one BaseRunner.fit iteration of the NeuMF head in ONE kernel: forward, BPR loss, backward and the
// optimizer update of every table row that only this batch position touches.
//
// Reference: models/general/NeuMF.py:56-76 (forward), models/BaseModel.py:182-185 (loss), loss.backward() +
// optimizer.step() of helpers/BaseRunner.py:193-206.  What the three-kernel step (neumf.hip: forward kernel -> loss
// kernel -> backward kernel) pays for and this one does not:
//   * the backward re-gathers all four table rows of every candidate and writes four per-occurrence gradient rows
//     ([B C, d] x 4 = 0.67 GB at the config-4 shape) that the table updates read back;
//   * the user rows and the user half of the hidden layer are the same for the C candidates of a tuple
//     (NeuMF.py:61 tiles the user ids): W1 [mlp_u ; mlp_i] = W1u mlp_u + W1i mlp_i, so W1u mlp_u, W1u^T sum_c dz_c and
//     dW1u += (sum_c dz_c) mlp_u^T are per-TUPLE products -- 0.6 x the forward FLOPs, 0.45 x the backward's.
//
// Layout.  A wave owns 16 TUPLES: MFMA row i <-> tuple i, and the candidate loop c = 0 .. C-1 walks candidate c of all
// sixteen, so a lane (i, g) meets every prediction of its tuple: the BPR loss and dL/dpred are computed in registers.
// v_mfma_f32_16x16x4_f32 throughout (exact fp32 FMA chains); operands as in neumf_fwd16_kernel: the 16-byte slice a lane
// loads from a table row, (row i, columns 16 c + 4 g ..), IS the B operand of Z^T = W1 h0^T, W1 waits in LDS.
//   pass 1   per c: gather mlp_i / mf_i rows, z = Zu + W1i hi, pred -> LDS strip [C][16] of the wave
//   loss     per lane over its tuple's C predictions -> dL/dpred back into the strip
//   pass 2   per c: gather again (L2 / MALL: the wave touched the rows microseconds ago), recompute z, dz,
//            dh0 = W1i^T dz lands in the SAME lane layout as the gathered row, so a single-occurrence row is updated in
//            place from registers (opt_row4: SGD / Adam / Adagrad) and only multi-occurrence rows write a gradient row
//            (to the per-occurrence arrays the bucket plan's pair update consumes; singleton / multi from a byte map over
//            item ids that two marking passes fill with plain stores -- schedule-independent, see below);
//            dW1i += dz^T hi contracts over CANDIDATES: dz and hi cross to "candidate in the K index" through a
//            wave-private LDS strip (ds in-order per wave: no barrier), accumulators 64 x 128 per wave (AGPRs);
//   tuple    dhu = W1u^T sum_c dz_c, d mf_u = w_mf * sum_c g_c mf_i: ONE gradient row per tuple and table (the plan's
//            user side is per tuple already); dW1u: the four waves exchange (sum dz, mlp_u) through LDS and each owns
//            a quarter of the 64 x 128 outputs (two workgroup barriers per 64 tuples -- the only ones in the loop).
// One wave per SIMD (512 registers: 160 accumulators + the rows of two candidates in flight); the next candidate's
// mlp_i rows are requested before the current one's MFMAs (1.2 K MFMA cycles per candidate cover the gather).
// Dense gradients leave as per-workgroup partials, summed in fixed order by neumf_reduce_partials_kernel.
#include "bpr_math.hpp"
#include "common.hpp"
#include "opt_math.hpp"
#include "philox.hpp"

namespace rc {

typedef float f32x4s __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4s mma16(float a, float b, f32x4s c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

struct NeumfStepArgs {
  float* mf_u;
  float* mf_i;
  float* mlp_u;
  float* mlp_i;
  float* m_mf_i;   // optimizer state of the item tables (in-place singleton updates); null where the optimizer has none
  float* v_mf_i;
  float* m_mlp_i;
  float* v_mlp_i;
  const float* W1;
  const float* b1;
  const float* w_out;
  const int64_t* uid;
  const int64_t* iid;
  int B, C;
  float inv_b;
  const uint8_t* multi;    // multi[id] != 0: item row id occurs at least twice in the batch; null: no row is updated in place
  float* loss_vec;         // [B]
  float* pred;             // [B, C] or null
  float* g_mf_i;           // [B C, D] gradient rows of multi-occurrence item rows (other positions are not written)
  float* g_mlp_i;
  float* gu_mf;            // [B, D] per-tuple gradient rows of the user tables
  float* gu_mlp;
  int64_t ld_u, ld_i;      // row strides (floats) of the user / item tables: D for nn.Embedding weights; the sharded step's fetched
  int64_t ld_gi, ld_gu;    // row blocks hold [mf | mlp] side by side (2 D), and so do the gradient rows it sends back
  float* pW1;              // per-workgroup partials [grid][L1 2D], [grid][L1], [grid][D + L1]
  float* pb1;
  float* pwout;
  OptScalars opt;
  // DROP instantiations: training-mode dropout on the hidden layer (NeuMF.py:58/70, nn.Dropout after the ReLU) with the mask of
  // rc_neumf_fwd_dropout: feature f of candidate n = b C + c is dropped iff word (f & 3) of Philox4x32-10(key = *seed_dev,
  // counter = (n, f >> 2)) < drop_thresh; kept values are scaled by keep_scale.  Pass 2 regenerates pass 1's mask.
  const uint64_t* seed_dev;
  uint32_t drop_thresh;
  float keep_scale;
};

template <int D, int L1>
struct StepCfg {
  static constexpr int K0 = 2 * D, SW = K0 + 4, NCU = D / 16, NT = L1 / 16;
  static constexpr int SZ = L1 + 16, SH = D + 16;          // strides of the transposition strips: 16 (mod 32) floats, so the four
                                                           // candidates a K step reads sit in four different 16-bank groups
  static constexpr int NTU = NT * NCU / 4;                 // dW1u output tiles per wave
  static constexpr int kFixedFloats = L1 * SW + L1 + (D + L1) + 4 * (2 * L1 + D);
  static constexpr int kWaveFloats = 16 * SZ + 16 * SH;    // + 16 C for the prediction strip
  static_assert(NT * NCU % 4 == 0, "dW1u tiles are dealt to four waves");
};

template <int N>
__device__ __forceinline__ void load_row_slices(float (&x)[N][4], const float* p) {
#pragma unroll
  for (int c = 0; c < N; ++c) {
    const float4 v = *reinterpret_cast<const float4*>(p + 16 * c);
    x[c][0] = v.x; x[c][1] = v.y; x[c][2] = v.z; x[c][3] = v.w;
  }
}

__device__ __forceinline__ void lds4(float (&w)[4], const float* p) {
  const float4 v = *reinterpret_cast<const float4*>(p);
  w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w;
}

// The three MFMA products of the step, each written as a software pipeline: the LDS operands of the NEXT group of MFMAs are
// requested before the current group is issued, and a group holds several independent accumulator chains.  (The first version
// left the order to the compiler, which at 256 + 255 registers placed every ds_read directly in front of its consumer:
// "ds_read, s_waitcnt, 4 dependent MFMAs" 32 times per candidate in the hidden layer and "ds_read_b32, s_waitcnt, MFMA" 128
// times in dh0 -- the matrix pipe idled for an LDS round trip per group.)

// z[nt] += W[16 nt + i][col + 16 cc + 4 g + e] x[cc][e]: wp = Ws + i SW + col + 4 g; NT chains, operands of cc + 1 in flight
template <int NT, int NCU, int SW>
__device__ __forceinline__ void hidden_half(f32x4s (&z)[NT], const float* wp, const float (&x)[NCU][4]) {
  float wc[NT][4], wn[NT][4];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) lds4(wc[nt], wp + 16 * nt * SW);
#pragma unroll
  for (int cc = 0; cc < NCU; ++cc) {
    if (cc + 1 < NCU) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) lds4(wn[nt], wp + 16 * nt * SW + 16 * (cc + 1));
    }
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) z[nt] = mma16(wc[nt][e], x[cc][e], z[nt]);
    if (cc + 1 < NCU) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int e = 0; e < 4; ++e) wc[nt][e] = wn[nt][e];
    }
  }
}

// acc[k] += sum over (nt, r) of W[16 nt + 4 g + r][col + 16 (kt0 + k) + i] dz[nt][r]: wp = Ws + 4 g SW + col + 16 kt0 + i.
// MFMA (nt, r) contracts the features 16 nt + 4 g' + r; register r' of acc[k] <-> column 16 (kt0 + k) + 4 g + r', the slice
// this lane gathered.  KG chains, the KG operands of the next (nt, r) in flight.
template <int NT, int KG, int SW>
__device__ __forceinline__ void back_tiles(f32x4s (&acc)[KG], const float* wp, const f32x4s (&dz)[NT]) {
  float ac[KG], an[KG];
#pragma unroll
  for (int k = 0; k < KG; ++k) ac[k] = wp[16 * k];
#pragma unroll
  for (int idx = 0; idx < 4 * NT; ++idx) {
    if (idx + 1 < 4 * NT) {
#pragma unroll
      for (int k = 0; k < KG; ++k) an[k] = wp[(16 * ((idx + 1) / 4) + ((idx + 1) % 4)) * SW + 16 * k];
    }
#pragma unroll
    for (int k = 0; k < KG; ++k) acc[k] = mma16(ac[k], dz[idx / 4][idx % 4], acc[k]);
    if (idx + 1 < 4 * NT) {
#pragma unroll
      for (int k = 0; k < KG; ++k) ac[k] = an[k];
    }
  }
}

// keep bits of the lane's hidden features of candidate n: bit 4 nt + r <-> feature 16 nt + 4 g + r, i.e. word r of the Philox block
// 4 nt + g -- one block per MFMA tile, exactly the four features the lane holds of it
template <int NT>
__device__ __forceinline__ uint32_t keep_bits(uint64_t seed, uint64_t n, int g, uint32_t thresh) {
  uint32_t bits = 0;
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    uint32_t w[4];
    philox4x32_10(seed, n, (uint32_t)(4 * nt + g), w);
#pragma unroll
    for (int r = 0; r < 4; ++r) bits |= (w[r] < thresh ? 0u : 1u) << (4 * nt + r);
  }
  return bits;
}

template <int D, int L1, int MODE, bool DROP>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void neumf_step_kernel(NeumfStepArgs a) {
  using Cfg = StepCfg<D, L1>;
  constexpr int K0 = Cfg::K0, SW = Cfg::SW, NCU = Cfg::NCU, NT = Cfg::NT, SZ = Cfg::SZ, SH = Cfg::SH, NTU = Cfg::NTU;
  constexpr int KG = NCU < 4 ? NCU : 4;   // dh0 column tiles per MFMA group
  extern __shared__ float lds[];
  float* Ws = lds;                       // [L1][SW]  W1, user half in columns 0 .. D-1
  float* sb1 = Ws + L1 * SW;             // [L1]
  float* swo = sb1 + L1;                 // [D + L1]  w_mf | w_h
  float* wred = swo + (D + L1);          // [4 waves][db1 L1 | dw_h L1 | dw_mf D]
  float* tb = wred + 4 * (2 * L1 + D);   // per wave: Tz [16][SZ], Th [16][SH], sp [C][16]
  const int C = a.C;
  const int per_wave = Cfg::kWaveFloats + 16 * C;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, i = lane & 15, g = lane >> 4;
  float* Tz = tb + wave * per_wave;
  float* Th = Tz + 16 * SZ;
  float* sp = Th + 16 * SH;
  float* wr = wred + wave * (2 * L1 + D);

  for (int k = threadIdx.x; k < L1 * K0 / 4; k += 256)
    *reinterpret_cast<float4*>(Ws + (k / (K0 / 4)) * SW + 4 * (k % (K0 / 4))) = reinterpret_cast<const float4*>(a.W1)[k];
  for (int k = threadIdx.x; k < L1; k += 256) sb1[k] = a.b1[k];
  for (int k = threadIdx.x; k < D + L1; k += 256) swo[k] = a.w_out[k];
  for (int k = threadIdx.x; k < 4 * (2 * L1 + D); k += 256) wred[k] = 0.f;
  __syncthreads();
  const uint64_t seed = DROP ? *a.seed_dev : 0;

  f32x4s accW[NT][NCU];   // dW1[:, D:] of this wave's tuples: tile (ft, kt), register r <-> W1[16 ft + 4 g + r][D + 16 kt + i]
  f32x4s accU[NTU];       // dW1[:, :D] tiles wave + 4 q of the workgroup's tuples
#pragma unroll
  for (int ft = 0; ft < NT; ++ft)
#pragma unroll
    for (int kt = 0; kt < NCU; ++kt) accW[ft][kt] = f32x4s{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int q = 0; q < NTU; ++q) accU[q] = f32x4s{0.f, 0.f, 0.f, 0.f};

  const float* wfwd_u = Ws + i * SW + 4 * g;            // hidden layer, user / item half
  const float* wfwd_i = wfwd_u + D;
  const float* wbwd_u = Ws + (4 * g) * SW + i;          // W1^T products, user / item half
  const float* wbwd_i = wbwd_u + D;

  const int64_t n_tiles = ((int64_t)a.B + 15) / 16;
  const int64_t n_rounds = (n_tiles + 3) / 4;
  for (int64_t round = blockIdx.x; round < n_rounds; round += gridDim.x) {
    asm volatile("" ::: "memory");   // LDS operands are re-read per use (hoisted, the weights alone are 128 registers)
    const int64_t tup = (round * 4 + wave) * 16 + i;
    const bool valid = tup < a.B;    // a tail lane works on the last tuple with dL/dpred = 0 and stores nothing
    const int64_t tt = valid ? tup : (int64_t)a.B - 1;
    const int64_t u = a.uid[tt];
    const int64_t* ip = a.iid + tt * C;
    const float* hup = a.mlp_u + u * a.ld_u + 4 * g;
    const float* mup = a.mf_u + u * a.ld_u + 4 * g;

    // ---- per tuple: Zu = W1u mlp_u (accumulator r of tile nt <-> hidden feature 16 nt + 4 g + r) ------------------------
    f32x4s Zu[NT];
    {
      float hu[NCU][4];
      load_row_slices<NCU>(hu, hup);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) Zu[nt] = f32x4s{0.f, 0.f, 0.f, 0.f};
      hidden_half<NT, NCU, SW>(Zu, wfwd_u, hu);
    }

    // ---- pass 1: predictions --------------------------------------------------------------------------------------
    // item ids run two candidates ahead of the rows, the mlp rows one candidate ahead of the MFMAs
    {
      float muw[NCU][4];
      load_row_slices<NCU>(muw, mup);
#pragma unroll
      for (int cc = 0; cc < NCU; ++cc) {
        const float4 w = *reinterpret_cast<const float4*>(swo + 16 * cc + 4 * g);
        muw[cc][0] *= w.x; muw[cc][1] *= w.y; muw[cc][2] *= w.z; muw[cc][3] *= w.w;
      }
      float hn[NCU][4];
      int64_t it0 = ip[0], it1 = ip[C > 1 ? 1 : 0];
      load_row_slices<NCU>(hn, a.mlp_i + it0 * a.ld_i + 4 * g);
      for (int c = 0; c < C; ++c) {
        asm volatile("" ::: "memory");
        float hx[NCU][4], mx[NCU][4];
#pragma unroll
        for (int cc = 0; cc < NCU; ++cc)
#pragma unroll
          for (int e = 0; e < 4; ++e) hx[cc][e] = hn[cc][e];
        load_row_slices<NCU>(mx, a.mf_i + it0 * a.ld_i + 4 * g);
        it0 = it1;
        if (c + 1 < C) {   // the next candidate's mlp rows travel during this one's MFMAs
          load_row_slices<NCU>(hn, a.mlp_i + it0 * a.ld_i + 4 * g);
          it1 = ip[c + 2 < C ? c + 2 : C - 1];
        }
        f32x4s z[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) z[nt] = Zu[nt];
        // (the mask does not depend on anything loaded: its integer arithmetic can issue beside the MFMAs below)
        const uint32_t kb = DROP ? keep_bits<NT>(seed, (uint64_t)(tt * C + c), g, a.drop_thresh) : 0u;
        hidden_half<NT, NCU, SW>(z, wfwd_i, hx);
        float pp = 0.f;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const float4 b = *reinterpret_cast<const float4*>(sb1 + 16 * nt + 4 * g);
          const float4 o = *reinterpret_cast<const float4*>(swo + D + 16 * nt + 4 * g);
          if (DROP) {
            const float bb[4] = {b.x, b.y, b.z, b.w}, oo[4] = {o.x, o.y, o.z, o.w};
#pragma unroll
            for (int r = 0; r < 4; ++r)
              pp = fmaf(oo[r], fmaxf(z[nt][r] + bb[r], 0.f) * (((kb >> (4 * nt + r)) & 1u) ? a.keep_scale : 0.f), pp);
          } else {
            pp = fmaf(o.x, fmaxf(z[nt][0] + b.x, 0.f), pp);
            pp = fmaf(o.y, fmaxf(z[nt][1] + b.y, 0.f), pp);
            pp = fmaf(o.z, fmaxf(z[nt][2] + b.z, 0.f), pp);
            pp = fmaf(o.w, fmaxf(z[nt][3] + b.w, 0.f), pp);
          }
        }
#pragma unroll
        for (int cc = 0; cc < NCU; ++cc)
#pragma unroll
          for (int e = 0; e < 4; ++e) pp = fmaf(muw[cc][e], mx[cc][e], pp);
        pp += __shfl_xor(pp, 16, 64);
        pp += __shfl_xor(pp, 32, 64);
        if (g == 0) {
          sp[c * 16 + i] = pp;
          if (a.pred && valid) a.pred[tup * C + c] = pp;
        }
      }
    }
    __builtin_amdgcn_wave_barrier();

    // ---- GeneralModel.loss on the tuple's C predictions (every lane of row i computes the same values) ----------------
    {
      const float pos = sp[i];
      float mxv = -INFINITY;
      for (int c = 1; c < C; ++c) mxv = fmaxf(mxv, sp[c * 16 + i]);
      float se = 0.f;
      for (int c = 1; c < C; ++c) se += expf(sp[c * 16 + i] - mxv);
      const float inv_se = 1.0f / se;
      float P = 0.f, A = 0.f;
      for (int c = 1; c < C; ++c) {
        const float x = sp[c * 16 + i];
        const float w = expf(x - mxv) * inv_se;
        const float s = sigmoidf_(pos - x);
        P = fmaf(w, s, P);
        A = fmaf(w, s * (1.0f - s), A);
      }
      const BprRow br = bpr_row(P, a.inv_b);
      if (g == 0 && valid) a.loss_vec[tup] = br.loss;
      const float dl = valid ? br.dLdP : 0.f;
      for (int c = 1; c < C; ++c) {
        const float x = sp[c * 16 + i];
        const float w = expf(x - mxv) * inv_se;
        const float s = sigmoidf_(pos - x);
        const float gc = dl * bpr_dP_dneg(w, s, P);
        if (g == 0) sp[c * 16 + i] = gc;
      }
      if (g == 0) sp[i] = dl * A;
    }
    __builtin_amdgcn_wave_barrier();

    // ---- pass 2: backward of the MLP branch and the mlp_i row updates ---------------------------------------------------
    f32x4s dzs[NT];      // sum_c dz_c: db1, and the tuple-level products of the user half
    float dwh[NT][4];    // sum_c g_c h1_c
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      dzs[nt] = f32x4s{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int r = 0; r < 4; ++r) dwh[nt][r] = 0.f;
    }
    {
      float hn[NCU][4];
      int64_t it0 = ip[0], it1 = ip[C > 1 ? 1 : 0];
      load_row_slices<NCU>(hn, a.mlp_i + it0 * a.ld_i + 4 * g);
      for (int c = 0; c < C; ++c) {
        asm volatile("" ::: "memory");
        float hx[NCU][4];
#pragma unroll
        for (int cc = 0; cc < NCU; ++cc)
#pragma unroll
          for (int e = 0; e < 4; ++e) hx[cc][e] = hn[cc][e];
        const int64_t item_c = it0;
        const uint8_t mflag = a.multi ? a.multi[item_c] : (uint8_t)1;
        it0 = it1;
        if (c + 1 < C) {
          load_row_slices<NCU>(hn, a.mlp_i + it0 * a.ld_i + 4 * g);
          it1 = ip[c + 2 < C ? c + 2 : C - 1];
        }
        const float gc = sp[c * 16 + i];
        const int64_t n = tup * C + c;

        // hidden layer again, dz = g w_h relu'(z)
        f32x4s dz[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) dz[nt] = Zu[nt];
        const uint32_t kb = DROP ? keep_bits<NT>(seed, (uint64_t)(tt * C + c), g, a.drop_thresh) : 0u;   // pass 1's mask again
        hidden_half<NT, NCU, SW>(dz, wfwd_i, hx);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const float4 b = *reinterpret_cast<const float4*>(sb1 + 16 * nt + 4 * g);
          const float4 o = *reinterpret_cast<const float4*>(swo + D + 16 * nt + 4 * g);
          const float bb[4] = {b.x, b.y, b.z, b.w}, oo[4] = {o.x, o.y, o.z, o.w};
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float zz = dz[nt][r] + bb[r];
            float dv, hh;
            if (DROP) {    // as the three-kernel step (neumf.hip): dz = g w_h keep, h1 = relu(z) keep
              const float kp = ((kb >> (4 * nt + r)) & 1u) ? a.keep_scale : 0.f;
              dv = zz > 0.f ? gc * oo[r] * kp : 0.f;
              hh = fmaxf(zz, 0.f) * kp;
            } else {
              dv = zz > 0.f ? gc * oo[r] : 0.f;
              hh = fmaxf(zz, 0.f);
            }
            dz[nt][r] = dv;
            dzs[nt][r] += dv;
            dwh[nt][r] = fmaf(gc, hh, dwh[nt][r]);
          }
          *reinterpret_cast<float4*>(Tz + i * SZ + 16 * nt + 4 * g) = make_float4(dz[nt][0], dz[nt][1], dz[nt][2], dz[nt][3]);
        }
#pragma unroll
        for (int cc = 0; cc < NCU; ++cc)
          *reinterpret_cast<float4*>(Th + i * SH + 16 * cc + 4 * g) = make_float4(hx[cc][0], hx[cc][1], hx[cc][2], hx[cc][3]);

        // d mlp_i = W1i^T dz in the lane layout of the gathered row: update in place, or hand the gradient row to the plan's update.
        // SGD: ONE store per slice either way (address and value selected per lane, no divergent branch)
        const bool single = mflag == 0;
        float* grow = a.g_mlp_i + n * a.ld_gi + 4 * g;
#pragma unroll
        for (int kt0 = 0; kt0 < NCU; kt0 += KG) {
          f32x4s acc[KG];
#pragma unroll
          for (int k = 0; k < KG; ++k) acc[k] = f32x4s{0.f, 0.f, 0.f, 0.f};
          // Adam / Adagrad: the optimizer state of the slices this group updates is requested BEFORE the group's 64 MFMAs and
          // consumed after them (opt_row4 at the point of use waited a full memory latency per slice with nothing beside it)
          float4 pm[KG], pv[KG];
          const bool upd = valid && single;
          if (MODE != MODE_SGD) {
#pragma unroll
            for (int k = 0; k < KG; ++k) {
              pm[k] = pv[k] = make_float4(0.f, 0.f, 0.f, 0.f);
              const size_t idx4 = (size_t)(item_c * D + 16 * (kt0 + k) + 4 * g) / 4;
              if (upd && mode_has_m(MODE)) pm[k] = load_stream4(reinterpret_cast<const float4*>(a.m_mlp_i) + idx4);
              if (upd && mode_has_v(MODE)) pv[k] = load_stream4(reinterpret_cast<const float4*>(a.v_mlp_i) + idx4);
            }
          }
          back_tiles<NT, KG, SW>(acc, wbwd_i + 16 * kt0, dz);
          if (valid) {
#pragma unroll
            for (int k = 0; k < KG; ++k) {
              const int kt = kt0 + k;
              const float4 gr = make_float4(acc[k][0], acc[k][1], acc[k][2], acc[k][3]);
              const float4 w0 = make_float4(hx[kt][0], hx[kt][1], hx[kt][2], hx[kt][3]);
              if (MODE == MODE_SGD) {
                float4 w1 = w0, m0 = w0, v0 = w0;
                opt_apply4<MODE_SGD>(a.opt, w1, m0, v0, gr);
                float* dst = single ? (a.mlp_i + item_c * a.ld_i + 16 * kt + 4 * g) : (grow + 16 * kt);
                store_row4(reinterpret_cast<float4*>(dst), single ? w1 : gr);
              } else if (single) {
                const size_t idx4 = (size_t)(item_c * D + 16 * kt + 4 * g) / 4;
                float4 w1 = w0;
                opt_apply4<MODE>(a.opt, w1, pm[k], pv[k], gr);
                store_row4(reinterpret_cast<float4*>(a.mlp_i) + idx4, w1);
                if (mode_has_m(MODE)) store_row4(reinterpret_cast<float4*>(a.m_mlp_i) + idx4, pm[k]);
                if (mode_has_v(MODE)) store_row4(reinterpret_cast<float4*>(a.v_mlp_i) + idx4, pv[k]);
              } else {
                *reinterpret_cast<float4*>(grow + 16 * kt) = gr;
              }
            }
          }
        }
        // dW1i += dz^T hi over the 16 candidates of this step (K step s: candidates 4 s + g'); operands of s + 1 in flight
        __builtin_amdgcn_wave_barrier();
        {
          float av[NT], bv[NCU], an[NT], bn[NCU];
#pragma unroll
          for (int ft = 0; ft < NT; ++ft) av[ft] = Tz[g * SZ + 16 * ft + i];
#pragma unroll
          for (int kt = 0; kt < NCU; ++kt) bv[kt] = Th[g * SH + 16 * kt + i];
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            if (s + 1 < 4) {
#pragma unroll
              for (int ft = 0; ft < NT; ++ft) an[ft] = Tz[(4 * (s + 1) + g) * SZ + 16 * ft + i];
#pragma unroll
              for (int kt = 0; kt < NCU; ++kt) bn[kt] = Th[(4 * (s + 1) + g) * SH + 16 * kt + i];
            }
#pragma unroll
            for (int ft = 0; ft < NT; ++ft)
#pragma unroll
              for (int kt = 0; kt < NCU; ++kt) accW[ft][kt] = mma16(av[ft], bv[kt], accW[ft][kt]);
            if (s + 1 < 4) {
#pragma unroll
              for (int ft = 0; ft < NT; ++ft) av[ft] = an[ft];
#pragma unroll
              for (int kt = 0; kt < NCU; ++kt) bv[kt] = bn[kt];
            }
          }
        }
        __builtin_amdgcn_wave_barrier();
      }
    }

    // ---- per tuple: mlp_u gradient row, per-wave sums of db1 / dw_h, the strips of the dW1u exchange -------------------------
    {
#pragma unroll
      for (int kt0 = 0; kt0 < NCU; kt0 += KG) {
        f32x4s acc[KG];
#pragma unroll
        for (int k = 0; k < KG; ++k) acc[k] = f32x4s{0.f, 0.f, 0.f, 0.f};
        back_tiles<NT, KG, SW>(acc, wbwd_u + 16 * kt0, dzs);
        if (valid) {
#pragma unroll
          for (int k = 0; k < KG; ++k)
            *reinterpret_cast<float4*>(a.gu_mlp + tup * a.ld_gu + 16 * (kt0 + k) + 4 * g) = make_float4(acc[k][0], acc[k][1], acc[k][2], acc[k][3]);
        }
      }
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float x = row_allreduce_sum<16>(dzs[nt][r]);
          const float y = row_allreduce_sum<16>(dwh[nt][r]);
          if (i == 0) {
            wr[16 * nt + 4 * g + r] += x;
            wr[L1 + 16 * nt + 4 * g + r] += y;
          }
        }
      float hu[NCU][4];
      load_row_slices<NCU>(hu, hup);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        *reinterpret_cast<float4*>(Tz + i * SZ + 16 * nt + 4 * g) = make_float4(dzs[nt][0], dzs[nt][1], dzs[nt][2], dzs[nt][3]);
#pragma unroll
      for (int cc = 0; cc < NCU; ++cc)
        *reinterpret_cast<float4*>(Th + i * SH + 16 * cc + 4 * g) = make_float4(hu[cc][0], hu[cc][1], hu[cc][2], hu[cc][3]);
    }

    // ---- pass 3: backward of the GMF branch (no MFMA: d mf_i = g w_mf mf_u, d mf_u = w_mf sum_c g_c mf_i) and the mf_i row
    // updates -- kept out of pass 2, whose MFMA pipelines need the registers these rows would occupy.  (65 us of the kernel at
    // the config-4 shape: 168 MB read + 168 MB written by one wave per SIMD.  Requesting the rows of four candidates together
    // and running the mlp rows of pass 1 two candidates ahead measured SLOWER on the same box -- 0.573 against 0.553 ms per
    // step, profiles/r05a_ab_neumf_step.txt; plain instead of non-temporal row stores: 0.655 ms.) -------------------------------
    {
      float mu[NCU][4], S[NCU][4], mn[NCU][4];
      load_row_slices<NCU>(mu, mup);
#pragma unroll
      for (int cc = 0; cc < NCU; ++cc)
#pragma unroll
        for (int e = 0; e < 4; ++e) S[cc][e] = 0.f;
      int64_t it0 = ip[0], it1 = ip[C > 1 ? 1 : 0];
      load_row_slices<NCU>(mn, a.mf_i + it0 * a.ld_i + 4 * g);
      // the singleton flag of a candidate travels one candidate ahead, like its row: this pass has no MFMA group to cover a load
      // that the very next store's address depends on
      uint8_t mflag_next = a.multi ? a.multi[it0] : (uint8_t)1;
      for (int c = 0; c < C; ++c) {
        float mx[NCU][4];
#pragma unroll
        for (int cc = 0; cc < NCU; ++cc)
#pragma unroll
          for (int e = 0; e < 4; ++e) mx[cc][e] = mn[cc][e];
        const int64_t item_c = it0;
#ifdef RC_NEUMF_FLAG_LATE     // (experiment switch: the flag requested where it is used, as before)
        const uint8_t mflag = a.multi ? a.multi[item_c] : (uint8_t)1;
#else
        const uint8_t mflag = mflag_next;
#endif
        it0 = it1;
        if (c + 1 < C) {
          load_row_slices<NCU>(mn, a.mf_i + it0 * a.ld_i + 4 * g);
#ifndef RC_NEUMF_FLAG_LATE
          mflag_next = a.multi ? a.multi[it0] : (uint8_t)1;
#endif
          it1 = ip[c + 2 < C ? c + 2 : C - 1];
        }
        const float gc = sp[c * 16 + i];
        const bool single = mflag == 0;
        float* grow = a.g_mf_i + (tup * C + c) * a.ld_gi + 4 * g;
        // Adam / Adagrad: the state slices of this candidate's mf_i row, requested together before the arithmetic
        float4 qm[MODE == MODE_SGD ? 1 : NCU], qv[MODE == MODE_SGD ? 1 : NCU];
        if (MODE != MODE_SGD) {
#pragma unroll
          for (int cc = 0; cc < NCU; ++cc) {
            qm[cc] = qv[cc] = make_float4(0.f, 0.f, 0.f, 0.f);
            const size_t idx4 = (size_t)(item_c * D + 16 * cc + 4 * g) / 4;
            if (valid && single && mode_has_m(MODE)) qm[cc] = load_stream4(reinterpret_cast<const float4*>(a.m_mf_i) + idx4);
            if (valid && single && mode_has_v(MODE)) qv[cc] = load_stream4(reinterpret_cast<const float4*>(a.v_mf_i) + idx4);
          }
        }
#pragma unroll
        for (int cc = 0; cc < NCU; ++cc) {
          const float4 w = *reinterpret_cast<const float4*>(swo + 16 * cc + 4 * g);
          const float4 gr = make_float4(gc * (w.x * mu[cc][0]), gc * (w.y * mu[cc][1]), gc * (w.z * mu[cc][2]), gc * (w.w * mu[cc][3]));
#pragma unroll
          for (int e = 0; e < 4; ++e) S[cc][e] = fmaf(gc, mx[cc][e], S[cc][e]);
          if (valid) {
            const float4 w0 = make_float4(mx[cc][0], mx[cc][1], mx[cc][2], mx[cc][3]);
            if (MODE == MODE_SGD) {
              float4 w1 = w0, m0 = w0, v0 = w0;
              opt_apply4<MODE_SGD>(a.opt, w1, m0, v0, gr);
              float* dst = single ? (a.mf_i + item_c * a.ld_i + 16 * cc + 4 * g) : (grow + 16 * cc);
              store_row4(reinterpret_cast<float4*>(dst), single ? w1 : gr);
            } else if (single) {
              const size_t idx4 = (size_t)(item_c * D + 16 * cc + 4 * g) / 4;
              float4 w1 = w0;
              opt_apply4<MODE>(a.opt, w1, qm[MODE == MODE_SGD ? 0 : cc], qv[MODE == MODE_SGD ? 0 : cc], gr);
              store_row4(reinterpret_cast<float4*>(a.mf_i) + idx4, w1);
              if (mode_has_m(MODE)) store_row4(reinterpret_cast<float4*>(a.m_mf_i) + idx4, qm[MODE == MODE_SGD ? 0 : cc]);
              if (mode_has_v(MODE)) store_row4(reinterpret_cast<float4*>(a.v_mf_i) + idx4, qv[MODE == MODE_SGD ? 0 : cc]);
            } else {
              *reinterpret_cast<float4*>(grow + 16 * cc) = gr;
            }
          }
        }
      }
#pragma unroll
      for (int cc = 0; cc < NCU; ++cc) {
        const float4 w = *reinterpret_cast<const float4*>(swo + 16 * cc + 4 * g);
        if (valid)
          *reinterpret_cast<float4*>(a.gu_mf + tup * a.ld_gu + 16 * cc + 4 * g) =
              make_float4(w.x * S[cc][0], w.y * S[cc][1], w.z * S[cc][2], w.w * S[cc][3]);
#pragma unroll
        for (int e = 0; e < 4; ++e) {   // dw_mf[k] += sum over the wave's tuples of mf_u[k] S[k]
          const float t = row_allreduce_sum<16>(mu[cc][e] * S[cc][e]);
          if (i == 0) wr[2 * L1 + 16 * cc + 4 * g + e] += t;
        }
      }
    }

    // ---- dW1u += (sum_c dz_c)^T mlp_u over the workgroup's 64 tuples: strips of all four waves, a quarter of the tiles each ----
    __syncthreads();
    for (int w2 = 0; w2 < 4; ++w2) {
      const float* Tz2 = tb + w2 * per_wave;
      const float* Th2 = Tz2 + 16 * SZ;
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int q = 0; q < NTU; ++q) {
          const int tile = wave + 4 * q, ft = tile % NT, kt = tile / NT;
          accU[q] = mma16(Tz2[(4 * s + g) * SZ + 16 * ft + i], Th2[(4 * s + g) * SH + 16 * kt + i], accU[q]);
        }
    }
    __syncthreads();
  }

  // ---- per-workgroup partials of the dense gradients --------------------------------------------------------------
  const size_t wg = blockIdx.x;
  float* pw = a.pW1 + wg * (size_t)(L1 * K0);
#pragma unroll
  for (int q = 0; q < NTU; ++q) {
    const int tile = wave + 4 * q, ft = tile % NT, kt = tile / NT;
#pragma unroll
    for (int r = 0; r < 4; ++r) pw[(size_t)(16 * ft + 4 * g + r) * K0 + 16 * kt + i] = accU[q][r];
  }
  // the item half: the four waves' accumulators summed in wave order through LDS (W1 is not needed any more)
  float* red = Ws;   // [L1][D]
  for (int w2 = 0; w2 < 4; ++w2) {
    if (wave == w2) {
#pragma unroll
      for (int ft = 0; ft < NT; ++ft)
#pragma unroll
        for (int kt = 0; kt < NCU; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float* p = red + (16 * ft + 4 * g + r) * D + 16 * kt + i;
            *p = (w2 == 0 ? 0.f : *p) + accW[ft][kt][r];
          }
    }
    __syncthreads();
  }
  for (int k = threadIdx.x; k < L1 * D; k += 256) pw[(size_t)(k / D) * K0 + D + (k % D)] = red[k];
  for (int k = threadIdx.x; k < 2 * L1 + D; k += 256) {
    const float t = ((wred[k] + wred[(2 * L1 + D) + k]) + wred[2 * (2 * L1 + D) + k]) + wred[3 * (2 * L1 + D) + k];
    if (k < L1) a.pb1[wg * L1 + k] = t;
    else if (k < 2 * L1) a.pwout[wg * (D + L1) + D + (k - L1)] = t;
    else a.pwout[wg * (D + L1) + (k - 2 * L1)] = t;
  }
}

// ---- singleton / multi-occurrence classification of the batch's item ids ------------------------------------------------
// Plain stores only (the first version used atomicOr on two bitmaps: 60 us, the Zipf head's ~4 K occurrences of ONE id
// serialise on one L2 address).  Pass 1: owner[id] = batch position -- whichever occurrence lands last owns the row.  Pass 2:
// every occurrence that does not own its row marks it, multi[id] = 1.  A row with one occurrence has no loser; a row with k >= 2
// has k - 1 of them, whoever won: the FLAGS do not depend on the schedule.  owner needs no initial value (read only where this
// batch wrote it); multi is all zero between steps (neumf_unmark_kernel clears what the batch set).
// Ids outside [0, n_items) touch nothing here (nn.Embedding raises on them in the reference; the plan's status word reports them).
__global__ __launch_bounds__(256) void neumf_mark_owner_kernel(const int64_t* __restrict__ ids, int64_t n, uint64_t n_items,
                                                               uint32_t* __restrict__ owner) {
  const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (k >= n) return;
  const int64_t id = ids[k];
  if ((uint64_t)id < n_items) owner[id] = (uint32_t)k;
}

__global__ __launch_bounds__(256) void neumf_mark_multi_kernel(const int64_t* __restrict__ ids, int64_t n, uint64_t n_items,
                                                               const uint32_t* __restrict__ owner, uint8_t* __restrict__ multi) {
  const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (k >= n) return;
  const int64_t id = ids[k];
  if ((uint64_t)id < n_items && owner[id] != (uint32_t)k) multi[id] = 1;
}

__global__ __launch_bounds__(256) void neumf_unmark_kernel(const int64_t* __restrict__ ids, int64_t n, uint64_t n_items,
                                                           uint8_t* __restrict__ multi) {
  const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (k >= n) return;
  const int64_t id = ids[k];
  if ((uint64_t)id < n_items) multi[id] = 0;
}

// (defined in neumf.hip) out[i] = sum_w p[w][i] for the three partial arrays, fixed order
int neumf_reduce_partials(const float* pW1, const float* pb1, const float* pwout, float* dW1, float* db1, float* dw_out, int cW, int cb,
                          int co, int n_wg, hipStream_t s);

static size_t step_lds_bytes(int d, int l1, int C) {
  const size_t k0 = 2 * (size_t)d;
  const size_t fixed = (size_t)l1 * (k0 + 4) + l1 + (d + l1) + 4 * (2 * (size_t)l1 + d);
  const size_t per_wave = 16 * ((size_t)l1 + 16) + 16 * ((size_t)d + 16) + 16 * (size_t)C;
  return sizeof(float) * (fixed + 4 * per_wave);
}

static bool step_shape(int d, int l1) {
  if (!(d == 32 || d == 64 || d == 128)) return false;
  if (l1 == 32 || l1 == 64) return true;
  // (hidden 128: eight MFMA chains per product keep 64 operand registers in flight -- built, 64 to 900 bytes of scratch per lane
  //  in every instantiation: those towers stay on the three-kernel step)
  if (l1 == 16) return d != 32;      // (the dW1u tiles are dealt to four waves: 16 (d / 16) / 16 tiles must be a multiple of 4)
  return false;
}

int device_cus();   // bucket_plan.hip

static int step_grid(int B) {
  const int64_t rounds = ((int64_t)B + 63) / 64;
  static const int per_cu = [] {
    const char* v = getenv("RC_NEUMF_STEP_WGS_PER_CU");   // workgroups per CU over the launch (A/B: partial size vs balance)
    const int k = (v && *v) ? atoi(v) : 1;
    return k < 1 ? 1 : k;
  }();
  int64_t grid = (int64_t)device_cus() * per_cu;
  if (grid > rounds) grid = rounds;
  return (int)(grid < 1 ? 1 : grid);
}

template <int D, int L1, int MODE, bool DROP>
static int launch_step(const NeumfStepArgs& a, int grid, hipStream_t s) {
  const size_t lds_bytes = step_lds_bytes(D, L1, a.C);
  auto kern = neumf_step_kernel<D, L1, MODE, DROP>;
  RC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds_bytes, s, a);
  RC_LAUNCH_CHECK();
  return RC_OK;
}

template <int MODE>
static int dispatch_step(const NeumfStepArgs& a, int d, int l1, int grid, hipStream_t s) {
#define RC_NS(D_, L_) \
  if (d == D_ && l1 == L_) return a.seed_dev ? launch_step<D_, L_, MODE, true>(a, grid, s) : launch_step<D_, L_, MODE, false>(a, grid, s)
  RC_NS(128, 64); RC_NS(128, 32); RC_NS(128, 16);
  RC_NS(64, 64); RC_NS(64, 32); RC_NS(64, 16);
  RC_NS(32, 64); RC_NS(32, 32);
#undef RC_NS
  return fail(RC_ERR_UNSUPPORTED, "rc_neumf_train_step: no kernel for d=%d, hidden=%d", d, l1);
}

}  // namespace rc

using namespace rc;

extern "C" int rc_neumf_train_step_supported(int C, int d, int l1) {
  return (step_shape(d, l1) && C >= 2 && step_lds_bytes(d, l1, C) <= 160 * 1024) ? 1 : 0;
}

extern "C" size_t rc_neumf_train_step_workspace_bytes(int B, int C, int d, int l1) {
  if (!rc_neumf_train_step_supported(C, d, l1) || B < 1) return 0;
  const size_t per = (size_t)l1 * 2 * d + l1 + (d + l1);
  return align_up((size_t)step_grid(B) * per * sizeof(float), 256) + 256;
}

extern "C" size_t rc_neumf_train_step_marks_bytes(int64_t n_items) {
  return n_items < 1 ? 0 : align_up((size_t)n_items * sizeof(uint32_t), 256) + align_up((size_t)n_items, 256);
}

// the two marking passes / the clearing pass on their own: a caller that knows the NEXT batch (BaseRunner.fit does) marks it on a
// second stream while this step's updates run and hands the prepared buffer to rc_neumf_train_step_marked
extern "C" int rc_neumf_mark_rows(const int64_t* iid, int64_t n, int64_t n_items, void* marks, rc_stream_t stream) {
  if (n == 0) return RC_OK;
  RC_REQUIRE(iid && marks && n > 0 && n < ((int64_t)1 << 32) && n_items >= 1, "rc_neumf_mark_rows: bad arguments");
  uint32_t* owner = reinterpret_cast<uint32_t*>(marks);
  uint8_t* multi = reinterpret_cast<uint8_t*>(marks) + align_up((size_t)n_items * sizeof(uint32_t), 256);
  const unsigned blocks = (unsigned)((n + 255) / 256);
  hipLaunchKernelGGL(neumf_mark_owner_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), iid, n, (uint64_t)n_items, owner);
  hipLaunchKernelGGL(neumf_mark_multi_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), iid, n, (uint64_t)n_items, owner, multi);
  RC_LAUNCH_CHECK();
  return RC_OK;
}

extern "C" int rc_neumf_unmark_rows(const int64_t* iid, int64_t n, int64_t n_items, void* marks, rc_stream_t stream) {
  if (n == 0) return RC_OK;
  RC_REQUIRE(iid && marks && n > 0 && n_items >= 1, "rc_neumf_unmark_rows: bad arguments");
  uint8_t* multi = reinterpret_cast<uint8_t*>(marks) + align_up((size_t)n_items * sizeof(uint32_t), 256);
  hipLaunchKernelGGL(neumf_unmark_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream), iid, n, (uint64_t)n_items, multi);
  RC_LAUNCH_CHECK();
  return RC_OK;
}

static int neumf_train_step_impl(bool marked, float drop_p, const uint64_t* seed_dev, float* mf_u, float* mf_i, float* mlp_u, float* mlp_i, float* m_mf_i, float* v_mf_i,
                                 float* m_mlp_i, float* v_mlp_i, const float* W1, const float* b1, const float* w_out,
                                 const int64_t* uid, const int64_t* iid, int B, int C, int d, int l1, int64_t n_items,
                                 void* marks, const rc_opt_hyper* h, float inv_b, float* loss_vec, float* pred,
                                 float* g_mf_i, float* g_mlp_i, float* gu_mf, float* gu_mlp, float* dW1, float* db1,
                                 float* dw_out, void* ws, size_t ws_bytes, rc_stream_t stream) {
  if (B == 0) return RC_OK;
  RC_REQUIRE(mf_u && mf_i && mlp_u && mlp_i && W1 && b1 && w_out && uid && iid && marks && loss_vec && g_mf_i && g_mlp_i && gu_mf &&
                 gu_mlp && dW1 && db1 && dw_out && ws,
             "rc_neumf_train_step: null pointer");
  RC_REQUIRE(B > 0 && C >= 2 && n_items >= 1, "rc_neumf_train_step: bad shape B=%d C=%d n_items=%lld", B, C, (long long)n_items);
  if (!rc_neumf_train_step_supported(C, d, l1))
    return fail(RC_ERR_UNSUPPORTED, "rc_neumf_train_step: d=%d hidden=%d C=%d not supported (d in {32,64,128}, hidden in {16,32,64}; hidden 16 needs d >= 64; LDS <= 160 KB)",
                d, l1, C);
  if (ws_bytes < rc_neumf_train_step_workspace_bytes(B, C, d, l1))
    return fail(RC_ERR_WORKSPACE, "rc_neumf_train_step: workspace %zu < %zu", ws_bytes, rc_neumf_train_step_workspace_bytes(B, C, d, l1));
  NeumfStepArgs a;
  memset(&a, 0, sizeof(a));
  RC_TRY(fill_opt_scalars(h, &a.opt));
  if (!(drop_p >= 0.f && drop_p < 1.f)) return fail(RC_ERR_INVALID_ARG, "rc_neumf_train_step: dropout p=%g outside [0, 1)", (double)drop_p);
  if (drop_p > 0.f) {
    if (!seed_dev) return fail(RC_ERR_INVALID_ARG, "rc_neumf_train_step: dropout p=%g needs a device seed", (double)drop_p);
    a.seed_dev = seed_dev;      // (as rc_neumf_fwd_dropout)
    a.drop_thresh = (uint32_t)((double)drop_p * 4294967296.0);
    a.keep_scale = 1.0f / (1.0f - drop_p);
  }
  const int mode = mode_of(h);
  RC_REQUIRE(!mode_has_m(mode) || (m_mf_i && m_mlp_i), "rc_neumf_train_step: optimizer %d needs the m state of the item tables", h->opt);
  RC_REQUIRE(!mode_has_v(mode) || (v_mf_i && v_mlp_i), "rc_neumf_train_step: optimizer %d needs the v state of the item tables", h->opt);
  hipStream_t s = as_stream(stream);
  const int64_t n = (int64_t)B * C;
  RC_REQUIRE(n < ((int64_t)1 << 32), "rc_neumf_train_step: B C = %lld positions do not fit the 32-bit owner map", (long long)n);
  uint32_t* owner = reinterpret_cast<uint32_t*>(marks);
  uint8_t* multi = reinterpret_cast<uint8_t*>(marks) + align_up((size_t)n_items * sizeof(uint32_t), 256);
  const unsigned mark_blocks = (unsigned)((n + 255) / 256);
  if (!marked) {
    hipLaunchKernelGGL(neumf_mark_owner_kernel, dim3(mark_blocks), dim3(256), 0, s, iid, n, (uint64_t)n_items, owner);
    hipLaunchKernelGGL(neumf_mark_multi_kernel, dim3(mark_blocks), dim3(256), 0, s, iid, n, (uint64_t)n_items, owner, multi);
    RC_LAUNCH_CHECK();
  }
  a.mf_u = mf_u; a.mf_i = mf_i; a.mlp_u = mlp_u; a.mlp_i = mlp_i;
  a.ld_u = a.ld_i = a.ld_gi = a.ld_gu = d;
  a.m_mf_i = m_mf_i; a.v_mf_i = v_mf_i; a.m_mlp_i = m_mlp_i; a.v_mlp_i = v_mlp_i;
  a.W1 = W1; a.b1 = b1; a.w_out = w_out; a.uid = uid; a.iid = iid; a.B = B; a.C = C; a.inv_b = inv_b;
  a.multi = multi; a.loss_vec = loss_vec; a.pred = pred; a.g_mf_i = g_mf_i; a.g_mlp_i = g_mlp_i; a.gu_mf = gu_mf; a.gu_mlp = gu_mlp;
  const int grid = step_grid(B);
  const int cW = l1 * 2 * d, cb = l1, co = d + l1;
  float* p = reinterpret_cast<float*>(ws);
  a.pW1 = p;
  a.pb1 = p + (size_t)grid * cW;
  a.pwout = a.pb1 + (size_t)grid * cb;
  int rc = RC_OK;
  if (mode == MODE_SGD) rc = dispatch_step<MODE_SGD>(a, d, l1, grid, s);
  else if (mode == MODE_ADAM) rc = dispatch_step<MODE_ADAM>(a, d, l1, grid, s);
  else rc = dispatch_step<MODE_ADAGRAD>(a, d, l1, grid, s);
  // the flags go back to zero whatever happened to the step (prepared marks: the caller clears them, rc_neumf_unmark_rows)
  if (!marked) hipLaunchKernelGGL(neumf_unmark_kernel, dim3(mark_blocks), dim3(256), 0, s, iid, n, (uint64_t)n_items, multi);
  RC_TRY(rc);
  RC_LAUNCH_CHECK();
  return neumf_reduce_partials(a.pW1, a.pb1, a.pwout, dW1, db1, dw_out, cW, cb, co, grid, s);
}

extern "C" int rc_neumf_train_step(float* mf_u, float* mf_i, float* mlp_u, float* mlp_i, float* m_mf_i, float* v_mf_i,
                                   float* m_mlp_i, float* v_mlp_i, const float* W1, const float* b1, const float* w_out,
                                   const int64_t* uid, const int64_t* iid, int B, int C, int d, int l1, int64_t n_items,
                                   void* marks, const rc_opt_hyper* h, float inv_b, float* loss_vec, float* pred,
                                   float* g_mf_i, float* g_mlp_i, float* gu_mf, float* gu_mlp, float* dW1, float* db1,
                                   float* dw_out, void* ws, size_t ws_bytes, rc_stream_t stream) {
  return neumf_train_step_impl(false, 0.f, nullptr, mf_u, mf_i, mlp_u, mlp_i, m_mf_i, v_mf_i, m_mlp_i, v_mlp_i, W1, b1, w_out, uid, iid, B, C, d, l1,
                               n_items, marks, h, inv_b, loss_vec, pred, g_mf_i, g_mlp_i, gu_mf, gu_mlp, dW1, db1, dw_out, ws, ws_bytes, stream);
}

extern "C" int rc_neumf_train_step_marked(float* mf_u, float* mf_i, float* mlp_u, float* mlp_i, float* m_mf_i, float* v_mf_i,
                                          float* m_mlp_i, float* v_mlp_i, const float* W1, const float* b1, const float* w_out,
                                          const int64_t* uid, const int64_t* iid, int B, int C, int d, int l1, int64_t n_items,
                                          void* marks, const rc_opt_hyper* h, float inv_b, float* loss_vec, float* pred,
                                          float* g_mf_i, float* g_mlp_i, float* gu_mf, float* gu_mlp, float* dW1, float* db1,
                                          float* dw_out, void* ws, size_t ws_bytes, rc_stream_t stream) {
  return neumf_train_step_impl(true, 0.f, nullptr, mf_u, mf_i, mlp_u, mlp_i, m_mf_i, v_mf_i, m_mlp_i, v_mlp_i, W1, b1, w_out, uid, iid, B, C, d, l1,
                               n_items, marks, h, inv_b, loss_vec, pred, g_mf_i, g_mlp_i, gu_mf, gu_mlp, dW1, db1, dw_out, ws, ws_bytes, stream);
}

extern "C" int rc_neumf_train_step_dropout(float* mf_u, float* mf_i, float* mlp_u, float* mlp_i, float* m_mf_i, float* v_mf_i,
                                           float* m_mlp_i, float* v_mlp_i, const float* W1, const float* b1, const float* w_out,
                                           const int64_t* uid, const int64_t* iid, int B, int C, int d, int l1, int64_t n_items,
                                           void* marks, int marks_prepared, const rc_opt_hyper* h, float inv_b, float drop_p,
                                           const uint64_t* seed_dev, float* loss_vec, float* pred, float* g_mf_i, float* g_mlp_i,
                                           float* gu_mf, float* gu_mlp, float* dW1, float* db1, float* dw_out, void* ws,
                                           size_t ws_bytes, rc_stream_t stream) {
  return neumf_train_step_impl(marks_prepared != 0, drop_p, seed_dev, mf_u, mf_i, mlp_u, mlp_i, m_mf_i, v_mf_i, m_mlp_i, v_mlp_i, W1, b1, w_out,
                               uid, iid, B, C, d, l1, n_items, marks, h, inv_b, loss_vec, pred, g_mf_i, g_mlp_i, gu_mf, gu_mlp, dW1, db1,
                               dw_out, ws, ws_bytes, stream);
}

extern "C" int rc_neumf_head_fwd_bwd(const float* mf_u, const float* mlp_u, int64_t ld_u, const float* mf_i, const float* mlp_i,
                                     int64_t ld_i, const float* W1, const float* b1, const float* w_out, const int64_t* uid,
                                     const int64_t* iid, int B, int C, int d, int l1, float inv_b, float* loss_vec, float* pred,
                                     float* g_mf_i, float* g_mlp_i, int64_t ld_gi, float* gu_mf, float* gu_mlp, int64_t ld_gu,
                                     float* dW1, float* db1, float* dw_out, void* ws, size_t ws_bytes, rc_stream_t stream) {
  if (B == 0) return RC_OK;
  RC_REQUIRE(mf_u && mf_i && mlp_u && mlp_i && W1 && b1 && w_out && uid && iid && loss_vec && g_mf_i && g_mlp_i && gu_mf && gu_mlp && dW1 &&
                 db1 && dw_out && ws,
             "rc_neumf_head_fwd_bwd: null pointer");
  RC_REQUIRE(B > 0 && C >= 2, "rc_neumf_head_fwd_bwd: bad shape B=%d C=%d", B, C);
  RC_REQUIRE(ld_u >= d && ld_i >= d && ld_gi >= d && ld_gu >= d && ld_u % 4 == 0 && ld_i % 4 == 0 && ld_gi % 4 == 0 && ld_gu % 4 == 0,
             "rc_neumf_head_fwd_bwd: row strides must be multiples of 4 floats and at least d");
  if (!rc_neumf_train_step_supported(C, d, l1))
    return fail(RC_ERR_UNSUPPORTED, "rc_neumf_head_fwd_bwd: d=%d hidden=%d C=%d not supported (see rc_neumf_train_step_supported)", d, l1, C);
  if (ws_bytes < rc_neumf_train_step_workspace_bytes(B, C, d, l1))
    return fail(RC_ERR_WORKSPACE, "rc_neumf_head_fwd_bwd: workspace %zu < %zu", ws_bytes, rc_neumf_train_step_workspace_bytes(B, C, d, l1));
  NeumfStepArgs a;
  memset(&a, 0, sizeof(a));
  a.opt.neg_lr = 0.f;   // no row is updated: every position writes its gradient row (multi == nullptr)
  a.mf_u = const_cast<float*>(mf_u); a.mf_i = const_cast<float*>(mf_i); a.mlp_u = const_cast<float*>(mlp_u); a.mlp_i = const_cast<float*>(mlp_i);
  a.ld_u = ld_u; a.ld_i = ld_i; a.ld_gi = ld_gi; a.ld_gu = ld_gu;
  a.W1 = W1; a.b1 = b1; a.w_out = w_out; a.uid = uid; a.iid = iid; a.B = B; a.C = C; a.inv_b = inv_b;
  a.multi = nullptr; a.loss_vec = loss_vec; a.pred = pred; a.g_mf_i = g_mf_i; a.g_mlp_i = g_mlp_i; a.gu_mf = gu_mf; a.gu_mlp = gu_mlp;
  hipStream_t s = as_stream(stream);
  const int grid = step_grid(B);
  const int cW = l1 * 2 * d, cb = l1, co = d + l1;
  float* p = reinterpret_cast<float*>(ws);
  a.pW1 = p;
  a.pb1 = p + (size_t)grid * cW;
  a.pwout = a.pb1 + (size_t)grid * cb;
  RC_TRY(dispatch_step<MODE_SGD>(a, d, l1, grid, s));
  return neumf_reduce_partials(a.pW1, a.pb1, a.pwout, dW1, db1, dw_out, cW, cb, co, grid, s);
}
