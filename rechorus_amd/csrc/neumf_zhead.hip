// neumf_zhead.hip -- the NeuMF head of a row-sharded step whose ITEM half of the hidden layer was computed by the rows' owners.
//
// Reference: models/general/NeuMF.py:61-75.  NeuMF tiles the user ids over the candidates (:61), so the hidden layer splits:
//     h = relu(W1 [mlp_u ; mlp_i] + b1) = relu(W1u mlp_u + W1i mlp_i + b1).
// In the sharded step (rechorus_amd/sharded.py, ShardedNeumf) the item rows live on other ranks.  Round 5 fetched both 128-float
// item rows of every candidate (mf_i | mlp_i: 1,024 B each way per distinct id).  Here the OWNER of an item row forms
// zi = W1i mlp_i (l1 = 64 floats) next to the row and returns (mf_i | zi): 768 B; the gradient that travels back is (d mf_i | dz)
// instead of (d mf_i | d mlp_i), the owner forms d mlp_i = W1i^T dz and its share of dW1i = dz^T mlp_i itself (GEMMs of mlp.hip
// on the rows it served).  -25 % bytes on both exchanges of the config-4 step, and the home rank's head loses its MFMA work on
// the candidates: what is left here is per-candidate ELEMENTWISE work plus the BPR loss --
//     pred_c = sum_k w_mf[k] mf_u[k] mf_i,c[k] + sum_f w_h[f] relu(zu[f] + zi,c[f])       (zu = W1u mlp_u + b1: a GEMM of the caller)
//     g = dL/dpred (models/BaseModel.py:182-185, closed form);   dz_c[f] = g_c w_h[f] [h_c[f] > 0];   d mf_i,c[k] = g_c w_mf[k] mf_u[k]
//     d mf_u[k] = w_mf[k] sum_c g_c mf_i,c[k];   dzu[f] = sum_c dz_c[f];   dw_mf[k] += mf_u[k] sum_c g_c mf_i,c[k];   dw_h[f] += sum_c g_c h_c[f]
// One wave per tuple, lanes over the features, two passes over the candidates (scores, then gradients: the loss needs all C
// scores first); the rows are re-read in pass 2 (a tuple's 5 x 768 B were touched microseconds ago).  HBM-bound: per tuple
// C (d + l1) floats read twice and written once.  dw_out leaves as per-workgroup partials summed in workgroup order.
#include "bpr_math.hpp"
#include "common.hpp"

namespace rc {

constexpr int kZheadMaxC = 256;        // predictions of a tuple kept in an LDS strip
constexpr int kZheadMaxW = 1024;       // d and l1 up to 1,024 (16 floats per lane)
constexpr int kZheadBlocks = 256;      // grid cap = number of dw_out partials (one per thread of the combining workgroup)

struct ZheadArgs {
  const float* mf_u;     // [B] rows of d floats, stride ld_u
  const float* zu;       // [B, l1] = W1u mlp_u + b1
  const float* irows;    // [B * C] rows (mf_i d floats | zi l1 floats), stride ld_i
  const float* w_out;    // [d + l1]: w_mf | w_h
  int64_t ld_u, ld_i, ld_gi, ld_gu;
  int B, C, d, l1;
  float inv_b;
  float* loss_vec;       // [B]
  float* pred;           // [B, C] | null
  float* gi;             // [B * C] rows (d mf_i | dz), stride ld_gi
  float* gu_mf;          // [B] rows of d floats, stride ld_gu
  float* dzu;            // [B, l1]
  float* part;           // [grid][d + l1] partial dw_out
};

// CREG > 0: C <= CREG candidates whose rows (d, l1 <= 128: two + two floats per lane and candidate) stay in registers between the
// passes -- all of a tuple's loads are requested together and nothing is re-read; CREG == 0: any shape, rows re-read in pass 2.
template <int CREG>
__global__ __launch_bounds__(kBlock) void neumf_zhead_kernel(ZheadArgs a) {
  __shared__ float strip[(kBlock / 64) * kZheadMaxC];
  __shared__ float comb[(kBlock / 64) * 2 * kZheadMaxW];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* sp = strip + wave * kZheadMaxC;
  const int d = a.d, l1 = a.l1, C = a.C;
  constexpr int NV = CREG > 0 ? 2 : kZheadMaxW / 64;
  float dwm[NV], dwh[NV];           // this wave's share of dw_mf / dw_h: feature lane + 64 v
#pragma unroll
  for (int v = 0; v < NV; ++v) dwm[v] = dwh[v] = 0.f;
  float wm[NV], wh[NV];             // w_out slices of this lane
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    wm[v] = lane + 64 * v < d ? a.w_out[lane + 64 * v] : 0.f;
    wh[v] = lane + 64 * v < l1 ? a.w_out[d + lane + 64 * v] : 0.f;
  }
  const int64_t n_waves = (int64_t)gridDim.x * (kBlock / 64);
  for (int64_t t = (int64_t)blockIdx.x * (kBlock / 64) + wave; t < a.B; t += n_waves) {
    const float* mu = a.mf_u + t * a.ld_u;
    const float* zu = a.zu + t * l1;
    constexpr int CR = CREG > 0 ? CREG : 1;
    float rm[CR][2], rz[CR][2], um[2], uz[2];
    if (CREG > 0) {
      // ---- every load of the tuple in flight together
#pragma unroll
      for (int v = 0; v < 2; ++v) {
        um[v] = lane + 64 * v < d ? mu[lane + 64 * v] : 0.f;
        uz[v] = lane + 64 * v < l1 ? zu[lane + 64 * v] : 0.f;
      }
#pragma unroll
      for (int c = 0; c < CREG; ++c) {
        const float* row = a.irows + (t * C + (c < C ? c : 0)) * a.ld_i;
#pragma unroll
        for (int v = 0; v < 2; ++v) {
          rm[c][v] = lane + 64 * v < d ? row[lane + 64 * v] : 0.f;
          rz[c][v] = lane + 64 * v < l1 ? row[d + lane + 64 * v] : 0.f;
        }
      }
#pragma unroll
      for (int c = 0; c < CREG; ++c) {
        if (c >= C) break;
        float pp = 0.f;
#pragma unroll
        for (int v = 0; v < 2; ++v) {
          pp = fmaf(wm[v] * um[v], rm[c][v], pp);
          pp = fmaf(wh[v], fmaxf(uz[v] + rz[c][v], 0.f), pp);
        }
        pp = wave_allreduce_sum(pp);
        if (lane == 0) {
          sp[c] = pp;
          if (a.pred) a.pred[t * C + c] = pp;
        }
      }
    } else {
      // ---- pass 1: predictions
      for (int c = 0; c < C; ++c) {
        const float* row = a.irows + (t * C + c) * a.ld_i;
        float pp = 0.f;
        for (int k = lane; k < d; k += 64) pp = fmaf(a.w_out[k] * mu[k], row[k], pp);
        for (int f = lane; f < l1; f += 64) pp = fmaf(a.w_out[d + f], fmaxf(zu[f] + row[d + f], 0.f), pp);
        pp = wave_allreduce_sum(pp);
        if (lane == 0) {
          sp[c] = pp;
          if (a.pred) a.pred[t * C + c] = pp;
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
    // ---- GeneralModel.loss on the C predictions (every lane the same values), dL/dpred back into the strip
    {
      const float pos = sp[0];
      float mx = -INFINITY;
      for (int c = 1; c < C; ++c) mx = fmaxf(mx, sp[c]);
      float se = 0.f;
      for (int c = 1; c < C; ++c) se += expf(sp[c] - mx);
      const float inv_se = 1.0f / se;
      float P = 0.f, A = 0.f;
      for (int c = 1; c < C; ++c) {
        const float x = sp[c];
        const float w = expf(x - mx) * inv_se;
        const float s = sigmoidf_(pos - x);
        P = fmaf(w, s, P);
        A = fmaf(w, s * (1.0f - s), A);
      }
      const BprRow br = bpr_row(P, a.inv_b);
      if (lane == 0) a.loss_vec[t] = br.loss;
      __builtin_amdgcn_wave_barrier();      // every lane has read the scores
      for (int c = 1 + lane; c < C; c += 64) {
        const float x = sp[c];
        const float w = expf(x - mx) * inv_se;
        const float s = sigmoidf_(pos - x);
        sp[c] = br.dLdP * bpr_dP_dneg(w, s, P);
      }
      if (lane == 0) sp[0] = br.dLdP * A;
    }
    __builtin_amdgcn_wave_barrier();
    // ---- pass 2: gradients.  GMF half: columns k = lane + 64 v; hidden half: features f = lane + 64 v
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      if (v * 64 >= d) break;
      const int k = lane + 64 * v;
      if (k < d) {
        const float m = CREG > 0 ? um[v < 2 ? v : 0] : mu[k];
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < (CREG > 0 ? CREG : 1); ++c) {
          if (CREG == 0 || c >= C) break;
          const float g = sp[c];
          s = fmaf(g, rm[c][v < 2 ? v : 0], s);
          a.gi[(t * C + c) * a.ld_gi + k] = g * wm[v] * m;
        }
        if (CREG == 0) {
          for (int c = 0; c < C; ++c) {
            const float g = sp[c];
            s = fmaf(g, a.irows[(t * C + c) * a.ld_i + k], s);
            a.gi[(t * C + c) * a.ld_gi + k] = g * wm[v] * m;
          }
        }
        a.gu_mf[t * a.ld_gu + k] = wm[v] * s;
        dwm[v] = fmaf(m, s, dwm[v]);
      }
    }
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      if (v * 64 >= l1) break;
      const int f = lane + 64 * v;
      if (f < l1) {
        const float z0 = CREG > 0 ? uz[v < 2 ? v : 0] : zu[f];
        float sdz = 0.f, sgh = 0.f;
#pragma unroll
        for (int c = 0; c < (CREG > 0 ? CREG : 1); ++c) {
          if (CREG == 0 || c >= C) break;
          const float g = sp[c];
          const float h = fmaxf(z0 + rz[c][v < 2 ? v : 0], 0.f);
          const float dz = h > 0.f ? g * wh[v] : 0.f;
          a.gi[(t * C + c) * a.ld_gi + d + f] = dz;
          sdz += dz;
          sgh = fmaf(g, h, sgh);
        }
        if (CREG == 0) {
          for (int c = 0; c < C; ++c) {
            const float g = sp[c];
            const float h = fmaxf(z0 + a.irows[(t * C + c) * a.ld_i + d + f], 0.f);
            const float dz = h > 0.f ? g * wh[v] : 0.f;
            a.gi[(t * C + c) * a.ld_gi + d + f] = dz;
            sdz += dz;
            sgh = fmaf(g, h, sgh);
          }
        }
        a.dzu[t * l1 + f] = sdz;
        dwh[v] += sgh;
      }
    }
    __builtin_amdgcn_wave_barrier();        // sp[] is rewritten by the next tuple
  }
  // this workgroup's partial of dw_out: the four waves in wave order
  float* cm = comb + wave * 2 * kZheadMaxW;
#pragma unroll
  for (int v = 0; v < kZheadMaxW / 64; ++v) {
    cm[lane + 64 * v] = v < NV ? dwm[v < NV ? v : 0] : 0.f;
    cm[kZheadMaxW + lane + 64 * v] = v < NV ? dwh[v < NV ? v : 0] : 0.f;
  }
  __syncthreads();
  for (int k = threadIdx.x; k < d + l1; k += kBlock) {
    const int idx = k < d ? k : kZheadMaxW + (k - d);
    float s = comb[idx];
    for (int q = 1; q < kBlock / 64; ++q) s += comb[q * 2 * kZheadMaxW + idx];
    a.part[(size_t)blockIdx.x * (d + l1) + k] = s;
  }
}

// dw_out[k] = sum of the workgroups' partials in workgroup order: one workgroup per column, thread q holds partial q, a fixed tree
// (the first cut walked 1,024 partials per column in one thread: 234 us for 192 columns)
__global__ __launch_bounds__(kBlock) void neumf_zhead_reduce_kernel(const float* __restrict__ part, int blocks, int w, float* __restrict__ out) {
  __shared__ float sm[kBlock];
  const int k = blockIdx.x;
  sm[threadIdx.x] = (int)threadIdx.x < blocks ? part[(size_t)threadIdx.x * w + k] : 0.f;
  __syncthreads();
  for (int off = kBlock / 2; off >= 1; off >>= 1) {
    if ((int)threadIdx.x < off) sm[threadIdx.x] += sm[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[k] = sm[0];
}

}  // namespace rc

using namespace rc;

extern "C" int rc_neumf_zhead_supported(int C, int d, int l1) {
  return (C >= 2 && C <= kZheadMaxC && d >= 1 && d <= kZheadMaxW && l1 >= 1 && l1 <= kZheadMaxW) ? 1 : 0;
}

extern "C" size_t rc_neumf_zhead_workspace_bytes(int d, int l1) {
  if (d < 1) d = 1;
  if (l1 < 1) l1 = 1;
  return align_up((size_t)kZheadBlocks * (size_t)(d + l1) * sizeof(float), 256);
}

extern "C" int rc_neumf_zhead_fwd_bwd(const float* mf_u, int64_t ld_u, const float* zu, const float* irows, int64_t ld_i, const float* w_out,
                                      int B, int C, int d, int l1, float inv_b, float* loss_vec, float* pred, float* gi, int64_t ld_gi,
                                      float* gu_mf, int64_t ld_gu, float* dzu, float* dw_out, void* ws, size_t ws_bytes, rc_stream_t stream) {
  RC_REQUIRE(dw_out != nullptr, "rc_neumf_zhead_fwd_bwd: null pointer");
  hipStream_t s = as_stream(stream);
  if (B == 0) {
    RC_HIP(hipMemsetAsync(dw_out, 0, (size_t)(d + l1) * sizeof(float), s));
    return RC_OK;
  }
  RC_REQUIRE(mf_u && zu && irows && w_out && loss_vec && gi && gu_mf && dzu && ws, "rc_neumf_zhead_fwd_bwd: null pointer");
  if (!rc_neumf_zhead_supported(C, d, l1))
    return fail(RC_ERR_UNSUPPORTED, "rc_neumf_zhead_fwd_bwd: C=%d (2 .. %d), d=%d, hidden %d (<= %d) not covered", C, kZheadMaxC, d, l1, kZheadMaxW);
  RC_REQUIRE(B > 0 && ld_u >= d && ld_i >= d + l1 && ld_gi >= d + l1 && ld_gu >= d, "rc_neumf_zhead_fwd_bwd: bad shape / strides");
  RC_REQUIRE(ws_bytes >= rc_neumf_zhead_workspace_bytes(d, l1), "rc_neumf_zhead_fwd_bwd: workspace %zu < %zu", ws_bytes,
             rc_neumf_zhead_workspace_bytes(d, l1));
  ZheadArgs a;
  memset(&a, 0, sizeof(a));
  a.mf_u = mf_u; a.zu = zu; a.irows = irows; a.w_out = w_out; a.ld_u = ld_u; a.ld_i = ld_i; a.ld_gi = ld_gi; a.ld_gu = ld_gu;
  a.B = B; a.C = C; a.d = d; a.l1 = l1; a.inv_b = inv_b; a.loss_vec = loss_vec; a.pred = pred; a.gi = gi; a.gu_mf = gu_mf; a.dzu = dzu;
  a.part = static_cast<float*>(ws);
  int blocks = (B + kBlock / 64 - 1) / (kBlock / 64);
  if (blocks > kZheadBlocks) blocks = kZheadBlocks;
  if (C <= 8 && d <= 128 && l1 <= 128)
    hipLaunchKernelGGL((neumf_zhead_kernel<8>), dim3((unsigned)blocks), dim3(kBlock), 0, s, a);      // a tuple's rows stay in registers
  else
    hipLaunchKernelGGL((neumf_zhead_kernel<0>), dim3((unsigned)blocks), dim3(kBlock), 0, s, a);
  RC_LAUNCH_CHECK();
  hipLaunchKernelGGL(neumf_zhead_reduce_kernel, dim3((unsigned)(d + l1)), dim3(kBlock), 0, s, a.part, blocks, d + l1, dw_out);
  RC_LAUNCH_CHECK();
  return RC_OK;
}
