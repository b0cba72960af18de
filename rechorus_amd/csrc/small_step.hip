// small_step.hip -- the BPRMF training step for SMALL batches (B (1 + K) + B <= 32,768 row ids, e.g. the reference's
// default --batch_size 256 with 99 negatives, helpers/BaseRunner.py:33) in TWO launches instead of nine:
//
//   launch 1  small_front_kernel (bprmf_fused.hip): 128 workgroups group the ids by row (small_plan.hpp) while the
//             others run the fused gather / dot / loss / backward and snapshot the batch's user rows
//   launch 2  small_update_kernel (here): every touched item row (gradient sum_occ g[o] * Ub[o / C] in ascending
//             position) and user row (sum_b ugrad[b]) read once, updated, written once; the loss mean
//
// The step is launch- and latency-bound at this size (103 us for ~20 dependent launches in round 1), not
// bandwidth-bound, so the design removes dependent launches: no counters to zero, no atomics across workgroups, and the
// user-row snapshot lets item and user rows update in the same launch (the item side never reads U again).
// Same arithmetic as the large-batch step for rows of up to 32 occurrences; hotter rows are summed by a whole wave
// (lane-group g takes occurrences g, g + G, ...; groups combined in a fixed butterfly), still deterministic.
// Reference semantics: helpers/BaseRunner.py:193-206 around models/general/BPRMF.py:34-45, models/BaseModel.py:182-185.
#include <mutex>

#include "opt_math.hpp"
#include "small_plan.hpp"
#include "numeric_grads.hpp"

namespace rc {

int small_front_launch(const float* U, const float* I, const int64_t* uid, const int64_t* iid, int B, int C, int d, float inv_b,
                       float* pred, float* loss_vec, float* gpred, float* ugrad, float* ub, const SmallPlanArgs& plan,
                       hipStream_t s);

struct SmallUpdArgs {
  float *I, *mI, *vI, *U, *mU, *vU;
  const rc_plan_row* rows;   // [kSmallPlanWgs][n]
  const uint32_t* occ;       // [kSmallPlanWgs][n]
  const SmallCnt* cnt;
  uint32_t n, n_a;
  int C;
  const float* gpred;        // [n_a]
  const float* ub;           // [B, D] pre-step user rows of the batch
  const float* ugrad;        // [B, D]
  OptScalars o;
  const float* loss_vec;
  int B;
  float loss_scale;
  float* loss_out;
};

__device__ __forceinline__ void padd4s(float4& x, const float4& y) {
  x.x += y.x; x.y += y.y; x.z += y.z; x.w += y.w;
}

constexpr int kSmallSeq = 32;   // occurrences one lane-group sums on its own

// gradient row (this lane's float4) of occurrence slot k of a row
template <int D>
__device__ __forceinline__ float4 small_grad4_at(const SmallUpdArgs& a, bool side_b, uint32_t o, int l) {
  constexpr int LPR = D / 4;
  if (side_b) return reinterpret_cast<const float4*>(a.ugrad)[(size_t)(o - a.n_a) * LPR + l];
  const float c = a.gpred[o];
  float4 v = reinterpret_cast<const float4*>(a.ub)[(size_t)(o / (uint32_t)a.C) * LPR + l];
  v.x *= c; v.y *= c; v.z *= c; v.w *= c;
  return v;
}
template <int D>
__device__ __forceinline__ float4 small_grad4(const SmallUpdArgs& a, bool side_b, uint32_t slot, int l) {
  return small_grad4_at<D>(a, side_b, a.occ[slot], l);
}

template <int D, int MODE>
__global__ __launch_bounds__(kBlock) void small_update_kernel(SmallUpdArgs a) {
  constexpr int LPR = D / 4;
  constexpr int GPW = 64 / LPR;
  __shared__ uint32_t pre_a[kSmallPlanWgs + 1], pre_b[kSmallPlanWgs + 1], cnt_a[kSmallPlanWgs];
  __shared__ float red[kBlock];
  __shared__ uint32_t wsum[8];
  const int tid = threadIdx.x, lane = tid & 63;
  const int l = lane % LPR, grp = lane / LPR;

  if (blockIdx.x == gridDim.x - 1) {  // the loss mean: fixed order (thread t sums t, t + 256, ...; LDS tree)
    if (a.loss_out == nullptr) return;
    // same order as plan_final_kernel / reduce_sum_kernel: the step's loss is bit-identical across pipelines
    const float acc = fixed_order_partial<kBlock, false>(a.loss_vec, (int64_t)a.B, tid);
    red[tid] = acc;
    __syncthreads();
    for (int off = kBlock / 2; off >= 1; off >>= 1) {
      if (tid < off) red[tid] += red[tid + off];
      __syncthreads();
    }
    if (tid == 0) a.loss_out[0] = red[0] * a.loss_scale;
    return;
  }

  {  // exclusive prefix of the per-workgroup row counts: one thread per plan workgroup
    static_assert(kSmallPlanWgs <= kBlock && kSmallPlanWgs % 64 == 0, "one thread per plan workgroup, whole waves");
    SmallCnt c;
    c.rows_a = c.rows_b = c.occ = c.pad = 0;
    if (tid < kSmallPlanWgs) c = a.cnt[tid];
    uint32_t xa = c.rows_a, xb = c.rows_b;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t ya = __shfl_up(xa, off, 64), yb = __shfl_up(xb, off, 64);
      if (lane >= off) { xa += ya; xb += yb; }
    }
    if (lane == 63) { wsum[tid >> 6] = xa; wsum[4 + (tid >> 6)] = xb; }
    __syncthreads();
    for (int q = 0; q < (tid >> 6); ++q) { xa += wsum[q]; xb += wsum[4 + q]; }
    if (tid < kSmallPlanWgs) { pre_a[tid] = xa - c.rows_a; pre_b[tid] = xb - c.rows_b; cnt_a[tid] = c.rows_a; }
    if (tid == kSmallPlanWgs - 1) { pre_a[kSmallPlanWgs] = xa; pre_b[kSmallPlanWgs] = xb; }
  }
  __syncthreads();
  const uint32_t Ra = pre_a[kSmallPlanWgs], R = Ra + pre_b[kSmallPlanWgs];
  const uint32_t n_waves = (gridDim.x - 1) * (kBlock / 64);
  const uint32_t wave = blockIdx.x * (kBlock / 64) + (tid >> 6);
  for (uint32_t r0 = wave * GPW; r0 < R; r0 += n_waves * GPW) {
    // this lane-group's row: flat index -> (side, plan workgroup, index in its segment)
    const uint32_t r = r0 + grp;
    const bool on = r < R;
    const bool side_b = on && r >= Ra;
    rc_plan_row e;
    e.row = 0; e.start = 0; e.n = 0; e.reserved = 0;
    if (on) {
      const uint32_t* pre = side_b ? pre_b : pre_a;
      const uint32_t q = side_b ? r - Ra : r;
      int lo = 0, hi = kSmallPlanWgs;   // last w with pre[w] <= q
      while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (pre[mid] <= q) lo = mid; else hi = mid;
      }
      const uint32_t j = q - pre[lo] + (side_b ? cnt_a[lo] : 0u);
      e = a.rows[(size_t)lo * a.n + j];
    }
    float* W = side_b ? a.U : a.I;
    float* M = side_b ? a.mU : a.mI;
    float* V = side_b ? a.vU : a.vI;
    const size_t idx4 = (size_t)e.row * LPR + l;
    float4 w4 = make_float4(0.f, 0.f, 0.f, 0.f), m4 = w4, v4 = w4;
    if (on) {
      w4 = reinterpret_cast<const float4*>(W)[idx4];
      if (mode_has_m(MODE)) m4 = reinterpret_cast<const float4*>(M)[idx4];
      if (mode_has_v(MODE)) v4 = reinterpret_cast<const float4*>(V)[idx4];
    }
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const bool seq = on && e.n <= (uint32_t)kSmallSeq;
    if (seq) {
      acc = small_grad4_at<D>(a, side_b, e.reserved, l);   // the record carries the row's first position
      for (uint32_t k = 1; k < e.n; ++k) padd4s(acc, small_grad4<D>(a, side_b, e.start + k, l));
    }
    // hot rows of this wave's groups, one after the other, summed by the whole wave
    uint64_t hot = __ballot(on && !seq && l == 0);
    while (hot) {
      const int src = __ffsll((long long)hot) - 1;   // lane 0 of the owning group
      hot &= hot - 1;
      const uint32_t hs = __shfl(e.start, src, 64), hn = __shfl(e.n, src, 64);
      const bool hb = __shfl((int)side_b, src, 64) != 0;
      float4 part = make_float4(0.f, 0.f, 0.f, 0.f);
      for (uint32_t k = grp; k < hn; k += GPW) padd4s(part, small_grad4<D>(a, hb, hs + k, l));
      part.x = groups_allreduce_sum<LPR, 64>(part.x);
      part.y = groups_allreduce_sum<LPR, 64>(part.y);
      part.z = groups_allreduce_sum<LPR, 64>(part.z);
      part.w = groups_allreduce_sum<LPR, 64>(part.w);
      if (lane / LPR == src / LPR) acc = part;
    }
    if (on) {
      opt_apply4<MODE>(a.o, w4, m4, v4, acc);
      reinterpret_cast<float4*>(W)[idx4] = w4;
      if (mode_has_m(MODE)) reinterpret_cast<float4*>(M)[idx4] = m4;
      if (mode_has_v(MODE)) reinterpret_cast<float4*>(V)[idx4] = v4;
    }
  }
}

template <int D>
static int small_update_launch_d(const SmallUpdArgs& a, int mode, hipStream_t s) {
  // one lane-group per row, a few rows per group: the grid covers the worst case (every key a distinct row)
  constexpr int GPB = kBlock / (D / 4);
  unsigned blocks = (a.n + GPB - 1) / GPB;
  if (blocks > 2048u) blocks = 2048u;
  if (blocks < 1u) blocks = 1u;
  switch (mode) {
    case MODE_SGD: hipLaunchKernelGGL((small_update_kernel<D, MODE_SGD>), dim3(blocks + 1), dim3(kBlock), 0, s, a); break;
    case MODE_ADAM: hipLaunchKernelGGL((small_update_kernel<D, MODE_ADAM>), dim3(blocks + 1), dim3(kBlock), 0, s, a); break;
    default: hipLaunchKernelGGL((small_update_kernel<D, MODE_ADAGRAD>), dim3(blocks + 1), dim3(kBlock), 0, s, a); break;
  }
  RC_LAUNCH_CHECK();
  return RC_OK;
}

// ---- aten::embedding_dense_backward for a SMALL id list (<= 32,768 ids) in two launches ---------------------------------
// The dense gradient of an embedding table (helpers/BaseRunner.py:205 behind loss.backward(): zero-fill + index_add of
// every occurrence's gradient row) at the reference's own batch sizes -- 1,024 rows x 8 fields of a CTR step, 256 x
// (1 + K) candidates -- went through the radix sort + head list + chunk planning: 15-20 dependent launches of a few
// microseconds for a few thousand ids (a third of the replayed DeepFM step).  Here the 128 plan workgroups of
// small_plan.hpp group the ids (launch 1) and one lane-group per touched row sums its occurrences' gradient rows in
// ascending position (launch 2); rows past 32 occurrences by a whole wave with a fixed butterfly.  No atomics, no
// counters to zero, deterministic.  d in {16, 32, 64, 128}; d <= 4 (the [vocab, 1] first-order tables of the FM family):
// one wave per row, lanes stride over the occurrences.
__global__ __launch_bounds__(kSmallThreads) void small_plan_kernel(SmallPlanArgs plan) {
  extern __shared__ __attribute__((aligned(16))) unsigned char small_plan_smem[];
  small_plan_block<kSmallCapBig, kSmallWaveCapBig>(plan, blockIdx.x, small_plan_smem);
}

constexpr int kSmallNumeric = 4;
constexpr int kSmallNumericSplits = 4;    // workgroups per numeric field (disjoint column ranges)

struct SmallSumArgs {
  const rc_plan_row* rows;   // [kSmallPlanWgs][n]
  const uint32_t* occ;       // [kSmallPlanWgs][n]
  const SmallCnt* cnt;
  uint32_t n;
  const float* src;          // [n, d] gradient row of every occurrence
  float* out;                // [n_rows, d]; only touched rows are written
  int d;
  const float* src1;         // optional second, ONE-float-wide gradient of the same occurrences ([n]) and its table [n_rows]:
  float* out1;               // the [vocab, 1] first-order weights of the FM family ride along with the [vocab, d] vectors
  // rc_small_row_sums_pair_numeric: the weight gradients of up to kSmallNumeric numeric fields (numeric_grads.hpp) by one extra
  // workgroup each behind the row workgroups -- independent work that would otherwise be a launch of its own (~13 us of a replayed
  // DeepFM step at B = 1,024 on MIND's field set)
  uint32_t blocks_rows;      // workgroups of the row sums (the grid may be longer)
  int n_numeric;
  NumericSlot num[kSmallNumeric];
  NumericCommon numc;
  FmTap fm;                  // rc_small_row_sums_planned: the FM term's backward on top of src's rows (src may then be null)
};

// flat row index -> record (rows of plan workgroup w are the w-th segment; per-workgroup counts prefix-summed here)
struct SmallRowIndex {
  uint32_t pre[kSmallPlanWgs + 1];
};
__device__ __forceinline__ void small_row_prefix(const SmallCnt* cnt, SmallRowIndex* ix, uint32_t* wsum) {
  const int tid = threadIdx.x, lane = tid & 63;
  uint32_t c = tid < kSmallPlanWgs ? cnt[tid].rows_a : 0u, x = c;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t y = __shfl_up(x, off, 64);
    if (lane >= off) x += y;
  }
  if (lane == 63) wsum[tid >> 6] = x;
  __syncthreads();
  for (int q = 0; q < (tid >> 6); ++q) x += wsum[q];
  if (tid < kSmallPlanWgs) ix->pre[tid] = x - c;
  if (tid == kSmallPlanWgs - 1) ix->pre[kSmallPlanWgs] = x;
  __syncthreads();
}
__device__ __forceinline__ rc_plan_row small_row_at(const SmallSumArgs& a, const SmallRowIndex& ix, uint32_t q) {
  int lo = 0, hi = kSmallPlanWgs;   // last w with pre[w] <= q
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (ix.pre[mid] <= q) lo = mid; else hi = mid;
  }
  return a.rows[(size_t)lo * a.n + (q - ix.pre[lo])];
}

// the float4 of NB occurrences' gradient rows this lane sums (o[j]; in[j]: the slot is in use); FM: the FM term's backward added
// where the rows are read (numeric_grads.hpp) -- every load of the batch is requested before the first is used (an unused slot reads
// occurrence 0: no branch between the requests)
template <int D, bool FM, int NB>
__device__ __forceinline__ void small_occ_rows4(const SmallSumArgs& a, const float4* src4, const uint32_t (&o)[NB], const bool (&in)[NB], int l,
                                                float4 (&v)[NB]) {
  constexpr int LPR = D / 4;
  const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
  if (!FM) {
#pragma unroll
    for (int j = 0; j < NB; ++j) v[j] = in[j] ? src4[(size_t)o[j] * LPR + l] : zero;
    return;
  }
  const float4* V4 = reinterpret_cast<const float4*>(a.fm.V);
  const float4* S4 = reinterpret_cast<const float4*>(a.fm.S);
  float4 x[NB], sv[NB];
  float g[NB];
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    const uint32_t oj = in[j] ? o[j] : 0u;
    const uint32_t r = a.fm.magic_F == 0u ? oj : __umulhi(oj, a.fm.magic_F);
    v[j] = src4 != nullptr ? src4[(size_t)oj * LPR + l] : zero;
    x[j] = V4[(size_t)oj * LPR + l];
    sv[j] = S4[(size_t)r * LPR + l];
    g[j] = a.fm.g[r];
  }
#pragma unroll
  for (int j = 0; j < NB; ++j) v[j] = in[j] ? fm_tap4(v[j], g[j], sv[j], x[j]) : zero;
}

template <int D, bool FM>
__global__ __launch_bounds__(kBlock) void small_row_sums_kernel(SmallSumArgs a) {
  constexpr int LPR = D / 4;
  constexpr int GPW = 64 / LPR;
  if (blockIdx.x >= a.blocks_rows) {   // (workgroup-uniform) a numeric field's weight gradients over the whole batch
    // kSmallNumericSplits workgroups per field, each a quarter of the columns: 4 lanes per row at d = 64, i.e. 64 rows per load
    // instruction and 1,024 rows -- the whole batch at B = 1,024 -- in flight in ONE trip of sixteen
    const int w = (int)(blockIdx.x - a.blocks_rows), j = w / kSmallNumericSplits;
    numeric_slot_grads<4, kBlock, 16>(a.num[j], a.numc, j, 0u, true, w % kSmallNumericSplits, kSmallNumericSplits);
    return;
  }
  __shared__ SmallRowIndex ix;
  __shared__ uint32_t wsum[8];
  const int tid = threadIdx.x, lane = tid & 63;
  const int l = lane % LPR, grp = lane / LPR;
  small_row_prefix(a.cnt, &ix, wsum);
  const uint32_t R = ix.pre[kSmallPlanWgs];
  const uint32_t n_waves = a.blocks_rows * (kBlock / 64);
  const uint32_t wave = blockIdx.x * (kBlock / 64) + (tid >> 6);
  const float4* src4 = reinterpret_cast<const float4*>(a.src);
  for (uint32_t r0 = wave * GPW; r0 < R; r0 += n_waves * GPW) {
    const uint32_t r = r0 + grp;
    const bool on = r < R;
    rc_plan_row e;
    e.row = 0; e.start = 0; e.n = 0; e.reserved = 0;
    if (on) e = small_row_at(a, ix, r);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    float acc1 = 0.f;
    // Rows of up to kSmallSeq occurrences: the lane-group on its own, in ascending position like the sort route (bit-identical to it),
    // eight gradient rows in flight and the next eight occurrence numbers requested beside them -- 1 + ceil(n / 8) memory latencies.
    // (The first version walked "occ[k] -> gradient row" one after the other: a field of 50 values at B = 1,024 is 20 dependent
    // round trips per row, and such rows set the kernel's 20 us.)
    constexpr int kB = 8;
    const bool seq = on && e.n <= (uint32_t)kSmallSeq;
    if (seq) {
      uint32_t oc[kB], on_[kB];
#pragma unroll
      for (int j = 0; j < kB; ++j) oc[j] = (j > 0 && (uint32_t)j < e.n) ? a.occ[e.start + j] : 0u;
      oc[0] = e.reserved;   // the record carries the row's first position
      for (uint32_t k0 = 0; k0 < e.n; k0 += kB) {
#pragma unroll
        for (int j = 0; j < kB; ++j) on_[j] = k0 + kB + j < e.n ? a.occ[e.start + k0 + kB + j] : 0u;
        float4 v[kB];
        float s1[kB];
        bool in_[kB];
#pragma unroll
        for (int j = 0; j < kB; ++j) in_[j] = k0 + j < e.n;
        small_occ_rows4<D, FM, kB>(a, src4, oc, in_, l, v);
#pragma unroll
        for (int j = 0; j < kB; ++j) s1[j] = (a.src1 && in_[j]) ? a.src1[oc[j]] : 0.f;
#pragma unroll
        for (int j = 0; j < kB; ++j) {
          if (k0 + j < e.n) {
            if (k0 + j == 0) { acc = v[0]; acc1 = s1[0]; }
            else { padd4s(acc, v[j]); acc1 += s1[j]; }
          }
        }
#pragma unroll
        for (int j = 0; j < kB; ++j) oc[j] = on_[j];
      }
    }
    // Longer rows of this wave's groups, one after the other, by the whole wave: up to 192 occurrence numbers in one round of
    // coalesced loads (three per lane), handed out by shuffles; then 32 gradient rows in flight per round (U per lane-group, fixed
    // pattern -> fixed order).  A row of 146 occurrences (a field of 7 values at B = 1,024) is 1 + 5 memory latencies.
    uint64_t hot = __ballot(on && !seq && l == 0);
    while (hot) {
      const int srcl = __ffsll((long long)hot) - 1;
      hot &= hot - 1;
      const uint32_t hs = __shfl(e.start, srcl, 64), hn = __shfl(e.n, srcl, 64);
      constexpr int U = 32 / GPW >= 1 ? 32 / GPW : 1;     // U * GPW = 32 (d = 16 .. 128): a round never straddles a multiple of 64
      static_assert(U * GPW == 32 || GPW > 32, "rounds of 32 occurrences");
      float4 part = make_float4(0.f, 0.f, 0.f, 0.f);
      float part1 = 0.f;
      for (uint32_t c0 = 0; c0 < hn; c0 += 192) {   // (wave-uniform)
        const uint32_t cn = hn - c0 < 192u ? hn - c0 : 192u;
        const uint32_t pm0 = (uint32_t)lane < cn ? a.occ[hs + c0 + lane] : 0u;
        const uint32_t pm1 = 64u + lane < cn ? a.occ[hs + c0 + 64 + lane] : 0u;
        const uint32_t pm2 = 128u + lane < cn ? a.occ[hs + c0 + 128 + lane] : 0u;
        for (uint32_t base = 0; base < cn; base += U * GPW) {
          const uint32_t k64 = base >> 6;   // wave-uniform: the whole round reads one of the three registers
          const uint32_t mine = k64 == 0 ? pm0 : (k64 == 1 ? pm1 : pm2);
          float4 sv[U];
          float s1[U];
          uint32_t ou[U];
          bool in_[U];
#pragma unroll
          for (int u = 0; u < U; ++u) {
            const uint32_t idx = base + u * GPW + grp;
            ou[u] = __shfl(mine, idx & 63, 64);
            in_[u] = idx < cn;
          }
          small_occ_rows4<D, FM, U>(a, src4, ou, in_, l, sv);
#pragma unroll
          for (int u = 0; u < U; ++u) s1[u] = (a.src1 && in_[u]) ? a.src1[ou[u]] : 0.f;
#pragma unroll
          for (int u = 0; u < U; ++u) {
            padd4s(part, sv[u]);
            part1 += s1[u];
          }
        }
      }
      part1 = groups_allreduce_sum<LPR, 64>(part1);
      if (lane / LPR == srcl / LPR) acc1 = part1;
      part.x = groups_allreduce_sum<LPR, 64>(part.x);
      part.y = groups_allreduce_sum<LPR, 64>(part.y);
      part.z = groups_allreduce_sum<LPR, 64>(part.z);
      part.w = groups_allreduce_sum<LPR, 64>(part.w);
      if (lane / LPR == srcl / LPR) acc = part;
    }
    if (on) reinterpret_cast<float4*>(a.out)[(size_t)e.row * LPR + l] = acc;
    if (on && l == 0 && a.out1) a.out1[e.row] = acc1;
  }
}

// d <= 4: one wave per row, lane k takes occurrences k, k + 64, ... (ascending), a fixed butterfly combines the lanes
__global__ __launch_bounds__(kBlock) void small_row_sums_narrow_kernel(SmallSumArgs a) {
  __shared__ SmallRowIndex ix;
  __shared__ uint32_t wsum[8];
  const int tid = threadIdx.x, lane = tid & 63;
  small_row_prefix(a.cnt, &ix, wsum);
  const uint32_t R = ix.pre[kSmallPlanWgs];
  const uint32_t n_waves = gridDim.x * (kBlock / 64);
  for (uint32_t r = blockIdx.x * (kBlock / 64) + (tid >> 6); r < R; r += n_waves) {
    const rc_plan_row e = small_row_at(a, ix, r);
    float part[4] = {0.f, 0.f, 0.f, 0.f};
    for (uint32_t k = lane; k < e.n; k += 64) {
      const float* sp = a.src + (size_t)a.occ[e.start + k] * a.d;
#pragma unroll
      for (int c = 0; c < 4; ++c)
        if (c < a.d) part[c] += sp[c];
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float t = wave_allreduce_sum(part[c]);
      if (c < a.d && lane == c) a.out[(size_t)e.row * a.d + c] = t;
    }
  }
}

// workspace of the small-batch step beyond gpred / ugrad / loss_vec
size_t small_step_extra_bytes(int64_t n, int64_t B, int d) {
  size_t t = 0;
  t += align_up((size_t)B * d * sizeof(float), 256);                                   // ub
  t += align_up((size_t)kSmallPlanWgs * (size_t)n * sizeof(rc_plan_row), 256);         // rows
  t += align_up((size_t)kSmallPlanWgs * (size_t)n * sizeof(uint32_t), 256);            // occ
  t += align_up((size_t)kSmallPlanWgs * sizeof(SmallCnt), 256);                       // cnt
  return t;
}

bool small_step_supported(int64_t n_i, int64_t B, int64_t n_items, int64_t n_users, int d) {
  return n_i + B <= kSmallMaxKeys && n_items + n_users < ((int64_t)1 << 32) && (d == 16 || d == 32 || d == 64 || d == 128);
}

int small_step_launch(float* U, float* I, float* mU, float* vU, float* mI, float* vI, const int64_t* uid, const int64_t* iid,
                      int B, int C, int d, int64_t n_items, const rc_opt_hyper* h, float inv_b, float* loss_out, float* pred,
                      float* gpred, float* ugrad, float* loss_vec, void* extra, hipStream_t s, hipEvent_t* ev_mid) {
  const int64_t n_i = (int64_t)B * C, n = n_i + B;
  Carver cv(extra);
  float* ub = cv.take<float>((size_t)B * d);
  rc_plan_row* rows = cv.take<rc_plan_row>((size_t)kSmallPlanWgs * (size_t)n);
  uint32_t* occ = cv.take<uint32_t>((size_t)kSmallPlanWgs * (size_t)n);
  SmallCnt* cnt = cv.take<SmallCnt>(kSmallPlanWgs);
  SmallPlanArgs p;
  memset(&p, 0, sizeof(p));
  p.ids_a = iid; p.ids_b = uid; p.n_a = (uint32_t)n_i; p.n = (uint32_t)n; p.base_b = (uint32_t)n_items;
  p.rows = rows; p.occ = occ; p.cnt = cnt;
  RC_TRY(small_front_launch(U, I, uid, iid, B, C, d, inv_b, pred, loss_vec, gpred, ugrad, ub, p, s));
  if (ev_mid) {  // two consecutive marks: [0] closes the front launch, [1] opens the update launch
    RC_HIP(hipEventRecord(ev_mid[0], s));
    RC_HIP(hipEventRecord(ev_mid[1], s));
  }
  SmallUpdArgs a;
  memset(&a, 0, sizeof(a));
  RC_TRY(fill_opt_scalars(h, &a.o));
  const int mode = mode_of(h);
  RC_REQUIRE(mode != MODE_ADAM || (mU && vU && mI && vI), "rc_bprmf_train_step: Adam needs m and v tables");
  RC_REQUIRE(mode != MODE_ADAGRAD || (mU && mI), "rc_bprmf_train_step: Adagrad needs the state_sum tables");
  a.I = I; a.mI = mI; a.vI = vI; a.U = U; a.mU = mU; a.vU = vU;
  a.rows = rows; a.occ = occ; a.cnt = cnt; a.n = (uint32_t)n; a.n_a = (uint32_t)n_i; a.C = C;
  a.gpred = gpred; a.ub = ub; a.ugrad = ugrad; a.loss_vec = loss_vec; a.B = B; a.loss_scale = inv_b; a.loss_out = loss_out;
  switch (d) {
    case 16: return small_update_launch_d<16>(a, mode, s);
    case 32: return small_update_launch_d<32>(a, mode, s);
    case 64: return small_update_launch_d<64>(a, mode, s);
    default: return small_update_launch_d<128>(a, mode, s);
  }
}

}  // namespace rc

using namespace rc;

// the plan workgroups keep their tables in ~148 KB of dynamic LDS: a part with less (not gfx950) has no instance
static bool small_lds_fits() {
  static std::mutex mu;
  static int fits[64];
  static bool known[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return false;
  std::lock_guard<std::mutex> lock(mu);
  if (!known[dev]) {
    int lds = 0;
    fits[dev] = hipDeviceGetAttribute(&lds, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) == hipSuccess && (size_t)lds >= kSmallLdsBytesBig;
    known[dev] = true;
  }
  return fits[dev] != 0;
}

extern "C" int rc_small_row_sums_supported(int64_t n, int64_t n_rows, int d) {
  return (n >= 1 && n <= kSmallMaxKeys && n_rows >= 1 && n_rows < ((int64_t)1 << 32) - 1 &&
          ((d >= 1 && d <= 4) || d == 16 || d == 32 || d == 64 || d == 128) && small_lds_fits()) ? 1 : 0;
}

extern "C" size_t rc_small_row_sums_workspace_bytes(int64_t n) {
  if (n < 1) n = 1;
  return align_up((size_t)kSmallPlanWgs * (size_t)n * sizeof(rc_plan_row), 256) +
         align_up((size_t)kSmallPlanWgs * (size_t)n * sizeof(uint32_t), 256) + align_up((size_t)kSmallPlanWgs * sizeof(SmallCnt), 256);
}

struct SmallNumericCall {     // host-side description of the numeric fields that ride in the launch
  int n_numeric;
  NumericSlot num[kSmallNumeric];
  NumericCommon c;
};

static int small_row_sums_impl(bool build_plan, const int64_t* ids, int64_t n, int64_t n_rows, const float* src, int d, float* out, void* ws,
                               size_t ws_bytes, rc_stream_t stream, const float* src1 = nullptr, float* out1 = nullptr,
                               const SmallNumericCall* numeric = nullptr, const FmTap* fm = nullptr) {
  if (n == 0) return RC_OK;
  RC_REQUIRE((ids || !build_plan) && (src || fm) && out && ws, "rc_small_row_sums: null pointer");
  if (!rc_small_row_sums_supported(n, n_rows, d))
    return fail(RC_ERR_UNSUPPORTED, "rc_small_row_sums: n=%lld (<= %d), n_rows=%lld, d=%d (1..4, 16, 32, 64, 128) not covered",
                (long long)n, kSmallMaxKeys, (long long)n_rows, d);
  if (ws_bytes < rc_small_row_sums_workspace_bytes(n))
    return fail(RC_ERR_WORKSPACE, "rc_small_row_sums: workspace %zu < %zu", ws_bytes, rc_small_row_sums_workspace_bytes(n));
  RC_REQUIRE(d <= 4 || (reinterpret_cast<uintptr_t>(src) % 16 == 0 && reinterpret_cast<uintptr_t>(out) % 16 == 0),
             "rc_small_row_sums: src / out must be 16-byte aligned");
  RC_REQUIRE(fm == nullptr || (d >= 16 && fm->V && fm->S && fm->g && fm->F >= 1 && n % fm->F == 0 && reinterpret_cast<uintptr_t>(fm->V) % 16 == 0 &&
                               reinterpret_cast<uintptr_t>(fm->S) % 16 == 0),
             "rc_small_row_sums_planned: the FM term's backward needs d >= 16, 16-byte aligned V [n / F, F, d] and S [n / F, d], and g [n / F]");
  hipStream_t s = as_stream(stream);
  Carver cv(ws);
  rc_plan_row* rows = cv.take<rc_plan_row>((size_t)kSmallPlanWgs * (size_t)n);
  uint32_t* occ = cv.take<uint32_t>((size_t)kSmallPlanWgs * (size_t)n);
  SmallCnt* cnt = cv.take<SmallCnt>(kSmallPlanWgs);
  if (build_plan) {
    SmallPlanArgs p;
    memset(&p, 0, sizeof(p));
    p.ids_a = ids; p.ids_b = nullptr; p.n_a = (uint32_t)n; p.n = (uint32_t)n; p.base_b = 0xFFFFFFFFu;   // one list: every key is a row of it
    p.rows = rows; p.occ = occ; p.cnt = cnt;
    {   // function attributes are per device: once per device of the process, under a lock (callers may drive several GPUs / threads)
      static std::mutex mu;
      static bool done[64] = {};
      int dev = 0;
      RC_HIP(hipGetDevice(&dev));
      std::lock_guard<std::mutex> lock(mu);
      if (dev < 0 || dev >= 64 || !done[dev]) {
        RC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(small_plan_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kSmallLdsBytesBig));
        if (dev >= 0 && dev < 64) done[dev] = true;
      }
    }
    hipLaunchKernelGGL(small_plan_kernel, dim3(kSmallPlanWgs), dim3(kSmallThreads), kSmallLdsBytesBig, s, p);
    RC_LAUNCH_CHECK();
  }
  SmallSumArgs a;
  a.rows = rows; a.occ = occ; a.cnt = cnt; a.n = (uint32_t)n; a.src = src; a.out = out; a.d = d;
  a.src1 = src1; a.out1 = out1;
  a.n_numeric = 0;
  memset(&a.numc, 0, sizeof(a.numc));
  memset(&a.fm, 0, sizeof(a.fm));
  if (fm != nullptr) a.fm = *fm;
  RC_REQUIRE((src1 == nullptr) == (out1 == nullptr) && (src1 == nullptr || d >= 16), "rc_small_row_sums_pair: the one-float-wide pair rides with d >= 16 only");
  RC_REQUIRE(numeric == nullptr || d >= 16, "rc_small_row_sums_pair_numeric: the numeric fields ride with the d >= 16 kernels only");
  if (numeric != nullptr) {
    a.n_numeric = numeric->n_numeric;
    for (int j = 0; j < numeric->n_numeric; ++j) a.num[j] = numeric->num[j];
    a.numc = numeric->c;
    a.numc.fm = a.fm;
  }
  if (d <= 4) {
    unsigned blocks = (unsigned)((n + kBlock / 64 - 1) / (kBlock / 64));
    if (blocks > 1024u) blocks = 1024u;
    hipLaunchKernelGGL(small_row_sums_narrow_kernel, dim3(blocks), dim3(kBlock), 0, s, a);
  } else {
    const int gpb = kBlock / (d / 4);
    unsigned blocks = (unsigned)((n + gpb - 1) / gpb);
    if (blocks > 2048u) blocks = 2048u;
    a.blocks_rows = blocks;
    const unsigned grid = blocks + (unsigned)(a.n_numeric * kSmallNumericSplits);     // the numeric fields' workgroups come last: the rows' are dispatched first
    if (fm != nullptr) {
      switch (d) {
        case 16: hipLaunchKernelGGL((small_row_sums_kernel<16, true>), dim3(grid), dim3(kBlock), 0, s, a); break;
        case 32: hipLaunchKernelGGL((small_row_sums_kernel<32, true>), dim3(grid), dim3(kBlock), 0, s, a); break;
        case 64: hipLaunchKernelGGL((small_row_sums_kernel<64, true>), dim3(grid), dim3(kBlock), 0, s, a); break;
        default: hipLaunchKernelGGL((small_row_sums_kernel<128, true>), dim3(grid), dim3(kBlock), 0, s, a); break;
      }
    } else {
      switch (d) {
        case 16: hipLaunchKernelGGL((small_row_sums_kernel<16, false>), dim3(grid), dim3(kBlock), 0, s, a); break;
        case 32: hipLaunchKernelGGL((small_row_sums_kernel<32, false>), dim3(grid), dim3(kBlock), 0, s, a); break;
        case 64: hipLaunchKernelGGL((small_row_sums_kernel<64, false>), dim3(grid), dim3(kBlock), 0, s, a); break;
        default: hipLaunchKernelGGL((small_row_sums_kernel<128, false>), dim3(grid), dim3(kBlock), 0, s, a); break;
      }
    }
  }
  RC_LAUNCH_CHECK();
  return RC_OK;
}

extern "C" int rc_small_row_sums(const int64_t* ids, int64_t n, int64_t n_rows, const float* src, int d, float* out, void* ws,
                                 size_t ws_bytes, rc_stream_t stream) {
  return small_row_sums_impl(true, ids, n, n_rows, src, d, out, ws, ws_bytes, stream);
}

extern "C" int rc_small_row_sums_again(int64_t n, int64_t n_rows, const float* src, int d, float* out, void* ws, size_t ws_bytes,
                                       rc_stream_t stream) {
  return small_row_sums_impl(false, nullptr, n, n_rows, src, d, out, ws, ws_bytes, stream);
}

extern "C" int rc_small_row_sums_pair(const int64_t* ids, int64_t n, int64_t n_rows, const float* src, int d, float* out,
                                      const float* src1, float* out1, void* ws, size_t ws_bytes, rc_stream_t stream) {
  RC_REQUIRE(src1 && out1, "rc_small_row_sums_pair: null pointer");
  return small_row_sums_impl(true, ids, n, n_rows, src, d, out, ws, ws_bytes, stream, src1, out1);
}

/* rc_small_row_sums_pair with the weight gradients of the numeric fields of the same gradient blocks (rc_numeric_field_grads)
 * riding in the row-sums launch: src = gV [rows, F, d] seen as n = rows * F occurrence rows, src1 = gL [rows * F]. */
extern "C" int rc_small_row_sums_pair_numeric(const int64_t* ids, int64_t n, int64_t n_rows, const float* src, int d, float* out,
                                              const float* src1, float* out1, const void* const* values, const int* per_row,
                                              const int* kind, const int* field, int n_numeric, int F, int64_t B, int C,
                                              float* const* dW, float* const* dw1, void* ws, size_t ws_bytes, rc_stream_t stream) {
  RC_REQUIRE(src1 && out1 && values && per_row && kind && field && dW && dw1, "rc_small_row_sums_pair_numeric: null pointer");
  RC_REQUIRE(n_numeric >= 1 && n_numeric <= kSmallNumeric && n_numeric <= F && F <= kMaxFields,
             "rc_small_row_sums_pair_numeric: %d numeric fields (1 .. %d) of F = %d", n_numeric, kSmallNumeric, F);
  RC_REQUIRE(B >= 1 && C >= 1 && B * C * F == n && d % 4 == 0 && d <= 4 * kBlock, "rc_small_row_sums_pair_numeric: bad shape B=%lld C=%d F=%d n=%lld d=%d",
             (long long)B, C, F, (long long)n, d);
  SmallNumericCall nc;
  memset(&nc, 0, sizeof(nc));
  nc.n_numeric = n_numeric;
  for (int j = 0; j < n_numeric; ++j) {
    RC_REQUIRE(values[j] && dW[j] && dw1[j] && field[j] >= 0 && field[j] < F && kind[j] >= RC_FIELD_F32 && kind[j] <= RC_FIELD_I64,
               "rc_small_row_sums_pair_numeric: bad numeric field %d", j);
    nc.num[j].values = values[j]; nc.num[j].dW = dW[j]; nc.num[j].dw1 = dw1[j];
    nc.num[j].kind = kind[j]; nc.num[j].per_row = per_row[j]; nc.num[j].field = field[j];
  }
  nc.c.gV = src; nc.c.gL = src1; nc.c.part = nullptr; nc.c.n = B * C; nc.c.n_numeric = n_numeric; nc.c.F = F; nc.c.C = C; nc.c.d = d;
  return small_row_sums_impl(true, ids, n, n_rows, src, d, out, ws, ws_bytes, stream, src1, out1, &nc);
}

/* The row sums of a backward pass whose grouping rc_gather_fields_fused already left in `ws` (no plan launch), for both table
 * families of the FM models, with -- each optional -- the numeric fields' weight gradients riding along (n_numeric > 0, as
 * rc_small_row_sums_pair_numeric) and the FM pairwise term's backward folded in (fm_V != null): occurrence o = r F + f then
 * contributes  src[o] + fm_g[r] * (fm_S[r] - fm_V[o])  (rc_fm_second_order_bwd_add's rows, never written out; src may be null:
 * no other consumer of the field vectors). */
extern "C" int rc_small_row_sums_planned(int64_t n, int64_t n_rows, const float* src, int d, float* out, const float* src1, float* out1,
                                         const void* const* values, const int* per_row, const int* kind, const int* field,
                                         int n_numeric, int F, int64_t B, int C, float* const* dW, float* const* dw1, const float* fm_V,
                                         const float* fm_S, const float* fm_g, void* ws, size_t ws_bytes, rc_stream_t stream) {
  RC_REQUIRE(src1 && out1, "rc_small_row_sums_planned: null pointer");
  RC_REQUIRE(n_numeric >= 0 && n_numeric <= kSmallNumeric && n_numeric <= F && F >= 1 && F <= kMaxFields,
             "rc_small_row_sums_planned: %d numeric fields (0 .. %d) of F = %d", n_numeric, kSmallNumeric, F);
  RC_REQUIRE(B >= 1 && C >= 1 && B * C * F == n && d % 4 == 0 && d >= 16, "rc_small_row_sums_planned: bad shape B=%lld C=%d F=%d n=%lld d=%d",
             (long long)B, C, F, (long long)n, d);
  RC_REQUIRE((fm_V == nullptr) == (fm_S == nullptr) && (fm_V == nullptr) == (fm_g == nullptr), "rc_small_row_sums_planned: fm_V, fm_S and fm_g come together");
  FmTap fm;
  fm.V = fm_V; fm.S = fm_S; fm.g = fm_g; fm.F = (uint32_t)F; fm.magic_F = small_div_magic((uint32_t)F);
  SmallNumericCall nc;
  memset(&nc, 0, sizeof(nc));
  if (n_numeric > 0) {
    RC_REQUIRE(values && per_row && kind && field && dW && dw1, "rc_small_row_sums_planned: null pointer (numeric fields)");
    nc.n_numeric = n_numeric;
    for (int j = 0; j < n_numeric; ++j) {
      RC_REQUIRE(values[j] && dW[j] && dw1[j] && field[j] >= 0 && field[j] < F && kind[j] >= RC_FIELD_F32 && kind[j] <= RC_FIELD_I64,
                 "rc_small_row_sums_planned: bad numeric field %d", j);
      nc.num[j].values = values[j]; nc.num[j].dW = dW[j]; nc.num[j].dw1 = dw1[j];
      nc.num[j].kind = kind[j]; nc.num[j].per_row = per_row[j]; nc.num[j].field = field[j];
    }
    nc.c.gV = src; nc.c.gL = src1; nc.c.part = nullptr; nc.c.n = B * C; nc.c.n_numeric = n_numeric; nc.c.F = F; nc.c.C = C; nc.c.d = d;
  }
  return small_row_sums_impl(false, nullptr, n, n_rows, src, d, out, ws, ws_bytes, stream, src1, out1, n_numeric > 0 ? &nc : nullptr,
                             fm_V != nullptr ? &fm : nullptr);
}
