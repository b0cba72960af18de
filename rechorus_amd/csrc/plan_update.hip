// plan_update.hip -- optimizer row updates driven by a bucket plan (bucket_plan.hip): per listed row, the
// gradient rows of its occurrences are summed in ascending batch position (fixed order, no float atomics),
// the table row is read once, updated, written once.
//
// Reference semantics being replaced: aten::embedding_dense_backward (index_add of every occurrence into a
// dense [n_rows, d] gradient) followed by torch.optim's step (helpers/BaseRunner.py:193,205-206; optimizer
// built at :110-114), restricted to the rows the batch touches ("row-wise").
//
//   plan_rows_kernel    grid-stride over rc_plan_row records, one lane-group (d/4 lanes, a float4 each) per
//                       row, two rows per lane-group in flight.  Per-occurrence gradient rows are rebuilt on the
//                       fly (BPRMF items: g[b,c] * U[uid[b]]) and never exist in HBM.  Rows with more than
//                       kPlanLongSeg occurrences (the head of a Zipf distribution) are cut into chunks here
//                       (row record + chunk list, integer atomics only) and left to:
//   plan_chunk_kernel   one workgroup per 256-occurrence chunk, fixed LDS tree -> partial sums
//   plan_final_kernel   chunk partials of a row combined in chunk order, optimizer, row written
#include "plan.hpp"

namespace rc {

constexpr bool LPR_OK(int D) { return D / 4 <= 64 && 64 % (D / 4) == 0; }  // a row's lane-group fits a wave

struct PlanSide {       // one table and how the gradient rows of its occurrences are obtained
  PlanTable t;
  PlanTable tb;         // pair mode: the second table
  PlanGrad g;
  const rc_plan_row* rows;
  const uint32_t* n_rows;
};

struct PlanUpdArgs {
  PlanSide side[2];     // BPRMF step: 0 = item table, 1 = user table
  const uint32_t* occ;
  uint32_t* counters;   // PC_LONG, PC_CHUNKS
  PlanLongWs lw;
  OptScalars o;
  // loss mean folded into the last launch (one extra workgroup): out[0] = scale * sum(vec[0..n))
  const float* loss_vec;
  int64_t loss_n;
  float loss_scale;
  float* loss_out;
  // the plan registered the hot rows already (bucket_plan.hip, emit_long): their chunks are reduced by extra workgroups
  // of the row-update launch instead of a launch of their own
  int long_planned;
  // occurrences per chunk of a hot row.  kPlanChunk where the plan registered the hot rows (emit_long); where this file does
  // (rc_plan_update*, rc_plan_row_sums) it follows the row width: a chunk is walked 16 occurrences per lane-group and trip, and
  // a 256-float row (a NeuMF table pair) has 4 lane-groups per workgroup -- 256 occurrences were 16 dependent trips (67 us for
  // the few hot rows of a NeuMF batch), 64 are four
  uint32_t chunk;
};

template <int D>
constexpr uint32_t side_chunk() { return D >= 256 ? 64u : (D == 128 ? 128u : (uint32_t)kPlanChunk); }

__device__ __forceinline__ void padd4(float4& x, const float4& y) {
  x.x += y.x; x.y += y.y; x.z += y.z; x.w += y.w;
}

// gradient row of the occurrence in slot `slot` of occ[], this lane's float4
template <int D>
__device__ __forceinline__ float4 plan_grad4(const PlanGrad& s, const uint32_t* __restrict__ occ, uint32_t slot, int l) {
  constexpr int LPR = D / 4;
  const uint32_t o = occ[slot];
  if (s.pair) {  // kernel-uniform
    constexpr int H = LPR / 2;
    const bool up = l >= H;
    const size_t ldq = s.pair_ld4 ? (size_t)s.pair_ld4 : (size_t)H;
    return reinterpret_cast<const float4*>(up ? s.src2b : s.src2)[(size_t)(o - s.n_split) * ldq + (up ? l - H : l)];
  }
  if (o >= s.n_split) return reinterpret_cast<const float4*>(s.src2)[(size_t)(o - s.n_split) * LPR + l];
  int64_t sr = (s.div == 1) ? (int64_t)o : (int64_t)(o / (uint32_t)s.div);
  if (s.src_index) sr = s.src_index[sr];
  const float c = s.coef ? s.coef[o] : 1.0f;
  float4 v = reinterpret_cast<const float4*>(s.src)[(size_t)sr * LPR + l];
  v.x *= c; v.y *= c; v.z *= c; v.w *= c;
  return v;
}

// the table (and float4 slot) a lane works on: the side's table, or in pair mode table a / b by lane half
template <int D>
__device__ __forceinline__ PlanTable plan_lane_table(const PlanSide& sd, uint32_t row, int l, size_t& idx4) {
  constexpr int LPR = D / 4;
  if (!sd.g.pair) {
    idx4 = (size_t)row * LPR + l;
    return sd.t;
  }
  constexpr int H = LPR / 2;
  const bool up = l >= H;
  idx4 = (size_t)row * H + (up ? l - H : l);
  return up ? sd.tb : sd.t;
}

// hot row: chunk records for plan_chunk_kernel / plan_final_kernel (lane-group cooperative, l = lane in group)
template <int LPR>
__device__ __forceinline__ void plan_long_row(const PlanUpdArgs& a, const rc_plan_row& e, int side, int l) {
  const uint32_t nch = (e.n + a.chunk - 1) / a.chunk;
  uint32_t slot = 0, cbase = 0;
  if (l == 0) {
    slot = atomicAdd(&a.counters[PC_LONG], 1u);
    cbase = atomicAdd(&a.counters[PC_CHUNKS], nch);
    if (slot < a.lw.long_cap) {
      PlanLongRow r;
      r.row = e.row; r.start = e.start; r.n = e.n; r.cbase = cbase; r.nchunks = nch; r.side = (uint32_t)side;
      r.pad0 = r.pad1 = 0;
      a.lw.lrows[slot] = r;
    }
  }
  const int src_lane = (threadIdx.x & 63) / LPR * LPR;
  slot = __shfl(slot, src_lane, 64);
  cbase = __shfl(cbase, src_lane, 64);
  for (uint32_t k = l; k < nch; k += LPR)
    if (cbase + k < a.lw.chunk_cap) {
      PlanChunkInfo c;
      c.lrow = slot;
      c.k = k;
      a.lw.chunks[cbase + k] = c;
    }
}

// SIDE: which table this launch updates.  PLAN_ONLY_OTHER: additionally walk the OTHER side's rows and hand
// its hot rows to the chunk path without updating anything (BPRMF step: user rows are updated later, but their
// hot rows have to be planned before plan_chunk_kernel runs).
template <int D, int MODE>
__device__ __forceinline__ void plan_rows_body(const PlanUpdArgs& a, int side, bool plan_long, uint32_t first_block,
                                               uint32_t n_blocks) {
  constexpr int LPR = D / 4;
  constexpr int GPB = kBlock / LPR;
  constexpr int H = 2;
  const PlanSide& sd = a.side[side];
  const int l = threadIdx.x % LPR;
  const uint32_t nr = *sd.n_rows;
  const uint32_t stride = n_blocks * GPB * H;
  for (uint32_t g0 = ((blockIdx.x - first_block) * GPB + threadIdx.x / LPR) * H; g0 < nr; g0 += stride) {
    rc_plan_row e[H];
    bool act[H];
#pragma unroll
    for (int h = 0; h < H; ++h) {
      e[h] = sd.rows[g0 + h < nr ? g0 + h : g0];
      act[h] = g0 + h < nr && e[h].n <= (uint32_t)kPlanLongSeg;
    }
    float4 w[H], acc[H], s1[H];
    PlanTable tab[H];
    size_t idx4[H];
#pragma unroll
    for (int h = 0; h < H; ++h) tab[h] = plan_lane_table<D>(sd, e[h].row, l, idx4[h]);
#pragma unroll
    for (int h = 0; h < H; ++h)
      if (act[h]) w[h] = load_stream4(reinterpret_cast<const float4*>(tab[h].W) + idx4[h]);
#pragma unroll
    for (int h = 0; h < H; ++h)
      if (act[h]) acc[h] = plan_grad4<D>(sd.g, a.occ, e[h].start, l);
#pragma unroll
    for (int h = 0; h < H; ++h)
      if (act[h] && e[h].n > 1) s1[h] = plan_grad4<D>(sd.g, a.occ, e[h].start + 1, l);
#pragma unroll
    for (int h = 0; h < H; ++h) {
      if (g0 + h >= nr) break;
      if (!act[h]) {
        if (plan_long) plan_long_row<LPR>(a, e[h], side, l);
        continue;
      }
      if (e[h].n > 1) padd4(acc[h], s1[h]);
      for (uint32_t k = 2; k < e[h].n; ++k) padd4(acc[h], plan_grad4<D>(sd.g, a.occ, e[h].start + k, l));
      opt_row4<MODE>(a.o, tab[h].W, tab[h].M, tab[h].V, idx4[h], w[h], acc[h]);
    }
  }
}

// Fast path for sides whose gradient rows are coef[o] * src[src_index[o / div]] for EVERY occurrence (BPRMF item side:
// g[b,c] * U[uid[b]]; no pair mode, no second source).  plan_rows_body walks a chain of four dependent loads per row
// (record -> occ[] -> coef / src_index -> source row) with 8 rows per wave in flight: latency-bound (0.34 ms for the
// 1.4 M multi-occurrence rows of config 2, 2.1 TB/s).  Here the chain is resolved ONE LANE PER ROW -- 64 rows per wave
// travel through the three index levels together -- and only then the lane-groups stream the rows: per step a group
// shuffles the resolved (row, coefficients, source rows) of H rows out of the index lanes and issues the table row,
// its optimizer state and the first two source rows of all H rows back to back (one memory latency per step).
// Summation order per row is unchanged (ascending position: occurrence 0, 1, then the rest) => bit-identical.
template <int D, int MODE>
__device__ __forceinline__ void plan_rows_indexed_body(const PlanUpdArgs& a, int side, bool plan_long, uint32_t first_block,
                                                       uint32_t n_blocks) {
  constexpr int LPR = D / 4;
  constexpr int GPW = 64 / LPR;   // lane-groups per wave
  constexpr int H = 2;            // rows per lane-group per step
#ifndef RC_PLAN_IDX_OCC
#define RC_PLAN_IDX_OCC 4
#endif
  constexpr uint32_t kIdxOcc = RC_PLAN_IDX_OCC;   // occurrences per row resolved in the index phase (2: the round-4 kernel, for A/B)
  const PlanSide& sd = a.side[side];
  const PlanGrad& gr = sd.g;
  const int lane = threadIdx.x & 63;
  const int l = lane % LPR, grp = lane / LPR;
  const uint32_t nr = *sd.n_rows;
  const uint32_t wave = (blockIdx.x - first_block) * (kBlock / 64) + (threadIdx.x >> 6);
  const uint32_t n_waves = n_blocks * (kBlock / 64);
  const float4* src4 = reinterpret_cast<const float4*>(gr.src);
  for (uint32_t base = wave * 64u; base < nr; base += n_waves * 64u) {
    // ---- index phase: lane i resolves row base + i
    const uint32_t gi = base + lane;
    rc_plan_row e;
    e.row = 0; e.start = 0; e.n = 0; e.reserved = 0;
    if (gi < nr) e = sd.rows[gi];
#ifdef RC_X_SKIP_PAIR_ROWS
    // EXPERIMENT ONLY (wrong results): rows with exactly two occurrences are dropped -- the upper bound of what this launch would save
    // if they were resolved inside the fused kernel (profiles/r09_bprmf_pair_ticket_negative.txt)
    if (e.n == 2) e.n = 0;
#endif
    const bool shortrow = e.n >= 1 && e.n <= (uint32_t)kPlanLongSeg;
    // the first FOUR occurrences of the row are resolved here (64 rows' chains together); a row's fifth and later ones walk the
    // chain on their own in the data phase.  (With two, a third occurrence -- 17 % of the multi-occurrence rows of config 2, so
    // 85 % of the eight-row steps had one -- stalled its wave for three dependent loads.)
    uint32_t o0 = 0, o1 = 0, o2 = 0, o3 = 0;
    if (shortrow) {
      o0 = a.occ[e.start];
      if (e.n > 1) o1 = a.occ[e.start + 1];
      if (kIdxOcc > 2 && e.n > 2) o2 = a.occ[e.start + 2];
      if (kIdxOcc > 3 && e.n > 3) o3 = a.occ[e.start + 3];
    }
    float c0 = 1.0f, c1 = 1.0f, c2 = 1.0f, c3 = 1.0f;
    typedef int64_t src_row_t;     // (32-bit source rows: the same 108 registers)
    src_row_t s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    if (shortrow) {
      auto resolve = [&](uint32_t o, float& c, src_row_t& sr) {
        int64_t t = (gr.div == 1) ? (int64_t)o : (int64_t)(o / (uint32_t)gr.div);
        if (gr.src_index) t = gr.src_index[t];
        sr = (src_row_t)t;
        if (gr.coef) c = gr.coef[o];
      };
      resolve(o0, c0, s0);
      if (e.n > 1) resolve(o1, c1, s1);
      if (kIdxOcc > 2 && e.n > 2) resolve(o2, c2, s2);
      if (kIdxOcc > 3 && e.n > 3) resolve(o3, c3, s3);
    }
    // ---- data phase: GPW * H rows per step
    for (int r0 = 0; r0 < 64; r0 += GPW * H) {
      if (base + (uint32_t)r0 >= nr) break;  // wave-uniform
      rc_plan_row eh[H];
      float ch0[H], ch1[H], ch2[H], ch3[H];
      src_row_t sh0[H], sh1[H], sh2[H], sh3[H];
      auto shfl64 = [](int64_t x, int sl) {
        return (int64_t)(((uint64_t)(uint32_t)__shfl((int)(uint32_t)(x >> 32), sl, 64) << 32) | (uint32_t)__shfl((int)(uint32_t)x, sl, 64));
      };
      bool on[H];
#pragma unroll
      for (int h = 0; h < H; ++h) {
        const int sl = r0 + h * GPW + grp;
        eh[h].row = __shfl(e.row, sl, 64);
        eh[h].start = __shfl(e.start, sl, 64);
        eh[h].n = __shfl(e.n, sl, 64);
        eh[h].reserved = 0;
        ch0[h] = __shfl(c0, sl, 64);
        ch1[h] = __shfl(c1, sl, 64);
        ch2[h] = __shfl(c2, sl, 64);
        ch3[h] = __shfl(c3, sl, 64);
        sh0[h] = shfl64(s0, sl);
        sh1[h] = shfl64(s1, sl);
        sh2[h] = shfl64(s2, sl);
        sh3[h] = shfl64(s3, sl);
        on[h] = eh[h].n >= 1 && eh[h].n <= (uint32_t)kPlanLongSeg;
      }
      float4 w[H], m[H], v[H], u0[H], u1[H], u2[H], u3[H];
      size_t idx4[H];
#pragma unroll
      for (int h = 0; h < H; ++h) {
        idx4[h] = (size_t)eh[h].row * LPR + l;
        m[h] = v[h] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (on[h]) {
          w[h] = load_stream4(reinterpret_cast<const float4*>(sd.t.W) + idx4[h]);
          if (mode_has_m(MODE)) m[h] = load_stream4(reinterpret_cast<const float4*>(sd.t.M) + idx4[h]);
          if (mode_has_v(MODE)) v[h] = load_stream4(reinterpret_cast<const float4*>(sd.t.V) + idx4[h]);
          u0[h] = src4[(size_t)sh0[h] * LPR + l];
          if (eh[h].n > 1) u1[h] = src4[(size_t)sh1[h] * LPR + l];
          if (kIdxOcc > 2 && eh[h].n > 2) u2[h] = src4[(size_t)sh2[h] * LPR + l];
          if (kIdxOcc > 3 && eh[h].n > 3) u3[h] = src4[(size_t)sh3[h] * LPR + l];
        }
      }
#pragma unroll
      for (int h = 0; h < H; ++h) {
        if (!on[h]) {
          if (plan_long && eh[h].n > (uint32_t)kPlanLongSeg) plan_long_row<LPR>(a, eh[h], side, l);
          continue;
        }
        float4 acc = u0[h];
        acc.x *= ch0[h]; acc.y *= ch0[h]; acc.z *= ch0[h]; acc.w *= ch0[h];
        if (eh[h].n > 1) {
          float4 t = u1[h];
          t.x *= ch1[h]; t.y *= ch1[h]; t.z *= ch1[h]; t.w *= ch1[h];
          padd4(acc, t);
        }
        if (kIdxOcc > 2 && eh[h].n > 2) {
          float4 t = u2[h];
          t.x *= ch2[h]; t.y *= ch2[h]; t.z *= ch2[h]; t.w *= ch2[h];
          padd4(acc, t);
        }
        if (kIdxOcc > 3 && eh[h].n > 3) {
          float4 t = u3[h];
          t.x *= ch3[h]; t.y *= ch3[h]; t.z *= ch3[h]; t.w *= ch3[h];
          padd4(acc, t);
        }
        // (the later occurrences one after the other: keeping eight of their chains in flight costs 22 registers and a wave per SIMD,
        //  and measured 7 % SLOWER on this launch and on NeuMF's pair update -- profiles/r08_plan_rows_index_four_occurrences.txt)
        for (uint32_t k = kIdxOcc; k < eh[h].n; ++k) padd4(acc, plan_grad4<D>(gr, a.occ, eh[h].start + k, l));
        opt_apply4<MODE>(a.o, w[h], m[h], v[h], acc);
        store_row4(reinterpret_cast<float4*>(sd.t.W) + idx4[h], w[h]);
        if (mode_has_m(MODE)) store_row4(reinterpret_cast<float4*>(sd.t.M) + idx4[h], m[h]);
        if (mode_has_v(MODE)) store_row4(reinterpret_cast<float4*>(sd.t.V) + idx4[h], v[h]);
      }
    }
  }
}

// does `side` qualify for plan_rows_indexed_body?  (kernel-uniform)
__device__ __forceinline__ bool plan_side_indexed(const PlanSide& sd) {
  return !sd.g.pair && sd.g.src != nullptr && sd.g.src2 == nullptr;
}

// hot rows of `side` only: chunk records, no update
template <int D>
__device__ __forceinline__ void plan_long_only_body(const PlanUpdArgs& a, int side, uint32_t first_block,
                                                    uint32_t n_blocks) {
  constexpr int LPR = D / 4;
  constexpr int GPB = kBlock / LPR;
  const PlanSide& sd = a.side[side];
  const int l = threadIdx.x % LPR;
  const uint32_t nr = *sd.n_rows;
  for (uint32_t g = (blockIdx.x - first_block) * GPB + threadIdx.x / LPR; g < nr; g += n_blocks * GPB) {
    const rc_plan_row e = sd.rows[g];
    if (e.n > (uint32_t)kPlanLongSeg) plan_long_row<LPR>(a, e, side, l);
  }
}

// launch 1 of the BPRMF step's update: item rows (side 0) updated, hot rows of BOTH sides planned
template <int D>
__device__ __forceinline__ void plan_chunk_body(const PlanUpdArgs& a, uint32_t first_block, uint32_t n_blocks, float4* part);

// RC_ROWS_WAVES_MAX (experiment switch): cap the waves per SIMD of the row-update kernel, so that a CU keeps wave slots
// free for the plan kernels of the NEXT batch that run beside it on the second stream
#ifdef RC_ROWS_WAVES_MAX
#define RC_ROWS_OCC __attribute__((amdgpu_waves_per_eu(1, RC_ROWS_WAVES_MAX)))
#else
#define RC_ROWS_OCC
#endif
template <int D, int MODE>
__global__ __launch_bounds__(kBlock) RC_ROWS_OCC void plan_rows_kernel(PlanUpdArgs a, uint32_t blocks_main, int update_side,
                                                                      int plan_other, uint32_t blocks_chunk) {
  __shared__ float4 part[kBlock];   // chunk workgroups only
  const bool plan_long = a.long_planned == 0;
  if (blockIdx.x < blocks_main) {
    if (LPR_OK(D) && plan_side_indexed(a.side[update_side])) plan_rows_indexed_body<D, MODE>(a, update_side, plan_long, 0, blocks_main);
    else plan_rows_body<D, MODE>(a, update_side, plan_long, 0, blocks_main);
  } else if (blockIdx.x < blocks_main + blocks_chunk) {
    plan_chunk_body<D>(a, blocks_main, blocks_chunk, part);   // hot rows of BOTH sides, registered by the plan
  } else if (plan_other) plan_long_only_body<D>(a, 1 - update_side, blocks_main + blocks_chunk, gridDim.x - blocks_main - blocks_chunk);
}

// LDS tree over the lane-groups of a block, fixed order; result in group 0's slots
template <int LPR>
__device__ __forceinline__ void plan_tree_sum(float4* part, int g) {
  constexpr int GPB = kBlock / LPR;
#pragma unroll
  for (int off = GPB / 2; off >= 1; off >>= 1) {
    __syncthreads();
    if (g < off) {
      float4 x = part[threadIdx.x];
      padd4(x, part[threadIdx.x + off * LPR]);
      part[threadIdx.x] = x;
    }
  }
  __syncthreads();
}

template <int D>
__device__ __forceinline__ void plan_chunk_body(const PlanUpdArgs& a, uint32_t first_block, uint32_t n_blocks, float4* part) {
  constexpr int LPR = D / 4;
  constexpr int GPB = kBlock / LPR;
  const int l = threadIdx.x % LPR;
  const int g = threadIdx.x / LPR;
  uint32_t n_chunks = a.counters[PC_CHUNKS];
  if (n_chunks > a.lw.chunk_cap) n_chunks = a.lw.chunk_cap;
  for (uint32_t c = blockIdx.x - first_block; c < n_chunks; c += n_blocks) {
    const PlanChunkInfo ci = a.lw.chunks[c];
    const PlanLongRow r = a.lw.lrows[ci.lrow];
    const PlanGrad& gs = a.side[r.side].g;
    const uint32_t start = r.start + ci.k * a.chunk;
    const uint32_t end = (start + a.chunk < r.start + r.n) ? start + a.chunk : r.start + r.n;
    // four independent occurrences per lane-group per trip (fixed pattern -> fixed order)
    float4 acc = make_float4(0, 0, 0, 0);
    for (uint32_t jj = start + g; jj < end; jj += 4 * GPB) {
      const float4 z = make_float4(0, 0, 0, 0);
      const float4 s0 = plan_grad4<D>(gs, a.occ, jj, l);
      const float4 s1 = (jj + GPB < end) ? plan_grad4<D>(gs, a.occ, jj + GPB, l) : z;
      const float4 s2 = (jj + 2 * GPB < end) ? plan_grad4<D>(gs, a.occ, jj + 2 * GPB, l) : z;
      const float4 s3 = (jj + 3 * GPB < end) ? plan_grad4<D>(gs, a.occ, jj + 3 * GPB, l) : z;
      padd4(acc, s0); padd4(acc, s1); padd4(acc, s2); padd4(acc, s3);
    }
    part[threadIdx.x] = acc;
    plan_tree_sum<LPR>(part, g);
    if (g == 0) reinterpret_cast<float4*>(a.lw.partial)[(size_t)(r.cbase + ci.k) * LPR + l] = part[threadIdx.x];
    __syncthreads();  // part[] is reused by the next chunk
  }
}

template <int D>
__global__ __launch_bounds__(kBlock) void plan_chunk_kernel(PlanUpdArgs a) {
  __shared__ float4 part[kBlock];
  plan_chunk_body<D>(a, 0, gridDim.x, part);
}

// last launch: short rows of `update_side` (BPRMF step: the user table, whose pre-step rows every earlier launch
// has finished reading), the hot rows of both sides from their chunk partials, and the loss mean
template <int D, int MODE>
__global__ __launch_bounds__(kBlock) void plan_final_kernel(PlanUpdArgs a, uint32_t blocks_rows, uint32_t blocks_long,
                                                           int update_side) {
  constexpr int LPR = D / 4;
  constexpr int GPB = kBlock / LPR;
  __shared__ float4 part[kBlock];
  if (blockIdx.x < blocks_rows) {
    if (update_side >= 0) plan_rows_body<D, MODE>(a, update_side, false, 0, blocks_rows);
    return;
  }
  if (blockIdx.x < blocks_rows + blocks_long) {
    const int l = threadIdx.x % LPR;
    const int g = threadIdx.x / LPR;
    uint32_t n_long = a.counters[PC_LONG];
    if (n_long > a.lw.long_cap) n_long = a.lw.long_cap;
    for (uint32_t i = blockIdx.x - blocks_rows; i < n_long; i += blocks_long) {
      const PlanLongRow r = a.lw.lrows[i];
      float4 acc = make_float4(0, 0, 0, 0);
      for (uint32_t k = g; k < r.nchunks; k += GPB)
        if (r.cbase + k < a.lw.chunk_cap)
          padd4(acc, reinterpret_cast<const float4*>(a.lw.partial)[(size_t)(r.cbase + k) * LPR + l]);
      part[threadIdx.x] = acc;
      plan_tree_sum<LPR>(part, g);
      if (g == 0) {
        size_t idx;
        const PlanTable t = plan_lane_table<D>(a.side[r.side], r.row, l, idx);
        opt_row4<MODE>(a.o, t.W, t.M, t.V, idx, load_stream4(reinterpret_cast<const float4*>(t.W) + idx), part[threadIdx.x]);
      }
      __syncthreads();
    }
    return;
  }
  // loss mean: thread t sums elements t, t+256, ... then a fixed LDS tree (same order as reduce_sum_kernel)
  if (a.loss_out == nullptr) return;
  float* sm = reinterpret_cast<float*>(part);
  const float acc = fixed_order_partial<kBlock, false>(a.loss_vec, a.loss_n, (int)threadIdx.x);
  sm[threadIdx.x] = acc;
  __syncthreads();
  for (int off = kBlock / 2; off >= 1; off >>= 1) {
    if ((int)threadIdx.x < off) sm[threadIdx.x] += sm[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) a.loss_out[0] = sm[0] * a.loss_scale;
}

// one table (or table pair) updated from its row list: rows, chunks of hot rows, hot rows
template <int D, int MODE>
static int launch_side_update(const PlanUpdArgs& a_in, int64_t n_occ, hipStream_t s) {
  PlanUpdArgs a = a_in;
  a.chunk = side_chunk<D>();
  const uint32_t cus = (uint32_t)device_cus();
  const uint32_t blocks_main = cus * 32;
  hipLaunchKernelGGL((plan_rows_kernel<D, MODE>), dim3(blocks_main), dim3(kBlock), 0, s, a, blocks_main, 0, 0, 0u);
  RC_LAUNCH_CHECK();
  if (n_occ > kPlanLongSeg) {
    hipLaunchKernelGGL((plan_chunk_kernel<D>), dim3(1024), dim3(kBlock), 0, s, a);
    RC_LAUNCH_CHECK();
    hipLaunchKernelGGL((plan_final_kernel<D, MODE>), dim3(256 + 1), dim3(kBlock), 0, s, a, 0u, 256u, -1);
    RC_LAUNCH_CHECK();
  }
  return RC_OK;
}

template <int MODE>
static int launch_side_update_d(const PlanUpdArgs& a, int d, int64_t n_occ, hipStream_t s) {
  switch (d) {
    case 16: return launch_side_update<16, MODE>(a, n_occ, s);
    case 32: return launch_side_update<32, MODE>(a, n_occ, s);
    case 64: return launch_side_update<64, MODE>(a, n_occ, s);
    case 128: return launch_side_update<128, MODE>(a, n_occ, s);
    case 256: return launch_side_update<256, MODE>(a, n_occ, s);
    default: return fail(RC_ERR_UNSUPPORTED, "rc_plan_update: row width %d floats (16/32/64/128/256)", d);
  }
}

template <int D, int MODE>
static int launch_step_updates(const PlanUpdArgs& a, int64_t n_occ, hipStream_t s, hipEvent_t* ev_items_done) {
  const uint32_t cus = (uint32_t)device_cus();
  // grid-stride over the rows with 4x more workgroups than fit at once: a grid of exactly "8 per CU" ran in two
  // rounds whenever the kernel's registers allowed only 7 (Adam: 72 VGPRs -> item update 2.05 instead of 1.5 ms)
  const uint32_t blocks_main = cus * 32;
  if (a.long_planned) {
    // hot rows came with the plan: their chunk sums ride in this launch (workgroups behind the row workgroups; they only
    // read pre-step rows and write the partial buffer, so they need no ordering against the row updates)
    const uint32_t blocks_chunk = n_occ > kPlanLongSeg ? 512 : 0;
    hipLaunchKernelGGL((plan_rows_kernel<D, MODE>), dim3(blocks_main + blocks_chunk), dim3(kBlock), 0, s, a, blocks_main, 0, 0,
                       blocks_chunk);
    RC_LAUNCH_CHECK();
  } else {
    const uint32_t blocks_other = 64;
    hipLaunchKernelGGL((plan_rows_kernel<D, MODE>), dim3(blocks_main + blocks_other), dim3(kBlock), 0, s, a, blocks_main, 0, 1, 0u);
    RC_LAUNCH_CHECK();
    if (n_occ > kPlanLongSeg) {  // otherwise no row can be hot
      hipLaunchKernelGGL((plan_chunk_kernel<D>), dim3(1024), dim3(kBlock), 0, s, a);
      RC_LAUNCH_CHECK();
    }
  }
  if (ev_items_done) RC_HIP(hipEventRecord(*ev_items_done, s));
  const uint32_t blocks_rows = cus * 2;
  const uint32_t blocks_long = n_occ > kPlanLongSeg ? 256 : 0;
  hipLaunchKernelGGL((plan_final_kernel<D, MODE>), dim3(blocks_rows + blocks_long + 1), dim3(kBlock), 0, s, a, blocks_rows,
                     blocks_long, 1);
  RC_LAUNCH_CHECK();
  return RC_OK;
}

template <int MODE>
static int launch_step_updates_d(const PlanUpdArgs& a, int d, int64_t n_occ, hipStream_t s, hipEvent_t* ev) {
  switch (d) {
    case 16: return launch_step_updates<16, MODE>(a, n_occ, s, ev);
    case 32: return launch_step_updates<32, MODE>(a, n_occ, s, ev);
    case 64: return launch_step_updates<64, MODE>(a, n_occ, s, ev);
    case 128: return launch_step_updates<128, MODE>(a, n_occ, s, ev);
    default: return fail(RC_ERR_UNSUPPORTED, "plan update: d=%d (16/32/64/128)", d);
  }
}

// The three update launches of a BPRMF step (train_step.hip): item rows, chunks of hot rows, user rows + hot rows
// + loss mean.  Item-side gradients read pre-step U rows, so every read of U precedes the last launch.
int plan_bprmf_step_updates(float* U, float* mU, float* vU, float* I, float* mI, float* vI, int d, const int64_t* uid,
                            int C, int64_t n_i, int64_t B, const float* gpred, const float* ugrad,
                            const rc_plan_row* rows_i, const uint32_t* n_rows_i, const rc_plan_row* rows_u,
                            const uint32_t* n_rows_u, const uint32_t* occ, uint32_t* counters,
                            const PlanLongWs& lw, bool long_planned,
                            const rc_opt_hyper* h, const float* loss_vec, float loss_scale, float* loss_out,
                            hipStream_t s, hipEvent_t* ev_items_done) {
  PlanUpdArgs a;
  memset(&a, 0, sizeof(a));
  a.long_planned = long_planned ? 1 : 0;
  a.chunk = kPlanChunk;
  RC_TRY(fill_opt_scalars(h, &a.o));
  const int mode = mode_of(h);
  RC_REQUIRE(mode != MODE_ADAM || (mU && vU && mI && vI), "rc_bprmf_train_step: Adam needs m and v tables");
  RC_REQUIRE(mode != MODE_ADAGRAD || (mU && mI), "rc_bprmf_train_step: Adagrad needs the state_sum tables");
  a.side[0].t = PlanTable{I, mI, vI};
  a.side[0].g = PlanGrad{gpred, U, uid, C, nullptr, 0xFFFFFFFFu, nullptr, 0};
  a.side[0].rows = rows_i;
  a.side[0].n_rows = n_rows_i;
  a.side[1].t = PlanTable{U, mU, vU};
  a.side[1].g = PlanGrad{nullptr, nullptr, nullptr, 1, ugrad, (uint32_t)n_i, nullptr, 0};  // user occurrence p = n_i + b -> ugrad[b]
  a.side[1].rows = rows_u;
  a.side[1].n_rows = n_rows_u;
  a.occ = occ;
  a.counters = counters;
  a.lw = lw;
  a.loss_vec = loss_vec;
  a.loss_n = B;
  a.loss_scale = loss_scale;
  a.loss_out = loss_out;
  switch (mode) {
    case MODE_SGD: return launch_step_updates_d<MODE_SGD>(a, d, n_i + B, s, ev_items_done);
    case MODE_ADAM: return launch_step_updates_d<MODE_ADAM>(a, d, n_i + B, s, ev_items_done);
    default: return launch_step_updates_d<MODE_ADAGRAD>(a, d, n_i + B, s, ev_items_done);
  }
}

}  // namespace rc

using namespace rc;

namespace {
struct UpdWs {
  uint32_t* counters;
  PlanLongWs lw;
  size_t total;
};
UpdWs carve_upd_ws(void* base, int64_t n_occ, int d) {
  Carver cv(base);
  UpdWs w;
  w.counters = cv.take<uint32_t>(PC_N);
  const PlanLongWs lw = carve_plan_long_ws(base ? reinterpret_cast<char*>(base) + cv.off : nullptr, n_occ, d);
  w.lw = lw;
  w.total = cv.off + lw.total;
  return w;
}

// h == nullptr: no optimizer, the listed rows of "W" receive the summed gradient rows (MODE_DENSE_GRAD)
// zeroed_counters: PC_N words the caller has zero-filled where it cost nothing (beside other work, on another stream) and that nobody
// has used since -- the memset in front of the update (a launch of its own on the step's critical path) is then left out
int run_side_update(PlanUpdArgs& a, const rc_opt_hyper* h, int d_eff, int64_t n_occ, void* ws, size_t ws_bytes,
                    rc_stream_t stream, const char* who, uint32_t* zeroed_counters = nullptr) {
  const UpdWs w = carve_upd_ws(ws, n_occ, d_eff);
  if (ws_bytes < w.total) return fail(RC_ERR_WORKSPACE, "%s: workspace %zu < %zu", who, ws_bytes, w.total);
  if (h != nullptr) RC_TRY(fill_opt_scalars(h, &a.o));
  a.counters = zeroed_counters != nullptr ? zeroed_counters : w.counters;
  a.lw = w.lw;
  hipStream_t s = as_stream(stream);
  if (zeroed_counters == nullptr) RC_HIP(hipMemsetAsync(w.counters, 0, PC_N * sizeof(uint32_t), s));
  if (h == nullptr) return launch_side_update_d<MODE_DENSE_GRAD>(a, d_eff, n_occ, s);
  switch (mode_of(h)) {
    case MODE_SGD: return launch_side_update_d<MODE_SGD>(a, d_eff, n_occ, s);
    case MODE_ADAM: return launch_side_update_d<MODE_ADAM>(a, d_eff, n_occ, s);
    default: return launch_side_update_d<MODE_ADAGRAD>(a, d_eff, n_occ, s);
  }
}
bool al16(const void* p) { return reinterpret_cast<uintptr_t>(p) % 16 == 0; }
}  // namespace

extern "C" size_t rc_plan_update_workspace_bytes(int64_t n_occ, int d) {
  if (n_occ < 1) n_occ = 1;
  if (d < 1) d = 1;
  return carve_upd_ws(nullptr, n_occ, d).total;
}

extern "C" int rc_plan_update(float* W, float* m, float* v, int d, const rc_plan_row* rows, const uint32_t* n_rows,
                              const uint32_t* occ, int64_t n_occ, const float* coef, const float* src,
                              const int64_t* src_index, int div, const float* src2, int64_t n_split,
                              const rc_opt_hyper* h, void* ws, size_t ws_bytes, rc_stream_t stream) {
  if (n_occ == 0) return RC_OK;
  RC_REQUIRE(W && rows && n_rows && occ && h && ws, "rc_plan_update: null pointer");
  RC_REQUIRE(n_occ > 0 && n_occ < ((int64_t)1 << 31) && div >= 1 && n_split >= 0 && n_split <= n_occ,
             "rc_plan_update: bad sizes n_occ=%lld div=%d n_split=%lld", (long long)n_occ, div, (long long)n_split);
  RC_REQUIRE(src || src2, "rc_plan_update: gradient source missing (src for positions < n_split, src2 for the others)");
  RC_REQUIRE(mode_of(h) != MODE_ADAM || (m && v), "rc_plan_update: Adam needs m and v");
  RC_REQUIRE(mode_of(h) != MODE_ADAGRAD || m, "rc_plan_update: Adagrad needs m (state_sum)");
  RC_REQUIRE(al16(W) && al16(m) && al16(v) && al16(src) && al16(src2), "rc_plan_update: buffers must be 16-byte aligned");
  PlanUpdArgs a;
  memset(&a, 0, sizeof(a));
  a.side[0].t = PlanTable{W, m, v};
  a.side[0].g = PlanGrad{coef, src, src_index, div, src2, (uint32_t)n_split, nullptr, 0};
  a.side[0].rows = rows;
  a.side[0].n_rows = n_rows;
  a.occ = occ;
  return run_side_update(a, h, d, n_occ, ws, ws_bytes, stream, "rc_plan_update");
}

static int plan_update_pair_impl(uint32_t* zeroed_counters, int64_t src_ld, float* W_a, float* m_a, float* v_a, float* W_b, float* m_b, float* v_b, int d,
                                   const rc_plan_row* rows, const uint32_t* n_rows, const uint32_t* occ, int64_t n_occ,
                                   const float* src_a, const float* src_b, int64_t occ_base, const rc_opt_hyper* h,
                                   void* ws, size_t ws_bytes, rc_stream_t stream) {
  if (n_occ == 0) return RC_OK;
  RC_REQUIRE(W_a && W_b && rows && n_rows && occ && src_a && src_b && h && ws, "rc_plan_update_pair: null pointer");
  RC_REQUIRE(n_occ > 0 && n_occ < ((int64_t)1 << 31) && occ_base >= 0 && occ_base < ((int64_t)1 << 31),
             "rc_plan_update_pair: bad sizes");
  RC_REQUIRE(mode_of(h) != MODE_ADAM || (m_a && v_a && m_b && v_b), "rc_plan_update_pair: Adam needs m and v");
  RC_REQUIRE(mode_of(h) != MODE_ADAGRAD || (m_a && m_b), "rc_plan_update_pair: Adagrad needs m (state_sum)");
  RC_REQUIRE(al16(W_a) && al16(W_b) && al16(m_a) && al16(m_b) && al16(v_a) && al16(v_b) && al16(src_a) && al16(src_b),
             "rc_plan_update_pair: buffers must be 16-byte aligned");
  if (d != 8 && d != 16 && d != 32 && d != 64 && d != 128)
    return fail(RC_ERR_UNSUPPORTED, "rc_plan_update_pair: d=%d (2 d must be 16/32/64/128/256)", d);
  PlanUpdArgs a;
  memset(&a, 0, sizeof(a));
  a.side[0].t = PlanTable{W_a, m_a, v_a};
  a.side[0].tb = PlanTable{W_b, m_b, v_b};
  a.side[0].g = PlanGrad{nullptr, nullptr, nullptr, 1, src_a, (uint32_t)occ_base, src_b, 1, (uint32_t)(src_ld / 4)};
  a.side[0].rows = rows;
  a.side[0].n_rows = n_rows;
  a.occ = occ;
  return run_side_update(a, h, 2 * d, n_occ, ws, ws_bytes, stream, "rc_plan_update_pair", zeroed_counters);
}

extern "C" int rc_plan_update_pair(float* W_a, float* m_a, float* v_a, float* W_b, float* m_b, float* v_b, int d,
                                   const rc_plan_row* rows, const uint32_t* n_rows, const uint32_t* occ, int64_t n_occ,
                                   const float* src_a, const float* src_b, int64_t occ_base, const rc_opt_hyper* h,
                                   void* ws, size_t ws_bytes, rc_stream_t stream) {
  return plan_update_pair_impl(nullptr, 0, W_a, m_a, v_a, W_b, m_b, v_b, d, rows, n_rows, occ, n_occ, src_a, src_b, occ_base, h, ws, ws_bytes, stream);
}

/* rc_plan_update_pair with both gradient sources in ONE block: occurrence o reads src_block[o - occ_base, 0 .. d) for table a and
 * [d .. 2 d) for table b, rows src_ld floats apart (a multiple of 4, >= 2 d) -- the (d mf | d mlp) rows a row-sharded NeuMF rank
 * receives from the others, used where they lie instead of through two contiguous copies.  counters: as rc_plan_update_pair_zeroed,
 * or NULL (the call zeroes its own). */
extern "C" int rc_plan_update_pair_block(float* W_a, float* m_a, float* v_a, float* W_b, float* m_b, float* v_b, int d,
                                         const rc_plan_row* rows, const uint32_t* n_rows, const uint32_t* occ, int64_t n_occ,
                                         const float* src_block, int64_t src_ld, int64_t occ_base, const rc_opt_hyper* h,
                                         uint32_t* counters, void* ws, size_t ws_bytes, rc_stream_t stream) {
  RC_REQUIRE(src_block != nullptr && src_ld >= 2 * (int64_t)d && src_ld % 4 == 0 && src_ld / 4 < ((int64_t)1 << 32),
             "rc_plan_update_pair_block: src_ld = %lld floats (a multiple of 4, at least 2 d)", (long long)src_ld);
  return plan_update_pair_impl(counters, src_ld, W_a, m_a, v_a, W_b, m_b, v_b, d, rows, n_rows, occ, n_occ, src_block, src_block + d, occ_base, h, ws,
                               ws_bytes, stream);
}

/* rc_plan_update_pair with the update's eight ticket counters supplied by the caller, ZERO-FILLED where that cost nothing (with the
 * plan's own buffers, beside other work on another stream) and used by nobody since: the one-launch memset in front of the update --
 * on the critical path of a step between its fused kernel and its table updates -- is left out.  counters: 8 uint32, device. */
extern "C" int rc_plan_update_pair_zeroed(float* W_a, float* m_a, float* v_a, float* W_b, float* m_b, float* v_b, int d,
                                          const rc_plan_row* rows, const uint32_t* n_rows, const uint32_t* occ, int64_t n_occ,
                                          const float* src_a, const float* src_b, int64_t occ_base, const rc_opt_hyper* h,
                                          uint32_t* counters, void* ws, size_t ws_bytes, rc_stream_t stream) {
  RC_REQUIRE(counters != nullptr, "rc_plan_update_pair_zeroed: null pointer");
  return plan_update_pair_impl(counters, 0, W_a, m_a, v_a, W_b, m_b, v_b, d, rows, n_rows, occ, n_occ, src_a, src_b, occ_base, h, ws, ws_bytes, stream);
}

/* out[row, :] = sum over the row's occurrences of their gradient rows, for the rows the plan lists (other rows of `out`
 * are left as they are): aten::embedding_dense_backward's index_add as a plan consumer, no optimizer. */
extern "C" int rc_plan_row_sums(float* out, int d, const rc_plan_row* rows, const uint32_t* n_rows, const uint32_t* occ,
                                int64_t n_occ, const float* coef, const float* src, const int64_t* src_index, int div,
                                const float* src2, int64_t n_split, void* ws, size_t ws_bytes, rc_stream_t stream) {
  if (n_occ == 0) return RC_OK;
  RC_REQUIRE(out && rows && n_rows && occ && ws, "rc_plan_row_sums: null pointer");
  RC_REQUIRE(n_occ > 0 && n_occ < ((int64_t)1 << 31) && div >= 1 && n_split >= 0 && n_split <= n_occ,
             "rc_plan_row_sums: bad sizes n_occ=%lld div=%d n_split=%lld", (long long)n_occ, div, (long long)n_split);
  RC_REQUIRE(src || src2, "rc_plan_row_sums: gradient source missing (src for positions < n_split, src2 for the others)");
  RC_REQUIRE(al16(out) && al16(src) && al16(src2), "rc_plan_row_sums: buffers must be 16-byte aligned");
  PlanUpdArgs a;
  memset(&a, 0, sizeof(a));
  a.side[0].t = PlanTable{out, nullptr, nullptr};
  a.side[0].g = PlanGrad{coef, src, src_index, div, src2, (uint32_t)n_split, nullptr, 0};
  a.side[0].rows = rows;
  a.side[0].n_rows = n_rows;
  a.occ = occ;
  return run_side_update(a, nullptr, d, n_occ, ws, ws_bytes, stream, "rc_plan_row_sums");
}

namespace rc {
// one thread per listed row: its id, and its record index at every position that touches it
__global__ __launch_bounds__(kBlock) void plan_distinct_kernel(const rc_plan_row* __restrict__ rows, const uint32_t* __restrict__ n_rows,
                                                               const uint32_t* __restrict__ occ, uint32_t occ_base,
                                                               int64_t* __restrict__ uniq, int64_t* __restrict__ inverse) {
  const uint32_t nr = *n_rows;
  for (uint32_t r = blockIdx.x * kBlock + threadIdx.x; r < nr; r += gridDim.x * kBlock) {
    const rc_plan_row e = rows[r];
    uniq[r] = (int64_t)e.row;
    for (uint32_t k = 0; k < e.n; ++k) inverse[occ[e.start + k] - occ_base] = (int64_t)r;
  }
}
}  // namespace rc

/* The distinct ids of a planned list and the inverse index (torch.unique(return_inverse=True) without the sort: the
 * order of `uniq` is the plan's record order): uniq[r] = id of record r, inverse[p - occ_base] = r for every position p
 * of the record.  The count stays on the device (*n_rows).  uniq: capacity = list length. */
extern "C" int rc_plan_distinct(const rc_plan_row* rows, const uint32_t* n_rows, const uint32_t* occ, int64_t occ_base,
                                int64_t n_list, int64_t* uniq, int64_t* inverse, rc_stream_t stream) {
  if (n_list == 0) return RC_OK;
  RC_REQUIRE(rows && n_rows && occ && uniq && inverse, "rc_plan_distinct: null pointer");
  RC_REQUIRE(n_list > 0 && occ_base >= 0 && occ_base < ((int64_t)1 << 31), "rc_plan_distinct: bad sizes");
  const int64_t blocks = (n_list + kBlock - 1) / kBlock;
  hipLaunchKernelGGL(plan_distinct_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(kBlock), 0, as_stream(stream),
                     rows, n_rows, occ, (uint32_t)occ_base, uniq, inverse);
  RC_LAUNCH_CHECK();
  return RC_OK;
}
