// bench_mix.hip -- a measurement aid, not a step of the path: the ACCESS MIX of the fused BPRMF kernel without its
// arithmetic, on the caller's own table and id list, so that bench.py can print what THIS box delivers for that mix next to
// what the kernel reaches (roofline.box_ceiling_gbps / frac_of_box_ceiling).  Rounds 2-3 quoted a ceiling measured once,
// on another box of the pool (tools/hbm_ceiling.hip, profiles/r02b_hbm_ceiling.json); boxes differ by +-7 %.
//
// The mix: every occurrence reads its table row (d floats, one lane-group per row, R rows in flight per group, non-temporal
// like the kernel's candidate loads), a pseudo-random fraction `write_frac` of the occurrences writes its row back
// UNCHANGED (the fused kernel writes back the single-occurrence rows: pass their share of the occurrences), the ids are
// read as int64 like the reference's batches (models/BaseModel.py:198).  The table keeps its contents.
#include "common.hpp"
#include "opt_math.hpp"

namespace rc {

template <int LPR, int R>
__global__ __launch_bounds__(256) void bench_mix_kernel(float* __restrict__ tab, const int64_t* __restrict__ ids, int64_t n_occ,
                                                        uint32_t wthresh, float* sink) {
  const int l = threadIdx.x % LPR;
  const int64_t grp = ((int64_t)blockIdx.x * 256 + threadIdx.x) / LPR;
  const int64_t o0 = grp * R;
  if (o0 >= n_occ) return;
  int64_t id[R];
  float4 r[R];
#pragma unroll
  for (int k = 0; k < R; ++k) id[k] = ids[o0 + k < n_occ ? o0 + k : o0];
#pragma unroll
  for (int k = 0; k < R; ++k) r[k] = load_stream4(reinterpret_cast<const float4*>(tab + id[k] * (4 * LPR)) + l);
  float acc = 0.f;
#pragma unroll
  for (int k = 0; k < R; ++k) acc += r[k].x + r[k].w;
#pragma unroll
  for (int k = 0; k < R; ++k) {
    const uint32_t h = (uint32_t)(o0 + k) * 2654435761u;   // which occurrences write back: a hash of the position
    if (h < wthresh && o0 + k < n_occ) store_row4(reinterpret_cast<float4*>(tab + id[k] * (4 * LPR)) + l, r[k]);
  }
  if (acc == 12345.678f) sink[0] = acc;   // keeps the loads alive
}

template <int LPR>
static void launch_mix(float* tab, const int64_t* ids, int64_t n_occ, uint32_t wthresh, float* sink, hipStream_t s) {
  constexpr int R = 8;
  const int64_t groups = (n_occ + R - 1) / R;
  const int64_t blocks = (groups * LPR + 255) / 256;
  hipLaunchKernelGGL((bench_mix_kernel<LPR, R>), dim3((unsigned)blocks), dim3(256), 0, s, tab, ids, n_occ, wthresh, sink);
}

// The matrix pipes' sustained fp32 rate on this box: every SIMD of the chip issues v_mfma_f32_32x32x2_f32 back to back from two
// waves, four independent accumulators per wave (no operand traffic at all: the operands are two registers).  What bench.py prints as
// roofline.box_mfma_tflops next to the datasheet's 157.3: clocks under a sustained matrix load are below the peak clock.
typedef float bm_f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void bench_mfma_kernel(int iters, float a0, float b0, float* sink) {
  bm_f32x16 acc[4];
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;
  float a = a0 + (float)(threadIdx.x & 7), b = b0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, acc[1], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, a, acc[2], 0, 0, 0);
      acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(b, b, acc[3], 0, 0, 0);
    }
  }
  float t = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int r = 0; r < 16; ++r) t += acc[k][r];
  if (t == 12345.678f) sink[0] = t;   // keeps the accumulators alive
}

int device_cus();   // bucket_plan.hip

}  // namespace rc

using namespace rc;

extern "C" int rc_bench_mfma(int iters, float* sink_dev, float* tflops_out, rc_stream_t stream) {
  RC_REQUIRE(sink_dev && tflops_out && iters >= 1, "rc_bench_mfma: bad arguments");
  hipStream_t s = as_stream(stream);
  const int wgs = device_cus() * 2;   // two waves per SIMD
  hipEvent_t a, b;
  RC_HIP(hipEventCreate(&a));
  RC_HIP(hipEventCreate(&b));
  hipLaunchKernelGGL(bench_mfma_kernel, dim3(wgs), dim3(256), 0, s, iters, 1.0f, 0.5f, sink_dev);   // warm-up (clocks)
  RC_HIP(hipEventRecord(a, s));
  hipLaunchKernelGGL(bench_mfma_kernel, dim3(wgs), dim3(256), 0, s, iters, 1.0f, 0.5f, sink_dev);
  RC_HIP(hipEventRecord(b, s));
  RC_HIP(hipEventSynchronize(b));
  float ms = 0.f;
  RC_HIP(hipEventElapsedTime(&ms, a, b));
  hipEventDestroy(a);
  hipEventDestroy(b);
  RC_LAUNCH_CHECK();
  const double flops = (double)wgs * 4.0 * (double)iters * 16.0 * 4096.0;   // waves x MFMAs x 32 * 32 * 2 * 2
  *tflops_out = (float)(flops / ((double)ms * 1e-3) / 1e12);
  return RC_OK;
}

extern "C" int rc_bench_mix(float* table, int d, const int64_t* ids, int64_t n_occ, float write_frac, int iters, float* sink_dev,
                            float* ms_out, rc_stream_t stream) {
  RC_REQUIRE(table && ids && sink_dev && ms_out, "rc_bench_mix: null pointer");
  RC_REQUIRE(d == 32 || d == 64 || d == 128, "rc_bench_mix: d=%d (32, 64 or 128)", d);
  RC_REQUIRE(n_occ >= 1 && iters >= 1 && write_frac >= 0.f && write_frac <= 1.f, "rc_bench_mix: bad arguments");
  hipStream_t s = as_stream(stream);
  const uint32_t wthresh = write_frac >= 1.f ? 0xFFFFFFFFu : (uint32_t)((double)write_frac * 4294967296.0);
  auto go = [&] {
    if (d == 32) launch_mix<8>(table, ids, n_occ, wthresh, sink_dev, s);
    else if (d == 64) launch_mix<16>(table, ids, n_occ, wthresh, sink_dev, s);
    else launch_mix<32>(table, ids, n_occ, wthresh, sink_dev, s);
  };
  hipEvent_t a, b;
  RC_HIP(hipEventCreate(&a));
  RC_HIP(hipEventCreate(&b));
  for (int k = 0; k < 3; ++k) go();
  RC_HIP(hipEventRecord(a, s));
  for (int k = 0; k < iters; ++k) go();
  RC_HIP(hipEventRecord(b, s));
  RC_HIP(hipEventSynchronize(b));
  float ms = 0.f;
  RC_HIP(hipEventElapsedTime(&ms, a, b));
  hipEventDestroy(a);
  hipEventDestroy(b);
  RC_LAUNCH_CHECK();
  *ms_out = ms / iters;
  return RC_OK;
}
