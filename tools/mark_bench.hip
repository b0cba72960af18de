// mark_bench.hip -- how fast can a batch's item ids be classified "occurs once / occurs several times" with two
// bitmaps and integer atomics (seen |= bit; if it was set: multi |= bit)?  6.55 M uniform ids over 10 M rows is the
// BPRMF config-2 step (models/BaseModel.py:207 negatives).  Stand-alone, prints one JSON line.
//   hipcc -O3 --offload-arch=gfx950 tools/mark_bench.hip -o tools/bin/mark_bench
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int PER>
__global__ __launch_bounds__(256) void mark_kernel(const int64_t* __restrict__ ids, uint32_t n, uint32_t* seen, uint32_t* multi) {
  const uint32_t base = (blockIdx.x * 256u + threadIdx.x);
  const uint32_t stride = gridDim.x * 256u;
  int64_t id[PER];
#pragma unroll
  for (int k = 0; k < PER; ++k) { const uint32_t p = base + k * stride; id[k] = p < n ? ids[p] : -1; }
  uint32_t old[PER];
#pragma unroll
  for (int k = 0; k < PER; ++k) if (id[k] >= 0) old[k] = atomicOr(&seen[id[k] >> 5], 1u << (id[k] & 31));
#pragma unroll
  for (int k = 0; k < PER; ++k) if (id[k] >= 0 && (old[k] >> (id[k] & 31) & 1u)) atomicOr(&multi[id[k] >> 5], 1u << (id[k] & 31));
}

// byte cells instead of bits: seen[id] = 1 via atomicAdd on the containing word (returns the count so far)
__global__ __launch_bounds__(256) void lookup_kernel(const int64_t* __restrict__ ids, uint32_t n, const uint32_t* __restrict__ multi, uint32_t* out) {
  const uint32_t p = blockIdx.x * 256u + threadIdx.x;
  uint32_t acc = 0;
  if (p < n) { const int64_t id = ids[p]; acc = (multi[id >> 5] >> (id & 31)) & 1u; }
  acc = __popcll(__ballot(acc));
  if ((threadIdx.x & 63) == 0 && acc) atomicAdd(out, acc);
}

int main(int argc, char** argv) {
  const uint32_t rows = argc > 1 ? atoi(argv[1]) : 10000001u;
  const uint32_t n = argc > 2 ? atoi(argv[2]) : 6553600u;
  std::vector<int64_t> h(n);
  uint64_t s = 88172645463325252ull;
  for (uint32_t i = 0; i < n; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i] = 1 + (int64_t)(s % (rows - 1)); }
  int64_t* ids; uint32_t *seen, *multi, *out;
  const size_t words = (rows + 31) / 32;
  CK(hipMalloc(&ids, n * 8)); CK(hipMalloc(&seen, words * 4)); CK(hipMalloc(&multi, words * 4)); CK(hipMalloc(&out, 4));
  CK(hipMemcpy(ids, h.data(), n * 8, hipMemcpyHostToDevice));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto run = [&](int per, int iters) {
    float best = 1e9f, tot = 0;
    for (int it = 0; it < iters; ++it) {
      CK(hipMemsetAsync(seen, 0, words * 4)); CK(hipMemsetAsync(multi, 0, words * 4));
      CK(hipEventRecord(e0));
      const uint32_t blocks = (n + 256 * per - 1) / (256 * per);
      if (per == 1) mark_kernel<1><<<blocks, 256>>>(ids, n, seen, multi);
      else if (per == 4) mark_kernel<4><<<blocks, 256>>>(ids, n, seen, multi);
      else mark_kernel<8><<<blocks, 256>>>(ids, n, seen, multi);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (it) { tot += ms; if (ms < best) best = ms; }
    }
    return best;
  };
  const float t1 = run(1, 6), t4 = run(4, 6), t8 = run(8, 6);
  CK(hipMemset(out, 0, 4));
  CK(hipEventRecord(e0));
  lookup_kernel<<<(n + 255) / 256, 256>>>(ids, n, multi, out);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float tl; CK(hipEventElapsedTime(&tl, e0, e1));
  uint32_t cnt; CK(hipMemcpy(&cnt, out, 4, hipMemcpyDeviceToHost));
  // host check of the multi count
  std::vector<uint8_t> c(rows, 0); uint32_t want = 0;
  for (uint32_t i = 0; i < n; ++i) if (c[h[i]] < 2) ++c[h[i]];
  for (uint32_t i = 0; i < n; ++i) if (c[h[i]] >= 2) ++want;
  float tm;
  CK(hipEventRecord(e0)); CK(hipMemsetAsync(seen, 0, words * 4)); CK(hipMemsetAsync(multi, 0, words * 4)); CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&tm, e0, e1));
  printf("{\"rows\": %u, \"n\": %u, \"mark_ms_per1\": %.4f, \"mark_ms_per4\": %.4f, \"mark_ms_per8\": %.4f, \"lookup_ms\": %.4f, \"memset2_ms\": %.4f, \"multi_occ\": %u, \"multi_occ_host\": %u}\n",
         rows, n, t1, t4, t8, tl, tm, cnt, want);
  return cnt == want ? 0 : 2;
}
