// pair_ticket_probe.hip -- what the "two-occurrence rows resolved inside the fused kernel by an integer ticket" idea (VERDICT
// round 5, item 2a) would pay in memory operations, measured on its own: the BPRMF contract batch has 1.1 M item rows with
// exactly two occurrences; each of their 2.2 M occurrences would do one 64-bit exchange on a per-row slot (the first arriver
// parks (g, user id), the second gets it back), and the second arriver then gathers the parked user row (256 B), adds, applies SGD
// to the row it holds, stores the row (256 B) and clears the slot.
//   part A: 2.2 M device-scope 64-bit exchanges on 1.1 M random slots of a 10 M-slot array (16 lanes of a wave idle, as in the
//           fused kernel: lane l == 0 of a lane-group issues, the group waits for the result)
//   part B: A + the second arriver's dependent 256-byte gather + 256-byte row store + slot clear
//   part C: the same 1.1 M row gathers + stores with NO ticket (what the plan-driven update does for them: rows streamed)
// Prints one JSON object.   hipcc -O3 --offload-arch=gfx950 tools/pair_ticket_probe.hip -o tools/bin/pair_ticket_probe
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float v4f __attribute__((ext_vector_type(4)));

// one lane-group (16 lanes) per occurrence, 4 occurrences per wave in flight per trip, U trips unrolled
template <int MODE>   // 0: exchange only, 1: exchange + dependent gather / store / clear, 2: gather + store, no ticket
__global__ __launch_bounds__(256) void probe(unsigned long long* slot, const uint32_t* row_of, const uint32_t* user_of, uint32_t n_occ,
                                             const v4f* U, v4f* I, float* sink) {
  const int lane = threadIdx.x & 63, l = lane & 15;
  const uint32_t group = (blockIdx.x * 256 + threadIdx.x) >> 4;
  const uint32_t n_groups = gridDim.x * 16;
  v4f acc = {0, 0, 0, 0};
  for (uint32_t o = group; o < n_occ; o += n_groups) {
    const uint32_t row = row_of[o];
    const uint32_t user = user_of[o];
    unsigned long long old = 0;
    if (MODE != 2) {
      if (l == 0) old = atomicExch(&slot[row], ((unsigned long long)0x3F800000u << 32) | (user + 1u));
      old = ((unsigned long long)(uint32_t)__shfl((int)(old >> 32), lane & ~15, 64) << 32) | (uint32_t)__shfl((int)(uint32_t)old, lane & ~15, 64);
    }
    const bool second = MODE == 2 ? (o & 1u) != 0 : old != 0;
    if (MODE >= 1 && second) {
      const uint32_t u1 = MODE == 2 ? user : (uint32_t)old - 1u;
      const v4f a = U[(size_t)u1 * 16 + l];
      v4f w = I[(size_t)row * 16 + l];
      w += a * 0.001f;
      I[(size_t)row * 16 + l] = w;
      if (MODE == 1 && l == 0) slot[row] = 0ull;
      acc += a;
    } else {
      acc.x += (float)(old & 1u);
    }
  }
  if (acc.x + acc.y + acc.z + acc.w == 12345.678f) sink[0] = acc.x;
}

int main() {
  const uint32_t n_items = 10000001, n_users = 1000001, n_pairs = 1100000, n_occ = 2 * n_pairs;
  std::mt19937_64 rng(7);
  std::vector<uint32_t> row_of(n_occ), user_of(n_occ);
  // every pair row twice, at two random places of the occurrence stream (the two tuples that hold it are unrelated)
  std::vector<uint32_t> rows(n_pairs);
  for (auto& r : rows) r = 1 + (uint32_t)(rng() % (n_items - 1));
  for (uint32_t k = 0; k < n_pairs; ++k) row_of[2 * k] = row_of[2 * k + 1] = rows[k];
  for (uint32_t i = n_occ - 1; i > 0; --i) std::swap(row_of[i], row_of[rng() % (i + 1)]);
  for (auto& u : user_of) u = (uint32_t)(rng() % 27400);       // 27.4 K distinct users per batch (Zipf head: L2 / MALL resident)
  unsigned long long* slot; uint32_t *d_row, *d_user; v4f *U, *I; float* sink;
  CK(hipMalloc(&slot, (size_t)n_items * 8)); CK(hipMemset(slot, 0, (size_t)n_items * 8));
  CK(hipMalloc(&d_row, n_occ * 4)); CK(hipMalloc(&d_user, n_occ * 4));
  CK(hipMemcpy(d_row, row_of.data(), n_occ * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_user, user_of.data(), n_occ * 4, hipMemcpyHostToDevice));
  CK(hipMalloc(&U, (size_t)n_users * 256)); CK(hipMemset(U, 0, (size_t)n_users * 256));
  CK(hipMalloc(&I, (size_t)n_items * 256)); CK(hipMemset(I, 0, (size_t)n_items * 256));
  CK(hipMalloc(&sink, 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float ms[3] = {0, 0, 0};
  const int reps = 10;
  for (int mode = 0; mode < 3; ++mode) {
    for (int r = -2; r < reps; ++r) {
      CK(hipMemset(slot, 0, (size_t)n_items * 8));
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0));
      if (mode == 0) hipLaunchKernelGGL(probe<0>, dim3(256 * 12), dim3(256), 0, 0, slot, d_row, d_user, n_occ, U, I, sink);
      if (mode == 1) hipLaunchKernelGGL(probe<1>, dim3(256 * 12), dim3(256), 0, 0, slot, d_row, d_user, n_occ, U, I, sink);
      if (mode == 2) hipLaunchKernelGGL(probe<2>, dim3(256 * 12), dim3(256), 0, 0, slot, d_row, d_user, n_occ, U, I, sink);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float t; CK(hipEventElapsedTime(&t, e0, e1));
      if (r >= 0) ms[mode] += t / reps;
    }
  }
  // every slot is clear again after mode 1 (each pair row: first parks, second clears)
  std::vector<unsigned long long> h(1000);
  CK(hipMemcpy(h.data(), slot + rows[0] - (rows[0] > 500 ? 500 : 0), 1000 * 8, hipMemcpyDeviceToHost));
  printf("{\"pair_rows\": %u, \"exchanges\": %u, \"ms_exchange_only\": %.4f, \"ms_exchange_gather_store_clear\": %.4f, \"ms_gather_store_no_ticket\": %.4f, "
         "\"exchanges_per_us\": %.1f}\n", n_pairs, n_occ, ms[0], ms[1], ms[2], n_occ / (ms[0] * 1e3));
  return 0;
}
