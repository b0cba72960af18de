// sampler.hip -- training-batch assembly on the device: negative sampling and history windows.
//
// Reference: GeneralModel.Dataset.actions_before_epoch (models/BaseModel.py:206-214): for every
// training row draw num_neg ids uniformly from [1, n_items) and redraw while the id is in the user's
// TRAIN clicked set; SequentialModel.Dataset._get_feed_dict (models/BaseModel.py:236-245): the last
// history_max items of the user's time-ordered history before `position`, right-padded with 0 by
// collate_batch (:135-152).  The reference does this in Python loops on the CPU (30-50 % of a K = 99
// epoch); here it is integer work at HBM speed, one thread per drawn id / history slot.
//
// Randomness: counter-based Philox4x32-10 (Salmon et al., SC'11), key = seed, counter =
// (element index, attempt block): any element can be regenerated independently, so the result does
// not depend on the launch geometry and the numpy restatement (oracle/sampler_oracle.py) is bit-exact.
// Each Philox block yields two 64-bit words = two attempts; a word w maps to 1 + mulhi64(w, n_items-1)
// (bias <= n_items / 2^64).  The reference's numpy MT19937 stream is not reproduced: parity with the
// reference is distributional (uniform over the user's non-clicked items), parity with the oracle exact.
#include "common.hpp"
#include "philox.hpp"

namespace rc {

constexpr int kMaxAttempts = 1024;  // (clicked fraction)^1024: never reached unless a user clicked ~everything

__device__ __forceinline__ bool contains_sorted(const int64_t* __restrict__ a, int64_t lo, int64_t hi, int64_t x) {
  while (lo < hi) {  // [lo, hi)
    const int64_t mid = lo + ((hi - lo) >> 1);
    const int64_t v = a[mid];
    if (v == x) return true;
    if (v < x) lo = mid + 1; else hi = mid;
  }
  return false;
}

__global__ __launch_bounds__(kBlock) void sample_negatives_kernel(
    const int64_t* __restrict__ users, int64_t n, int K, int64_t n_items,
    const int64_t* __restrict__ clicked_ptr, const int64_t* __restrict__ clicked_items, uint64_t seed,
    uint64_t base_index, int64_t* __restrict__ neg) {
  const int64_t total = n * K;
  const uint64_t range = (uint64_t)(n_items - 1);
  for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < total; e += (int64_t)gridDim.x * kBlock) {
    const int64_t u = users[e / K];
    const int64_t lo = clicked_ptr ? clicked_ptr[u] : 0, hi = clicked_ptr ? clicked_ptr[u + 1] : 0;
    int64_t cand = 1;
    bool found = false;
    for (int attempt = 0; attempt < kMaxAttempts && !found; attempt += 2) {
      uint32_t r[4];
      philox4x32_10(seed, base_index + (uint64_t)e, (uint32_t)(attempt >> 1), r);
      const uint64_t w0 = ((uint64_t)r[1] << 32) | r[0], w1 = ((uint64_t)r[3] << 32) | r[2];
      cand = 1 + (int64_t)__umul64hi(w0, range);
      found = !contains_sorted(clicked_items, lo, hi, cand);
      if (found) break;
      cand = 1 + (int64_t)__umul64hi(w1, range);
      found = !contains_sorted(clicked_items, lo, hi, cand);
    }
    if (!found) {
      // The user clicked (nearly) the whole catalogue: the reference's `while neg in clicked` loop (models/BaseModel.py:
      // 209-210) would keep drawing; here the r-th NON-clicked id is selected directly, r uniform -- the same
      // distribution, no loop.  Non-clicked ids before the j-th clicked id c_j (sorted, distinct, >= 1): c_j - 1 - j.
      const int64_t nc = hi - lo, free_ids = (n_items - 1) - nc;
      if (free_ids > 0) {
        uint32_t r[4];
        philox4x32_10(seed, base_index + (uint64_t)e, (uint32_t)(kMaxAttempts >> 1), r);
        const int64_t rank = (int64_t)__umul64hi(((uint64_t)r[1] << 32) | r[0], (uint64_t)free_ids);
        int64_t a = 0, b = nc;  // smallest j with clicked[j] - 1 - j > rank (nc if none)
        while (a < b) {
          const int64_t mid = a + ((b - a) >> 1);
          if (clicked_items[lo + mid] - 1 - mid > rank) b = mid; else a = mid + 1;
        }
        cand = rank + 1 + a;
      }  // (a user who clicked EVERY item keeps the last draw: there is nothing else to return)
    }
    neg[e] = cand;
  }
}

// out[b, 0] = items[idx[b]], out[b, 1 + k] = neg[idx[b], k]; users_out[b] = users[idx[b]]
__global__ __launch_bounds__(kBlock) void assemble_candidates_kernel(
    const int64_t* __restrict__ idx, int64_t B, int K, const int64_t* __restrict__ users,
    const int64_t* __restrict__ items, const int64_t* __restrict__ neg, int64_t* __restrict__ users_out,
    int64_t* __restrict__ cand_out) {
  const int C = 1 + K;
  const int64_t total = B * C;
  for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < total; e += (int64_t)gridDim.x * kBlock) {
    const int64_t b = e / C;
    const int c = (int)(e - b * C);
    const int64_t i = idx[b];
    cand_out[e] = c == 0 ? items[i] : neg[i * K + (c - 1)];
    if (c == 0) users_out[b] = users[i];
  }
}

// history window: hist[b, t] = his_items[his_ptr[u] + pos - len + t] for t < len = min(pos, L), else 0
__global__ __launch_bounds__(kBlock) void gather_history_kernel(
    const int64_t* __restrict__ idx, int64_t B, int L, const int64_t* __restrict__ users,
    const int64_t* __restrict__ position, const int64_t* __restrict__ his_ptr,
    const int64_t* __restrict__ his_items, const int64_t* __restrict__ his_times, int64_t* __restrict__ hist,
    int64_t* __restrict__ times, int64_t* __restrict__ lengths) {
  const int64_t total = B * L;
  for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < total; e += (int64_t)gridDim.x * kBlock) {
    const int64_t b = e / L;
    const int t = (int)(e - b * L);
    const int64_t i = idx ? idx[b] : b;
    const int64_t pos = position[i];
    const int64_t len = pos < L ? pos : L;
    const int64_t src = his_ptr[users[i]] + pos - len + t;
    hist[e] = t < len ? his_items[src] : 0;
    if (times) times[e] = t < len ? his_times[src] : 0;
    if (t == 0) lengths[b] = len;
  }
}

}  // namespace rc

using namespace rc;

static unsigned grid_for(int64_t total) {
  int64_t blocks = (total + kBlock - 1) / kBlock;
  if (blocks > 256 * 16) blocks = 256 * 16;
  return (unsigned)blocks;
}

extern "C" int rc_sample_negatives(const int64_t* users, int64_t n, int K, int64_t n_items,
                                   const int64_t* clicked_ptr, const int64_t* clicked_items, uint64_t seed,
                                   uint64_t base_index, int64_t* neg, rc_stream_t stream) {
  if (n == 0 || K == 0) return RC_OK;
  RC_REQUIRE(users && neg, "rc_sample_negatives: null pointer");
  RC_REQUIRE(n > 0 && K > 0 && n_items >= 2, "rc_sample_negatives: bad shape n=%lld K=%d n_items=%lld", (long long)n, K,
             (long long)n_items);
  RC_REQUIRE((clicked_ptr == nullptr) == (clicked_items == nullptr),
             "rc_sample_negatives: clicked_ptr and clicked_items go together");
  hipLaunchKernelGGL(sample_negatives_kernel, dim3(grid_for(n * K)), dim3(kBlock), 0, as_stream(stream), users, n, K,
                     n_items, clicked_ptr, clicked_items, seed, base_index, neg);
  RC_LAUNCH_CHECK();
  return RC_OK;
}

extern "C" int rc_assemble_candidates(const int64_t* idx, int64_t B, int K, const int64_t* users, const int64_t* items,
                                      const int64_t* neg, int64_t* users_out, int64_t* cand_out, rc_stream_t stream) {
  if (B == 0) return RC_OK;
  RC_REQUIRE(idx && users && items && users_out && cand_out && (neg || K == 0), "rc_assemble_candidates: null pointer");
  RC_REQUIRE(B > 0 && K >= 0, "rc_assemble_candidates: bad shape B=%lld K=%d", (long long)B, K);
  hipLaunchKernelGGL(assemble_candidates_kernel, dim3(grid_for(B * (1 + K))), dim3(kBlock), 0, as_stream(stream), idx, B,
                     K, users, items, neg, users_out, cand_out);
  RC_LAUNCH_CHECK();
  return RC_OK;
}

extern "C" int rc_gather_history(const int64_t* idx, int64_t B, int L, const int64_t* users, const int64_t* position,
                                 const int64_t* his_ptr, const int64_t* his_items, const int64_t* his_times,
                                 int64_t* hist, int64_t* times, int64_t* lengths, rc_stream_t stream) {
  if (B == 0) return RC_OK;
  RC_REQUIRE(users && position && his_ptr && his_items && hist && lengths, "rc_gather_history: null pointer");
  RC_REQUIRE(B > 0 && L >= 1, "rc_gather_history: bad shape B=%lld L=%d", (long long)B, L);
  RC_REQUIRE((times == nullptr) || his_times, "rc_gather_history: times requested without his_times");
  hipLaunchKernelGGL(gather_history_kernel, dim3(grid_for(B * L)), dim3(kBlock), 0, as_stream(stream), idx, B, L, users,
                     position, his_ptr, his_items, his_times, hist, times, lengths);
  RC_LAUNCH_CHECK();
  return RC_OK;
}
