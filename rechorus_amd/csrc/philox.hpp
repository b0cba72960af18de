// philox.hpp -- counter-based Philox4x32-10 (Salmon et al., SC'11), shared by the negative sampler and the
// dropout mask of the NeuMF head; pinned by the Random123 known-answer vectors (tests/test_sampler_cpu.py).
#pragma once
#include "common.hpp"

namespace rc {

struct Philox {
  uint32_t c[4];
  uint32_t k[2];
};

__device__ __forceinline__ void philox_round(Philox& p) {
  constexpr uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
  const uint32_t hi0 = __umulhi(M0, p.c[0]), lo0 = M0 * p.c[0];
  const uint32_t hi1 = __umulhi(M1, p.c[2]), lo1 = M1 * p.c[2];
  const uint32_t n0 = hi1 ^ p.c[1] ^ p.k[0], n2 = hi0 ^ p.c[3] ^ p.k[1];
  p.c[0] = n0; p.c[1] = lo1; p.c[2] = n2; p.c[3] = lo0;
}

// 4 x 32 random bits for (seed, index, block)
__device__ __forceinline__ void philox4x32_10(uint64_t seed, uint64_t index, uint32_t block, uint32_t out[4]) {
  constexpr uint32_t W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
  Philox p;
  p.c[0] = (uint32_t)index; p.c[1] = (uint32_t)(index >> 32); p.c[2] = block; p.c[3] = 0u;
  p.k[0] = (uint32_t)seed; p.k[1] = (uint32_t)(seed >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    philox_round(p);
    p.k[0] += W0; p.k[1] += W1;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) out[i] = p.c[i];
}

}  // namespace rc
