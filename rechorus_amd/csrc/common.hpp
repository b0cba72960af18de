// common.hpp -- shared host/device helpers for librechorus_hip.so (gfx950 only).
#pragma once
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <cmath>

#include <hip/hip_runtime.h>

#include "../../include/rechorus_hip.h"

namespace rc {

// ---- error reporting ---------------------------------------------------------------
char* last_error_buf();  // thread-local, 512 bytes (defined in library.hip)

inline int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(last_error_buf(), 512, fmt, ap);
  va_end(ap);
  return code;
}

#define RC_REQUIRE(cond, ...)                                   \
  do {                                                          \
    if (!(cond)) return ::rc::fail(RC_ERR_INVALID_ARG, __VA_ARGS__); \
  } while (0)

#define RC_HIP(expr)                                                              \
  do {                                                                            \
    hipError_t e_ = (expr);                                                       \
    if (e_ != hipSuccess)                                                         \
      return ::rc::fail(RC_ERR_HIP, "%s failed: %s (%s:%d)", #expr,              \
                        hipGetErrorString(e_), __FILE__, __LINE__);               \
  } while (0)

#define RC_LAUNCH_CHECK() RC_HIP(hipGetLastError())

#define RC_TRY(expr)          \
  do {                        \
    int rc_ = (expr);         \
    if (rc_ != RC_OK) return rc_; \
  } while (0)

inline hipStream_t as_stream(rc_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// carve a sub-buffer out of a caller workspace (256-byte aligned slices)
struct Carver {
  char* base;
  size_t off = 0;
  explicit Carver(void* p) : base(reinterpret_cast<char*>(p)) {}
  template <typename T>
  T* take(size_t count) {
    T* p = reinterpret_cast<T*>(base + off);
    off += align_up(count * sizeof(T), 256);
    return p;
  }
};

constexpr int kWave = 64;          // gfx950 wavefront
constexpr int kBlock = 256;        // default workgroup: 4 waves, one per SIMD
constexpr int kMaxGridX = 1 << 30;

// ---- device helpers ----------------------------------------------------------------
#if defined(__HIPCC__)

// DPP row (16-lane) permutations; ctrl codes per the AMDGPU ISA (DppCtrl):
//   quad_perm [1,0,3,2] = 0xB1, quad_perm [2,3,0,1] = 0x4E,
//   row_half_mirror = 0x141, row_mirror = 0x140.
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float x) {
#ifdef RC_NO_DPP
  // debugging fall-back: same permutations through ds_bpermute
  const int lane = __lane_id();
  int src = lane;
  if (CTRL == 0xB1) src = lane ^ 1;
  else if (CTRL == 0x4E) src = lane ^ 2;
  else if (CTRL == 0x141) src = (lane & ~7) | (7 - (lane & 7));
  else if (CTRL == 0x140) src = (lane & ~15) | (15 - (lane & 15));
  return __shfl(x, src, 64);
#else
  int xi = __float_as_int(x);
  int r = __builtin_amdgcn_update_dpp(xi, xi, CTRL, 0xF, 0xF, false);
  return __int_as_float(r);
#endif
}

// all-reduce (sum) over the LPR consecutive lanes that share one table row.
// This thread's share of a fixed-order sum over x[0, n) by a workgroup of NT threads: elements t, t + NT, ... as float4 chunks
// ((v.x + v.y) + (v.z + v.w) per chunk, chunks in ascending order; a tail of single floats behind them) -- the order every loss
// mean of the library uses (reduce_sum_kernel, plan_final_kernel, small_update_kernel), so the step's loss is bit-identical across
// pipelines.  STAGED: up to 32 chunks are REQUESTED together before the first is added (the adds stay in order): a batch of 65,536
// losses is two memory latencies per thread instead of eight -- 27 -> 8 us for a kernel that is nothing but this sum
// (reduce_sum_kernel).  Not for kernels that do other work beside it: the staging's registers cost the small-batch step's update
// kernel 1.8 us of 13 (same-box A/B), and plan_final_kernel's duration is set by its row workgroups, not by this one.
template <int NT, bool STAGED>
__device__ __forceinline__ float fixed_order_partial(const float* __restrict__ x, int64_t n, int tid) {
  float acc = 0.f;
  int64_t done = 0;
  if (reinterpret_cast<uintptr_t>(x) % 16 == 0) {
    const int64_t n4 = n / 4;
    const float4* x4 = reinterpret_cast<const float4*>(x);
    if (!STAGED || n4 <= (int64_t)4 * NT) {      // (workgroup-uniform)
#pragma unroll 8
      for (int64_t i = tid; i < n4; i += NT) {
        const float4 v = x4[i];
        acc += (v.x + v.y) + (v.z + v.w);
      }
    } else {
      constexpr int Q = 32;
      for (int64_t i0 = tid; i0 < n4; i0 += (int64_t)Q * NT) {
        float4 v[Q];
#pragma unroll
        for (int q = 0; q < Q; ++q) {
          const int64_t i = i0 + (int64_t)q * NT;
          v[q] = x4[i < n4 ? i : i0];      // (no branch between the requests; an unused slot re-reads the first chunk)
        }
#pragma unroll
        for (int q = 0; q < Q; ++q)
          if (i0 + (int64_t)q * NT < n4) acc += (v[q].x + v[q].y) + (v[q].z + v[q].w);
      }
    }
    done = n4 * 4;
  }
  for (int64_t i = done + tid; i < n; i += NT) acc += x[i];
  return acc;
}

// LPR is a power of two in [1, 64]; lanes are LPR-aligned.
template <int LPR>
__device__ __forceinline__ float row_allreduce_sum(float x) {
  if (LPR >= 2) x += dpp_mov<0xB1>(x);
  if (LPR >= 4) x += dpp_mov<0x4E>(x);
  if (LPR >= 8) x += dpp_mov<0x141>(x);
  if (LPR >= 16) x += dpp_mov<0x140>(x);
  if (LPR >= 32) x += __shfl_xor(x, 16, 64);
  if (LPR >= 64) x += __shfl_xor(x, 32, 64);
  return x;
}

// all-reduce over the GS lane-groups of one tuple: lane offsets LPR, 2*LPR, ... S/2.
template <int LPR, int S>
__device__ __forceinline__ float groups_allreduce_sum(float x) {
#pragma unroll
  for (int off = LPR; off < S; off <<= 1) x += __shfl_xor(x, off, 64);
  return x;
}
template <int LPR, int S>
__device__ __forceinline__ float groups_allreduce_max(float x) {
#pragma unroll
  for (int off = LPR; off < S; off <<= 1) x = fmaxf(x, __shfl_xor(x, off, 64));
  return x;
}

__device__ __forceinline__ float wave_allreduce_sum(float x) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) x += __shfl_xor(x, off, 64);
  return x;
}
__device__ __forceinline__ float wave_allreduce_max(float x) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) x = fmaxf(x, __shfl_xor(x, off, 64));
  return x;
}

// streaming row load: read-once table rows; non-temporal (measured: fused BPRMF kernel 0.739 -> 0.629 ms; -DRC_NO_NT restores plain accesses)
__device__ __forceinline__ float4 load_stream4(const float4* p) {
#if !defined(RC_NO_NT) && !defined(RC_NO_NT_LOAD)
  typedef float v4f __attribute__((ext_vector_type(4)));
  const v4f v = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(p));
  return make_float4(v.x, v.y, v.z, v.w);
#else
  return *p;
#endif
}

__device__ __forceinline__ float dot4(const float4& a, const float4& b) {
  return fmaf(a.w, b.w, fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)));
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

#endif  // __HIPCC__

}  // namespace rc
