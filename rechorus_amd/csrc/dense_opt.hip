// dense_opt.hip -- exact dense optimizer step over a whole tensor: torch.optim semantics
// including weight decay on EVERY element (reference: helpers/BaseRunner.py:110-114 builds
// torch.optim.<name>(..., weight_decay=l2); :206 steps it).  Used for small tables, where
// reproducing the reference's dense Adam/SGD/Adagrad bit-for-bit-in-semantics is affordable,
// and for the dense (MLP / attention) parameters.  Streaming, HBM-bound: W, G (+m, v) read
// once, W (+m, v) written once, float4 per lane.
#include "common.hpp"
#include "opt_math.hpp"

namespace rc {

template <int MODE>
__global__ __launch_bounds__(kBlock) void dense_update_kernel(OptScalars o, float* __restrict__ W,
                                                              const float* __restrict__ G,
                                                              float* __restrict__ M,
                                                              float* __restrict__ V, int64_t n) {
  const int64_t n4 = n / 4;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n4;
       i += (int64_t)gridDim.x * kBlock) {
    const float4 g = reinterpret_cast<const float4*>(G)[i];
    const float4 w = reinterpret_cast<const float4*>(W)[i];
    opt_row4<MODE>(o, W, M, V, (size_t)i, w, g);
  }
  const int64_t i = n4 * 4 + (int64_t)blockIdx.x * kBlock + threadIdx.x;  // n % 4 tail
  if (i < n) {
    float w = W[i], m = 0.f, v = 0.f;
    if (MODE == MODE_ADAM || MODE == MODE_ADAGRAD) m = M[i];
    if (MODE == MODE_ADAM) v = V[i];
    opt_elem<MODE>(o, G[i], w, m, v);
    W[i] = w;
    if (MODE == MODE_ADAM || MODE == MODE_ADAGRAD) M[i] = m;
    if (MODE == MODE_ADAM) V[i] = v;
  }
}

template <int MODE>
__global__ __launch_bounds__(kBlock) void dense_update_scalar_kernel(
    OptScalars o, float* __restrict__ W, const float* __restrict__ G, float* __restrict__ M,
    float* __restrict__ V, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * kBlock) {
    float w = W[i], m = 0.f, v = 0.f;
    if (MODE == MODE_ADAM || MODE == MODE_ADAGRAD) m = M[i];
    if (MODE == MODE_ADAM) v = V[i];
    opt_elem<MODE>(o, G[i], w, m, v);
    W[i] = w;
    if (MODE == MODE_ADAM || MODE == MODE_ADAGRAD) M[i] = m;
    if (MODE == MODE_ADAM) V[i] = v;
  }
}

template <int MODE>
static int launch_dense(const OptScalars& o, float* W, const float* G, float* M, float* V,
                        int64_t n, bool aligned, hipStream_t s) {
  const int64_t work = aligned ? (n / 4 > 0 ? n / 4 : 1) : n;
  int64_t blocks = (work + kBlock - 1) / kBlock;
  if (blocks > 256 * 16) blocks = 256 * 16;  // grid-stride beyond 16 blocks per CU
  if (aligned)
    hipLaunchKernelGGL((dense_update_kernel<MODE>), dim3((unsigned)blocks), dim3(kBlock), 0, s, o,
                       W, G, M, V, n);
  else
    hipLaunchKernelGGL((dense_update_scalar_kernel<MODE>), dim3((unsigned)blocks), dim3(kBlock), 0,
                       s, o, W, G, M, V, n);
  RC_LAUNCH_CHECK();
  return RC_OK;
}

}  // namespace rc

using namespace rc;

extern "C" int rc_dense_update(float* W, const float* G, float* m, float* v, int64_t n,
                               const rc_opt_hyper* h, rc_stream_t stream) {
  if (n == 0) return RC_OK;
  RC_REQUIRE(W && G, "rc_dense_update: null pointer");
  RC_REQUIRE(n > 0, "rc_dense_update: n < 0");
  OptScalars o;
  RC_TRY(fill_opt_scalars(h, &o));
  hipStream_t s = as_stream(stream);
  auto al = [](const void* p) { return reinterpret_cast<uintptr_t>(p) % 16 == 0; };
  bool aligned = al(W) && al(G);
  switch (h->opt) {
    case RC_OPT_SGD:
      return launch_dense<MODE_SGD>(o, W, G, m, v, n, aligned, s);
    case RC_OPT_ADAM:
      RC_REQUIRE(m && v, "rc_dense_update: Adam needs m and v");
      return launch_dense<MODE_ADAM>(o, W, G, m, v, n, aligned && al(m) && al(v), s);
    case RC_OPT_ADAGRAD:
      RC_REQUIRE(m, "rc_dense_update: Adagrad needs m (state_sum)");
      return launch_dense<MODE_ADAGRAD>(o, W, G, m, v, n, aligned && al(m), s);
    default:
      return fail(RC_ERR_INVALID_ARG, "rc_dense_update: unknown optimizer %d", h->opt);
  }
}
