// opt_math.hpp -- element-wise optimizer arithmetic shared by every kernel that updates a
// table row (segmented update, dense update, the fused BPRMF singleton path).
// Formulas: torch/optim/{sgd,adam,adagrad}.py single-tensor paths, which the reference
// reaches through helpers/BaseRunner.py:110-114 (construction) and :206 (step).
#pragma once
#include "common.hpp"

namespace rc {

enum { MODE_SGD = 0, MODE_ADAM = 1, MODE_ADAGRAD = 2, MODE_DENSE_GRAD = 3, MODE_NONE = 4, MODE_ADADELTA = 5 };

// which optimizer state tensors a mode reads and writes (m: exp_avg / state_sum / square_avg, v: exp_avg_sq / acc_delta)
constexpr bool mode_has_m(int mode) { return mode == MODE_ADAM || mode == MODE_ADAGRAD || mode == MODE_ADADELTA; }
constexpr bool mode_has_v(int mode) { return mode == MODE_ADAM || mode == MODE_ADADELTA; }

// python-double hyper-parameters narrowed to fp32 at the points where torch narrows them
struct OptScalars {
  float neg_lr;    // SGD/Adagrad: -lr
  float l2;        // weight_decay
  float one_m_b1;  // Adam: 1 - beta1
  float b2;        // Adam: beta2
  float one_m_b2;  // Adam: 1 - beta2
  float neg_step;  // Adam: -(lr / (1 - beta1^t))
  float bc2_sqrt;  // Adam: sqrt(1 - beta2^t)
  float eps;
};

inline int fill_opt_scalars(const rc_opt_hyper* h, OptScalars* a, bool dense = false) {
  RC_REQUIRE(h != nullptr, "optimizer hyper-parameters missing");
  RC_REQUIRE(h->opt == RC_OPT_SGD || h->opt == RC_OPT_ADAM || h->opt == RC_OPT_ADAGRAD || (dense && h->opt == RC_OPT_ADADELTA),
             h->opt == RC_OPT_ADADELTA ? "Adadelta (optimizer %d) is built for dense steps only (rc_dense_update*)"
                                       : "unknown optimizer %d", h->opt);
  a->l2 = (float)h->l2;
  a->neg_lr = (float)(-h->lr);
  a->eps = (float)h->eps;
  a->one_m_b1 = a->b2 = a->one_m_b2 = a->neg_step = 0.f;
  a->bc2_sqrt = 1.f;
  if (h->opt == RC_OPT_ADADELTA) {  // torch/optim/adadelta.py: rho in beta1
    a->b2 = (float)h->beta1;
    a->one_m_b2 = (float)(1.0 - h->beta1);
  }
  if (h->opt == RC_OPT_ADAM) {
    RC_REQUIRE(h->step >= 1, "Adam needs step >= 1 (got %lld)", (long long)h->step);
    const double bc1 = 1.0 - pow(h->beta1, (double)h->step);
    const double bc2 = 1.0 - pow(h->beta2, (double)h->step);
    a->one_m_b1 = (float)(1.0 - h->beta1);
    a->b2 = (float)h->beta2;
    a->one_m_b2 = (float)(1.0 - h->beta2);
    a->neg_step = (float)(-(h->lr / bc1));
    a->bc2_sqrt = (float)sqrt(bc2);
  }
  return RC_OK;
}

inline int mode_of(const rc_opt_hyper* h) {
  return h->opt == RC_OPT_SGD ? MODE_SGD : (h->opt == RC_OPT_ADAM ? MODE_ADAM : (h->opt == RC_OPT_ADADELTA ? MODE_ADADELTA : MODE_ADAGRAD));
}

#if defined(__HIPCC__)
template <int MODE>
__device__ __forceinline__ void opt_elem(const OptScalars& a, float g, float& w, float& m,
                                         float& v) {
  if (MODE == MODE_SGD) {
    g = fmaf(a.l2, w, g);  // grad.add(param, alpha=weight_decay)
    w = fmaf(a.neg_lr, g, w);
  } else if (MODE == MODE_ADAM) {
    g = fmaf(a.l2, w, g);
    m = fmaf(a.one_m_b1, g - m, m);         // exp_avg.lerp_(grad, 1-beta1)
    v = fmaf(a.one_m_b2, g * g, v * a.b2);  // mul_(beta2).addcmul_(g, g, 1-beta2)
    const float denom = sqrtf(v) / a.bc2_sqrt + a.eps;
    w = fmaf(a.neg_step, m / denom, w);     // addcdiv_(exp_avg, denom, -step_size)
  } else if (MODE == MODE_ADAGRAD) {
    g = fmaf(a.l2, w, g);
    m = fmaf(g, g, m);                      // state_sum.addcmul_(g, g, 1)
    w = fmaf(a.neg_lr, g / (sqrtf(m) + a.eps), w);
  } else if (MODE == MODE_ADADELTA) {       // torch/optim/adadelta.py, single-tensor path
    g = fmaf(a.l2, w, g);
    m = fmaf(a.one_m_b2, g * g, m * a.b2);  // square_avg.mul_(rho).addcmul_(g, g, 1 - rho)
    const float std = sqrtf(m + a.eps);     // square_avg.add(eps).sqrt_()
    const float delta = sqrtf(v + a.eps) / std * g;   // acc_delta.add(eps).sqrt_().div_(std).mul_(g)
    v = fmaf(a.one_m_b2, delta * delta, v * a.b2);    // acc_delta.mul_(rho).addcmul_(delta, delta, 1 - rho)
    w = fmaf(a.neg_lr, delta, w);           // param.add_(delta, alpha=-lr)
  } else if (MODE == MODE_DENSE_GRAD) {     // no optimizer: the "table" is a gradient buffer, its row := the summed gradient
    w = g;
  }
}

// row write-back: the row is not read again in this step -> non-temporal store (-DRC_NO_NT: plain)
__device__ __forceinline__ void store_row4(float4* p, const float4& x) {
#if defined(RC_STORE_SC1)
  // experiment switch: write-through store that does not keep the line in this XCD's L2 (MI355X_MICROARCH.md, store flavours)
  typedef float v4f __attribute__((ext_vector_type(4)));
  const v4f v = {x.x, x.y, x.z, x.w};
  asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
#elif !defined(RC_NO_NT) && !defined(RC_NO_NT_STORE)
  typedef float v4f __attribute__((ext_vector_type(4)));
  v4f v = {x.x, x.y, x.z, x.w};
  __builtin_nontemporal_store(v, reinterpret_cast<v4f*>(p));
#else
  *p = x;
#endif
}

// the optimizer on one float4 slice held in registers (state already loaded)
template <int MODE>
__device__ __forceinline__ void opt_apply4(const OptScalars& a, float4& w, float4& m, float4& v, const float4& g) {
  opt_elem<MODE>(a, g.x, w.x, m.x, v.x);
  opt_elem<MODE>(a, g.y, w.y, m.y, v.y);
  opt_elem<MODE>(a, g.z, w.z, m.z, v.z);
  opt_elem<MODE>(a, g.w, w.w, m.w, v.w);
}

// update one float4 slice of a table row in place (row index `row`, slice l of LPR)
template <int MODE>
__device__ __forceinline__ void opt_row4(const OptScalars& a, float* __restrict__ W,
                                         float* __restrict__ M, float* __restrict__ V, size_t idx4,
                                         float4 w, const float4& g) {
  float4 m = make_float4(0, 0, 0, 0), v = make_float4(0, 0, 0, 0);
  if (mode_has_m(MODE)) m = load_stream4(reinterpret_cast<const float4*>(M) + idx4);
  if (mode_has_v(MODE)) v = load_stream4(reinterpret_cast<const float4*>(V) + idx4);
  opt_elem<MODE>(a, g.x, w.x, m.x, v.x);
  opt_elem<MODE>(a, g.y, w.y, m.y, v.y);
  opt_elem<MODE>(a, g.z, w.z, m.z, v.z);
  opt_elem<MODE>(a, g.w, w.w, m.w, v.w);
  store_row4(reinterpret_cast<float4*>(W) + idx4, w);
  if (mode_has_m(MODE)) store_row4(reinterpret_cast<float4*>(M) + idx4, m);
  if (mode_has_v(MODE)) store_row4(reinterpret_cast<float4*>(V) + idx4, v);
}
#endif

}  // namespace rc
