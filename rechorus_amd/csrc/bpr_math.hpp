// bpr_math.hpp -- per-row pieces of GeneralModel.loss (models/BaseModel.py:182-185)
// shared by the stand-alone loss kernel and the fused BPRMF kernel.
#pragma once
#include <hip/hip_runtime.h>

namespace rc {

struct BprRow {
  float loss;  // -log(clamp(P))
  float dLdP;  // d(inv_b * loss)/dP, 0 where the clamp saturates
};

// Python evaluates `1-1e-8` in double and torch converts the bound to fp32, which rounds
// it to exactly 1.0f; torch's clamp backward passes the gradient for lo <= P <= hi.
__device__ __forceinline__ BprRow bpr_row(float P, float inv_b) {
  const float lo = 1e-8f;
  const float hi = (float)(1.0 - 1e-8);
  const float Pc = fminf(fmaxf(P, lo), hi);
  BprRow r;
  r.loss = -logf(Pc);
  r.dLdP = (P >= lo && P <= hi) ? (-inv_b / Pc) : 0.0f;
  return r;
}

// dP/dneg_j for softmax weight w_j, s_j = sigmoid(pos - neg_j)
__device__ __forceinline__ float bpr_dP_dneg(float w, float s, float P) {
  return w * ((s - P) - s * (1.0f - s));
}

}  // namespace rc
