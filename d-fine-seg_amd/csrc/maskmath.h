// Element math of the mask loss (reference src/d_fine/dfine_criterion.py:272-357: _focal_loss_mask, _dice_loss), shared
// by the row kernels over materialised logits (maskloss.hip) and the GEMM epilogues that never write them (igemm2.hip).
//     p = sigmoid(z), p_t = p t + (1-p)(1-t), bce = softplus(z) - z t, a_t = alpha t + (1-alpha)(1-t)
//     stats = ( sum a_t (1-p_t)^2 bce ,  sum p t ,  sum p )
//     dz = g0 * a_t [ 2 (1-p_t)(1-2t) p(1-p) bce + (1-p_t)^2 (p - t) ]  +  (g1 t + g2) p (1-p)
#pragma once
#include "common.h"

namespace dfine {

// sigmoid and softplus from ONE exponential: e = exp(-|z|), p = (z >= 0 ? 1 : e) / (1 + e), softplus = max(z, 0) + log(1 + e)
__device__ __forceinline__ void sigmoid_softplus(float z, float& p, float& softplus) {
  const float e = __expf(-fabsf(z));
  const float r = __frcp_rn(1.f + e);
  p = z >= 0.f ? r : e * r;
  softplus = fmaxf(z, 0.f) + __logf(1.f + e);
}

__device__ __forceinline__ void mask_terms(float z, float t, float alpha, float& focal, float& p_out) {
  float p, sp;
  sigmoid_softplus(z, p, sp);
  const float bce = sp - z * t;                 // softplus(z) - z t
  const float one_m_pt = p + t - 2.f * p * t;   // 1 - p_t
  const float a_t = alpha * t + (1.f - alpha) * (1.f - t);
  focal = a_t * one_m_pt * one_m_pt * bce;
  p_out = p;
}

__device__ __forceinline__ float mask_dz(float z, float t, float alpha, float g0, float g1, float g2) {
  float p, sp;
  sigmoid_softplus(z, p, sp);
  const float pq = p * (1.f - p);
  const float bce = sp - z * t;
  const float one_m_pt = p + t - 2.f * p * t;
  const float a_t = alpha * t + (1.f - alpha) * (1.f - t);
  // d(1 - p_t)/dz = (1 - 2t) p (1-p);  d bce/dz = p - t
  const float dfocal = a_t * (2.f * one_m_pt * (1.f - 2.f * t) * pq * bce + one_m_pt * one_m_pt * (p - t));
  return g0 * dfocal + (g1 * t + g2) * pq;
}

}  // namespace dfine
