// Optimizer tail of the training step as two streaming kernels over flat fp32 buffers:
//   1. global gradient norm  (torch.nn.utils.clip_grad_norm_, reference src/dl/train.py:496-503)
//   2. clip + AdamW (4 parameter groups, build_optimizer src/d_fine/dfine.py:87-124) + EMA of the
//      weights (ModelEMA.update, src/dl/train.py:61-72) + bf16 shadow copy of the weights for the next
//      forward + zeroing of the gradient buffer (optimizer.zero_grad).
// The stock path is ~650-tensor foreach launches for each of these; here every element is read and
// written once: 20 B read + 18 B written per parameter, HBM-bound.
// All per-step scalars (learning rates, bias corrections, EMA decay) live in a small device array that
// the host refreshes with one async copy, so the launch sequence is identical every step (graph-safe).
#include "common.h"

namespace dfine {

struct OptSegment {   // a parameter group = a contiguous range of the flat buffers
  long long begin, end;
};
struct OptSegments {
  int n;
  OptSegment seg[8];
};

// hyper layout (floats): [0] clip max_norm (<=0: no clipping), [1] beta1, [2] beta2, [3] eps,
// [4] 1-beta1^t, [5] 1-beta2^t, [6] ema decay (<0: no EMA), [8+2g] lr of group g, [9+2g] weight decay.
#define HYP_MAXNORM 0
#define HYP_B1 1
#define HYP_B2 2
#define HYP_EPS 3
#define HYP_BC1 4
#define HYP_BC2 5
#define HYP_EMA 6
#define HYP_GROUP 8

__global__ __launch_bounds__(256) void grad_sumsq_kernel(const float* __restrict__ grad, long long n4,
                                                         long long n, double* __restrict__ out) {
  float acc = 0.f;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    const float4 g = reinterpret_cast<const float4*>(grad)[i];
    acc += g.x * g.x + g.y * g.y + g.z * g.z + g.w * g.w;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0)
    for (long long i = n4 * 4; i < n; ++i) acc += grad[i] * grad[i];
  acc = group_sum<64>(acc);
  __shared__ float partial[4];
  if ((threadIdx.x & 63) == 0) partial[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(out, (double)(partial[0] + partial[1] + partial[2] + partial[3]));
}

__global__ __launch_bounds__(256) void adamw_ema_kernel(float* __restrict__ param, float* __restrict__ grad,
                                                        float* __restrict__ exp_avg, float* __restrict__ exp_avg_sq,
                                                        float* __restrict__ ema, bf16_t* __restrict__ shadow,
                                                        const float* __restrict__ hyper,
                                                        const double* __restrict__ grad_sumsq, OptSegments segs,
                                                        long long n) {
  const float max_norm = hyper[HYP_MAXNORM];
  float coef = 1.f;
  if (max_norm > 0.f) {
    const float total = sqrtf((float)grad_sumsq[0]);
    coef = fminf(max_norm / (total + 1e-6f), 1.f);
  }
  const float b1 = hyper[HYP_B1], b2 = hyper[HYP_B2], eps = hyper[HYP_EPS];
  const float bc1 = hyper[HYP_BC1], rsqrt_bc2 = rsqrtf(hyper[HYP_BC2]);
  const float decay = hyper[HYP_EMA];
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    int gi = 0;
#pragma unroll
    for (int s = 1; s < 8; ++s)
      if (s < segs.n && i >= segs.seg[s].begin) gi = s;
    const float lr = hyper[HYP_GROUP + 2 * gi], wd = hyper[HYP_GROUP + 2 * gi + 1];
    const float g = grad[i] * coef;
    float p = param[i] * (1.f - lr * wd);
    const float m = b1 * exp_avg[i] + (1.f - b1) * g;
    const float v = b2 * exp_avg_sq[i] + (1.f - b2) * g * g;
    p -= (lr / bc1) * m / (sqrtf(v) * rsqrt_bc2 + eps);
    param[i] = p;
    exp_avg[i] = m;
    exp_avg_sq[i] = v;
    grad[i] = 0.f;
    if (decay >= 0.f) ema[i] = decay * ema[i] + (1.f - decay) * p;
    if (shadow) shadow[i] = f32_to_bf16(p);
  }
}

}  // namespace dfine

using namespace dfine;

extern "C" int dfine_grad_sumsq(const float* grad, long long n, double* out, void* stream) {
  if (!grad || !out || n <= 0) return DFINE_E_BADARG;
  if (((uintptr_t)grad & 15) != 0) return DFINE_E_BADARG;
  hipLaunchKernelGGL(grad_sumsq_kernel, dim3(1024), dim3(256), 0, (hipStream_t)stream, grad, n / 4, n, out);
  DFINE_LAUNCH_CHECK();
  return 0;
}

extern "C" int dfine_adamw_ema_step(float* param, float* grad, float* exp_avg, float* exp_avg_sq, float* ema,
                                    void* bf16_shadow, const float* hyper, const double* grad_sumsq,
                                    const long long* group_begin, int n_groups, long long n, void* stream) {
  if (!param || !grad || !exp_avg || !exp_avg_sq || !hyper || !grad_sumsq || !group_begin) return DFINE_E_BADARG;
  if (n <= 0 || n_groups <= 0 || n_groups > 8) return DFINE_E_BADARG;
  OptSegments segs;
  segs.n = n_groups;
  for (int g = 0; g < n_groups; ++g) {
    segs.seg[g].begin = group_begin[g];
    segs.seg[g].end = g + 1 < n_groups ? group_begin[g + 1] : n;
    if (segs.seg[g].begin < 0 || segs.seg[g].begin > segs.seg[g].end) return DFINE_E_BADARG;
  }
  if (group_begin[0] != 0) return DFINE_E_BADARG;
  hipLaunchKernelGGL(adamw_ema_kernel, dim3(2048), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg,
                     exp_avg_sq, ema, (bf16_t*)bf16_shadow, hyper, grad_sumsq, segs, n);
  DFINE_LAUNCH_CHECK();
  return 0;
}
