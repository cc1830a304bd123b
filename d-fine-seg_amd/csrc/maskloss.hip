// Mask loss of the matched queries (reference src/d_fine/dfine_criterion.py:272-357: _focal_loss_mask,
// _dice_loss, loss_masks), per (image, target slot) row of mask logits z[HW] against its ground-truth
// mask t[HW] in [0,1]:
//     p = sigmoid(z), p_t = p t + (1-p)(1-t), bce = softplus(z) - z t, a_t = alpha t + (1-alpha)(1-t)
//     stats = ( sum a_t (1-p_t)^2 bce ,  sum p t ,  sum p )
// The tiny rest (means, dice ratio, average over the matched pairs) stays in PyTorch on [B, M] tensors, so
// autograd hands the backward kernel d(loss)/d(stats) and it writes d(loss)/dz in one pass:
//     dz = g0 * a_t [ -2 (1-p_t)(2t-1) p(1-p) bce + (1-p_t)^2 (p - t) ]  +  (g1 t + g2) p (1-p)
// The stock formulation is ~15 elementwise launches forward and as many backward over fp32 [B, M, HW]
// tensors (315 MB each for the denoising heads of D-FINE-m at batch 32).
//
// Forward: one workgroup per row, 16-byte loads, wave shuffle + LDS reduction, no atomics.
// Target slot m of image b is ground truth m % G (M = groups * G): the denoising groups share targets.
#include "common.h"
#include "maskmath.h"

namespace dfine {

struct MaskGeom {
  int M, G, HW;
};

template <typename T>
__global__ __launch_bounds__(256) void mask_stats_kernel(const T* __restrict__ logits, const float* __restrict__ tgt,
                                                         const float* __restrict__ alpha, float* __restrict__ stats,
                                                         MaskGeom g) {
  constexpr int V = VecTraits<T>::kElems;
  const int row = blockIdx.x;  // b * M + m
  const int b = row / g.M, gt = (row % g.M) % g.G;
  const T* z = logits + (size_t)row * g.HW;
  const float* t = tgt + ((size_t)b * g.G + gt) * g.HW;
  const float a = alpha[b * g.G + gt];
  float s_f = 0.f, s_i = 0.f, s_p = 0.f;
  // a row is shared by gridDim.y workgroups (few rows -- the 8 ground truths of the own heads -- would
  // otherwise leave most CUs idle); partial sums are added to the zero-initialised stats
  for (int i = (blockIdx.y * blockDim.x + threadIdx.x) * V; i < g.HW; i += gridDim.y * blockDim.x * V) {  // HW % V == 0
    float zz[V], tt[V];
    load_vec<T, V>(z + i, zz);
#pragma unroll
    for (int k = 0; k < V; k += 4) load_vec<float, 4>(t + i + k, tt + k);
#pragma unroll
    for (int k = 0; k < V; ++k) {
      float f, p;
      mask_terms(zz[k], tt[k], a, f, p);
      s_f += f;
      s_i += p * tt[k];
      s_p += p;
    }
  }
  s_f = group_sum<64>(s_f), s_i = group_sum<64>(s_i), s_p = group_sum<64>(s_p);
  __shared__ float red[4][3];
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) red[wave][0] = s_f, red[wave][1] = s_i, red[wave][2] = s_p;
  __syncthreads();
  if (threadIdx.x < 3)
    atomicAdd(stats + (size_t)row * 3 + threadIdx.x, red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]);
}

template <typename T>
__global__ __launch_bounds__(256) void mask_grad_kernel(const T* __restrict__ logits, const float* __restrict__ tgt,
                                                        const float* __restrict__ alpha, const float* __restrict__ g_stats,
                                                        T* __restrict__ d_logits, MaskGeom g) {
  constexpr int V = VecTraits<T>::kElems;
  const int row = blockIdx.y;
  const int b = row / g.M, gt = (row % g.M) % g.G;
  const int i = (blockIdx.x * blockDim.x + threadIdx.x) * V;
  if (i >= g.HW) return;
  const float a = alpha[b * g.G + gt];
  const float g0 = g_stats[(size_t)row * 3], g1 = g_stats[(size_t)row * 3 + 1], g2 = g_stats[(size_t)row * 3 + 2];
  float zz[V], tt[V], out[V];
  load_vec<T, V>(logits + (size_t)row * g.HW + i, zz);
  const float* t = tgt + ((size_t)b * g.G + gt) * g.HW + i;
#pragma unroll
  for (int k = 0; k < V; k += 4) load_vec<float, 4>(t + k, tt + k);
#pragma unroll
  for (int k = 0; k < V; ++k) {
    out[k] = mask_dz(zz[k], tt[k], a, g0, g1, g2);
  }
  store_vec<T, V>(d_logits + (size_t)row * g.HW + i, out);
}

template <typename T>
static int launch_mask(bool fwd, const void* logits, const float* tgt, const float* alpha, const float* g_stats, void* out,
                       int B, int M, int G, int HW, hipStream_t st) {
  constexpr int V = VecTraits<T>::kElems;
  if (HW % V != 0 || HW % 4 != 0) return DFINE_E_UNSUPPORTED;
  const MaskGeom g{M, G, HW};
  const long long rows = (long long)B * M;
  if (rows > 0x7fffffffLL) return DFINE_E_UNSUPPORTED;
  if (fwd) {
    int split = (int)((2048 + rows - 1) / rows);  // >= 2048 workgroups
    const int max_split = (HW / V + 255) / 256;
    if (split > max_split) split = max_split;
    if (split < 1) split = 1;
    hipLaunchKernelGGL(mask_stats_kernel<T>, dim3((unsigned)rows, split), dim3(256), 0, st, (const T*)logits, tgt, alpha, (float*)out, g);
  } else {
    if (rows > 65535) return DFINE_E_UNSUPPORTED;
    const int bx = (HW / V + 255) / 256;
    hipLaunchKernelGGL(mask_grad_kernel<T>, dim3(bx, (unsigned)rows), dim3(256), 0, st, (const T*)logits, tgt, alpha, g_stats, (T*)out, g);
  }
  DFINE_LAUNCH_CHECK();
  return 0;
}

}  // namespace dfine

using namespace dfine;

extern "C" int dfine_mask_loss_stats(const void* logits, int dtype, const float* tgt, const float* alpha, float* stats,
                                     int B, int M, int G, int HW, void* stream) {
  if (!logits || !tgt || !alpha || !stats) return DFINE_E_BADARG;
  if (B <= 0 || M <= 0 || G <= 0 || HW <= 0 || M % G != 0) return DFINE_E_BADARG;
  if (dtype == DFINE_DT_F32) return launch_mask<float>(true, logits, tgt, alpha, nullptr, stats, B, M, G, HW, (hipStream_t)stream);
  if (dtype == DFINE_DT_BF16) return launch_mask<bf16_t>(true, logits, tgt, alpha, nullptr, stats, B, M, G, HW, (hipStream_t)stream);
  return DFINE_E_UNSUPPORTED;
}

extern "C" int dfine_mask_loss_grad(const void* logits, int dtype, const float* tgt, const float* alpha,
                                    const float* g_stats, void* d_logits, int B, int M, int G, int HW, void* stream) {
  if (!logits || !tgt || !alpha || !g_stats || !d_logits) return DFINE_E_BADARG;
  if (B <= 0 || M <= 0 || G <= 0 || HW <= 0 || M % G != 0) return DFINE_E_BADARG;
  if (dtype == DFINE_DT_F32) return launch_mask<float>(false, logits, tgt, alpha, g_stats, d_logits, B, M, G, HW, (hipStream_t)stream);
  if (dtype == DFINE_DT_BF16) return launch_mask<bf16_t>(false, logits, tgt, alpha, g_stats, d_logits, B, M, G, HW, (hipStream_t)stream);
  return DFINE_E_UNSUPPORTED;
}
