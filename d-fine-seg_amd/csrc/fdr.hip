// FDR head of one decoder layer, fused: for every query and box side
//     p = softmax(corners[side])                      (dfine_decoder.py:276-296 Integral, :298-313 LQE)
//     d[side] = sum_j p_j * W(n)_j                    Integral: F.linear(prob, project)
//     box = distance2bbox(ref_box, d, reg_scale)      (arch/utils.py:119-142), cxcywh
//     stat[side] = (top-4 of p, their mean)           LQE input (prob.topk(4), cat with the mean)
// and the matching backward (softmax' (W-weighted box gradient + the top-4 scatter)). The stock path is
// ~30 launches per layer: two softmaxes, a [.,33]x[33] matmul, a radix top-k of 4 out of 33 that alone
// takes 290 us, stack/cat/mean and all their backward nodes.
//
// One lane per (query, side); the 4 sides of a query sit in 4 adjacent lanes, so the box is assembled
// with quad shuffles. Nothing is kept in arrays: the 33 logits are re-read (L1/L2) for each pass.
#include "common.h"

namespace dfine {

constexpr int FDR_K = 4;  // LQE(k=4, ...), dfine_decoder.py:598

struct SoftmaxNorm {
  float max, inv_sum;
};

__device__ __forceinline__ SoftmaxNorm softmax_norm(const float* __restrict__ c, int n) {
  float m = c[0];
  for (int j = 1; j < n; ++j) m = fmaxf(m, c[j]);
  float s = 0.f;
  for (int j = 0; j < n; ++j) s += expf(c[j] - m);
  return {m, 1.f / s};
}

__global__ __launch_bounds__(256) void fdr_head_fwd_kernel(const float* __restrict__ corners,
                                                           const float* __restrict__ ref_boxes,
                                                           const float* __restrict__ W, float* __restrict__ boxes,
                                                           float* __restrict__ stat, uint8_t* __restrict__ top_idx,
                                                           long long rows, int n_bins, float reg_scale) {
  // the workgroup's 256 x n_bins logits are one contiguous block: through LDS with coalesced loads (see fdr_head_bwd_kernel)
  extern __shared__ float tile[];
  const long long t0 = (long long)blockIdx.x * blockDim.x, n_items = rows * 4;
  const long long span = min((long long)blockDim.x, n_items - t0) * n_bins;
  for (long long i = threadIdx.x; i < span; i += blockDim.x) tile[i] = corners[t0 * n_bins + i];
  __syncthreads();
  const long long t = t0 + threadIdx.x;  // (row, side)
  const bool live = t < n_items;
  const long long tt = live ? t : n_items - 1;  // idle lanes of the last wave mirror the last item (shuffles below)
  const long long row = tt >> 2;
  const int side = (int)(tt & 3);
  const float* c = tile + (size_t)(tt - t0) * n_bins;
  const SoftmaxNorm nm = softmax_norm(c, n_bins);
  float d = 0.f;
  float v[FDR_K];
  int id[FDR_K];
#pragma unroll
  for (int k = 0; k < FDR_K; ++k) v[k] = -1.f, id[k] = 0;
  for (int j = 0; j < n_bins; ++j) {
    const float p = expf(c[j] - nm.max) * nm.inv_sum;
    d += p * W[j];
    // insert into the descending top-4 (strict >: the earlier bin wins a tie)
    float pv = p;
    int pi = j;
#pragma unroll
    for (int k = 0; k < FDR_K; ++k) {
      if (pv > v[k]) {
        const float tv = v[k];
        const int ti = id[k];
        v[k] = pv, id[k] = pi;
        pv = tv, pi = ti;
      }
    }
  }
  // the four sides' distances of this query: lanes 4q .. 4q+3
  const int lane = threadIdx.x & 63, q0 = lane & ~3;
  const float d0 = __shfl(d, q0, 64), d1 = __shfl(d, q0 + 1, 64), d2 = __shfl(d, q0 + 2, 64), d3 = __shfl(d, q0 + 3, 64);
  if (!live) return;
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < FDR_K; ++k) {
    stat[tt * (FDR_K + 1) + k] = v[k];
    top_idx[tt * FDR_K + k] = (uint8_t)id[k];
    s += v[k];
  }
  stat[tt * (FDR_K + 1) + FDR_K] = s / FDR_K;
  const float* r = ref_boxes + row * 4;
  const float half = 0.5f * reg_scale;
  const float wq = r[2] / reg_scale, hq = r[3] / reg_scale;
  const float x1 = r[0] - (half + d0) * wq, y1 = r[1] - (half + d1) * hq;
  const float x2 = r[0] + (half + d2) * wq, y2 = r[1] + (half + d3) * hq;
  const float out = side == 0 ? (x1 + x2) / 2 : side == 1 ? (y1 + y2) / 2 : side == 2 ? x2 - x1 : y2 - y1;
  boxes[tt] = out;
}

// One thread per (row, side); the n_bins logits of 256 consecutive (row, side) pairs are ONE contiguous block of memory, so
// the workgroup moves it through LDS with coalesced 4-byte accesses in both directions (a thread's own row sits at pitch
// n_bins, odd for the model's 33 bins: conflict-free). Per-thread row pointers into global memory -- 132-byte rows, 132 bytes
// apart -- made every wave store touch 64 different cache lines: 53 MB of HBM writes for an 18 MB gradient (PMC, round 1).
__global__ __launch_bounds__(256) void fdr_head_bwd_kernel(const float* __restrict__ corners,
                                                           const float* __restrict__ ref_boxes,
                                                           const float* __restrict__ W,
                                                           const uint8_t* __restrict__ top_idx,
                                                           const float* __restrict__ g_boxes,
                                                           const float* __restrict__ g_stat,
                                                           float* __restrict__ g_corners, long long rows, int n_bins,
                                                           float reg_scale) {
  extern __shared__ float tile[];  // [256][n_bins]
  const long long t0 = (long long)blockIdx.x * blockDim.x;
  const long long n_items = rows * 4;
  const long long span = min((long long)blockDim.x, n_items - t0) * n_bins;  // floats of this workgroup's block
  for (long long i = threadIdx.x; i < span; i += blockDim.x) tile[i] = corners[t0 * n_bins + i];
  __syncthreads();
  const long long t = t0 + threadIdx.x;
  if (t < n_items) {
    const long long row = t >> 2;
    const int side = (int)(t & 3);
    float* c = tile + (size_t)threadIdx.x * n_bins;
    // d(loss)/d(distance of this side) from the box gradient
    float gd = 0.f;
    if (g_boxes) {
      const float* g = g_boxes + row * 4;
      const float* r = ref_boxes + row * 4;
      const float wq = r[2] / reg_scale, hq = r[3] / reg_scale;
      // box = ((x1+x2)/2, (y1+y2)/2, x2-x1, y2-y1); x1 = cx - (half+d0) wq, x2 = cx + (half+d2) wq, ...
      gd = side == 0 ? -(0.5f * g[0] - g[2]) * wq
         : side == 1 ? -(0.5f * g[1] - g[3]) * hq
         : side == 2 ? (0.5f * g[0] + g[2]) * wq
                     : (0.5f * g[1] + g[3]) * hq;
    }
    float gtop[FDR_K];
    int id[FDR_K];
#pragma unroll
    for (int k = 0; k < FDR_K; ++k) {
      id[k] = top_idx[t * FDR_K + k];
      gtop[k] = g_stat ? g_stat[t * (FDR_K + 1) + k] + g_stat[t * (FDR_K + 1) + FDR_K] * (1.f / FDR_K) : 0.f;
    }
    const SoftmaxNorm nm = softmax_norm(c, n_bins);
    auto dprob = [&](int j) {
      float g = gd * W[j];
#pragma unroll
      for (int k = 0; k < FDR_K; ++k) g += (id[k] == j) ? gtop[k] : 0.f;
      return g;
    };
    float dot = 0.f;
    for (int j = 0; j < n_bins; ++j) dot += expf(c[j] - nm.max) * nm.inv_sum * dprob(j);
    for (int j = 0; j < n_bins; ++j) c[j] = expf(c[j] - nm.max) * nm.inv_sum * (dprob(j) - dot);  // in place: own row only
  }
  __syncthreads();
  for (long long i = threadIdx.x; i < span; i += blockDim.x) g_corners[t0 * n_bins + i] = tile[i];
}

}  // namespace dfine

using namespace dfine;

extern "C" int dfine_fdr_head_fwd(const float* corners, const float* ref_boxes, const float* W, float* boxes,
                                  float* stat, unsigned char* top_idx, long long rows, int n_bins, int k,
                                  float reg_scale, void* stream) {
  if (!corners || !ref_boxes || !W || !boxes || !stat || !top_idx) return DFINE_E_BADARG;
  if (rows < 0 || n_bins <= 0 || reg_scale == 0.f) return DFINE_E_BADARG;
  if (k != FDR_K || n_bins < FDR_K || n_bins > 60) return DFINE_E_UNSUPPORTED;  // 256 rows of n_bins floats in LDS (<= 60 KB)
  if (rows == 0) return 0;
  const long long blocks = (rows * 4 + 255) / 256;
  if (blocks > 0x7fffffffLL) return DFINE_E_UNSUPPORTED;
  hipLaunchKernelGGL(fdr_head_fwd_kernel, dim3((unsigned)blocks), dim3(256), (size_t)256 * n_bins * sizeof(float), (hipStream_t)stream, corners,
                     ref_boxes, W, boxes, stat, top_idx, rows, n_bins, fabsf(reg_scale));
  DFINE_LAUNCH_CHECK();
  return 0;
}

extern "C" int dfine_fdr_head_bwd(const float* corners, const float* ref_boxes, const float* W,
                                  const unsigned char* top_idx, const float* g_boxes, const float* g_stat,
                                  float* g_corners, long long rows, int n_bins, int k, float reg_scale,
                                  void* stream) {
  if (!corners || !ref_boxes || !W || !top_idx || !g_corners) return DFINE_E_BADARG;
  if (rows < 0 || n_bins <= 0 || reg_scale == 0.f) return DFINE_E_BADARG;
  if (k != FDR_K || n_bins < FDR_K || n_bins > 60) return DFINE_E_UNSUPPORTED;  // 256 rows of n_bins floats in LDS (<= 60 KB)
  if (rows == 0) return 0;
  const long long blocks = (rows * 4 + 255) / 256;
  if (blocks > 0x7fffffffLL) return DFINE_E_UNSUPPORTED;
  hipLaunchKernelGGL(fdr_head_bwd_kernel, dim3((unsigned)blocks), dim3(256), (size_t)256 * n_bins * sizeof(float), (hipStream_t)stream, corners,
                     ref_boxes, W, top_idx, g_boxes, g_stat, g_corners, rows, n_bins, fabsf(reg_scale));
  DFINE_LAUNCH_CHECK();
  return 0;
}
