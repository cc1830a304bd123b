// Hungarian matching on the device: cost matrix, linear sum assignment and the "GO" union of
// matchings, so the criterion never copies a cost matrix to the host (the reference does
// C.cpu() + scipy per head, src/d_fine/matcher.py:112-116, i.e. L+2 device->host syncs a step).
//
// Semantics:
//   cost   -- HungarianMatcher.forward, matcher.py:76-112 (focal class cost, L1, -GIoU, NaN -> 1)
//   lsap   -- scipy.optimize.linear_sum_assignment as called at matcher.py:116 (Crouse 2016
//             shortest augmenting paths in double precision; same column scan order and
//             tie-breaking as scipy's rectangular_lsap, see oracle/lsap.c)
//   union  -- DFINECriterion._get_go_indices, dfine_criterion.py:371-392
// All three are small, latency-bound integer/float kernels: one wave per assignment problem.
#include <math.h>

#include "common.h"

namespace dfine {

__device__ __forceinline__ float finite_or(float x) {  // torch.nan_to_num(nan=1.0) defaults
  if (isnan(x)) return 1.0f;
  if (isinf(x)) return x > 0 ? 3.4028234663852886e38f : -3.4028234663852886e38f;
  return x;
}

// cost[p][g][q] for problem p = head*B + image, ground truth g < num_gt[image], query q.
__global__ __launch_bounds__(256) void match_cost_kernel(
    const float* __restrict__ logits, const float* __restrict__ boxes,
    const int64_t* __restrict__ tgt_labels, const float* __restrict__ tgt_boxes,
    const int32_t* __restrict__ num_gt, float* __restrict__ cost, int N, int B, int Q, int C, int G,
    float alpha, float gamma, float w_class, float w_bbox, float w_giou) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  const int g = blockIdx.y, p = blockIdx.z;
  const int b = p % B;
  if (q >= Q || g >= num_gt[b]) return;
  const float4 pb = *reinterpret_cast<const float4*>(boxes + ((size_t)p * Q + q) * 4);
  const float4 tb = *reinterpret_cast<const float4*>(tgt_boxes + ((size_t)b * G + g) * 4);
  const int label = (int)tgt_labels[(size_t)b * G + g];
  const float logit = logits[((size_t)p * Q + q) * C + label];
  const float prob = 1.0f / (1.0f + expf(-logit));
  const float neg = (1.0f - alpha) * powf(prob, gamma) * (-logf(1.0f - prob + 1e-8f));
  const float pos = alpha * powf(1.0f - prob, gamma) * (-logf(prob + 1e-8f));
  const float c_class = pos - neg;
  const float c_bbox = fabsf(pb.x - tb.x) + fabsf(pb.y - tb.y) + fabsf(pb.z - tb.z) + fabsf(pb.w - tb.w);
  // cxcywh -> xyxy with w, h clamped at 0 (arch/utils.py:59-67)
  const float pw = fmaxf(pb.z, 0.f), ph = fmaxf(pb.w, 0.f), tw = fmaxf(tb.z, 0.f), th = fmaxf(tb.w, 0.f);
  const float px0 = pb.x - 0.5f * pw, py0 = pb.y - 0.5f * ph, px1 = pb.x + 0.5f * pw, py1 = pb.y + 0.5f * ph;
  const float tx0 = tb.x - 0.5f * tw, ty0 = tb.y - 0.5f * th, tx1 = tb.x + 0.5f * tw, ty1 = tb.y + 0.5f * th;
  const float area_p = (px1 - px0) * (py1 - py0), area_t = (tx1 - tx0) * (ty1 - ty0);
  const float iw = fmaxf(fminf(px1, tx1) - fmaxf(px0, tx0), 0.f);
  const float ih = fmaxf(fminf(py1, ty1) - fmaxf(py0, ty0), 0.f);
  const float inter = iw * ih;
  const float uni = area_p + area_t - inter;
  const float iou = inter / uni;
  const float hw = fmaxf(fmaxf(px1, tx1) - fminf(px0, tx0), 0.f);
  const float hh = fmaxf(fmaxf(py1, ty1) - fminf(py0, ty0), 0.f);
  const float hull = hw * hh;
  const float giou = iou - (hull - uni) / hull;
  const float c = w_bbox * c_bbox + w_class * c_class + w_giou * (-giou);
  cost[((size_t)p * G + g) * Q + q] = finite_or(c);
}

// ------------------------------------------------------------------------------------------
// LSAP: one 64-lane workgroup per problem. Rows = ground truths (n <= Q), columns = queries,
// which is the orientation scipy solves after its internal transpose. All state lives in LDS.
// ------------------------------------------------------------------------------------------
struct Cand {  // a scan candidate: path cost, position in `remaining`, column unassigned?
  double val;
  int it;
  int unassigned;
};

// scipy's sequential rule over equal costs: the last unassigned column wins if there is one,
// otherwise the first column in scan order.
__device__ __forceinline__ bool better(const Cand& a, const Cand& b) {
  if (a.val != b.val) return a.val < b.val;
  if (a.unassigned != b.unassigned) return a.unassigned > b.unassigned;
  return a.unassigned ? a.it > b.it : a.it < b.it;
}

__global__ __launch_bounds__(64) void lsap_kernel(const float* __restrict__ cost,
                                                   const int32_t* __restrict__ num_gt,
                                                   int32_t* __restrict__ query_of_gt,
                                                   int32_t* __restrict__ gt_of_query, int B, int G,
                                                   int Q) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int p = blockIdx.x, lane = threadIdx.x;
  const int n = num_gt[p % B];
  double* u = reinterpret_cast<double*>(smem);             // [G]
  double* v = u + G;                                       // [Q]
  double* spc = v + Q;                                     // [Q] shortest path costs
  int* path = reinterpret_cast<int*>(spc + Q);             // [Q]
  int* row4col = path + Q;                                 // [Q]
  int* remaining = row4col + Q;                            // [Q]
  int* col4row = remaining + Q;                            // [G]
  unsigned char* SC = reinterpret_cast<unsigned char*>(col4row + G);  // [Q]
  unsigned char* SR = SC + Q;                                         // [G]
  __shared__ int s_i, s_sink, s_num_remaining;
  __shared__ double s_min;

  const float* cmat = cost + (size_t)p * G * Q;
  for (int j = lane; j < Q; j += 64) { v[j] = 0.0; path[j] = -1; row4col[j] = -1; }
  for (int i = lane; i < G; i += 64) { u[i] = 0.0; col4row[i] = -1; }
  __syncthreads();

  for (int cur = 0; cur < n; ++cur) {
    for (int j = lane; j < Q; j += 64) { remaining[j] = Q - j - 1; spc[j] = INFINITY; SC[j] = 0; }
    for (int i = lane; i < G; i += 64) SR[i] = 0;
    if (lane == 0) { s_i = cur; s_sink = -1; s_num_remaining = Q; s_min = 0.0; }
    __syncthreads();
    while (true) {  // uniform across the wave: every lane reads the same LDS words
      const int i = s_i, num_remaining = s_num_remaining;
      const double min_val = s_min, ui = u[i];
      if (lane == 0) SR[i] = 1;
      Cand best{INFINITY, 0x7fffffff, 0};
      for (int it = lane; it < num_remaining; it += 64) {
        const int j = remaining[it];
        const double r = min_val + (double)cmat[(size_t)i * Q + j] - ui - v[j];
        double s = spc[j];
        if (r < s) { path[j] = i; spc[j] = r; s = r; }
        const Cand c{s, it, row4col[j] == -1};
        if (better(c, best)) best = c;
      }
#pragma unroll
      for (int m = 32; m >= 1; m >>= 1) {
        Cand o;
        o.val = __shfl_xor(best.val, m, 64);
        o.it = __shfl_xor(best.it, m, 64);
        o.unassigned = __shfl_xor(best.unassigned, m, 64);
        if (better(o, best)) best = o;
      }
      __syncthreads();
      if (lane == 0) {
        s_min = best.val;
        if (best.val == INFINITY) {
          s_sink = -2;  // infeasible (cannot happen after nan_to_num; bounded exit anyway)
        } else {
          const int j = remaining[best.it];
          if (row4col[j] == -1) s_sink = j; else s_i = row4col[j];
          SC[j] = 1;
          remaining[best.it] = remaining[num_remaining - 1];
          s_num_remaining = num_remaining - 1;
        }
      }
      __syncthreads();
      if (s_sink != -1) break;
    }
    const int sink = s_sink;
    if (sink < 0) break;
    const double min_val = s_min;
    // dual updates
    for (int i = lane; i < G; i += 64)
      if (SR[i] && i != cur) u[i] += min_val - spc[col4row[i]];
    for (int j = lane; j < Q; j += 64)
      if (SC[j]) v[j] -= min_val - spc[j];
    __syncthreads();
    if (lane == 0) {
      u[cur] += min_val;
      int j = sink;
      while (true) {  // augment along the path; at most cur+1 hops
        const int i = path[j];
        row4col[j] = i;
        const int t = col4row[i];
        col4row[i] = j;
        j = t;
        if (i == cur) break;
      }
    }
    __syncthreads();
  }
  for (int j = lane; j < Q; j += 64) gt_of_query[(size_t)p * Q + j] = row4col[j];
  for (int i = lane; i < G; i += 64) query_of_gt[(size_t)p * G + i] = (i < n) ? col4row[i] : -1;
}

// go[b][q] = the ground truth query q is matched to most often over the H heads (ties: the
// smallest index, i.e. the order a stable count-descending sort of the sorted unique pairs
// gives), -1 if no head matched q.
__global__ __launch_bounds__(256) void go_union_kernel(const int32_t* __restrict__ gt_of_query,
                                                       int32_t* __restrict__ go, int H, int BQ) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= BQ) return;
  int best = -1, best_count = 0;
  for (int h = 0; h < H; ++h) {
    const int g = gt_of_query[(size_t)h * BQ + idx];
    if (g < 0) continue;
    int count = 0;
    for (int k = 0; k < H; ++k) count += gt_of_query[(size_t)k * BQ + idx] == g;
    if (count > best_count || (count == best_count && g < best)) { best = g; best_count = count; }
  }
  go[idx] = best;
}

}  // namespace dfine

using namespace dfine;

extern "C" int dfine_match_cost(const float* logits, const float* boxes, const int64_t* tgt_labels,
                                const float* tgt_boxes, const int32_t* num_gt, float* cost, int N,
                                int B, int Q, int C, int G, float alpha, float gamma, float w_class,
                                float w_bbox, float w_giou, void* stream) {
  if (!logits || !boxes || !tgt_labels || !tgt_boxes || !num_gt || !cost) return DFINE_E_BADARG;
  if (N <= 0 || B <= 0 || Q <= 0 || C <= 0 || G <= 0 || N % B != 0) return DFINE_E_BADARG;
  if (G > 65535 || N > 65535) return DFINE_E_UNSUPPORTED;
  dim3 grid((Q + 255) / 256, G, N);
  hipLaunchKernelGGL(match_cost_kernel, grid, dim3(256), 0, (hipStream_t)stream, logits, boxes,
                     tgt_labels, tgt_boxes, num_gt, cost, N, B, Q, C, G, alpha, gamma, w_class,
                     w_bbox, w_giou);
  DFINE_LAUNCH_CHECK();
  return 0;
}

extern "C" int dfine_lsap(const float* cost, const int32_t* num_gt, int32_t* query_of_gt,
                          int32_t* gt_of_query, int N, int B, int G, int Q, void* stream) {
  if (!cost || !num_gt || !query_of_gt || !gt_of_query) return DFINE_E_BADARG;
  if (N <= 0 || B <= 0 || G <= 0 || Q <= 0 || N % B != 0) return DFINE_E_BADARG;
  if (G > Q) return DFINE_E_UNSUPPORTED;  // more ground truths than queries: not the matcher's regime
  const size_t lds = (size_t)G * 8 + (size_t)Q * 16 + (size_t)Q * 12 + (size_t)G * 4 + Q + G + 16;
  if (lds > 64 * 1024) return DFINE_E_UNSUPPORTED;
  hipLaunchKernelGGL(lsap_kernel, dim3(N), dim3(64), lds, (hipStream_t)stream, cost, num_gt,
                     query_of_gt, gt_of_query, B, G, Q);
  DFINE_LAUNCH_CHECK();
  return 0;
}

extern "C" int dfine_go_union(const int32_t* gt_of_query, int32_t* go, int H, int B, int Q,
                              void* stream) {
  if (!gt_of_query || !go) return DFINE_E_BADARG;
  if (H <= 0 || B <= 0 || Q <= 0) return DFINE_E_BADARG;
  const int BQ = B * Q;
  hipLaunchKernelGGL(go_union_kernel, dim3((BQ + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                     gt_of_query, go, H, BQ);
  DFINE_LAUNCH_CHECK();
  return 0;
}
