// Evaluation post-processing (reference src/dl/train.py:227-319 Trainer.preds_postprocess and its helpers
// src/dl/utils.py:160-183 norm_xywh_to_abs_xyxy, :623-672 scale_boxes(_ratio_kept), :715-771 process_masks,
// :774-789 cleanup_masks) as two device kernels, so an evaluation / inference step leaves the GPU with its
// final records instead of with [B, Q, C] scores, [B, Q, Hm, Wm] masks and a chain of host loops:
//
//   postprocess_topk: per image, sigmoid over the Q*C logits, the K best (score, query, class) triples in
//       descending score order (ties: lower flat index first) and their boxes mapped to the original image
//       (cxcywh in [0,1] -> rounded absolute xyxy on the network input -> original size, with or without
//       letterbox padding), exactly the reference's arithmetic incl. where it runs in float64.
//   postprocess_masks: per kept instance, the reference's two bilinear resizes (mask grid -> network input,
//       crop of the letterbox padding, -> original size; each stage rounded to fp16 like the reference's
//       half-precision tensors), clamp, threshold and the "zero outside its box" clean-up, written as uint8.
#include <hip/hip_fp16.h>

#include "common.h"

namespace dfine {

constexpr int PP_THREADS = 256;
constexpr int PP_CAP = 1024;  // candidates (score >= the K-th best) kept for the final sort; K <= 512

struct BoxMap {
  int proc_h, proc_w, keep_ratio;
};

__device__ __forceinline__ unsigned int score_key(float logit) {
  const float s = 1.f / (1.f + expf(-logit));
  return __float_as_uint(s);  // scores are in [0, 1]: the bit pattern orders like the value
}

// norm_xywh_to_abs_xyxy(to_round=True) on the processed size, then scale_boxes / scale_boxes_ratio_kept
__device__ __forceinline__ void map_box(const float* __restrict__ b, int oh, int ow, const BoxMap& m, float* out) {
  const float W = (float)m.proc_w, H = (float)m.proc_h;
  const float xc = b[0] * W, yc = b[1] * H, bw = b[2] * W, bh = b[3] * H;
  float x0 = fmaxf(floorf(xc - bw / 2.f), 1.f), y0 = fmaxf(floorf(yc - bh / 2.f), 1.f);
  float x1 = fminf(ceilf(xc + bw / 2.f), W - 1.f), y1 = fminf(ceilf(yc + bh / 2.f), H - 1.f);
  if (m.keep_ratio) {
    const double gain = fmin((double)m.proc_h / (double)oh, (double)m.proc_w / (double)ow);
    const float padw = (float)rint(((double)m.proc_w - (double)ow * gain) / 2.0 - 0.1);  // Python round(): half to even
    const float padh = (float)rint(((double)m.proc_h - (double)oh * gain) / 2.0 - 0.1);
    x0 = (float)((double)(x0 - padw) / gain), x1 = (float)((double)(x1 - padw) / gain);
    y0 = (float)((double)(y0 - padh) / gain), y1 = (float)((double)(y1 - padh) / gain);
    x0 = fminf(fmaxf(x0, 0.f), (float)ow), x1 = fminf(fmaxf(x1, 0.f), (float)ow);
    y0 = fminf(fmaxf(y0, 0.f), (float)oh), y1 = fminf(fmaxf(y1, 0.f), (float)oh);
  } else {
    const double sx = (double)ow / (double)m.proc_w, sy = (double)oh / (double)m.proc_h;
    x0 = (float)((double)x0 * sx), x1 = (float)((double)x1 * sx);
    y0 = (float)((double)y0 * sy), y1 = (float)((double)y1 * sy);
  }
  out[0] = x0, out[1] = y0, out[2] = x1, out[3] = y1;
}

template <typename T>
__global__ __launch_bounds__(PP_THREADS) void postprocess_topk_kernel(const T* __restrict__ logits, const float* __restrict__ boxes,
                                                                     const long long* __restrict__ orig_sizes, float* __restrict__ scores,
                                                                     long long* __restrict__ labels, long long* __restrict__ qidx,
                                                                     float* __restrict__ out_boxes, int Q, int C, int K, BoxMap bm) {
  __shared__ unsigned int hist[256];
  __shared__ unsigned int s_prefix, s_need, s_count;
  __shared__ unsigned long long cand[PP_CAP];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int N = Q * C;
  const T* lg = logits + (size_t)b * N;
  // ---- radix select, most significant byte first: the K-th largest key
  unsigned int prefix = 0, need = (unsigned)K;
  for (int pass = 0; pass < 4; ++pass) {
    const int shift = 24 - 8 * pass;
    const unsigned int mask = pass == 0 ? 0u : (0xffffffffu << (shift + 8));
    hist[tid] = 0;
    __syncthreads();
    for (int i = tid; i < N; i += PP_THREADS) {
      const unsigned int key = score_key(VecTraits<T>::load1(lg + i));
      if ((key & mask) == prefix) atomicAdd(&hist[(key >> shift) & 0xff], 1u);
    }
    __syncthreads();
    if (tid == 0) {
      unsigned int acc = 0;
      int bin = 255;
      for (; bin > 0; --bin) {
        if (acc + hist[bin] >= need) break;
        acc += hist[bin];
      }
      s_prefix = prefix | ((unsigned)bin << shift);
      s_need = need - acc;
    }
    __syncthreads();
    prefix = s_prefix, need = s_need;
    __syncthreads();
  }
  // ---- candidates: every key >= the K-th largest (ties beyond the capacity are dropped from the high-index end)
  const unsigned int kth = prefix;
  if (tid == 0) s_count = 0;
  for (int i = tid; i < PP_CAP; i += PP_THREADS) cand[i] = 0ull;
  __syncthreads();
  for (int i0 = 0; i0 < N; i0 += PP_THREADS) {  // index order, so that ties at the threshold keep the lowest indices
    const int i = i0 + tid;
    const unsigned int key = i < N ? score_key(VecTraits<T>::load1(lg + i)) : 0u;
    if (i < N && key >= kth) {
      const unsigned int slot = atomicAdd(&s_count, 1u);
      if (slot < PP_CAP) cand[slot] = ((unsigned long long)key << 32) | (unsigned long long)(0xffffffffu - (unsigned)i);
    }
  }
  __syncthreads();
  // ---- bitonic sort, descending by (score, -index)
  for (int k = 2; k <= PP_CAP; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = tid; i < PP_CAP; i += PP_THREADS) {
        const int ixj = i ^ j;
        if (ixj > i) {
          const unsigned long long a = cand[i], c = cand[ixj];
          const bool desc = (i & k) == 0;
          if (desc ? a < c : a > c) cand[i] = c, cand[ixj] = a;
        }
      }
      __syncthreads();
    }
  const int oh = (int)orig_sizes[2 * b], ow = (int)orig_sizes[2 * b + 1];
  for (int r = tid; r < K; r += PP_THREADS) {
    const unsigned long long e = cand[r];
    const unsigned int idx = 0xffffffffu - (unsigned)(e & 0xffffffffull);
    const size_t o = (size_t)b * K + r;
    const bool ok = e != 0ull && idx < (unsigned)N;  // fewer than K candidates only when K > N
    const int q = ok ? (int)(idx / (unsigned)C) : 0;
    scores[o] = ok ? __uint_as_float((unsigned)(e >> 32)) : 0.f;
    labels[o] = ok ? (long long)(idx - (unsigned)q * (unsigned)C) : 0;
    qidx[o] = q;
    float bx[4];
    map_box(boxes + ((size_t)b * Q + q) * 4, oh, ow, bm, bx);
#pragma unroll
    for (int c = 0; c < 4; ++c) out_boxes[o * 4 + c] = bx[c];
  }
}

// PyTorch's bilinear source index (align_corners=False): max(scale * (dst + 0.5) - 0.5, 0)
__device__ __forceinline__ void src_index(int dst, float scale, int in_size, int& i0, int& i1, float& l1) {
  float s = scale * ((float)dst + 0.5f) - 0.5f;
  s = s < 0.f ? 0.f : s;
  i0 = (int)s;
  if (i0 > in_size - 1) i0 = in_size - 1;
  i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
  l1 = s - (float)i0;
}

__device__ __forceinline__ float round_half(float v) { return __half2float(__float2half_rn(v)); }

struct MaskMap {
  int Hm, Wm, proc_h, proc_w, crop_y, crop_x, crop_h, crop_w, H0, W0;
  float thresh;
};

template <typename T>
__global__ __launch_bounds__(256) void postprocess_masks_kernel(const T* __restrict__ probs, const long long* __restrict__ qidx,
                                                               const float* __restrict__ boxes, unsigned char* __restrict__ out, int n_inst,
                                                               MaskMap mm) {
  const int n = blockIdx.y;
  const int pix = blockIdx.x * blockDim.x + threadIdx.x;
  if (pix >= mm.H0 * mm.W0) return;
  const int y = pix / mm.W0, x = pix - y * mm.W0;
  const T* src = probs + (size_t)qidx[n] * mm.Hm * mm.Wm;
  // stage 2: (crop of the processed-size map) -> original size
  const float s2y = (float)mm.crop_h / (float)mm.H0, s2x = (float)mm.crop_w / (float)mm.W0;
  int y0, y1, x0, x1;
  float ly, lx;
  src_index(y, s2y, mm.crop_h, y0, y1, ly);
  src_index(x, s2x, mm.crop_w, x0, x1, lx);
  // stage 1: mask grid -> processed size, evaluated at the four stage-2 taps, each rounded to fp16
  const float s1y = (float)mm.Hm / (float)mm.proc_h, s1x = (float)mm.Wm / (float)mm.proc_w;
  auto stage1 = [&](int py, int px) {
    int a0, a1, b0, b1;
    float la, lb;
    src_index(py + mm.crop_y, s1y, mm.Hm, a0, a1, la);
    src_index(px + mm.crop_x, s1x, mm.Wm, b0, b1, lb);
    const float v00 = round_half(VecTraits<T>::load1(src + a0 * mm.Wm + b0)), v01 = round_half(VecTraits<T>::load1(src + a0 * mm.Wm + b1));
    const float v10 = round_half(VecTraits<T>::load1(src + a1 * mm.Wm + b0)), v11 = round_half(VecTraits<T>::load1(src + a1 * mm.Wm + b1));
    return round_half((1.f - la) * ((1.f - lb) * v00 + lb * v01) + la * ((1.f - lb) * v10 + lb * v11));
  };
  const float v = round_half((1.f - ly) * ((1.f - lx) * stage1(y0, x0) + lx * stage1(y0, x1)) +
                             ly * ((1.f - lx) * stage1(y1, x0) + lx * stage1(y1, x1)));
  const float c = fminf(fmaxf(v, 0.f), 1.f);
  const float* bx = boxes + 4 * n;
  const bool inside = (float)x >= bx[0] && (float)x < bx[2] && (float)y >= bx[1] && (float)y < bx[3];
  out[(size_t)n * mm.H0 * mm.W0 + pix] = (c >= mm.thresh && inside) ? 1 : 0;
}

}  // namespace dfine

using namespace dfine;

extern "C" int dfine_postprocess_topk(const void* logits, int dtype, const float* boxes, const long long* orig_sizes, float* scores,
                                      long long* labels, long long* qidx, float* out_boxes, int B, int Q, int C, int K, int proc_h,
                                      int proc_w, int keep_ratio, void* stream) {
  if (!logits || !boxes || !orig_sizes || !scores || !labels || !qidx || !out_boxes) return DFINE_E_BADARG;
  if (B <= 0 || Q <= 0 || C <= 0 || K <= 0 || proc_h <= 0 || proc_w <= 0) return DFINE_E_BADARG;
  if (K > 512 || K > Q * C || (long long)Q * C > 0x7fffffffLL) return DFINE_E_UNSUPPORTED;
  const BoxMap bm{proc_h, proc_w, keep_ratio};
  if (dtype == DFINE_DT_F32)
    hipLaunchKernelGGL(postprocess_topk_kernel<float>, dim3(B), dim3(PP_THREADS), 0, (hipStream_t)stream, (const float*)logits, boxes,
                       orig_sizes, scores, labels, qidx, out_boxes, Q, C, K, bm);
  else if (dtype == DFINE_DT_BF16)
    hipLaunchKernelGGL(postprocess_topk_kernel<bf16_t>, dim3(B), dim3(PP_THREADS), 0, (hipStream_t)stream, (const bf16_t*)logits, boxes,
                       orig_sizes, scores, labels, qidx, out_boxes, Q, C, K, bm);
  else
    return DFINE_E_UNSUPPORTED;
  DFINE_LAUNCH_CHECK();
  return 0;
}

extern "C" int dfine_postprocess_masks(const void* probs, int dtype, const long long* qidx, const float* boxes, unsigned char* out,
                                       int n_inst, int Hm, int Wm, int proc_h, int proc_w, int H0, int W0, int keep_ratio, float thresh,
                                       void* stream) {
  if (!probs || !qidx || !boxes || !out) return DFINE_E_BADARG;
  if (n_inst <= 0 || Hm <= 0 || Wm <= 0 || proc_h <= 0 || proc_w <= 0 || H0 <= 0 || W0 <= 0) return DFINE_E_BADARG;
  if (n_inst > 65535) return DFINE_E_UNSUPPORTED;
  MaskMap mm{Hm, Wm, proc_h, proc_w, 0, 0, proc_h, proc_w, H0, W0, thresh};
  if (keep_ratio) {  // the letterbox padding process_masks removes before the final resize (src/dl/utils.py:742-755)
    const double gain = fmin((double)proc_h / H0, (double)proc_w / W0);
    const int padw = (int)rint((proc_w - W0 * gain) / 2 - 0.1), padh = (int)rint((proc_h - H0 * gain) / 2 - 0.1);
    mm.crop_y = padh > 0 ? padh : 0, mm.crop_x = padw > 0 ? padw : 0;
    mm.crop_h = proc_h - 2 * mm.crop_y, mm.crop_w = proc_w - 2 * mm.crop_x;
    if (mm.crop_h <= 0 || mm.crop_w <= 0) return DFINE_E_BADARG;
  }
  const dim3 grid((unsigned)((H0 * W0 + 255) / 256), (unsigned)n_inst);
  if (dtype == DFINE_DT_F32)
    hipLaunchKernelGGL(postprocess_masks_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)probs, qidx, boxes, out,
                       n_inst, mm);
  else if (dtype == DFINE_DT_BF16)
    hipLaunchKernelGGL(postprocess_masks_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)probs, qidx, boxes, out,
                       n_inst, mm);
  else
    return DFINE_E_UNSUPPORTED;
  DFINE_LAUNCH_CHECK();
  return 0;
}
