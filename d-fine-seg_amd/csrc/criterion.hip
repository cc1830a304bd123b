// D-FINE criterion for one prediction head, forward AND gradients in one pass: varifocal loss,
// L1 + GIoU box loss, fine-grained localisation loss (FGL) and decoupled distillation focal loss
// (DDF). Semantics: DFINECriterion.loss_labels_vfl / loss_boxes / loss_local
// (reference src/d_fine/dfine_criterion.py:92-237) with bbox2distance / translate_gt
// (src/d_fine/arch/utils.py:267-354) and box_iou / generalized_box_iou (utils.py:12-51), evaluated on
// the dense matchings produced by match.hip: assign[b][q] = ground-truth index or -1.
//
// One wavefront per (image, query): lanes sweep the classes (VFL) and the 33 distribution bins of
// each box edge (FGL/DDF); soft-max statistics are wave reductions. The stock path runs ~400 small
// elementwise launches per head (x 11 heads); this is 1 launch per head. Latency-bound, tiny bytes.
#include "common.h"

namespace dfine {

struct CritArgs {
  int B, Q, C, G, NB;       // NB = reg_max + 1 bins per edge
  int has_local, has_ddf;   // corners present / teacher present
  float alpha, gamma;       // varifocal
  float reg_scale, temp;    // FDR reg_scale, DDF temperature
  // elements between consecutive images of each INPUT (rows of an image are dense): the decoder's outputs are [B, Nd + Q, .]
  // and the heads are given the [:, :Nd] / [:, Nd:] halves in place instead of contiguous copies
  long long sb_logits, sb_boxes, sb_corners, sb_ref, sb_tcorners, sb_tlogits;
};

__device__ __forceinline__ float wave_sum(float v) { return group_sum<64>(v); }
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, __shfl_xor(v, m, 64));
  return v;
}

struct Box { float x0, y0, x1, y1; };
__device__ __forceinline__ Box to_xyxy(const float4 b) {  // width/height clamped at 0 (utils.py:59-67)
  const float w = fmaxf(b.z, 0.f), h = fmaxf(b.w, 0.f);
  return Box{b.x - 0.5f * w, b.y - 0.5f * h, b.x + 0.5f * w, b.y + 0.5f * h};
}
__device__ __forceinline__ float iou_of(const Box& p, const Box& t) {
  const float iw = fmaxf(fminf(p.x1, t.x1) - fmaxf(p.x0, t.x0), 0.f);
  const float ih = fmaxf(fminf(p.y1, t.y1) - fmaxf(p.y0, t.y0), 0.f);
  const float inter = iw * ih;
  return inter / ((p.x1 - p.x0) * (p.y1 - p.y0) + (t.x1 - t.x0) * (t.y1 - t.y0) - inter);
}

// 1 - GIoU(pred, target) and its gradient w.r.t. the predicted cxcywh box.
__device__ __forceinline__ float giou_loss_grad(const float4 pb, const Box& p, const Box& t, float4* grad) {
  const float ix0 = fmaxf(p.x0, t.x0), iy0 = fmaxf(p.y0, t.y0), ix1 = fminf(p.x1, t.x1), iy1 = fminf(p.y1, t.y1);
  const float iw = fmaxf(ix1 - ix0, 0.f), ih = fmaxf(iy1 - iy0, 0.f);
  const float inter = iw * ih;
  const float pw = p.x1 - p.x0, ph = p.y1 - p.y0;
  const float uni = pw * ph + (t.x1 - t.x0) * (t.y1 - t.y0) - inter;
  const float hx0 = fminf(p.x0, t.x0), hy0 = fminf(p.y0, t.y0), hx1 = fmaxf(p.x1, t.x1), hy1 = fmaxf(p.y1, t.y1);
  const float hw = fmaxf(hx1 - hx0, 0.f), hh = fmaxf(hy1 - hy0, 0.f);
  const float hull = hw * hh;
  const float giou = inter / uni - (hull - uni) / hull;
  // reverse mode, d(loss)/d(giou) = -1
  const float d_giou = -1.f;
  float d_uni = d_giou / hull - d_giou * inter / (uni * uni);
  const float d_hull = -d_giou * uni / (hull * hull);
  float d_inter = d_giou / uni - d_uni;
  const float d_area_p = d_uni;
  const float d_iw = d_inter * ih, d_ih = d_inter * iw;
  float d_x0 = 0.f, d_y0 = 0.f, d_x1 = 0.f, d_y1 = 0.f;
  if (ix1 - ix0 > 0.f) {
    if (p.x1 < t.x1) d_x1 += d_iw;
    if (p.x0 > t.x0) d_x0 -= d_iw;
  }
  if (iy1 - iy0 > 0.f) {
    if (p.y1 < t.y1) d_y1 += d_ih;
    if (p.y0 > t.y0) d_y0 -= d_ih;
  }
  const float d_hw = d_hull * hh, d_hh = d_hull * hw;
  if (p.x1 > t.x1) d_x1 += d_hw;
  if (p.x0 < t.x0) d_x0 -= d_hw;
  if (p.y1 > t.y1) d_y1 += d_hh;
  if (p.y0 < t.y0) d_y0 -= d_hh;
  d_x1 += d_area_p * ph;
  d_x0 -= d_area_p * ph;
  d_y1 += d_area_p * pw;
  d_y0 -= d_area_p * pw;
  grad->x = d_x0 + d_x1;
  grad->y = d_y0 + d_y1;
  grad->z = pb.z > 0.f ? 0.5f * (d_x1 - d_x0) : 0.f;
  grad->w = pb.w > 0.f ? 0.5f * (d_y1 - d_y0) : 0.f;
  return 1.f - giou;
}

constexpr int CRIT_TERM_REPLICAS = 64;  // terms: f32 [64][8], columns 0..4 used

template <typename T>
__device__ __forceinline__ void criterion_head_body(
    const T* __restrict__ logits, const float* __restrict__ boxes, const float* __restrict__ corners,
    const float* __restrict__ ref, const float* __restrict__ teacher_corners, const T* __restrict__ teacher_logits,
    const int32_t* __restrict__ own, const int32_t* __restrict__ go, const int64_t* __restrict__ tgt_labels,
    const float* __restrict__ tgt_boxes, const float* __restrict__ Wn, const float* __restrict__ scal,
    float* __restrict__ terms, T* __restrict__ d_logits, float* __restrict__ d_box_l1, float* __restrict__ d_box_giou,
    float* __restrict__ d_corners_fgl, float* __restrict__ d_corners_ddf, const CritArgs& a) {
  // scal: [0] 1/nb for vfl, [1] 1/nb for boxes+local, [2] ddf coefficient (matched), [3] (unmatched),
  //       [4..8] loss weights (vfl, bbox, giou, fgl, ddf), already folded into the normalisers below
  __shared__ float part[4][5];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row = blockIdx.x * 4 + wave;  // (b, q)
  float t_vfl = 0.f, t_l1 = 0.f, t_giou = 0.f, t_fgl = 0.f, t_ddf = 0.f;
  if (row < a.B * a.Q) {
    const int b = row / a.Q, q = row - b * a.Q;
    logits += (size_t)b * a.sb_logits + (size_t)q * a.C;  // this row's inputs; outputs stay dense [B * Q, .]
    boxes += (size_t)b * a.sb_boxes + (size_t)q * 4;
    if (a.has_local) {
      corners += (size_t)b * a.sb_corners + (size_t)q * 4 * a.NB;
      ref += (size_t)b * a.sb_ref + (size_t)q * 4;
    }
    if (a.has_ddf) {
      teacher_corners += (size_t)b * a.sb_tcorners + (size_t)q * 4 * a.NB;
      teacher_logits += (size_t)b * a.sb_tlogits + (size_t)q * a.C;
    }
    const float inv_nb_cls = scal[0] * scal[4];  // weight_dict["loss_vfl"] / num_boxes
    const float inv_nb_l1 = scal[1] * scal[5], inv_nb_giou = scal[1] * scal[6], inv_nb_fgl = scal[1] * scal[7];
    const float4 pb = *reinterpret_cast<const float4*>(boxes);
    const Box pxy = to_xyxy(pb);
    // ---------------------------------------------------------------- varifocal (own matching)
    {
      const int g = own[row];
      float iou = 0.f;
      int label = -1;
      if (g >= 0) {
        iou = iou_of(pxy, to_xyxy(*reinterpret_cast<const float4*>(tgt_boxes + ((size_t)b * a.G + g) * 4)));
        label = (int)tgt_labels[(size_t)b * a.G + g];
      }
      float acc = 0.f;
      for (int c = lane; c < a.C; c += 64) {
        const float x = VecTraits<T>::load1(logits + c);
        const float p = 1.f / (1.f + __expf(-x));
        const bool hit = c == label;
        const float t = hit ? iou : 0.f;
        const float w = hit ? t : a.alpha * powf(p, a.gamma);
        const float bce = fmaxf(x, 0.f) - x * t + log1pf(__expf(-fabsf(x)));
        acc += w * bce;
        store1(d_logits + (size_t)row * a.C + c, w * (p - t) * inv_nb_cls);
      }
      t_vfl = wave_sum(acc) * inv_nb_cls;
    }
    // ---------------------------------------------------------------- boxes + local (GO matching)
    const int g = go[row];
    const bool matched = g >= 0;
    float4 tbox = make_float4(0.5f, 0.5f, 1.f, 1.f);
    if (matched) tbox = *reinterpret_cast<const float4*>(tgt_boxes + ((size_t)b * a.G + g) * 4);
    const Box txy = to_xyxy(tbox);
    float4 gl1 = make_float4(0.f, 0.f, 0.f, 0.f), ggi = gl1;
    if (matched) {
      t_l1 = (fabsf(pb.x - tbox.x) + fabsf(pb.y - tbox.y) + fabsf(pb.z - tbox.z) + fabsf(pb.w - tbox.w)) * inv_nb_l1;
      const float sx = pb.x > tbox.x ? 1.f : (pb.x < tbox.x ? -1.f : 0.f), sy = pb.y > tbox.y ? 1.f : (pb.y < tbox.y ? -1.f : 0.f);
      const float sz = pb.z > tbox.z ? 1.f : (pb.z < tbox.z ? -1.f : 0.f), sw = pb.w > tbox.w ? 1.f : (pb.w < tbox.w ? -1.f : 0.f);
      gl1 = make_float4(sx * inv_nb_l1, sy * inv_nb_l1, sz * inv_nb_l1, sw * inv_nb_l1);
      t_giou = giou_loss_grad(pb, pxy, txy, &ggi) * inv_nb_giou;
      ggi.x *= inv_nb_giou; ggi.y *= inv_nb_giou; ggi.z *= inv_nb_giou; ggi.w *= inv_nb_giou;
    }
    if (lane == 0) {
      *reinterpret_cast<float4*>(d_box_l1 + (size_t)row * 4) = gl1;
      *reinterpret_cast<float4*>(d_box_giou + (size_t)row * 4) = ggi;
    }
    if (a.has_local) {
      const float iou = matched ? iou_of(pxy, txy) : 0.f;
      float w_ddf = iou;  // DDF weight: IoU where matched, else the teacher's best class probability
      if (a.has_ddf && !matched) {
        float mx = -INFINITY;
        for (int c = lane; c < a.C; c += 64) mx = fmaxf(mx, VecTraits<T>::load1(teacher_logits + c));
        mx = wave_max(mx);
        w_ddf = 1.f / (1.f + __expf(-mx));
      }
      const float ddf_coef = (matched ? scal[2] : scal[3]) * scal[8];
      const float4 rp = *reinterpret_cast<const float4*>(ref);
      const float rs = a.reg_scale;
      const float wq = rp.z / rs + 1e-16f, hq = rp.w / rs + 1e-16f;
      const float dist[4] = {(rp.x - txy.x0) / wq - 0.5f * rs, (rp.y - txy.y0) / hq - 0.5f * rs,
                             (txy.x1 - rp.x) / wq - 0.5f * rs, (txy.y1 - rp.y) / hq - 0.5f * rs};
      const bool live = lane < a.NB;
      const float Wj = live ? Wn[lane] : INFINITY;
      const int reg_max = a.NB - 1;
      const float temp = a.temp, inv_T = 1.f / a.temp;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const size_t off = ((size_t)row * 4 + e) * a.NB + lane;  // dense outputs
        const size_t in_off = (size_t)e * a.NB + lane;           // this row's inputs
        const float x = live ? corners[in_off] : -INFINITY;
        const float mx = wave_max(x);
        const float ex = live ? __expf(x - mx) : 0.f;
        const float se = wave_sum(ex);
        const float logp = x - mx - __logf(se);
        float g_fgl = 0.f;
        if (matched) {  // translate_gt: left bin, linear weights to its two neighbours
          const float d = dist[e];
          const int idx = __popcll(__ballot(live && Wj <= d)) - 1;
          int left;
          float wl, wr;
          if (idx < 0) { left = 0; wl = 1.f; wr = 0.f; }
          else if (idx >= reg_max) { left = reg_max - 1; wl = 0.f; wr = 1.f; }
          else {
            left = idx;
            const float dl = fabsf(d - __shfl(Wj, idx, 64)), dr = fabsf(__shfl(Wj, idx + 1, 64) - d);
            wr = dl / (dl + dr);
            wl = 1.f - wr;
          }
          const float contrib = (lane == left ? wl : 0.f) + (lane == left + 1 ? wr : 0.f);
          t_fgl += -wave_sum(live ? contrib * logp : 0.f) * iou * inv_nb_fgl;
          g_fgl = live ? (ex / se * (wl + wr) - contrib) * iou * inv_nb_fgl : 0.f;
        }
        if (live) d_corners_fgl[off] = g_fgl;
        if (a.has_ddf) {
          const float tc = live ? teacher_corners[in_off] : -INFINITY;
          const float es = live ? __expf((x - mx) * inv_T) : 0.f;  // student at temperature T (same max)
          const float ss = wave_sum(es);
          const float tmx = wave_max(tc);
          const float et = live ? __expf((tc - tmx) * inv_T) : 0.f;
          const float st = wave_sum(et);
          const float ps = es / ss, pt = et / st;
          const float log_ps = (x - mx) * inv_T - __logf(ss), log_pt = (tc - tmx) * inv_T - __logf(st);
          const float kl = wave_sum(live && pt > 0.f ? pt * (log_pt - log_ps) : 0.f);
          t_ddf += w_ddf * temp * temp * kl * ddf_coef;
          if (live) d_corners_ddf[off] = w_ddf * temp * ddf_coef * (ps - pt);
        }
      }
    }
  }
  if (lane == 0) {
    part[wave][0] = t_vfl; part[wave][1] = t_l1; part[wave][2] = t_giou; part[wave][3] = t_fgl; part[wave][4] = t_ddf;
  }
  __syncthreads();
  if (threadIdx.x < 5) {
    // 64 replicas of the five sums (the caller adds them up): ~4 000 workgroups adding into the SAME five floats -- one
    // cache line, one L2 atomic unit -- serialised for ~40 us of this 52 us kernel
    const int k = threadIdx.x;
    atomicAdd(terms + (blockIdx.x % CRIT_TERM_REPLICAS) * 8 + k, part[0][k] + part[1][k] + part[2][k] + part[3][k]);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void criterion_head_kernel(
    const T* __restrict__ logits, const float* __restrict__ boxes, const float* __restrict__ corners,
    const float* __restrict__ ref, const float* __restrict__ teacher_corners, const T* __restrict__ teacher_logits,
    const int32_t* __restrict__ own, const int32_t* __restrict__ go, const int64_t* __restrict__ tgt_labels,
    const float* __restrict__ tgt_boxes, const float* __restrict__ Wn, const float* __restrict__ scal,
    float* __restrict__ terms, T* __restrict__ d_logits, float* __restrict__ d_box_l1, float* __restrict__ d_box_giou,
    float* __restrict__ d_corners_fgl, float* __restrict__ d_corners_ddf, CritArgs a) {
  criterion_head_body<T>(logits, boxes, corners, ref, teacher_corners, teacher_logits, own, go, tgt_labels, tgt_boxes, Wn, scal, terms, d_logits,
                         d_box_l1, d_box_giou, d_corners_fgl, d_corners_ddf, a);
}

// Every head of a criterion call in ONE launch (blockIdx.y = head): a head's kernel is a few thousand one-row waves of long
// dependent chains (softmax / log / IoU per row) that leave most of the chip idle -- 25 us per head, eleven heads one after
// the other per step -- and the heads do not depend on each other. The table travels as a kernel argument (<= 12 heads).
constexpr int CRIT_MAX_HEADS = 12;
struct CritHead {
  const void* logits;
  const float* boxes;
  const float* corners;
  const float* ref;
  const float* teacher_corners;
  const void* teacher_logits;
  const int32_t* own;
  const int32_t* go;
  const int64_t* tgt_labels;
  const float* tgt_boxes;
  const float* Wn;
  const float* scal;
  float* terms;
  void* d_logits;
  float* d_box_l1;
  float* d_box_giou;
  float* d_corners_fgl;
  float* d_corners_ddf;
  float* out;  // f32 [5]: the replicas' sums
  CritArgs a;
};
struct CritHeads {
  int n;
  CritHead h[CRIT_MAX_HEADS];
};

template <typename T>
__global__ __launch_bounds__(256) void criterion_heads_kernel(const CritHeads tab) {
  const CritHead& e = tab.h[blockIdx.y];
  if ((int)blockIdx.x * 4 >= e.a.B * e.a.Q) return;  // a shorter head's surplus blocks (whole workgroup, before any barrier)
  criterion_head_body<T>((const T*)e.logits, e.boxes, e.corners, e.ref, e.teacher_corners, (const T*)e.teacher_logits, e.own, e.go, e.tgt_labels,
                         e.tgt_boxes, e.Wn, e.scal, e.terms, (T*)e.d_logits, e.d_box_l1, e.d_box_giou, e.d_corners_fgl, e.d_corners_ddf, e.a);
}

__global__ __launch_bounds__(64) void criterion_terms_reduce_kernel(const CritHeads tab) {
  const CritHead& e = tab.h[blockIdx.x];
  if (threadIdx.x < 5) {
    float s = 0.f;
    for (int r = 0; r < CRIT_TERM_REPLICAS; ++r) s += e.terms[r * 8 + threadIdx.x];
    e.out[threadIdx.x] = s;
  }
}

// Backward of a head's five weighted terms: the forward already holds d(term)/d(input); autograd hands five scalar
// gradients (the five terms' d(loss)/d(term)) and the chain rule is
//     d logits = d_logits * g_vfl     d boxes = d_l1 * g_l1 + d_giou * g_giou     d corners = d_fgl * g_fgl + d_ddf * g_ddf
// -- five or six elementwise launches over small tensors as PyTorch ops, one here.
template <typename T>
__global__ __launch_bounds__(256) void criterion_scale_kernel(const T* __restrict__ d_logits, const float* __restrict__ d_l1,
                                                              const float* __restrict__ d_giou, const float* __restrict__ d_fgl,
                                                              const float* __restrict__ d_ddf, const float* __restrict__ g_vfl,
                                                              const float* __restrict__ g_l1, const float* __restrict__ g_giou,
                                                              const float* __restrict__ g_fgl, const float* __restrict__ g_ddf,
                                                              T* __restrict__ o_logits, float* __restrict__ o_boxes,
                                                              float* __restrict__ o_corners, long long n_logits, long long n_boxes,
                                                              long long n_corners) {
  // (the PyTorch form casts the scalar to the logits' dtype first)
  const float sv = g_vfl ? (sizeof(T) == 2 ? bf16_to_f32(f32_to_bf16(g_vfl[0])) : g_vfl[0]) : 0.f;
  const float s1 = g_l1 ? g_l1[0] : 0.f, s2 = g_giou ? g_giou[0] : 0.f;
  const float s3 = g_fgl ? g_fgl[0] : 0.f, s4 = g_ddf ? g_ddf[0] : 0.f;
  const long long total = n_logits + n_boxes + n_corners;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    if (i < n_logits) {
      store1(o_logits + i, VecTraits<T>::load1(d_logits + i) * sv);
    } else if (i < n_logits + n_boxes) {
      const long long j = i - n_logits;
      o_boxes[j] = d_l1[j] * s1 + d_giou[j] * s2;
    } else {
      const long long j = i - n_logits - n_boxes;
      o_corners[j] = d_fgl[j] * s3 + (d_ddf ? d_ddf[j] * s4 : 0.f);
    }
  }
}

}  // namespace dfine

using namespace dfine;

extern "C" int dfine_criterion_head_scale(const void* d_logits, int logits_dtype, const float* d_boxes_l1, const float* d_boxes_giou,
                                          const float* d_corners_fgl, const float* d_corners_ddf, const float* g_vfl, const float* g_l1,
                                          const float* g_giou, const float* g_fgl, const float* g_ddf, void* out_logits, float* out_boxes,
                                          float* out_corners, long long n_logits, long long n_boxes, long long n_corners, void* stream) {
  if (!d_logits || !d_boxes_l1 || !d_boxes_giou || !out_logits || !out_boxes) return DFINE_E_BADARG;
  if (n_logits <= 0 || n_boxes <= 0 || n_corners < 0 || (n_corners > 0 && (!d_corners_fgl || !out_corners))) return DFINE_E_BADARG;
  const long long total = n_logits + n_boxes + n_corners;
  long long blocks = (total + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipStream_t st = (hipStream_t)stream;
  if (logits_dtype == DFINE_DT_F32)
    hipLaunchKernelGGL(criterion_scale_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, st, (const float*)d_logits, d_boxes_l1, d_boxes_giou,
                       d_corners_fgl, d_corners_ddf, g_vfl, g_l1, g_giou, g_fgl, g_ddf, (float*)out_logits, out_boxes, out_corners, n_logits, n_boxes,
                       n_corners);
  else if (logits_dtype == DFINE_DT_BF16)
    hipLaunchKernelGGL(criterion_scale_kernel<bf16_t>, dim3((unsigned)blocks), dim3(256), 0, st, (const bf16_t*)d_logits, d_boxes_l1, d_boxes_giou,
                       d_corners_fgl, d_corners_ddf, g_vfl, g_l1, g_giou, g_fgl, g_ddf, (bf16_t*)out_logits, out_boxes, out_corners, n_logits,
                       n_boxes, n_corners);
  else
    return DFINE_E_UNSUPPORTED;
  DFINE_LAUNCH_CHECK();
  return 0;
}

static int make_crit_args(CritArgs* out, const float* corners, const float* ref_points, const float* teacher_corners, const void* teacher_logits,
                          const float* Wn, const float* d_corners_fgl, const float* d_corners_ddf, int B, int Q, int C, int G, int n_bins,
                          float alpha, float gamma, float reg_scale, float temperature, const long long* batch_strides) {
  if (B <= 0 || Q <= 0 || C <= 0 || G <= 0) return DFINE_E_BADARG;
  const int has_local = corners != nullptr;
  const int has_ddf = has_local && teacher_corners != nullptr;
  if (has_local && (!ref_points || !Wn || !d_corners_fgl || n_bins <= 1 || n_bins > 64)) return DFINE_E_BADARG;
  if (has_ddf && (!teacher_logits || !d_corners_ddf)) return DFINE_E_BADARG;
  CritArgs a{B, Q, C, G, n_bins, has_local, has_ddf, alpha, gamma, reg_scale, temperature,
             (long long)Q * C, (long long)Q * 4, (long long)Q * 4 * n_bins, (long long)Q * 4, (long long)Q * 4 * n_bins, (long long)Q * C};
  if (batch_strides) {  // logits, boxes, corners, ref_points, teacher_corners, teacher_logits (0 = dense)
    long long* sb[6] = {&a.sb_logits, &a.sb_boxes, &a.sb_corners, &a.sb_ref, &a.sb_tcorners, &a.sb_tlogits};
    for (int i = 0; i < 6; ++i) {
      if (batch_strides[i] < 0 || (batch_strides[i] > 0 && batch_strides[i] < *sb[i])) return DFINE_E_BADARG;
      if (batch_strides[i] > 0) *sb[i] = batch_strides[i];
    }
    if (a.sb_boxes % 4 != 0 || (has_local && a.sb_ref % 4 != 0)) return DFINE_E_UNSUPPORTED;  // float4 loads
  }
  *out = a;
  return 0;
}

extern "C" int dfine_criterion_head(const void* logits, int logits_dtype, const float* boxes, const float* corners,
                                    const float* ref_points, const float* teacher_corners, const void* teacher_logits,
                                    const int32_t* own_assign, const int32_t* go_assign, const int64_t* tgt_labels,
                                    const float* tgt_boxes, const float* Wn, const float* scalars, float* terms,
                                    void* d_logits, float* d_boxes_l1, float* d_boxes_giou, float* d_corners_fgl,
                                    float* d_corners_ddf, int B, int Q, int C, int G, int n_bins, float alpha,
                                    float gamma, float reg_scale, float temperature, const long long* batch_strides, void* stream) {
  if (!logits || !boxes || !own_assign || !go_assign || !tgt_labels || !tgt_boxes || !scalars || !terms || !d_logits ||
      !d_boxes_l1 || !d_boxes_giou)
    return DFINE_E_BADARG;
  CritArgs a;
  if (int rc = make_crit_args(&a, corners, ref_points, teacher_corners, teacher_logits, Wn, d_corners_fgl, d_corners_ddf, B, Q, C, G, n_bins, alpha,
                              gamma, reg_scale, temperature, batch_strides))
    return rc;
  const int rows = B * Q;
  dim3 grid((rows + 3) / 4), block(256);
  hipStream_t st = (hipStream_t)stream;
  if (logits_dtype == DFINE_DT_BF16)
    hipLaunchKernelGGL(criterion_head_kernel<bf16_t>, grid, block, 0, st, (const bf16_t*)logits, boxes, corners, ref_points,
                       teacher_corners, (const bf16_t*)teacher_logits, own_assign, go_assign, tgt_labels, tgt_boxes, Wn,
                       scalars, terms, (bf16_t*)d_logits, d_boxes_l1, d_boxes_giou, d_corners_fgl, d_corners_ddf, a);
  else if (logits_dtype == DFINE_DT_F32)
    hipLaunchKernelGGL(criterion_head_kernel<float>, grid, block, 0, st, (const float*)logits, boxes, corners, ref_points,
                       teacher_corners, (const float*)teacher_logits, own_assign, go_assign, tgt_labels, tgt_boxes, Wn,
                       scalars, terms, (float*)d_logits, d_boxes_l1, d_boxes_giou, d_corners_fgl, d_corners_ddf, a);
  else
    return DFINE_E_UNSUPPORTED;
  DFINE_LAUNCH_CHECK();
  return 0;
}

extern "C" int dfine_criterion_heads(const dfine_criterion_head_args* heads, int n_heads, int logits_dtype, void* stream) {
  if (!heads || n_heads <= 0) return DFINE_E_BADARG;
  if (logits_dtype != DFINE_DT_BF16 && logits_dtype != DFINE_DT_F32) return DFINE_E_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  for (int first = 0; first < n_heads; first += CRIT_MAX_HEADS) {
    CritHeads tab;
    tab.n = n_heads - first < CRIT_MAX_HEADS ? n_heads - first : CRIT_MAX_HEADS;
    int max_rows = 0;
    for (int i = 0; i < tab.n; ++i) {
      const dfine_criterion_head_args& q = heads[first + i];
      if (!q.logits || !q.boxes || !q.own_assign || !q.go_assign || !q.tgt_labels || !q.tgt_boxes || !q.scalars || !q.terms || !q.d_logits ||
          !q.d_boxes_l1 || !q.d_boxes_giou || !q.out)
        return DFINE_E_BADARG;
      CritHead& e = tab.h[i];
      if (int rc = make_crit_args(&e.a, q.corners, q.ref_points, q.teacher_corners, q.teacher_logits, q.Wn, q.d_corners_fgl, q.d_corners_ddf, q.B, q.Q,
                                  q.C, q.G, q.n_bins, q.alpha, q.gamma, q.reg_scale, q.temperature, q.has_strides ? q.batch_strides : nullptr))
        return rc;
      e.logits = q.logits, e.boxes = q.boxes, e.corners = q.corners, e.ref = q.ref_points;
      e.teacher_corners = e.a.has_ddf ? q.teacher_corners : nullptr, e.teacher_logits = e.a.has_ddf ? q.teacher_logits : nullptr;
      e.own = q.own_assign, e.go = q.go_assign, e.tgt_labels = q.tgt_labels, e.tgt_boxes = q.tgt_boxes, e.Wn = q.Wn, e.scal = q.scalars;
      e.terms = q.terms, e.d_logits = q.d_logits, e.d_box_l1 = q.d_boxes_l1, e.d_box_giou = q.d_boxes_giou;
      e.d_corners_fgl = q.d_corners_fgl, e.d_corners_ddf = q.d_corners_ddf, e.out = q.out;
      if (q.B * q.Q > max_rows) max_rows = q.B * q.Q;
    }
    const dim3 grid((max_rows + 3) / 4, tab.n), block(256);
    if (logits_dtype == DFINE_DT_BF16) hipLaunchKernelGGL(criterion_heads_kernel<bf16_t>, grid, block, 0, st, tab);
    else hipLaunchKernelGGL(criterion_heads_kernel<float>, grid, block, 0, st, tab);
    hipLaunchKernelGGL(criterion_terms_reduce_kernel, dim3(tab.n), dim3(64), 0, st, tab);
  }
  DFINE_LAUNCH_CHECK();
  return 0;
}
