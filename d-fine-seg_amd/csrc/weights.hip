// Two small kernels around the parameters:
//   * dfine_transpose_weights: bf16 [K][T][C] -> [C][T][K] for every weight of the model in one launch --
//     the "transposed shadow" the input-gradient GEMMs read (dx = dy W needs W with the output channel as
//     the contiguous reduction axis; the forward wants the input channel contiguous). Refreshed once per
//     optimizer step over the whole flat bf16 shadow: ~2 x 40 MB of traffic for D-FINE-m.
//   * dfine_embedding_wgrad: dW[idx[m]] += g[m] (nn.Embedding backward for the denoising class embedding,
//     reference dfine_decoder.py:434 / arch/utils.py:434) as float atomics into the flat fp32 gradient.
#include "common.h"

namespace dfine {

__global__ __launch_bounds__(256) void transpose_weights_kernel(const bf16_t* __restrict__ src, bf16_t* __restrict__ dst,
                                                               const long long* __restrict__ table,
                                                               const int* __restrict__ blocks) {
  __shared__ bf16_t tile[32][33];
  const int w = blocks[2 * blockIdx.x], t_idx = blocks[2 * blockIdx.x + 1];
  const long long* e = table + 5 * (long long)w;
  const long long so = e[0], dof = e[1];
  const int K = (int)e[2], T = (int)e[3], C = (int)e[4];
  const int kt = (K + 31) / 32, ct = (C + 31) / 32;
  const int t = t_idx / (kt * ct), rem = t_idx % (kt * ct);
  const int k0 = (rem / ct) * 32, c0 = (rem % ct) * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int k = k0 + ty + 8 * i, c = c0 + tx;
    if (k < K && c < C) tile[ty + 8 * i][tx] = src[so + ((long long)k * T + t) * C + c];
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = c0 + ty + 8 * i, k = k0 + tx;
    if (k < K && c < C) dst[dof + ((long long)c * T + t) * K + k] = tile[tx][ty + 8 * i];
  }
}

template <typename T>
__global__ __launch_bounds__(256) void embedding_wgrad_kernel(const long long* __restrict__ idx, const T* __restrict__ g,
                                                             float* __restrict__ dW, long long M, int D, int rows,
                                                             int padding_idx) {
  // one wave per row m (grid-stride): each wave instruction adds 64 consecutive floats of one dW row
  const int lane = threadIdx.x & 63;
  const long long wave = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const long long n_waves = (long long)gridDim.x * (blockDim.x >> 6);
  for (long long m = wave; m < M; m += n_waves) {
    const long long r = idx[m];
    if (r < 0 || r >= rows || r == padding_idx) continue;
    for (int c = lane; c < D; c += 64) atomicAdd(dW + r * D + c, VecTraits<T>::load1(g + m * D + c));
  }
}

// GT masks -> loss targets: F.interpolate(mask.float(), size=(Hm, Wm), mode="nearest").clamp(0, 1) for every (image, slot)
// of a packed uint8 batch (reference dfine_criterion.py:239-270). PyTorch's nearest source index: min(floor(dst * scale), in-1)
// with scale = float(in) / float(out).
// (Hv, Wv) != (H, W): the batch went through the multiscale augmentation first (reference src/dl/dataset.py:645-670: masks
// resized to the new batch size with mode="nearest", thresholded back to uint8), i.e. TWO nearest resizes compose.
__global__ __launch_bounds__(256) void mask_targets_nearest_kernel(const unsigned char* __restrict__ masks, float* __restrict__ out,
                                                                  long long n_maps, int H, int W, int Hv, int Wv, int Hm, int Wm) {
  const float sy = (float)Hv / (float)Hm, sx = (float)Wv / (float)Wm;
  const float ty = (float)H / (float)Hv, tx = (float)W / (float)Wv;
  const long long total = n_maps * Hm * Wm;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int x = (int)(i % Wm);
    const long long t = i / Wm;
    const int y = (int)(t % Hm);
    const long long m = t / Hm;
    int iy = min((int)floorf((float)y * sy), Hv - 1), ix = min((int)floorf((float)x * sx), Wv - 1);
    if (Hv != H || Wv != W) iy = min((int)floorf((float)iy * ty), H - 1), ix = min((int)floorf((float)ix * tx), W - 1);
    out[i] = masks[(m * H + iy) * W + ix] ? 1.f : 0.f;
  }
}

// Image staging: uint8 HWC pixels (what the data loader decodes) -> normalised float NHWC on the device, optionally resized
// (bilinear, align_corners = False: the multiscale augmentation of the collate function). One thread = one output pixel.
//   v = (u8 - mean_c * 255) * (1 / (std_c * 255))       albumentations' A.Normalize in float32 (src/dl/dataset.py:128-131)
//   F.interpolate(v, size, mode="bilinear", align_corners=False): src = scale * (dst + 0.5) - 0.5 clamped at 0,
//   scale = float(in) / float(out), value = hl0 * (wl0 * p00 + wl1 * p01) + hl1 * (wl0 * p10 + wl1 * p11)
struct StageNorm {
  float sub[3], mul[3];
};
template <typename T>
__global__ __launch_bounds__(256) void stage_images_kernel(const unsigned char* __restrict__ src, T* __restrict__ dst, int B, int H, int W,
                                                          int OH, int OW, StageNorm nm) {
  const long long total = (long long)B * OH * OW;
  const bool resize = OH != H || OW != W;
  const float sy = (float)H / (float)OH, sx = (float)W / (float)OW;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int x = (int)(i % OW);
    const long long t = i / OW;
    const int y = (int)(t % OH);
    const long long b = t / OH;
    float v[3];
    if (!resize) {
      const unsigned char* p = src + ((b * H + y) * W + x) * 3;
#pragma unroll
      for (int c = 0; c < 3; ++c) v[c] = ((float)p[c] - nm.sub[c]) * nm.mul[c];
    } else {
      float fy = sy * ((float)y + 0.5f) - 0.5f, fx = sx * ((float)x + 0.5f) - 0.5f;
      fy = fy < 0.f ? 0.f : fy, fx = fx < 0.f ? 0.f : fx;
      const int y0 = min((int)fy, H - 1), x0 = min((int)fx, W - 1);
      const int y1 = y0 + (y0 < H - 1 ? 1 : 0), x1 = x0 + (x0 < W - 1 ? 1 : 0);
      const float hl1 = fy - (float)y0, wl1 = fx - (float)x0, hl0 = 1.f - hl1, wl0 = 1.f - wl1;
      const unsigned char* p00 = src + ((b * H + y0) * W + x0) * 3;
      const unsigned char* p01 = src + ((b * H + y0) * W + x1) * 3;
      const unsigned char* p10 = src + ((b * H + y1) * W + x0) * 3;
      const unsigned char* p11 = src + ((b * H + y1) * W + x1) * 3;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float a = ((float)p00[c] - nm.sub[c]) * nm.mul[c], bq = ((float)p01[c] - nm.sub[c]) * nm.mul[c];
        const float cq = ((float)p10[c] - nm.sub[c]) * nm.mul[c], d = ((float)p11[c] - nm.sub[c]) * nm.mul[c];
        v[c] = hl0 * (wl0 * a + wl1 * bq) + hl1 * (wl0 * cq + wl1 * d);
      }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) store1(dst + i * 3 + c, v[c]);
  }
}

// Row maximum of a [M, C] matrix (query selection: the best class logit of each of the 8 400 encoder tokens, reference
// dfine_decoder.py:905 enc_outputs_logits.max(-1).values feeding topk): 16 lanes per row, shuffle reduction. ATen's generic
// inner-dimension reduction takes 166 us for the 43 MB of D-FINE-m at batch 32.
template <typename T>
__global__ __launch_bounds__(256) void row_max_kernel(const T* __restrict__ x, float* __restrict__ out, long long M, int C) {
  const int sub = threadIdx.x & 15;
  const long long row = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
  float m = -INFINITY;
  if (row < M)
    for (int c = sub; c < C; c += 16) m = fmaxf(m, VecTraits<T>::load1(x + row * C + c));
#pragma unroll
  for (int o = 8; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if (row < M && sub == 0) out[row] = m;
}

}  // namespace dfine

using namespace dfine;

extern "C" int dfine_row_max(const void* x, int dtype, float* out, long long M, int C, void* stream) {
  if (!x || !out) return DFINE_E_BADARG;
  if (M <= 0 || C <= 0) return DFINE_E_BADARG;
  const long long blocks = (M * 16 + 255) / 256;
  if (blocks > 0x7fffffffLL) return DFINE_E_UNSUPPORTED;
  if (dtype == DFINE_DT_F32)
    hipLaunchKernelGGL(row_max_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const float*)x, out, M, C);
  else if (dtype == DFINE_DT_BF16)
    hipLaunchKernelGGL(row_max_kernel<bf16_t>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, out, M, C);
  else
    return DFINE_E_UNSUPPORTED;
  DFINE_LAUNCH_CHECK();
  return 0;
}

extern "C" int dfine_mask_targets_nearest(const unsigned char* masks, float* out, long long n_maps, int H, int W, int Hv, int Wv,
                                          int Hm, int Wm, void* stream) {
  if (!masks || !out) return DFINE_E_BADARG;
  if (Hv == 0 && Wv == 0) Hv = H, Wv = W;
  if (n_maps <= 0 || H <= 0 || W <= 0 || Hv <= 0 || Wv <= 0 || Hm <= 0 || Wm <= 0) return DFINE_E_BADARG;
  long long blocks = (n_maps * Hm * Wm + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(mask_targets_nearest_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, masks, out, n_maps, H, W, Hv,
                     Wv, Hm, Wm);
  DFINE_LAUNCH_CHECK();
  return 0;
}

extern "C" int dfine_stage_images(const unsigned char* src, void* dst, int dtype, int B, int H, int W, int OH, int OW,
                                  const float* mean, const float* std, void* stream) {
  if (!src || !dst || !mean || !std) return DFINE_E_BADARG;
  if (B <= 0 || H <= 0 || W <= 0 || OH <= 0 || OW <= 0) return DFINE_E_BADARG;
  StageNorm nm;
  for (int c = 0; c < 3; ++c) {
    if (!(std[c] > 0.f)) return DFINE_E_BADARG;
    nm.sub[c] = mean[c] * 255.f;          // albumentations: mean * max_pixel_value, reciprocal(std * max_pixel_value), float32
    nm.mul[c] = 1.f / (std[c] * 255.f);
  }
  long long blocks = ((long long)B * OH * OW + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  if (dtype == DFINE_DT_F32)
    hipLaunchKernelGGL(stage_images_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, src, (float*)dst, B, H, W, OH,
                       OW, nm);
  else if (dtype == DFINE_DT_BF16)
    hipLaunchKernelGGL(stage_images_kernel<bf16_t>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, src, (bf16_t*)dst, B, H, W,
                       OH, OW, nm);
  else
    return DFINE_E_UNSUPPORTED;
  DFINE_LAUNCH_CHECK();
  return 0;
}

extern "C" int dfine_transpose_weights(const void* src, void* dst, const long long* table, const int* blocks,
                                       int n_blocks, void* stream) {
  if (!src || !dst || !table || !blocks) return DFINE_E_BADARG;
  if (n_blocks <= 0) return DFINE_E_BADARG;
  hipLaunchKernelGGL(transpose_weights_kernel, dim3(n_blocks), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)src,
                     (bf16_t*)dst, table, blocks);
  DFINE_LAUNCH_CHECK();
  return 0;
}

extern "C" int dfine_embedding_wgrad(const long long* idx, const void* g, int dtype, float* dW, long long M, int D,
                                     int rows, int padding_idx, void* stream) {
  if (!idx || !g || !dW) return DFINE_E_BADARG;
  if (M <= 0 || D <= 0 || rows <= 0) return DFINE_E_BADARG;
  long long blocks = (M + 3) / 4;
  if (blocks > 2048) blocks = 2048;
  if (dtype == DFINE_DT_F32)
    hipLaunchKernelGGL(embedding_wgrad_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, idx,
                       (const float*)g, dW, M, D, rows, padding_idx);
  else if (dtype == DFINE_DT_BF16)
    hipLaunchKernelGGL(embedding_wgrad_kernel<bf16_t>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, idx,
                       (const bf16_t*)g, dW, M, D, rows, padding_idx);
  else
    return DFINE_E_UNSUPPORTED;
  DFINE_LAUNCH_CHECK();
  return 0;
}
