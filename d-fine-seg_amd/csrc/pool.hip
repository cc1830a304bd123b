// Stem pooling of HGNetv2: F.max_pool2d(F.pad(x, (0, 1, 0, 1)), kernel_size=2, stride=1, ceil_mode=True)
// (reference src/d_fine/arch/hgnetv2.py:213-224: the map is zero-extended by one row/column at the
// bottom/right, then max-pooled 2x2 stride 1, so the output keeps H x W) on a channels_last map, without
// materialising the padded copy. ATen's NHWC max-pool takes 460 us forward + 450 us backward on the
// [32, 24, 321, 321] bf16 stem map; these are plain streaming kernels (one 16-byte channel chunk per lane).
//
// Backward routes each output gradient to the FIRST maximum of its window in scan order (y, x), (y, x+1),
// (y+1, x), (y+1, x+1) -- ATen's rule -- formulated as a gather: an input pixel looks at the (up to) four
// windows that contain it and recomputes their arg-max from its 3x3 neighbourhood.
#include "common.h"

namespace dfine {

template <typename T>
__device__ __forceinline__ void load_px(const T* x, int b, int y, int xx, int H, int W, int C, int c0, float* f) {
  constexpr int V = VecTraits<T>::kElems;
  if (y < H && xx < W) {
    load_vec<T, V>(x + (((size_t)b * H + y) * W + xx) * C + c0, f);
  } else {
#pragma unroll
    for (int i = 0; i < V; ++i) f[i] = 0.f;  // the zero extension
  }
}

template <typename T>
__global__ __launch_bounds__(256) void pool_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, int B, int H, int W,
                                                       int C, int ldy) {
  constexpr int V = VecTraits<T>::kElems;
  const int cpp = C / V;  // chunks per pixel
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (size_t)B * H * W * cpp) return;
  const int c0 = (int)(t % cpp) * V;
  size_t p = t / cpp;
  const int xx = (int)(p % W);
  p /= W;
  const int yy = (int)(p % H), b = (int)(p / H);
  float a[V], o[V];
  load_px(x, b, yy, xx, H, W, C, c0, o);
  load_px(x, b, yy, xx + 1, H, W, C, c0, a);
#pragma unroll
  for (int i = 0; i < V; ++i) o[i] = a[i] > o[i] ? a[i] : o[i];
  load_px(x, b, yy + 1, xx, H, W, C, c0, a);
#pragma unroll
  for (int i = 0; i < V; ++i) o[i] = a[i] > o[i] ? a[i] : o[i];
  load_px(x, b, yy + 1, xx + 1, H, W, C, c0, a);
#pragma unroll
  for (int i = 0; i < V; ++i) o[i] = a[i] > o[i] ? a[i] : o[i];
  store_vec<T, V>(y + (((size_t)b * H + yy) * W + xx) * ldy + c0, o);  // ldy > C: a channel slice of a concat buffer
}

template <typename T>
__global__ __launch_bounds__(256) void pool_bwd_kernel(const T* __restrict__ x, const T* __restrict__ gy,
                                                       T* __restrict__ gx, int B, int H, int W, int C, int ld_gy) {
  constexpr int V = VecTraits<T>::kElems;
  const int cpp = C / V;
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (size_t)B * H * W * cpp) return;
  const int c0 = (int)(t % cpp) * V;
  size_t p = t / cpp;
  const int xx = (int)(p % W);
  p /= W;
  const int yy = (int)(p % H), b = (int)(p / H);
  float n[3][3][V];  // the neighbourhood, n[1][1] = this pixel; rows/columns before the map are never used
#pragma unroll
  for (int dy = 0; dy < 3; ++dy)
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      const int y2 = yy + dy - 1, x2 = xx + dx - 1;
      if (y2 >= 0 && x2 >= 0) load_px(x, b, y2, x2, H, W, C, c0, n[dy][dx]);
      else {
#pragma unroll
        for (int i = 0; i < V; ++i) n[dy][dx][i] = 0.f;
      }
    }
  float acc[V];
#pragma unroll
  for (int i = 0; i < V; ++i) acc[i] = 0.f;
  // window with top-left (yy + wy - 1, xx + wx - 1) holds this pixel at scan position (1 - wy, 1 - wx)
#pragma unroll
  for (int wy = 0; wy < 2; ++wy)
#pragma unroll
    for (int wx = 0; wx < 2; ++wx) {
      const int oy = yy + wy - 1, ox = xx + wx - 1;
      if (oy < 0 || ox < 0) continue;
      float g[V];
      load_vec<T, V>(gy + (((size_t)b * H + oy) * W + ox) * ld_gy + c0, g);
      const int mine = (1 - wy) * 2 + (1 - wx);  // scan index of this pixel inside that window
#pragma unroll
      for (int i = 0; i < V; ++i) {
        // first maximum in scan order
        int best = 0;
        float bv = n[wy][wx][i];
        if (n[wy][wx + 1][i] > bv) bv = n[wy][wx + 1][i], best = 1;
        if (n[wy + 1][wx][i] > bv) bv = n[wy + 1][wx][i], best = 2;
        if (n[wy + 1][wx + 1][i] > bv) bv = n[wy + 1][wx + 1][i], best = 3;
        acc[i] += best == mine ? g[i] : 0.f;
      }
    }
  store_vec<T, V>(gx + (((size_t)b * H + yy) * W + xx) * C + c0, acc);
}

// Nearest x2 upsampling of the FPN (reference hybrid_encoder.py:377 F.interpolate(scale_factor=2., mode="nearest")), forward with
// an output pitch -- the upsampled map is written straight into its half of the FPN concat buffer -- and backward (each input
// pixel sums its 2x2 output gradients, read with a pitch from the buffer's gradient).
template <typename T>
__global__ __launch_bounds__(256) void upsample2x_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, int B, int H, int W, int C,
                                                            int ldy) {
  constexpr int V = VecTraits<T>::kElems;
  const int cpp = C / V;
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (size_t)B * 2 * H * 2 * W * cpp) return;
  const int c0 = (int)(t % cpp) * V;
  size_t p = t / cpp;
  const int ox = (int)(p % (2 * W));
  p /= 2 * W;
  const int oy = (int)(p % (2 * H)), b = (int)(p / (2 * H));
  const Vec16 v = *reinterpret_cast<const Vec16*>(x + (((size_t)b * H + (oy >> 1)) * W + (ox >> 1)) * C + c0);
  *reinterpret_cast<Vec16*>(y + (((size_t)b * 2 * H + oy) * 2 * W + ox) * ldy + c0) = v;
}

template <typename T>
__global__ __launch_bounds__(256) void upsample2x_bwd_kernel(const T* __restrict__ gy, T* __restrict__ gx, int B, int H, int W, int C,
                                                            int ld_gy) {
  constexpr int V = VecTraits<T>::kElems;
  const int cpp = C / V;
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (size_t)B * H * W * cpp) return;
  const int c0 = (int)(t % cpp) * V;
  size_t p = t / cpp;
  const int xx = (int)(p % W);
  p /= W;
  const int yy = (int)(p % H), b = (int)(p / H);
  float acc[V], g[V];
#pragma unroll
  for (int i = 0; i < V; ++i) acc[i] = 0.f;
#pragma unroll
  for (int dy = 0; dy < 2; ++dy)
#pragma unroll
    for (int dx = 0; dx < 2; ++dx) {
      load_vec<T, V>(gy + (((size_t)b * 2 * H + 2 * yy + dy) * 2 * W + 2 * xx + dx) * ld_gy + c0, g);
#pragma unroll
      for (int i = 0; i < V; ++i) acc[i] += g[i];
    }
  store_vec<T, V>(gx + (((size_t)b * H + yy) * W + xx) * C + c0, acc);
}

template <typename T>
static int launch_upsample(bool fwd, const void* in, void* out, int B, int H, int W, int C, int ld, hipStream_t st) {
  constexpr int V = VecTraits<T>::kElems;
  if (C % V != 0) return DFINE_E_UNSUPPORTED;
  if (ld == 0) ld = C;
  if (ld < C) return DFINE_E_BADARG;
  if (ld % V != 0 || (uintptr_t)(fwd ? out : in) % (V * sizeof(T)) != 0) return DFINE_E_UNSUPPORTED;
  const size_t items = (size_t)B * H * W * (C / V) * (fwd ? 4 : 1);
  const size_t blocks = (items + 255) / 256;
  if (blocks > 0x7fffffffULL) return DFINE_E_UNSUPPORTED;
  if (fwd) hipLaunchKernelGGL(upsample2x_fwd_kernel<T>, dim3((unsigned)blocks), dim3(256), 0, st, (const T*)in, (T*)out, B, H, W, C, ld);
  else hipLaunchKernelGGL(upsample2x_bwd_kernel<T>, dim3((unsigned)blocks), dim3(256), 0, st, (const T*)in, (T*)out, B, H, W, C, ld);
  DFINE_LAUNCH_CHECK();
  return 0;
}

template <typename T>
static int launch_pool(bool fwd, const void* x, const void* gy, void* out, int B, int H, int W, int C, int ld, hipStream_t st) {
  constexpr int V = VecTraits<T>::kElems;
  if (C % V != 0) return DFINE_E_UNSUPPORTED;
  if (ld == 0) ld = C;
  // the pitched operand (forward: y, backward: gy) must keep its channel vectors aligned
  if (ld < C || ld % V != 0 || (uintptr_t)(fwd ? out : gy) % (V * sizeof(T)) != 0) return ld < C ? DFINE_E_BADARG : DFINE_E_UNSUPPORTED;
  const size_t items = (size_t)B * H * W * (C / V);
  const size_t blocks = (items + 255) / 256;
  if (blocks > 0x7fffffffULL) return DFINE_E_UNSUPPORTED;
  if (fwd) hipLaunchKernelGGL(pool_fwd_kernel<T>, dim3((unsigned)blocks), dim3(256), 0, st, (const T*)x, (T*)out, B, H, W, C, ld);
  else hipLaunchKernelGGL(pool_bwd_kernel<T>, dim3((unsigned)blocks), dim3(256), 0, st, (const T*)x, (const T*)gy, (T*)out, B, H, W, C, ld);
  DFINE_LAUNCH_CHECK();
  return 0;
}

}  // namespace dfine

using namespace dfine;

extern "C" int dfine_stem_pool_fwd(const void* x, int dtype, void* y, int B, int H, int W, int C, int ldy, void* stream) {
  if (!x || !y) return DFINE_E_BADARG;
  if (B <= 0 || H <= 0 || W <= 0 || C <= 0) return DFINE_E_BADARG;
  if (dtype == DFINE_DT_F32) return launch_pool<float>(true, x, nullptr, y, B, H, W, C, ldy, (hipStream_t)stream);
  if (dtype == DFINE_DT_BF16) return launch_pool<bf16_t>(true, x, nullptr, y, B, H, W, C, ldy, (hipStream_t)stream);
  return DFINE_E_UNSUPPORTED;
}

extern "C" int dfine_stem_pool_bwd(const void* x, const void* gy, int ld_gy, int dtype, void* gx, int B, int H, int W, int C,
                                   void* stream) {
  if (!x || !gy || !gx) return DFINE_E_BADARG;
  if (B <= 0 || H <= 0 || W <= 0 || C <= 0) return DFINE_E_BADARG;
  if (dtype == DFINE_DT_F32) return launch_pool<float>(false, x, gy, gx, B, H, W, C, ld_gy, (hipStream_t)stream);
  if (dtype == DFINE_DT_BF16) return launch_pool<bf16_t>(false, x, gy, gx, B, H, W, C, ld_gy, (hipStream_t)stream);
  return DFINE_E_UNSUPPORTED;
}

extern "C" int dfine_upsample2x_fwd(const void* x, int dtype, void* y, int B, int H, int W, int C, int ldy, void* stream) {
  if (!x || !y) return DFINE_E_BADARG;
  if (B <= 0 || H <= 0 || W <= 0 || C <= 0) return DFINE_E_BADARG;
  if (dtype == DFINE_DT_F32) return launch_upsample<float>(true, x, y, B, H, W, C, ldy, (hipStream_t)stream);
  if (dtype == DFINE_DT_BF16) return launch_upsample<bf16_t>(true, x, y, B, H, W, C, ldy, (hipStream_t)stream);
  return DFINE_E_UNSUPPORTED;
}

extern "C" int dfine_upsample2x_bwd(const void* gy, int ld_gy, int dtype, void* gx, int B, int H, int W, int C, void* stream) {
  if (!gy || !gx) return DFINE_E_BADARG;
  if (B <= 0 || H <= 0 || W <= 0 || C <= 0) return DFINE_E_BADARG;
  if (dtype == DFINE_DT_F32) return launch_upsample<float>(false, gy, gx, B, H, W, C, ld_gy, (hipStream_t)stream);
  if (dtype == DFINE_DT_BF16) return launch_upsample<bf16_t>(false, gy, gx, B, H, W, C, ld_gy, (hipStream_t)stream);
  return DFINE_E_UNSUPPORTED;
}
