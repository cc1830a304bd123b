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
  const size_t t = (size_t)xcd_block_id() * blockDim.x + threadIdx.x;
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
  const size_t t = (size_t)xcd_block_id() * blockDim.x + threadIdx.x;
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
  const size_t t = (size_t)xcd_block_id() * blockDim.x + threadIdx.x;
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
  const size_t t = (size_t)xcd_block_id() * blockDim.x + threadIdx.x;
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
  if (blocks > 0x7ffffff0ULL) return DFINE_E_UNSUPPORTED;
  if (fwd) hipLaunchKernelGGL(upsample2x_fwd_kernel<T>, dim3(xcd_grid((unsigned)blocks)), dim3(256), 0, st, (const T*)in, (T*)out, B, H, W, C, ld);
  else hipLaunchKernelGGL(upsample2x_bwd_kernel<T>, dim3(xcd_grid((unsigned)blocks)), dim3(256), 0, st, (const T*)in, (T*)out, B, H, W, C, ld);
  DFINE_LAUNCH_CHECK();
  return 0;
}

// Bilinear resize (align_corners = false, PyTorch's source-index rule) of a channel-last map, ADDED to a base map of the output
// size: out = base + resize(x) -- the top-down sum of MaskPixelDecoder (dfine_decoder.py:352-360: `x = x + F.interpolate(lat,
// size, mode="bilinear")`) in one launch; backward = each INPUT pixel gathers its weights over the output pixels that read it
// (no atomics; the stock backward scatters with float atomics in fp32). fp32 arithmetic, one rounding.
struct BilinearTap {
  int i0, i1;
  float l0, l1;
};
__device__ __forceinline__ BilinearTap bilinear_tap(int dst, float scale, int n_in) {
  float src = scale * ((float)dst + 0.5f) - 0.5f;
  if (src < 0.f) src = 0.f;
  BilinearTap t;
  t.i0 = (int)src;
  if (t.i0 > n_in - 1) t.i0 = n_in - 1;
  t.i1 = t.i0 + (t.i0 < n_in - 1 ? 1 : 0);
  t.l1 = src - (float)t.i0;
  t.l0 = 1.f - t.l1;
  return t;
}

template <typename T>
__global__ __launch_bounds__(256) void bilinear_add_fwd_kernel(const T* __restrict__ x, const T* __restrict__ base, T* __restrict__ y, int B,
                                                              int h, int w, int H, int W, int C, float sy, float sx) {
  constexpr int V = VecTraits<T>::kElems;
  const int cpp = C / V;
  const size_t t = (size_t)xcd_block_id() * blockDim.x + threadIdx.x;
  if (t >= (size_t)B * H * W * cpp) return;
  const int c0 = (int)(t % cpp) * V;
  size_t p = t / cpp;
  const int X = (int)(p % W);
  p /= W;
  const int Y = (int)(p % H), b = (int)(p / H);
  const BilinearTap ty = bilinear_tap(Y, sy, h), tx = bilinear_tap(X, sx, w);
  const T* xb = x + (size_t)b * h * w * C + c0;
  float v00[V], v01[V], v10[V], v11[V], o[V];
  load_vec<T, V>(xb + ((size_t)ty.i0 * w + tx.i0) * C, v00);
  load_vec<T, V>(xb + ((size_t)ty.i0 * w + tx.i1) * C, v01);
  load_vec<T, V>(xb + ((size_t)ty.i1 * w + tx.i0) * C, v10);
  load_vec<T, V>(xb + ((size_t)ty.i1 * w + tx.i1) * C, v11);
  const size_t off = (((size_t)b * H + Y) * W + X) * C + c0;
  if (base) load_vec<T, V>(base + off, o);
#pragma unroll
  for (int i = 0; i < V; ++i) {
    const float r = ty.l0 * (tx.l0 * v00[i] + tx.l1 * v01[i]) + ty.l1 * (tx.l0 * v10[i] + tx.l1 * v11[i]);
    o[i] = base ? o[i] + r : r;
  }
  store_vec<T, V>(y + off, o);
}

constexpr int BILINEAR_MAX_CAND = 20;  // output pixels along one axis that can read one input pixel: <= 2 * (out / in) + 2

template <typename T>
__global__ __launch_bounds__(256) void bilinear_bwd_kernel(const T* __restrict__ gy, T* __restrict__ gx, int B, int h, int w, int H, int W,
                                                          int C, float sy, float sx) {
  constexpr int V = VecTraits<T>::kElems;
  const int cpp = C / V;
  const size_t t = (size_t)xcd_block_id() * blockDim.x + threadIdx.x;
  if (t >= (size_t)B * h * w * cpp) return;
  const int c0 = (int)(t % cpp) * V;
  size_t p = t / cpp;
  const int xx = (int)(p % w);
  p /= w;
  const int yy = (int)(p % h), b = (int)(p / h);
  // candidate output rows / columns and this input pixel's weight in each (0 for a candidate that does not read it)
  auto span = [](int i, float scale, int n_out, int& lo, int& hi) {
    const float inv = 1.f / scale;
    lo = (int)floorf(((float)i - 0.5f) * inv - 0.5f) - 1;
    hi = (int)ceilf(((float)i + 1.5f) * inv - 0.5f) + 1;
    if (lo < 0 || i == 0) lo = 0;
    if (hi > n_out - 1) hi = n_out - 1;
  };
  int ylo, yhi, xlo, xhi;
  span(yy, sy, H, ylo, yhi);
  span(xx, sx, W, xlo, xhi);
  float wx[BILINEAR_MAX_CAND];
  const int nx = min(xhi - xlo + 1, BILINEAR_MAX_CAND);
  for (int j = 0; j < nx; ++j) {
    const BilinearTap tx = bilinear_tap(xlo + j, sx, w);
    wx[j] = (tx.i0 == xx ? tx.l0 : 0.f) + (tx.i1 == xx ? tx.l1 : 0.f);
  }
  float acc[V], g[V];
#pragma unroll
  for (int i = 0; i < V; ++i) acc[i] = 0.f;
  for (int Y = ylo; Y <= yhi; ++Y) {
    const BilinearTap ty = bilinear_tap(Y, sy, h);
    const float wy = (ty.i0 == yy ? ty.l0 : 0.f) + (ty.i1 == yy ? ty.l1 : 0.f);
    if (wy == 0.f) continue;
    const T* row = gy + (((size_t)b * H + Y) * W) * C + c0;
    for (int j = 0; j < nx; ++j) {
      if (wx[j] == 0.f) continue;
      load_vec<T, V>(row + (size_t)(xlo + j) * C, g);
      const float wgt = wy * wx[j];
#pragma unroll
      for (int i = 0; i < V; ++i) acc[i] += wgt * g[i];
    }
  }
  store_vec<T, V>(gx + (((size_t)b * h + yy) * w + xx) * C + c0, acc);
}

template <typename T>
static int launch_bilinear(bool fwd, const void* in, const void* base, void* out, int B, int h, int w, int H, int W, int C, hipStream_t st) {
  constexpr int V = VecTraits<T>::kElems;
  if (C % V != 0) return DFINE_E_UNSUPPORTED;
  if (H < h || W < w || 2 * ((H + h - 1) / h) + 4 > BILINEAR_MAX_CAND || 2 * ((W + w - 1) / w) + 4 > BILINEAR_MAX_CAND) return DFINE_E_UNSUPPORTED;  // up-sampling by <= 8
  const float sy = (float)h / (float)H, sx = (float)w / (float)W;
  const size_t items = (size_t)B * (fwd ? H : h) * (fwd ? W : w) * (C / V);
  const size_t blocks = (items + 255) / 256;
  if (blocks > 0x7ffffff0ULL) return DFINE_E_UNSUPPORTED;
  if (fwd) hipLaunchKernelGGL(bilinear_add_fwd_kernel<T>, dim3(xcd_grid((unsigned)blocks)), dim3(256), 0, st, (const T*)in, (const T*)base, (T*)out, B, h, w, H, W, C, sy, sx);
  else hipLaunchKernelGGL(bilinear_bwd_kernel<T>, dim3(xcd_grid((unsigned)blocks)), dim3(256), 0, st, (const T*)in, (T*)out, B, h, w, H, W, C, sy, sx);
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
  if (blocks > 0x7ffffff0ULL) return DFINE_E_UNSUPPORTED;
  if (fwd) hipLaunchKernelGGL(pool_fwd_kernel<T>, dim3(xcd_grid((unsigned)blocks)), dim3(256), 0, st, (const T*)x, (T*)out, B, H, W, C, ld);
  else hipLaunchKernelGGL(pool_bwd_kernel<T>, dim3(xcd_grid((unsigned)blocks)), dim3(256), 0, st, (const T*)x, (const T*)gy, (T*)out, B, H, W, C, ld);
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

extern "C" int dfine_bilinear_add_fwd(const void* x, const void* base, int dtype, void* y, int B, int h, int w, int H, int W, int C,
                                      void* stream) {
  if (!x || !y) return DFINE_E_BADARG;
  if (B <= 0 || h <= 0 || w <= 0 || H <= 0 || W <= 0 || C <= 0) return DFINE_E_BADARG;
  if (dtype == DFINE_DT_F32) return launch_bilinear<float>(true, x, base, y, B, h, w, H, W, C, (hipStream_t)stream);
  if (dtype == DFINE_DT_BF16) return launch_bilinear<bf16_t>(true, x, base, y, B, h, w, H, W, C, (hipStream_t)stream);
  return DFINE_E_UNSUPPORTED;
}

extern "C" int dfine_bilinear_bwd(const void* gy, int dtype, void* gx, int B, int h, int w, int H, int W, int C, void* stream) {
  if (!gy || !gx) return DFINE_E_BADARG;
  if (B <= 0 || h <= 0 || w <= 0 || H <= 0 || W <= 0 || C <= 0) return DFINE_E_BADARG;
  if (dtype == DFINE_DT_F32) return launch_bilinear<float>(false, gy, nullptr, gx, B, h, w, H, W, C, (hipStream_t)stream);
  if (dtype == DFINE_DT_BF16) return launch_bilinear<bf16_t>(false, gy, nullptr, gx, B, h, w, H, W, C, (hipStream_t)stream);
  return DFINE_E_UNSUPPORTED;
}
