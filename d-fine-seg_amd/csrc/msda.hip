// Multi-scale deformable attention core for gfx950: the bilinear gather behind the D-FINE
// decoder's cross-attention. Semantics follow deformable_attention_core_func_v2
// (reference src/d_fine/arch/utils.py:191-264: grid = 2*loc-1, F.grid_sample bilinear /
// zeros / align_corners=False  =>  pixel = loc*size - 0.5, out-of-image corners read 0),
// re-laid-out channel-last so one corner of one head is one contiguous segment.
//
// Both kernels are HBM/L2-gather bound (no contraction worth an MFMA): the forward moves
// Lq*nH*P*4 corner segments of c elements per image, the backward the same reads plus the
// same number of float-atomic segments into grad_value.
#include <stdlib.h>

#include "common.h"

namespace dfine {

struct MsdaGeom {
  int n_levels;
  int P;                        // points per head, all levels
  int H[DFINE_MAX_LEVELS];
  int W[DFINE_MAX_LEVELS];
  int start[DFINE_MAX_LEVELS];  // first row of level l inside S
  int pt_begin[DFINE_MAX_LEVELS + 1];  // level l owns points [pt_begin[l], pt_begin[l+1])
};

// One sampling point -> 4 clamped row indices + 4 bilinear weights (0 for corners outside
// the map, so loads can stay unconditional and in-bounds).
struct Corners {
  int r00, r01, r10, r11;  // row = start + y*W + x
  float w00, w01, w10, w11;
  float fx, fy;
  bool vx0, vx1, vy0, vy1;
};

__device__ __forceinline__ Corners make_corners(float lx, float ly, int H, int W, int start) {
  Corners k;
  const float x = lx * (float)W - 0.5f;
  const float y = ly * (float)H - 0.5f;
  // Anything not strictly inside (-1, size) touches no pixel (also catches NaN/Inf).
  const bool live = (x > -1.0f) && (x < (float)W) && (y > -1.0f) && (y < (float)H);
  const float xs = live ? x : 0.0f, ys = live ? y : 0.0f;
  const float xf = floorf(xs), yf = floorf(ys);
  const int x0 = (int)xf, y0 = (int)yf;
  k.fx = xs - xf;
  k.fy = ys - yf;
  k.vx0 = live && x0 >= 0;
  k.vx1 = live && (x0 + 1) < W;
  k.vy0 = live && y0 >= 0;
  k.vy1 = live && (y0 + 1) < H;
  const int xc0 = max(x0, 0), xc1 = min(x0 + 1, W - 1);
  const int yc0 = max(y0, 0), yc1 = min(y0 + 1, H - 1);
  k.r00 = start + yc0 * W + xc0;
  k.r01 = start + yc0 * W + xc1;
  k.r10 = start + yc1 * W + xc0;
  k.r11 = start + yc1 * W + xc1;
  k.w00 = (k.vx0 && k.vy0) ? (1.0f - k.fx) * (1.0f - k.fy) : 0.0f;
  k.w01 = (k.vx1 && k.vy0) ? k.fx * (1.0f - k.fy) : 0.0f;
  k.w10 = (k.vx0 && k.vy1) ? (1.0f - k.fx) * k.fy : 0.0f;
  k.w11 = (k.vx1 && k.vy1) ? k.fx * k.fy : 0.0f;
  return k;
}

// ------------------------------------------------------------------------------------------
// Forward. A "pair" is one (batch, query, head). C/VEC lanes share a pair, each owning a
// 16-byte slice of the head's channels, so every corner is read as one 16 B/lane access and
// a pair's corner is one contiguous C*sizeof(T) segment (64 B for bf16, c=32).
// ------------------------------------------------------------------------------------------
template <typename T, int C>
__global__ __launch_bounds__(256) void msda_fwd_kernel(const T* __restrict__ value,
                                                        const float* __restrict__ loc,
                                                        const float* __restrict__ attn,
                                                        T* __restrict__ out, MsdaGeom g,
                                                        int n_pairs, int S, int Lq, int nH) {
  constexpr int VEC = VecTraits<T>::kElems;
  constexpr int LPP = C / VEC;  // lanes per pair
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  const int pair = tid / LPP;
  if (pair >= n_pairs) return;
  const int sub = tid % LPP;
  const int h = pair % nH;
  const int b = pair / (nH * Lq);
  const int row_stride = nH * C;  // elements between consecutive rows of S
  const T* vbase = value + ((size_t)b * S * nH + h) * C + sub * VEC;
  const float* lp = loc + (size_t)pair * g.P * 2;
  const float* ap = attn + (size_t)pair * g.P;

  float acc[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) acc[i] = 0.0f;

  for (int l = 0; l < g.n_levels; ++l) {
    const int H = g.H[l], W = g.W[l], start = g.start[l];
#pragma unroll 3
    for (int p = g.pt_begin[l]; p < g.pt_begin[l + 1]; ++p) {
      const float2 xy = *reinterpret_cast<const float2*>(lp + 2 * p);
      const float a = ap[p];
      const Corners k = make_corners(xy.x, xy.y, H, W, start);
      const Vec16 q00 = *reinterpret_cast<const Vec16*>(vbase + (size_t)k.r00 * row_stride);
      const Vec16 q01 = *reinterpret_cast<const Vec16*>(vbase + (size_t)k.r01 * row_stride);
      const Vec16 q10 = *reinterpret_cast<const Vec16*>(vbase + (size_t)k.r10 * row_stride);
      const Vec16 q11 = *reinterpret_cast<const Vec16*>(vbase + (size_t)k.r11 * row_stride);
      float v00[VEC], v01[VEC], v10[VEC], v11[VEC];
      VecTraits<T>::unpack(q00, v00);
      VecTraits<T>::unpack(q01, v01);
      VecTraits<T>::unpack(q10, v10);
      VecTraits<T>::unpack(q11, v11);
      const float c00 = a * k.w00, c01 = a * k.w01, c10 = a * k.w10, c11 = a * k.w11;
#pragma unroll
      for (int i = 0; i < VEC; ++i)
        acc[i] += c00 * v00[i] + c01 * v01[i] + c10 * v10[i] + c11 * v11[i];
    }
  }
  *reinterpret_cast<Vec16*>(out + (size_t)pair * C + sub * VEC) = VecTraits<T>::pack(acc);
}

// ------------------------------------------------------------------------------------------
// Backward w.r.t. the sampling locations and attention weights ONLY (the value gradient is formed by the sorted gather below):
// the forward's mapping -- C/VEC lanes per (batch, query, head), every corner one 16-byte load per lane -- with three dot
// products over the head's channels per point (sampled value, d/dx, d/dy against grad_out), finished by a shuffle tree over the
// pair's C/VEC lanes. The one-lane-per-channel form below needs C lanes, 2-byte loads and a 5-level tree for the same numbers:
// 200 us per decoder layer against this kernel's time in tools/time_msda_value.py.
// ------------------------------------------------------------------------------------------
template <typename T, int C>
__global__ __launch_bounds__(256) void msda_bwd_points_kernel(const T* __restrict__ value, const float* __restrict__ loc,
                                                             const float* __restrict__ attn, const T* __restrict__ grad_out,
                                                             float* __restrict__ grad_loc, float* __restrict__ grad_attn, MsdaGeom g,
                                                             int n_pairs, int S, int Lq, int nH) {
  constexpr int VEC = VecTraits<T>::kElems;
  constexpr int LPP = C / VEC;  // lanes per pair
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  const int pair = tid / LPP;
  const bool live = pair < n_pairs;  // the host pads the grid to whole waves: dead lanes still take part in the shuffles
  const int spair = live ? pair : 0;
  const int sub = tid % LPP;
  const int h = spair % nH;
  const int b = spair / (nH * Lq);
  const int row_stride = nH * C;
  const T* vbase = value + ((size_t)b * S * nH + h) * C + sub * VEC;
  const float* lp = loc + (size_t)spair * g.P * 2;
  const float* ap = attn + (size_t)spair * g.P;
  float go[VEC];
  VecTraits<T>::unpack(*reinterpret_cast<const Vec16*>(grad_out + (size_t)spair * C + sub * VEC), go);
  // the block's results go through LDS and leave as contiguous runs (its pairs' grad_attn / grad_loc rows are adjacent in
  // memory): a pair's first lane storing 4 + 8 bytes per point made 24 scattered stores per pair
  constexpr int PPB = 256 / LPP;  // pairs per workgroup
  __shared__ float res_a[PPB * DFINE_MAX_POINTS], res_l[PPB * DFINE_MAX_POINTS * 2];
  const int pl = threadIdx.x / LPP;

  for (int l = 0; l < g.n_levels; ++l) {
    const int H = g.H[l], W = g.W[l], start = g.start[l];
#pragma unroll 3
    for (int p = g.pt_begin[l]; p < g.pt_begin[l + 1]; ++p) {
      const float2 xy = *reinterpret_cast<const float2*>(lp + 2 * p);
      const float a = ap[p];
      const Corners k = make_corners(xy.x, xy.y, H, W, start);
      float v00[VEC], v01[VEC], v10[VEC], v11[VEC];
      VecTraits<T>::unpack(*reinterpret_cast<const Vec16*>(vbase + (size_t)k.r00 * row_stride), v00);
      VecTraits<T>::unpack(*reinterpret_cast<const Vec16*>(vbase + (size_t)k.r01 * row_stride), v01);
      VecTraits<T>::unpack(*reinterpret_cast<const Vec16*>(vbase + (size_t)k.r10 * row_stride), v10);
      VecTraits<T>::unpack(*reinterpret_cast<const Vec16*>(vbase + (size_t)k.r11 * row_stride), v11);
      // corners outside the map read a clamped (valid) row and count as zero
      const float m00 = (k.vx0 && k.vy0) ? 1.f : 0.f, m01 = (k.vx1 && k.vy0) ? 1.f : 0.f;
      const float m10 = (k.vx0 && k.vy1) ? 1.f : 0.f, m11 = (k.vx1 && k.vy1) ? 1.f : 0.f;
      float s_v = 0.f, s_x = 0.f, s_y = 0.f;
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        const float a00 = m00 * v00[i], a01 = m01 * v01[i], a10 = m10 * v10[i], a11 = m11 * v11[i];
        s_v += go[i] * (k.w00 * a00 + k.w01 * a01 + k.w10 * a10 + k.w11 * a11);
        s_x += go[i] * ((a01 - a00) * (1.0f - k.fy) + (a11 - a10) * k.fy);
        s_y += go[i] * ((a10 - a00) * (1.0f - k.fx) + (a11 - a01) * k.fx);
      }
      s_v = group_sum<LPP>(s_v), s_x = group_sum<LPP>(s_x), s_y = group_sum<LPP>(s_y);
      if (sub == 0) {
        res_a[pl * g.P + p] = s_v;
        res_l[(pl * g.P + p) * 2] = s_x * a * (float)W;
        res_l[(pl * g.P + p) * 2 + 1] = s_y * a * (float)H;
      }
    }
  }
  __syncthreads();
  const long long pair0 = (long long)blockIdx.x * PPB;
  const long long n_live = min((long long)PPB, (long long)n_pairs - pair0) * g.P;  // floats of grad_attn this block owns
  for (int i = threadIdx.x; i < n_live; i += 256) grad_attn[pair0 * g.P + i] = res_a[i];
  for (int i = threadIdx.x; i < 2 * n_live; i += 256) grad_loc[pair0 * g.P * 2 + i] = res_l[i];
}

// ------------------------------------------------------------------------------------------
// Backward. One lane per channel, C lanes per pair, so every float-atomic wave instruction
// is 64/C contiguous segments of C*4 bytes (two 128-B segments at c=32: the full-rate shape
// for global_atomic_add_f32 on gfx950). grad_attn / grad_loc need a sum over the pair's
// channels: a shuffle tree over C lanes, written by the pair's first lane.
// ------------------------------------------------------------------------------------------
// PK: grad_value is a bf16 map and the scatter uses global_atomic_pk_add_bf16 -- even lanes add (own, right neighbour's)
// channel pair, 64 bytes per corner instead of 128. The float-atomic rate is the floor of this kernel (788 MB per layer
// at ~1.3 TB/s); measured with the same access pattern (tools/micro/atomic_rate.hip) the packed form moves the same
// bytes per second, i.e. twice the adds: 585 -> 296 us per layer's worth. Accumulating in bf16 is what stock autocast does
// for this op (grid_sample's backward adds into a bf16 tensor with packed atomics as well).
template <typename T, int C, bool SCATTER, bool PK = false>
__global__ __launch_bounds__(256) void msda_bwd_kernel(
    const T* __restrict__ value, const float* __restrict__ loc, const float* __restrict__ attn,
    const T* __restrict__ grad_out, void* __restrict__ grad_value_, float* __restrict__ grad_loc,
    float* __restrict__ grad_attn, MsdaGeom g, int n_pairs, int S, int Lq, int nH) {
  float* const grad_value = static_cast<float*>(grad_value_);
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  const int pair = tid / C;
  // n_pairs*C is padded to the block size by the host, so whole C-lane groups are either
  // all live or all dead; dead groups still take part in the shuffles below.
  const bool live = pair < n_pairs;
  const int spair = live ? pair : 0;
  const int ch = tid % C;
  const int h = spair % nH;
  const int b = spair / (nH * Lq);
  const int row_stride = nH * C;
  const size_t head_off = ((size_t)b * S * nH + h) * C + ch;
  const T* vbase = value + head_off;
  float* gvbase = grad_value + head_off;
  const float* lp = loc + (size_t)spair * g.P * 2;
  const float* ap = attn + (size_t)spair * g.P;
  const float go = live ? VecTraits<T>::load1(grad_out + (size_t)spair * C + ch) : 0.0f;

  for (int l = 0; l < g.n_levels; ++l) {
    const int H = g.H[l], W = g.W[l], start = g.start[l];
#pragma unroll 3
    for (int p = g.pt_begin[l]; p < g.pt_begin[l + 1]; ++p) {
      const float2 xy = *reinterpret_cast<const float2*>(lp + 2 * p);
      const float a = ap[p];
      const Corners k = make_corners(xy.x, xy.y, H, W, start);
      const size_t o00 = (size_t)k.r00 * row_stride, o01 = (size_t)k.r01 * row_stride;
      const size_t o10 = (size_t)k.r10 * row_stride, o11 = (size_t)k.r11 * row_stride;
      const bool b00 = k.vx0 && k.vy0, b01 = k.vx1 && k.vy0;
      const bool b10 = k.vx0 && k.vy1, b11 = k.vx1 && k.vy1;
      const float v00 = b00 ? VecTraits<T>::load1(vbase + o00) : 0.0f;
      const float v01 = b01 ? VecTraits<T>::load1(vbase + o01) : 0.0f;
      const float v10 = b10 ? VecTraits<T>::load1(vbase + o10) : 0.0f;
      const float v11 = b11 ? VecTraits<T>::load1(vbase + o11) : 0.0f;

      // d out / d attn, d out / d loc (pixel = loc*size - 0.5  =>  d pixel / d loc = size)
      const float sampled = k.w00 * v00 + k.w01 * v01 + k.w10 * v10 + k.w11 * v11;
      const float dvdx = (v01 - v00) * (1.0f - k.fy) + (v11 - v10) * k.fy;
      const float dvdy = (v10 - v00) * (1.0f - k.fx) + (v11 - v01) * k.fx;
      const float ga = group_sum<C>(go * sampled);
      const float gx = group_sum<C>(go * dvdx) * a * (float)W;
      const float gy = group_sum<C>(go * dvdy) * a * (float)H;
      if (live && ch == 0) {
        grad_attn[(size_t)pair * g.P + p] = ga;
        *reinterpret_cast<float2*>(grad_loc + ((size_t)pair * g.P + p) * 2) = make_float2(gx, gy);
      }
      // d out / d value: scatter-add, fp32, no return value.
      const float ga_go = go * a;
      if (SCATTER && PK) {
        typedef short short2v __attribute__((ext_vector_type(2)));
        typedef __attribute__((address_space(1))) short2v gptr;
        bf16_t* const gv16 = static_cast<bf16_t*>(grad_value_) + head_off;
        const float c00 = ga_go * k.w00, c01 = ga_go * k.w01, c10 = ga_go * k.w10, c11 = ga_go * k.w11;
        // the right neighbour's contributions (every lane of the wave takes part in the shuffles)
        const float n00 = __shfl_xor(c00, 1, 64), n01 = __shfl_xor(c01, 1, 64), n10 = __shfl_xor(c10, 1, 64), n11 = __shfl_xor(c11, 1, 64);
        if (live && (ch & 1) == 0) {
          short2v v;
          if (b00) { v[0] = (short)f32_to_bf16(c00), v[1] = (short)f32_to_bf16(n00); __builtin_amdgcn_global_atomic_fadd_v2bf16((gptr*)(gv16 + o00), v); }
          if (b01) { v[0] = (short)f32_to_bf16(c01), v[1] = (short)f32_to_bf16(n01); __builtin_amdgcn_global_atomic_fadd_v2bf16((gptr*)(gv16 + o01), v); }
          if (b10) { v[0] = (short)f32_to_bf16(c10), v[1] = (short)f32_to_bf16(n10); __builtin_amdgcn_global_atomic_fadd_v2bf16((gptr*)(gv16 + o10), v); }
          if (b11) { v[0] = (short)f32_to_bf16(c11), v[1] = (short)f32_to_bf16(n11); __builtin_amdgcn_global_atomic_fadd_v2bf16((gptr*)(gv16 + o11), v); }
        }
      } else if (SCATTER && live) {
        if (b00) atomicAdd(gvbase + o00, ga_go * k.w00);
        if (b01) atomicAdd(gvbase + o01, ga_go * k.w01);
        if (b10) atomicAdd(gvbase + o10, ga_go * k.w10);
        if (b11) atomicAdd(gvbase + o11, ga_go * k.w11);
      }
    }
  }
}

static int make_geom(MsdaGeom* g, int S, int n_levels, const int32_t* level_hw,
                     const int32_t* num_points) {
  if (n_levels <= 0 || n_levels > DFINE_MAX_LEVELS) return DFINE_E_UNSUPPORTED;
  int start = 0, p = 0;
  g->n_levels = n_levels;
  for (int l = 0; l < n_levels; ++l) {
    const int H = level_hw[2 * l], W = level_hw[2 * l + 1];
    if (H <= 0 || W <= 0 || num_points[l] <= 0) return DFINE_E_BADARG;
    g->H[l] = H;
    g->W[l] = W;
    g->start[l] = start;
    g->pt_begin[l] = p;
    start += H * W;
    p += num_points[l];
  }
  g->pt_begin[n_levels] = p;
  g->P = p;
  if (start != S) return DFINE_E_BADARG;  // levels must tile S exactly
  if (p > DFINE_MAX_POINTS) return DFINE_E_UNSUPPORTED;
  return 0;
}

template <typename T, int C>
static int launch_fwd(const void* value, const float* loc, const float* attn, void* out,
                      const MsdaGeom& g, int B, int S, int Lq, int nH, hipStream_t st) {
  constexpr int LPP = C / VecTraits<T>::kElems;
  const long long n_pairs = (long long)B * Lq * nH;
  const long long threads = n_pairs * LPP;
  const int block = 256;
  const long long grid = (threads + block - 1) / block;
  if (grid > 0x7fffffffLL) return DFINE_E_UNSUPPORTED;
  hipLaunchKernelGGL((msda_fwd_kernel<T, C>), dim3((unsigned)grid), dim3(block), 0, st,
                     (const T*)value, loc, attn, (T*)out, g, (int)n_pairs, S, Lq, nH);
  DFINE_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------
// d(loss)/d(value) WITHOUT global atomics, for every deformable-attention call that sampled the same value
// tensor (all decoder layers do) in ONE launch. The scatter of the straightforward backward is bound by the
// chip's float-atomic rate (788 MB of atomic adds per layer at ~1.26 TB/s = 625 us, 4-6 times per step). Here a
// workgroup owns a SLAB of one level's pixels for one (image, head) -- up to 768 pixels x 32 channels of fp32 =
// 96 KB of LDS --, scans every (layer, query, point) of that level, adds the corners that fall into its slab
// with LDS atomics, and finally WRITES the slab (plain 16-byte stores: every element of grad_value is written
// exactly once, no zero-fill, no read-modify-write). Scanning is cheap (12 bytes per point); the atomic traffic
// never leaves the CU.
// ------------------------------------------------------------------------------------------
constexpr int MSDA_MAX_CALLS = 8;
constexpr int MSDA_SLAB_FLOATS = 24576;  // 96 KB

struct MsdaCalls {
  int n;
  const float* loc[MSDA_MAX_CALLS];
  const float* attn[MSDA_MAX_CALLS];
  const void* grad_out[MSDA_MAX_CALLS];
};

struct MsdaSlabs {
  int n_slabs;                      // over all levels
  int first[DFINE_MAX_LEVELS + 1];  // level l owns slabs [first[l], first[l+1])
  int rows[DFINE_MAX_LEVELS];       // pixel rows per slab of level l
};

constexpr int MSDA_VALUE_THREADS = 1024;  // 16 waves share one slab: the only latency hiding a 96 KB-LDS workgroup has

template <typename T, int C>
__global__ __launch_bounds__(MSDA_VALUE_THREADS) void msda_bwd_value_kernel(MsdaCalls calls, float* __restrict__ grad_value, MsdaGeom g,
                                                                          MsdaSlabs sl, int S, int Lq, int nH) {
  __shared__ __attribute__((aligned(16))) float tile[MSDA_SLAB_FLOATS];
  const int slab = blockIdx.x, bh = blockIdx.y, b = bh / nH, h = bh % nH;
  int l = 0;
  while (l + 1 < g.n_levels && slab >= sl.first[l + 1]) ++l;
  const int H = g.H[l], W = g.W[l];
  const int y_begin = (slab - sl.first[l]) * sl.rows[l], y_end = min(y_begin + sl.rows[l], H);
  const int n_px = (y_end - y_begin) * W;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr int NW = MSDA_VALUE_THREADS / 64;
  for (int i = tid * 4; i < n_px * C; i += MSDA_VALUE_THREADS * 4) *reinterpret_cast<float4*>(tile + i) = make_float4(0.f, 0.f, 0.f, 0.f);
  __syncthreads();
  constexpr int SUBS = 64 / C;  // lane groups of C channels: 2 (C = 32) or 4 (C = 16)
  const int ch = lane % C, sub = lane / C;
  const int p0 = g.pt_begin[l], n_pt = g.pt_begin[l + 1] - p0;
  const int n_pairs = Lq * n_pt;
  for (int c = 0; c < calls.n; ++c) {
    const float* loc = calls.loc[c];
    const float* attn = calls.attn[c];
    const T* go_base = reinterpret_cast<const T*>(calls.grad_out[c]);
    // a wave takes 64 (query, point) pairs at a time: every lane decodes ONE pair (coalesced-ish 8-byte loads, one memory
    // latency per 64 pairs), then the wave walks the pairs that touch this slab, four at a time so that the four
    // grad_out rows they need are in flight together
    for (int base = wave * 64; base < n_pairs; base += NW * 64) {
      const int idx = base + lane;
      bool hit = false;
      int r_up = 0, r_dn = 0, x_step = 0;
      float w00 = 0.f, w01 = 0.f, w10 = 0.f, w11 = 0.f;
      int pair32 = 0;  // (image, query, head) index: < 2^31 / C, checked on the host
      if (idx < n_pairs) {
        const int q = idx / n_pt, p = p0 + idx - q * n_pt;
        const size_t pair = ((size_t)b * Lq + q) * nH + h;
        pair32 = (int)pair;
        const float2 xy = *reinterpret_cast<const float2*>(loc + (pair * g.P + p) * 2);
        const Corners k = make_corners(xy.x, xy.y, H, W, 0);
        const int y0 = k.r00 / W, y1 = k.r10 / W;  // exact: r = y*W + x with x < W
        const bool in0 = k.vy0 && y0 >= y_begin && y0 < y_end, in1 = k.vy1 && y1 >= y_begin && y1 < y_end;
        hit = in0 || in1;
        if (hit) {
          const float a = attn[pair * g.P + p];
          w00 = in0 ? a * k.w00 : 0.f, w01 = in0 ? a * k.w01 : 0.f;
          w10 = in1 ? a * k.w10 : 0.f, w11 = in1 ? a * k.w11 : 0.f;
          // rows are clamped into the slab when their weight is zero (no add is issued for a zero weight)
          r_up = in0 ? k.r00 - y_begin * W : 0;
          r_dn = in1 ? k.r10 - y_begin * W : 0;
          x_step = k.r01 - k.r00;  // 0 or 1 (the right neighbour; 0 at the border, where its weight is 0)
        }
      }
      unsigned long long todo = __ballot(hit);
      while (todo) {
        int j[4];
        float go[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          j[u] = todo ? __ffsll((long long)todo) - 1 : -1;
          if (todo) todo &= todo - 1;
        }
        // broadcasts are scalar reads of one lane (v_readlane with a wave-uniform index), not LDS shuffles: with 16 waves
        // adding into the slab the LDS pipe is the kernel's bottleneck
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          go[u] = 0.f;
          if (j[u] >= 0) {
            const int ju = __builtin_amdgcn_readfirstlane(j[u]);
            const int pj = __builtin_amdgcn_readlane(pair32, ju);
            go[u] = VecTraits<T>::load1(go_base + (size_t)pj * C + ch);
          }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (j[u] < 0) continue;
          const int ju = __builtin_amdgcn_readfirstlane(j[u]);
          const int ru = __builtin_amdgcn_readlane(r_up, ju), rd = __builtin_amdgcn_readlane(r_dn, ju);
          const int xs = __builtin_amdgcn_readlane(x_step, ju);
          const float a00 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(w00), ju));
          const float a01 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(w01), ju));
          const float a10 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(w10), ju));
          const float a11 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(w11), ju));
          if (SUBS == 2) {  // sub 0: the upper corner pair, sub 1: the lower one
            const int r = sub == 0 ? ru : rd;
            const float wa = sub == 0 ? a00 : a10, wb = sub == 0 ? a01 : a11;
            if (wa != 0.f) atomicAdd(tile + r * C + ch, go[u] * wa);
            if (wb != 0.f) atomicAdd(tile + (r + xs) * C + ch, go[u] * wb);
          } else {  // one corner per lane group
            const int r = (sub < 2 ? ru : rd) + ((sub & 1) ? xs : 0);
            const float w = sub == 0 ? a00 : sub == 1 ? a01 : sub == 2 ? a10 : a11;
            if (w != 0.f) atomicAdd(tile + r * C + ch, go[u] * w);
          }
        }
      }
    }
  }
  __syncthreads();
  // flush: pixel px of the slab -> grad_value[b][start + y_begin*W + px][h][0..C)
  float* out = grad_value + (((size_t)b * S + g.start[l] + (size_t)y_begin * W) * nH + h) * C;
  constexpr int V4 = C / 4;
  for (int i = tid; i < n_px * V4; i += MSDA_VALUE_THREADS) {
    const int px = i / V4, v = i - px * V4;
    *reinterpret_cast<float4*>(out + (size_t)px * nH * C + v * 4) = *reinterpret_cast<const float4*>(tile + px * C + v * 4);
  }
}

template <int C>
static bool make_slabs(const MsdaGeom& g, MsdaSlabs* sl) {
  int n = 0;
  for (int l = 0; l < g.n_levels; ++l) {
    const int max_px = MSDA_SLAB_FLOATS / C;
    if (g.W[l] > max_px) return false;  // a single pixel row does not fit: the atomic kernel handles this map
    const int n_l = (g.H[l] * g.W[l] + max_px - 1) / max_px;
    int rows = (g.H[l] + n_l - 1) / n_l;
    while (rows * g.W[l] > max_px) --rows;
    sl->first[l] = n;
    sl->rows[l] = rows;
    n += (g.H[l] + rows - 1) / rows;
  }
  sl->first[g.n_levels] = n;
  sl->n_slabs = n;
  return true;
}

template <typename T, int C>
static int launch_bwd_value(const MsdaCalls& calls, float* grad_value, const MsdaGeom& g, int B, int S, int Lq, int nH,
                            hipStream_t st) {
  MsdaSlabs sl;
  if (!make_slabs<C>(g, &sl)) return DFINE_E_UNSUPPORTED;
  if ((long long)B * nH > 65535 || (long long)B * Lq * nH * C > 0x7fffffffLL) return DFINE_E_UNSUPPORTED;
  hipLaunchKernelGGL((msda_bwd_value_kernel<T, C>), dim3(sl.n_slabs, B * nH), dim3(MSDA_VALUE_THREADS), 0, st, calls, grad_value, g, sl, S, Lq, nH);
  DFINE_LAUNCH_CHECK();
  return 0;
}

template <typename T, int C>
static int launch_bwd_points(const void* value, const float* loc, const float* attn, const void* grad_out, float* grad_loc, float* grad_attn,
                             const MsdaGeom& g, int B, int S, int Lq, int nH, hipStream_t st) {
  constexpr int LPP = C / VecTraits<T>::kElems;
  const long long n_pairs = (long long)B * Lq * nH;
  const long long grid = (n_pairs * LPP + 255) / 256;
  if (grid > 0x7fffffffLL) return DFINE_E_UNSUPPORTED;
  hipLaunchKernelGGL((msda_bwd_points_kernel<T, C>), dim3((unsigned)grid), dim3(256), 0, st, (const T*)value, loc, attn, (const T*)grad_out,
                     grad_loc, grad_attn, g, (int)n_pairs, S, Lq, nH);
  DFINE_LAUNCH_CHECK();
  return 0;
}

template <typename T, int C, bool SCATTER = true, bool PK = false>
static int launch_bwd(const void* value, const float* loc, const float* attn,
                      const void* grad_out, void* grad_value, float* grad_loc, float* grad_attn,
                      const MsdaGeom& g, int B, int S, int Lq, int nH, hipStream_t st) {
  const long long n_pairs = (long long)B * Lq * nH;
  const long long threads = n_pairs * C;
  const int block = 256;
  const long long grid = (threads + block - 1) / block;
  if (grid > 0x7fffffffLL) return DFINE_E_UNSUPPORTED;
  hipLaunchKernelGGL((msda_bwd_kernel<T, C, SCATTER, PK>), dim3((unsigned)grid), dim3(block), 0, st,
                     (const T*)value, loc, attn, (const T*)grad_out, grad_value, grad_loc,
                     grad_attn, g, (int)n_pairs, S, Lq, nH);
  DFINE_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------
// d(loss)/d(value) as a GATHER over per-token lists ("sorted"), for every deformable-attention call that sampled the same value
// tensor, with no float atomics. The scatter above is bound by the chip's atomic rate: every (query, head, point, corner) adds
// c = 32 channels, 193 M float adds per decoder layer. The ADDRESSES are shared by the 32 channels -- there are only 6 M
// (token, head) hits per layer -- so the hits are counting-sorted by destination and each destination's list is summed in
// registers from the grad_out rows (8 MB per call: L2 resident), every element of grad_value written exactly once. A first
// version sorted with GLOBAL integer atomics (one counter per (token, head)): the L2 serves one atomic per cache LINE per
// operation, and 64 lanes with 64 different counters are 64 line operations where the float scatter's 64 lanes cover two --
// count 0.83 ms + fill 1.11 ms per backward pass of D-FINE-m at batch 32, against 2.5 ms for the whole float scatter. The hits of
// one (image, head) only ever land on that (image, head)'s S tokens, so ONE workgroup per (image, head[, token range]) sorts
// next to its own LDS:
//   count   its calls' (query, point, corner) hits into an LDS histogram over its tokens (LDS integer atomics)
//   scan    the histogram in LDS (offsets stay in LDS)
//   fill    8-byte records {call | query, coefficient} into the workgroup's own segment of the record buffer
//   gather  one half-wave (c lanes) per token: the list's records are read 32 at a time (one coalesced load, then lane
//           broadcasts), every record's grad_out row is an independent load, the sum stays in registers, the token's
//           gradient is written once
// grad_value is WRITTEN (every element once). The record buffer needs n_calls * Lq * P * 4 records per workgroup.
// ------------------------------------------------------------------------------------------
constexpr int MSDA_SORT_THREADS = 1024;
constexpr int MSDA_SORT_MAX_TOKENS = 16384;  // tokens per workgroup: 2 LDS int arrays of this size = 128 KB

template <typename T, int C>
__global__ __launch_bounds__(MSDA_SORT_THREADS) void msda_bwd_value_sorted_kernel(MsdaCalls calls, void* __restrict__ grad_value, int out_bf16,
                                                                                 uint2* __restrict__ entries, long long seg_cap, MsdaGeom g,
                                                                                 int S, int Lq, int nH, int splits, int tok_per_split, int dbg) {
  extern __shared__ int smem[];
  int* const cur = smem;                     // [tok_per_split + 1] histogram -> running cursors
  int* const off = smem + tok_per_split + 1;  // [tok_per_split + 1] list offsets
  __shared__ const T* go_tab[MSDA_MAX_CALLS];
  __shared__ int wave_tot[MSDA_SORT_THREADS / 64];
  const int tid = threadIdx.x;
  const int part = blockIdx.x % splits, bh = blockIdx.x / splits, b = bh / nH, h = bh % nH;
  const int lo = part * tok_per_split, hi = min(lo + tok_per_split, S), n_tok = hi - lo;
  uint2* const seg = entries + (size_t)blockIdx.x * seg_cap;
  if (tid < MSDA_MAX_CALLS) go_tab[tid] = static_cast<const T*>(calls.grad_out[tid]);
  for (int i = tid; i <= tok_per_split; i += MSDA_SORT_THREADS) cur[i] = 0;
  __syncthreads();

  const int n_pts = calls.n * Lq * g.P;  // (call, query, point) of this (image, head)
  auto for_hits = [&](auto&& f) {
    for (int i = tid; i < n_pts; i += MSDA_SORT_THREADS) {
      const int p = i % g.P, rest = i / g.P;
      const int q = rest % Lq, call = rest / Lq;
      int l = 0;
      while (l + 1 < g.n_levels && p >= g.pt_begin[l + 1]) ++l;
      const float* lp = nullptr;
      const float* ap = nullptr;
#pragma unroll
      for (int c = 0; c < MSDA_MAX_CALLS; ++c)
        if (c == call) lp = calls.loc[c], ap = calls.attn[c];
      const size_t pt = (((size_t)b * Lq + q) * nH + h) * g.P + p;
      const float2 xy = *reinterpret_cast<const float2*>(lp + pt * 2);
      const float a = ap[pt];
      const Corners k = make_corners(xy.x, xy.y, g.H[l], g.W[l], g.start[l]);
      const bool ok[4] = {k.vx0 && k.vy0, k.vx1 && k.vy0, k.vx0 && k.vy1, k.vx1 && k.vy1};
      const int row[4] = {k.r00, k.r01, k.r10, k.r11};
      const float w[4] = {k.w00, k.w01, k.w10, k.w11};
#pragma unroll
      for (int c = 0; c < 4; ++c)
        if (ok[c] && row[c] >= lo && row[c] < hi) f(row[c] - lo, ((unsigned)call << 28) | (unsigned)q, a * w[c]);
    }
  };
  // ---- count
  for_hits([&](int t, unsigned, float) { atomicAdd(cur + t, 1); });
  __syncthreads();
  // ---- exclusive scan of cur[0 .. n_tok) -> off[], cur[] (thread t owns a contiguous run of tokens)
  {
    const int per = (n_tok + MSDA_SORT_THREADS - 1) / MSDA_SORT_THREADS;
    const int t0 = min(tid * per, n_tok), t1 = min(t0 + per, n_tok);
    int tot = 0;
    for (int t = t0; t < t1; ++t) tot += cur[t];
    int incl = tot;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int o = __shfl_up(incl, d, 64);
      if ((tid & 63) >= d) incl += o;
    }
    if ((tid & 63) == 63) wave_tot[tid >> 6] = incl;
    __syncthreads();
    int run = incl - tot;
    for (int w = 0; w < (tid >> 6); ++w) run += wave_tot[w];
    for (int t = t0; t < t1; ++t) {
      const int c = cur[t];
      off[t] = run, cur[t] = run;
      run += c;
    }
    if (t0 < n_tok && t1 == n_tok) off[n_tok] = run;  // the owner of the last token publishes the total
    if (n_tok == 0 && tid == 0) off[0] = 0;
  }
  __syncthreads();
  if (dbg == 1) return;  // diagnostics (tools/time_msda_value.py): count + scan only
  // ---- fill
  for_hits([&](int t, unsigned src, float coef) {
    const int slot = atomicAdd(cur + t, 1);
    if (slot < seg_cap) seg[slot] = make_uint2(src, __float_as_uint(coef));
  });
  __syncthreads();  // (also orders this workgroup's record stores before its own loads of them)
  if (dbg == 2) return;  // diagnostics: no gather
  // ---- gather: C/4 lanes per token, 4 channels per lane (8-byte bf16 / 16-byte f32 row pieces): with 32 single-channel lanes
  // per token a workgroup had 32 tokens in flight and the phase was one L2 round trip per token long (0.95 of the kernel's
  // 1.2 ms, tools/time_msda_value.py); 8 lanes per token put 128 tokens in flight
  constexpr int LPT = C / 4, GROUPS = MSDA_SORT_THREADS / LPT;
  const int grp = tid / LPT, sub = tid % LPT;
  const size_t go_row = ((size_t)b * Lq * nH + h) * C + sub * 4;  // + query * nH * C
  auto row4 = [&](unsigned src, float (&v)[4]) {
    const T* p = go_tab[src >> 28] + go_row + (size_t)(src & 0x0fffffffu) * nH * C;
    if (sizeof(T) == 2) {
      const uint2 q = *reinterpret_cast<const uint2*>(p);
      v[0] = __uint_as_float(q.x << 16), v[1] = __uint_as_float(q.x & 0xffff0000u);
      v[2] = __uint_as_float(q.y << 16), v[3] = __uint_as_float(q.y & 0xffff0000u);
    } else {
      const float4 q = *reinterpret_cast<const float4*>(p);
      v[0] = q.x, v[1] = q.y, v[2] = q.z, v[3] = q.w;
    }
  };
  for (int t = grp; t < n_tok; t += GROUPS) {
    const int beg = off[t];
    int end = off[t + 1];
    if (end > seg_cap) end = (int)seg_cap;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int j0 = beg; j0 < end; j0 += 2 * LPT) {  // two record chunks per trip: a long list (a token many queries sit on) is a chain of trips
      uint2 e0 = make_uint2(0u, 0u), e1 = make_uint2(0u, 0u);  // a zero record adds 0 * row(call 0, query 0): keeps the loop branch-free
      if (j0 + sub < end) e0 = seg[j0 + sub];
      if (j0 + LPT + sub < end) e1 = seg[j0 + LPT + sub];
      const bool second = j0 + LPT < end;
#pragma unroll
      for (int j = 0; j < LPT; ++j) {
        const unsigned src = __shfl(e0.x, j, LPT);
        const float coef = __uint_as_float(__shfl(e0.y, j, LPT));
        float v[4];
        row4(src, v);
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] += coef * v[i];
      }
      if (second) {
#pragma unroll
        for (int j = 0; j < LPT; ++j) {
          const unsigned src = __shfl(e1.x, j, LPT);
          const float coef = __uint_as_float(__shfl(e1.y, j, LPT));
          float v[4];
          row4(src, v);
#pragma unroll
          for (int i = 0; i < 4; ++i) acc[i] += coef * v[i];
        }
      }
    }
    const size_t o = (((size_t)b * S + lo + t) * nH + h) * C + sub * 4;
    if (out_bf16) {  // the sum is formed in fp32 registers either way; a bf16 map saves the caller's conversion pass
      uint2 pk;
      pk.x = (uint32_t)f32_to_bf16(acc[0]) | ((uint32_t)f32_to_bf16(acc[1]) << 16);
      pk.y = (uint32_t)f32_to_bf16(acc[2]) | ((uint32_t)f32_to_bf16(acc[3]) << 16);
      *reinterpret_cast<uint2*>(static_cast<bf16_t*>(grad_value) + o) = pk;
    } else {
      *reinterpret_cast<float4*>(static_cast<float*>(grad_value) + o) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    }
  }
}

// token ranges per (image, head): enough workgroups to fill the chip (DFINE_MSDA_SORT_WGS, read per call, for measurements)
static int sorted_splits(long long bh, int S) {
  const char* e = getenv("DFINE_MSDA_SORT_WGS");
  const long long want = e ? atoll(e) : 256;
  int splits = 1;
  while (bh * splits < want && splits < 64) splits <<= 1;
  while ((S + splits - 1) / splits > MSDA_SORT_MAX_TOKENS) splits <<= 1;
  return splits;
}

template <typename T, int C>
static int launch_bwd_value_sorted(const MsdaCalls& calls, void* grad_value, int out_bf16, uint2* entries, long long cap, const MsdaGeom& g,
                                   int B, int S, int Lq, int nH, hipStream_t st) {
  if (Lq >= (1 << 28)) return DFINE_E_UNSUPPORTED;
  const long long bh = (long long)B * nH;
  const int splits = sorted_splits(bh, S);
  const int tok_per_split = (S + splits - 1) / splits;
  const long long seg_cap = (long long)calls.n * Lq * g.P * 4;  // every hit of an (image, head) may fall into one range
  const long long blocks = bh * splits;
  if (blocks > 0x7fffffffLL || seg_cap > 0x7fffffffLL) return DFINE_E_UNSUPPORTED;
  if (blocks * seg_cap > cap) return DFINE_E_BADARG;
  const size_t lds = (size_t)2 * (tok_per_split + 1) * sizeof(int);
  static bool attr_set[2][2] = {};
  bool& done = attr_set[sizeof(T) == 2][C == 32];
  if (!done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&msda_bwd_value_sorted_kernel<T, C>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            2 * (MSDA_SORT_MAX_TOKENS + 1) * (int)sizeof(int)) != hipSuccess)
      return DFINE_E_UNSUPPORTED;
    done = true;
  }
  hipLaunchKernelGGL((msda_bwd_value_sorted_kernel<T, C>), dim3((unsigned)blocks), dim3(MSDA_SORT_THREADS), lds, st, calls, grad_value, out_bf16, entries,
                     seg_cap, g, S, Lq, nH, splits, tok_per_split, getenv("DFINE_MSDA_SORT_DEBUG") ? atoi(getenv("DFINE_MSDA_SORT_DEBUG")) : 0);
  DFINE_LAUNCH_CHECK();
  return 0;
}

// records dfine_msda_bwd_value_sorted needs for this geometry (host helper behind dfine_msda_sorted_records)
static long long sorted_records(int n_calls, int B, int S, int Lq, int nH, int P) {
  const long long bh = (long long)B * nH;
  return bh * sorted_splits(bh, S) * ((long long)n_calls * Lq * P * 4);
}

// The glue in front of the gather (reference dfine_decoder.py:148-166): sampling locations from the offset projection and the
// reference boxes, attention weights = softmax over the P points of a head -- one thread per (token, head), forward and
// backward, instead of ~7 + ~10 elementwise launches per decoder layer:
//     loc[p]  = ref.xy + off[p] * (scale[p] * 0.5) * ref.wh          w[p] = softmax_p(logit[p])
//     d off[p] = d loc[p] * (scale[p] * 0.5) * ref.wh                d logit[p] = w[p] * (d w[p] - sum_q w[q] d w[q])
// (the reference boxes are detached in the decoder, dfine_decoder.py:476: no gradient flows into them)
// P > 0: the point count is a compile-time multiple of 4 and a thread's rows (48 / 24 bytes of bf16 in, 96 / 48 bytes of fp32
// out for P = 12) move as 8- and 16-byte accesses from registers; P == 0: any point count (n_points), element by element.
template <typename T, int P>
__global__ __launch_bounds__(256) void msda_prepare_fwd_kernel(const T* __restrict__ off, const T* __restrict__ logit,
                                                              const float* __restrict__ ref, const float* __restrict__ scale,
                                                              float* __restrict__ loc, float* __restrict__ w, long long n_tok, int nH, int n_points) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;  // (token, head)
  if (t >= n_tok * nH) return;
  const long long tok = t / nH;
  const float4 r = *reinterpret_cast<const float4*>(ref + tok * 4);
  if constexpr (P > 0) {
    float o[2 * P], lg[P];
#pragma unroll
    for (int i = 0; i < 2 * P; i += 4) load_vec<T, 4>(off + t * 2 * P + i, o + i);
#pragma unroll
    for (int i = 0; i < P; i += 4) load_vec<T, 4>(logit + t * P + i, lg + i);
    float m = lg[0];
#pragma unroll
    for (int p = 1; p < P; ++p) m = fmaxf(m, lg[p]);
    float sum = 0.f;
#pragma unroll
    for (int p = 0; p < P; ++p) lg[p] = __expf(lg[p] - m), sum += lg[p];
    const float inv = 1.f / sum;
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const float k = scale[p] * 0.5f;
      o[2 * p] = r.x + o[2 * p] * k * r.z;
      o[2 * p + 1] = r.y + o[2 * p + 1] * k * r.w;
      lg[p] *= inv;
    }
#pragma unroll
    for (int i = 0; i < 2 * P; i += 4) store_vec<float, 4>(loc + t * 2 * P + i, o + i);
#pragma unroll
    for (int i = 0; i < P; i += 4) store_vec<float, 4>(w + t * P + i, lg + i);
  } else {
    const int np = n_points;
    const T* o = off + t * np * 2;
    const T* lg = logit + t * np;
    float m = -INFINITY;
    for (int p = 0; p < np; ++p) m = fmaxf(m, VecTraits<T>::load1(lg + p));
    float sum = 0.f;
    for (int p = 0; p < np; ++p) sum += __expf(VecTraits<T>::load1(lg + p) - m);
    const float inv = 1.f / sum;
    for (int p = 0; p < np; ++p) {
      const float k = scale[p] * 0.5f;
      loc[(t * np + p) * 2 + 0] = r.x + VecTraits<T>::load1(o + 2 * p) * k * r.z;
      loc[(t * np + p) * 2 + 1] = r.y + VecTraits<T>::load1(o + 2 * p + 1) * k * r.w;
      w[t * np + p] = __expf(VecTraits<T>::load1(lg + p) - m) * inv;
    }
  }
}

template <typename T, int P>
__global__ __launch_bounds__(256) void msda_prepare_bwd_kernel(const float* __restrict__ g_loc, const float* __restrict__ g_w,
                                                              const float* __restrict__ w, const float* __restrict__ ref,
                                                              const float* __restrict__ scale, T* __restrict__ d_off,
                                                              T* __restrict__ d_logit, long long n_tok, int nH, int n_points) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_tok * nH) return;
  const long long tok = t / nH;
  const float4 r = *reinterpret_cast<const float4*>(ref + tok * 4);
  if constexpr (P > 0) {
    float gl[2 * P], gw[P], ww[P];
#pragma unroll
    for (int i = 0; i < 2 * P; i += 4) load_vec<float, 4>(g_loc + t * 2 * P + i, gl + i);
#pragma unroll
    for (int i = 0; i < P; i += 4) load_vec<float, 4>(g_w + t * P + i, gw + i), load_vec<float, 4>(w + t * P + i, ww + i);
    float dot = 0.f;
#pragma unroll
    for (int p = 0; p < P; ++p) dot += ww[p] * gw[p];
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const float k = scale[p] * 0.5f;
      gl[2 * p] *= k * r.z;
      gl[2 * p + 1] *= k * r.w;
      gw[p] = ww[p] * (gw[p] - dot);
    }
#pragma unroll
    for (int i = 0; i < 2 * P; i += 4) store_vec<T, 4>(d_off + t * 2 * P + i, gl + i);
#pragma unroll
    for (int i = 0; i < P; i += 4) store_vec<T, 4>(d_logit + t * P + i, gw + i);
  } else {
    const int np = n_points;
    float dot = 0.f;
    for (int p = 0; p < np; ++p) dot += w[t * np + p] * g_w[t * np + p];
    for (int p = 0; p < np; ++p) {
      const float k = scale[p] * 0.5f;
      store1(d_off + (t * np + p) * 2 + 0, g_loc[(t * np + p) * 2 + 0] * k * r.z);
      store1(d_off + (t * np + p) * 2 + 1, g_loc[(t * np + p) * 2 + 1] * k * r.w);
      store1(d_logit + t * np + p, w[t * np + p] * (g_w[t * np + p] - dot));
    }
  }
}

template <typename T>
static void launch_prepare_fwd(unsigned blocks, hipStream_t st, const void* off, const void* logit, const float* ref, const float* scale,
                               float* loc, float* w, long long n_tok, int nH, int P) {
#define DFINE_PREP_FWD(PP) hipLaunchKernelGGL((msda_prepare_fwd_kernel<T, PP>), dim3(blocks), dim3(256), 0, st, (const T*)off, (const T*)logit, ref, scale, loc, w, n_tok, nH, P)
  if (P == 12) DFINE_PREP_FWD(12);
  else if (P == 8) DFINE_PREP_FWD(8);
  else if (P == 16) DFINE_PREP_FWD(16);
  else DFINE_PREP_FWD(0);
#undef DFINE_PREP_FWD
}

template <typename T>
static void launch_prepare_bwd(unsigned blocks, hipStream_t st, const float* g_loc, const float* g_w, const float* w, const float* ref,
                               const float* scale, void* d_off, void* d_logit, long long n_tok, int nH, int P) {
#define DFINE_PREP_BWD(PP) hipLaunchKernelGGL((msda_prepare_bwd_kernel<T, PP>), dim3(blocks), dim3(256), 0, st, g_loc, g_w, w, ref, scale, (T*)d_off, (T*)d_logit, n_tok, nH, P)
  if (P == 12) DFINE_PREP_BWD(12);
  else if (P == 8) DFINE_PREP_BWD(8);
  else if (P == 16) DFINE_PREP_BWD(16);
  else DFINE_PREP_BWD(0);
#undef DFINE_PREP_BWD
}

}  // namespace dfine

using namespace dfine;

extern "C" int dfine_msda_prepare_fwd(const void* offsets, const void* logits, int dtype, const float* ref_boxes, const float* point_scale,
                                      float* loc, float* weights, long long n_tokens, int nH, int P, void* stream) {
  if (!offsets || !logits || !ref_boxes || !point_scale || !loc || !weights) return DFINE_E_BADARG;
  if (n_tokens <= 0 || nH <= 0 || P <= 0) return DFINE_E_BADARG;
  const long long blocks = (n_tokens * nH + 255) / 256;
  if (blocks > 0x7fffffffLL) return DFINE_E_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == DFINE_DT_BF16)
    launch_prepare_fwd<bf16_t>((unsigned)blocks, st, offsets, logits, ref_boxes, point_scale, loc, weights, n_tokens, nH, P);
  else if (dtype == DFINE_DT_F32)
    launch_prepare_fwd<float>((unsigned)blocks, st, offsets, logits, ref_boxes, point_scale, loc, weights, n_tokens, nH, P);
  else
    return DFINE_E_UNSUPPORTED;
  DFINE_LAUNCH_CHECK();
  return 0;
}

extern "C" int dfine_msda_prepare_bwd(const float* grad_loc, const float* grad_weights, const float* weights, const float* ref_boxes,
                                      const float* point_scale, int dtype, void* d_offsets, void* d_logits, long long n_tokens, int nH,
                                      int P, void* stream) {
  if (!grad_loc || !grad_weights || !weights || !ref_boxes || !point_scale || !d_offsets || !d_logits) return DFINE_E_BADARG;
  if (n_tokens <= 0 || nH <= 0 || P <= 0) return DFINE_E_BADARG;
  const long long blocks = (n_tokens * nH + 255) / 256;
  if (blocks > 0x7fffffffLL) return DFINE_E_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == DFINE_DT_BF16)
    launch_prepare_bwd<bf16_t>((unsigned)blocks, st, grad_loc, grad_weights, weights, ref_boxes, point_scale, d_offsets, d_logits, n_tokens, nH, P);
  else if (dtype == DFINE_DT_F32)
    launch_prepare_bwd<float>((unsigned)blocks, st, grad_loc, grad_weights, weights, ref_boxes, point_scale, d_offsets, d_logits, n_tokens, nH, P);
  else
    return DFINE_E_UNSUPPORTED;
  DFINE_LAUNCH_CHECK();
  return 0;
}

extern "C" int dfine_abi_version(void) { return DFINE_ABI_VERSION; }

extern "C" int dfine_msda_fwd(const void* value, int value_dtype, const float* loc,
                              const float* attn, void* out, int B, int S, int Lq, int nH, int c,
                              int n_levels, const int32_t* level_hw, const int32_t* num_points,
                              void* stream) {
  if (!value || !loc || !attn || !out || !level_hw || !num_points) return DFINE_E_BADARG;
  if (B <= 0 || S <= 0 || Lq <= 0 || nH <= 0) return DFINE_E_BADARG;
  MsdaGeom g;
  if (int rc = make_geom(&g, S, n_levels, level_hw, num_points)) return rc;
  hipStream_t st = (hipStream_t)stream;
  if (value_dtype == DFINE_DT_BF16 && c == 32)
    return launch_fwd<bf16_t, 32>(value, loc, attn, out, g, B, S, Lq, nH, st);
  if (value_dtype == DFINE_DT_BF16 && c == 16)
    return launch_fwd<bf16_t, 16>(value, loc, attn, out, g, B, S, Lq, nH, st);
  if (value_dtype == DFINE_DT_F32 && c == 32)
    return launch_fwd<float, 32>(value, loc, attn, out, g, B, S, Lq, nH, st);
  if (value_dtype == DFINE_DT_F32 && c == 16)
    return launch_fwd<float, 16>(value, loc, attn, out, g, B, S, Lq, nH, st);
  return DFINE_E_UNSUPPORTED;
}

extern "C" int dfine_msda_bwd(const void* value, int value_dtype, const float* loc,
                              const float* attn, const void* grad_out, void* grad_value, int grad_value_dtype,
                              float* grad_loc, float* grad_attn, int B, int S, int Lq, int nH,
                              int c, int n_levels, const int32_t* level_hw,
                              const int32_t* num_points, void* stream) {
  if (!value || !loc || !attn || !grad_out || !grad_value || !grad_loc || !grad_attn ||
      !level_hw || !num_points)
    return DFINE_E_BADARG;
  if (B <= 0 || S <= 0 || Lq <= 0 || nH <= 0) return DFINE_E_BADARG;
  MsdaGeom g;
  if (int rc = make_geom(&g, S, n_levels, level_hw, num_points)) return rc;
  hipStream_t st = (hipStream_t)stream;
  if (grad_value_dtype == DFINE_DT_BF16) {  // packed bf16 atomics into a bf16 gradient map
    if (value_dtype != DFINE_DT_BF16 || (uintptr_t)grad_value % 4 != 0) return DFINE_E_UNSUPPORTED;
    if (c == 32) return launch_bwd<bf16_t, 32, true, true>(value, loc, attn, grad_out, grad_value, grad_loc, grad_attn, g, B, S, Lq, nH, st);
    if (c == 16) return launch_bwd<bf16_t, 16, true, true>(value, loc, attn, grad_out, grad_value, grad_loc, grad_attn, g, B, S, Lq, nH, st);
    return DFINE_E_UNSUPPORTED;
  }
  if (grad_value_dtype != DFINE_DT_F32) return DFINE_E_UNSUPPORTED;
  if (value_dtype == DFINE_DT_BF16 && c == 32)
    return launch_bwd<bf16_t, 32>(value, loc, attn, grad_out, grad_value, grad_loc, grad_attn, g,
                                  B, S, Lq, nH, st);
  if (value_dtype == DFINE_DT_BF16 && c == 16)
    return launch_bwd<bf16_t, 16>(value, loc, attn, grad_out, grad_value, grad_loc, grad_attn, g,
                                  B, S, Lq, nH, st);
  if (value_dtype == DFINE_DT_F32 && c == 32)
    return launch_bwd<float, 32>(value, loc, attn, grad_out, grad_value, grad_loc, grad_attn, g,
                                 B, S, Lq, nH, st);
  if (value_dtype == DFINE_DT_F32 && c == 16)
    return launch_bwd<float, 16>(value, loc, attn, grad_out, grad_value, grad_loc, grad_attn, g,
                                 B, S, Lq, nH, st);
  return DFINE_E_UNSUPPORTED;
}

// grad_loc / grad_attn only (the value gradient is left to dfine_msda_bwd_value)
extern "C" int dfine_msda_bwd_points(const void* value, int value_dtype, const float* loc, const float* attn, const void* grad_out,
                                     float* grad_loc, float* grad_attn, int B, int S, int Lq, int nH, int c, int n_levels,
                                     const int32_t* level_hw, const int32_t* num_points, void* stream) {
  if (!value || !loc || !attn || !grad_out || !grad_loc || !grad_attn || !level_hw || !num_points) return DFINE_E_BADARG;
  if (B <= 0 || S <= 0 || Lq <= 0 || nH <= 0) return DFINE_E_BADARG;
  MsdaGeom g;
  if (int rc = make_geom(&g, S, n_levels, level_hw, num_points)) return rc;
  hipStream_t st = (hipStream_t)stream;
  if (value_dtype == DFINE_DT_BF16 && c == 32) return launch_bwd_points<bf16_t, 32>(value, loc, attn, grad_out, grad_loc, grad_attn, g, B, S, Lq, nH, st);
  if (value_dtype == DFINE_DT_BF16 && c == 16) return launch_bwd_points<bf16_t, 16>(value, loc, attn, grad_out, grad_loc, grad_attn, g, B, S, Lq, nH, st);
  if (value_dtype == DFINE_DT_F32 && c == 32) return launch_bwd_points<float, 32>(value, loc, attn, grad_out, grad_loc, grad_attn, g, B, S, Lq, nH, st);
  if (value_dtype == DFINE_DT_F32 && c == 16) return launch_bwd_points<float, 16>(value, loc, attn, grad_out, grad_loc, grad_attn, g, B, S, Lq, nH, st);
  return DFINE_E_UNSUPPORTED;
}

extern "C" int dfine_msda_bwd_value(int value_dtype, const void* const* loc, const void* const* attn, const void* const* grad_out,
                                    int n_calls, float* grad_value, int B, int S, int Lq, int nH, int c, int n_levels,
                                    const int32_t* level_hw, const int32_t* num_points, void* stream) {
  if (!loc || !attn || !grad_out || !grad_value || !level_hw || !num_points) return DFINE_E_BADARG;
  if (n_calls <= 0 || B <= 0 || S <= 0 || Lq <= 0 || nH <= 0) return DFINE_E_BADARG;
  if (n_calls > MSDA_MAX_CALLS) return DFINE_E_UNSUPPORTED;
  MsdaGeom g;
  if (int rc = make_geom(&g, S, n_levels, level_hw, num_points)) return rc;
  MsdaCalls calls;
  calls.n = n_calls;
  for (int i = 0; i < n_calls; ++i) {
    if (!loc[i] || !attn[i] || !grad_out[i]) return DFINE_E_BADARG;
    calls.loc[i] = (const float*)loc[i], calls.attn[i] = (const float*)attn[i], calls.grad_out[i] = grad_out[i];
  }
  hipStream_t st = (hipStream_t)stream;
  if (value_dtype == DFINE_DT_BF16 && c == 32) return launch_bwd_value<bf16_t, 32>(calls, grad_value, g, B, S, Lq, nH, st);
  if (value_dtype == DFINE_DT_BF16 && c == 16) return launch_bwd_value<bf16_t, 16>(calls, grad_value, g, B, S, Lq, nH, st);
  if (value_dtype == DFINE_DT_F32 && c == 32) return launch_bwd_value<float, 32>(calls, grad_value, g, B, S, Lq, nH, st);
  if (value_dtype == DFINE_DT_F32 && c == 16) return launch_bwd_value<float, 16>(calls, grad_value, g, B, S, Lq, nH, st);
  return DFINE_E_UNSUPPORTED;
}

extern "C" long long dfine_msda_sorted_records(int n_calls, int B, int S, int Lq, int nH, int P) {
  if (n_calls <= 0 || B <= 0 || S <= 0 || Lq <= 0 || nH <= 0 || P <= 0) return DFINE_E_BADARG;
  return sorted_records(n_calls, B, S, Lq, nH, P);
}

extern "C" int dfine_msda_bwd_value_sorted(int value_dtype, const void* const* loc, const void* const* attn, const void* const* grad_out,
                                           int n_calls, void* grad_value, int grad_dtype, void* entries, long long entries_cap, int B, int S,
                                           int Lq, int nH, int c, int n_levels, const int32_t* level_hw, const int32_t* num_points, void* stream) {
  if (!loc || !attn || !grad_out || !grad_value || !entries || !level_hw || !num_points) return DFINE_E_BADARG;
  if (grad_dtype != DFINE_DT_F32 && grad_dtype != DFINE_DT_BF16) return DFINE_E_UNSUPPORTED;
  const int ob = grad_dtype == DFINE_DT_BF16;
  if (n_calls <= 0 || B <= 0 || S <= 0 || Lq <= 0 || nH <= 0 || entries_cap <= 0) return DFINE_E_BADARG;
  if (n_calls > MSDA_MAX_CALLS) return DFINE_E_UNSUPPORTED;
  MsdaGeom g;
  if (int rc = make_geom(&g, S, n_levels, level_hw, num_points)) return rc;
  MsdaCalls calls;
  calls.n = n_calls;
  for (int i = 0; i < MSDA_MAX_CALLS; ++i) {
    if (i < n_calls && (!loc[i] || !attn[i] || !grad_out[i])) return DFINE_E_BADARG;
    const int j = i < n_calls ? i : 0;  // unused slots alias call 0: every table entry is a valid pointer
    calls.loc[i] = (const float*)loc[j], calls.attn[i] = (const float*)attn[j], calls.grad_out[i] = grad_out[j];
  }
  hipStream_t st = (hipStream_t)stream;
  uint2* en = static_cast<uint2*>(entries);
  if (value_dtype == DFINE_DT_BF16 && c == 32) return launch_bwd_value_sorted<bf16_t, 32>(calls, grad_value, ob, en, entries_cap, g, B, S, Lq, nH, st);
  if (value_dtype == DFINE_DT_BF16 && c == 16) return launch_bwd_value_sorted<bf16_t, 16>(calls, grad_value, ob, en, entries_cap, g, B, S, Lq, nH, st);
  if (value_dtype == DFINE_DT_F32 && c == 32) return launch_bwd_value_sorted<float, 32>(calls, grad_value, ob, en, entries_cap, g, B, S, Lq, nH, st);
  if (value_dtype == DFINE_DT_F32 && c == 16) return launch_bwd_value_sorted<float, 16>(calls, grad_value, ob, en, entries_cap, g, B, S, Lq, nH, st);
  return DFINE_E_UNSUPPORTED;
}
