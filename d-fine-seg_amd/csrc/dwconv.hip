// Depthwise kxk convolutions (k in {3,5}, stride in {1,2}, pad (k-1)/2, no bias) on channel-last
// maps: the 5x5 light-block convs and 3x3 stride-2 downsamples of HGNetv2 and the SCDown convs of
// the encoder (reference src/d_fine/arch/hgnetv2.py:83-112,295-304, hybrid_encoder.py:96-103,
// all nn.Conv2d(groups=channels)). These have ~k*k MACs per byte moved: HBM/L2-bound, no MFMA.
// The stock path (MIOpen) falls back to naive / generic grouped-conv kernels for them: 6.4 ms per
// weight-gradient call measured at B=32 (profiles/), for ~13 MB of traffic.
//
// Layout: x [B, H, W, C], y [B, OH, OW, C] (T = bf16 or f32), w [C, k, k] f32 (the fp32 master
// weight, read directly: no per-step cast), dw [C, k, k] f32.
#include <stdlib.h>

#include "common.h"

namespace dfine {

struct DwGeom {
  int B, H, W, C, OH, OW;
};

template <int K>
__device__ __forceinline__ void stage_weights(float* w_s, const float* __restrict__ w, int c_base,
                                              int c_count, int tid, int nthreads) {
  // w_s[tap][c] <- w[(c_base + c)][tap]: channel innermost so a thread's channels are contiguous
  for (int i = tid; i < K * K * c_count; i += nthreads) {
    const int tap = i / c_count, c = i - tap * c_count;
    w_s[i] = w[(size_t)(c_base + c) * (K * K) + tap];
  }
}

// Forward: one thread = one output pixel x VEC channels. blockDim = (channel groups, pixels).
template <typename T, int K, int S>
__global__ __launch_bounds__(256) void dwconv_fwd_kernel(const T* __restrict__ x,
                                                         const float* __restrict__ w,
                                                         T* __restrict__ y, DwGeom g, int n_pix) {
  constexpr int VEC = VecTraits<T>::kElems;
  constexpr int PAD = (K - 1) / 2;
  extern __shared__ __attribute__((aligned(16))) float w_s[];
  const int c_slice = blockDim.x * VEC;
  const int c_base = blockIdx.y * c_slice;
  stage_weights<K>(w_s, w, c_base, c_slice, threadIdx.y * blockDim.x + threadIdx.x, blockDim.x * blockDim.y);
  __syncthreads();
  const int pix = xcd_block_id() * blockDim.y + threadIdx.y;
  if (pix >= n_pix) return;
  const int ox = pix % g.OW, oy = (pix / g.OW) % g.OH, b = pix / (g.OW * g.OH);
  const int c_loc = threadIdx.x * VEC;
  float acc[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) acc[i] = 0.f;
  const T* xb = x + (size_t)b * g.H * g.W * g.C + c_base + c_loc;
#pragma unroll
  for (int ky = 0; ky < K; ++ky) {
    const int iy = oy * S + ky - PAD;
    if (iy < 0 || iy >= g.H) continue;
#pragma unroll
    for (int kx = 0; kx < K; ++kx) {
      const int ix = ox * S + kx - PAD;
      if (ix < 0 || ix >= g.W) continue;
      float v[VEC];
      VecTraits<T>::unpack(*reinterpret_cast<const Vec16*>(xb + ((size_t)iy * g.W + ix) * g.C), v);
      const float* wt = w_s + (ky * K + kx) * c_slice + c_loc;
#pragma unroll
      for (int i = 0; i < VEC; ++i) acc[i] += v[i] * wt[i];
    }
  }
  *reinterpret_cast<Vec16*>(y + (size_t)pix * g.C + c_base + c_loc) = VecTraits<T>::pack(acc);
}

// Backward w.r.t. the input: dx[iy,ix] = sum over taps of dy[(iy+pad-ky)/S, (ix+pad-kx)/S] * w[ky,kx].
template <typename T, int K, int S>
__global__ __launch_bounds__(256) void dwconv_bwd_data_kernel(const T* __restrict__ dy,
                                                              const float* __restrict__ w,
                                                              T* __restrict__ dx, DwGeom g,
                                                              int n_pix) {
  constexpr int VEC = VecTraits<T>::kElems;
  constexpr int PAD = (K - 1) / 2;
  extern __shared__ __attribute__((aligned(16))) float w_s[];
  const int c_slice = blockDim.x * VEC;
  const int c_base = blockIdx.y * c_slice;
  stage_weights<K>(w_s, w, c_base, c_slice, threadIdx.y * blockDim.x + threadIdx.x, blockDim.x * blockDim.y);
  __syncthreads();
  const int pix = xcd_block_id() * blockDim.y + threadIdx.y;  // input pixel
  if (pix >= n_pix) return;
  const int ix = pix % g.W, iy = (pix / g.W) % g.H, b = pix / (g.W * g.H);
  const int c_loc = threadIdx.x * VEC;
  float acc[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) acc[i] = 0.f;
  const T* dyb = dy + (size_t)b * g.OH * g.OW * g.C + c_base + c_loc;
#pragma unroll
  for (int ky = 0; ky < K; ++ky) {
    const int ny = iy + PAD - ky;
    if (ny < 0 || (ny % S) != 0) continue;
    const int oy = ny / S;
    if (oy >= g.OH) continue;
#pragma unroll
    for (int kx = 0; kx < K; ++kx) {
      const int nx = ix + PAD - kx;
      if (nx < 0 || (nx % S) != 0) continue;
      const int ox = nx / S;
      if (ox >= g.OW) continue;
      float v[VEC];
      VecTraits<T>::unpack(*reinterpret_cast<const Vec16*>(dyb + ((size_t)oy * g.OW + ox) * g.C), v);
      const float* wt = w_s + (ky * K + kx) * c_slice + c_loc;
#pragma unroll
      for (int i = 0; i < VEC; ++i) acc[i] += v[i] * wt[i];
    }
  }
  *reinterpret_cast<Vec16*>(dx + (size_t)pix * g.C + c_base + c_loc) = VecTraits<T>::pack(acc);
}

// Stride-1 forward AND input gradient (the same correlation with the taps flipped: dx[iy, ix] = sum dy[iy + PAD - ky,
// ix + PAD - kx] * w[ky, kx]) as a STRIP kernel: one thread = VEC channels x STRIP consecutive pixels of one output row. Per
// kernel row it loads the STRIP + K - 1 input pixels the strip touches ONCE into registers (packed) and slides the K taps
// over them: (STRIP + K - 1) * K loads for STRIP outputs -- 7.5 per output at K = 5, STRIP = 8, against 25 for the
// one-pixel-per-thread kernel above, which is bound by load issue / L1 bandwidth, not by HBM (measured 1.9x the
// algorithmic HBM bytes and 0.8 TB/s effective). blockDim = (channel groups, strips).
template <typename T, int K, int STRIP, bool FLIP>
__global__ __launch_bounds__(256) void dwconv_strip_kernel(const T* __restrict__ x, const float* __restrict__ w, T* __restrict__ y,
                                                           double* __restrict__ stats, DwGeom g, int n_strips, int strips_per_row) {
  constexpr int VEC = VecTraits<T>::kElems;
  constexpr int PAD = (K - 1) / 2;
  constexpr int WIN = STRIP + K - 1;
  extern __shared__ __attribute__((aligned(16))) float w_s[];
  const int c_slice = blockDim.x * VEC;
  const int c_base = blockIdx.y * c_slice;
  stage_weights<K>(w_s, w, c_base, c_slice, threadIdx.y * blockDim.x + threadIdx.x, blockDim.x * blockDim.y);
  __syncthreads();
  const int strip = xcd_block_id() * blockDim.y + threadIdx.y;  // neighbouring strips share an L2 (common.h)
  const bool live = strip < n_strips;
  const int c_loc = threadIdx.x * VEC;
  if (!live && stats == nullptr) return;
  float s1[VEC], s2[VEC];  // per-channel sum / sum of squares of the ROUNDED outputs (the BatchNorm that follows reads them)
#pragma unroll
  for (int i = 0; i < VEC; ++i) s1[i] = s2[i] = 0.f;
  if (live) {
  const int sx = strip % strips_per_row, row = strip / strips_per_row;  // row = (b, oy); stride 1: OH = H, OW = W
  const int oy = row % g.H, b = row / g.H, ox0 = sx * STRIP;
  float acc[STRIP][VEC];
#pragma unroll
  for (int s_ = 0; s_ < STRIP; ++s_)
#pragma unroll
    for (int i = 0; i < VEC; ++i) acc[s_][i] = 0.f;
  const T* xb = x + (size_t)b * g.H * g.W * g.C + c_base + c_loc;
#pragma unroll
  for (int ky = 0; ky < K; ++ky) {
    const int iy = oy + ky - PAD;
    if (iy < 0 || iy >= g.H) continue;
    const T* xr = xb + (size_t)iy * g.W * g.C;
    Vec16 win[WIN];
#pragma unroll
    for (int j = 0; j < WIN; ++j) {
      const int ix = ox0 + j - PAD;
      win[j].w[0] = win[j].w[1] = win[j].w[2] = win[j].w[3] = 0u;
      if (ix >= 0 && ix < g.W) win[j] = *reinterpret_cast<const Vec16*>(xr + (size_t)ix * g.C);
    }
#pragma unroll
    for (int j = 0; j < WIN; ++j) {
      float v[VEC];
      VecTraits<T>::unpack(win[j], v);
#pragma unroll
      for (int kx = 0; kx < K; ++kx) {
        const int s_ = j - kx;  // output pixel of the strip this input pixel feeds through tap kx
        if (s_ < 0 || s_ >= STRIP) continue;
        const float* wt = w_s + (FLIP ? ((K - 1 - ky) * K + (K - 1 - kx)) : (ky * K + kx)) * c_slice + c_loc;
#pragma unroll
        for (int i = 0; i < VEC; ++i) acc[s_][i] += v[i] * wt[i];
      }
    }
  }
#pragma unroll
  for (int s_ = 0; s_ < STRIP; ++s_) {
    const int ox = ox0 + s_;
    if (ox >= g.W) continue;
    const Vec16 pk = VecTraits<T>::pack(acc[s_]);
    *reinterpret_cast<Vec16*>(y + ((size_t)row * g.W + ox) * g.C + c_base + c_loc) = pk;
    if (stats != nullptr) {
      float r[VEC];
      VecTraits<T>::unpack(pk, r);
#pragma unroll
      for (int i = 0; i < VEC; ++i) s1[i] += r[i], s2[i] += r[i] * r[i];
    }
  }
  }  // live
  if (stats != nullptr) {
    // column sums over the block's strips through LDS (behind the staged weights), then one fp64 atomic per (quantity,
    // channel) into the 8-replica layout dfine_bn_act_fwd reads: the depthwise conv's BatchNorm needs no statistics pass
    float* part = w_s + K * K * c_slice;  // [2][blockDim.y][c_slice]
    __syncthreads();
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      part[(0 * blockDim.y + threadIdx.y) * c_slice + c_loc + i] = s1[i];
      part[(1 * blockDim.y + threadIdx.y) * c_slice + c_loc + i] = s2[i];
    }
    __syncthreads();
    const int tid = threadIdx.y * blockDim.x + threadIdx.x;
    for (int e = tid; e < 2 * c_slice; e += blockDim.x * blockDim.y) {  // c_slice can be the whole block (256 channels)
      const int q = e / c_slice, c = e % c_slice;
      float a = 0.f;
      for (int r = 0; r < (int)blockDim.y; ++r) a += part[(q * blockDim.y + r) * c_slice + c];
      atomicAdd(stats + ((size_t)(blockIdx.x & 7) * 2 + q) * g.C + c_base + c, (double)a);
    }
  }
}

// Backward w.r.t. the weights: dw[c][tap] = sum over (b,oy,ox) of dy[b,oy,ox,c] * x[b,iy,ix,c].
template <typename T>
struct Quad;  // 4 channels as one 8-byte (bf16) or 16-byte (f32) load
template <>
struct Quad<bf16_t> {
  __device__ static __forceinline__ void load(const bf16_t* p, float* f) {
    const uint2 q = *reinterpret_cast<const uint2*>(p);
    f[0] = __uint_as_float(q.x << 16);
    f[1] = __uint_as_float(q.x & 0xffff0000u);
    f[2] = __uint_as_float(q.y << 16);
    f[3] = __uint_as_float(q.y & 0xffff0000u);
  }
};
template <>
struct Quad<float> {
  __device__ static __forceinline__ void load(const float* p, float* f) {
    const float4 q = *reinterpret_cast<const float4*>(p);
    f[0] = q.x; f[1] = q.y; f[2] = q.z; f[3] = q.w;
  }
};

// One thread = 4 channels x one kernel row ky x a strided set of output rows (b, oy). Along a row it
// slides a K-wide register window over x, so each output pixel costs one dy load and S new x loads
// for K*4 FMAs. Its K*4 partial sums are written to a workspace and summed by a second small kernel:
// with thousands of threads adding into the same C*K*K addresses, direct float atomics serialise
// (measured 8.5 ms/step at 4k blocks vs 3.3 ms at 320 blocks; the two-pass form has no such trade).
// blockDim = (channel quads, K kernel rows, rows per block).
template <typename T, int K, int S>
__global__ __launch_bounds__(256) void dwconv_bwd_weight_kernel(const T* __restrict__ x,
                                                                const T* __restrict__ dy,
                                                                float* __restrict__ ws, DwGeom g,
                                                                int n_rows, int rows_per_thread) {
  constexpr int PAD = (K - 1) / 2;
  const int c0 = (blockIdx.y * blockDim.x + threadIdx.x) * 4;
  const int ky = threadIdx.y;
  float acc[K][4];
#pragma unroll
  for (int t = 0; t < K; ++t)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[t][i] = 0.f;
  const unsigned bx = xcd_block_id();  // adjacent row ranges (they share K - 1 rows of x) on one XCD's L2 (common.h)
  if ((long long)bx * blockDim.z * rows_per_thread >= n_rows) return;  // padding blocks of xcd_grid()
  const int first = bx * blockDim.z * rows_per_thread + threadIdx.z;
  for (int n = 0; n < rows_per_thread; ++n) {
    const int row = first + n * blockDim.z;  // (b, oy)
    if (row >= n_rows) break;
    const int oy = row % g.OH, b = row / g.OH;
    const int iy = oy * S + ky - PAD;
    if (iy < 0 || iy >= g.H) continue;
    const T* xr = x + ((size_t)(b * g.H + iy) * g.W) * g.C + c0;
    const T* dr = dy + ((size_t)row * g.OW) * g.C + c0;
    float w[K][4];
#pragma unroll
    for (int j = 0; j < K; ++j) {  // window for ox = 0: ix = j - PAD
      const int ix = j - PAD;
      if (ix >= 0 && ix < g.W) Quad<T>::load(xr + (size_t)ix * g.C, w[j]);
      else w[j][0] = w[j][1] = w[j][2] = w[j][3] = 0.f;
    }
    for (int ox = 0; ox < g.OW; ++ox) {
      float d[4];
      Quad<T>::load(dr + (size_t)ox * g.C, d);
#pragma unroll
      for (int j = 0; j < K; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[j][i] += d[i] * w[j][i];
      // slide by S: keep the K-S newest taps, fetch S new columns for ox+1
#pragma unroll
      for (int j = 0; j + S < K; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) w[j][i] = w[j + S][i];
#pragma unroll
      for (int j = K - S; j < K; ++j) {
        const int ix = (ox + 1) * S + j - PAD;
        if (ix >= 0 && ix < g.W) Quad<T>::load(xr + (size_t)ix * g.C, w[j]);
        else w[j][0] = w[j][1] = w[j][2] = w[j][3] = 0.f;
      }
    }
  }
  // partial sums of this thread's rows: ws[part][c][ky][kx], part = one (block, row-lane) pair
  float* out = ws + ((size_t)(bx * blockDim.z + threadIdx.z) * g.C + c0) * (K * K) + ky * K;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int kx = 0; kx < K; ++kx) out[(size_t)i * (K * K) + kx] = acc[kx][i];
}

// dw[e] += sum over a chunk of partials of ws[part][e]; gridDim.y chunks, one atomic per (e, chunk).
__global__ __launch_bounds__(256) void dwconv_reduce_partials_kernel(const float* __restrict__ ws, float* __restrict__ dw,
                                                                     int n_out, int n_parts) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n_out) return;
  const int per = (n_parts + gridDim.y - 1) / gridDim.y;
  const int lo = blockIdx.y * per, hi = min(lo + per, n_parts);
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;  // four loads in flight: the loop is a chain of ~2 us round trips otherwise
  int p = lo;
  for (; p + 3 < hi; p += 4) {
    s0 += ws[(size_t)p * n_out + e];
    s1 += ws[(size_t)(p + 1) * n_out + e];
    s2 += ws[(size_t)(p + 2) * n_out + e];
    s3 += ws[(size_t)(p + 3) * n_out + e];
  }
  for (; p < hi; ++p) s0 += ws[(size_t)p * n_out + e];
  if (hi > lo) atomicAdd(dw + e, (s0 + s1) + (s2 + s3));
}

static int largest_divisor_leq(int n, int cap) {
  for (int d = cap; d >= 1; --d)
    if (n % d == 0) return d;
  return 1;
}

template <typename T, int K, int S>
static int run_fwd_or_bwd_data(bool fwd, const void* in, const float* w, void* out, const DwGeom& g,
                               hipStream_t st, double* stats = nullptr) {
  constexpr int VEC = VecTraits<T>::kElems;
  if (g.C % VEC != 0) return DFINE_E_UNSUPPORTED;
  const int cgb = largest_divisor_leq(g.C / VEC, 32);
  const int pb = 256 / cgb;
  static const bool no_strip = getenv("DFINE_DWCONV_NO_STRIP") != nullptr;
  if (S == 1 && !no_strip) {  // forward and input gradient are the same strip kernel (taps flipped for the gradient)
    constexpr int STRIP = 8;
    const int spr = (g.W + STRIP - 1) / STRIP;
    const long long n_strips = (long long)g.B * g.H * spr;
    if (n_strips > 0x7fffffffLL) return DFINE_E_UNSUPPORTED;
    dim3 block(cgb, pb), grid(xcd_grid((unsigned)((n_strips + pb - 1) / pb)), g.C / (cgb * VEC));
    const size_t lds = (size_t)K * K * cgb * VEC * sizeof(float) + (stats ? (size_t)2 * pb * cgb * VEC * sizeof(float) : 0);
    if (fwd)
      hipLaunchKernelGGL((dwconv_strip_kernel<T, K, STRIP, false>), grid, block, lds, st, (const T*)in, w, (T*)out, stats, g, (int)n_strips, spr);
    else
      hipLaunchKernelGGL((dwconv_strip_kernel<T, K, STRIP, true>), grid, block, lds, st, (const T*)in, w, (T*)out, nullptr, g, (int)n_strips, spr);
    DFINE_LAUNCH_CHECK();
    return 0;
  }
  if (stats) return DFINE_E_UNSUPPORTED;  // the statistics epilogue exists in the stride-1 strip kernel only
  const long long n_pix = fwd ? (long long)g.B * g.OH * g.OW : (long long)g.B * g.H * g.W;
  dim3 block(cgb, pb), grid(xcd_grid((unsigned)((n_pix + pb - 1) / pb)), g.C / (cgb * VEC));
  const size_t lds = (size_t)K * K * cgb * VEC * sizeof(float);
  if (fwd)
    hipLaunchKernelGGL((dwconv_fwd_kernel<T, K, S>), grid, block, lds, st, (const T*)in, w, (T*)out, g, (int)n_pix);
  else
    hipLaunchKernelGGL((dwconv_bwd_data_kernel<T, K, S>), grid, block, lds, st, (const T*)in, w, (T*)out, g, (int)n_pix);
  DFINE_LAUNCH_CHECK();
  return 0;
}

struct WrwPlan {
  dim3 grid, block;
  int rpt;
  long long parts;  // partial-sum slots = grid.x * block.z
};

template <int K>
static int plan_wrw(WrwPlan* P, const DwGeom& g) {
  if (g.C % 4 != 0) return DFINE_E_UNSUPPORTED;
  const int cgb = largest_divisor_leq(g.C / 4, 32);
  int rb = 256 / (cgb * K);
  if (rb < 1) rb = 1;
  const long long n_rows = (long long)g.B * g.OH;
  long long want_blocks = 2048 / (g.C / (cgb * 4));  // enough waves to hide the load latency
  if (want_blocks < 1) want_blocks = 1;
  long long rpt = (n_rows + want_blocks * rb - 1) / (want_blocks * rb);
  if (rpt < 1) rpt = 1;
  P->block = dim3(cgb, K, rb);
  P->grid = dim3((unsigned)((n_rows + (long long)rb * rpt - 1) / ((long long)rb * rpt)), g.C / (cgb * 4));
  P->rpt = (int)rpt;
  P->parts = (long long)P->grid.x * rb;
  return 0;
}

template <typename T, int K, int S>
static int run_bwd_weight(const void* x, const void* dy, float* dw, float* ws, long long ws_floats, const DwGeom& g,
                          hipStream_t st) {
  WrwPlan P;
  if (int rc = plan_wrw<K>(&P, g)) return rc;
  const long long n_out = (long long)g.C * K * K;
  if (!ws || ws_floats < P.parts * n_out) return DFINE_E_BADARG;
  hipLaunchKernelGGL((dwconv_bwd_weight_kernel<T, K, S>), dim3(xcd_grid(P.grid.x), P.grid.y), P.block, 0, st, (const T*)x, (const T*)dy, ws, g,
                     g.B * g.OH, P.rpt);
  DFINE_LAUNCH_CHECK();
  const int chunks = P.parts >= 256 ? 64 : P.parts >= 64 ? 16 : 1;
  hipLaunchKernelGGL(dwconv_reduce_partials_kernel, dim3((unsigned)((n_out + 255) / 256), chunks), dim3(256), 0, st, ws, dw,
                     (int)n_out, (int)P.parts);
  DFINE_LAUNCH_CHECK();
  return 0;
}

#define DFINE_DW_DISPATCH(CALL)                                                     \
  do {                                                                              \
    if (dtype == DFINE_DT_BF16) {                                                   \
      if (k == 3 && stride == 1) return CALL(bf16_t, 3, 1);                         \
      if (k == 3 && stride == 2) return CALL(bf16_t, 3, 2);                         \
      if (k == 5 && stride == 1) return CALL(bf16_t, 5, 1);                         \
      if (k == 5 && stride == 2) return CALL(bf16_t, 5, 2);                         \
    } else if (dtype == DFINE_DT_F32) {                                             \
      if (k == 3 && stride == 1) return CALL(float, 3, 1);                          \
      if (k == 3 && stride == 2) return CALL(float, 3, 2);                          \
      if (k == 5 && stride == 1) return CALL(float, 5, 1);                          \
      if (k == 5 && stride == 2) return CALL(float, 5, 2);                          \
    }                                                                               \
    return DFINE_E_UNSUPPORTED;                                                     \
  } while (0)

static int make_dw_geom(DwGeom* g, int B, int H, int W, int C, int k, int stride) {
  if (B <= 0 || H <= 0 || W <= 0 || C <= 0) return DFINE_E_BADARG;
  if ((k != 3 && k != 5) || (stride != 1 && stride != 2)) return DFINE_E_UNSUPPORTED;
  const int pad = (k - 1) / 2;
  *g = DwGeom{B, H, W, C, (H + 2 * pad - k) / stride + 1, (W + 2 * pad - k) / stride + 1};
  return 0;
}

}  // namespace dfine

using namespace dfine;

extern "C" int dfine_dwconv_fwd(const void* x, const float* w, void* y, int dtype, int B, int H,
                                int W, int C, int k, int stride, double* stats, void* stream) {
  if (!x || !w || !y) return DFINE_E_BADARG;
  DwGeom g;
  if (int rc = make_dw_geom(&g, B, H, W, C, k, stride)) return rc;
  if (stats && stride != 1) return DFINE_E_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
#define CALL(T, K, S) run_fwd_or_bwd_data<T, K, S>(true, x, w, y, g, st, stats)
  DFINE_DW_DISPATCH(CALL);
#undef CALL
}

extern "C" int dfine_dwconv_bwd_data(const void* dy, const float* w, void* dx, int dtype, int B,
                                     int H, int W, int C, int k, int stride, void* stream) {
  if (!dy || !w || !dx) return DFINE_E_BADARG;
  DwGeom g;
  if (int rc = make_dw_geom(&g, B, H, W, C, k, stride)) return rc;
  hipStream_t st = (hipStream_t)stream;
#define CALL(T, K, S) run_fwd_or_bwd_data<T, K, S>(false, dy, w, dx, g, st)
  DFINE_DW_DISPATCH(CALL);
#undef CALL
}

extern "C" long long dfine_dwconv_bwd_weight_workspace(int B, int H, int W, int C, int k, int stride) {
  DwGeom g;
  if (make_dw_geom(&g, B, H, W, C, k, stride)) return -1;
  WrwPlan P;
  if ((k == 3 ? plan_wrw<3>(&P, g) : plan_wrw<5>(&P, g))) return -1;
  return P.parts * (long long)C * k * k;
}

extern "C" int dfine_dwconv_bwd_weight(const void* x, const void* dy, float* dw, float* workspace,
                                       long long workspace_floats, int dtype, int B, int H, int W, int C, int k,
                                       int stride, void* stream) {
  if (!x || !dy || !dw) return DFINE_E_BADARG;
  DwGeom g;
  if (int rc = make_dw_geom(&g, B, H, W, C, k, stride)) return rc;
  hipStream_t st = (hipStream_t)stream;
#define CALL(T, K, S) run_bwd_weight<T, K, S>(x, dy, dw, workspace, workspace_floats, g, st)
  DFINE_DW_DISPATCH(CALL);
#undef CALL
}
