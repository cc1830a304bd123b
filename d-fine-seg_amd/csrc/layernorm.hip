// Residual add + LayerNorm over the last dimension, forward and backward, in the activations' own dtype:
//     s = clamp(x + res), y = (s - mean(s)) * rstd(s) * gamma + beta
// (post-norm blocks of the decoder and the AIFI layer: reference src/d_fine/arch/dfine_decoder.py:233-255
// `norm1(tgt + dropout(attn))`, `norm3(clamp(tgt + ffn))`, Gate :258-271; hybrid_encoder.py:274-290).
// Under autocast the stock LayerNorm runs in fp32: a cast in, a cast out, three backward kernels and the
// residual add are separate launches over [B*Lq, D]; here each direction is one launch and bf16 stays bf16.
//
// One wave per row (D = 64 * E elements, E in {2, 4, 8}: each lane owns E consecutive elements = one
// 4/8/16-byte access for bf16), statistics by wave shuffles in fp32. Backward accumulates d gamma / d beta
// per lane over the rows a wave walks, then one LDS reduction and float atomics per workgroup.
#include <stdlib.h>

#include "common.h"

namespace dfine {

template <typename T>
__device__ __forceinline__ float round_to(float v);
template <>
__device__ __forceinline__ float round_to<float>(float v) { return v; }
template <>
__device__ __forceinline__ float round_to<bf16_t>(float v) { return bf16_to_f32(f32_to_bf16(v)); }

// E consecutive elements per lane; E = 6 (D = 384, the x-size AIFI layer) is not a power-of-two vector: three 2-element pieces
template <typename T, int E>
__device__ __forceinline__ void lv(const T* p, float* f) {
  if constexpr (E == 6) {
    load_vec<T, 2>(p, f), load_vec<T, 2>(p + 2, f + 2), load_vec<T, 2>(p + 4, f + 4);
  } else {
    load_vec<T, E>(p, f);
  }
}
template <typename T, int E>
__device__ __forceinline__ void sv(T* p, const float* f) {
  if constexpr (E == 6) {
    store_vec<T, 2>(p, f), store_vec<T, 2>(p + 2, f + 2), store_vec<T, 2>(p + 4, f + 4);
  } else {
    store_vec<T, E>(p, f);
  }
}

template <typename T, int E>
__global__ __launch_bounds__(256) void add_ln_fwd_kernel(const T* __restrict__ x, const T* __restrict__ res,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         T* __restrict__ y, T* __restrict__ s_out, float* __restrict__ mean_out,
                                                         float* __restrict__ rstd_out, long long M, float eps, float clamp) {
  constexpr int D = 64 * E;
  const int lane = threadIdx.x & 63;
  const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6), n_waves = (long long)gridDim.x * 4;
  float g[E], b[E];
#pragma unroll
  for (int i = 0; i < E; ++i) g[i] = gamma[lane * E + i], b[i] = beta[lane * E + i];
  for (long long row = wave; row < M; row += n_waves) {
    float v[E];
    lv<T, E>(x + row * D + lane * E, v);
    if (res) {
      float r[E];
      lv<T, E>(res + row * D + lane * E, r);
#pragma unroll
      for (int i = 0; i < E; ++i) v[i] += r[i];
    }
    if (clamp > 0.f) {
#pragma unroll
      for (int i = 0; i < E; ++i) v[i] = fminf(fmaxf(v[i], -clamp), clamp);
    }
    if (s_out) {  // the normalised quantity as the backward will read it (rounded to T)
      sv<T, E>(s_out + row * D + lane * E, v);
#pragma unroll
      for (int i = 0; i < E; ++i) v[i] = round_to<T>(v[i]);
    }
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < E; ++i) sum += v[i];
    const float mean = group_sum<64>(sum) * (1.f / D);
    float var = 0.f;
#pragma unroll
    for (int i = 0; i < E; ++i) var += (v[i] - mean) * (v[i] - mean);
    const float rstd = rsqrtf(group_sum<64>(var) * (1.f / D) + eps);
    float o[E];
#pragma unroll
    for (int i = 0; i < E; ++i) o[i] = (v[i] - mean) * rstd * g[i] + b[i];
    sv<T, E>(y + row * D + lane * E, o);
    if (lane == 0) mean_out[row] = mean, rstd_out[row] = rstd;
  }
}

template <typename T, int E>
__global__ __launch_bounds__(256) void add_ln_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ s,
                                                         const float* __restrict__ gamma, const float* __restrict__ mean_in,
                                                         const float* __restrict__ rstd_in, T* __restrict__ ds,
                                                         float* __restrict__ dgamma, float* __restrict__ dbeta, long long M,
                                                         float clamp) {
  constexpr int D = 64 * E;
  __shared__ float red[2][4][D];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const long long wave = (long long)blockIdx.x * 4 + w, n_waves = (long long)gridDim.x * 4;
  float g[E], ag[E], ab[E];
#pragma unroll
  for (int i = 0; i < E; ++i) g[i] = gamma[lane * E + i], ag[i] = 0.f, ab[i] = 0.f;
  for (long long row = wave; row < M; row += n_waves) {
    float d[E], v[E];
    lv<T, E>(dy + row * D + lane * E, d);
    lv<T, E>(s + row * D + lane * E, v);
    const float mean = mean_in[row], rstd = rstd_in[row];
    float c1 = 0.f, c2 = 0.f, xh[E];
#pragma unroll
    for (int i = 0; i < E; ++i) {
      xh[i] = (v[i] - mean) * rstd;
      const float dg = d[i] * g[i];
      c1 += dg;
      c2 += dg * xh[i];
      ag[i] += d[i] * xh[i];
      ab[i] += d[i];
    }
    c1 = group_sum<64>(c1) * (1.f / D);
    c2 = group_sum<64>(c2) * (1.f / D);
    float o[E];
#pragma unroll
    for (int i = 0; i < E; ++i) {
      o[i] = rstd * (d[i] * g[i] - c1 - xh[i] * c2);
      if (clamp > 0.f && (v[i] <= -clamp || v[i] >= clamp)) o[i] = 0.f;  // clamp's gradient (saturated: 0)
    }
    sv<T, E>(ds + row * D + lane * E, o);
  }
#pragma unroll
  for (int i = 0; i < E; ++i) red[0][w][lane * E + i] = ag[i], red[1][w][lane * E + i] = ab[i];
  __syncthreads();
  for (int c = threadIdx.x; c < D; c += blockDim.x) {
    atomicAdd(dgamma + c, red[0][0][c] + red[0][1][c] + red[0][2][c] + red[0][3][c]);
    atomicAdd(dbeta + c, red[1][0][c] + red[1][1][c] + red[1][2][c] + red[1][3][c]);
  }
}

// bf16, D = 128 / 256: LPR = 16 lanes per row, FOUR rows per wave. A lane owns NCH chunks of 8 elements (chunk k of the row
// at element (k * LPR + sub) * 8: every load / store instruction covers whole rows contiguously). One row per wave made the
// 15 744-token launches of the decoder a chain of two 64-lane reductions per row with nothing else in flight (22 us backward
// for 24 MB); four rows per wave run four such chains side by side, each 4 shuffle levels deep instead of 6.
template <int LPR, int NCH>
__global__ __launch_bounds__(256) void add_ln_fwd_rows_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ res,
                                                              const float* __restrict__ gamma, const float* __restrict__ beta,
                                                              bf16_t* __restrict__ y, bf16_t* __restrict__ s_out, float* __restrict__ mean_out,
                                                              float* __restrict__ rstd_out, long long M, float eps, float clamp) {
  constexpr int D = LPR * NCH * 8, RPW = 64 / LPR;
  const int lane = threadIdx.x & 63, sub = lane % LPR, rsel = lane / LPR;
  const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6), n_waves = (long long)gridDim.x * 4;
  float g[NCH][8], b[NCH][8];
#pragma unroll
  for (int k = 0; k < NCH; ++k)
#pragma unroll
    for (int i = 0; i < 8; ++i) g[k][i] = gamma[(k * LPR + sub) * 8 + i], b[k][i] = beta[(k * LPR + sub) * 8 + i];
  for (long long row0 = wave * RPW; row0 < M; row0 += n_waves * RPW) {
    const long long row = row0 + rsel;
    const bool live = row < M;
    const long long r = live ? row : M - 1;  // dead row slots of the last wave recompute a valid row and store nothing
    float v[NCH][8];
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
      const long long o = r * D + (k * LPR + sub) * 8;
      load_vec<bf16_t, 8>(x + o, v[k]);
      if (res) {
        float t[8];
        load_vec<bf16_t, 8>(res + o, t);
#pragma unroll
        for (int i = 0; i < 8; ++i) v[k][i] += t[i];
      }
      if (clamp > 0.f) {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[k][i] = fminf(fmaxf(v[k][i], -clamp), clamp);
      }
      if (s_out) {  // the normalised quantity as the backward will read it (rounded to bf16)
        if (live) store_vec<bf16_t, 8>(s_out + o, v[k]);
#pragma unroll
        for (int i = 0; i < 8; ++i) v[k][i] = round_to<bf16_t>(v[k][i]);
      }
    }
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < NCH; ++k)
#pragma unroll
      for (int i = 0; i < 8; ++i) sum += v[k][i];
    const float mean = group_sum<LPR>(sum) * (1.f / D);
    float var = 0.f;
#pragma unroll
    for (int k = 0; k < NCH; ++k)
#pragma unroll
      for (int i = 0; i < 8; ++i) var += (v[k][i] - mean) * (v[k][i] - mean);
    const float rstd = rsqrtf(group_sum<LPR>(var) * (1.f / D) + eps);
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
      float o[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) o[i] = (v[k][i] - mean) * rstd * g[k][i] + b[k][i];
      if (live) store_vec<bf16_t, 8>(y + r * D + (k * LPR + sub) * 8, o);
    }
    if (live && sub == 0) mean_out[row] = mean, rstd_out[row] = rstd;
  }
}

template <int LPR, int NCH>
__global__ __launch_bounds__(256) void add_ln_bwd_rows_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ s,
                                                              const float* __restrict__ gamma, const float* __restrict__ mean_in,
                                                              const float* __restrict__ rstd_in, bf16_t* __restrict__ ds,
                                                              float* __restrict__ dgamma, float* __restrict__ dbeta, long long M, float clamp) {
  constexpr int D = LPR * NCH * 8, RPW = 64 / LPR;
  __shared__ float red[2][4][D];
  const int lane = threadIdx.x & 63, sub = lane % LPR, rsel = lane / LPR, w = threadIdx.x >> 6;
  const long long wave = (long long)blockIdx.x * 4 + w, n_waves = (long long)gridDim.x * 4;
  float g[NCH][8], ag[NCH][8], ab[NCH][8];
#pragma unroll
  for (int k = 0; k < NCH; ++k)
#pragma unroll
    for (int i = 0; i < 8; ++i) g[k][i] = gamma[(k * LPR + sub) * 8 + i], ag[k][i] = 0.f, ab[k][i] = 0.f;
  for (long long row0 = wave * RPW; row0 < M; row0 += n_waves * RPW) {
    const long long row = row0 + rsel;
    const bool live = row < M;
    const long long r = live ? row : M - 1;
    const float mean = mean_in[r], rstd = rstd_in[r], keep = live ? 1.f : 0.f;
    float d[NCH][8], v[NCH][8], xh[NCH][8];
    float c1 = 0.f, c2 = 0.f;
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
      const long long o = r * D + (k * LPR + sub) * 8;
      load_vec<bf16_t, 8>(dy + o, d[k]);
      load_vec<bf16_t, 8>(s + o, v[k]);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        xh[k][i] = (v[k][i] - mean) * rstd;
        const float dg = d[k][i] * g[k][i];
        c1 += dg;
        c2 += dg * xh[k][i];
        ag[k][i] += keep * d[k][i] * xh[k][i];
        ab[k][i] += keep * d[k][i];
      }
    }
    c1 = group_sum<LPR>(c1) * (1.f / D);
    c2 = group_sum<LPR>(c2) * (1.f / D);
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
      float o[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        o[i] = rstd * (d[k][i] * g[k][i] - c1 - xh[k][i] * c2);
        if (clamp > 0.f && (v[k][i] <= -clamp || v[k][i] >= clamp)) o[i] = 0.f;  // clamp's gradient (saturated: 0)
      }
      if (live) store_vec<bf16_t, 8>(ds + r * D + (k * LPR + sub) * 8, o);
    }
  }
  // the wave's RPW row slots hold partial column sums of the same columns: fold them, then waves through LDS
#pragma unroll
  for (int k = 0; k < NCH; ++k)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
#pragma unroll
      for (int m = LPR; m < 64; m <<= 1) ag[k][i] += __shfl_xor(ag[k][i], m, 64), ab[k][i] += __shfl_xor(ab[k][i], m, 64);
    }
  if (rsel == 0) {
#pragma unroll
    for (int k = 0; k < NCH; ++k)
#pragma unroll
      for (int i = 0; i < 8; ++i) red[0][w][(k * LPR + sub) * 8 + i] = ag[k][i], red[1][w][(k * LPR + sub) * 8 + i] = ab[k][i];
  }
  __syncthreads();
  for (int c = threadIdx.x; c < D; c += blockDim.x) {
    atomicAdd(dgamma + c, red[0][0][c] + red[0][1][c] + red[0][2][c] + red[0][3][c]);
    atomicAdd(dbeta + c, red[1][0][c] + red[1][1][c] + red[1][2][c] + red[1][3][c]);
  }
}

template <int LPR, int NCH>
static int launch_ln_rows(bool fwd, const void* a, const void* b, const float* gamma, const float* beta, void* out, void* s_out,
                          float* mean, float* rstd, float* dgamma, float* dbeta, long long M, float eps, float clamp, hipStream_t st) {
  constexpr int RPB = 4 * (64 / LPR);  // rows per workgroup and trip
  long long blocks = (M + RPB - 1) / RPB;
  static const int bwd_cap = getenv("DFINE_LN_BWD_BLOCKS") ? atoi(getenv("DFINE_LN_BWD_BLOCKS")) : 256;  // see launch_ln
  const long long cap = fwd ? 2048 : bwd_cap;
  if (blocks > cap) blocks = cap;
  if (fwd)
    hipLaunchKernelGGL((add_ln_fwd_rows_kernel<LPR, NCH>), dim3((unsigned)blocks), dim3(256), 0, st, (const bf16_t*)a, (const bf16_t*)b, gamma,
                       beta, (bf16_t*)out, (bf16_t*)s_out, mean, rstd, M, eps, clamp);
  else
    hipLaunchKernelGGL((add_ln_bwd_rows_kernel<LPR, NCH>), dim3((unsigned)blocks), dim3(256), 0, st, (const bf16_t*)a, (const bf16_t*)b, gamma,
                       mean, rstd, (bf16_t*)out, dgamma, dbeta, M, clamp);
  DFINE_LAUNCH_CHECK();
  return 0;
}

template <typename T, int E>
static int launch_ln(bool fwd, const void* a, const void* b, const float* gamma, const float* beta, void* out, void* s_out,
                     float* mean, float* rstd, float* dgamma, float* dbeta, long long M, float eps, float clamp, hipStream_t st) {
  // grid-stride over rows. Backward: every workgroup ends with 2*D float atomics on the SAME 2*D addresses;
  // 2048 workgroups made that tail 60 us of serialised same-address adds, 256 keep it under 10 us
  long long blocks = (M + 3) / 4;
  static const int bwd_cap = getenv("DFINE_LN_BWD_BLOCKS") ? atoi(getenv("DFINE_LN_BWD_BLOCKS")) : 256;
  const long long cap = fwd ? 2048 : bwd_cap;
  if (blocks > cap) blocks = cap;
  if (fwd)
    hipLaunchKernelGGL((add_ln_fwd_kernel<T, E>), dim3((unsigned)blocks), dim3(256), 0, st, (const T*)a, (const T*)b, gamma, beta,
                       (T*)out, (T*)s_out, mean, rstd, M, eps, clamp);
  else
    hipLaunchKernelGGL((add_ln_bwd_kernel<T, E>), dim3((unsigned)blocks), dim3(256), 0, st, (const T*)a, (const T*)b, gamma, mean, rstd,
                       (T*)out, dgamma, dbeta, M, clamp);
  DFINE_LAUNCH_CHECK();
  return 0;
}

#define DFINE_LN_DISPATCH(CALL)                                 \
  do {                                                          \
    if (dtype == DFINE_DT_BF16) {                               \
      if (D == 128) return CALL(bf16_t, 2);                     \
      if (D == 256) return CALL(bf16_t, 4);                     \
      if (D == 384) return CALL(bf16_t, 6);                     \
      if (D == 512) return CALL(bf16_t, 8);                     \
    } else if (dtype == DFINE_DT_F32) {                         \
      if (D == 128) return CALL(float, 2);                      \
      if (D == 256) return CALL(float, 4);                      \
      if (D == 384) return CALL(float, 6);                      \
    }                                                           \
    return DFINE_E_UNSUPPORTED;                                 \
  } while (0)

// The decoder layer's gateway (reference dfine_decoder.py:181-196 Gate): gates = sigmoid(Linear(cat(a, b))), out = norm(g1 * a + g2 * b).
// The Linear and the LayerNorm have their own kernels; this is the elementwise middle -- mix = sigmoid(pre[:, :D]) * a +
// sigmoid(pre[:, D:]) * b -- and its backward (d pre, d a, d b from d mix) as ONE kernel each instead of sigmoid, two products
// (forward) and four products, a concat and sigmoid_backward (autograd's backward): ~9 launches of 4-8 us per layer.
template <typename T>
__global__ __launch_bounds__(256) void gate_mix_fwd_kernel(const T* __restrict__ pre, const T* __restrict__ a, const T* __restrict__ b,
                                                          T* __restrict__ out, long long M, int D) {
  const long long n = M * D;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const long long m = i / D;
    const int d = (int)(i - m * D);
    const float p1 = VecTraits<T>::load1(pre + m * 2 * D + d), p2 = VecTraits<T>::load1(pre + m * 2 * D + D + d);
    // each factor is rounded to T like the separate ops round their outputs (sigmoid -> T, product -> T), the sum is the
    // LayerNorm's fused residual add
    const float s1 = round_to<T>(1.f / (1.f + __expf(-p1))), s2 = round_to<T>(1.f / (1.f + __expf(-p2)));
    const float x = round_to<T>(s1 * VecTraits<T>::load1(a + i)), y = round_to<T>(s2 * VecTraits<T>::load1(b + i));
    store1(out + i, x + y);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void gate_mix_bwd_kernel(const T* __restrict__ pre, const T* __restrict__ a, const T* __restrict__ b,
                                                          const T* __restrict__ g, T* __restrict__ dpre, T* __restrict__ da,
                                                          T* __restrict__ db, long long M, int D) {
  const long long n = M * D;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const long long m = i / D;
    const int d = (int)(i - m * D);
    const float p1 = VecTraits<T>::load1(pre + m * 2 * D + d), p2 = VecTraits<T>::load1(pre + m * 2 * D + D + d);
    const float s1 = round_to<T>(1.f / (1.f + __expf(-p1))), s2 = round_to<T>(1.f / (1.f + __expf(-p2)));
    const float gv = VecTraits<T>::load1(g + i), av = VecTraits<T>::load1(a + i), bv = VecTraits<T>::load1(b + i);
    store1(da + i, gv * s1);
    store1(db + i, gv * s2);
    store1(dpre + m * 2 * D + d, gv * av * s1 * (1.f - s1));
    store1(dpre + m * 2 * D + D + d, gv * bv * s2 * (1.f - s2));
  }
}

static bool ln_rows_enabled() {  // DFINE_LN_ROWS=0: the one-row-per-wave kernels for every shape (A/B measurements), read per call
  const char* e = getenv("DFINE_LN_ROWS");
  return !(e && atoi(e) == 0);
}

}  // namespace dfine

using namespace dfine;

extern "C" int dfine_gate_mix_fwd(const void* pre, const void* a, const void* b, int dtype, void* out, long long M, int D, void* stream) {
  if (!pre || !a || !b || !out) return DFINE_E_BADARG;
  if (M <= 0 || D <= 0) return DFINE_E_BADARG;
  long long blocks = (M * D + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == DFINE_DT_BF16)
    hipLaunchKernelGGL(gate_mix_fwd_kernel<bf16_t>, dim3((unsigned)blocks), dim3(256), 0, st, (const bf16_t*)pre, (const bf16_t*)a, (const bf16_t*)b, (bf16_t*)out, M, D);
  else if (dtype == DFINE_DT_F32)
    hipLaunchKernelGGL(gate_mix_fwd_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, st, (const float*)pre, (const float*)a, (const float*)b, (float*)out, M, D);
  else
    return DFINE_E_UNSUPPORTED;
  DFINE_LAUNCH_CHECK();
  return 0;
}

extern "C" int dfine_gate_mix_bwd(const void* pre, const void* a, const void* b, const void* g, int dtype, void* dpre, void* da, void* db,
                                  long long M, int D, void* stream) {
  if (!pre || !a || !b || !g || !dpre || !da || !db) return DFINE_E_BADARG;
  if (M <= 0 || D <= 0) return DFINE_E_BADARG;
  long long blocks = (M * D + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == DFINE_DT_BF16)
    hipLaunchKernelGGL(gate_mix_bwd_kernel<bf16_t>, dim3((unsigned)blocks), dim3(256), 0, st, (const bf16_t*)pre, (const bf16_t*)a, (const bf16_t*)b,
                       (const bf16_t*)g, (bf16_t*)dpre, (bf16_t*)da, (bf16_t*)db, M, D);
  else if (dtype == DFINE_DT_F32)
    hipLaunchKernelGGL(gate_mix_bwd_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, st, (const float*)pre, (const float*)a, (const float*)b,
                       (const float*)g, (float*)dpre, (float*)da, (float*)db, M, D);
  else
    return DFINE_E_UNSUPPORTED;
  DFINE_LAUNCH_CHECK();
  return 0;
}

extern "C" int dfine_add_layernorm_fwd(const void* x, const void* res, int dtype, const float* gamma, const float* beta,
                                       void* y, void* s_out, float* mean, float* rstd, long long M, int D, float eps,
                                       float clamp, void* stream) {
  if (!x || !gamma || !beta || !y || !mean || !rstd) return DFINE_E_BADARG;
  if (M <= 0 || D <= 0 || eps <= 0.f) return DFINE_E_BADARG;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == DFINE_DT_BF16 && ln_rows_enabled()) {  // four rows per wave (see add_ln_fwd_rows_kernel)
    if (D == 256) return launch_ln_rows<16, 2>(true, x, res, gamma, beta, y, s_out, mean, rstd, nullptr, nullptr, M, eps, clamp, st);
    if (D == 128) return launch_ln_rows<16, 1>(true, x, res, gamma, beta, y, s_out, mean, rstd, nullptr, nullptr, M, eps, clamp, st);
  }
#define FWD(T, E) launch_ln<T, E>(true, x, res, gamma, beta, y, s_out, mean, rstd, nullptr, nullptr, M, eps, clamp, st)
  DFINE_LN_DISPATCH(FWD);
#undef FWD
}

extern "C" int dfine_add_layernorm_bwd(const void* dy, const void* s, int dtype, const float* gamma, const float* mean,
                                       const float* rstd, void* ds, float* dgamma, float* dbeta, long long M, int D,
                                       float clamp, void* stream) {
  if (!dy || !s || !gamma || !mean || !rstd || !ds || !dgamma || !dbeta) return DFINE_E_BADARG;
  if (M <= 0 || D <= 0) return DFINE_E_BADARG;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == DFINE_DT_BF16 && ln_rows_enabled()) {
    if (D == 256) return launch_ln_rows<16, 2>(false, dy, s, gamma, nullptr, ds, nullptr, const_cast<float*>(mean), const_cast<float*>(rstd), dgamma, dbeta, M, 0.f, clamp, st);
    if (D == 128) return launch_ln_rows<16, 1>(false, dy, s, gamma, nullptr, ds, nullptr, const_cast<float*>(mean), const_cast<float*>(rstd), dgamma, dbeta, M, 0.f, clamp, st);
  }
#define BWD(T, E) launch_ln<T, E>(false, dy, s, gamma, nullptr, ds, nullptr, const_cast<float*>(mean), const_cast<float*>(rstd), dgamma, dbeta, M, 0.f, clamp, st)
  DFINE_LN_DISPATCH(BWD);
#undef BWD
}
