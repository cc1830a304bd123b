// Fused training BatchNorm2d + optional addend + activation + learnable affine, channel-last.
//
//   z = (x - mean_c) * rstd_c * gamma_c + beta_c  (+ addend)     mean/var over (B, H, W) per channel
//   y = lab_scale * act(z) + lab_bias                            act in {identity, relu, silu}
//
// This is the tail of every conv block on the path: ConvBNAct (bn -> ReLU -> LearnableAffineBlock,
// reference src/d_fine/arch/hgnetv2.py:35-80), ConvNormLayer(_fuse) (norm -> SiLU,
// hybrid_encoder.py:21-93) and VGGBlock (silu(bn(conv3x3) + bn(conv1x1)), hybrid_encoder.py:115-121,
// which is what the addend is for). The stock path spends 5 launches and 4 passes over the map in the
// forward (MIOpen mean/var, final, normalise; activation; addcmul) and ~9 in the backward; here it is
// 2 + 2 kernels, each map read twice per direction. All four kernels are HBM-bound streaming kernels.
//
// x / addend / y / dy / dx / dadd: [n_pix, C] (= [B, H, W, C]) of T (bf16 or f32). Statistics are
// accumulated as fp32 partials per thread and combined with fp64 atomics (the conv outputs of this
// model can have |mean| >> std; fp64 keeps E[x^2] - E[x]^2 exact enough).
#include <stdlib.h>

#include "common.h"

namespace dfine {

enum { ACT_NONE = 0, ACT_RELU = 1, ACT_SILU = 2 };

// The per-channel sums are accumulated in NA_REPLICAS copies (block b adds into copy b % NA_REPLICAS)
// and summed by the consumer: thousands of blocks adding into one set of C addresses serialise.
constexpr int NA_REPLICAS = 8;

struct NaGeom {
  int n_pix, C;
  int pix_per_thread;
  int act, has_add, has_lab, training;
  int ranks;          // SyncBatchNorm: ranks whose AVERAGED sums the workspace holds (1 = this rank's own sums)
  float eps, momentum;
  int ldy;            // forward: pixel pitch of the OUTPUT (>= C: a channel slice of a concat buffer)
  int ld_post;        // forward: pixel pitch of the optional residual added AFTER activation and LAB
  int ld_dy, ld_dy2;  // backward: pixel pitches of the output gradient and of its optional second part (dy + dy2)
  int dbg;            // diagnostics (tools/bench_bn.py, DFINE_BN_DEBUG): 1 = skip the global atomics of the column reduction,
                      // 2 = skip the column reduction altogether
};

// sigmoid through v_exp + v_rcp (1 ulp; an IEEE division is a ~10-instruction sequence per element, and these kernels are
// bound by instruction issue, not by HBM: profiles/r03_bn_kernels.txt)
__device__ __forceinline__ float fast_sigmoid(float z) { return __builtin_amdgcn_rcpf(1.f + __expf(-z)); }
__device__ __forceinline__ float act_fwd(float z, int act) {
  if (act == ACT_RELU) return fmaxf(z, 0.f);
  if (act == ACT_SILU) return z * fast_sigmoid(z);
  return z;
}
__device__ __forceinline__ float act_grad(float z, int act) {
  if (act == ACT_RELU) return z > 0.f ? 1.f : 0.f;
  if (act == ACT_SILU) {
    const float s = fast_sigmoid(z);
    return s * (1.f + z * (1.f - s));
  }
  return 1.f;
}
// Compile-time specialisation of the streaming kernels: FLAGS >= 0 fixes (bit 0 addend, bit 1 second gradient part / POST
// residual, bit 2 LAB, bits 3-4 activation) so the per-element code has no branch and no dead transcendental; FLAGS < 0
// reads them at run time (the rarely used vector widths: compile time).
#define NA_OPTS(FLAGS, g, second_ptr)                                              \
  const bool ADD = (FLAGS) < 0 ? (g).has_add != 0 : (((FLAGS)&1) != 0);            \
  const bool SEC = (FLAGS) < 0 ? (second_ptr) != nullptr : (((FLAGS)&2) != 0);     \
  const bool LAB = (FLAGS) < 0 ? (g).has_lab != 0 : (((FLAGS)&4) != 0);            \
  const int ACTV = (FLAGS) < 0 ? (g).act : (((FLAGS) >> 3) & 3);

// blockDim = (channel groups of VEC, pixel rows). Thread (cx, py) owns channels
// [ (blockIdx.y*blockDim.x + cx)*VEC, +VEC ) and pixels blockIdx.x*blockDim.y*ppt + py + n*blockDim.y.
template <typename T, int NQ>
struct ColReduce {
  // Sum NQ quantities x VEC channels over the block's pixel rows through LDS, then one fp64 atomic
  // per (quantity, channel) per block into out[q*C + c].
  template <int VEC>
  __device__ static __forceinline__ void run(float (&acc)[NQ][VEC], float* red, double* out, int C, int c0, int dbg = 0) {
    if (dbg & 2) {
      if (acc[0][0] == 12345.678f) out[0] = 1.0;  // keeps the accumulation alive
      return;
    }
    const int lanes = blockDim.x * VEC;  // channels covered by the block
    float* mine = red + (size_t)threadIdx.y * NQ * lanes + threadIdx.x * VEC;
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
      for (int i = 0; i < VEC; ++i) mine[q * lanes + i] = acc[q][i];
    __syncthreads();
    const int tid = threadIdx.y * blockDim.x + threadIdx.x, nthreads = blockDim.x * blockDim.y;
    for (int e = tid; e < NQ * lanes; e += nthreads) {
      float s = 0.f;
      for (int r = 0; r < (int)blockDim.y; ++r) s += red[(size_t)r * NQ * lanes + e];
      const int q = e / lanes, c = e - q * lanes;
      if (dbg & 1) {
        if (s == 12345.678f) out[0] = 1.0;
        continue;
      }
      atomicAdd(out + ((size_t)(blockIdx.x % NA_REPLICAS) * NQ + q) * C + (c0 - threadIdx.x * VEC) + c, (double)s);
    }
  }
};

// Every kernel walks its pixels four at a time with the loads of all four issued before any is used
// (the maps are 6-100 MB: at that size a kernel is latency-bound unless each wave keeps several
// 16-byte loads in flight), and fetches its per-channel constants once per block through LDS.
#define DFINE_NA_UNROLL 4
#define DFINE_NA_UNROLL_BWD 2  // x, dy (+addend) per pixel: 2 pixels already keep 4-6 loads in flight

template <typename T, int VEC>
__global__ __launch_bounds__(256) void bn_stats_kernel(const T* __restrict__ x, double* __restrict__ sums, NaGeom g) {
  extern __shared__ __attribute__((aligned(16))) float red[];
  const int c0 = (blockIdx.y * blockDim.x + threadIdx.x) * VEC;
  float acc[2][VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) acc[0][i] = acc[1][i] = 0.f;
  const int first = blockIdx.x * blockDim.y * g.pix_per_thread + threadIdx.y;
  for (int n = 0; n < g.pix_per_thread; n += DFINE_NA_UNROLL) {
    float v[DFINE_NA_UNROLL][VEC];
#pragma unroll
    for (int u = 0; u < DFINE_NA_UNROLL; ++u) {
      const int pix = first + (n + u) * blockDim.y;
      if (n + u < g.pix_per_thread && pix < g.n_pix) {
        load_vec<T, VEC>(x + (size_t)pix * g.C + c0, v[u]);
      } else {
#pragma unroll
        for (int i = 0; i < VEC; ++i) v[u][i] = 0.f;
      }
    }
#pragma unroll
    for (int u = 0; u < DFINE_NA_UNROLL; ++u)
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        acc[0][i] += v[u][i];
        acc[1][i] += v[u][i] * v[u][i];
      }
  }
  ColReduce<T, 2>::template run<VEC>(acc, red, sums, g.C, c0);
}

// Forward apply. One thread per channel of the block's slice turns the fp64 sums (training) or the
// running statistics (eval) into scale/shift in LDS; block row 0 also keeps the books.
template <typename T, int VEC, int FLAGS>
__global__ __launch_bounds__(256) void bn_apply_kernel(
    const T* __restrict__ x, const T* __restrict__ addend, const T* __restrict__ post, T* __restrict__ y,
    const double* __restrict__ sums, const float* __restrict__ gamma, const float* __restrict__ beta,
    float* __restrict__ running_mean, float* __restrict__ running_var, float* __restrict__ save_mean,
    float* __restrict__ save_rstd, const float* __restrict__ lab_scale, const float* __restrict__ lab_bias, NaGeom g) {
  extern __shared__ __attribute__((aligned(16))) float cst[];
  const int lanes = blockDim.x * VEC;
  const int cbase = blockIdx.y * lanes;
  const int tid = threadIdx.y * blockDim.x + threadIdx.x;
  for (int k = tid; k < lanes; k += blockDim.x * blockDim.y) {
    const int c = cbase + k;
    float mean, rstd;
    if (g.training) {
      double s1 = 0, s2 = 0;
#pragma unroll
      for (int rep = 0; rep < NA_REPLICAS; ++rep) {
        s1 += sums[(size_t)rep * 2 * g.C + c];
        s2 += sums[(size_t)rep * 2 * g.C + g.C + c];
      }
      const double m = s1 / g.n_pix;
      double v = s2 / g.n_pix - m * m;
      if (v < 0) v = 0;
      mean = (float)m;
      const float var = (float)v;
      rstd = rsqrtf(var + g.eps);
      if (blockIdx.x == 0) {
        save_mean[c] = mean;
        save_rstd[c] = rstd;
        const float count = (float)g.n_pix * (float)g.ranks;  // pixels behind the statistics, over all ranks
        const float unbiased = count > 1.f ? var * (count / (count - 1.f)) : var;
        running_mean[c] = (1.f - g.momentum) * running_mean[c] + g.momentum * mean;
        running_var[c] = (1.f - g.momentum) * running_var[c] + g.momentum * unbiased;
      }
    } else {
      mean = running_mean[c];
      rstd = rsqrtf(running_var[c] + g.eps);
    }
    const float sc = rstd * gamma[c];
    cst[k] = sc;
    cst[lanes + k] = beta[c] - mean * sc;
  }
  __syncthreads();
  const int c0 = cbase + threadIdx.x * VEC;
  float scale[VEC], shift[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) {
    scale[i] = cst[threadIdx.x * VEC + i];
    shift[i] = cst[lanes + threadIdx.x * VEC + i];
  }
  NA_OPTS(FLAGS, g, post)
  const bool POST = SEC;
  const float ls = LAB ? lab_scale[0] : 1.f, lb = LAB ? lab_bias[0] : 0.f;
  const int first = blockIdx.x * blockDim.y * g.pix_per_thread + threadIdx.y;
  for (int n = 0; n < g.pix_per_thread; n += DFINE_NA_UNROLL) {
    float v[DFINE_NA_UNROLL][VEC], a[DFINE_NA_UNROLL][VEC], r[DFINE_NA_UNROLL][VEC];
    bool ok[DFINE_NA_UNROLL];
#pragma unroll
    for (int u = 0; u < DFINE_NA_UNROLL; ++u) {
      const int pix = first + (n + u) * blockDim.y;
      ok[u] = n + u < g.pix_per_thread && pix < g.n_pix;
      if (ok[u]) {
        const size_t off = (size_t)pix * g.C + c0;
        load_vec<T, VEC>(x + off, v[u]);
        if (ADD) load_vec<T, VEC>(addend + off, a[u]);
        if (POST) load_vec<T, VEC>(post + (size_t)pix * g.ld_post + c0, r[u]);
      }
    }
#pragma unroll
    for (int u = 0; u < DFINE_NA_UNROLL; ++u) {
      if (!ok[u]) continue;
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        float z = v[u][i] * scale[i] + shift[i];
        if (ADD) z += a[u][i];
        v[u][i] = ls * act_fwd(z, ACTV) + lb;
        if (POST) v[u][i] += r[u][i];  // the block's residual: out = lab(act(norm(x))) + residual, one rounding
      }
      store_vec<T, VEC>(y + (size_t)(first + (n + u) * blockDim.y) * g.ldy + c0, v[u]);
    }
  }
}

// Backward pass 1: per channel sum(dz), sum(dz * xhat), sum(dy * act(z)) [-> d lab_scale], sum(dy) [-> d lab_bias]
template <typename T, int VEC, int UNROLL, int FLAGS>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(
    const T* __restrict__ x, const T* __restrict__ addend, const T* __restrict__ dy, const T* __restrict__ dy2,
    const float* __restrict__ save_mean, const float* __restrict__ save_rstd,
    const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ lab_scale,
    double* __restrict__ red_out, NaGeom g) {
  extern __shared__ __attribute__((aligned(16))) float red[];
  const int lanes = blockDim.x * VEC;
  const int cbase = blockIdx.y * lanes;
  const int tid = threadIdx.y * blockDim.x + threadIdx.x;
  for (int k = tid; k < lanes; k += blockDim.x * blockDim.y) {
    const int c = cbase + k;
    red[k] = save_mean[c];
    red[lanes + k] = save_rstd[c];
    red[2 * lanes + k] = gamma[c];
    red[3 * lanes + k] = beta[c];
  }
  __syncthreads();
  const int c0 = cbase + threadIdx.x * VEC;
  float mean[VEC], rstd[VEC], gam[VEC], bet[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) {
    const int k = threadIdx.x * VEC + i;
    mean[i] = red[k];
    rstd[i] = red[lanes + k];
    gam[i] = red[2 * lanes + k];
    bet[i] = red[3 * lanes + k];
  }
  __syncthreads();  // `red` is reused by the column reduction below
  NA_OPTS(FLAGS, g, dy2)
  const bool DY2 = SEC;
  const float ls = LAB ? lab_scale[0] : 1.f;
  float acc[4][VEC];
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int i = 0; i < VEC; ++i) acc[q][i] = 0.f;
  const int first = blockIdx.x * blockDim.y * g.pix_per_thread + threadIdx.y;
  for (int n = 0; n < g.pix_per_thread; n += UNROLL) {
    PackedVec<T, VEC> px[UNROLL], pa[UNROLL], pd[UNROLL], pe[UNROLL];
    bool ok[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const int pix = first + (n + u) * blockDim.y;
      ok[u] = n + u < g.pix_per_thread && pix < g.n_pix;
      if (ok[u]) {
        const size_t off = (size_t)pix * g.C + c0;
        px[u].load(x + off);
        pd[u].load(dy + (size_t)pix * g.ld_dy + c0);
        if (DY2) pe[u].load(dy2 + (size_t)pix * g.ld_dy2 + c0);
        if (ADD) pa[u].load(addend + off);
      }
    }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      if (!ok[u]) continue;
      float v[VEC], a[VEC], d[VEC];
      px[u].unpack(v);
      pd[u].unpack(d);
      if (DY2) {  // the output feeds two consumers (next layer + a concat slice): their gradients are summed on load
        float d2[VEC];
        pe[u].unpack(d2);
#pragma unroll
        for (int i = 0; i < VEC; ++i) d[i] += d2[i];
      }
      if (ADD) pa[u].unpack(a);
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        const float xhat = (v[i] - mean[i]) * rstd[i];
        float z = xhat * gam[i] + bet[i];
        if (ADD) z += a[i];
        const float dz = d[i] * ls * act_grad(z, ACTV);
        acc[0][i] += dz;
        acc[1][i] += dz * xhat;
        if (LAB) {  // d lab_scale / d lab_bias only; without LAB this saves SiLU's second exponential
          acc[2][i] += d[i] * act_fwd(z, ACTV);
          acc[3][i] += d[i];
        }
      }
    }
  }
  ColReduce<T, 4>::template run<VEC>(acc, red, red_out, g.C, c0, g.dbg);
}

// Backward pass 2: dx = gamma*rstd * (dz - mean(dz) - xhat*mean(dz*xhat)); dadd = dz;
// block row 0 also publishes dgamma/dbeta (fp32) and adds its channels' share of d lab.
template <typename T, int VEC, int FLAGS>
__global__ __launch_bounds__(256) void bn_bwd_dx_kernel(
    const T* __restrict__ x, const T* __restrict__ addend, const T* __restrict__ dy, const T* __restrict__ dy2,
    const float* __restrict__ save_mean, const float* __restrict__ save_rstd,
    const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ lab_scale,
    const double* __restrict__ red_in, T* __restrict__ dx, T* __restrict__ dadd,
    float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ dlab_scale, float* __restrict__ dlab_bias,
    NaGeom g) {
  extern __shared__ __attribute__((aligned(16))) float cst[];
  const int lanes = blockDim.x * VEC;
  const int cbase = blockIdx.y * lanes;
  const int tid = threadIdx.y * blockDim.x + threadIdx.x;
  NA_OPTS(FLAGS, g, dy2)
  const bool DY2 = SEC;
  const float inv_n = 1.f / (float)g.n_pix;
  // d lab_scale / d lab_bias are sums over ALL channels: the channels of this block are added up in LDS first (the global
  // adds of a 128..1536-channel map all hit the same two floats and would serialise behind workgroup 0 otherwise)
  __shared__ float lab_part[2];
  if (tid < 2) lab_part[tid] = 0.f;
  __syncthreads();
  for (int k = tid; k < lanes; k += blockDim.x * blockDim.y) {
    const int c = cbase + k;
    double r0 = 0, r1 = 0, r2 = 0, r3 = 0;
#pragma unroll
    for (int rep = 0; rep < NA_REPLICAS; ++rep) {
      const double* rr = red_in + (size_t)rep * 4 * g.C;
      r0 += rr[c]; r1 += rr[g.C + c]; r2 += rr[2 * g.C + c]; r3 += rr[3 * g.C + c];
    }
    const float s_dz = (float)r0, s_dzx = (float)r1;
    cst[k] = save_mean[c];
    cst[lanes + k] = save_rstd[c];
    cst[2 * lanes + k] = gamma[c];
    cst[3 * lanes + k] = beta[c];
    cst[4 * lanes + k] = g.training ? s_dz * inv_n : 0.f;  // eval-mode BN: statistics are constants
    cst[5 * lanes + k] = g.training ? s_dzx * inv_n : 0.f;
    if (blockIdx.x == 0) {  // one writer per channel: accumulate (the caller may point these at .grad)
      dgamma[c] += s_dzx;
      dbeta[c] += s_dz;
      if (LAB) {
        atomicAdd(&lab_part[0], (float)r2);
        atomicAdd(&lab_part[1], (float)r3);
      }
    }
  }
  __syncthreads();
  if (blockIdx.x == 0 && LAB && tid < 2) atomicAdd(tid == 0 ? dlab_scale : dlab_bias, lab_part[tid]);
  const int c0 = cbase + threadIdx.x * VEC;
  float mean[VEC], rstd[VEC], gam[VEC], bet[VEC], m_dz[VEC], m_dzx[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) {
    const int k = threadIdx.x * VEC + i;
    mean[i] = cst[k];
    rstd[i] = cst[lanes + k];
    gam[i] = cst[2 * lanes + k];
    bet[i] = cst[3 * lanes + k];
    m_dz[i] = cst[4 * lanes + k];
    m_dzx[i] = cst[5 * lanes + k];
  }
  const float ls = LAB ? lab_scale[0] : 1.f;
  const int first = blockIdx.x * blockDim.y * g.pix_per_thread + threadIdx.y;
  for (int n = 0; n < g.pix_per_thread; n += DFINE_NA_UNROLL_BWD) {
    PackedVec<T, VEC> px[DFINE_NA_UNROLL_BWD], pa[DFINE_NA_UNROLL_BWD], pd[DFINE_NA_UNROLL_BWD], pe[DFINE_NA_UNROLL_BWD];
    bool ok[DFINE_NA_UNROLL_BWD];
#pragma unroll
    for (int u = 0; u < DFINE_NA_UNROLL_BWD; ++u) {
      const int pix = first + (n + u) * blockDim.y;
      ok[u] = n + u < g.pix_per_thread && pix < g.n_pix;
      if (ok[u]) {
        const size_t off = (size_t)pix * g.C + c0;
        px[u].load(x + off);
        pd[u].load(dy + (size_t)pix * g.ld_dy + c0);
        if (DY2) pe[u].load(dy2 + (size_t)pix * g.ld_dy2 + c0);
        if (ADD) pa[u].load(addend + off);
      }
    }
#pragma unroll
    for (int u = 0; u < DFINE_NA_UNROLL_BWD; ++u) {
      if (!ok[u]) continue;
      const size_t off = (size_t)(first + (n + u) * blockDim.y) * g.C + c0;
      float v[VEC], a[VEC], d[VEC];
      px[u].unpack(v);
      pd[u].unpack(d);
      if (DY2) {
        float d2[VEC];
        pe[u].unpack(d2);
#pragma unroll
        for (int i = 0; i < VEC; ++i) d[i] += d2[i];
      }
      if (ADD) pa[u].unpack(a);
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        const float xhat = (v[i] - mean[i]) * rstd[i];
        float z = xhat * gam[i] + bet[i];
        if (ADD) z += a[i];
        const float dz = d[i] * ls * act_grad(z, ACTV);
        a[i] = dz;
        v[i] = gam[i] * rstd[i] * (dz - m_dz[i] - xhat * m_dzx[i]);
      }
      store_vec<T, VEC>(dx + off, v);
      if (ADD) store_vec<T, VEC>(dadd + off, a);
    }
  }
}


// ---------------------------------------------------------------------------------------------------------------
// Two norms summed under one activation:  y = act( bn_a(xa) + bn_b(xb) )  -- RepVgg's training form
// silu(norm(conv3x3(x)) + norm(conv1x1(x))) (reference hybrid_encoder.py:106-121). As two single-norm calls the
// second branch's output is written, re-read as the first's addend, and in backward the shared dz travels through
// memory as `dadd`: 4 + 4 launches and 13 map passes backward. Here: 1 launch forward (xa, xb -> y), 2 backward
// (reduce: xa, xb, dy -> 3 sums per channel; dx: xa, xb, dy -> dxa, dxb), 8 passes.
// ---------------------------------------------------------------------------------------------------------------
template <typename T, int VEC, int ACT>
__global__ __launch_bounds__(256) void bn_pair_apply_kernel(
    const T* __restrict__ xa, const T* __restrict__ xb, T* __restrict__ y, const double* __restrict__ sums_a,
    const double* __restrict__ sums_b, const float* __restrict__ gamma_a, const float* __restrict__ beta_a,
    const float* __restrict__ gamma_b, const float* __restrict__ beta_b, float* __restrict__ rm_a, float* __restrict__ rv_a,
    float* __restrict__ rm_b, float* __restrict__ rv_b, float* __restrict__ save_a, float* __restrict__ save_b, NaGeom g) {
  extern __shared__ __attribute__((aligned(16))) float cst[];  // [3][lanes]: scale_a, scale_b, shift_a + shift_b
  const int lanes = blockDim.x * VEC;
  const int cbase = blockIdx.y * lanes;
  const int tid = threadIdx.y * blockDim.x + threadIdx.x;
  for (int k = tid; k < lanes; k += blockDim.x * blockDim.y) {
    const int c = cbase + k;
    float sc[2], sh[2];
#pragma unroll
    for (int w = 0; w < 2; ++w) {
      const double* sums = w ? sums_b : sums_a;
      float* rm = w ? rm_b : rm_a;
      float* rv = w ? rv_b : rv_a;
      float* save = w ? save_b : save_a;  // [2][C]: mean, rstd
      float mean, rstd;
      if (g.training) {
        double s1 = 0, s2 = 0;
#pragma unroll
        for (int rep = 0; rep < NA_REPLICAS; ++rep) {
          s1 += sums[(size_t)rep * 2 * g.C + c];
          s2 += sums[(size_t)rep * 2 * g.C + g.C + c];
        }
        const double m = s1 / g.n_pix;
        double v = s2 / g.n_pix - m * m;
        if (v < 0) v = 0;
        mean = (float)m;
        const float var = (float)v;
        rstd = rsqrtf(var + g.eps);
        if (blockIdx.x == 0) {
          save[c] = mean;
          save[g.C + c] = rstd;
          const float count = (float)g.n_pix * (float)g.ranks;
          const float unbiased = count > 1.f ? var * (count / (count - 1.f)) : var;
          rm[c] = (1.f - g.momentum) * rm[c] + g.momentum * mean;
          rv[c] = (1.f - g.momentum) * rv[c] + g.momentum * unbiased;
        }
      } else {
        mean = rm[c];
        rstd = rsqrtf(rv[c] + g.eps);
      }
      sc[w] = rstd * (w ? gamma_b : gamma_a)[c];
      sh[w] = (w ? beta_b : beta_a)[c] - mean * sc[w];
    }
    cst[k] = sc[0];
    cst[lanes + k] = sc[1];
    cst[2 * lanes + k] = sh[0] + sh[1];
  }
  __syncthreads();
  const int c0 = cbase + threadIdx.x * VEC;
  float sa[VEC], sb[VEC], sh[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) {
    sa[i] = cst[threadIdx.x * VEC + i];
    sb[i] = cst[lanes + threadIdx.x * VEC + i];
    sh[i] = cst[2 * lanes + threadIdx.x * VEC + i];
  }
  const int first = blockIdx.x * blockDim.y * g.pix_per_thread + threadIdx.y;
  for (int n = 0; n < g.pix_per_thread; n += DFINE_NA_UNROLL) {
    PackedVec<T, VEC> pa[DFINE_NA_UNROLL], pb[DFINE_NA_UNROLL];
    bool ok[DFINE_NA_UNROLL];
#pragma unroll
    for (int u = 0; u < DFINE_NA_UNROLL; ++u) {
      const int pix = first + (n + u) * blockDim.y;
      ok[u] = n + u < g.pix_per_thread && pix < g.n_pix;
      if (ok[u]) {
        pa[u].load(xa + (size_t)pix * g.C + c0);
        pb[u].load(xb + (size_t)pix * g.C + c0);
      }
    }
#pragma unroll
    for (int u = 0; u < DFINE_NA_UNROLL; ++u) {
      if (!ok[u]) continue;
      float va[VEC], vb[VEC];
      pa[u].unpack(va);
      pb[u].unpack(vb);
#pragma unroll
      for (int i = 0; i < VEC; ++i) va[i] = act_fwd(va[i] * sa[i] + vb[i] * sb[i] + sh[i], ACT < 0 ? g.act : ACT);
      store_vec<T, VEC>(y + (size_t)(first + (n + u) * blockDim.y) * g.ldy + c0, va);
    }
  }
}

// backward pass 1 of the pair: per channel sum(dz), sum(dz * xhat_a), sum(dz * xhat_b)
template <typename T, int VEC, int ACT>
__global__ __launch_bounds__(256) void bn_pair_bwd_reduce_kernel(
    const T* __restrict__ xa, const T* __restrict__ xb, const T* __restrict__ dy, const T* __restrict__ dy2,
    const float* __restrict__ save_a, const float* __restrict__ save_b, const float* __restrict__ gamma_a,
    const float* __restrict__ beta_a, const float* __restrict__ gamma_b, const float* __restrict__ beta_b,
    double* __restrict__ red_out, NaGeom g) {
  extern __shared__ __attribute__((aligned(16))) float red[];
  const int lanes = blockDim.x * VEC;
  const int cbase = blockIdx.y * lanes;
  const int tid = threadIdx.y * blockDim.x + threadIdx.x;
  for (int k = tid; k < lanes; k += blockDim.x * blockDim.y) {
    const int c = cbase + k;
    red[k] = save_a[c];
    red[lanes + k] = save_a[g.C + c];
    red[2 * lanes + k] = save_b[c];
    red[3 * lanes + k] = save_b[g.C + c];
    red[4 * lanes + k] = gamma_a[c];
    red[5 * lanes + k] = gamma_b[c];
    red[6 * lanes + k] = beta_a[c] + beta_b[c];
  }
  __syncthreads();
  float mean_a[VEC], rstd_a[VEC], mean_b[VEC], rstd_b[VEC], gam_a[VEC], gam_b[VEC], bet[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) {
    const int k = threadIdx.x * VEC + i;
    mean_a[i] = red[k], rstd_a[i] = red[lanes + k], mean_b[i] = red[2 * lanes + k], rstd_b[i] = red[3 * lanes + k];
    gam_a[i] = red[4 * lanes + k], gam_b[i] = red[5 * lanes + k], bet[i] = red[6 * lanes + k];
  }
  __syncthreads();  // `red` is reused by the column reduction below
  const int c0 = cbase + threadIdx.x * VEC;
  float acc[3][VEC];
#pragma unroll
  for (int q = 0; q < 3; ++q)
#pragma unroll
    for (int i = 0; i < VEC; ++i) acc[q][i] = 0.f;
  const int first = blockIdx.x * blockDim.y * g.pix_per_thread + threadIdx.y;
  for (int n = 0; n < g.pix_per_thread; n += DFINE_NA_UNROLL_BWD) {
    PackedVec<T, VEC> pa[DFINE_NA_UNROLL_BWD], pb[DFINE_NA_UNROLL_BWD], pd[DFINE_NA_UNROLL_BWD], pe[DFINE_NA_UNROLL_BWD];
    bool ok[DFINE_NA_UNROLL_BWD];
#pragma unroll
    for (int u = 0; u < DFINE_NA_UNROLL_BWD; ++u) {
      const int pix = first + (n + u) * blockDim.y;
      ok[u] = n + u < g.pix_per_thread && pix < g.n_pix;
      if (ok[u]) {
        pa[u].load(xa + (size_t)pix * g.C + c0);
        pb[u].load(xb + (size_t)pix * g.C + c0);
        pd[u].load(dy + (size_t)pix * g.ld_dy + c0);
        if (dy2) pe[u].load(dy2 + (size_t)pix * g.ld_dy2 + c0);
      }
    }
#pragma unroll
    for (int u = 0; u < DFINE_NA_UNROLL_BWD; ++u) {
      if (!ok[u]) continue;
      float va[VEC], vb[VEC], d[VEC];
      pa[u].unpack(va);
      pb[u].unpack(vb);
      pd[u].unpack(d);
      if (dy2) {
        float d2[VEC];
        pe[u].unpack(d2);
#pragma unroll
        for (int i = 0; i < VEC; ++i) d[i] += d2[i];
      }
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        const float ha = (va[i] - mean_a[i]) * rstd_a[i], hb = (vb[i] - mean_b[i]) * rstd_b[i];
        const float dz = d[i] * act_grad(ha * gam_a[i] + hb * gam_b[i] + bet[i], ACT < 0 ? g.act : ACT);
        acc[0][i] += dz;
        acc[1][i] += dz * ha;
        acc[2][i] += dz * hb;
      }
    }
  }
  ColReduce<T, 3>::template run<VEC>(acc, red, red_out, g.C, c0);
}

// backward pass 2 of the pair: dxa, dxb; block row 0 publishes dgamma / dbeta of both norms
template <typename T, int VEC, int ACT>
__global__ __launch_bounds__(256) void bn_pair_bwd_dx_kernel(
    const T* __restrict__ xa, const T* __restrict__ xb, const T* __restrict__ dy, const T* __restrict__ dy2,
    const float* __restrict__ save_a, const float* __restrict__ save_b, const float* __restrict__ gamma_a,
    const float* __restrict__ beta_a, const float* __restrict__ gamma_b, const float* __restrict__ beta_b,
    const double* __restrict__ red_in, T* __restrict__ dxa, T* __restrict__ dxb, float* __restrict__ dgamma_a,
    float* __restrict__ dbeta_a, float* __restrict__ dgamma_b, float* __restrict__ dbeta_b, NaGeom g) {
  extern __shared__ __attribute__((aligned(16))) float cst[];  // [10][lanes]
  const int lanes = blockDim.x * VEC;
  const int cbase = blockIdx.y * lanes;
  const int tid = threadIdx.y * blockDim.x + threadIdx.x;
  const float inv_n = 1.f / (float)g.n_pix;
  for (int k = tid; k < lanes; k += blockDim.x * blockDim.y) {
    const int c = cbase + k;
    double r0 = 0, r1 = 0, r2 = 0;
#pragma unroll
    for (int rep = 0; rep < NA_REPLICAS; ++rep) {
      const double* rr = red_in + (size_t)rep * 3 * g.C;
      r0 += rr[c]; r1 += rr[g.C + c]; r2 += rr[2 * g.C + c];
    }
    cst[k] = save_a[c];
    cst[lanes + k] = save_a[g.C + c];
    cst[2 * lanes + k] = save_b[c];
    cst[3 * lanes + k] = save_b[g.C + c];
    cst[4 * lanes + k] = gamma_a[c];
    cst[5 * lanes + k] = gamma_b[c];
    cst[6 * lanes + k] = beta_a[c] + beta_b[c];
    cst[7 * lanes + k] = g.training ? (float)r0 * inv_n : 0.f;
    cst[8 * lanes + k] = g.training ? (float)r1 * inv_n : 0.f;
    cst[9 * lanes + k] = g.training ? (float)r2 * inv_n : 0.f;
    if (blockIdx.x == 0) {  // one writer per channel: accumulate (the caller may point these at .grad)
      dgamma_a[c] += (float)r1;
      dgamma_b[c] += (float)r2;
      dbeta_a[c] += (float)r0;
      dbeta_b[c] += (float)r0;
    }
  }
  __syncthreads();
  float mean_a[VEC], rstd_a[VEC], mean_b[VEC], rstd_b[VEC], gam_a[VEC], gam_b[VEC], bet[VEC], m_dz[VEC], m_a[VEC], m_b[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) {
    const int k = threadIdx.x * VEC + i;
    mean_a[i] = cst[k], rstd_a[i] = cst[lanes + k], mean_b[i] = cst[2 * lanes + k], rstd_b[i] = cst[3 * lanes + k];
    gam_a[i] = cst[4 * lanes + k], gam_b[i] = cst[5 * lanes + k], bet[i] = cst[6 * lanes + k];
    m_dz[i] = cst[7 * lanes + k], m_a[i] = cst[8 * lanes + k], m_b[i] = cst[9 * lanes + k];
  }
  const int c0 = cbase + threadIdx.x * VEC;
  const int first = blockIdx.x * blockDim.y * g.pix_per_thread + threadIdx.y;
  for (int n = 0; n < g.pix_per_thread; n += DFINE_NA_UNROLL_BWD) {
    PackedVec<T, VEC> pa[DFINE_NA_UNROLL_BWD], pb[DFINE_NA_UNROLL_BWD], pd[DFINE_NA_UNROLL_BWD], pe[DFINE_NA_UNROLL_BWD];
    bool ok[DFINE_NA_UNROLL_BWD];
#pragma unroll
    for (int u = 0; u < DFINE_NA_UNROLL_BWD; ++u) {
      const int pix = first + (n + u) * blockDim.y;
      ok[u] = n + u < g.pix_per_thread && pix < g.n_pix;
      if (ok[u]) {
        pa[u].load(xa + (size_t)pix * g.C + c0);
        pb[u].load(xb + (size_t)pix * g.C + c0);
        pd[u].load(dy + (size_t)pix * g.ld_dy + c0);
        if (dy2) pe[u].load(dy2 + (size_t)pix * g.ld_dy2 + c0);
      }
    }
#pragma unroll
    for (int u = 0; u < DFINE_NA_UNROLL_BWD; ++u) {
      if (!ok[u]) continue;
      const size_t off = (size_t)(first + (n + u) * blockDim.y) * g.C + c0;
      float va[VEC], vb[VEC], d[VEC];
      pa[u].unpack(va);
      pb[u].unpack(vb);
      pd[u].unpack(d);
      if (dy2) {
        float d2[VEC];
        pe[u].unpack(d2);
#pragma unroll
        for (int i = 0; i < VEC; ++i) d[i] += d2[i];
      }
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        const float ha = (va[i] - mean_a[i]) * rstd_a[i], hb = (vb[i] - mean_b[i]) * rstd_b[i];
        const float dz = d[i] * act_grad(ha * gam_a[i] + hb * gam_b[i] + bet[i], ACT < 0 ? g.act : ACT);
        va[i] = gam_a[i] * rstd_a[i] * (dz - m_dz[i] - ha * m_a[i]);
        vb[i] = gam_b[i] * rstd_b[i] * (dz - m_dz[i] - hb * m_b[i]);
      }
      store_vec<T, VEC>(dxa + off, va);
      store_vec<T, VEC>(dxb + off, vb);
    }
  }
}

static int apply_target_blocks() {  // read per call: tools/bench_bn.py switches the knobs at run time
  const char* e = getenv("DFINE_BN_APPLY_BLOCKS");
  const int v = e ? atoi(e) : 1024;
  return v < 64 ? 64 : v;
}

static int reduce_target_blocks() {
  const char* e = getenv("DFINE_BN_REDUCE_BLOCKS");
  const int v = e ? atoi(e) : 512;
  return v < 64 ? 64 : v;
}

// 24 specialisations (8 flag combinations x 3 activations) for the 16-byte bf16 vector that carries the model's maps; the
// narrow vectors (channel counts that are not multiples of 8) and fp32 (parity runs) share one run-time-flag instantiation
// each (a code object with ~600 kernels failed to resolve its later symbols at launch: keep the count moderate)
#define NA_CASE(L, F) case F: L(F); break;
#define NA_FLAG_SWITCH(flags, L)                                                                                       \
  if constexpr (sizeof(T) == 2 && VEC == 8) {                                                                          \
    switch (flags) {                                                                                                   \
      NA_CASE(L, 0) NA_CASE(L, 1) NA_CASE(L, 2) NA_CASE(L, 3) NA_CASE(L, 4) NA_CASE(L, 5) NA_CASE(L, 6) NA_CASE(L, 7)    \
      NA_CASE(L, 8) NA_CASE(L, 9) NA_CASE(L, 10) NA_CASE(L, 11) NA_CASE(L, 12) NA_CASE(L, 13) NA_CASE(L, 14) NA_CASE(L, 15) \
      NA_CASE(L, 16) NA_CASE(L, 17) NA_CASE(L, 18) NA_CASE(L, 19) NA_CASE(L, 20) NA_CASE(L, 21) NA_CASE(L, 22)           \
      default: L(23); break;                                                                                           \
    }                                                                                                                  \
  } else {                                                                                                             \
    L(-1);                                                                                                             \
  }

struct NaLaunch {
  dim3 grid, block;
  int ppt;
};

template <int VEC>
static int plan(NaLaunch* L, int n_pix, int C, int target_blocks) {
  if (C % VEC != 0) return DFINE_E_UNSUPPORTED;
  int cgb = 1;
  for (int d = 32; d >= 1; --d)
    if ((C / VEC) % d == 0) { cgb = d; break; }
  const int pb = 256 / cgb;
  const int slices = C / (cgb * VEC);
  long long want = (long long)target_blocks / slices;
  if (want < 1) want = 1;
  long long ppt = ((long long)n_pix + want * pb - 1) / (want * pb);
  if (ppt < 1) ppt = 1;
  L->ppt = (int)ppt;
  L->block = dim3(cgb, pb);
  L->grid = dim3((unsigned)(((long long)n_pix + (long long)pb * ppt - 1) / ((long long)pb * ppt)), slices);
  return 0;
}

static NaGeom make_geom(int n_pix, int C, int ppt, int act, bool has_add, bool has_lab, int training, float eps, float momentum) {
  NaGeom g;
  g.n_pix = n_pix; g.C = C; g.pix_per_thread = ppt; g.act = act; g.has_add = has_add; g.has_lab = has_lab;
  g.training = training & 0xff; g.ranks = (training >> 8) > 1 ? (training >> 8) : 1; g.eps = eps; g.momentum = momentum;
  g.ldy = g.ld_dy = g.ld_dy2 = g.ld_post = C;
  const char* e = getenv("DFINE_BN_DEBUG");
  g.dbg = e ? atoi(e) : 0;
  return g;
}

template <typename T, int VEC>
static int fwd_impl(const void* x, const void* addend, void* y, double* sums, const float* gamma, const float* beta,
                    float* running_mean, float* running_var, float* save_mean, float* save_rstd, const float* lab_scale,
                    const float* lab_bias, int n_pix, int C, int act, int training, float eps, float momentum, int phase,
                    int ldy, const void* post, int ld_post, hipStream_t st) {
  NaLaunch R, A;
  if (int rc = plan<VEC>(&R, n_pix, C, reduce_target_blocks())) return rc;   // reductions: few blocks -> few contended atomics
  if (int rc = plan<VEC>(&A, n_pix, C, apply_target_blocks())) return rc;  // streaming: ~4 blocks per CU, several pixels per thread
  if (training && phase != 2) {
    NaGeom g = make_geom(n_pix, C, R.ppt, act, addend != nullptr, lab_scale != nullptr, training, eps, momentum);
    const size_t lds = (size_t)R.block.y * 2 * R.block.x * VEC * sizeof(float);
    hipLaunchKernelGGL((bn_stats_kernel<T, VEC>), R.grid, R.block, lds, st, (const T*)x, sums, g);
    DFINE_LAUNCH_CHECK();
  }
  if (phase == 1) return 0;
  NaGeom g = make_geom(n_pix, C, A.ppt, act, addend != nullptr, lab_scale != nullptr, training, eps, momentum);
  g.ldy = ldy;
  g.ld_post = ld_post;
  const int flags = (addend ? 1 : 0) | (post ? 2 : 0) | (lab_scale ? 4 : 0) | (act << 3);
#define APPLY_LAUNCH(F)                                                                                                              \
  hipLaunchKernelGGL((bn_apply_kernel<T, VEC, F>), A.grid, A.block, (size_t)2 * A.block.x * VEC * sizeof(float), st, (const T*)x,     \
                     (const T*)addend, (const T*)post, (T*)y, sums, gamma, beta, running_mean, running_var, save_mean, save_rstd,      \
                     lab_scale, lab_bias, g)
  NA_FLAG_SWITCH(flags, APPLY_LAUNCH)
#undef APPLY_LAUNCH
  DFINE_LAUNCH_CHECK();
  return 0;
}

template <typename T, int VEC>
static int bwd_impl(const void* x, const void* addend, const void* dy, int ld_dy, const void* dy2, int ld_dy2, const float* save_mean, const float* save_rstd,
                    const float* gamma, const float* beta, const float* lab_scale, double* red, void* dx, void* dadd,
                    float* dgamma, float* dbeta, float* dlab_scale, float* dlab_bias, int n_pix, int C, int act, int training,
                    float eps, int phase, hipStream_t st) {
  NaLaunch R, A;
  if (int rc = plan<VEC>(&R, n_pix, C, reduce_target_blocks())) return rc;
  if (int rc = plan<VEC>(&A, n_pix, C, apply_target_blocks())) return rc;
  NaGeom g = make_geom(n_pix, C, R.ppt, act, addend != nullptr, lab_scale != nullptr, training, eps, 0.f);
  g.ld_dy = ld_dy, g.ld_dy2 = ld_dy2;
  const int flags = (addend ? 1 : 0) | (dy2 ? 2 : 0) | (lab_scale ? 4 : 0) | (act << 3);
  if (phase != 2) {
    const size_t lds = (size_t)R.block.y * 4 * R.block.x * VEC * sizeof(float);
#define RED_LAUNCH(F)                                                                                                            \
  hipLaunchKernelGGL((bn_bwd_reduce_kernel<T, VEC, 2, F>), R.grid, R.block, lds, st, (const T*)x, (const T*)addend, (const T*)dy, \
                     (const T*)dy2, save_mean, save_rstd, gamma, beta, lab_scale, red, g)
    NA_FLAG_SWITCH(flags, RED_LAUNCH)
#undef RED_LAUNCH
    DFINE_LAUNCH_CHECK();
  }
  if (phase == 1) return 0;
  g.pix_per_thread = A.ppt;
#define DX_LAUNCH(F)                                                                                                                  \
  hipLaunchKernelGGL((bn_bwd_dx_kernel<T, VEC, F>), A.grid, A.block, (size_t)6 * A.block.x * VEC * sizeof(float), st, (const T*)x,     \
                     (const T*)addend, (const T*)dy, (const T*)dy2, save_mean, save_rstd, gamma, beta, lab_scale, red, (T*)dx, (T*)dadd, \
                     dgamma, dbeta, dlab_scale, dlab_bias, g)
  NA_FLAG_SWITCH(flags, DX_LAUNCH)
#undef DX_LAUNCH
  DFINE_LAUNCH_CHECK();
  return 0;
}

template <typename T, int VEC>
static int pair_fwd_impl(const void* xa, const void* xb, void* y, const double* sums_a, const double* sums_b, const float* gamma_a,
                         const float* beta_a, const float* gamma_b, const float* beta_b, float* rm_a, float* rv_a, float* rm_b,
                         float* rv_b, float* save_a, float* save_b, int n_pix, int C, int act, int training, float eps,
                         float momentum, int ldy, hipStream_t st) {
  NaLaunch A;
  if (int rc = plan<VEC>(&A, n_pix, C, apply_target_blocks())) return rc;
  NaGeom g = make_geom(n_pix, C, A.ppt, act, false, false, training, eps, momentum);
  g.ldy = ldy;
#define PAIR_APPLY(ACT_)                                                                                                               \
  hipLaunchKernelGGL((bn_pair_apply_kernel<T, VEC, ACT_>), A.grid, A.block, (size_t)3 * A.block.x * VEC * sizeof(float), st, (const T*)xa, \
                     (const T*)xb, (T*)y, sums_a, sums_b, gamma_a, beta_a, gamma_b, beta_b, rm_a, rv_a, rm_b, rv_b, save_a, save_b, g)
  if (act == ACT_SILU) PAIR_APPLY(ACT_SILU); else PAIR_APPLY(-1);
#undef PAIR_APPLY
  DFINE_LAUNCH_CHECK();
  return 0;
}

template <typename T, int VEC>
static int pair_bwd_impl(const void* xa, const void* xb, const void* dy, int ld_dy, const void* dy2, int ld_dy2, const float* save_a,
                         const float* save_b, const float* gamma_a, const float* beta_a, const float* gamma_b, const float* beta_b,
                         double* red, void* dxa, void* dxb, float* dgamma_a, float* dbeta_a, float* dgamma_b, float* dbeta_b,
                         int n_pix, int C, int act, int training, float eps, hipStream_t st) {
  NaLaunch R, A;
  if (int rc = plan<VEC>(&R, n_pix, C, reduce_target_blocks())) return rc;
  if (int rc = plan<VEC>(&A, n_pix, C, apply_target_blocks())) return rc;
  NaGeom g = make_geom(n_pix, C, R.ppt, act, false, false, training, eps, 0.f);
  g.ld_dy = ld_dy, g.ld_dy2 = ld_dy2;
  const size_t lds_r = (size_t)R.block.x * VEC * sizeof(float) * (R.block.y * 3 > 7 ? R.block.y * 3 : 7);
#define PAIR_RED(ACT_)                                                                                                        \
  hipLaunchKernelGGL((bn_pair_bwd_reduce_kernel<T, VEC, ACT_>), R.grid, R.block, lds_r, st, (const T*)xa, (const T*)xb, (const T*)dy, \
                     (const T*)dy2, save_a, save_b, gamma_a, beta_a, gamma_b, beta_b, red, g)
  if (act == ACT_SILU) PAIR_RED(ACT_SILU); else PAIR_RED(-1);
#undef PAIR_RED
  DFINE_LAUNCH_CHECK();
  g.pix_per_thread = A.ppt;
#define PAIR_DX(ACT_)                                                                                                                   \
  hipLaunchKernelGGL((bn_pair_bwd_dx_kernel<T, VEC, ACT_>), A.grid, A.block, (size_t)10 * A.block.x * VEC * sizeof(float), st, (const T*)xa, \
                     (const T*)xb, (const T*)dy, (const T*)dy2, save_a, save_b, gamma_a, beta_a, gamma_b, beta_b, red, (T*)dxa, (T*)dxb,   \
                     dgamma_a, dbeta_a, dgamma_b, dbeta_b, g)
  if (act == ACT_SILU) PAIR_DX(ACT_SILU); else PAIR_DX(-1);
#undef PAIR_DX
  DFINE_LAUNCH_CHECK();
  return 0;
}

// widest channel vector (<= 16 bytes) that divides C: any channel count works, wide ones stream
#define DFINE_NA_DISPATCH(CALL)                      \
  do {                                               \
    if (dtype == DFINE_DT_BF16) {                    \
      if (C % 8 == 0) return CALL(bf16_t, 8);        \
      if (C % 4 == 0) return CALL(bf16_t, 4);        \
      if (C % 2 == 0) return CALL(bf16_t, 2);        \
      return CALL(bf16_t, 1);                        \
    }                                                \
    if (dtype == DFINE_DT_F32) {                     \
      if (C % 4 == 0) return CALL(float, 4);         \
      if (C % 2 == 0) return CALL(float, 2);         \
      return CALL(float, 1);                         \
    }                                                \
    return DFINE_E_UNSUPPORTED;                      \
  } while (0)

}  // namespace dfine

using namespace dfine;

extern "C" int dfine_bn_act_fwd(const void* x, const void* addend, void* y, int dtype, double* sums,
                                const float* gamma, const float* beta, float* running_mean,
                                float* running_var, float* save_mean, float* save_rstd,
                                const float* lab_scale, const float* lab_bias, int n_pix, int C, int act, int training, float eps,
                                float momentum, int phase, int ldy, const void* post, int ld_post, void* stream) {
  if ((training & 0xff) == 0) training = 0;  // bits 8..: SyncBatchNorm rank count, meaningful with batch statistics only
  if (!x || !y || !gamma || !beta || !running_mean || !running_var) return DFINE_E_BADARG;
  if (training && (!sums || !save_mean || !save_rstd)) return DFINE_E_BADARG;
  if (n_pix <= 0 || C <= 0 || act < 0 || act > 2 || phase < 0 || phase > 2) return DFINE_E_BADARG;
  if ((lab_scale != nullptr) != (lab_bias != nullptr)) return DFINE_E_BADARG;
  if (ldy == 0) ldy = C;
  if (ldy < C) return DFINE_E_BADARG;
  if (ld_post == 0) ld_post = C;
  if (post && ld_post < C) return DFINE_E_BADARG;
  hipStream_t st = (hipStream_t)stream;
  const int es = dtype == DFINE_DT_BF16 ? 2 : 4;
  // a pitched output must keep the channel vectors aligned: pitch and base a multiple of the vector width
#define FWD(T, V)                                                                                                                  \
  ((ldy % V != 0 || (uintptr_t)y % (V * es) != 0 || (post && (ld_post % V != 0 || (uintptr_t)post % (V * es) != 0)))                                                                                  \
       ? DFINE_E_UNSUPPORTED                                                                                                       \
       : fwd_impl<T, V>(x, addend, y, sums, gamma, beta, running_mean, running_var, save_mean, save_rstd, lab_scale, lab_bias, n_pix, C, \
                        act, training, eps, momentum, phase, ldy, post, ld_post, st))
  DFINE_NA_DISPATCH(FWD);
#undef FWD
}

extern "C" int dfine_bn_act_bwd(const void* x, const void* addend, const void* dy, int ld_dy, const void* dy2, int ld_dy2, int dtype,
                                const float* save_mean, const float* save_rstd, const float* gamma,
                                const float* beta, const float* lab_scale, double* red, void* dx, void* dadd,
                                float* dgamma, float* dbeta, float* dlab_scale, float* dlab_bias, int n_pix,
                                int C, int act, int training, float eps, int phase, void* stream) {
  if ((training & 0xff) == 0) training = 0;  // bits 8..: SyncBatchNorm rank count, meaningful with batch statistics only
  if (!x || !dy || !save_mean || !save_rstd || !gamma || !beta || !red || !dx || !dgamma || !dbeta) return DFINE_E_BADARG;
  if ((addend != nullptr) != (dadd != nullptr)) return DFINE_E_BADARG;
  if (lab_scale && (!dlab_scale || !dlab_bias)) return DFINE_E_BADARG;
  if (n_pix <= 0 || C <= 0 || act < 0 || act > 2 || phase < 0 || phase > 2) return DFINE_E_BADARG;
  if (ld_dy == 0) ld_dy = C;
  if (ld_dy2 == 0) ld_dy2 = C;
  if (ld_dy < C || (dy2 && ld_dy2 < C)) return DFINE_E_BADARG;
  hipStream_t st = (hipStream_t)stream;
  const int es = dtype == DFINE_DT_BF16 ? 2 : 4;
#define BWD(T, V)                                                                                                                   \
  ((ld_dy % V != 0 || (uintptr_t)dy % (V * es) != 0 || (dy2 && (ld_dy2 % V != 0 || (uintptr_t)dy2 % (V * es) != 0)))                \
       ? DFINE_E_UNSUPPORTED                                                                                                        \
       : bwd_impl<T, V>(x, addend, dy, ld_dy, dy2, ld_dy2, save_mean, save_rstd, gamma, beta, lab_scale, red, dx, dadd, dgamma, dbeta, \
                        dlab_scale, dlab_bias, n_pix, C, act, training, eps, phase, st))
  DFINE_NA_DISPATCH(BWD);
#undef BWD
}

extern "C" int dfine_bn_pair_act_fwd(const void* xa, const void* xb, void* y, int dtype, const double* sums_a, const double* sums_b,
                                     const float* gamma_a, const float* beta_a, const float* gamma_b, const float* beta_b,
                                     float* running_mean_a, float* running_var_a, float* running_mean_b, float* running_var_b,
                                     float* save_a, float* save_b, int n_pix, int C, int act, int training, float eps, float momentum,
                                     int ldy, void* stream) {
  if ((training & 0xff) == 0) training = 0;  // bits 8..: SyncBatchNorm rank count, meaningful with batch statistics only
  if (!xa || !xb || !y || !gamma_a || !beta_a || !gamma_b || !beta_b || !running_mean_a || !running_var_a || !running_mean_b ||
      !running_var_b)
    return DFINE_E_BADARG;
  if (training && (!sums_a || !sums_b || !save_a || !save_b)) return DFINE_E_BADARG;
  if (n_pix <= 0 || C <= 0 || act < 0 || act > 2) return DFINE_E_BADARG;
  if (ldy == 0) ldy = C;
  if (ldy < C) return DFINE_E_BADARG;
  hipStream_t st = (hipStream_t)stream;
  const int es = dtype == DFINE_DT_BF16 ? 2 : 4;
#define PFWD(T, V)                                                                                                              \
  ((ldy % V != 0 || (uintptr_t)y % (V * es) != 0)                                                                               \
       ? DFINE_E_UNSUPPORTED                                                                                                    \
       : pair_fwd_impl<T, V>(xa, xb, y, sums_a, sums_b, gamma_a, beta_a, gamma_b, beta_b, running_mean_a, running_var_a,          \
                             running_mean_b, running_var_b, save_a, save_b, n_pix, C, act, training, eps, momentum, ldy, st))
  DFINE_NA_DISPATCH(PFWD);
#undef PFWD
}

extern "C" int dfine_bn_pair_act_bwd(const void* xa, const void* xb, const void* dy, int ld_dy, const void* dy2, int ld_dy2, int dtype,
                                     const float* save_a, const float* save_b, const float* gamma_a, const float* beta_a,
                                     const float* gamma_b, const float* beta_b, double* red, void* dxa, void* dxb, float* dgamma_a,
                                     float* dbeta_a, float* dgamma_b, float* dbeta_b, int n_pix, int C, int act, int training,
                                     float eps, void* stream) {
  if ((training & 0xff) == 0) training = 0;  // bits 8..: SyncBatchNorm rank count, meaningful with batch statistics only
  if (!xa || !xb || !dy || !save_a || !save_b || !gamma_a || !beta_a || !gamma_b || !beta_b || !red || !dxa || !dxb || !dgamma_a ||
      !dbeta_a || !dgamma_b || !dbeta_b)
    return DFINE_E_BADARG;
  if (n_pix <= 0 || C <= 0 || act < 0 || act > 2) return DFINE_E_BADARG;
  if (ld_dy == 0) ld_dy = C;
  if (ld_dy2 == 0) ld_dy2 = C;
  if (ld_dy < C || (dy2 && ld_dy2 < C)) return DFINE_E_BADARG;
  hipStream_t st = (hipStream_t)stream;
  const int es = dtype == DFINE_DT_BF16 ? 2 : 4;
#define PBWD(T, V)                                                                                                                  \
  ((ld_dy % V != 0 || (uintptr_t)dy % (V * es) != 0 || (dy2 && (ld_dy2 % V != 0 || (uintptr_t)dy2 % (V * es) != 0)))                \
       ? DFINE_E_UNSUPPORTED                                                                                                        \
       : pair_bwd_impl<T, V>(xa, xb, dy, ld_dy, dy2, ld_dy2, save_a, save_b, gamma_a, beta_a, gamma_b, beta_b, red, dxa, dxb, dgamma_a, \
                             dbeta_a, dgamma_b, dbeta_b, n_pix, C, act, training, eps, st))
  DFINE_NA_DISPATCH(PBWD);
#undef PBWD
}
