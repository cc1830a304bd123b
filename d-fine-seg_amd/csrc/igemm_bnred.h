// Backward reductions of the BatchNorm that PRODUCED a GEMM's output tensor's forward twin, done in the GEMM's epilogue.
//
// A conv / Linear input-gradient GEMM writes dy = d(loss)/d(y) where y = lab(act(bn(x))) is the output of the previous
// layer's fused norm (reference hgnetv2.py:35-80 ConvBNAct, hybrid_encoder.py:21-45 ConvNormLayer). That norm's backward
// starts with a pass over (x, dy) for per-channel sum(dz), sum(dz * xhat) (+ the two LAB sums), dz = dy * ls * act'(z)
// (normact.hip: bn_bwd_reduce). Here the GEMM's epilogue, which holds the finished dy tile, reads the matching x tile
// (+ the second gradient part dy2 when y was also written into a concat slice) and adds those sums to the norm's fp64
// replicas -- the reduction launch (and one read of dy) disappears; the norm's backward runs its dx pass only.
#pragma once
#include "common.h"

namespace dfine {

struct IgBnRed {
  const bf16_t* x;         // the norm's input [n_pix, ldx] (n_pix = rows of the GEMM's output map)
  const bf16_t* dy2;       // optional second part of the output gradient [n_pix, ld_dy2]
  const float* mean;       // [N] saved batch mean (or running mean)
  const float* rstd;       // [N]
  const float* gamma;      // [N]
  const float* beta;       // [N]
  const float* lab_scale;  // [1] or nullptr
  double* red;             // [8 replicas][4][N] fp64, zero-initialised by the caller
  int ldx, ld_dy2, act;    // act: 0 none, 1 relu, 2 silu
};

__device__ __forceinline__ float br_sigmoid(float z) { return __builtin_amdgcn_rcpf(1.f + __expf(-z)); }

// per-thread state of the epilogue's store loop: 8 consecutive channels
struct BnRedLane {
  float mean[8], rstd[8], gam[8], bet[8], ls;
  float a[4][8];
  __device__ __forceinline__ void init(const IgBnRed& br, int ncol, bool live) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      mean[e] = live ? br.mean[ncol + e] : 0.f;
      rstd[e] = live ? br.rstd[ncol + e] : 0.f;
      gam[e] = live ? br.gamma[ncol + e] : 0.f;
      bet[e] = live ? br.beta[ncol + e] : 0.f;
      a[0][e] = a[1][e] = a[2][e] = a[3][e] = 0.f;
    }
    ls = br.lab_scale ? br.lab_scale[0] : 1.f;
  }
  // fin: the 8 finished bf16 outputs of this row chunk (packed), pix: row of the output map
  __device__ __forceinline__ void row(const IgBnRed& br, const uint32_t (&fin)[4], size_t pix, int ncol) {
    const uint4 xv = *reinterpret_cast<const uint4*>(br.x + pix * br.ldx + ncol);
    const uint32_t xw[4] = {xv.x, xv.y, xv.z, xv.w};
    uint32_t ew[4] = {0u, 0u, 0u, 0u};
    if (br.dy2 != nullptr) {
      const uint4 ev = *reinterpret_cast<const uint4*>(br.dy2 + pix * br.ld_dy2 + ncol);
      ew[0] = ev.x, ew[1] = ev.y, ew[2] = ev.z, ew[3] = ev.w;
    }
    const bool lab = br.lab_scale != nullptr;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int sh = 16 * (e & 1);
      const float d = __uint_as_float((fin[e >> 1] >> sh) << 16) + __uint_as_float((ew[e >> 1] >> sh) << 16);
      const float xh = (__uint_as_float((xw[e >> 1] >> sh) << 16) - mean[e]) * rstd[e];
      const float z = xh * gam[e] + bet[e];
      float g = 1.f, f = z;
      if (br.act == 1) {
        g = z > 0.f ? 1.f : 0.f;
        f = fmaxf(z, 0.f);
      } else if (br.act == 2) {
        const float s = br_sigmoid(z);
        g = s * (1.f + z * (1.f - s));
        f = z * s;
      }
      const float dz = d * ls * g;
      a[0][e] += dz;
      a[1][e] += dz * xh;
      if (lab) {
        a[2][e] += d * f;
        a[3][e] += d;
      }
    }
  }
};

// Column sums of two per-thread quantities over the tile's rows through LDS (`part`: [2][ORPP][BN] floats), then one fp64
// atomic per (quantity, column) into out[(replica * 4 + q0 + q) * N + n]. Every thread of the workgroup calls it.
template <int BN, int ORPP, int NT>
__device__ __forceinline__ void bnred_columns(float* part, const float (&s1)[8], const float (&s2)[8], int o_r, int o_j, int tid,
                                              double* out, int q0, int N, int n0, bool skip_atomics = false) {
  float4* p1 = reinterpret_cast<float4*>(part + (0 * ORPP + o_r) * BN + o_j * 8);
  float4* p2 = reinterpret_cast<float4*>(part + (1 * ORPP + o_r) * BN + o_j * 8);
  p1[0] = make_float4(s1[0], s1[1], s1[2], s1[3]), p1[1] = make_float4(s1[4], s1[5], s1[6], s1[7]);
  p2[0] = make_float4(s2[0], s2[1], s2[2], s2[3]), p2[1] = make_float4(s2[4], s2[5], s2[6], s2[7]);
  __syncthreads();
  if (tid < 2 * BN) {
    const int q = tid / BN, c = tid % BN;
    float a = 0.f;
#pragma unroll
    for (int r = 0; r < ORPP; ++r) a += part[(q * ORPP + r) * BN + c];
    if (n0 + c < N && !skip_atomics) atomicAdd(out + ((size_t)(blockIdx.x & 7) * 4 + q0 + q) * N + n0 + c, (double)a);
  }
  __syncthreads();  // `part` may be reused by the next pair of quantities / the next row block
}

}  // namespace dfine
