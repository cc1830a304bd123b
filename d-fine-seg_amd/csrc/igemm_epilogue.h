// Epilogue of the register-staged implicit-GEMM body (igemm.hip), kept apart from its K loop so that another loop over the
// same 2 x 2-wave tile can end in it (round 3 tried a 3x3 halo-tile loop on it): accumulators (+bias, act) -> bf16 tile in LDS -> coalesced 16-byte rows at pitch ldc (C += result on request),
// per-channel statistics of the rounded outputs for the BatchNorm that follows (8-replica fp64 layout of dfine_bn_act_fwd),
// or the producing norm's backward sums (igemm_bnred.h). `rowmap(ml)` says which row of the output map tile row ml is (-1: none).
// 256 threads = 2 x 2 waves, wave tile 16*FM x 16*FN, D layout of mfma_f32_16x16x32 with the weight tile as operand A:
// row (= n) = (lane>>4)*4 + reg, col (= m) = lane&15. Every thread of the workgroup calls it after a barrier that ends all
// reads of the operand stages (the C tile reuses that LDS).
#pragma once
#include "common.h"
#include "igemm_bnred.h"

namespace dfine {

typedef short short8v __attribute__((ext_vector_type(8)));
typedef float float4v __attribute__((ext_vector_type(4)));
typedef dfine_igemm_geom IGemmGeom;

template <int FM, int FN>
struct IgEpi {
  static constexpr int BM = 32 * FM, BN = 32 * FN;
  static constexpr int CPITCH = BN + 8;  // tile row pitch (elements): 16 bytes of padding
  static constexpr int LDS_B = BM * CPITCH * 2 + 2 * (256 / (BN / 8)) * BN * 4;  // C tile + statistics partials
};

template <int FM, int FN, int OVEC, bool BNRED, typename RowMap>
__device__ __forceinline__ void ig_epilogue(float4v (&acc)[FN][FM], unsigned char* lds_raw, bf16_t* __restrict__ C,
                                            const float* __restrict__ bias, double* __restrict__ stats, const IGemmGeom& g,
                                            int n0, int tid, const IgBnRed& br, RowMap rowmap) {
  constexpr int BM = 32 * FM, BN = 32 * FN, CPITCH = IgEpi<FM, FN>::CPITCH;
  // ---- epilogue: accumulators (+bias, act) -> bf16 tile in LDS -> coalesced rows (+ per-channel statistics)
  // D layout of mfma_f32_16x16x32 with the weight tile as operand A: row (= n) = (lane>>4)*4 + reg, col (= m) = lane&15
  bf16_t* ct = reinterpret_cast<bf16_t*>(lds_raw);
  const int lane = tid & 63, wave = tid >> 6;
  const int wave_m = wave >> 1, wave_n = wave & 1;
  const int f_r = lane & 15, f_c = lane >> 4;
#pragma unroll
  for (int i = 0; i < FN; ++i) {
    const int nl = wave_n * (BN / 2) + 16 * i + 4 * f_c;
    float bv[4] = {0.f, 0.f, 0.f, 0.f};
    if (bias != nullptr) {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (n0 + nl + e < g.N) bv[e] = bias[n0 + nl + e];
    }
#pragma unroll
    for (int j = 0; j < FM; ++j) {
      const int ml = wave_m * (BM / 2) + 16 * j + f_r;
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[e] = acc[i][j][e] + bv[e];
        if ((g.act & 0xff) == 1) v[e] = fmaxf(v[e], 0.f);
      }
      uint2 pk;
      pk.x = (uint32_t)f32_to_bf16(v[0]) | ((uint32_t)f32_to_bf16(v[1]) << 16);
      pk.y = (uint32_t)f32_to_bf16(v[2]) | ((uint32_t)f32_to_bf16(v[3]) << 16);
      *reinterpret_cast<uint2*>(ct + ml * CPITCH + nl) = pk;
    }
  }
  __syncthreads();
  constexpr int OCPR = BN / 8, ORPP = 256 / OCPR;  // 16-byte chunks per output row, rows per pass
  const int o_j = tid % OCPR, o_r = tid / OCPR;
  float s1[8], s2[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) s1[e] = s2[e] = 0.f;
  const int ncol = n0 + o_j * 8;
  const int nvalid = min(max(g.N - ncol, 0), 8);
  const bool accumulate = (g.act & 0x400) != 0;
  BnRedLane brl;  // igemm_bnred.h: the producing norm's backward sums over this tile (input-gradient launches)
  if (BNRED) brl.init(br, ncol, nvalid == 8);
#pragma unroll
  for (int i = 0; i < BM / ORPP; ++i) {
    const int ml = o_r + i * ORPP;
    const long long opix = rowmap(ml);  // output-map row of tile row ml, or -1 (beyond the problem)
    if (opix < 0 || nvalid == 0) continue;
    const uint4 v = *reinterpret_cast<const uint4*>(ct + ml * CPITCH + o_j * 8);
    const size_t orow = (size_t)opix * g.ldc;
    bf16_t* dst = C + orow + ncol;
    if (accumulate) {  // C += result: the second of two input gradients of one tensor (RepVgg's 3x3 + 1x1, CSP's two 1x1)
      const uint32_t w[4] = {v.x, v.y, v.z, v.w};
      if (OVEC == 8 && nvalid == 8) {
        const uint4 old = *reinterpret_cast<const uint4*>(dst);
        const uint32_t o[4] = {old.x, old.y, old.z, old.w};
        uint32_t r[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float lo = __uint_as_float(w[q] << 16) + __uint_as_float(o[q] << 16);
          const float hi = __uint_as_float(w[q] & 0xffff0000u) + __uint_as_float(o[q] & 0xffff0000u);
          r[q] = (uint32_t)f32_to_bf16(lo) | ((uint32_t)f32_to_bf16(hi) << 16);
        }
        *reinterpret_cast<uint4*>(dst) = make_uint4(r[0], r[1], r[2], r[3]);
        if (BNRED) brl.row(br, r, orow / g.ldc, ncol);  // the SUM is the finished gradient
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (e < nvalid)
            dst[e] = f32_to_bf16(__uint_as_float((w[e >> 1] >> (16 * (e & 1))) << 16) + bf16_to_f32(dst[e]));
      }
    } else if (OVEC == 8 && nvalid == 8) {
      *reinterpret_cast<uint4*>(dst) = v;
      if (BNRED) {
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
        brl.row(br, w, orow / g.ldc, ncol);
      }
    } else {
      const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int e = 0; e < 8; ++e)
        if (e < nvalid) dst[e] = (bf16_t)(w[e >> 1] >> (16 * (e & 1)));
    }
    if (stats != nullptr) {
      const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float f = __uint_as_float((w[e >> 1] >> (16 * (e & 1))) << 16);
        s1[e] += f;
        s2[e] += f * f;
      }
    }
  }
  if (stats != nullptr && (g.act & 0x200) == 0) {
    // column sums over the tile's rows: every thread parks its 8 + 8 partial sums (its chunk column, its ORPP-strided
    // rows) in LDS behind the C tile, then thread c (< BN) adds the ORPP partials of column c and thread BN + c those of
    // its squares -- plain LDS traffic, no shuffles and no LDS atomics (that form cost 9 us on a 30 us launch)
    float* part = reinterpret_cast<float*>(lds_raw + BM * CPITCH * 2);  // [2][ORPP][BN]
    float4* p1 = reinterpret_cast<float4*>(part + (0 * ORPP + o_r) * BN + o_j * 8);
    float4* p2 = reinterpret_cast<float4*>(part + (1 * ORPP + o_r) * BN + o_j * 8);
    p1[0] = make_float4(s1[0], s1[1], s1[2], s1[3]), p1[1] = make_float4(s1[4], s1[5], s1[6], s1[7]);
    p2[0] = make_float4(s2[0], s2[1], s2[2], s2[3]), p2[1] = make_float4(s2[4], s2[5], s2[6], s2[7]);
    __syncthreads();
    if (tid < 2 * BN) {
      const int q = tid / BN, c = tid % BN;
      float acc = 0.f;
#pragma unroll
      for (int r = 0; r < ORPP; ++r) acc += part[(q * ORPP + r) * BN + c];
      if (n0 + c < g.N && (g.act & 0x100) == 0)
        atomicAdd(stats + ((size_t)(blockIdx.x & 7) * 2 + q) * g.N + n0 + c, (double)acc);
    }
  }
  if (BNRED) {
    float* part = reinterpret_cast<float*>(lds_raw + BM * CPITCH * 2);  // [2][ORPP][BN]
    bnred_columns<BN, ORPP, 256>(part, brl.a[0], brl.a[1], o_r, o_j, tid, br.red, 0, g.N, n0, (g.act & 0x2000) != 0);
    if (br.lab_scale != nullptr) bnred_columns<BN, ORPP, 256>(part, brl.a[2], brl.a[3], o_r, o_j, tid, br.red, 2, g.N, n0, (g.act & 0x2000) != 0);
  }
}

}  // namespace dfine
