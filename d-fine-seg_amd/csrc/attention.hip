// Multi-head self-attention, forward and backward, for the two attention sites of the model:
//   * decoder self-attention over <= 500 queries with the denoising bool mask
//     (reference src/d_fine/arch/dfine_decoder.py:233-240: nn.MultiheadAttention(q = k = tgt + pos, v = tgt, attn_mask))
//   * AIFI over the stride-32 tokens (100 / 400 / 900) without a mask (hybrid_encoder.py:243-290)
//         O = softmax(scale * Q K^T + mask) V        per (image, head), head_dim in {16, 32, 48}
// Operands are read where the projection GEMMs left them: q, k, v are [B*L, ...] row-major matrices with a row
// pitch (q and k are the two halves of one [B*L, 2D] projection output), head h = columns [h*dh, (h+1)*dh); O
// and the gradients are written the same way, so there are no head permutes / copies around the kernels.
//
// Mapping (all three kernels): one workgroup = 4 waves = 64 rows of one (image, head); it walks the other
// sequence axis in tiles of 64 staged through LDS. MFMA 16x16x32 bf16:
//   forward / dQ : S^T = K Q^T ("swapped": a lane owns ONE query, lane&15, and 4 consecutive keys per 16-key
//                  block) so softmax statistics are per lane; P^T then IS the A operand of P V up to a fixed
//                  permutation of the 32 keys of a k-step, which the V operand reproduces by reading its
//                  4(key) x 16(column) blocks with ds_read_b64_tr_b16 -- no LDS round trip for P.
//   dK / dV      : the mirror image (a lane owns one key, 4 consecutive queries per block).
// Softmax is two-pass over the keys (row maximum first, then exp / sum / PV): the products are tiny
// (7.9 GFLOP per decoder layer at batch 32) and the second QK^T costs less than rescaling O.
// RESIDENT builds (default for the backward kernels): the whole K and V (dK/dV kernel: Q and dO) of one (image, head) --
// <= 512 rows x 32..64 columns -- are staged ONCE into LDS (80..144 KB of the 160), one barrier, and every wave then walks
// its 16-row groups through all tiles with no further barrier or global load. Faster for the backward kernels, slower for
// the forward (see attn_resident below); both forms are tested.
// The mask is read bit-packed (AttnGeom::bits_q / bits_k): as single bytes it cost 16 loads per lane and tile and DOUBLED
// every masked launch (187 -> 98 us forward, 476 -> 250 us forward + backward at B 32, L 492).
// Backward = two kernels without atomics: dQ (one pass over the keys per query tile, also emits
// D = rowsum(dO * O)) and dK/dV (one pass over the queries per key tile), each recomputing P from the saved
// log-sum-exp.
#include <stdlib.h>

#include "common.h"

namespace dfine {

typedef short short4v __attribute__((ext_vector_type(4)));
typedef short short8v __attribute__((ext_vector_type(8)));
typedef float float4v __attribute__((ext_vector_type(4)));

constexpr int AT_T = 64;  // rows per workgroup and per staged tile

struct AttnGeom {
  int B, H, L, dh;
  int ldq, ldk, ldv, ldo;
  float scale_log2;  // scale * log2(e)
  // the mask, bit-packed (optional; else the kernels read `blocked` byte by byte -- 16 single-byte loads per lane and key
  // tile, which DOUBLED the time of every masked launch): bits_q[q][t] bit j = blocked[q][64 t + j] (a lane's query vs the
  // 64 keys of tile t), bits_k[k][t] bit j = blocked[64 t + j][k] (a lane's key vs the 64 queries of tile t)
  const unsigned long long* bits_q;
  const unsigned long long* bits_k;
};

// tile [AT_T][PITCH] bf16 in LDS, PITCH = DHP + 8 (rows stay 16-byte aligned; 8-byte aligned for tr reads)
template <int DHP>
struct Tile {
  static constexpr int PITCH = DHP + 8;
  // rows [row0, row0+64) of a [*, ld] matrix (columns col0 .. col0+dh) -> LDS; rows >= nrows and columns >= dh are zero
  static __device__ __forceinline__ void stage(bf16_t* dst, const bf16_t* __restrict__ src, int ld, long long row_base,
                                               int row0, int nrows, int col0, int dh, int tid, int nthreads = 256) {
    constexpr int CPR = DHP / 8;
    for (int idx = tid; idx < AT_T * CPR; idx += nthreads) {
      const int r = idx / CPR, c = idx % CPR;
      uint4 v = make_uint4(0u, 0u, 0u, 0u);
      if (row0 + r < nrows && c * 8 < dh) v = *reinterpret_cast<const uint4*>(src + (row_base + row0 + r) * ld + col0 + c * 8);
      *reinterpret_cast<uint4*>(dst + r * PITCH + c * 8) = v;
    }
  }
  // A/B operand whose reduction index is the head dimension: row (r0 + lane&15), columns ks*32 + 8*(lane>>4) ..
  static __device__ __forceinline__ short8v row_frag(const bf16_t* t, int r0, int ks, int lane) {
    return *reinterpret_cast<const short8v*>(t + (r0 + (lane & 15)) * PITCH + ks * 32 + (lane >> 4) * 8);
  }
  // B operand whose reduction index is the tile ROW (keys / queries): for the k-step of rows [rb, rb+32), lane
  // (g = lane>>4, i = lane&15) receives column c0+i of rows rb+4g..+3 (elements 0-3) and rb+16+4g..+3 (elements 4-7)
  static __device__ __forceinline__ short8v tr_frag(const bf16_t* t, int rb, int c0, int lane) {
    const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
    typedef __attribute__((address_space(3))) short4v lds_v4;
    const bf16_t* a = t + (rb + 4 * g + q) * PITCH + c0 + 4 * p;
    const short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)a);
    const short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)(a + 16 * PITCH));
    short8v f;
    f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
    f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
    return f;
  }
};

__device__ __forceinline__ short8v pack_p(const float4v& a, const float4v& b) {
  short8v f;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    f[e] = (short)f32_to_bf16(a[e]);
    f[4 + e] = (short)f32_to_bf16(b[e]);
  }
  return f;
}

// row fragment straight from global memory (the wave's own 16 rows: loaded once, kept in registers)
template <int DHP>
__device__ __forceinline__ void own_rows(const bf16_t* __restrict__ src, int ld, long long row_base, int row0, int nrows,
                                         int col0, int dh, int lane, short8v (&f)[DHP / 32]) {
#pragma unroll
  for (int ks = 0; ks < DHP / 32; ++ks) {
    const int r = row0 + (lane & 15), c = ks * 32 + (lane >> 4) * 8;
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (r < nrows && c < dh) v = *reinterpret_cast<const uint4*>(src + (row_base + r) * ld + col0 + c);
    f[ks] = *reinterpret_cast<short8v*>(&v);
  }
}

// ---------------------------------------------------------------------------------------------------- forward
template <int DHP, bool RES>
__global__ __launch_bounds__(RES ? 512 : 256) void attn_fwd_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k,
                                                       const bf16_t* __restrict__ v, const unsigned char* __restrict__ blocked,
                                                       bf16_t* __restrict__ o, float* __restrict__ lse, AttnGeom g) {
  constexpr int PITCH = Tile<DHP>::PITCH, NC = DHP / 16;  // NC: 16-column blocks of the head dimension
  extern __shared__ __attribute__((aligned(16))) bf16_t lds[];
  const int n_tiles = (g.L + AT_T - 1) / AT_T;
  bf16_t* kt = lds;                                          // RES: tile t at kt + t * AT_T * PITCH
  bf16_t* vt = lds + (RES ? n_tiles : 1) * AT_T * PITCH;
  const int bh = blockIdx.x, b = bh / g.H, h = bh % g.H;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long long row_base = (long long)b * g.L;
  const int col0 = h * g.dh;
  const int grp = lane >> 4;
  if (RES) {
    for (int t = 0; t < n_tiles; ++t) {
      Tile<DHP>::stage(kt + t * AT_T * PITCH, k, g.ldk, row_base, t * AT_T, g.L, col0, g.dh, tid, blockDim.x);
      Tile<DHP>::stage(vt + t * AT_T * PITCH, v, g.ldv, row_base, t * AT_T, g.L, col0, g.dh, tid, blockDim.x);
    }
    __syncthreads();
  }
  const int n_waves = blockDim.x >> 6, n_groups = (g.L + 15) / 16;
  // RES: the waves of the workgroup share out ALL 16-query groups of this (image, head); else: this workgroup's 64 queries
  for (int rg = RES ? wave : blockIdx.y * 4 + wave; rg < (RES ? n_groups : blockIdx.y * 4 + wave + 1); rg += n_waves) {
  const int q0 = rg * 16;  // this wave's 16 queries
  const int my_q = q0 + (lane & 15);
  short8v qf[DHP / 32];
  own_rows<DHP>(q, g.ldq, row_base, q0, g.L, col0, g.dh, lane, qf);
  const unsigned long long* mbits = (g.bits_q != nullptr && my_q < g.L) ? g.bits_q + (size_t)my_q * n_tiles : nullptr;
  const unsigned char* mrow = (g.bits_q == nullptr && blocked != nullptr && my_q < g.L) ? blocked + (size_t)my_q * g.L : nullptr;

  auto scores = [&](int t, float4v (&s)[4]) {  // S^T of key tile t (staged in kt): s[blk][reg] = key t*64 + 16*blk + 4*grp + reg
    const bf16_t* ktt = kt + (RES ? t * AT_T * PITCH : 0);
    const unsigned long long mw = mbits != nullptr ? mbits[t] : 0ull;
#pragma unroll
    for (int blk = 0; blk < 4; ++blk) {
      float4v acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < DHP / 32; ++ks)
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Tile<DHP>::row_frag(ktt, 16 * blk, ks, lane), qf[ks], acc, 0, 0, 0);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int kl = 16 * blk + 4 * grp + e, key = t * AT_T + kl;
        float x = acc[e] * g.scale_log2;
        if (key >= g.L || ((mw >> kl) & 1ull) || (mrow != nullptr && mrow[key])) x = -INFINITY;
        s[blk][e] = x;
      }
    }
  };

  // pass 1: row maximum
  float m = -INFINITY;
  for (int t = 0; t < n_tiles; ++t) {
    if (!RES) {
      __syncthreads();
      Tile<DHP>::stage(kt, k, g.ldk, row_base, t * AT_T, g.L, col0, g.dh, tid);
      __syncthreads();
    }
    float4v s[4];
    scores(t, s);
#pragma unroll
    for (int blk = 0; blk < 4; ++blk)
#pragma unroll
      for (int e = 0; e < 4; ++e) m = fmaxf(m, s[blk][e]);
  }
  m = fmaxf(m, __shfl_xor(m, 16, 64));
  m = fmaxf(m, __shfl_xor(m, 32, 64));
  const float m_use = m == -INFINITY ? 0.f : m;  // fully masked row: p = 0 everywhere, output 0

  // pass 2: p = exp2(s - m), l = sum p, O = P V
  float l = 0.f;
  float4v oacc[NC];
#pragma unroll
  for (int n = 0; n < NC; ++n) oacc[n] = float4v{0.f, 0.f, 0.f, 0.f};
  for (int t = 0; t < n_tiles; ++t) {
    if (!RES) {
      __syncthreads();
      Tile<DHP>::stage(kt, k, g.ldk, row_base, t * AT_T, g.L, col0, g.dh, tid);
      Tile<DHP>::stage(vt, v, g.ldv, row_base, t * AT_T, g.L, col0, g.dh, tid);
      __syncthreads();
    }
    const bf16_t* vtt = vt + (RES ? t * AT_T * PITCH : 0);
    float4v s[4];
    scores(t, s);
#pragma unroll
    for (int blk = 0; blk < 4; ++blk)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float p = exp2f(s[blk][e] - m_use);
        s[blk][e] = p;
        l += p;
      }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {  // 32 keys per k-step
      const short8v pf = pack_p(s[2 * ks], s[2 * ks + 1]);
#pragma unroll
      for (int n = 0; n < NC; ++n)
        oacc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pf, Tile<DHP>::tr_frag(vtt, 32 * ks, 16 * n, lane), oacc[n], 0, 0, 0);
    }
  }
  l += __shfl_xor(l, 16, 64);
  l += __shfl_xor(l, 32, 64);
  if (grp == 0 && my_q < g.L) lse[((size_t)b * g.H + h) * g.L + my_q] = l > 0.f ? m_use + log2f(l) : -INFINITY;  // log2 domain
  // O layout: row (query) = 4*grp + reg, column = 16n + (lane&15); 1/l of query 4*grp+reg lives in lane 4*grp+reg
  const float inv_l = l > 0.f ? 1.f / l : 0.f;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float il = __shfl(inv_l, 4 * grp + e, 64);
    const int qq = q0 + 4 * grp + e;
#pragma unroll
    for (int n = 0; n < NC; ++n) {
      const int c = 16 * n + (lane & 15);
      if (qq < g.L && c < g.dh) o[(row_base + qq) * g.ldo + col0 + c] = f32_to_bf16(oacc[n][e] * il);
    }
  }
  }  // row groups
}

// ---------------------------------------------------------------------------------------------------- backward: dQ (+ D)
template <int DHP, bool RES>
__global__ __launch_bounds__(RES ? 512 : 256) void attn_bwd_dq_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k,
                                                          const bf16_t* __restrict__ v, const unsigned char* __restrict__ blocked,
                                                          const bf16_t* __restrict__ o, const float* __restrict__ lse,
                                                          const bf16_t* __restrict__ dout, int lddo, bf16_t* __restrict__ dq,
                                                          int lddq, float* __restrict__ dsum, AttnGeom g, float scale) {
  constexpr int PITCH = Tile<DHP>::PITCH, NC = DHP / 16;
  extern __shared__ __attribute__((aligned(16))) bf16_t lds[];
  const int n_tiles = (g.L + AT_T - 1) / AT_T;
  bf16_t* kt = lds;
  bf16_t* vt = lds + (RES ? n_tiles : 1) * AT_T * PITCH;
  const int bh = blockIdx.x, b = bh / g.H, h = bh % g.H;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long long row_base = (long long)b * g.L;
  const int col0 = h * g.dh;
  const int grp = lane >> 4;
  if (RES) {
    for (int t = 0; t < n_tiles; ++t) {
      Tile<DHP>::stage(kt + t * AT_T * PITCH, k, g.ldk, row_base, t * AT_T, g.L, col0, g.dh, tid, blockDim.x);
      Tile<DHP>::stage(vt + t * AT_T * PITCH, v, g.ldv, row_base, t * AT_T, g.L, col0, g.dh, tid, blockDim.x);
    }
    __syncthreads();
  }
  const int n_waves = blockDim.x >> 6, n_groups = (g.L + 15) / 16;
  for (int rg = RES ? wave : blockIdx.y * 4 + wave; rg < (RES ? n_groups : blockIdx.y * 4 + wave + 1); rg += n_waves) {
  const int q0 = rg * 16;
  const int my_q = q0 + (lane & 15);
  short8v qf[DHP / 32], dof[DHP / 32];
  own_rows<DHP>(q, g.ldq, row_base, q0, g.L, col0, g.dh, lane, qf);
  own_rows<DHP>(dout, lddo, row_base, q0, g.L, col0, g.dh, lane, dof);
  // D = rowsum(dO * O) of my query: this lane holds 8 (or 16) of its dh columns; the 4 lanes of a query add up
  float d = 0.f;
  {
    short8v of[DHP / 32];
    own_rows<DHP>(o, g.ldo, row_base, q0, g.L, col0, g.dh, lane, of);
#pragma unroll
    for (int ks = 0; ks < DHP / 32; ++ks)
#pragma unroll
      for (int e = 0; e < 8; ++e) d += bf16_to_f32((bf16_t)of[ks][e]) * bf16_to_f32((bf16_t)dof[ks][e]);
    d += __shfl_xor(d, 16, 64);
    d += __shfl_xor(d, 32, 64);
    if (grp == 0 && my_q < g.L) dsum[((size_t)b * g.H + h) * g.L + my_q] = d;
  }
  const float my_lse = my_q < g.L ? lse[((size_t)b * g.H + h) * g.L + my_q] : 0.f;
  const bool dead = !(my_lse > -INFINITY);  // fully masked row
  const unsigned long long* mbits = (g.bits_q != nullptr && my_q < g.L) ? g.bits_q + (size_t)my_q * n_tiles : nullptr;
  const unsigned char* mrow = (g.bits_q == nullptr && blocked != nullptr && my_q < g.L) ? blocked + (size_t)my_q * g.L : nullptr;
  float4v acc_dq[NC];
#pragma unroll
  for (int n = 0; n < NC; ++n) acc_dq[n] = float4v{0.f, 0.f, 0.f, 0.f};
  for (int t = 0; t < n_tiles; ++t) {
    if (!RES) {
      __syncthreads();
      Tile<DHP>::stage(kt, k, g.ldk, row_base, t * AT_T, g.L, col0, g.dh, tid);
      Tile<DHP>::stage(vt, v, g.ldv, row_base, t * AT_T, g.L, col0, g.dh, tid);
      __syncthreads();
    }
    const bf16_t* ktt = kt + (RES ? t * AT_T * PITCH : 0);
    const bf16_t* vtt = vt + (RES ? t * AT_T * PITCH : 0);
    const unsigned long long mw = mbits != nullptr ? mbits[t] : 0ull;
    float4v ds[4];
#pragma unroll
    for (int blk = 0; blk < 4; ++blk) {
      float4v s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < DHP / 32; ++ks) {
        s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Tile<DHP>::row_frag(ktt, 16 * blk, ks, lane), qf[ks], s, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Tile<DHP>::row_frag(vtt, 16 * blk, ks, lane), dof[ks], dp, 0, 0, 0);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int kl = 16 * blk + 4 * grp + e, key = t * AT_T + kl;
        const bool off = dead || key >= g.L || ((mw >> kl) & 1ull) || (mrow != nullptr && mrow[key]);
        const float p = off ? 0.f : exp2f(s[e] * g.scale_log2 - my_lse);
        ds[blk][e] = p * (dp[e] - d);
      }
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const short8v f = pack_p(ds[2 * ks], ds[2 * ks + 1]);
#pragma unroll
      for (int n = 0; n < NC; ++n)
        acc_dq[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f, Tile<DHP>::tr_frag(ktt, 32 * ks, 16 * n, lane), acc_dq[n], 0, 0, 0);
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int qq = q0 + 4 * grp + e;
#pragma unroll
    for (int n = 0; n < NC; ++n) {
      const int c = 16 * n + (lane & 15);
      if (qq < g.L && c < g.dh) dq[(row_base + qq) * lddq + col0 + c] = f32_to_bf16(acc_dq[n][e] * scale);
    }
  }
  }  // row groups
}

// ---------------------------------------------------------------------------------------------------- backward: dK, dV
template <int DHP, bool RES>
__global__ __launch_bounds__(RES ? 512 : 256) void attn_bwd_dkv_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k,
                                                           const bf16_t* __restrict__ v, const unsigned char* __restrict__ blocked,
                                                           const float* __restrict__ lse, const float* __restrict__ dsum,
                                                           const bf16_t* __restrict__ dout, int lddo, bf16_t* __restrict__ dk,
                                                           int lddk, bf16_t* __restrict__ dv, int lddv, AttnGeom g, float scale) {
  constexpr int PITCH = Tile<DHP>::PITCH, NC = DHP / 16;
  extern __shared__ __attribute__((aligned(16))) bf16_t lds[];
  const int n_tiles = (g.L + AT_T - 1) / AT_T;
  const int res_tiles = RES ? n_tiles : 1;
  bf16_t* qt = lds;
  bf16_t* dot = lds + res_tiles * AT_T * PITCH;
  float* s_lse = reinterpret_cast<float*>(lds + 2 * res_tiles * AT_T * PITCH);  // [res_tiles * 64]
  float* s_d = s_lse + res_tiles * AT_T;
  const int bh = blockIdx.x, b = bh / g.H, h = bh % g.H;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long long row_base = (long long)b * g.L;
  const int col0 = h * g.dh;
  const int grp = lane >> 4;
  if (RES) {
    for (int t = 0; t < n_tiles; ++t) {
      Tile<DHP>::stage(qt + t * AT_T * PITCH, q, g.ldq, row_base, t * AT_T, g.L, col0, g.dh, tid, blockDim.x);
      Tile<DHP>::stage(dot + t * AT_T * PITCH, dout, lddo, row_base, t * AT_T, g.L, col0, g.dh, tid, blockDim.x);
    }
    for (int qq = tid; qq < n_tiles * AT_T; qq += blockDim.x) {
      s_lse[qq] = qq < g.L ? lse[((size_t)b * g.H + h) * g.L + qq] : -INFINITY;
      s_d[qq] = qq < g.L ? dsum[((size_t)b * g.H + h) * g.L + qq] : 0.f;
    }
    __syncthreads();
  }
  const int n_waves = blockDim.x >> 6, n_groups = (g.L + 15) / 16;
  for (int rg = RES ? wave : blockIdx.y * 4 + wave; rg < (RES ? n_groups : blockIdx.y * 4 + wave + 1); rg += n_waves) {
  const int k0 = rg * 16;  // this wave's 16 keys
  const int my_k = k0 + (lane & 15);
  short8v kf[DHP / 32], vf[DHP / 32];
  own_rows<DHP>(k, g.ldk, row_base, k0, g.L, col0, g.dh, lane, kf);
  own_rows<DHP>(v, g.ldv, row_base, k0, g.L, col0, g.dh, lane, vf);
  float4v acc_dk[NC], acc_dv[NC];
#pragma unroll
  for (int n = 0; n < NC; ++n) acc_dk[n] = acc_dv[n] = float4v{0.f, 0.f, 0.f, 0.f};
  for (int t = 0; t < n_tiles; ++t) {
    if (!RES) {
      __syncthreads();
      Tile<DHP>::stage(qt, q, g.ldq, row_base, t * AT_T, g.L, col0, g.dh, tid);
      Tile<DHP>::stage(dot, dout, lddo, row_base, t * AT_T, g.L, col0, g.dh, tid);
      if (tid < AT_T) {
        const int qq = t * AT_T + tid;
        s_lse[tid] = qq < g.L ? lse[((size_t)b * g.H + h) * g.L + qq] : -INFINITY;
        s_d[tid] = qq < g.L ? dsum[((size_t)b * g.H + h) * g.L + qq] : 0.f;
      }
      __syncthreads();
    }
    const int toff = RES ? t * AT_T : 0;
    const bf16_t* qtt = qt + toff * PITCH;
    const bf16_t* dott = dot + toff * PITCH;
    const unsigned long long mw = (g.bits_k != nullptr && my_k < g.L) ? g.bits_k[(size_t)my_k * n_tiles + t] : 0ull;
    const bool bytes = g.bits_k == nullptr && blocked != nullptr;
    float4v p[4], ds[4];
#pragma unroll
    for (int blk = 0; blk < 4; ++blk) {  // S / dP: row (query) = t*64 + 16*blk + 4*grp + reg, column (key) = lane&15
      float4v s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < DHP / 32; ++ks) {
        s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Tile<DHP>::row_frag(qtt, 16 * blk, ks, lane), kf[ks], s, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Tile<DHP>::row_frag(dott, 16 * blk, ks, lane), vf[ks], dp, 0, 0, 0);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int ql = 16 * blk + 4 * grp + e, qq = t * AT_T + ql;
        const float ls = s_lse[toff + ql];
        const bool off = qq >= g.L || my_k >= g.L || !(ls > -INFINITY) || ((mw >> ql) & 1ull) ||
                         (bytes && blocked[(size_t)qq * g.L + my_k]);
        const float pe = off ? 0.f : exp2f(s[e] * g.scale_log2 - ls);
        p[blk][e] = pe;
        ds[blk][e] = pe * (dp[e] - s_d[toff + ql]);
      }
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {  // 32 queries per k-step
      const short8v pf = pack_p(p[2 * ks], p[2 * ks + 1]), dsf = pack_p(ds[2 * ks], ds[2 * ks + 1]);
#pragma unroll
      for (int n = 0; n < NC; ++n) {
        acc_dv[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pf, Tile<DHP>::tr_frag(dott, 32 * ks, 16 * n, lane), acc_dv[n], 0, 0, 0);
        acc_dk[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dsf, Tile<DHP>::tr_frag(qtt, 32 * ks, 16 * n, lane), acc_dk[n], 0, 0, 0);
      }
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int kk = k0 + 4 * grp + e;
#pragma unroll
    for (int n = 0; n < NC; ++n) {
      const int c = 16 * n + (lane & 15);
      if (kk < g.L && c < g.dh) {
        dk[(row_base + kk) * lddk + col0 + c] = f32_to_bf16(acc_dk[n][e] * scale);
        dv[(row_base + kk) * lddv + col0 + c] = f32_to_bf16(acc_dv[n][e]);
      }
    }
  }
  }  // row groups
}

static bool attn_aligned(const void* p, int ld) { return ((uintptr_t)p % 16 == 0) && ld % 8 == 0; }

// dynamic LDS of one launch: two operand arrays of `tiles` 64-row tiles (+ lse / D rows for the dK/dV kernel)
static size_t attn_lds_bytes(int tiles, int dhp, int with_rows) {
  return (size_t)2 * tiles * AT_T * (dhp + 8) * sizeof(bf16_t) + (with_rows ? (size_t)2 * tiles * AT_T * sizeof(float) : 0);
}
// whole-sequence residency (everything must fit the 160 KB of a CU). Default: BACKWARD only. Measured at the decoder's shape
// (B 32, L 492, 8 heads of 32; tools/time_attention.py, bench.py): forward 99 us resident vs 82 us tiled -- one 80 KB
// workgroup per CU leaves 2 waves per SIMD and the tiled form's 8 workgroups per CU hide its staging round trips better
// than residency removes them -- but dQ + dK/dV 150 vs 166 us (twice the operands per tile to re-stage).
// DFINE_ATTN_RESIDENT=0 / 1 forces tiled / resident for both directions.
static bool attn_resident(int n_tiles, int dhp, int backward) {
  static const int forced = getenv("DFINE_ATTN_RESIDENT") ? atoi(getenv("DFINE_ATTN_RESIDENT")) : -1;
  const bool want = forced < 0 ? backward != 0 : forced != 0;
  return want && attn_lds_bytes(n_tiles, dhp, backward) <= 152 * 1024;
}
template <typename K>
static int attn_allow_lds(K kernel, size_t bytes) {
  static size_t allowed = 64 * 1024;  // per kernel instantiation: raise the cap once, outside of any capture after the first use
  if (bytes <= allowed) return 0;
  allowed = bytes;
  const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  return e == hipSuccess ? 0 : (int)e;
}

}  // namespace dfine

using namespace dfine;

extern "C" int dfine_attention_fwd(const void* q, const void* k, const void* v, int ldq, int ldk, int ldv,
                                   const unsigned char* blocked, const unsigned long long* bits_q, void* o, int ldo, float* lse,
                                   int B, int H, int L, int dh, float scale, void* stream) {
  if (!q || !k || !v || !o || !lse) return DFINE_E_BADARG;
  if (B <= 0 || H <= 0 || L <= 0 || dh <= 0 || ldq < H * dh || ldk < H * dh || ldv < H * dh || ldo < H * dh) return DFINE_E_BADARG;
  if (dh % 8 != 0 || dh > 64 || (long long)B * H > 0x7fffffffLL) return DFINE_E_UNSUPPORTED;
  if (!attn_aligned(q, ldq) || !attn_aligned(k, ldk) || !attn_aligned(v, ldv)) return DFINE_E_UNSUPPORTED;
  const AttnGeom g{B, H, L, dh, ldq, ldk, ldv, ldo, scale * 1.4426950408889634f, blocked ? bits_q : nullptr, nullptr};
  const int n_tiles = (L + AT_T - 1) / AT_T;
  const bool res = attn_resident(n_tiles, dh <= 32 ? 32 : 64, 0);
  const dim3 grid((unsigned)(B * H), res ? 1u : (unsigned)n_tiles), block(res ? 512 : 256);
#define DFINE_ATTN_FWD(DHP_, RES_)                                                                                              \
  do {                                                                                                                         \
    const size_t lds = attn_lds_bytes(RES_ ? n_tiles : 1, DHP_, 0);                                                            \
    if (int rc = attn_allow_lds(attn_fwd_kernel<DHP_, RES_>, lds)) return rc;                                                  \
    hipLaunchKernelGGL((attn_fwd_kernel<DHP_, RES_>), grid, block, lds, (hipStream_t)stream, (const bf16_t*)q, (const bf16_t*)k, \
                       (const bf16_t*)v, blocked, (bf16_t*)o, lse, g);                                                         \
  } while (0)
  if (dh <= 32) {
    if (res) DFINE_ATTN_FWD(32, true);
    else DFINE_ATTN_FWD(32, false);
  } else {
    if (res) DFINE_ATTN_FWD(64, true);
    else DFINE_ATTN_FWD(64, false);
  }
#undef DFINE_ATTN_FWD
  DFINE_LAUNCH_CHECK();
  return 0;
}

extern "C" int dfine_attention_bwd(const void* q, const void* k, const void* v, int ldq, int ldk, int ldv,
                                   const unsigned char* blocked, const unsigned long long* bits_q, const unsigned long long* bits_k,
                                   const void* o, int ldo, const float* lse, const void* dout, int lddo, void* dq, int lddq,
                                   void* dk, int lddk, void* dv, int lddv, float* dsum, int B, int H, int L, int dh, float scale,
                                   void* stream) {
  if (!q || !k || !v || !o || !lse || !dout || !dq || !dk || !dv || !dsum) return DFINE_E_BADARG;
  if (B <= 0 || H <= 0 || L <= 0 || dh <= 0) return DFINE_E_BADARG;
  const int D = H * dh;
  if (ldq < D || ldk < D || ldv < D || ldo < D || lddo < D || lddq < D || lddk < D || lddv < D) return DFINE_E_BADARG;
  if (dh % 8 != 0 || dh > 64) return DFINE_E_UNSUPPORTED;
  if (!attn_aligned(q, ldq) || !attn_aligned(k, ldk) || !attn_aligned(v, ldv) || !attn_aligned(o, ldo) || !attn_aligned(dout, lddo))
    return DFINE_E_UNSUPPORTED;
  if ((bits_q != nullptr) != (bits_k != nullptr)) return DFINE_E_BADARG;
  const AttnGeom g{B, H, L, dh, ldq, ldk, ldv, ldo, scale * 1.4426950408889634f, blocked ? bits_q : nullptr, blocked ? bits_k : nullptr};
  const int n_tiles = (L + AT_T - 1) / AT_T;
  const bool res = attn_resident(n_tiles, dh <= 32 ? 32 : 64, 1);
  const dim3 grid((unsigned)(B * H), res ? 1u : (unsigned)n_tiles), block(res ? 512 : 256);
  hipStream_t st = (hipStream_t)stream;
#define DFINE_ATTN_BWD(DHP_, RES_)                                                                                                  \
  do {                                                                                                                             \
    const size_t lds_q = attn_lds_bytes(RES_ ? n_tiles : 1, DHP_, 0), lds_kv = attn_lds_bytes(RES_ ? n_tiles : 1, DHP_, 1);        \
    if (int rc = attn_allow_lds(attn_bwd_dq_kernel<DHP_, RES_>, lds_q)) return rc;                                                 \
    if (int rc = attn_allow_lds(attn_bwd_dkv_kernel<DHP_, RES_>, lds_kv)) return rc;                                               \
    hipLaunchKernelGGL((attn_bwd_dq_kernel<DHP_, RES_>), grid, block, lds_q, st, (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, \
                       blocked, (const bf16_t*)o, lse, (const bf16_t*)dout, lddo, (bf16_t*)dq, lddq, dsum, g, scale);              \
    hipLaunchKernelGGL((attn_bwd_dkv_kernel<DHP_, RES_>), grid, block, lds_kv, st, (const bf16_t*)q, (const bf16_t*)k,             \
                       (const bf16_t*)v, blocked, lse, dsum, (const bf16_t*)dout, lddo, (bf16_t*)dk, lddk, (bf16_t*)dv, lddv, g,   \
                       scale);                                                                                                     \
  } while (0)
  if (dh <= 32) {
    if (res) DFINE_ATTN_BWD(32, true);
    else DFINE_ATTN_BWD(32, false);
  } else {
    if (res) DFINE_ATTN_BWD(64, true);
    else DFINE_ATTN_BWD(64, false);
  }
#undef DFINE_ATTN_BWD
  DFINE_LAUNCH_CHECK();
  return 0;
}
