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
// Backward = two kernels without atomics: dQ (one pass over the keys per query tile, also emits
// D = rowsum(dO * O)) and dK/dV (one pass over the queries per key tile), each recomputing P from the saved
// log-sum-exp.
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
};

// tile [AT_T][PITCH] bf16 in LDS, PITCH = DHP + 8 (rows stay 16-byte aligned; 8-byte aligned for tr reads)
template <int DHP>
struct Tile {
  static constexpr int PITCH = DHP + 8;
  // rows [row0, row0+64) of a [*, ld] matrix (columns col0 .. col0+dh) -> LDS; rows >= nrows and columns >= dh are zero
  static __device__ __forceinline__ void stage(bf16_t* dst, const bf16_t* __restrict__ src, int ld, long long row_base,
                                               int row0, int nrows, int col0, int dh, int tid) {
    constexpr int CPR = DHP / 8;
#pragma unroll
    for (int i = 0; i < AT_T * CPR / 256; ++i) {
      const int idx = tid + i * 256, r = idx / CPR, c = idx % CPR;
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
template <int DHP>
__global__ __launch_bounds__(256) void attn_fwd_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k,
                                                       const bf16_t* __restrict__ v, const unsigned char* __restrict__ blocked,
                                                       bf16_t* __restrict__ o, float* __restrict__ lse, AttnGeom g) {
  constexpr int PITCH = Tile<DHP>::PITCH, NC = DHP / 16;  // NC: 16-column blocks of the head dimension
  __shared__ __attribute__((aligned(16))) bf16_t lds[2 * AT_T * PITCH];
  bf16_t* kt = lds;
  bf16_t* vt = lds + AT_T * PITCH;
  const int bh = blockIdx.x, b = bh / g.H, h = bh % g.H;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int q0 = blockIdx.y * AT_T + wave * 16;  // this wave's 16 queries
  const long long row_base = (long long)b * g.L;
  const int col0 = h * g.dh;
  const int my_q = q0 + (lane & 15), grp = lane >> 4;
  short8v qf[DHP / 32];
  own_rows<DHP>(q, g.ldq, row_base, q0, g.L, col0, g.dh, lane, qf);
  const int n_tiles = (g.L + AT_T - 1) / AT_T;
  const unsigned char* mrow = (blocked != nullptr && my_q < g.L) ? blocked + (size_t)my_q * g.L : nullptr;

  auto scores = [&](int t, float4v (&s)[4]) {  // S^T of key tile t (staged in kt): s[blk][reg] = key t*64 + 16*blk + 4*grp + reg
#pragma unroll
    for (int blk = 0; blk < 4; ++blk) {
      float4v acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < DHP / 32; ++ks)
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Tile<DHP>::row_frag(kt, 16 * blk, ks, lane), qf[ks], acc, 0, 0, 0);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int key = t * AT_T + 16 * blk + 4 * grp + e;
        float x = acc[e] * g.scale_log2;
        if (key >= g.L || (mrow != nullptr && mrow[key])) x = -INFINITY;
        s[blk][e] = x;
      }
    }
  };

  // pass 1: row maximum
  float m = -INFINITY;
  for (int t = 0; t < n_tiles; ++t) {
    __syncthreads();
    Tile<DHP>::stage(kt, k, g.ldk, row_base, t * AT_T, g.L, col0, g.dh, tid);
    __syncthreads();
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
    __syncthreads();
    Tile<DHP>::stage(kt, k, g.ldk, row_base, t * AT_T, g.L, col0, g.dh, tid);
    Tile<DHP>::stage(vt, v, g.ldv, row_base, t * AT_T, g.L, col0, g.dh, tid);
    __syncthreads();
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
        oacc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pf, Tile<DHP>::tr_frag(vt, 32 * ks, 16 * n, lane), oacc[n], 0, 0, 0);
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
}

// ---------------------------------------------------------------------------------------------------- backward: dQ (+ D)
template <int DHP>
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k,
                                                          const bf16_t* __restrict__ v, const unsigned char* __restrict__ blocked,
                                                          const bf16_t* __restrict__ o, const float* __restrict__ lse,
                                                          const bf16_t* __restrict__ dout, int lddo, bf16_t* __restrict__ dq,
                                                          int lddq, float* __restrict__ dsum, AttnGeom g, float scale) {
  constexpr int PITCH = Tile<DHP>::PITCH, NC = DHP / 16;
  __shared__ __attribute__((aligned(16))) bf16_t lds[2 * AT_T * PITCH];
  bf16_t* kt = lds;
  bf16_t* vt = lds + AT_T * PITCH;
  const int bh = blockIdx.x, b = bh / g.H, h = bh % g.H;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int q0 = blockIdx.y * AT_T + wave * 16;
  const long long row_base = (long long)b * g.L;
  const int col0 = h * g.dh;
  const int my_q = q0 + (lane & 15), grp = lane >> 4;
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
  const unsigned char* mrow = (blocked != nullptr && my_q < g.L) ? blocked + (size_t)my_q * g.L : nullptr;
  const int n_tiles = (g.L + AT_T - 1) / AT_T;
  float4v acc_dq[NC];
#pragma unroll
  for (int n = 0; n < NC; ++n) acc_dq[n] = float4v{0.f, 0.f, 0.f, 0.f};
  for (int t = 0; t < n_tiles; ++t) {
    __syncthreads();
    Tile<DHP>::stage(kt, k, g.ldk, row_base, t * AT_T, g.L, col0, g.dh, tid);
    Tile<DHP>::stage(vt, v, g.ldv, row_base, t * AT_T, g.L, col0, g.dh, tid);
    __syncthreads();
    float4v ds[4];
#pragma unroll
    for (int blk = 0; blk < 4; ++blk) {
      float4v s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < DHP / 32; ++ks) {
        s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Tile<DHP>::row_frag(kt, 16 * blk, ks, lane), qf[ks], s, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Tile<DHP>::row_frag(vt, 16 * blk, ks, lane), dof[ks], dp, 0, 0, 0);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int key = t * AT_T + 16 * blk + 4 * grp + e;
        const bool off = dead || key >= g.L || (mrow != nullptr && mrow[key]);
        const float p = off ? 0.f : exp2f(s[e] * g.scale_log2 - my_lse);
        ds[blk][e] = p * (dp[e] - d);
      }
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const short8v f = pack_p(ds[2 * ks], ds[2 * ks + 1]);
#pragma unroll
      for (int n = 0; n < NC; ++n)
        acc_dq[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f, Tile<DHP>::tr_frag(kt, 32 * ks, 16 * n, lane), acc_dq[n], 0, 0, 0);
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
}

// ---------------------------------------------------------------------------------------------------- backward: dK, dV
template <int DHP>
__global__ __launch_bounds__(256) void attn_bwd_dkv_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k,
                                                           const bf16_t* __restrict__ v, const unsigned char* __restrict__ blocked,
                                                           const float* __restrict__ lse, const float* __restrict__ dsum,
                                                           const bf16_t* __restrict__ dout, int lddo, bf16_t* __restrict__ dk,
                                                           int lddk, bf16_t* __restrict__ dv, int lddv, AttnGeom g, float scale) {
  constexpr int PITCH = Tile<DHP>::PITCH, NC = DHP / 16;
  __shared__ __attribute__((aligned(16))) bf16_t lds[2 * AT_T * PITCH];
  __shared__ float s_lse[AT_T], s_d[AT_T];
  bf16_t* qt = lds;
  bf16_t* dot = lds + AT_T * PITCH;
  const int bh = blockIdx.x, b = bh / g.H, h = bh % g.H;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int k0 = blockIdx.y * AT_T + wave * 16;  // this wave's 16 keys
  const long long row_base = (long long)b * g.L;
  const int col0 = h * g.dh;
  const int my_k = k0 + (lane & 15), grp = lane >> 4;
  short8v kf[DHP / 32], vf[DHP / 32];
  own_rows<DHP>(k, g.ldk, row_base, k0, g.L, col0, g.dh, lane, kf);
  own_rows<DHP>(v, g.ldv, row_base, k0, g.L, col0, g.dh, lane, vf);
  const int n_tiles = (g.L + AT_T - 1) / AT_T;
  float4v acc_dk[NC], acc_dv[NC];
#pragma unroll
  for (int n = 0; n < NC; ++n) acc_dk[n] = acc_dv[n] = float4v{0.f, 0.f, 0.f, 0.f};
  for (int t = 0; t < n_tiles; ++t) {
    __syncthreads();
    Tile<DHP>::stage(qt, q, g.ldq, row_base, t * AT_T, g.L, col0, g.dh, tid);
    Tile<DHP>::stage(dot, dout, lddo, row_base, t * AT_T, g.L, col0, g.dh, tid);
    if (tid < AT_T) {
      const int qq = t * AT_T + tid;
      s_lse[tid] = qq < g.L ? lse[((size_t)b * g.H + h) * g.L + qq] : -INFINITY;
      s_d[tid] = qq < g.L ? dsum[((size_t)b * g.H + h) * g.L + qq] : 0.f;
    }
    __syncthreads();
    float4v p[4], ds[4];
#pragma unroll
    for (int blk = 0; blk < 4; ++blk) {  // S / dP: row (query) = t*64 + 16*blk + 4*grp + reg, column (key) = lane&15
      float4v s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < DHP / 32; ++ks) {
        s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Tile<DHP>::row_frag(qt, 16 * blk, ks, lane), kf[ks], s, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Tile<DHP>::row_frag(dot, 16 * blk, ks, lane), vf[ks], dp, 0, 0, 0);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int ql = 16 * blk + 4 * grp + e, qq = t * AT_T + ql;
        const float ls = s_lse[ql];
        const bool off = qq >= g.L || my_k >= g.L || !(ls > -INFINITY) || (blocked != nullptr && blocked[(size_t)qq * g.L + my_k]);
        const float pe = off ? 0.f : exp2f(s[e] * g.scale_log2 - ls);
        p[blk][e] = pe;
        ds[blk][e] = pe * (dp[e] - s_d[ql]);
      }
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {  // 32 queries per k-step
      const short8v pf = pack_p(p[2 * ks], p[2 * ks + 1]), dsf = pack_p(ds[2 * ks], ds[2 * ks + 1]);
#pragma unroll
      for (int n = 0; n < NC; ++n) {
        acc_dv[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pf, Tile<DHP>::tr_frag(dot, 32 * ks, 16 * n, lane), acc_dv[n], 0, 0, 0);
        acc_dk[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dsf, Tile<DHP>::tr_frag(qt, 32 * ks, 16 * n, lane), acc_dk[n], 0, 0, 0);
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
}

static bool attn_aligned(const void* p, int ld) { return ((uintptr_t)p % 16 == 0) && ld % 8 == 0; }

}  // namespace dfine

using namespace dfine;

extern "C" int dfine_attention_fwd(const void* q, const void* k, const void* v, int ldq, int ldk, int ldv,
                                   const unsigned char* blocked, void* o, int ldo, float* lse, int B, int H, int L, int dh,
                                   float scale, void* stream) {
  if (!q || !k || !v || !o || !lse) return DFINE_E_BADARG;
  if (B <= 0 || H <= 0 || L <= 0 || dh <= 0 || ldq < H * dh || ldk < H * dh || ldv < H * dh || ldo < H * dh) return DFINE_E_BADARG;
  if (dh % 8 != 0 || dh > 64 || (long long)B * H > 0x7fffffffLL) return DFINE_E_UNSUPPORTED;
  if (!attn_aligned(q, ldq) || !attn_aligned(k, ldk) || !attn_aligned(v, ldv)) return DFINE_E_UNSUPPORTED;
  const AttnGeom g{B, H, L, dh, ldq, ldk, ldv, ldo, scale * 1.4426950408889634f};
  const dim3 grid((unsigned)(B * H), (unsigned)((L + AT_T - 1) / AT_T));
  if (dh <= 32)
    hipLaunchKernelGGL(attn_fwd_kernel<32>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)q, (const bf16_t*)k,
                       (const bf16_t*)v, blocked, (bf16_t*)o, lse, g);
  else
    hipLaunchKernelGGL(attn_fwd_kernel<64>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)q, (const bf16_t*)k,
                       (const bf16_t*)v, blocked, (bf16_t*)o, lse, g);
  DFINE_LAUNCH_CHECK();
  return 0;
}

extern "C" int dfine_attention_bwd(const void* q, const void* k, const void* v, int ldq, int ldk, int ldv,
                                   const unsigned char* blocked, const void* o, int ldo, const float* lse, const void* dout,
                                   int lddo, void* dq, int lddq, void* dk, int lddk, void* dv, int lddv, float* dsum, int B,
                                   int H, int L, int dh, float scale, void* stream) {
  if (!q || !k || !v || !o || !lse || !dout || !dq || !dk || !dv || !dsum) return DFINE_E_BADARG;
  if (B <= 0 || H <= 0 || L <= 0 || dh <= 0) return DFINE_E_BADARG;
  const int D = H * dh;
  if (ldq < D || ldk < D || ldv < D || ldo < D || lddo < D || lddq < D || lddk < D || lddv < D) return DFINE_E_BADARG;
  if (dh % 8 != 0 || dh > 64) return DFINE_E_UNSUPPORTED;
  if (!attn_aligned(q, ldq) || !attn_aligned(k, ldk) || !attn_aligned(v, ldv) || !attn_aligned(o, ldo) || !attn_aligned(dout, lddo))
    return DFINE_E_UNSUPPORTED;
  const AttnGeom g{B, H, L, dh, ldq, ldk, ldv, ldo, scale * 1.4426950408889634f};
  const dim3 grid((unsigned)(B * H), (unsigned)((L + AT_T - 1) / AT_T));
  hipStream_t st = (hipStream_t)stream;
  if (dh <= 32) {
    hipLaunchKernelGGL(attn_bwd_dq_kernel<32>, grid, dim3(256), 0, st, (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, blocked,
                       (const bf16_t*)o, lse, (const bf16_t*)dout, lddo, (bf16_t*)dq, lddq, dsum, g, scale);
    hipLaunchKernelGGL(attn_bwd_dkv_kernel<32>, grid, dim3(256), 0, st, (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, blocked,
                       lse, dsum, (const bf16_t*)dout, lddo, (bf16_t*)dk, lddk, (bf16_t*)dv, lddv, g, scale);
  } else {
    hipLaunchKernelGGL(attn_bwd_dq_kernel<64>, grid, dim3(256), 0, st, (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, blocked,
                       (const bf16_t*)o, lse, (const bf16_t*)dout, lddo, (bf16_t*)dq, lddq, dsum, g, scale);
    hipLaunchKernelGGL(attn_bwd_dkv_kernel<64>, grid, dim3(256), 0, st, (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, blocked,
                       lse, dsum, (const bf16_t*)dout, lddo, (bf16_t*)dk, lddk, (bf16_t*)dv, lddv, g, scale);
  }
  DFINE_LAUNCH_CHECK();
  return 0;
}
