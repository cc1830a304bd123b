// Weight (and bias) gradient of a Linear layer as a split-M MFMA kernel:
//     dW[n][k] += sum_m dY[m][n] * X[m][k]        db[n] += sum_m dY[m][n]
// (autograd of F.linear as used by every MLP / attention projection of the decoder and the AIFI layer,
// reference src/d_fine/arch/dfine_decoder.py:33-46,181-255 and hybrid_encoder.py:243-290).
//
// Why a kernel of our own: the reduction runs over M = tokens (15 744 for the decoder at batch 32,
// 268 800 for the encoder memory) while the output is tiny (e.g. 256 x 1024). The stock GEMM
// (hipBLASLt, no split-K chosen) covers the output with 16-64 tiles of 64x64 and so runs on 16-64 of
// the 256 CUs: 92-178 us per call, 7.2 ms per step (profiles/). Here the M axis is split across
// workgroups as well, each accumulating a 64x64 fp32 partial tile with bf16 MFMA and adding it to
// the fp32 gradient with float atomics -- straight into the flat fp32 gradient buffer, so the bf16
// grad tensor, its fp32 cast and the accumulate kernel of the stock path disappear too.
//
// Both operands are row-major with the reduction index m as the SLOW axis, while an MFMA lane wants 8
// consecutive reduction elements of one output row/column. Tiles are staged row-major in LDS and read
// back with ds_read_b64_tr_b16 (gfx950's transposing LDS read): a 16-lane group reads a 4(m) x 16(col)
// block and each lane receives one column's 4 m-values -- two such reads are one MFMA 16x16x32 fragment.
#include "common.h"

namespace dfine {

typedef short short4v __attribute__((ext_vector_type(4)));
typedef short short8v __attribute__((ext_vector_type(8)));
typedef float float4v __attribute__((ext_vector_type(4)));

constexpr int WG_TILE = 64;   // output tile edge (n and k)
constexpr int WG_MSTEP = 32;  // reduction rows per MFMA step

__device__ __forceinline__ short8v tr_frag(const bf16_t* tile, int row0, int col0, int lane) {
  // rows row0 + 8*(lane>>4) + {0..7}, column col0 + (lane&15), from a row-major [32][64] tile
  const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
  const bf16_t* a0 = tile + (row0 + 8 * g + q) * WG_TILE + col0 + 4 * p;
  typedef __attribute__((address_space(3))) short4v lds_v4;
  const short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)a0);
  const short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)(a0 + 4 * WG_TILE));
  short8v f;
  f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
  f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
  return f;
}

__global__ __launch_bounds__(256) void linear_wgrad_kernel(const bf16_t* __restrict__ dY,
                                                           const bf16_t* __restrict__ X,
                                                           float* __restrict__ dW, float* __restrict__ db,
                                                           int M, int N, int K, int ldw, int m_chunk,
                                                           int k_tiles) {
  // two LDS stages: while the MFMAs consume stage `cur`, the next 32 rows are already in flight from
  // global memory and are written to the other stage -- one barrier per 32-row step
  __shared__ __attribute__((aligned(16))) bf16_t a_tile[2][WG_MSTEP * WG_TILE];  // dY rows m, cols n
  __shared__ __attribute__((aligned(16))) bf16_t b_tile[2][WG_MSTEP * WG_TILE];  // X  rows m, cols k
  __shared__ float bias_red[WG_MSTEP][WG_TILE + 1];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n0 = (blockIdx.x / k_tiles) * WG_TILE, k0 = (blockIdx.x % k_tiles) * WG_TILE;
  const int m_begin = blockIdx.y * m_chunk, m_end = min(m_begin + m_chunk, M);
  const int r = tid >> 3, c8 = (tid & 7) * 8;  // this thread stages row r, 8 columns from c8
  const bool a_cols = n0 + c8 < N, b_cols = k0 + c8 < K;  // N, K are multiples of 8: a chunk is all in or all out
  const bool want_bias = db != nullptr && k0 == 0;

  float4v acc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) acc[j] = float4v{0.f, 0.f, 0.f, 0.f};
  float bsum[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) bsum[i] = 0.f;

  auto fetch = [&](int m0, Vec16& va, Vec16& vb) {
    const int m = m0 + r;
    va.w[0] = va.w[1] = va.w[2] = va.w[3] = 0u;
    vb = va;
    if (m < m_end) {
      if (a_cols) va = *reinterpret_cast<const Vec16*>(dY + (size_t)m * N + n0 + c8);
      if (b_cols) vb = *reinterpret_cast<const Vec16*>(X + (size_t)m * K + k0 + c8);
    }
  };
  auto stage = [&](int buf, const Vec16& va, const Vec16& vb) {
    if (want_bias) {
      float f[8];
      VecTraits<bf16_t>::unpack(va, f);
#pragma unroll
      for (int i = 0; i < 8; ++i) bsum[i] += f[i];
    }
    *reinterpret_cast<Vec16*>(a_tile[buf] + r * WG_TILE + c8) = va;
    *reinterpret_cast<Vec16*>(b_tile[buf] + r * WG_TILE + c8) = vb;
  };

  Vec16 va, vb;
  fetch(m_begin, va, vb);
  stage(0, va, vb);
  __syncthreads();
  int cur = 0;
  for (int m0 = m_begin; m0 < m_end; m0 += WG_MSTEP) {
    const bool more = m0 + WG_MSTEP < m_end;
    if (more) fetch(m0 + WG_MSTEP, va, vb);  // global loads overlap the MFMAs below
    const short8v a = tr_frag(a_tile[cur], 0, 16 * wave, lane);  // this wave's 16 output rows (n)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const short8v b = tr_frag(b_tile[cur], 0, 16 * j, lane);
      acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[j], 0, 0, 0);
    }
    if (more) stage(cur ^ 1, va, vb);  // the other stage was last read one iteration ago (barrier below)
    __syncthreads();
    cur ^= 1;
  }
  // C/D layout of mfma_f32_16x16x32: col = lane&15, row = (lane>>4)*4 + reg
  const int col = lane & 15, rbase = (lane >> 4) * 4;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int k = k0 + 16 * j + col;
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const int n = n0 + 16 * wave + rbase + reg;
      if (n < N && k < K) atomicAdd(dW + (size_t)n * ldw + k, acc[j][reg]);
    }
  }
  if (want_bias) {
#pragma unroll
    for (int i = 0; i < 8; ++i) bias_red[r][c8 + i] = bsum[i];
    __syncthreads();
    if (tid < WG_TILE && n0 + tid < N) {
      float s = 0.f;
#pragma unroll 8
      for (int rr = 0; rr < WG_MSTEP; ++rr) s += bias_red[rr][tid];
      atomicAdd(db + n0 + tid, s);
    }
  }
}

}  // namespace dfine

using namespace dfine;

extern "C" int dfine_linear_wgrad(const void* dY, const void* X, float* dW, float* db, int M, int N, int K,
                                  int ldw, void* stream) {
  if (!dY || !X || !dW) return DFINE_E_BADARG;
  if (M <= 0 || N <= 0 || K <= 0 || ldw < K) return DFINE_E_BADARG;
  if (N % 8 != 0 || K % 8 != 0) return DFINE_E_UNSUPPORTED;  // 16-byte row chunks
  const int n_tiles = (N + WG_TILE - 1) / WG_TILE, k_tiles = (K + WG_TILE - 1) / WG_TILE;
  // aim at ~4 workgroups per CU; chunks of >= 256 rows keep the atomic traffic below the MFMA time
  long long chunks = 1024 / ((long long)n_tiles * k_tiles);
  if (chunks < 1) chunks = 1;
  long long m_chunk = ((long long)M + chunks - 1) / chunks;
  m_chunk = (m_chunk + WG_MSTEP - 1) / WG_MSTEP * WG_MSTEP;
  if (m_chunk < 256) m_chunk = 256;
  const long long grid_y = ((long long)M + m_chunk - 1) / m_chunk;
  if (grid_y > 65535) return DFINE_E_UNSUPPORTED;
  hipLaunchKernelGGL(linear_wgrad_kernel, dim3(n_tiles * k_tiles, (unsigned)grid_y), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)dY, (const bf16_t*)X, dW, db, M, N, K, ldw, (int)m_chunk, k_tiles);
  DFINE_LAUNCH_CHECK();
  return 0;
}
