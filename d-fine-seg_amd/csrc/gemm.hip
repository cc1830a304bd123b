// Weight (and bias) gradients as a split-M MFMA kernel:
//     Linear:  dW[n][k] += sum_m dY[m][n] * X[m][k]        db[n] += sum_m dY[m][n]
//     Conv:    dW[n][ky][kx][c] += sum_{b,oy,ox} dY[b][oy][ox][n] * X[b][oy*s-p+ky][ox*s-p+kx][c]
// (autograd of F.linear as used by every MLP / attention projection of the decoder and the AIFI layer,
// reference src/d_fine/arch/dfine_decoder.py:33-46,181-255 and hybrid_encoder.py:243-290; autograd of the
// dense nn.Conv2d inside ConvBNAct / ConvNormLayer, hgnetv2.py:35-80 and hybrid_encoder.py:21-93).
//
// Why a kernel of our own: the reduction runs over M = tokens or pixels (15 744 decoder tokens at batch
// 32, 268 800 memory tokens, 819 200 stride-4 pixels) while the output is tiny (e.g. 256 x 1024, or
// 32 x 288). The stock GEMM (hipBLASLt, no split-K chosen) covers the output with 16-64 tiles and so runs
// on 16-64 of the 256 CUs; MIOpen's split-K wrw needs a zeroed fp32 workspace, a cast and an accumulate
// launch around it. Here the M axis is split across workgroups as well, each accumulating an fp32 partial
// tile with bf16 MFMA and adding it to the fp32 gradient with float atomics -- straight into the flat
// fp32 gradient buffer.
//
// A conv is the same GEMM with K = KH*KW*CI ("im2col" column k = (ky*KW+kx)*CI + c, which IS the memory
// order of a channels_last weight) whose X rows are gathered: column chunk -> tap is fixed per thread,
// only the pixel moves.
//
// Both operands are row-major with the reduction index m as the SLOW axis, while an MFMA lane wants 8
// consecutive reduction elements of one output row/column. Tiles are staged row-major in LDS and read
// back with ds_read_b64_tr_b16 (gfx950's transposing LDS read): a 16-lane group reads a 4(m) x 16(col)
// block and each lane receives one column's 4 m-values -- two such reads are one MFMA 16x16x32 fragment.
// Banking of that read is per 32-lane half, i.e. 8 tile rows x 32 bytes at one column offset: with plain
// 128/256-byte rows those land on 2 (1) bank slots, a 4x (8x) conflict. The 32-byte chunks of a row are
// therefore XOR-swizzled by row (swz()) so the 8 rows of a half cover the whole 256-byte bank row.
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "common.h"

namespace dfine {

typedef short short4v __attribute__((ext_vector_type(4)));
typedef short short8v __attribute__((ext_vector_type(8)));
typedef float float4v __attribute__((ext_vector_type(4)));

constexpr int WG_MSTEP = 32;  // reduction rows per MFMA step

// XOR mask for the 32-byte chunk index of tile row r (TW = tile width in elements)
template <int TW>
__device__ __forceinline__ int swz(int r) {
  if (TW == 128) return (((r >> 3) & 1) << 2) | (r & 3);
  return (((r >> 3) & 1) << 1) | ((r >> 1) & 1);
}

// rows 8*(lane>>4) + {0..7}, column col0 + (lane&15) of a swizzled row-major [32][TW] tile; col0 % 16 == 0
template <int TW>
__device__ __forceinline__ short8v tr_frag(const bf16_t* tile, int col0, int lane) {
  const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
  const int r = 8 * g + q;
  const bf16_t* a0 = tile + r * TW + (((col0 >> 4) ^ swz<TW>(r)) << 4) + 4 * p;
  typedef __attribute__((address_space(3))) short4v lds_v4;
  const short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)a0);
  const short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)(a0 + 4 * TW));  // swz(r + 4) == swz(r)
  short8v f;
  f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
  f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
  return f;
}

// 8 consecutive bf16 from a 4-byte aligned address; only the first `valid` (1..8) are read, the rest are 0
struct __attribute__((packed, aligned(4))) Vec16u {
  uint32_t w[4];
};
template <bool EDGE>
__device__ __forceinline__ Vec16 load_chunk(const bf16_t* __restrict__ p, int valid) {
  Vec16 v;
  if (!EDGE || (valid == 8 && (reinterpret_cast<uintptr_t>(p) & 3) == 0)) {
    const Vec16u u = *reinterpret_cast<const Vec16u*>(p);
    v.w[0] = u.w[0], v.w[1] = u.w[1], v.w[2] = u.w[2], v.w[3] = u.w[3];
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const uint32_t lo = 2 * i < valid ? p[2 * i] : 0u, hi = 2 * i + 1 < valid ? p[2 * i + 1] : 0u;
      v.w[i] = lo | (hi << 16);
    }
  }
  return v;
}

struct ConvGeom {
  int H, W, CI, OH, OW, KW, stride, pad;
  int ldy, ldx;  // row pitch (elements) of dY rows / X pixels (>= N / CI: channel slices of wider maps)
};

// pixel cursor of one staged row: advances by WG_MSTEP output pixels per step without dividing
struct PixelCursor {
  int b, oy, ox;
  __device__ __forceinline__ void init(long long m, const ConvGeom& g) {
    ox = (int)(m % g.OW);
    const long long t = m / g.OW;
    oy = (int)(t % g.OH);
    b = (int)(t / g.OH);
  }
  __device__ __forceinline__ void advance(const ConvGeom& g) {
    ox += WG_MSTEP;
    while (ox >= g.OW) {
      ox -= g.OW;
      if (++oy == g.OH) { oy = 0; ++b; }
    }
  }
};

// EDGE: dY rows / N not 16-byte friendly (element-wise loads where a chunk is partial or misaligned).
// XVEC: widest aligned load of X: 8 = a 16-byte chunk is 8 channels of ONE tap (CI % 8 == 0) resp. 8 columns of a Linear
// row; 4 / 1 = a chunk is assembled from 8-byte / 2-byte pieces that may belong to different taps (CI = 12, 3, 21, ...).
template <int TN, int TK, bool CONV, int DEPTH, bool EDGE, int XVEC>
__device__ __forceinline__ void wgrad_tile(const bf16_t* __restrict__ dY, const bf16_t* __restrict__ X, float* __restrict__ dW,
                                           float* __restrict__ db, int N, int K, int ldw, int m_begin, int m_end, int n0,
                                           int k0, const ConvGeom& geom) {
  constexpr int FN = TN / 32, FK = TK / 32;        // 16x16 fragments per wave (waves tile the output 2 x 2)
  constexpr int A_ROWS = TN / 64, B_ROWS = TK / 64;  // rows each thread stages per step (16-byte chunks)
  constexpr int A_CPR = TN / 8, B_CPR = TK / 8;    // 16-byte chunks per tile row
  // ONE LDS array (two stages of both tiles; reused for the bias reduction after the loop)
  __shared__ __attribute__((aligned(16))) bf16_t lds[2 * WG_MSTEP * (TN + TK)];
  bf16_t* const a_tile0 = lds;
  bf16_t* const b_tile0 = lds + 2 * WG_MSTEP * TN;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wave_n = wave >> 1, wave_k = wave & 1;
  const bool want_bias = db != nullptr && k0 == 0;

  // staging coordinates: thread -> (row, 16-byte chunk) per operand
  const int a_r = tid / A_CPR, a_j = tid % A_CPR;  // further rows at +256/A_CPR
  const int b_r = tid / B_CPR, b_j = tid % B_CPR;
  constexpr int A_RSTEP = 256 / A_CPR, B_RSTEP = 256 / B_CPR;
  // valid elements of this thread's 8-column chunks (N, K need not be multiples of 8: the last chunk of a
  // row may be partial; rows are then only 4-byte aligned, which a global dwordx4 load does not mind)
  const int a_valid = min(max(N - (n0 + a_j * 8), 0), 8), b_valid = min(max(K - (k0 + b_j * 8), 0), 8);
  int tap_dy = 0, tap_dx = 0, chan = 0;
  // EDGE conv builds (CI % 8 != 0: the 3-channel image, 12 / 21-channel maps): a chunk's 8 columns can belong to
  // different taps, so every element carries its own (tap offset, channel)
  constexpr int NE = (CONV && XVEC < 8) ? 8 / XVEC : 1;  // pieces per chunk
  int e_dy[NE], e_dx[NE], e_ch[NE];
  PixelCursor cur_px[B_ROWS];
  if (CONV) {
    const int kk = k0 + b_j * 8;
    const int tap = kk / geom.CI;
    chan = kk % geom.CI;
    tap_dy = tap / geom.KW - geom.pad;
    tap_dx = tap % geom.KW - geom.pad;
    if (XVEC < 8) {
#pragma unroll
      for (int e = 0; e < NE; ++e) {
        const int k = kk + e * XVEC, t = k / geom.CI;
        e_ch[e] = k < K ? k % geom.CI : -1;
        e_dy[e] = t / geom.KW - geom.pad;
        e_dx[e] = t % geom.KW - geom.pad;
      }
    }
#pragma unroll
    for (int i = 0; i < B_ROWS; ++i) cur_px[i].init((long long)m_begin + b_r + i * B_RSTEP, geom);
  }

  float4v acc[FN][FK];
#pragma unroll
  for (int i = 0; i < FN; ++i)
#pragma unroll
    for (int j = 0; j < FK; ++j) acc[i][j] = float4v{0.f, 0.f, 0.f, 0.f};
  float bsum[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) bsum[i] = 0.f;

  // Register ring of DEPTH future steps: the loads of step s+DEPTH are issued while step s is computed,
  // so DEPTH * (TN + TK) * 64 bytes per workgroup are in flight -- one step ahead is not enough to cover
  // the ~2 us a loaded HBM/Infinity-Cache round trip takes (measured: 380 TFLOP/s, time per step = latency).
  Vec16 va[DEPTH][A_ROWS], vb[DEPTH][B_ROWS];
  auto fetch = [&](int slot, int m0) {
#pragma unroll
    for (int i = 0; i < A_ROWS; ++i) {
      const int m = m0 + a_r + i * A_RSTEP;
      va[slot][i].w[0] = va[slot][i].w[1] = va[slot][i].w[2] = va[slot][i].w[3] = 0u;
      if (m < m_end && a_valid > 0) va[slot][i] = load_chunk<EDGE>(dY + (size_t)m * geom.ldy + n0 + a_j * 8, a_valid);
    }
#pragma unroll
    for (int i = 0; i < B_ROWS; ++i) {
      const int m = m0 + b_r + i * B_RSTEP;
      vb[slot][i].w[0] = vb[slot][i].w[1] = vb[slot][i].w[2] = vb[slot][i].w[3] = 0u;
      if (CONV && XVEC < 8) {
        if (m < m_end && b_valid > 0) {
          const int oy = cur_px[i].oy * geom.stride, ox = cur_px[i].ox * geom.stride;
#pragma unroll
          for (int e = 0; e < NE; ++e) {
            const int iy = oy + e_dy[e], ix = ox + e_dx[e];
            if (e_ch[e] >= 0 && (unsigned)iy < (unsigned)geom.H && (unsigned)ix < (unsigned)geom.W) {
              const bf16_t* src = X + (((size_t)cur_px[i].b * geom.H + iy) * geom.W + ix) * geom.ldx + e_ch[e];
              if (XVEC == 4) {
                const uint2 v = *reinterpret_cast<const uint2*>(src);
                vb[slot][i].w[2 * e] = v.x, vb[slot][i].w[2 * e + 1] = v.y;
              } else {
                vb[slot][i].w[e >> 1] |= (uint32_t)(*src) << (16 * (e & 1));
              }
            }
          }
        }
        cur_px[i].advance(geom);
      } else if (CONV) {
        const int iy = cur_px[i].oy * geom.stride + tap_dy, ix = cur_px[i].ox * geom.stride + tap_dx;
        if (m < m_end && b_valid > 0 && (unsigned)iy < (unsigned)geom.H && (unsigned)ix < (unsigned)geom.W)
          vb[slot][i] = load_chunk<false>(X + (((size_t)cur_px[i].b * geom.H + iy) * geom.W + ix) * geom.ldx + chan, 8);
        cur_px[i].advance(geom);
      } else {
        if (m < m_end && b_valid > 0) vb[slot][i] = load_chunk<(XVEC < 8)>(X + (size_t)m * geom.ldx + k0 + b_j * 8, b_valid);
      }
    }
  };
  auto stage = [&](int buf, int slot) {
    bf16_t* at = a_tile0 + buf * WG_MSTEP * TN;
    bf16_t* bt = b_tile0 + buf * WG_MSTEP * TK;
#pragma unroll
    for (int i = 0; i < A_ROWS; ++i) {
      if (want_bias) {
        float f[8];
        VecTraits<bf16_t>::unpack(va[slot][i], f);
#pragma unroll
        for (int e = 0; e < 8; ++e) bsum[e] += f[e];
      }
      const int r = a_r + i * A_RSTEP;
      *reinterpret_cast<Vec16*>(at + r * TN + ((((a_j >> 1) ^ swz<TN>(r)) << 1) | (a_j & 1)) * 8) = va[slot][i];
    }
#pragma unroll
    for (int i = 0; i < B_ROWS; ++i) {
      const int r = b_r + i * B_RSTEP;
      *reinterpret_cast<Vec16*>(bt + r * TK + ((((b_j >> 1) ^ swz<TK>(r)) << 1) | (b_j & 1)) * 8) = vb[slot][i];
    }
  };

  // Two LDS stages, one barrier per 32-row step, and the MFMA operands double-buffered in registers: while
  // the MFMAs of step s run, the ds_read_b64_tr_b16 of step s+1 are already in flight (their latency was
  // exposed at the top of every step before). Ring slot k holds a step == k (mod DEPTH); the loop is
  // unrolled by DEPTH (even) so slots, LDS stages and fragment buffers are compile-time indices. Rows at or
  // beyond m_end are fetched as zeros, so every workgroup simply runs a multiple of DEPTH steps.
  const int n_steps = (m_end - m_begin + WG_MSTEP * DEPTH - 1) / (WG_MSTEP * DEPTH) * DEPTH;
  short8v fa[2][FN], fb[2][FK];
  auto read_frags = [&](int buf, int stage_idx) {
    const bf16_t* at = a_tile0 + stage_idx * WG_MSTEP * TN;
    const bf16_t* bt = b_tile0 + stage_idx * WG_MSTEP * TK;
#pragma unroll
    for (int i = 0; i < FN; ++i) fa[buf][i] = tr_frag<TN>(at, wave_n * (TN / 2) + 16 * i, lane);
#pragma unroll
    for (int j = 0; j < FK; ++j) fb[buf][j] = tr_frag<TK>(bt, wave_k * (TK / 2) + 16 * j, lane);
  };
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) fetch(d, m_begin + WG_MSTEP * d);
  stage(0, 0);  // step 0
  fetch(0, m_begin + WG_MSTEP * DEPTH);
  __syncthreads();
  read_frags(0, 0);
  stage(1, 1 % DEPTH);  // step 1
  fetch(1 % DEPTH, m_begin + WG_MSTEP * (DEPTH + 1));
  __syncthreads();
  for (int s0 = 0; s0 < n_steps; s0 += DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      // step s = s0 + d: its operands are in fragment buffer d&1; prefetch step s+1 from the other LDS stage
      read_frags((d + 1) & 1, (d + 1) & 1);
#pragma unroll
      for (int i = 0; i < FN; ++i)
#pragma unroll
        for (int j = 0; j < FK; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[d & 1][i], fb[d & 1][j], acc[i][j], 0, 0, 0);
      // step s+2 (ring slot (d+2) % DEPTH, loaded DEPTH steps ago) goes to LDS stage d&1: that stage held step s,
      // whose fragments every wave read one iteration ago (before the previous barrier)
      const int slot = (d + 2) % DEPTH;
      stage(d & 1, slot);
      fetch(slot, m_begin + WG_MSTEP * (s0 + d + 2 + DEPTH));
      __syncthreads();
    }
  }
  // C/D layout of mfma_f32_16x16x32: col = lane&15, row = (lane>>4)*4 + reg
  const int col = lane & 15, rbase = (lane >> 4) * 4;
#pragma unroll
  for (int i = 0; i < FN; ++i)
#pragma unroll
    for (int j = 0; j < FK; ++j) {
      const int k = k0 + wave_k * (TK / 2) + 16 * j + col;
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int n = n0 + wave_n * (TN / 2) + 16 * i + rbase + reg;
        if (n < N && k < K) atomicAdd(dW + (size_t)n * ldw + k, acc[i][j][reg]);
      }
    }
  if (want_bias) {  // the loop ended on a barrier: the tile memory is free
    float* red = reinterpret_cast<float*>(lds);
    if (tid < TN) red[tid] = 0.f;
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 8; ++e) atomicAdd(red + a_j * 8 + e, bsum[e]);
    __syncthreads();
    if (tid < TN && n0 + tid < N) atomicAdd(db + n0 + tid, red[tid]);
  }
}

// One problem per launch: workgroup -> (m-chunk, output tile).
template <int TN, int TK, bool CONV, int DEPTH, bool EDGE, int XVEC>
__global__ __launch_bounds__(256, 2) void wgrad_kernel(const bf16_t* __restrict__ dY, const bf16_t* __restrict__ X,
                                                    float* __restrict__ dW, float* __restrict__ db, int M, int N,
                                                    int K, int ldw, int m_chunk, int n_chunks, int k_tiles,
                                                    int tiles, ConvGeom geom) {
  // XCD-aware decode: consecutive workgroup ids go round-robin to the 8 XCDs; all tiles of one m-chunk
  // share the chunk's dY/X rows, so they are given ids of one residue class (one XCD, one L2)
  const int id = blockIdx.x, xcd = id & 7, slot = id >> 3;
  const int chunk = xcd + 8 * (slot / tiles), tile = slot % tiles;
  if (chunk >= n_chunks) return;  // whole workgroup, before any barrier
  const int m_begin = chunk * m_chunk;
  wgrad_tile<TN, TK, CONV, DEPTH, EDGE, XVEC>(dY, X, dW, db, N, K, ldw, m_begin, min(m_begin + m_chunk, M), (tile / k_tiles) * TN,
                                              (tile % k_tiles) * TK, geom);
}

// Many problems per launch (the deferred weight gradients of a whole backward pass): a block table built on the host maps
// workgroup -> (problem, m-chunk, tile); see dfine_wgrad_group_plan. Entries with problem < 0 pad the XCD interleave.
struct WgradProblem {
  const bf16_t* dY;
  const bf16_t* X;
  float* dW;
  float* db;
  int M, N, K, ldw;
  int m_chunk, k_tiles, variant, n_chunks;
  ConvGeom geom;
  int pad[2];
};
static_assert(sizeof(WgradProblem) == 112, "table layout");

template <int TN, int TK, bool CONV, bool NARROW = false>
__global__ __launch_bounds__(256, 2) void wgrad_group_kernel(const WgradProblem* __restrict__ problems, const int4* __restrict__ blocks) {
  const int4 e = blocks[blockIdx.x];
  if (e.x < 0) return;
  const WgradProblem* p = problems + e.x;
  const int M = p->M, m_chunk = p->m_chunk, k_tiles = p->k_tiles;
  const ConvGeom geom = p->geom;
  const int m_begin = e.y * m_chunk;
  wgrad_tile<TN, TK, CONV, 4, NARROW, NARROW ? 1 : 8>(p->dY, p->X, p->dW, p->db, p->N, p->K, p->ldw, m_begin, min(m_begin + m_chunk, M),
                                                      (e.z / k_tiles) * TN, (e.z % k_tiles) * TK, geom);
}

// a 128-wide tile only where it wastes no more padding than 64-wide tiles do
static inline int tile_width(int n) { return ((n + 127) / 128 * 128 == (n + 63) / 64 * 64) ? 128 : 64; }

template <bool CONV, int DEPTH, bool EDGE, int XVEC>
static int launch_wgrad(const void* dY, const void* X, float* dW, float* db, int M, int N, int K, int ldw,
                        const ConvGeom& geom, hipStream_t stream) {
  // Split of the M axis: all workgroups do the same amount of work, so their number should fill the
  // chip's resident capacity (256 CUs x the tile shape's occupancy) ONCE -- 1.5 rounds cost as much as 2.
  // Never more partial tiles than keeps the atomic traffic (chunks * N*K*4 bytes) below the operand
  // traffic (M*(N+K)*2 bytes); chunks of at least 256 rows; a multiple of 8 chunks (one residue class
  // per XCD). Tile shape: the largest that still fills half the capacity (128-wide tiles halve the LDS
  // and L2 reads per MFMA but a small problem needs the parallelism more).
  static const int atomic_w = getenv("DFINE_WGRAD_ATOMIC_W") ? atoi(getenv("DFINE_WGRAD_ATOMIC_W")) : 2;
  static const int fill_div = getenv("DFINE_WGRAD_FILL_DIV") ? atoi(getenv("DFINE_WGRAD_FILL_DIV")) : 2;
  long long by_traffic = ((long long)M * (N + K)) / ((long long)atomic_w * N * K);
  if (by_traffic > (M + 255) / 256) by_traffic = (M + 255) / 256;
  if (by_traffic < 1) by_traffic = 1;
  const int want_n = tile_width(N), want_k = tile_width(K);
  int cand[3][2] = {{want_n, want_k}, {64, 64}, {64, 64}};
  if (want_n == 128 && want_k == 128) cand[1][N >= K ? 0 : 1] = 128;  // halve the shorter side first
  int tn = 64, tk = 64, n_tiles = 0, k_tiles = 0;
  long long tiles = 0, chunks = 1;
  for (int c = 0; c < 3; ++c) {
    tn = cand[c][0], tk = cand[c][1];
    n_tiles = (N + tn - 1) / tn, k_tiles = (K + tk - 1) / tk;
    tiles = (long long)n_tiles * k_tiles;
    const long long capacity = 256LL * (tn + tk == 256 ? 2 : tn + tk == 192 ? 3 : 4);  // __launch_bounds__(256, 2) builds
    chunks = capacity / tiles;
    if (chunks > by_traffic) chunks = by_traffic;
    if (chunks >= 8) chunks = chunks / 8 * 8;
    if (chunks < 1) chunks = 1;
    if (tiles * chunks * fill_div >= capacity) break;
  }
  long long m_chunk = ((long long)M + chunks - 1) / chunks;
  m_chunk = (m_chunk + WG_MSTEP * DEPTH - 1) / (WG_MSTEP * DEPTH) * (WG_MSTEP * DEPTH);
  const long long n_chunks = ((long long)M + m_chunk - 1) / m_chunk;
  const long long grid = (n_chunks + 7) / 8 * 8 * tiles;
  if (grid > 0x7fffffffLL) return DFINE_E_UNSUPPORTED;
#define DFINE_WGRAD_LAUNCH(TN_, TK_)                                                                          \
  hipLaunchKernelGGL((wgrad_kernel<TN_, TK_, CONV, DEPTH, EDGE, XVEC>), dim3((unsigned)grid), dim3(256), 0, stream,        \
                     (const bf16_t*)dY, (const bf16_t*)X, dW, db, M, N, K, ldw, (int)m_chunk, (int)n_chunks, \
                     k_tiles, (int)tiles, geom)
  if (tn == 128 && tk == 128) DFINE_WGRAD_LAUNCH(128, 128);
  else if (tn == 128) DFINE_WGRAD_LAUNCH(128, 64);
  else if (tk == 128) DFINE_WGRAD_LAUNCH(64, 128);
  else DFINE_WGRAD_LAUNCH(64, 64);
#undef DFINE_WGRAD_LAUNCH
  DFINE_LAUNCH_CHECK();
  return 0;
}

// ---- grouped launch: host-side planning ------------------------------------------------------------------------------
// kernel variants of a grouped launch: bit 0 TK = 128, bit 1 TN = 128, bit 2 gathered conv rows; 8 = Linear problems that are
// not 16-byte friendly (N or K not a multiple of 8, odd pitches: the 4- / 132-wide heads), 64 x 64 tiles with narrow loads
constexpr int GROUP_VARIANTS = 9, GROUP_NARROW = 8;
struct GroupLaunch {
  int variant, grid;
  long long blocks_off;
};
struct GroupHeader {
  int magic, n_problems, n_launches, pad;
  long long problems_off, total_bytes;
  GroupLaunch launch[GROUP_VARIANTS];
};
constexpr int GROUP_MAGIC = 0x57475250;  // "WGRP"

// public problem -> device descriptor (without the split); false: not servable by the aligned (16-byte loads) build
static bool group_normalise(const dfine_wgrad_problem& q, WgradProblem* out) {
  WgradProblem w{};
  if (!q.dY || !q.X || !q.dW) return false;
  w.dY = (const bf16_t*)q.dY, w.X = (const bf16_t*)q.X, w.dW = q.dW, w.db = q.db;
  bool conv = false;
  if (q.conv) {
    if (q.B <= 0 || q.H <= 0 || q.W <= 0 || q.CI <= 0 || q.OH <= 0 || q.OW <= 0 || q.CO <= 0 || q.KH <= 0 || q.KW <= 0 ||
        q.stride <= 0 || q.pad < 0 || q.ldy < q.CO || q.ldx < q.CI || q.db)
      return false;
    if ((q.OH - 1) * q.stride - q.pad >= q.H || (q.OW - 1) * q.stride - q.pad >= q.W) return false;
    const long long M = (long long)q.B * q.OH * q.OW;
    if (M > 0x7fffffffLL - 4096 || (long long)q.B * q.H * q.W * q.ldx > 0x7fffffffffLL) return false;
    w.M = (int)M, w.N = q.CO, w.K = q.KH * q.KW * q.CI, w.ldw = w.K;
    w.geom = ConvGeom{q.H, q.W, q.CI, q.OH, q.OW, q.KW, q.stride, q.pad, q.ldy, q.ldx};
    conv = !(q.KH == 1 && q.KW == 1 && q.stride == 1 && q.pad == 0 && q.OH == q.H && q.OW == q.W);
  } else {
    if (q.M <= 0 || q.N <= 0 || q.K <= 0 || q.ldw < q.K || q.ldy < q.N || q.ldx < q.K || q.M > 0x7fffffff - 4096) return false;
    w.M = q.M, w.N = q.N, w.K = q.K, w.ldw = q.ldw;
    w.geom = ConvGeom{};
    w.geom.ldy = q.ldy, w.geom.ldx = q.ldx;
  }
  const bool wide = w.N % 8 == 0 && q.ldy % 8 == 0 && (uintptr_t)q.dY % 16 == 0 && q.ldx % 8 == 0 && (uintptr_t)q.X % 16 == 0 &&
                    (q.conv ? q.CI % 8 == 0 : q.K % 8 == 0);
  if (!wide && conv) return false;  // gathered rows with narrow loads (the 3-channel stem): single-problem launches only
  if (!wide && ((uintptr_t)q.dY % 2 != 0 || (uintptr_t)q.X % 2 != 0)) return false;
  const int tn = wide ? tile_width(w.N) : 64, tk = wide ? tile_width(w.K) : 64;
  w.variant = wide ? ((conv ? 4 : 0) | (tn == 128 ? 2 : 0) | (tk == 128 ? 1 : 0)) : GROUP_NARROW;
  w.k_tiles = (w.K + tk - 1) / tk;
  *out = w;
  return true;
}

static int env_int(const char* name, int dflt) {
  const char* e = getenv(name);
  return e ? atoi(e) : dflt;
}

}  // namespace dfine

using namespace dfine;

extern "C" int dfine_wgrad_group_eligible(const dfine_wgrad_problem* problem) {
  WgradProblem w;
  return problem && group_normalise(*problem, &w) ? 1 : 0;
}

extern "C" long long dfine_wgrad_group_plan(const dfine_wgrad_problem* problems, int n, void* table, long long capacity) {
  if (!problems || n <= 0) return DFINE_E_BADARG;
  // Split of the row axis. With hundreds of problems in one launch the grid is full whatever the split, so chunks are LONG:
  // the fp32 atomics of a partial tile (chunks * N*K*4 bytes at ~1.3 TB/s) stay below 1/16 of the operand bytes
  // (M*(N+K)*2 at ~5 TB/s), and a workgroup's prologue / epilogue is amortised over 32..ROWS_MAX/32 steps.
  static const int rows_max = env_int("DFINE_WGRAD_GROUP_ROWS_MAX", 4096), rows_min = env_int("DFINE_WGRAD_GROUP_ROWS_MIN", 1024);
  static const int atomic_w = env_int("DFINE_WGRAD_GROUP_ATOMIC_W", 32);
  std::vector<WgradProblem> ps(n);
  struct Group { int problem, chunk, tiles, steps; };
  std::vector<Group> groups[GROUP_VARIANTS];
  for (int i = 0; i < n; ++i) {
    WgradProblem& w = ps[i];
    if (!group_normalise(problems[i], &w)) return DFINE_E_UNSUPPORTED;
    const int tn = (w.variant != GROUP_NARROW && (w.variant & 2)) ? 128 : 64, tk = (w.variant != GROUP_NARROW && (w.variant & 1)) ? 128 : 64;
    const int tiles = ((w.N + tn - 1) / tn) * w.k_tiles;
    long long chunks = ((long long)w.M * (w.N + w.K)) / ((long long)atomic_w * w.N * w.K);
    const long long lo = ((long long)w.M + rows_max - 1) / rows_max, hi = ((long long)w.M + rows_min - 1) / rows_min;
    if (chunks < lo) chunks = lo;
    if (chunks > hi) chunks = hi;
    if (chunks < 1) chunks = 1;
    long long m_chunk = ((long long)w.M + chunks - 1) / chunks;
    m_chunk = (m_chunk + WG_MSTEP * 4 - 1) / (WG_MSTEP * 4) * (WG_MSTEP * 4);
    w.m_chunk = (int)m_chunk;
    w.n_chunks = (int)(((long long)w.M + m_chunk - 1) / m_chunk);
    for (int c = 0; c < w.n_chunks; ++c) groups[w.variant].push_back(Group{i, c, tiles, (int)(m_chunk / WG_MSTEP)});
  }
  // Block tables, one per kernel variant. Workgroup ids go round-robin to the 8 XCDs: the tiles of one (problem, chunk)
  // share that chunk's dY / X rows, so they get consecutive slots of ONE XCD (one L2); groups go to the least loaded XCD,
  // longest first.
  GroupHeader h{};
  h.magic = GROUP_MAGIC, h.n_problems = n;
  h.problems_off = sizeof(GroupHeader);
  long long off = h.problems_off + (long long)n * sizeof(WgradProblem);
  std::vector<std::vector<int4>> tables;
  for (int v = 0; v < GROUP_VARIANTS; ++v) {
    std::vector<Group>& g = groups[v];
    if (g.empty()) continue;
    std::stable_sort(g.begin(), g.end(), [](const Group& a, const Group& b) { return a.steps > b.steps; });
    std::vector<int4> seq[8];
    long long load[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (const Group& gr : g) {
      int x = 0;
      for (int j = 1; j < 8; ++j)
        if (load[j] < load[x]) x = j;
      for (int t = 0; t < gr.tiles; ++t) seq[x].push_back(int4{gr.problem, gr.chunk, t, 0});
      load[x] += (long long)gr.tiles * gr.steps;
    }
    size_t len = 0;
    for (int x = 0; x < 8; ++x) len = std::max(len, seq[x].size());
    if (len * 8 > 0x7fffffffULL) return DFINE_E_UNSUPPORTED;
    std::vector<int4> tab(len * 8, int4{-1, 0, 0, 0});
    for (int x = 0; x < 8; ++x)
      for (size_t s = 0; s < seq[x].size(); ++s) tab[s * 8 + x] = seq[x][s];
    GroupLaunch& L = h.launch[h.n_launches++];
    L.variant = v, L.grid = (int)tab.size(), L.blocks_off = off;
    off += (long long)tab.size() * sizeof(int4);
    tables.push_back(std::move(tab));
  }
  h.total_bytes = off;
  if (!table) return off;
  if (capacity < off) return DFINE_E_BADARG;
  char* out = (char*)table;
  memcpy(out, &h, sizeof(h));
  memcpy(out + h.problems_off, ps.data(), (size_t)n * sizeof(WgradProblem));
  for (int l = 0; l < h.n_launches; ++l) memcpy(out + h.launch[l].blocks_off, tables[l].data(), tables[l].size() * sizeof(int4));
  return off;
}

extern "C" int dfine_wgrad_group_launch(const void* table_host, const void* table_dev, void* stream) {
  if (!table_host || !table_dev) return DFINE_E_BADARG;
  const GroupHeader* h = (const GroupHeader*)table_host;
  if (h->magic != GROUP_MAGIC || h->n_launches < 0 || h->n_launches > GROUP_VARIANTS) return DFINE_E_BADARG;
  const char* dev = (const char*)table_dev;
  const WgradProblem* problems = (const WgradProblem*)(dev + h->problems_off);
  hipStream_t st = (hipStream_t)stream;
  for (int l = 0; l < h->n_launches; ++l) {
    const GroupLaunch& L = h->launch[l];
    const int4* blocks = (const int4*)(dev + L.blocks_off);
#define DFINE_GROUP_LAUNCH(TN_, TK_, CONV_) \
  hipLaunchKernelGGL((wgrad_group_kernel<TN_, TK_, CONV_>), dim3((unsigned)L.grid), dim3(256), 0, st, problems, blocks)
    switch (L.variant) {
      case 0: DFINE_GROUP_LAUNCH(64, 64, false); break;
      case 1: DFINE_GROUP_LAUNCH(64, 128, false); break;
      case 2: DFINE_GROUP_LAUNCH(128, 64, false); break;
      case 3: DFINE_GROUP_LAUNCH(128, 128, false); break;
      case 4: DFINE_GROUP_LAUNCH(64, 64, true); break;
      case 5: DFINE_GROUP_LAUNCH(64, 128, true); break;
      case 6: DFINE_GROUP_LAUNCH(128, 64, true); break;
      case 7: DFINE_GROUP_LAUNCH(128, 128, true); break;
      case GROUP_NARROW:
        hipLaunchKernelGGL((wgrad_group_kernel<64, 64, false, true>), dim3((unsigned)L.grid), dim3(256), 0, st, problems, blocks);
        break;
      default: return DFINE_E_BADARG;
    }
#undef DFINE_GROUP_LAUNCH
    DFINE_LAUNCH_CHECK();
  }
  return 0;
}

// `edge`: dY side (N, ldy, base) not 16-byte friendly; `xvec`: widest aligned piece of an X chunk (8, 4 or 1 elements)
template <bool CONV>
static int dispatch_wgrad(const void* dY, const void* X, float* dW, float* db, int M, int N, int K, int ldw,
                          const ConvGeom& geom, bool edge, int xvec, hipStream_t stream) {
  if (xvec == 8)
    return edge ? launch_wgrad<CONV, 4, true, 8>(dY, X, dW, db, M, N, K, ldw, geom, stream)
                : launch_wgrad<CONV, 4, false, 8>(dY, X, dW, db, M, N, K, ldw, geom, stream);
  if (CONV && xvec == 4)
    return edge ? launch_wgrad<CONV, 4, true, CONV ? 4 : 1>(dY, X, dW, db, M, N, K, ldw, geom, stream)
                : launch_wgrad<CONV, 4, false, CONV ? 4 : 1>(dY, X, dW, db, M, N, K, ldw, geom, stream);
  return edge ? launch_wgrad<CONV, 4, true, 1>(dY, X, dW, db, M, N, K, ldw, geom, stream)
              : launch_wgrad<CONV, 4, false, 1>(dY, X, dW, db, M, N, K, ldw, geom, stream);
}

extern "C" int dfine_linear_wgrad(const void* dY, const void* X, float* dW, float* db, int M, int N, int K,
                                  int ldw, int ldy, int ldx, void* stream) {
  if (!dY || !X || !dW) return DFINE_E_BADARG;
  if (M <= 0 || N <= 0 || K <= 0 || ldw < K || ldy < N || ldx < K) return DFINE_E_BADARG;
  ConvGeom geom{};
  geom.ldy = ldy, geom.ldx = ldx;
  const bool edge = !(N % 8 == 0 && ldy % 8 == 0 && (uintptr_t)dY % 16 == 0);
  const int xvec = (K % 8 == 0 && ldx % 8 == 0 && (uintptr_t)X % 16 == 0) ? 8 : 1;
  return dispatch_wgrad<false>(dY, X, dW, db, M, N, K, ldw, geom, edge, xvec, (hipStream_t)stream);
}

extern "C" int dfine_conv_wgrad(const void* dY, const void* X, float* dW, int B, int H, int W, int CI, int OH,
                                int OW, int CO, int KH, int KW, int stride, int pad, int ldy, int ldx, void* stream) {
  if (!dY || !X || !dW) return DFINE_E_BADARG;
  if (B <= 0 || H <= 0 || W <= 0 || CI <= 0 || OH <= 0 || OW <= 0 || CO <= 0 || KH <= 0 || KW <= 0 || stride <= 0 ||
      pad < 0 || ldy < CO || ldx < CI)
    return DFINE_E_BADARG;
  // the last output's first tap must start inside the (top/left padded) input: bottom/right padding is implied by OH/OW
  if ((OH - 1) * stride - pad >= H || (OW - 1) * stride - pad >= W) return DFINE_E_BADARG;
  const long long M = (long long)B * OH * OW;
  if (M > 0x7fffffffLL - 64 || (long long)B * H * W * ldx > 0x7fffffffffLL) return DFINE_E_UNSUPPORTED;
  ConvGeom geom{H, W, CI, OH, OW, KW, stride, pad, ldy, ldx};
  const int K = KH * KW * CI;
  const bool edge = !(CO % 8 == 0 && ldy % 8 == 0 && (uintptr_t)dY % 16 == 0);
  int xvec = 1;
  if (CI % 8 == 0 && ldx % 8 == 0 && (uintptr_t)X % 16 == 0) xvec = 8;
  else if (CI % 4 == 0 && ldx % 4 == 0 && (uintptr_t)X % 8 == 0) xvec = 4;
  if (KH == 1 && KW == 1 && stride == 1 && pad == 0 && OH == H && OW == W)  // rows are not gathered: the Linear path
    return dispatch_wgrad<false>(dY, X, dW, nullptr, (int)M, CO, K, K, geom, edge, xvec == 8 ? 8 : 1, (hipStream_t)stream);
  return dispatch_wgrad<true>(dY, X, dW, nullptr, (int)M, CO, K, K, geom, edge, xvec, (hipStream_t)stream);
}
