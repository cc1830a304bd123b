// Implicit-GEMM convolution / Linear on bf16 MFMA (gfx950), forward and input gradient:
//
//     C[m][n] = act( sum_k A(m, k) * Bw[n][k] + bias[n] )          fp32 accumulate, bf16 out
//
// One kernel family serves every dense contraction of the step:
//   * Linear forward       y = x W^T + b           A(m,k) = x[m][k],         Bw = W [N][K]
//   * Linear input grad    dx = dy W               A(m,k) = dy[m][k],        Bw = W^T [K_in][N_out]
//   * conv forward         (HGNetv2 ConvBNAct hgnetv2.py:35-80, hybrid_encoder ConvNormLayer :21-45)
//                          A(m,k) = X[b][py*s + dy(tap)][px*s + dx(tap)][c], k = tap*CI + c, Bw = W [CO][KH][KW][CI]
//   * conv input grad      the same gather over dY with mirrored taps and Bw = W^T [CI][KH][KW][CO];
//                          a strided conv's gradient is launched once per input-parity class
//   * ConvTranspose2d 2x2 s2 (MaskPixelDecoder.upconv, dfine_decoder.py:316-376): four 1-tap launches that
//     write interleaved output pixels; its input gradient is a 4-tap gather with input stride 2
// A "tap" is a (dy, dx, weight-slot) triple from a small table, so asymmetric padding (the stem's
// F.pad(...,(0,1,0,1)) + 2x2 conv, hgnetv2.py:156-166) needs no padded copy: out-of-range taps read zeros.
//
// Epilogue (the reason this is not a library call): bias + ReLU, the per-channel sum / sum of squares of
// the bf16-rounded outputs for the BatchNorm that follows every conv (fp64 atomics into the same replica
// layout dfine_bn_act_fwd reads, so its statistics pass disappears), and an output row pitch `ldc` so a
// layer can write straight into its slice of a concat buffer. `lda` likewise lets a layer read a channel
// slice of a wider map.
//
// Mapping: workgroup = 256 threads = 2 x 2 waves, tile BM x BN = 32*FM x 32*FN, K step BK. Both operands are
// K-contiguous, so MFMA 16x16x32 fragments are plain 16-byte LDS reads (row lane&15, 16-byte chunk lane>>4);
// the 16-byte chunks of a tile row are XOR-swizzled by row so that every ds_read_b128 lane group covers all
// 16 slots of the 256-byte bank row (checked exhaustively: conflict-free for BK = 32 and 64). The weight
// tile is MFMA operand A and the pixel tile operand B: an accumulator register quad then holds 4 CONSECUTIVE
// output channels of one pixel, the tile goes through LDS once as packed 8-byte writes and leaves as 16-byte
// row-contiguous stores. Global -> LDS staging is through registers (the gather zero-fills padding taps and
// rows beyond M), one K step ahead, one barrier per K step. Tiles of one M range share an XCD (its L2 holds
// the activation rows all N tiles re-read and the halo rows neighbouring M tiles share).
#include <stdlib.h>

#include "common.h"
#include "igemm_bnred.h"
#include "igemm_epilogue.h"

namespace dfine {

template <int BK>
__device__ __forceinline__ int ig_swz(int r) {
  if (BK == 64) return (r >> 1) & 7;
  return (0x1230 >> (4 * ((r >> 2) & 3))) & 3;  // {0, 3, 2, 1}
}

// VEC consecutive bf16 (VEC*2 bytes, naturally aligned) -> the low VEC halves of dst[]
template <int VEC>
__device__ __forceinline__ void ig_load(const bf16_t* __restrict__ p, uint32_t* dst) {
  if (VEC == 8) {
    const uint4 v = *reinterpret_cast<const uint4*>(p);
    dst[0] = v.x, dst[1] = v.y, dst[2] = v.z, dst[3] = v.w;
  } else if (VEC == 4) {
    const uint2 v = *reinterpret_cast<const uint2*>(p);
    dst[0] = v.x, dst[1] = v.y;
  }
}

struct IgRow {  // one staged pixel row: decoded once, constant over the K loop
  int base;     // gather: pixel index of the image's first pixel; plain: element offset of the row
  int iy0, ix0;
  bool ok;
};

template <int FM, int FN, int BK, int VEC, int OVEC, bool GATHER, bool BNRED = false>
__global__ __launch_bounds__(256, 2) void igemm_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ Bw,
                                                      bf16_t* __restrict__ C, const float* __restrict__ bias,
                                                      double* __restrict__ stats, const IGemmGeom g, int m_tiles,
                                                      int n_tiles, int m_per_xcd, const IgBnRed br) {
  constexpr int BM = 32 * FM, BN = 32 * FN;
  constexpr int CPR = BK / 8;                 // 16-byte chunks per tile row
  constexpr int RPP = 256 / CPR;              // rows staged per pass of the 256 threads
  constexpr int A_CH = BM / RPP, B_CH = (BN + RPP - 1) / RPP;  // chunks per thread per K step
  constexpr int STAGE = (BM + BN) * BK;       // bf16 elements per LDS stage
  constexpr int CPITCH = BN + 8;              // epilogue tile row pitch (elements): 16 bytes of padding
  constexpr int LDS_MAIN = 2 * STAGE * 2, LDS_EPI = BM * CPITCH * 2 + 2 * (256 / (BN / 8)) * BN * 4;  // C tile + statistics partials
  __shared__ __attribute__((aligned(16))) unsigned char lds_raw[(LDS_MAIN > LDS_EPI ? LDS_MAIN : LDS_EPI) + 64];
  bf16_t* const lds = reinterpret_cast<bf16_t*>(lds_raw);
  int* const tap_tab = reinterpret_cast<int*>(lds_raw + (LDS_MAIN > LDS_EPI ? LDS_MAIN : LDS_EPI));

  // XCD-aware decode: ids round-robin over the 8 XCDs; XCD x owns the contiguous M tiles [x*m_per_xcd, ...)
  const int id = blockIdx.x, xcd = id & 7, slot = id >> 3;
  const int m_tile = xcd * m_per_xcd + slot / n_tiles, n_tile = slot % n_tiles;
  if (slot / n_tiles >= m_per_xcd || m_tile >= m_tiles) return;  // whole workgroup, before any barrier
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wave_m = wave >> 1, wave_n = wave & 1;
  const int m0 = m_tile * BM, n0 = n_tile * BN;

  if (GATHER && tid < DFINE_IGEMM_MAX_TAPS)
    tap_tab[tid] = (g.tap_dy[tid] & 0xff) | ((g.tap_dx[tid] & 0xff) << 8) | ((g.tap_w[tid] & 0xff) << 16);

  // ---- staging coordinates
  const int st_j = tid % CPR, st_r = tid / CPR;
  IgRow rowA[A_CH];
#pragma unroll
  for (int i = 0; i < A_CH; ++i) {
    const int m = m0 + st_r + i * RPP;
    rowA[i].ok = m < g.M;
    if (GATHER) {
      const int mm = rowA[i].ok ? m : 0;
      const int px = mm % g.PW, t = mm / g.PW;
      const int py = t % g.PH, b = t / g.PH;
      rowA[i].base = b * g.IH * g.IW;
      rowA[i].iy0 = py * g.isy;
      rowA[i].ix0 = px * g.isx;
    } else {
      rowA[i].base = m * g.lda;  // host guarantees M*lda < 2^31
      rowA[i].iy0 = rowA[i].ix0 = 0;
    }
  }
  int rowB[B_CH];  // element offset of weight row n (or -1)
#pragma unroll
  for (int i = 0; i < B_CH; ++i) {
    const int r = st_r + i * RPP, n = n0 + r;
    rowB[i] = (r < BN && n < g.N) ? n * g.ldb : -1;
  }
  if (GATHER) __syncthreads();  // tap table visible

  uint32_t ra[A_CH][4], rb[B_CH][4];
  auto fetch = [&](int step) {
    const int kk = step * BK + st_j * 8;
#pragma unroll
    for (int i = 0; i < A_CH; ++i) ra[i][0] = ra[i][1] = ra[i][2] = ra[i][3] = 0u;
#pragma unroll
    for (int i = 0; i < B_CH; ++i) rb[i][0] = rb[i][1] = rb[i][2] = rb[i][3] = 0u;
    if (VEC == 1) {  // any alignment: element by element (D-FINE-n's 21-channel maps, the 3-channel image)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int k = kk + e;
        if (k >= g.K) continue;
        int acol = k, bcol = k, dy = 0, dx = 0;
        if (GATHER) {
          const int tap = k / g.CI, c = k - tap * g.CI, tt = tap_tab[tap];
          dy = (int)(signed char)(tt & 0xff), dx = (int)(signed char)((tt >> 8) & 0xff);
          acol = c, bcol = ((tt >> 16) & 0xff) * g.CI + c;
        }
#pragma unroll
        for (int i = 0; i < A_CH; ++i) {
          uint32_t v = 0;
          if (GATHER) {
            const int iy = rowA[i].iy0 + dy, ix = rowA[i].ix0 + dx;
            if (rowA[i].ok && (unsigned)iy < (unsigned)g.IH && (unsigned)ix < (unsigned)g.IW)
              v = A[(size_t)(rowA[i].base + iy * g.IW + ix) * g.lda + acol];
          } else if (rowA[i].ok) {
            v = A[(size_t)rowA[i].base + acol];
          }
          ra[i][e >> 1] |= v << (16 * (e & 1));
        }
#pragma unroll
        for (int i = 0; i < B_CH; ++i)
          if (rowB[i] >= 0) rb[i][e >> 1] |= (uint32_t)Bw[(size_t)rowB[i] + bcol] << (16 * (e & 1));
      }
    } else {
#pragma unroll
      for (int h = 0; h < 8 / VEC; ++h) {  // VEC-element groups of the chunk: a group never straddles two taps
        const int k = kk + h * VEC;
        if (k >= g.K) continue;
        int acol = k, bcol = k, dy = 0, dx = 0;
        if (GATHER) {
          const int tap = k / g.CI, c = k - tap * g.CI, tt = tap_tab[tap];
          dy = (int)(signed char)(tt & 0xff), dx = (int)(signed char)((tt >> 8) & 0xff);
          acol = c, bcol = ((tt >> 16) & 0xff) * g.CI + c;
        }
#pragma unroll
        for (int i = 0; i < A_CH; ++i) {
          if (GATHER) {
            const int iy = rowA[i].iy0 + dy, ix = rowA[i].ix0 + dx;
            if (rowA[i].ok && (unsigned)iy < (unsigned)g.IH && (unsigned)ix < (unsigned)g.IW)
              ig_load<VEC>(A + (size_t)(rowA[i].base + iy * g.IW + ix) * g.lda + acol, &ra[i][h * VEC / 2]);
          } else if (rowA[i].ok) {
            ig_load<VEC>(A + (size_t)rowA[i].base + acol, &ra[i][h * VEC / 2]);
          }
        }
#pragma unroll
        for (int i = 0; i < B_CH; ++i)
          if (rowB[i] >= 0) ig_load<VEC>(Bw + (size_t)rowB[i] + bcol, &rb[i][h * VEC / 2]);
      }
    }
  };
  auto stage = [&](int buf) {
    bf16_t* at = lds + buf * STAGE;
    bf16_t* bt = at + BM * BK;
#pragma unroll
    for (int i = 0; i < A_CH; ++i) {
      const int r = st_r + i * RPP;
      *reinterpret_cast<uint4*>(at + r * BK + ((st_j ^ ig_swz<BK>(r)) << 3)) = make_uint4(ra[i][0], ra[i][1], ra[i][2], ra[i][3]);
    }
#pragma unroll
    for (int i = 0; i < B_CH; ++i) {
      const int r = st_r + i * RPP;
      if (r < BN)
        *reinterpret_cast<uint4*>(bt + r * BK + ((st_j ^ ig_swz<BK>(r)) << 3)) = make_uint4(rb[i][0], rb[i][1], rb[i][2], rb[i][3]);
    }
  };

  float4v acc[FN][FM];
#pragma unroll
  for (int i = 0; i < FN; ++i)
#pragma unroll
    for (int j = 0; j < FM; ++j) acc[i][j] = float4v{0.f, 0.f, 0.f, 0.f};

  const int n_steps = (g.K + BK - 1) / BK;
  const int f_r = lane & 15, f_c = lane >> 4;
  fetch(0);
  stage(0);
  if (n_steps > 1) fetch(1);
  __syncthreads();
  for (int s = 0; s < n_steps; ++s) {
    const bf16_t* at = lds + (s & 1) * STAGE;
    const bf16_t* bt = at + BM * BK;
#pragma unroll
    for (int sub = 0; sub < BK / 32; ++sub) {
      short8v fw[FN], fx[FM];
#pragma unroll
      for (int i = 0; i < FN; ++i) {
        const int r = wave_n * (BN / 2) + 16 * i + f_r;
        fw[i] = *reinterpret_cast<const short8v*>(bt + r * BK + (((sub * 4 + f_c) ^ ig_swz<BK>(r)) << 3));
      }
#pragma unroll
      for (int j = 0; j < FM; ++j) {
        const int r = wave_m * (BM / 2) + 16 * j + f_r;
        fx[j] = *reinterpret_cast<const short8v*>(at + r * BK + (((sub * 4 + f_c) ^ ig_swz<BK>(r)) << 3));
      }
#pragma unroll
      for (int i = 0; i < FN; ++i)
#pragma unroll
        for (int j = 0; j < FM; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[i], fx[j], acc[i][j], 0, 0, 0);
    }
    if (s + 1 < n_steps) {
      stage((s + 1) & 1);  // stage (s+1)&1 held step s-1: every wave finished reading it before the last barrier
      if (s + 2 < n_steps) fetch(s + 2);
    }
    __syncthreads();
  }

  // ---- epilogue (igemm_epilogue.h)
  ig_epilogue<FM, FN, OVEC, BNRED>(acc, lds_raw, C, bias, stats, g, n0, tid, br, [&](int ml) -> long long {
    const int m = m0 + ml;
    if (m >= g.M) return -1;
    if (GATHER) {
      const int px = m % g.PW, t = m / g.PW;
      const int py = t % g.PH, b = t / g.PH;
      return (long long)(b * g.OHf + py * g.osy + g.ooy) * g.OWf + px * g.osx + g.oox;
    }
    return m;
  });
}

static inline int ig_tile_n(int N) {
  // cost of a tile width = padded columns (wasted MFMA work) + 16 per N tile (A is re-read once per tile); ties go to
  // the wider tile
  int best = 128, best_cost = (N + 127) / 128 * 128 + 16 * ((N + 127) / 128);
  const int cands[2] = {64, 32};
  for (int c : cands) {
    const int tiles = (N + c - 1) / c, cost = tiles * c + 16 * tiles;
    if (cost < best_cost) best = c, best_cost = cost;
  }
  return best;
}

template <int FN, int BK, int VEC, int OVEC, bool GATHER, bool BNRED = false>
static int ig_launch(const void* A, const void* Bw, void* C, const float* bias, double* stats, const IGemmGeom& g,
                     hipStream_t stream, const IgBnRed& br = IgBnRed{}) {
  constexpr int BM = 128, BN = 32 * FN;
  const int m_tiles = (g.M + BM - 1) / BM, n_tiles = (g.N + BN - 1) / BN;
  const int m_per_xcd = (m_tiles + 7) / 8;
  const long long grid = 8LL * m_per_xcd * n_tiles;
  if (grid > 0x7fffffffLL) return DFINE_E_UNSUPPORTED;
  hipLaunchKernelGGL((igemm_kernel<4, FN, BK, VEC, OVEC, GATHER, BNRED>), dim3((unsigned)grid), dim3(256), 0, stream,
                     (const bf16_t*)A, (const bf16_t*)Bw, (bf16_t*)C, bias, stats, g, m_tiles, n_tiles, m_per_xcd, br);
  DFINE_LAUNCH_CHECK();
  return 0;
}

template <int BK, int VEC, int OVEC, bool GATHER>
static int ig_dispatch_n(const void* A, const void* Bw, void* C, const float* bias, double* stats, const IGemmGeom& g,
                         hipStream_t stream) {
  // a problem with fewer output tiles than CUs (the 20x20 maps: 100 row tiles) runs on part of the chip: narrower N tiles
  // multiply the workgroups at no HBM cost (the extra A reads hit L2); DFINE_IGEMM_FILL=0 switches it off
  static const bool fill = !(getenv("DFINE_IGEMM_FILL") && atoi(getenv("DFINE_IGEMM_FILL")) == 0);
  if (fill) {
    const long long m_tiles = (g.M + 127) / 128;
    int bn = ig_tile_n(g.N);
    while (bn > 32 && m_tiles * ((g.N + bn - 1) / bn) < 200) bn >>= 1;
    if (bn != ig_tile_n(g.N)) {
      if (bn == 64) return ig_launch<2, BK, VEC, OVEC, GATHER>(A, Bw, C, bias, stats, g, stream);
      return ig_launch<1, BK, VEC, OVEC, GATHER>(A, Bw, C, bias, stats, g, stream);
    }
  }
  switch (ig_tile_n(g.N)) {
    case 128: return ig_launch<4, BK, VEC, OVEC, GATHER>(A, Bw, C, bias, stats, g, stream);
    case 64: return ig_launch<2, BK, VEC, OVEC, GATHER>(A, Bw, C, bias, stats, g, stream);
    default: return ig_launch<1, BK, VEC, OVEC, GATHER>(A, Bw, C, bias, stats, g, stream);
  }
}

template <int VEC, int OVEC, bool GATHER>
static int ig_dispatch_k(const void* A, const void* Bw, void* C, const float* bias, double* stats, const IGemmGeom& g,
                         hipStream_t stream) {
  // BK = 64 halves the barriers per MFMA; BK = 32 wastes less on a K that is not a multiple of 64 and, at half the LDS and
  // 30 fewer VGPRs, lets 4 workgroups share a CU instead of 2 (DFINE_IGEMM_BK=32|64 forces one for A/B measurements)
  static const int forced = [] {
    const char* e = getenv("DFINE_IGEMM_BK");
    return e ? atoi(e) : 0;
  }();
  const int pad64 = (g.K + 63) / 64 * 64, pad32 = (g.K + 31) / 32 * 32;
  bool use64 = (pad64 == pad32 && g.K > 64) || g.K >= 1024;  // K <= 64 (the stem's 2x2 convs): one or two short steps, occupancy wins
  if (forced == 32) use64 = false;
  if (forced == 64) use64 = true;
  if (use64) return ig_dispatch_n<64, VEC, OVEC, GATHER>(A, Bw, C, bias, stats, g, stream);
  return ig_dispatch_n<32, VEC, OVEC, GATHER>(A, Bw, C, bias, stats, g, stream);
}

// the gathered input gradient with the producing norm's reductions in the epilogue: 16-byte paths only
static int ig_dispatch_bnred(const void* A, const void* Bw, void* C, const IGemmGeom& g, const IgBnRed& br, hipStream_t st) {
  const long long m_tiles = (g.M + 127) / 128;
  int bn = ig_tile_n(g.N);
  while (bn > 32 && m_tiles * ((g.N + bn - 1) / bn) < 200) bn >>= 1;
  const int pad64 = (g.K + 63) / 64 * 64, pad32 = (g.K + 31) / 32 * 32;
  const bool use64 = (pad64 == pad32 && g.K > 64) || g.K >= 1024;
#define IG_BR(FN_, BK_) return ig_launch<FN_, BK_, 8, 8, true, true>(A, Bw, C, nullptr, nullptr, g, st, br)
  if (use64) {
    if (bn == 128) IG_BR(4, 64);
    if (bn == 64) IG_BR(2, 64);
    IG_BR(1, 64);
  }
  if (bn == 128) IG_BR(4, 32);
  if (bn == 64) IG_BR(2, 32);
  IG_BR(1, 32);
#undef IG_BR
}

template <int VEC, int OVEC>
static int ig_dispatch_g(const void* A, const void* Bw, void* C, const float* bias, double* stats, const IGemmGeom& g,
                         hipStream_t stream) {
  return g.gather ? ig_dispatch_k<VEC, OVEC, true>(A, Bw, C, bias, stats, g, stream)
                  : ig_dispatch_k<VEC, OVEC, false>(A, Bw, C, bias, stats, g, stream);
}

// igemm2.hip: the LDS-DMA ring body for operands on the 16-byte path
int igemm2_run(const void* A, const void* Bw, void* C, const float* bias, double* stats, const IGemmGeom& g,
               long long a_extent, bool ovec8, hipStream_t st, const IgBnRed* br);

}  // namespace dfine

using namespace dfine;

static int igemm_entry(const void* A, const void* Bw, void* C, const float* bias, double* stats, const dfine_igemm_geom* geom,
                       const IgBnRed* br, void* stream) {
  if (!A || !Bw || !C || !geom) return DFINE_E_BADARG;
  const IGemmGeom& g = *geom;
  if (g.M <= 0 || g.N <= 0 || g.K <= 0 || g.lda <= 0 || g.ldb < 0 || g.ldc < g.N) return DFINE_E_BADARG;
  if ((g.act & 0xff) > 1) return DFINE_E_BADARG;  // bits 8, 9: diagnostics of the statistics epilogue; bit 10: C += result
  if ((g.act & 0x400) && stats) return DFINE_E_BADARG;
  long long a_extent;  // elements addressed in A (pixel offsets are 32-bit in the kernel)
  if (g.gather) {
    if (g.PH <= 0 || g.PW <= 0 || g.IH <= 0 || g.IW <= 0 || g.CI <= 0 || g.ntaps <= 0 || g.ntaps > DFINE_IGEMM_MAX_TAPS)
      return DFINE_E_BADARG;
    if (g.K != g.ntaps * g.CI || g.M % (g.PH * g.PW) != 0 || g.lda < g.CI) return DFINE_E_BADARG;
    if (g.isy <= 0 || g.isx <= 0 || g.osy <= 0 || g.osx <= 0 || g.ooy < 0 || g.oox < 0) return DFINE_E_BADARG;
    // the output pixels this launch writes must lie inside the [OHf, OWf] map
    if ((g.PH - 1) * g.osy + g.ooy >= g.OHf || (g.PW - 1) * g.osx + g.oox >= g.OWf) return DFINE_E_BADARG;
    int max_w = 0;
    for (int t = 0; t < g.ntaps; ++t) {
      if (g.tap_w[t] < 0) return DFINE_E_BADARG;
      if (g.tap_w[t] > max_w) max_w = g.tap_w[t];
    }
    if (g.ldb < (max_w + 1) * g.CI) return DFINE_E_BADARG;
    const long long images = g.M / ((long long)g.PH * g.PW);
    a_extent = images * g.IH * g.IW * g.lda;
    if (images * g.OHf * g.OWf * (long long)g.ldc > 0x7fffffffLL * 64) return DFINE_E_UNSUPPORTED;
  } else {
    if (g.lda < g.K || g.ldb < g.K) return DFINE_E_BADARG;
    a_extent = (long long)g.M * g.lda;
  }
  if (a_extent > 0x7fffffffLL || (long long)g.N * g.ldb > 0x7fffffffLL) return DFINE_E_UNSUPPORTED;
  // widest access every row start and tap offset is aligned for
  const uintptr_t pa = (uintptr_t)A, pb = (uintptr_t)Bw, pc = (uintptr_t)C;
  const int unit = g.gather ? g.CI : g.K;  // granularity at which A/B columns start
  int vec = 1;
  if (unit % 8 == 0 && g.lda % 8 == 0 && g.ldb % 8 == 0 && pa % 16 == 0 && pb % 16 == 0) vec = 8;
  else if (unit % 4 == 0 && g.lda % 4 == 0 && g.ldb % 4 == 0 && pa % 8 == 0 && pb % 8 == 0) vec = 4;
  const bool ovec8 = g.ldc % 8 == 0 && pc % 16 == 0;
  hipStream_t st = (hipStream_t)stream;
  // Which body: the LDS-DMA ring (igemm2.hip) for plain rows (1x1 convs, Linears: 5-18 % faster on this model's shapes,
  // tools/bench_igemm_cfg.py), the register-staged one below for gathers (it wins on the 40x40 / 20x20 maps and ties on
  // 80x80). DFINE_IGEMM_V=1 / 2 forces one body for every launch (A/B measurements).
  const char* fb = getenv("DFINE_IGEMM_V");  // read per call: tools/bench_igemm_cfg.py switches it at run time
  const int forced_body = fb ? atoi(fb) : 0;
  const bool use_v2 = forced_body == 2 || (forced_body != 1 && !g.gather);
  if (br != nullptr) {  // input gradient + the producing norm's backward reductions (igemm_bnred.h) / a ReLU mask (red == NULL)
    if (vec != 8 || !ovec8 || g.N % 8 != 0 || stats || bias) return DFINE_E_UNSUPPORTED;
    if (br->red == nullptr && g.gather) return DFINE_E_UNSUPPORTED;  // the mask epilogue exists in the plain-rows body only
    if (!g.gather) return igemm2_run(A, Bw, C, bias, stats, g, a_extent, ovec8, st, br);
    return ig_dispatch_bnred(A, Bw, C, g, *br, st);
  }
  if (vec == 8 && use_v2) {
    const int rc = igemm2_run(A, Bw, C, bias, stats, g, a_extent, ovec8, st, nullptr);
    if (rc != DFINE_E_UNSUPPORTED) return rc;
  }
  if (vec == 8) return ovec8 ? ig_dispatch_g<8, 8>(A, Bw, C, bias, stats, g, st) : ig_dispatch_g<8, 1>(A, Bw, C, bias, stats, g, st);
  if (vec == 4) return ovec8 ? ig_dispatch_g<4, 8>(A, Bw, C, bias, stats, g, st) : ig_dispatch_g<4, 1>(A, Bw, C, bias, stats, g, st);
  return ovec8 ? ig_dispatch_g<1, 8>(A, Bw, C, bias, stats, g, st) : ig_dispatch_g<1, 1>(A, Bw, C, bias, stats, g, st);
}

extern "C" int dfine_igemm_bf16(const void* A, const void* Bw, void* C, const float* bias, double* stats,
                                const dfine_igemm_geom* geom, void* stream) {
  return igemm_entry(A, Bw, C, bias, stats, geom, nullptr, stream);
}

extern "C" int dfine_igemm_bf16_bnred(const void* A, const void* Bw, void* C, const dfine_igemm_geom* geom,
                                      const dfine_igemm_bnred* bnred, void* stream) {
  if (!bnred || !bnred->x || !bnred->mean || !bnred->rstd || !bnred->gamma || !bnred->beta || !bnred->red) return DFINE_E_BADARG;
  if (bnred->ldx < geom->N || (bnred->dy2 && bnred->ld_dy2 < geom->N) || bnred->act < 0 || bnred->act > 2) return DFINE_E_BADARG;
  if (bnred->ldx % 8 != 0 || (bnred->dy2 && bnred->ld_dy2 % 8 != 0) || (uintptr_t)bnred->x % 16 != 0 ||
      (bnred->dy2 && (uintptr_t)bnred->dy2 % 16 != 0))
    return DFINE_E_UNSUPPORTED;
  IgBnRed br;
  br.x = (const bf16_t*)bnred->x, br.dy2 = (const bf16_t*)bnred->dy2;
  br.mean = bnred->mean, br.rstd = bnred->rstd, br.gamma = bnred->gamma, br.beta = bnred->beta, br.lab_scale = bnred->lab_scale;
  br.red = bnred->red, br.ldx = bnred->ldx, br.ld_dy2 = bnred->ld_dy2, br.act = bnred->act;
  return igemm_entry(A, Bw, C, nullptr, nullptr, geom, &br, stream);
}

extern "C" int dfine_igemm_bf16_relumask(const void* A, const void* Bw, void* C, const dfine_igemm_geom* geom, const void* mask,
                                         int ld_mask, void* stream) {
  if (!mask || !geom) return DFINE_E_BADARG;
  if (ld_mask < geom->N) return DFINE_E_BADARG;
  if (geom->gather || ld_mask % 8 != 0 || (uintptr_t)mask % 16 != 0 || (geom->act & 0x400)) return DFINE_E_UNSUPPORTED;
  IgBnRed br{};
  br.x = (const bf16_t*)mask, br.ldx = ld_mask;
  return igemm_entry(A, Bw, C, nullptr, nullptr, geom, &br, stream);
}
