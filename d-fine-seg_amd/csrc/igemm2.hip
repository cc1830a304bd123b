// Implicit-GEMM convolution / Linear on bf16 MFMA (gfx950), second body of dfine_igemm_bf16: the operand tiles go
// global -> LDS by LDS-DMA (`buffer_load_dwordx4 ... lds`), through a ring of NS stages with COUNTED s_waitcnt, so
// the loads of the next NS-1 K steps are in flight while a step's MFMAs run and never sit between the MFMAs and the
// barrier (the first body, igemm.hip, stages through registers one step ahead: its waves spend ~40 % of their cycles
// parked on s_waitcnt / the barrier, profiles/r02_wave_stalls.json).
//
//     C[m][n] = act( sum_k A(m, k) * Bw[n][k] + bias[n] )          fp32 accumulate, bf16 out
//
// Same contract as igemm.hip (conv forward / input gradient as a tap-table gather, Linear, the 2x2 transposed conv;
// reference hgnetv2.py:35-80, hybrid_encoder.py:21-45, dfine_decoder.py:316-376), for operands on the 16-byte path
// (CI / K, lda, ldb multiples of 8, 16-byte aligned bases). What differs:
//   * a wave instruction moves 64 x 16 bytes into 1 KB of CONSECUTIVE LDS (base in M0 + lane*16), so the LDS image of a
//     tile is lane-linear and the XOR swizzle that keeps the ds_read_b128 fragment reads conflict-free is applied to the
//     SOURCE address: lane (row r, slot p) fetches 16-byte chunk p ^ swz(r) of its row's K step;
//   * the operands are addressed through buffer resources: 32-bit byte offsets, and a lane whose tap falls in the
//     padding, whose row is beyond M / N or whose k is beyond K asks for an offset past the end of the resource and
//     gets ZEROS (hardware range check) -- no branch around any load, no zero-fill pass;
//   * per K step and wave: wait for all but the youngest NS-2 stages (s_waitcnt vmcnt(N), N counted), one raw
//     s_barrier (all waves' pieces of this stage have landed AND everybody is done reading the stage about to be
//     refilled), issue the loads of step s+NS-1, read fragments, MFMA.
// Workgroup = WM x WN waves, wave tile (16 FMW) x (16 FNW): 128x128/64/32 with 4 waves, 256x128 with 4 waves (wave tile
// 128x64) or 8, 256x256 with 8; K step 32 or 64. What bounds the body (measured, tools/bench_igemm_cfg.py: with the MFMAs
// switched off a launch takes 80 % of its time, with the DMAs off 50 %) is the L2 -> LDS path: ~40 GB/s per CU with ~96 KB
// in flight per CU, so the lever is bytes per FLOP -- the tile shape -- and keeping that many bytes in flight.
#include <stdlib.h>

#include "common.h"
#include "igemm_bnred.h"
#include "maskmath.h"

namespace dfine {

typedef short short8v __attribute__((ext_vector_type(8)));
typedef float float4v __attribute__((ext_vector_type(4)));
typedef dfine_igemm_geom IGemmGeom;

template <int BK>
__device__ __forceinline__ int ig2_swz(int r) {
  if (BK == 64) return (r >> 1) & 7;
  return (0x1230 >> (4 * ((r >> 2) & 3))) & 3;  // {0, 3, 2, 1}
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

#define IG2_INVALID 0xFFFFFFF0u  // beyond every resource's num_records: the load returns zeros

typedef __attribute__((address_space(3))) void* lds_ptr_t;

// The mask head's contraction as this GEMM (reference dfine_decoder.py:937-940 `einsum('bqc,bchw->bqhw')` + the loss that
// consumes it, dfine_criterion.py:272-357): M = B*HW pixels (A = the channels-last feature map [B*HW, C]), N = the matched
// mask-embedding rows of one image, K = C, with a PER-IMAGE weight matrix (Bw + image * w_stride). The logits live only in
// the accumulators:
//   EPI_MASK_STATS  the focal / dice row statistics (maskmath.h) are reduced in the epilogue and added to stats[B][N][3];
//                   nothing else is written;
//   EPI_MASK_DZ     the accumulators are turned into d(loss)/d(logit) in place (given d(loss)/d(stats)) and leave through
//                   the ordinary store path as a bf16 [B*HW, N] matrix -- the operand of the two backward GEMMs.
enum { EPI_PLAIN = 0, EPI_MASK_STATS = 1, EPI_MASK_DZ = 2, EPI_BNRED = 3, EPI_RELUMASK = 4 };  // EPI_BNRED: igemm_bnred.h
// EPI_RELUMASK: an input-gradient launch whose output is d(loss)/d(x) for x = relu(.) of the PREVIOUS Linear (the MLPs of the
// decoder, reference dfine_decoder.py:33-46, and its FFN :233-271): the epilogue zeroes the elements where x <= 0 (x = `br.x`,
// pitch `br.ldx`: the tensor this launch's layer consumed) -- the ReLU's backward without its own pass over [M, N]
struct Ig2Mask {
  const float* tgt;     // [B][G][HW] fp32 ground-truth masks at the logit resolution
  const float* alpha;   // [B][G]
  const float* gstats;  // [B][N][3]  (EPI_MASK_DZ)
  float* stats;         // [B][N][3]  (EPI_MASK_STATS), zero-initialised by the caller
  int HW, G;            // rows of A per image (0: one weight matrix for all rows); row n is ground truth n % G
  long long w_stride;   // elements between the images' weight matrices
};

// geometry of a configuration: LDS bytes (ring, or the epilogue's C tile + statistics partials if larger) and the
// waves per SIMD its workgroups can reach = the register budget the compiler is given
template <int WM, int WN, int FMW, int FNW, int BK, int NS>
struct Ig2Cfg {
  static constexpr int NW = WM * WN, NT = NW * 64, BM = WM * 16 * FMW, BN = WN * 16 * FNW;
  static constexpr int RPI = 64 / (BK / 8);  // tile rows one wave instruction (1 KB) covers
  static constexpr int A_PIECES = BM / RPI, W_PIECES = BN / RPI;
  static constexpr int A_L = A_PIECES / NW, W_L = (W_PIECES + NW - 1) / NW;
  static constexpr int STAGE_B = (BM + BN) * BK * 2, RING_B = NS * STAGE_B;
  static constexpr int MAIN_B = RING_B + ((W_PIECES % NW) ? NW * 1024 : 0);  // + a scratch KB per wave for surplus DMA slots
  static constexpr int EH = BM > 128 ? BM / 128 : 1;  // the epilogue handles the tile in EH row blocks of BMH rows
  static constexpr int BMH = BM / EH;
  static constexpr int CPITCH = BN + 8, OCPR = BN / 8, ORPP = NT / OCPR;
  static constexpr int EPI_B = BMH * CPITCH * 2 + 2 * ORPP * BN * 4;
  static constexpr int LDS_B = MAIN_B > EPI_B ? MAIN_B : EPI_B;
  static constexpr int WGS = (160 * 1024) / (LDS_B + 64) < 1 ? 1 : (160 * 1024) / (LDS_B + 64);
  static constexpr int WPS_ = WGS * NW / 4;
  static constexpr int WAVES_PER_SIMD = WPS_ > 4 ? 4 : (WPS_ < 1 ? 1 : WPS_);
  static_assert(A_PIECES % NW == 0, "A pieces must divide over the waves");
  static_assert(BMH % ORPP == 0 && 16 * FMW <= BMH, "epilogue row blocks");
  static_assert(NT >= 2 * BN, "statistics reduction: one thread per (quantity, column)");
};

template <int WM, int WN, int FMW, int FNW, int BK, int NS, bool PIPE, bool GATHER, int EPI>
__global__ __launch_bounds__(WM * WN * 64, ((EPI == 3 && Ig2Cfg<WM, WN, FMW, FNW, BK, NS>::WAVES_PER_SIMD > 3)
                                                 ? 3 : Ig2Cfg<WM, WN, FMW, FNW, BK, NS>::WAVES_PER_SIMD)) void igemm2_kernel(
    const bf16_t* __restrict__ A, const bf16_t* __restrict__ Bw, bf16_t* __restrict__ C, const float* __restrict__ bias,
    double* __restrict__ stats, const IGemmGeom g, int m_tiles, int n_tiles, int m_per_xcd, unsigned a_bytes, unsigned b_bytes,
    int ovec8, const Ig2Mask mk, const IgBnRed br) {
  typedef Ig2Cfg<WM, WN, FMW, FNW, BK, NS> Cfg;
  constexpr int NT = Cfg::NT, NW = Cfg::NW, BM = Cfg::BM, BN = Cfg::BN;
  constexpr int CPR = BK / 8, RPI = Cfg::RPI;
  constexpr int W_PIECES = Cfg::W_PIECES, A_L = Cfg::A_L, W_L = Cfg::W_L;
  constexpr int L = A_L + W_L;  // DMA instructions per wave and stage
  constexpr int STAGE_B = Cfg::STAGE_B, RING_B = Cfg::RING_B, LDS_B = Cfg::LDS_B;
  constexpr int CPITCH = Cfg::CPITCH, OCPR = Cfg::OCPR, ORPP = Cfg::ORPP, EH = Cfg::EH, BMH = Cfg::BMH;
  constexpr int RM = 16 * FMW, RN = 16 * FNW;  // wave tile
  __shared__ __attribute__((aligned(1024))) unsigned char lds_raw[LDS_B + 64];
  int* const tap_tab = reinterpret_cast<int*>(lds_raw + LDS_B);

  const int id = blockIdx.x, xcd = id & 7, slot = id >> 3;
  const int m_tile = xcd * m_per_xcd + slot / n_tiles, n_tile = slot % n_tiles;
  if (slot / n_tiles >= m_per_xcd || m_tile >= m_tiles) return;  // whole workgroup, before any barrier
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wave_m = wave / WN, wave_n = wave % WN;
  const int m0 = m_tile * BM, n0 = n_tile * BN;

  if (GATHER && tid < DFINE_IGEMM_MAX_TAPS)
    tap_tab[tid] = tid < g.ntaps ? ((g.tap_dy[tid] & 0xff) | ((g.tap_dx[tid] & 0xff) << 8) | ((g.tap_w[tid] & 0xff) << 16)) : 0;

  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(A), 0, (int)a_bytes, 0x00020000);
  const int img = mk.HW > 0 ? m0 / mk.HW : 0;  // per-image weights: a tile lies inside one image (host checks HW % BM == 0)
  const __amdgpu_buffer_rsrc_t rsB =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(Bw) + (size_t)img * mk.w_stride, 0, (int)b_bytes, 0x00020000);

  // ---- loader coordinates (constant over the K loop)
  const int l_r = lane / CPR, l_p = lane % CPR;
  // A pieces: gather -> (pixel index of the image's first pixel, iy0, ix0) ; plain -> byte offset of the row
  unsigned a_base[A_L];
  int a_iy0[A_L], a_ix0[A_L];
  int a_tap[A_L], a_cc[A_L];  // gather: current tap and channel inside it; plain: a_cc = column (elements)
  bool a_ok[A_L];
#pragma unroll
  for (int i = 0; i < A_L; ++i) {
    const int r = (wave + NW * i) * RPI + l_r;
    const int m = m0 + r;
    a_ok[i] = m < g.M;
    const int kcol = (l_p ^ ig2_swz<BK>(r)) * 8;
    if (GATHER) {
      const int mm = a_ok[i] ? m : 0;
      const int px = mm % g.PW, t = mm / g.PW;
      const int py = t % g.PH, b = t / g.PH;
      a_base[i] = (unsigned)(b * g.IH * g.IW);
      a_iy0[i] = py * g.isy;
      a_ix0[i] = px * g.isx;
      a_tap[i] = kcol / g.CI;
      a_cc[i] = kcol - a_tap[i] * g.CI;
    } else {
      a_base[i] = (unsigned)m * (unsigned)g.lda * 2u;  // host guarantees the byte extent < 2^32
      a_iy0[i] = a_ix0[i] = 0;
      a_tap[i] = 0;
      a_cc[i] = kcol;
    }
  }
  unsigned w_base[W_L];
  int w_tap[W_L], w_cc[W_L];
  bool w_ok[W_L], w_real[W_L];
#pragma unroll
  for (int i = 0; i < W_L; ++i) {
    const int q = wave + NW * i;
    w_real[i] = q < W_PIECES;  // wave-uniform
    const int r = q * RPI + l_r;
    const int n = n0 + r;
    w_ok[i] = w_real[i] && n < g.N;
    w_base[i] = (unsigned)n * (unsigned)g.ldb * 2u;
    const int kcol = (l_p ^ ig2_swz<BK>(r)) * 8;
    if (GATHER) {
      w_tap[i] = kcol / g.CI;
      w_cc[i] = kcol - w_tap[i] * g.CI;
    } else {
      w_tap[i] = 0;
      w_cc[i] = kcol;
    }
  }
  if (GATHER) __syncthreads();  // tap table visible

  auto issue = [&](int buf) {  // the loads of the NEXT not yet issued K step into ring stage `buf`; advances the k state
    unsigned char* const stage = lds_raw + buf * STAGE_B;
#pragma unroll
    for (int i = 0; i < A_L; ++i) {
      unsigned voff = IG2_INVALID;
      if (GATHER) {
        if (a_ok[i] && a_tap[i] < g.ntaps) {
          const int tt = tap_tab[a_tap[i]];
          const int iy = a_iy0[i] + (int)(signed char)(tt & 0xff), ix = a_ix0[i] + (int)(signed char)((tt >> 8) & 0xff);
          if ((unsigned)iy < (unsigned)g.IH && (unsigned)ix < (unsigned)g.IW)
            voff = ((a_base[i] + (unsigned)(iy * g.IW + ix)) * (unsigned)g.lda + (unsigned)a_cc[i]) * 2u;
        }
        a_cc[i] += BK;
        while (a_cc[i] >= g.CI) a_cc[i] -= g.CI, ++a_tap[i];
      } else {
        if (a_ok[i] && a_cc[i] < g.K) voff = a_base[i] + (unsigned)a_cc[i] * 2u;
        a_cc[i] += BK;
      }
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr_t)(stage + (wave + NW * i) * 1024), 16, voff, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < W_L; ++i) {
      unsigned voff = IG2_INVALID;
      if (GATHER) {
        if (w_ok[i] && w_tap[i] < g.ntaps)
          voff = w_base[i] + (unsigned)(((tap_tab[w_tap[i]] >> 16) & 0xff) * g.CI + w_cc[i]) * 2u;
        w_cc[i] += BK;
        while (w_cc[i] >= g.CI) w_cc[i] -= g.CI, ++w_tap[i];
      } else {
        if (w_ok[i] && w_cc[i] < g.K) voff = w_base[i] + (unsigned)w_cc[i] * 2u;
        w_cc[i] += BK;
      }
      unsigned char* dst = w_real[i] ? stage + BM * BK * 2 + (wave + NW * i) * 1024 : lds_raw + RING_B + wave * 1024;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_ptr_t)dst, 16, voff, 0, 0, 0);
    }
  };

  float4v acc[FNW][FMW];
#pragma unroll
  for (int i = 0; i < FNW; ++i)
#pragma unroll
    for (int j = 0; j < FMW; ++j) acc[i][j] = float4v{0.f, 0.f, 0.f, 0.f};

  const int n_steps = (g.K + BK - 1) / BK;
  const int f_r = lane & 15, f_c = lane >> 4;
  const bool dbg_no_mma = (g.act & 0x800) != 0, dbg_no_dma = (g.act & 0x1000) != 0;  // diagnostics (tools/bench_igemm_cfg.py)
  auto read_frags = [&](short8v (&fw)[FNW], short8v (&fx)[FMW], int buf_, int sub) {
    const bf16_t* at = reinterpret_cast<const bf16_t*>(lds_raw + buf_ * STAGE_B);
    const bf16_t* bt = at + BM * BK;
#pragma unroll
    for (int i = 0; i < FNW; ++i) {
      const int r = wave_n * RN + 16 * i + f_r;
      fw[i] = *reinterpret_cast<const short8v*>(bt + r * BK + (((sub * 4 + f_c) ^ ig2_swz<BK>(r)) << 3));
    }
#pragma unroll
    for (int j = 0; j < FMW; ++j) {
      const int r = wave_m * RM + 16 * j + f_r;
      fx[j] = *reinterpret_cast<const short8v*>(at + r * BK + (((sub * 4 + f_c) ^ ig2_swz<BK>(r)) << 3));
    }
  };
  auto mma = [&](const short8v (&fw)[FNW], const short8v (&fx)[FMW]) {
    if (dbg_no_mma) return;
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < FNW; ++i)
#pragma unroll
      for (int j = 0; j < FMW; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[i], fx[j], acc[i][j], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
  };
  auto wait_behind = [&](int b) {  // all of this wave's DMAs done except those of the youngest b stages
    if (NS >= 6 && b >= 4) wait_vmcnt<(NS >= 6 ? 4 : 0) * L>();
    else if (NS >= 5 && b >= 3) wait_vmcnt<(NS >= 5 ? 3 : 0) * L>();
    else if (NS >= 4 && b >= 2) wait_vmcnt<(NS >= 4 ? 2 : 0) * L>();
    else if (NS >= 3 && b >= 1) wait_vmcnt<(NS >= 3 ? 1 : 0) * L>();
    else wait_vmcnt<0>();
  };
#pragma unroll
  for (int s = 0; s < NS - 1; ++s)
    if (s < n_steps && !dbg_no_dma) issue(s);
  int buf = 0, fill = NS - 1;  // ring stage read by this step / refilled by this step
  if (!PIPE) {
    for (int s = 0; s < n_steps; ++s) {
      wait_behind(min(NS - 2, n_steps - 1 - s));  // stages in flight BEHIND step s's own
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this wave's fragment reads of the stage about to be refilled
      __builtin_amdgcn_s_barrier();
      if (s + NS - 1 < n_steps && !dbg_no_dma) issue(fill);
#pragma unroll
      for (int sub = 0; sub < BK / 32; ++sub) {
        short8v fw[FNW], fx[FMW];
        read_frags(fw, fx, buf, sub);
        mma(fw, fx);
      }
      buf = buf + 1 == NS ? 0 : buf + 1;
      fill = fill + 1 == NS ? 0 : fill + 1;
    }
  } else {
    // software pipeline over 32-wide sub-steps with two fragment register sets: the ds_reads of sub-step u+1 are issued
    // BEFORE the MFMAs of sub-step u, so the LDS latency runs under the matrix pipe. A sub-step that opens a new stage
    // first waits for that stage's DMAs (counted), passes the step's one barrier and refills the stage everybody left a
    // step ago. At the barrier of step s the stages issued are 0 .. s+NS-2, so min(n_steps-2-s, NS-3) may stay in flight.
    short8v fwA[FNW], fxA[FMW], fwB[FNW], fxB[FMW];
    auto open_next = [&](int s) {  // called in step s with s + 1 < n_steps: make stage s+1 readable, refill stage s-1's buffer
      wait_behind(min(NS - 3, n_steps - 2 - s));
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (s + NS - 1 < n_steps && !dbg_no_dma) issue(fill);
      buf = buf + 1 == NS ? 0 : buf + 1;
      fill = fill + 1 == NS ? 0 : fill + 1;
    };
    wait_behind(min(NS - 2, n_steps - 1));
    __builtin_amdgcn_s_barrier();
    read_frags(fwA, fxA, 0, 0);
    if (BK == 64) {
      for (int s = 0; s < n_steps; ++s) {
        read_frags(fwB, fxB, buf, 1);
        mma(fwA, fxA);
        if (s + 1 < n_steps) {
          open_next(s);
          read_frags(fwA, fxA, buf, 0);
        }
        mma(fwB, fxB);
      }
    } else {
      for (int s = 0; s < n_steps; s += 2) {
        if (s + 1 < n_steps) {
          open_next(s);
          read_frags(fwB, fxB, buf, 0);
        }
        mma(fwA, fxA);
        if (s + 1 >= n_steps) break;
        if (s + 2 < n_steps) {
          open_next(s + 1);
          read_frags(fwA, fxA, buf, 0);
        }
        mma(fwB, fxB);
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // (n_steps == 0 never happens: K >= 1)
  __syncthreads();  // every wave is done with the ring: the epilogue reuses it

  if (EPI == EPI_MASK_STATS || EPI == EPI_MASK_DZ) {
    // the mask logits of this tile are the accumulators: accumulator quad (i, j) = rows n_base .. n_base+3 of pixel px
    const int px0 = m0 - img * mk.HW + wave_m * RM + f_r;
    float sf[FNW][4], si[FNW][4], sp[FNW][4];
#pragma unroll
    for (int i = 0; i < FNW; ++i) {
      const int nb = n0 + wave_n * RN + 16 * i + 4 * f_c;
      const float* tp[4];
      float al[4], g0[4], g1[4], g2[4];
      bool okn[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        okn[e] = nb + e < g.N;
        const int n = okn[e] ? nb + e : 0, gt = n % mk.G;
        tp[e] = mk.tgt + ((size_t)img * mk.G + gt) * mk.HW;
        al[e] = mk.alpha[img * mk.G + gt];
        sf[i][e] = si[i][e] = sp[i][e] = 0.f;
        if (EPI == EPI_MASK_DZ) {
          const float* gs = mk.gstats + ((size_t)img * g.N + n) * 3;
          g0[e] = gs[0], g1[e] = gs[1], g2[e] = gs[2];
        }
      }
#pragma unroll
      for (int j = 0; j < FMW; ++j) {
        const int px = px0 + 16 * j;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float z = bf16_to_f32(f32_to_bf16(acc[i][j][e]));  // the logit as the reference's bf16 einsum hands it to the loss
          const float t = tp[e][px];
          if (EPI == EPI_MASK_STATS) {
            float f, p;
            mask_terms(z, t, al[e], f, p);
            sf[i][e] += f, si[i][e] += p * t, sp[i][e] += p;
          } else {
            acc[i][j][e] = okn[e] ? mask_dz(z, t, al[e], g0[e], g1[e], g2[e]) : 0.f;
          }
        }
      }
      if (EPI == EPI_MASK_STATS) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float a = sf[i][e], b = si[i][e], c = sp[i][e];
#pragma unroll
          for (int sh = 1; sh < 16; sh <<= 1) {  // the 16 lanes of one f_c hold the 16 pixels of a fragment column
            a += __shfl_xor(a, sh, 64), b += __shfl_xor(b, sh, 64), c += __shfl_xor(c, sh, 64);
          }
          if (f_r == 0 && okn[e]) {
            float* dst = mk.stats + ((size_t)img * g.N + nb + e) * 3;
            atomicAdd(dst, a), atomicAdd(dst + 1, b), atomicAdd(dst + 2, c);
          }
        }
      }
    }
    if (EPI == EPI_MASK_STATS) return;  // no barrier follows for anybody
  }
  // ---- epilogue, in EH row blocks of BMH rows: accumulators (+bias, act) -> bf16 tile in LDS -> coalesced rows
  // (+ per-channel statistics). D layout of mfma_f32_16x16x32 with the weight tile as operand A: row (= n) =
  // (lane>>4)*4 + reg, col (= m) = lane&15
  bf16_t* ct = reinterpret_cast<bf16_t*>(lds_raw);
  const int o_j = tid % OCPR, o_r = tid / OCPR;
  const int ncol = n0 + o_j * 8;
  const int nvalid = min(max(g.N - ncol, 0), 8);
  const bool accumulate = (g.act & 0x400) != 0;
  BnRedLane brl;
  if (EPI == EPI_BNRED) brl.init(br, ncol, nvalid == 8);
#pragma unroll
  for (int h = 0; h < EH; ++h) {
    if (h > 0) __syncthreads();  // the previous block's tile and partials have been read
    if ((wave_m * RM) / BMH == h) {
      const int mrow0 = wave_m * RM - h * BMH;
#pragma unroll
      for (int i = 0; i < FNW; ++i) {
        const int nl = wave_n * RN + 16 * i + 4 * f_c;
        float bv[4] = {0.f, 0.f, 0.f, 0.f};
        if (bias != nullptr) {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (n0 + nl + e < g.N) bv[e] = bias[n0 + nl + e];
        }
#pragma unroll
        for (int j = 0; j < FMW; ++j) {
          const int ml = mrow0 + 16 * j + f_r;
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
    }
    __syncthreads();
    float s1[8], s2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) s1[e] = s2[e] = 0.f;
#pragma unroll
    for (int i = 0; i < BMH / ORPP; ++i) {
      const int ml = o_r + i * ORPP, m = m0 + h * BMH + ml;
      if (m >= g.M || nvalid == 0) continue;
      uint4 v = *reinterpret_cast<const uint4*>(ct + ml * CPITCH + o_j * 8);
      size_t orow;
      if (GATHER) {
        const int px = m % g.PW, t = m / g.PW;
        const int py = t % g.PH, b = t / g.PH;
        orow = ((size_t)(b * g.OHf + py * g.osy + g.ooy) * g.OWf + px * g.osx + g.oox) * g.ldc;
      } else {
        orow = (size_t)m * g.ldc;
      }
      if (EPI == EPI_RELUMASK && nvalid == 8) {  // the consumed activation's ReLU backward: keep where x > 0
        const uint4 xm = *reinterpret_cast<const uint4*>(br.x + (orow / g.ldc) * (size_t)br.ldx + ncol);
        auto keep = [](uint32_t val, uint32_t x) -> uint32_t {
          const uint32_t lo = ((x & 0x8000u) || !(x & 0x7fffu)) ? 0u : (val & 0xffffu);
          const uint32_t hi = ((x & 0x80000000u) || !(x & 0x7fff0000u)) ? 0u : (val & 0xffff0000u);
          return lo | hi;
        };
        v.x = keep(v.x, xm.x), v.y = keep(v.y, xm.y), v.z = keep(v.z, xm.z), v.w = keep(v.w, xm.w);
      }
      bf16_t* dst = C + orow + ncol;
      if (accumulate) {  // C += result: the second of two input gradients of one tensor
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
        if (ovec8 && nvalid == 8) {
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
          if (EPI == EPI_BNRED) brl.row(br, r, orow / g.ldc, ncol);  // the SUM is the finished gradient
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e)
            if (e < nvalid)
              dst[e] = f32_to_bf16(__uint_as_float((w[e >> 1] >> (16 * (e & 1))) << 16) + bf16_to_f32(dst[e]));
        }
      } else if (ovec8 && nvalid == 8) {
        *reinterpret_cast<uint4*>(dst) = v;
        if (EPI == EPI_BNRED) {
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
      // column sums over the block's rows (see igemm.hip): partials parked behind the C tile, summed by column
      float* part = reinterpret_cast<float*>(lds_raw + BMH * CPITCH * 2);  // [2][ORPP][BN]
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
        if (n0 + c < g.N && (g.act & 0x100) == 0)
          atomicAdd(stats + ((size_t)(blockIdx.x & 7) * 2 + q) * g.N + n0 + c, (double)a);
      }
    }
    if (EPI == EPI_BNRED) {
      float* part = reinterpret_cast<float*>(lds_raw + BMH * CPITCH * 2);  // [2][ORPP][BN]
      bnred_columns<BN, ORPP, NT>(part, brl.a[0], brl.a[1], o_r, o_j, tid, br.red, 0, g.N, n0, (g.act & 0x2000) != 0);
      if (br.lab_scale != nullptr) bnred_columns<BN, ORPP, NT>(part, brl.a[2], brl.a[3], o_r, o_j, tid, br.red, 2, g.N, n0, (g.act & 0x2000) != 0);
#pragma unroll
      for (int e = 0; e < 8; ++e) brl.a[0][e] = brl.a[1][e] = brl.a[2][e] = brl.a[3][e] = 0.f;
    }
  }
}

template <int WM, int WN, int FMW, int FNW, int BK, int NS, bool PIPE, bool GATHER, int EPI = EPI_PLAIN>
static int ig2_launch(const void* A, const void* Bw, void* C, const float* bias, double* stats, const IGemmGeom& g,
                      unsigned a_bytes, unsigned b_bytes, bool ovec8, hipStream_t stream, const Ig2Mask mk = Ig2Mask{},
                      const IgBnRed br = IgBnRed{}) {
  typedef Ig2Cfg<WM, WN, FMW, FNW, BK, NS> Cfg;
  const int m_tiles = (g.M + Cfg::BM - 1) / Cfg::BM, n_tiles = (g.N + Cfg::BN - 1) / Cfg::BN;
  const int m_per_xcd = (m_tiles + 7) / 8;
  const long long grid = 8LL * m_per_xcd * n_tiles;
  if (grid > 0x7fffffffLL) return DFINE_E_UNSUPPORTED;
  hipLaunchKernelGGL((igemm2_kernel<WM, WN, FMW, FNW, BK, NS, PIPE, GATHER, EPI>), dim3((unsigned)grid), dim3(Cfg::NT), 0, stream,
                     (const bf16_t*)A, (const bf16_t*)Bw, (bf16_t*)C, bias, stats, g, m_tiles, n_tiles, m_per_xcd, a_bytes, b_bytes,
                     (int)ovec8, mk, br);
  DFINE_LAUNCH_CHECK();
  return 0;
}

static int env_int(const char* name, int dflt) {
  const char* e = getenv(name);
  return e ? atoi(e) : dflt;
}

template <bool GATHER>
static int ig2_dispatch(const void* A, const void* Bw, void* C, const float* bias, double* stats, const IGemmGeom& g,
                        unsigned a_bytes, unsigned b_bytes, bool ovec8, hipStream_t st, const IgBnRed* br) {
  // DFINE_IGEMM2_CFG=<wm><wn><fmw><fnw><bk/32><ns><pipe> (e.g. 2244130) forces one configuration for A/B measurements
  const int forced = env_int("DFINE_IGEMM2_CFG", 0);  // read per call: tools/bench_igemm_cfg.py switches it at run time
  // default: 128-row tiles; N tile as in igemm.hip (padded columns vs A re-reads), halved while the grid cannot fill the chip
  int bn = 128;
  {
    int best_cost = (g.N + 127) / 128 * 128 + 16 * ((g.N + 127) / 128);
    const int cands[2] = {64, 32};
    for (int c : cands) {
      const int tiles = (g.N + c - 1) / c, cost = tiles * c + 16 * tiles;
      if (cost < best_cost) bn = c, best_cost = cost;
    }
  }
  const long long m_tiles = (g.M + 127) / 128;
  while (bn > 32 && m_tiles * ((g.N + bn - 1) / bn) < 200) bn >>= 1;
  int cfg = 2240130 + (bn / 32) * 1000;  // 128 x bn, 4 waves, K step 32, 3 stages
  if (forced) cfg = forced;
  if (br != nullptr && br->red == nullptr) {  // ReLU-mask epilogue (EPI_RELUMASK): br carries the mask tensor only
    if (GATHER || !ovec8 || g.N % 8 != 0 || stats != nullptr || (g.act & 0x400)) return DFINE_E_UNSUPPORTED;
    switch (bn) {
      case 128: return ig2_launch<2, 2, 4, 4, 32, 3, false, false, EPI_RELUMASK>(A, Bw, C, bias, stats, g, a_bytes, b_bytes, ovec8, st, Ig2Mask{}, *br);
      case 64: return ig2_launch<2, 2, 4, 2, 32, 3, false, false, EPI_RELUMASK>(A, Bw, C, bias, stats, g, a_bytes, b_bytes, ovec8, st, Ig2Mask{}, *br);
      default: return ig2_launch<2, 2, 4, 1, 32, 3, false, false, EPI_RELUMASK>(A, Bw, C, bias, stats, g, a_bytes, b_bytes, ovec8, st, Ig2Mask{}, *br);
    }
  }
  if (br != nullptr) {  // the producing norm's backward reductions in the epilogue (igemm_bnred.h): default tiles only
    if (GATHER || !ovec8 || g.N % 8 != 0 || stats != nullptr) return DFINE_E_UNSUPPORTED;
    switch (bn) {
      case 128: return ig2_launch<2, 2, 4, 4, 32, 3, false, false, EPI_BNRED>(A, Bw, C, bias, stats, g, a_bytes, b_bytes, ovec8, st, Ig2Mask{}, *br);
      case 64: return ig2_launch<2, 2, 4, 2, 32, 3, false, false, EPI_BNRED>(A, Bw, C, bias, stats, g, a_bytes, b_bytes, ovec8, st, Ig2Mask{}, *br);
      default: return ig2_launch<2, 2, 4, 1, 32, 3, false, false, EPI_BNRED>(A, Bw, C, bias, stats, g, a_bytes, b_bytes, ovec8, st, Ig2Mask{}, *br);
    }
  }
#define IG2_CASE(WM_, WN_, FMW_, FNW_, BK_, NS_, P_)                                                               \
  if (cfg == WM_ * 1000000 + WN_ * 100000 + FMW_ * 10000 + FNW_ * 1000 + (BK_ / 32) * 100 + NS_ * 10 + P_) \
    return ig2_launch<WM_, WN_, FMW_, FNW_, BK_, NS_, P_ != 0, GATHER>(A, Bw, C, bias, stats, g, a_bytes, b_bytes, ovec8, st);
  IG2_CASE(2, 2, 4, 4, 32, 3, 0)  // 128 x 128, 3 workgroups per CU
  IG2_CASE(2, 2, 4, 2, 32, 3, 0)  // 128 x 64
  IG2_CASE(2, 2, 4, 1, 32, 3, 0)  // 128 x 32
  IG2_CASE(2, 2, 4, 4, 32, 3, 1)
  IG2_CASE(2, 2, 8, 4, 32, 3, 0)  // 256 x 128, 4 waves of 128 x 64, 2 workgroups per CU
  IG2_CASE(2, 2, 8, 4, 32, 3, 1)
  IG2_CASE(2, 2, 8, 4, 32, 4, 1)
  IG2_CASE(2, 2, 8, 4, 64, 3, 1)
  IG2_CASE(4, 2, 4, 4, 32, 5, 0)  // 256 x 128, 8 waves of 64 x 64
  IG2_CASE(4, 2, 4, 4, 32, 5, 1)
  IG2_CASE(2, 4, 8, 4, 32, 4, 0)  // 256 x 256, 8 waves of 128 x 64
  IG2_CASE(2, 4, 8, 4, 32, 4, 1)
  IG2_CASE(2, 2, 8, 2, 32, 3, 0)  // 256 x 64
  IG2_CASE(2, 2, 8, 2, 32, 4, 1)
#undef IG2_CASE
  return DFINE_E_UNSUPPORTED;
}

// called by dfine_igemm_bf16 (igemm.hip) after its argument checks, for operands on the 16-byte path
int igemm2_run(const void* A, const void* Bw, void* C, const float* bias, double* stats, const IGemmGeom& g,
               long long a_extent, bool ovec8, hipStream_t st, const IgBnRed* br) {
  const long long a_bytes = a_extent * 2, b_bytes = (long long)g.N * g.ldb * 2;
  if (a_bytes > 0xFFFFFF00LL || b_bytes > 0xFFFFFF00LL) return DFINE_E_UNSUPPORTED;  // 32-bit buffer offsets
  return g.gather ? ig2_dispatch<true>(A, Bw, C, bias, stats, g, (unsigned)a_bytes, (unsigned)b_bytes, ovec8, st, br)
                  : ig2_dispatch<false>(A, Bw, C, bias, stats, g, (unsigned)a_bytes, (unsigned)b_bytes, ovec8, st, br);
}


// the three launches of the mask head's contraction (see Ig2Mask); 128 x 128 tiles, plain rows
template <int EPI>
static int mask_rows_launch(const void* feat, const void* rows, void* out, int ldo, const Ig2Mask& mk, int B, int N, int K,
                            long long rows_ld, hipStream_t st) {
  if (B <= 0 || mk.HW <= 0 || N <= 0 || K <= 0 || mk.HW % 128 != 0 || K % 8 != 0 || rows_ld % 8 != 0) return DFINE_E_UNSUPPORTED;
  const long long M = (long long)B * mk.HW;
  if (M * K * 2 > 0xFFFFFF00LL || (long long)N * rows_ld * 2 > 0xFFFFFF00LL || M > 0x7fffffffLL) return DFINE_E_UNSUPPORTED;
  if (((uintptr_t)feat | (uintptr_t)rows) % 16 != 0) return DFINE_E_UNSUPPORTED;
  IGemmGeom g = {};
  g.M = (int)M, g.N = N, g.K = K, g.lda = K, g.ldb = (int)rows_ld, g.ldc = ldo > 0 ? ldo : N;
  const bool ovec8 = out != nullptr && g.ldc % 8 == 0 && (uintptr_t)out % 16 == 0;
  return ig2_launch<2, 2, 4, 4, 32, 3, false, false, EPI>(feat, rows, out, nullptr, nullptr, g, (unsigned)(M * K * 2),
                                                           (unsigned)((long long)N * rows_ld * 2), ovec8, st, mk);
}

}  // namespace dfine

using namespace dfine;

extern "C" int dfine_mask_rows_stats(const void* feat, const void* rows, const float* tgt, const float* alpha, float* stats,
                                     int B, int HW, int N, int G, int C, void* stream) {
  if (!feat || !rows || !tgt || !alpha || !stats || G <= 0) return DFINE_E_BADARG;
  Ig2Mask mk = {tgt, alpha, nullptr, stats, HW, G, (long long)N * C};
  return mask_rows_launch<EPI_MASK_STATS>(feat, rows, nullptr, 0, mk, B, N, C, C, (hipStream_t)stream);
}

extern "C" int dfine_mask_rows_dz(const void* feat, const void* rows, const float* tgt, const float* alpha, const float* g_stats,
                                  void* dz, int ldz, int B, int HW, int N, int G, int C, void* stream) {
  if (!feat || !rows || !tgt || !alpha || !g_stats || !dz || G <= 0 || ldz < N) return DFINE_E_BADARG;
  Ig2Mask mk = {tgt, alpha, g_stats, nullptr, HW, G, (long long)N * C};
  return mask_rows_launch<EPI_MASK_DZ>(feat, rows, dz, ldz, mk, B, N, C, C, (hipStream_t)stream);
}

extern "C" int dfine_mask_rows_dfeat(const void* dz, int ldz, const void* rows_t, int ld_rt, void* d_feat, int B, int HW, int Npad,
                                     int C, void* stream) {
  // d_feat[b] [HW, C] = dz[b] [HW, Npad] @ rows[b] [Npad, C]: this GEMM's weight operand is rows[b]^T [C][ld_rt]
  if (!dz || !rows_t || !d_feat || ldz < Npad || ld_rt < Npad) return DFINE_E_BADARG;
  if (ldz % 8 != 0) return DFINE_E_UNSUPPORTED;
  Ig2Mask mk = {nullptr, nullptr, nullptr, nullptr, HW, 1, (long long)C * ld_rt};
  // A = dz with row pitch ldz: the generic launcher takes lda = K, so build the geometry here
  if (B <= 0 || HW <= 0 || HW % 128 != 0 || Npad % 8 != 0 || ld_rt % 8 != 0) return DFINE_E_UNSUPPORTED;
  const long long M = (long long)B * HW;
  if (M * ldz * 2 > 0xFFFFFF00LL || M > 0x7fffffffLL || (((uintptr_t)dz | (uintptr_t)rows_t | (uintptr_t)d_feat) % 16) != 0) return DFINE_E_UNSUPPORTED;
  IGemmGeom g = {};
  g.M = (int)M, g.N = C, g.K = Npad, g.lda = ldz, g.ldb = ld_rt, g.ldc = C;
  return ig2_launch<2, 2, 4, 4, 32, 3, false, false, EPI_PLAIN>(dz, rows_t, d_feat, nullptr, nullptr, g, (unsigned)(M * ldz * 2),
                                                                 (unsigned)((long long)C * ld_rt * 2), C % 8 == 0, (hipStream_t)stream, mk);
}
