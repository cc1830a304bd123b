// Shared device helpers for libdfine_hip.so (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "dfine_hip.h"

#define DFINE_LAUNCH_CHECK()                       \
  do {                                             \
    hipError_t e__ = hipGetLastError();            \
    if (e__ != hipSuccess) return (int)e__;        \
  } while (0)

namespace dfine {

typedef uint16_t bf16_t;  // raw bf16 bits

__device__ __forceinline__ float bf16_to_f32(bf16_t v) {
  return __uint_as_float(((uint32_t)v) << 16);
}

// Round-to-nearest-even f32 -> bf16. A plain cast lowers to v_cvt_pk_bf16_f32 on gfx950,
// which keeps NaN a NaN (the integer-add trick does not).
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
  __bf16 b = (__bf16)f;
  return *reinterpret_cast<bf16_t*>(&b);
}

// 16-byte vector of T, as 4 dwords, for coalesced 16 B/lane global accesses.
struct alignas(16) Vec16 {
  uint32_t w[4];
};

template <typename T>
struct VecTraits;

template <>
struct VecTraits<float> {
  static constexpr int kElems = 4;
  __device__ static __forceinline__ void unpack(const Vec16& v, float* f) {
#pragma unroll
    for (int i = 0; i < 4; ++i) f[i] = __uint_as_float(v.w[i]);
  }
  __device__ static __forceinline__ Vec16 pack(const float* f) {
    Vec16 v;
#pragma unroll
    for (int i = 0; i < 4; ++i) v.w[i] = __float_as_uint(f[i]);
    return v;
  }
  __device__ static __forceinline__ float load1(const float* p) { return *p; }
};

template <>
struct VecTraits<bf16_t> {
  static constexpr int kElems = 8;
  __device__ static __forceinline__ void unpack(const Vec16& v, float* f) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      f[2 * i] = __uint_as_float(v.w[i] << 16);
      f[2 * i + 1] = __uint_as_float(v.w[i] & 0xffff0000u);
    }
  }
  __device__ static __forceinline__ Vec16 pack(const float* f) {
    Vec16 v;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      v.w[i] = (uint32_t)f32_to_bf16(f[2 * i]) | ((uint32_t)f32_to_bf16(f[2 * i + 1]) << 16);
    return v;
  }
  __device__ static __forceinline__ float load1(const bf16_t* p) { return bf16_to_f32(*p); }
};

// VEC consecutive elements of T <-> fp32, as ONE naturally aligned access of VEC*sizeof(T) bytes
// (16, 8, 4 or 2 bytes), for maps whose channel count is not a multiple of the 16-byte vector.
template <int BYTES>
struct RawBytes;
template <> struct RawBytes<16> { typedef uint4 type; };
template <> struct RawBytes<8> { typedef uint2 type; };
template <> struct RawBytes<4> { typedef uint32_t type; };
template <> struct RawBytes<2> { typedef uint16_t type; };

template <typename T, int VEC>
__device__ __forceinline__ void load_vec(const T* p, float* f) {
  typedef typename RawBytes<VEC * sizeof(T)>::type raw_t;
  union { raw_t raw; T e[VEC]; } u;
  u.raw = *reinterpret_cast<const raw_t*>(p);
#pragma unroll
  for (int i = 0; i < VEC; ++i) f[i] = VecTraits<T>::load1(&u.e[i]);
}

// the same access split in two: keep the packed bytes in registers (1/2 the VGPRs of fp32 for bf16)
// while several loads are in flight, convert when the values are used
template <typename T, int VEC>
struct PackedVec {
  typedef typename RawBytes<VEC * sizeof(T)>::type raw_t;
  raw_t raw;
  __device__ __forceinline__ void load(const T* p) { raw = *reinterpret_cast<const raw_t*>(p); }
  __device__ __forceinline__ void unpack(float* f) const {
    union { raw_t r; T e[VEC]; } u;
    u.r = raw;
#pragma unroll
    for (int i = 0; i < VEC; ++i) f[i] = VecTraits<T>::load1(&u.e[i]);
  }
};

__device__ __forceinline__ void store1(float* p, float v) { *p = v; }
__device__ __forceinline__ void store1(bf16_t* p, float v) { *p = f32_to_bf16(v); }

template <typename T, int VEC>
__device__ __forceinline__ void store_vec(T* p, const float* f) {
  typedef typename RawBytes<VEC * sizeof(T)>::type raw_t;
  union { raw_t raw; T e[VEC]; } u;
#pragma unroll
  for (int i = 0; i < VEC; ++i) store1(&u.e[i], f[i]);
  *reinterpret_cast<raw_t*>(p) = u.raw;
}

// Sum over aligned groups of WIDTH lanes (WIDTH a power of two <= 64); every lane of the
// group ends with the group's sum.
template <int WIDTH>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
  for (int m = WIDTH / 2; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  return v;
}

// Workgroups are handed to the 8 XCDs round-robin in launch order, and each XCD has its own L2: when blocks i and i+1 read
// neighbouring rows of a map (pooling / depthwise / resampling halos), every shared row is fetched from HBM once per XCD
// (profiles/r03_pmc_traffic.json: stem pool backward 2.6x its algorithmic bytes). Such kernels take their block index from
// xcd_block_id(): XCD k walks the k-th contiguous eighth of the block range, so neighbours share an L2. The grid must be
// xcd_grid(n) (a multiple of 8); ids past the real block count fall out at the kernels' own bounds checks.
__device__ __forceinline__ unsigned xcd_block_id() {
  const unsigned per = gridDim.x >> 3;
  return (blockIdx.x & 7u) * per + (blockIdx.x >> 3);
}
static inline unsigned xcd_grid(unsigned n_blocks) { return 8u * ((n_blocks + 7u) / 8u); }

}  // namespace dfine
