// Float vs packed-bf16 global atomic add rate in the MSDA value-gradient access pattern: each group of 32 (f32) or 16 (pk bf16)
// lanes adds one 32-channel row (128 B / 64 B) at a pseudo-random pixel of a [rows][32] map.   hipcc -O3 --offload-arch=gfx950
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef short short2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned hash(unsigned x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }
__global__ void k_f32(float* buf, unsigned rows, int iters) {
  const unsigned tid = blockIdx.x * blockDim.x + threadIdx.x, grp = tid >> 5, ch = tid & 31;
  for (int i = 0; i < iters; ++i) atomicAdd(buf + (size_t)(hash(grp * 977u + i) % rows) * 32 + ch, 1.0f);
}
__global__ void k_pk(short2v* buf, unsigned rows, int iters) {
  typedef __attribute__((address_space(1))) short2v gptr;
  const unsigned tid = blockIdx.x * blockDim.x + threadIdx.x, grp = tid >> 4, ch2 = tid & 15;
  short2v one; one[0] = 0x3f80; one[1] = 0x3f80;
  for (int i = 0; i < iters; ++i) __builtin_amdgcn_global_atomic_fadd_v2bf16((gptr*)buf + (size_t)(hash(grp * 977u + i) % rows) * 16 + ch2, one);
}
int main() {
  const unsigned rows = 32u * 8400u * 8u;  // B * S * heads
  float* a; short2v* b;
  hipMalloc(&a, (size_t)rows * 32 * 4); hipMalloc(&b, (size_t)rows * 32 * 2);
  hipMemset(a, 0, (size_t)rows * 32 * 4); hipMemset(b, 0, (size_t)rows * 32 * 2);
  const long long adds = 32LL * 492 * 8 * 12 * 4;  // row adds of one decoder layer
  hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
  for (int rep = 0; rep < 2; ++rep) {
    const int iters = 48;
    long long groups = adds / iters;
    float ms;
    hipEventRecord(s); hipLaunchKernelGGL(k_f32, dim3((unsigned)(groups * 32 / 256)), dim3(256), 0, 0, a, rows, iters); hipEventRecord(e); hipEventSynchronize(e);
    hipEventElapsedTime(&ms, s, e); printf("f32   : %.1f us  %.2f G row-adds/s  %.0f GB/s\n", ms * 1e3, adds / ms / 1e6, adds * 128.0 / ms / 1e6);
    hipEventRecord(s); hipLaunchKernelGGL(k_pk, dim3((unsigned)(groups * 16 / 256)), dim3(256), 0, 0, b, rows, iters); hipEventRecord(e); hipEventSynchronize(e);
    hipEventElapsedTime(&ms, s, e); printf("pk bf16: %.1f us  %.2f G row-adds/s  %.0f GB/s\n", ms * 1e3, adds / ms / 1e6, adds * 64.0 / ms / 1e6);
  }
  return 0;
}
