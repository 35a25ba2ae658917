// Shared device helpers of the proposal kernels (proposals.hip, rpn.hip).
#pragma once
#include "common.h"

typedef unsigned long long u64;
typedef unsigned int u32;

constexpr float SCALE_CLAMP = 4.135166556742356f;  // log(1000/16), d2 Box2BoxTransformRotated

__device__ __forceinline__ u32 float_key(float f) {
  // order-preserving map float -> uint (larger float => larger key); -0.0 and +0.0 compare equal in
  // torch.sort, so both map to the key of +0.0 (a stable sort then orders them by index)
  u32 u = __float_as_uint(f);
  if (u == 0x80000000u) u = 0u;
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key_float(u32 k) {
  u32 u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
  return __uint_as_float(u);
}

// descending bitonic sort of `npad` (power of two) u64 in LDS by all threads of the block
__device__ inline void bitonic_sort_desc(u64* a, int npad) {
  for (int k = 2; k <= npad; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = threadIdx.x; i < npad; i += blockDim.x) {
        const int ixj = i ^ j;
        if (ixj > i) {
          const u64 x = a[i], y = a[ixj];
          const bool desc = (i & k) == 0;
          if (desc ? (x < y) : (x > y)) { a[i] = y; a[ixj] = x; }
        }
      }
      __syncthreads();
    }
  }
}

__device__ __forceinline__ float floor_mod(float a, float b) {  // torch.remainder semantics
  float m = fmodf(a, b);
  if (m != 0.f && ((b < 0.f) != (m < 0.f))) m += b;
  return m;
}

