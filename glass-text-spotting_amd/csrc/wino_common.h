// Helpers shared by the Winograd kernels (winograd.hip: F(2x2,3x3); winograd43.hip: F(4x4,3x3)).
#pragma once
#include "common.h"
#include <cstdint>
#include <cstdlib>
#include <type_traits>
#include <utility>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr unsigned OOB = 0x7fffffffu;

struct WinoParams {
  const float* x;
  const float* u;
  const float* bias;
  const float* res;
  float* y;
  int N, H, W, Cin, Cout, TH, TW, ntiles, nk;   // nk: k-tiles a workgroup walks (a k-SLICE of the layer when gridDim.y > 1)
  int nkt;                         // k-tiles of the whole layer (stride of the packed weights); F(4x4) split-K: slice blockIdx.y starts at blockIdx.y * nk
  long y_slice;                    // F(4x4) split-K: floats between the slices' partial outputs (y = workspace + blockIdx.y * y_slice)
  int ldx, ldy, ycoff, ldr, relu, res_mode;
  int tiles_m, tiles_n;
  unsigned x_bytes, u_bytes, y_bytes, r_bytes;
  unsigned magic_tpi, magic_tw;    // floor(2^32 / d) for the two tile-index divisions (fast_div)
  unsigned long long* dbg;         // timing-instrumented builds only (GLASS_W43_ABL=4): per-workgroup cycle stamps
};

__device__ __forceinline__ float4 add4(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 sub4(float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
// t / d for 0 <= t < 2^31 with magic = floor(2^32 / d): the mul-high estimate is low by at most one (6 instructions
// instead of the ~40 of a 32-bit integer division; the epilogue alone did eight of those per thread = 1.5 us).
__device__ __forceinline__ int fast_div(int t, int d, unsigned magic) {
  int q = d == 1 ? t : (int)__umulhi((unsigned)t, magic);
  if (t - q * d >= d) ++q;
  return q;
}

// compile-time loops: the issue order below is written as straight-line code with static register indices
template <int I> using ic = std::integral_constant<int, I>;
template <int... Is, class F> __device__ __forceinline__ void static_for_impl(std::integer_sequence<int, Is...>, F&& f) {
  (f(ic<Is>{}), ...);
}
template <int N, class F> __device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(std::make_integer_sequence<int, N>{}, static_cast<F&&>(f));
}

__device__ __forceinline__ float comp(const float4& v, int s) { return s == 0 ? v.x : s == 1 ? v.y : s == 2 ? v.z : v.w; }

}  // namespace
