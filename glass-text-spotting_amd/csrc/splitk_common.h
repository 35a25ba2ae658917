// The second half of a split-K launch (conv.hip: glass_conv2d_nhwc_splitk; winograd43.hip: glass_conv3x3_winograd43_splitk_nhwc):
// the slices' raw partial sums workspace[s][M][Cout] are added IN SLICE ORDER (deterministic), then bias, ReLU, residual as the
// single-slice epilogues do, and y is written with its strides.
#pragma once
#include "common.h"
#include <cstdint>

namespace {
struct SplitReduce {
  const float* ws; const float* bias; const float* res; float* y;
  long M; int Cout, splits, ldy, ycoff, ycs, ldr, relu, res_mode;
};
template <int V>   // V = 4: four channels per thread (unit channel stride, aligned), 1: one
__global__ __launch_bounds__(256) void splitk_reduce_kernel(SplitReduce q) {
  const long per = q.M * q.Cout;
  const int cv = q.Cout / V;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < per / V; i += (long)gridDim.x * blockDim.x) {
    const long m = i / cv;
    const int c = (int)(i - m * cv) * V;
    float a[V];
#pragma unroll
    for (int e = 0; e < V; ++e) a[e] = 0.f;
    for (int s = 0; s < q.splits; ++s) {
      if constexpr (V == 4) {
        const float4 v = *reinterpret_cast<const float4*>(q.ws + (long)s * per + m * q.Cout + c);
        a[0] += v.x; a[1] += v.y; a[2] += v.z; a[3] += v.w;
      } else {
        a[0] += q.ws[(long)s * per + m * q.Cout + c];
      }
    }
#pragma unroll
    for (int e = 0; e < V; ++e) {
      float v = a[e] + (q.bias ? q.bias[c + e] : 0.f);
      if (q.relu == 2) v = fmaxf(v, 0.f);
      if (q.res_mode == 1) v += q.res[m * q.ldr + c + e];
      if (q.relu == 1) v = fmaxf(v, 0.f);
      a[e] = v;
    }
    if constexpr (V == 4) *reinterpret_cast<float4*>(q.y + m * q.ldy + q.ycoff + c) = make_float4(a[0], a[1], a[2], a[3]);
    else q.y[m * q.ldy + q.ycoff + (long)c * q.ycs] = a[0];
  }
}

// launches the reduction on `s`; y / residual with the caller's strides (ycs = channel stride of y)
static inline void launch_splitk_reduce(const float* ws, const float* bias, const float* res, float* y, long M, int Cout, int splits,
                                        int ldy, int ycoff, int ycs, int ldr, int relu, int res_mode, hipStream_t s) {
  SplitReduce q;
  q.ws = ws; q.bias = bias; q.res = res_mode ? res : nullptr; q.y = y;
  q.M = M; q.Cout = Cout; q.splits = splits; q.ldy = ldy; q.ycoff = ycoff; q.ycs = ycs; q.ldr = ldr; q.relu = relu; q.res_mode = res_mode;
  const bool vec = Cout % 4 == 0 && ycs == 1 && ldy % 4 == 0 && ycoff % 4 == 0 && ((uintptr_t)y & 15) == 0;
  const long n = M * Cout / (vec ? 4 : 1);
  const int blocks = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
  if (vec) hipLaunchKernelGGL(splitk_reduce_kernel<4>, dim3(blocks), dim3(256), 0, s, q);
  else hipLaunchKernelGGL(splitk_reduce_kernel<1>, dim3(blocks), dim3(256), 0, s, q);
}
}  // namespace
