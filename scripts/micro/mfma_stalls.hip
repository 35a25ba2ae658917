// Does the GPU pause periodically under sustained MFMA load?  Launches ~3 s of back-to-back ~250 us kernels
// (pure MFMA, f32 or f16 operands), records an event after each, prints the launch-to-launch intervals that are
// more than 5 ms longer than the median, with their time offsets.
// hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_stalls scripts/micro/mfma_stalls.hip && /tmp/mfma_stalls
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
template <bool HALF>
__global__ __launch_bounds__(256, 1) void k(float* out, int iters) {
  f32x16 acc[8];
  for (int i = 0; i < 8; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  float a = threadIdx.x * 1e-3f, b = threadIdx.x * 2e-3f;
  h8 ah, bh;
  for (int e = 0; e < 8; ++e) { ah[e] = (_Float16)(a + e); bh[e] = (_Float16)(b - e); }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (HALF) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[i], 0, 0, 0);
        else acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
      }
  }
  float s = 0.f;
  for (int i = 0; i < 8; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <bool HALF> void run(int iters, int n) {
  float* out; hipMalloc(&out, 256 * 8 * 256 * 4);
  std::vector<hipEvent_t> ev(n + 1);
  for (auto& e : ev) hipEventCreate(&e);
  k<HALF><<<512, 256>>>(out, iters); hipDeviceSynchronize();
  hipEventRecord(ev[0]);
  for (int i = 0; i < n; ++i) { k<HALF><<<512, 256>>>(out, iters); hipEventRecord(ev[i + 1]); }
  hipDeviceSynchronize();
  std::vector<float> dt(n);
  for (int i = 0; i < n; ++i) hipEventElapsedTime(&dt[i], ev[i], ev[i + 1]);
  std::vector<float> s = dt; std::sort(s.begin(), s.end());
  float med = s[n / 2], tot = 0; int nst = 0; float stall = 0;
  for (int i = 0; i < n; ++i) {
    if (dt[i] > med + 5.f) { printf("  %s stall %.1f ms at t=%.0f ms (launch %d)\n", HALF ? "f16" : "f32", dt[i] - med, tot, i); ++nst; stall += dt[i] - med; }
    tot += dt[i];
  }
  printf("%s: %d launches, median %.3f ms, total %.0f ms, %d stalls = %.0f ms (%.1f%%)\n", HALF ? "f16" : "f32", n, med, tot, nst, stall, 100 * stall / tot);
  hipFree(out);
}
int main() {
  run<false>(400, 6000);
  run<true>(400, 6000);
  run<false>(400, 6000);
  return 0;
}
