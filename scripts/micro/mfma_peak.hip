// Ceiling probe: back-to-back v_mfma_f32_32x32x2_f32 on 16 independent accumulators, no memory traffic.
// hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_peak scripts/micro/mfma_peak.hip && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(256, 1) void k(float* out, int iters) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  float a = threadIdx.x * 1e-3f, b = threadIdx.x * 2e-3f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC> void run(int blocks_per_cu, int iters) {
  float* out; hipMalloc(&out, 256 * 8 * 256 * 4);
  int grid = 256 * blocks_per_cu;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<NACC><<<grid, 256>>>(out, 10); hipDeviceSynchronize();
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0); k<NACC><<<grid, 256>>>(out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flops = (double)grid * 4 * iters * 4.0 * NACC * 32 * 32 * 2 * 2;
    printf("NACC=%d blocks/CU=%d iters=%d: %.3f ms  %.1f TFLOP/s\n", NACC, blocks_per_cu, iters, ms, flops / ms / 1e9);
  }
  hipFree(out);
}
int main() {
  run<16>(1, 20000);    // ~1.3 s of MFMA per launch at peak -> sustained clock
  run<16>(1, 2000);
  run<4>(1, 8000);
  run<4>(2, 8000);
  run<8>(2, 4000);
  return 0;
}
