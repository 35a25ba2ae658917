// What does s_memtime count on gfx950?  DESIGN (rounds 2-5) priced the F(4x4) kernel's phases in s_memtime ticks and called them
// shader cycles; the same blocks' wall time (s_memrealtime, the 100 MHz counter) put the tick at 2.0 GHz while hwmon and
// GRBM_GUI_ACTIVE say the shader clock is 2.3-2.4 GHz.  This probe runs a stream whose cycle count is architectural -
// back-to-back independent v_mfma_f32_32x32x2_f32 (16 passes x 4 = 64 cycles each per SIMD, MI355X_MICROARCH.md) and a
// dependent v_add_f32 chain - and reports, per MFMA: s_memtime ticks, nanoseconds (s_memrealtime), and the clocks they imply.
// hipcc --offload-arch=gfx950 -O3 -o /tmp/clock_calib scripts/micro/clock_calib.hip && /tmp/clock_calib
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256, 1) void k(float* out, unsigned long long* st, int iters) {
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  float a = threadIdx.x * 1e-3f, b = threadIdx.x * 2e-3f;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
  // the accumulators must be complete before the closing stamps
  asm volatile("s_nop 7\n\ts_nop 7" ::"v"(s));
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) { st[blockIdx.x * 2] = t1 - t0; st[blockIdx.x * 2 + 1] = r1 - r0; }
}

typedef float f32x4 __attribute__((ext_vector_type(4)));
// the F(4x4) kernel's instruction: v_mfma_f32_16x16x4_f32 (32 cycles per SIMD), 8 independent accumulators
__global__ __launch_bounds__(256, 1) void k16(float* out, unsigned long long* st, int iters) {
  f32x4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float a = threadIdx.x * 1e-3f, b = threadIdx.x * 2e-3f;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
  asm volatile("s_nop 7\n\ts_nop 7" ::"v"(s));
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) { st[blockIdx.x * 2] = t1 - t0; st[blockIdx.x * 2 + 1] = r1 - r0; }
}

__global__ __launch_bounds__(64, 1) void kv(float* out, unsigned long long* st, int iters) {
  float x = threadIdx.x;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 64; ++r) asm volatile("v_add_f32 %0, %0, %0" : "+v"(x));
  }
  asm volatile("s_nop 7" ::"v"(x));
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  out[blockIdx.x * 64 + threadIdx.x] = x;
  if (threadIdx.x == 0) { st[blockIdx.x * 2] = t1 - t0; st[blockIdx.x * 2 + 1] = r1 - r0; }
}

int main() {
  float* out; unsigned long long* st;
  hipMalloc(&out, 1024 * 256 * 4); hipMalloc(&st, 1024 * 16);
  std::vector<unsigned long long> h(2048);
  for (int grid : {1, 256}) {
    for (int iters : {2000, 20000}) {
      k<<<grid, 256>>>(out, st, 100); hipDeviceSynchronize();
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      hipEventRecord(e0); k<<<grid, 256>>>(out, st, iters); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      hipMemcpy(h.data(), st, grid * 16, hipMemcpyDeviceToHost);
      double ticks = 0, real = 0;
      for (int b = 0; b < grid; ++b) { ticks += h[2 * b]; real += h[2 * b + 1]; }
      ticks /= grid; real /= grid;
      const double n = 16.0 * iters, ns = real * 10.0;
      printf("mfma_f32_32x32x2 x %.0f per wave, %3d workgroup(s) of 4 waves: %.2f s_memtime ticks / MFMA, %.3f ns / MFMA (s_memrealtime, 100 MHz) "
             "-> s_memtime ticks at %.1f MHz; shader clock if 64 cycles / MFMA: %.0f MHz; kernel %.3f ms by events\n",
             n, grid, ticks / n, ns / n, ticks / ns * 1e3, 64.0 * n / ns * 1e3, ms);
    }
  }
  for (int grid : {1, 256}) {
    const int iters = 20000;
    k16<<<grid, 256>>>(out, st, 100); hipDeviceSynchronize();
    k16<<<grid, 256>>>(out, st, iters); hipDeviceSynchronize();
    hipMemcpy(h.data(), st, grid * 16, hipMemcpyDeviceToHost);
    double ticks = 0, real = 0;
    for (int b = 0; b < grid; ++b) { ticks += h[2 * b]; real += h[2 * b + 1]; }
    ticks /= grid; real /= grid;
    const double n = 32.0 * iters, ns = real * 10.0;
    printf("mfma_f32_16x16x4 x %.0f per wave, %3d workgroup(s) of 4 waves: %.2f s_memtime ticks / MFMA, %.3f ns / MFMA -> shader clock %.0f MHz; "
           "%.1f TFLOP/s\n", n, grid, ticks / n, ns / n, ticks / ns * 1e3, grid * 4 * n * 2048.0 / (ns * 1e-9) / 1e12);
  }
  for (int grid : {1, 1024}) {
    const int iters = 20000;
    kv<<<grid, 64>>>(out, st, 10); hipDeviceSynchronize();
    kv<<<grid, 64>>>(out, st, iters); hipDeviceSynchronize();
    hipMemcpy(h.data(), st, grid * 16, hipMemcpyDeviceToHost);
    double ticks = 0, real = 0;
    for (int b = 0; b < grid; ++b) { ticks += h[2 * b]; real += h[2 * b + 1]; }
    ticks /= grid; real /= grid;
    const double n = 64.0 * iters, ns = real * 10.0;
    printf("dependent v_add_f32 x %.0f, %4d single-wave workgroup(s): %.2f ticks / op, %.3f ns / op -> ticks at %.1f MHz\n", n, grid,
           ticks / n, ns / n, ticks / ns * 1e3);
  }
  return 0;
}
