// Do f32 MFMAs and f32 VALU work of ANOTHER wave on the same SIMD overlap, or do they share the vector ALUs?
// 512-thread workgroups (2 waves per SIMD), 1 per CU: waves 0-3 run an MFMA loop, waves 4-7 run (a) nothing,
// (b) a v_pk_add_f32 / v_add_f32 loop, (c) an LDS write loop.  Reports the MFMA rate of waves 0-3 in each case.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(512, 1) void k(float* out, int iters, int mode, long* tcycles) {
  __shared__ float lds[8192];
  const int wave = threadIdx.x >> 6;
  if (wave < 4) {
    f32x16 acc[8];
    for (int i = 0; i < 8; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    float a = threadIdx.x * 1e-3f, b = threadIdx.x * 2e-3f;
    const long t0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    const long t1 = wall_clock64();
    float s = 0.f;
    for (int i = 0; i < 8; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) tcycles[0] = t1 - t0;
  } else if (mode == 1) {          // scalar f32 VALU adds
    float v[16];
    for (int i = 0; i < 16; ++i) v[i] = threadIdx.x * 0.5f + i;
    for (int it = 0; it < iters * 4; ++it) {
#pragma unroll
      for (int i = 0; i < 16; ++i) v[i] = v[i] + v[(i + 1) & 15];
    }
    float s = 0.f; for (int i = 0; i < 16; ++i) s += v[i];
    out[blockIdx.x * 512 + threadIdx.x] = s;
  } else if (mode == 2) {          // packed f32 adds
    f32x2 v[8];
    for (int i = 0; i < 8; ++i) { v[i][0] = threadIdx.x * 0.5f + i; v[i][1] = i; }
    for (int it = 0; it < iters * 4; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = v[i] + v[(i + 1) & 7];
    }
    float s = 0.f; for (int i = 0; i < 8; ++i) s += v[i][0] + v[i][1];
    out[blockIdx.x * 512 + threadIdx.x] = s;
  } else if (mode == 3) {          // LDS 16-byte writes
    float4 v = make_float4(threadIdx.x, 1.f, 2.f, 3.f);
    for (int it = 0; it < iters * 2; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) *reinterpret_cast<float4*>(&lds[((threadIdx.x - 256) * 4 + i * 1024) & 8191]) = v;
      v.x += 1.f;
    }
    __syncthreads;
    out[blockIdx.x * 512 + threadIdx.x] = lds[threadIdx.x];
  } else if (mode == 4) {          // integer VALU
    int v[16];
    for (int i = 0; i < 16; ++i) v[i] = threadIdx.x + i;
    for (int it = 0; it < iters * 4; ++it) {
#pragma unroll
      for (int i = 0; i < 16; ++i) v[i] = v[i] + (v[(i + 1) & 15] ^ it);
    }
    int s = 0; for (int i = 0; i < 16; ++i) s += v[i];
    out[blockIdx.x * 512 + threadIdx.x] = (float)s;
  } else {
    out[blockIdx.x * 512 + threadIdx.x] = 0.f;
  }
}
int main() {
  float* out; long* tc; (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&tc, 8);
  const char* names[] = {"idle partner", "partner: v_add_f32 loop", "partner: v_pk_add_f32 loop", "partner: ds_write_b128 loop", "partner: integer VALU loop"};
  const int iters = 4000;
  for (int mode = 0; mode < 5; ++mode) {
    k<<<256, 512>>>(out, 100, mode, tc); (void)hipDeviceSynchronize();
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0); k<<<256, 512>>>(out, iters, mode, tc); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    long t; (void)hipMemcpy(&t, tc, 8, hipMemcpyDeviceToHost);
    const double flops = 256.0 * 4 * iters * 64.0 * 32 * 32 * 2 * 2;
    printf("%-32s kernel %.3f ms; MFMA waves busy %.3f ms -> %.1f TFLOP/s from the MFMA waves\n", names[mode], ms, t * 1e-5, flops / (t * 1e-5) / 1e9);
  }
  return 0;
}
