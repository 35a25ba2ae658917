// Does a wavefront's VALU arithmetic stay correct while wavefronts of ANOTHER kernel (another stream) issue matrix-core
// instructions on the same SIMD?  Stand-alone reproducer for what scripts/diag_fp16_pipeline.py found in round 3: results
// of roi_align_rotated_kernel (plain VALU code, no LDS, no scratch, no atomics) came back wrong in lanes 48..63 of the LOW
// halves of its v_pk_*_f32 results whenever conv_h16_kernel (v_mfma_f32_16x16x32_f16) ran beside it.
//
//   victim  (stream 1): every lane iterates  x = fma(x, a, b)  on registers, `victim` selects the instruction form:
//                       0 scalar v_fma_f32, 1 packed v_pk_fma_f32, 2 packed v_pk_mul_f32 + v_pk_add_f32
//   aggressor (stream 2): a loop of MFMAs, `aggr` selects  0 none, 1 v_mfma_f32_16x16x32_f16, 2 v_mfma_f32_32x32x16_f16,
//                       3 v_mfma_f32_16x16x4_f32, 4 v_mfma_f32_16x16x32_bf16, 5 a VALU-only loop (control)
// The victim's output is compared bit for bit with its own solo run.
//   hipcc --offload-arch=gfx950 -O3 -o pk_vs_mfma scripts/micro/pk_vs_mfma.hip && ./pk_vs_mfma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int FORM>
__global__ __launch_bounds__(256) void victim_kernel(float* out, int iters) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  f32x2 x[4], a, b;
  for (int i = 0; i < 4; ++i) { x[i][0] = 0.001f * (float)(t % 977) + (float)i; x[i][1] = 0.002f * (float)(t % 613) - (float)i; }
  a[0] = 0.999f; a[1] = 1.0005f; b[0] = 0.0123f; b[1] = -0.0077f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if constexpr (FORM == 0) {
        float lo, hi;
        asm volatile("v_fma_f32 %0, %2, %4, %6\n\tv_fma_f32 %1, %3, %5, %7" : "=&v"(lo), "=&v"(hi)
                     : "v"(x[i][0]), "v"(x[i][1]), "v"(a[0]), "v"(a[1]), "v"(b[0]), "v"(b[1]));
        x[i][0] = lo; x[i][1] = hi;
      } else if constexpr (FORM == 1) {
        f32x2 r;
        asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(x[i]), "v"(a), "v"(b));
        x[i] = r;
      } else {
        f32x2 r;
        asm volatile("v_pk_mul_f32 %0, %1, %2\n\ts_nop 0\n\tv_pk_add_f32 %0, %0, %3" : "=&v"(r) : "v"(x[i]), "v"(a), "v"(b));
        x[i] = r;
      }
    }
  }
  for (int i = 0; i < 4; ++i) { out[(long)t * 8 + 2 * i] = x[i][0]; out[(long)t * 8 + 2 * i + 1] = x[i][1]; }
}

template <int KIND>
__global__ __launch_bounds__(256, 2) void aggressor_kernel(float* out, int iters) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if constexpr (KIND == 5) {
    float v[16];
    for (int i = 0; i < 16; ++i) v[i] = (float)(t & 63) * 0.5f + (float)i;
    for (int it = 0; it < iters * 8; ++it)
#pragma unroll
      for (int i = 0; i < 16; ++i) v[i] = __builtin_fmaf(v[i], 0.999f, v[(i + 1) & 15] * 1e-3f);
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += v[i];
    out[t] = s;
    return;
  } else {
    f32x4 acc4[16];
    f32x16 acc16[4];
    for (int i = 0; i < 16; ++i) acc4[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) acc16[i][e] = 0.f;
    h8 ha, hb; b8 ba, bb;
    for (int e = 0; e < 8; ++e) { ha[e] = (_Float16)(0.01f * (float)((t + e) & 15)); hb[e] = (_Float16)(0.02f * (float)((t * 3 + e) & 7));
                                  ba[e] = (__bf16)(0.01f * (float)((t + e) & 15)); bb[e] = (__bf16)(0.02f * (float)((t * 3 + e) & 7)); }
    const float fa = 0.01f * (float)(t & 15), fb = 0.02f * (float)(t & 7);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        if constexpr (KIND == 1) acc4[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, acc4[i], 0, 0, 0);
        else if constexpr (KIND == 2) acc16[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, acc16[i & 3], 0, 0, 0);
        else if constexpr (KIND == 3) acc4[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, fb, acc4[i], 0, 0, 0);
        else acc4[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ba, bb, acc4[i], 0, 0, 0);
      }
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += acc4[i][0] + acc4[i][1] + acc4[i][2] + acc4[i][3];
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) s += acc16[i][e];
    out[t] = s;
  }
}

static void launch_victim(int form, float* out, int blocks, int iters, hipStream_t s) {
  if (form == 0) hipLaunchKernelGGL(victim_kernel<0>, dim3(blocks), dim3(256), 0, s, out, iters);
  else if (form == 1) hipLaunchKernelGGL(victim_kernel<1>, dim3(blocks), dim3(256), 0, s, out, iters);
  else hipLaunchKernelGGL(victim_kernel<2>, dim3(blocks), dim3(256), 0, s, out, iters);
}
static void launch_aggr(int kind, float* out, int blocks, int iters, hipStream_t s) {
  switch (kind) {
    case 1: hipLaunchKernelGGL(aggressor_kernel<1>, dim3(blocks), dim3(256), 0, s, out, iters); break;
    case 2: hipLaunchKernelGGL(aggressor_kernel<2>, dim3(blocks), dim3(256), 0, s, out, iters); break;
    case 3: hipLaunchKernelGGL(aggressor_kernel<3>, dim3(blocks), dim3(256), 0, s, out, iters); break;
    case 4: hipLaunchKernelGGL(aggressor_kernel<4>, dim3(blocks), dim3(256), 0, s, out, iters); break;
    case 5: hipLaunchKernelGGL(aggressor_kernel<5>, dim3(blocks), dim3(256), 0, s, out, iters); break;
    default: break;
  }
}

int main(int argc, char** argv) {
  const int rounds = argc > 1 ? atoi(argv[1]) : 40;
  const int vblocks = 512, ablocks = 1024, viters = 4000, aiters = 4000;
  const size_t vn = (size_t)vblocks * 256 * 8;
  float *vout, *aout;
  CK(hipMalloc(&vout, vn * 4));
  CK(hipMalloc(&aout, (size_t)ablocks * 256 * 4));
  hipStream_t sv, sa;
  CK(hipStreamCreateWithFlags(&sv, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
  std::vector<float> ref(vn), got(vn);
  const char* vname[] = {"v_fma_f32 (scalar)", "v_pk_fma_f32", "v_pk_mul_f32 + v_pk_add_f32"};
  const char* aname[] = {"none", "v_mfma_f32_16x16x32_f16", "v_mfma_f32_32x32x16_f16", "v_mfma_f32_16x16x4_f32", "v_mfma_f32_16x16x32_bf16", "VALU-only loop"};
  for (int form = 0; form < 3; ++form) {
    launch_victim(form, vout, vblocks, viters, sv);
    CK(hipStreamSynchronize(sv));
    CK(hipMemcpy(ref.data(), vout, vn * 4, hipMemcpyDeviceToHost));
    for (int kind = 0; kind < 6; ++kind) {
      long bad_launches = 0, bad_vals = 0, lane_hist[4] = {0, 0, 0, 0}, half_hist[2] = {0, 0};
      for (int r = 0; r < rounds; ++r) {
        CK(hipMemsetAsync(vout, 0, vn * 4, sv));
        launch_aggr(kind, aout, ablocks, aiters, sa);
        launch_victim(form, vout, vblocks, viters, sv);
        launch_aggr(kind, aout, ablocks, aiters, sa);
        CK(hipStreamSynchronize(sv));
        CK(hipStreamSynchronize(sa));
        CK(hipMemcpy(got.data(), vout, vn * 4, hipMemcpyDeviceToHost));
        long nb = 0;
        for (size_t i = 0; i < vn; ++i)
          if (memcmp(&got[i], &ref[i], 4) != 0) {
            ++nb;
            const int lane = (int)((i / 8) & 63);
            ++lane_hist[lane >> 4];
            ++half_hist[i & 1];
          }
        if (nb) { ++bad_launches; bad_vals += nb; }
      }
      printf("victim %-28s | aggressor %-26s: %ld of %d victim launches wrong, %ld values; by lane group 0-15/16-31/32-47/48-63: %ld %ld %ld %ld; "
             "lo / hi half of the pair: %ld / %ld\n", vname[form], aname[kind], bad_launches, rounds, bad_vals, lane_hist[0], lane_hist[1],
             lane_hist[2], lane_hist[3], half_hist[0], half_hist[1]);
      fflush(stdout);
    }
  }
  return 0;
}
