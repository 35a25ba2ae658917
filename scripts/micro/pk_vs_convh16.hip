// Second stand-alone reproducer (see pk_vs_mfma.hip): the aggressor is the product's own conv_h16_kernel, called through the
// C ABI of libglass_hip.so on a second stream; the victim is a register-only, SELF-CHECKING kernel - every lane iterates the
// same recurrence twice, once with packed-f32 instructions and once with their scalar equivalents, and counts the lanes /
// halves whose bits differ at the end.  Forms:
//   1  v_pk_fma_f32 (plain)                        2  v_pk_mul_f32 op_sel:[0,1] op_sel_hi:[0,1] (broadcast) + v_pk_fma_f32
//   3  v_pk_add_f32 neg_lo/neg_hi + v_pk_fma_f32   4  form 1 with a third of the lanes switched off (EXEC)
//   5  v_pk_fma_f32 op_sel_hi:[0,1,1]   6 / 7  v_pk_mul_f32 with only op_sel / only op_sel_hi   8  v_pk_mov_b32 op_sel
//   9  SDWA (v_cvt_f32_f16_sdwa src0_sel:WORD_1)   10  DPP (v_mov_b32_dpp quad_perm)
//   hipcc --offload-arch=gfx950 -O3 -I include -o pk_vs_convh16 scripts/micro/pk_vs_convh16.hip -ldl
//   ./pk_vs_convh16 glass-text-spotting_amd/libglass_hip.so [rounds] [aggressor: 1 conv 3x3 | 2 none | 3 conv 1x1 | 4 none, 16x victim grid | 5 VALU spinner | 6 + 64 KB LDS | 7 ~240 VGPRs]
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "glass_hip.h"

typedef float f32x2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct Report { unsigned long long bad_lo, bad_hi, lanes[4], waves, all_base[64], bad_base[64], all_slot[16], bad_slot[16], all_simd[4], bad_simd[4]; };

template <int FORM>
__global__ __launch_bounds__(256) void victim_kernel(Report* rep, int iters) {
  const int t = blockIdx.x * 256 + threadIdx.x, lane = threadIdx.x & 63;
  f32x2 x, a, b, c;
  x[0] = 0.001f * (float)(t % 977) + 1.f; x[1] = 0.002f * (float)(t % 613) - 1.f;
  a[0] = 0.999f; a[1] = 1.0005f; b[0] = 0.0123f; b[1] = -0.0077f; c[0] = 0.9991f; c[1] = 0.9985f;
  float slo = x[0], shi = x[1];
  const bool on = FORM != 4 || (lane % 3) != 0;
  if (on) {
    for (int it = 0; it < iters; ++it) {
      if constexpr (FORM == 1 || FORM == 4) {
        asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));
        asm volatile("v_fma_f32 %0, %0, %2, %4\n\tv_fma_f32 %1, %1, %3, %5" : "+v"(slo), "+v"(shi) : "v"(a[0]), "v"(a[1]), "v"(b[0]), "v"(b[1]));
      } else if constexpr (FORM == 2) {
        f32x2 m;
        asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[0,1]\n\ts_nop 1\n\tv_pk_fma_f32 %1, %0, %3, %4"
                     : "=&v"(m), "+v"(x) : "v"(a), "v"(c), "v"(b));
        float sm;
        asm volatile("v_mul_f32 %0, %1, %3\n\ts_nop 1\n\tv_fma_f32 %1, %0, %4, %6\n\tv_fma_f32 %2, %0, %5, %7"
                     : "=&v"(sm), "+v"(slo), "=&v"(shi) : "v"(a[1]), "v"(c[0]), "v"(c[1]), "v"(b[0]), "v"(b[1]));
      } else if constexpr (FORM == 6 || FORM == 7) {
        // only the LO selectors (op_sel) / only the HI selectors (op_sel_hi) differ from the default
        f32x2 m;
        if constexpr (FORM == 6) asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]\n\ts_nop 1\n\tv_pk_fma_f32 %1, %0, %3, %4" : "=&v"(m), "+v"(x) : "v"(a), "v"(c), "v"(b));
        else asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]\n\ts_nop 1\n\tv_pk_fma_f32 %1, %0, %3, %4" : "=&v"(m), "+v"(x) : "v"(a), "v"(c), "v"(b));
        // 6: m = (x.lo a.hi, x.hi a.hi)   7: m = (x.lo a.lo, x.lo a.hi)
        float m0, m1;
        if constexpr (FORM == 6) asm volatile("v_mul_f32 %0, %2, %5\n\tv_mul_f32 %1, %3, %5" : "=&v"(m0), "=&v"(m1) : "v"(slo), "v"(shi), "v"(a[0]), "v"(a[1]));
        else asm volatile("v_mul_f32 %0, %2, %4\n\tv_mul_f32 %1, %2, %5" : "=&v"(m0), "=&v"(m1) : "v"(slo), "v"(shi), "v"(a[0]), "v"(a[1]));
        asm volatile("s_nop 1\n\tv_fma_f32 %0, %2, %4, %6\n\tv_fma_f32 %1, %3, %5, %7" : "=&v"(slo), "=&v"(shi) : "v"(m0), "v"(m1), "v"(c[0]), "v"(c[1]), "v"(b[0]), "v"(b[1]));
      } else if constexpr (FORM == 8) {
        // v_pk_mov_b32 with op_sel (swap the halves), then the plain recurrence
        f32x2 sw;
        asm volatile("v_pk_mov_b32 %0, %1, %1 op_sel:[1,0]\n\ts_nop 1\n\tv_pk_fma_f32 %1, %0, %2, %3" : "=&v"(sw), "+v"(x) : "v"(a), "v"(b));
        float n0, n1;
        asm volatile("v_fma_f32 %0, %3, %4, %6\n\tv_fma_f32 %1, %2, %5, %7" : "=&v"(n0), "=&v"(n1) : "v"(slo), "v"(shi), "v"(a[0]), "v"(a[1]), "v"(b[0]), "v"(b[1]));
        slo = n0; shi = n1;
      } else if constexpr (FORM == 9) {
        // SDWA sub-dword select (the fp16 kernels' own epilogues use v_cvt_f32_f16_sdwa): convert the HIGH half word
        const unsigned packed = (__float_as_uint(x[0]) & 0xffff0000u) | 0x3c00u;       // hi = top bits of x.lo as an fp16 pattern
        float c_sdwa, c_ref;
        asm volatile("v_cvt_f32_f16_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(c_sdwa) : "v"(packed));
        asm volatile("v_lshrrev_b32 %0, 16, %1\n\ts_nop 1\n\tv_cvt_f32_f16 %0, %0" : "=&v"(c_ref) : "v"(packed));
        c_sdwa = (c_sdwa != c_sdwa || c_sdwa > 1e4f || c_sdwa < -1e4f) ? 0.5f : c_sdwa;
        c_ref = (c_ref != c_ref || c_ref > 1e4f || c_ref < -1e4f) ? 0.5f : c_ref;
        x[0] = __builtin_fmaf(x[0], 0.999f, 1e-3f * c_sdwa); x[1] = x[0];
        slo = __builtin_fmaf(slo, 0.999f, 1e-3f * c_ref); shi = slo;
      } else if constexpr (FORM == 10) {
        // DPP (wave reductions): neighbour exchange by quad_perm vs the LDS-crossbar path (ds_bpermute)
        float nb_dpp;
        asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=v"(nb_dpp) : "v"(x[0]));
        const float nb_ref = __uint_as_float(__builtin_amdgcn_ds_bpermute((lane ^ 1) * 4, __float_as_uint(slo)));
        x[0] = __builtin_fmaf(nb_dpp, 0.999f, 0.0123f); x[1] = x[0];
        slo = __builtin_fmaf(nb_ref, 0.999f, 0.0123f); shi = slo;
      } else if constexpr (FORM == 5) {
        // x = fma(bcast(a.lo), x, b): the compiler's idiom for scalar * vector
        asm volatile("v_pk_fma_f32 %0, %1, %0, %2 op_sel_hi:[0,1,1]" : "+v"(x) : "v"(a), "v"(b));
        asm volatile("v_fma_f32 %0, %2, %0, %3\n\tv_fma_f32 %1, %2, %1, %4" : "+v"(slo), "+v"(shi) : "v"(a[0]), "v"(b[0]), "v"(b[1]));
      } else {
        f32x2 d;
        asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]\n\ts_nop 1\n\tv_pk_fma_f32 %1, %0, %3, %2"
                     : "=&v"(d), "+v"(x) : "v"(b), "v"(a));
        float d0, d1;
        asm volatile("v_sub_f32 %0, %2, %6\n\tv_sub_f32 %1, %3, %7\n\ts_nop 1\n\tv_fma_f32 %2, %0, %4, %6\n\tv_fma_f32 %3, %1, %5, %7"
                     : "=&v"(d0), "=&v"(d1), "+v"(slo), "+v"(shi) : "v"(a[0]), "v"(a[1]), "v"(b[0]), "v"(b[1]));
      }
    }
  }
  const bool blo = __float_as_uint(x[0]) != __float_as_uint(slo), bhi = __float_as_uint(x[1]) != __float_as_uint(shi);
  if (blo) atomicAdd(&rep->bad_lo, 1ull);
  if (bhi) atomicAdd(&rep->bad_hi, 1ull);
  if (blo || bhi) atomicAdd(&rep->lanes[lane >> 4], 1ull);
  const bool wave_bad = __ballot(blo || bhi) != 0;
  if (lane == 0) {
    // where the wavefront lives: HW_REG_GPR_ALLOC (id 5) VGPR_BASE[5:0] in units of 8 registers; HW_REG_HW_ID (id 4) wave slot [3:0], SIMD [5:4]
    const unsigned alloc = __builtin_amdgcn_s_getreg(5 | (31 << 11)), hwid = __builtin_amdgcn_s_getreg(4 | (31 << 11));
    const int base = alloc & 63, slot = hwid & 15, simd = (hwid >> 4) & 3;
    atomicAdd(&rep->all_base[base], 1ull); atomicAdd(&rep->all_slot[slot], 1ull); atomicAdd(&rep->all_simd[simd], 1ull);
    if (wave_bad) { atomicAdd(&rep->waves, 1ull); atomicAdd(&rep->bad_base[base], 1ull); atomicAdd(&rep->bad_slot[slot], 1ull); atomicAdd(&rep->bad_simd[simd], 1ull); }
  }
}

// synthetic aggressors: OCC 0 = few registers, no LDS; 1 = 64 KB of LDS; 2 = ~240 live VGPRs; both spin on plain VALU work
template <int OCC>
__global__ __launch_bounds__(256, 2) void occupier_kernel(float* out, int iters) {
  __shared__ float lds[OCC == 1 ? 16384 : 1];
  const int t = blockIdx.x * 256 + threadIdx.x;
  constexpr int NV = OCC == 2 ? 224 : 8;
  float v[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) v[i] = (float)(t & 63) * 0.5f + (float)i;
  if (OCC == 1) lds[threadIdx.x] = v[0];
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = __builtin_fmaf(v[i], 0.999f, v[(i + 1) % NV] * 1e-3f);
  }
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) sum += v[i];
  if (OCC == 1) sum += lds[(threadIdx.x * 7) & 16383];
  out[t] = sum;
}

// instruction-class spinners: which instruction of the aggressor matters?
//   10 v_pk_add_f32 (default op_sel)   11 v_pk_mul_f32 op_sel:[1,0] op_sel_hi:[1,0]   12 v_mfma_f32_16x16x32_f16
//   13 v_cvt_pk_f16_f32 + v_max3_f32   14 ds_write_b128 / ds_read_b128 + s_barrier     15 buffer_load_dwordx4
typedef _Float16 mh8 __attribute__((ext_vector_type(8)));
typedef float mf4 __attribute__((ext_vector_type(4)));
typedef float mf16 __attribute__((ext_vector_type(16)));
typedef __bf16 mb8 __attribute__((ext_vector_type(8)));
template <int KIND>
__global__ __launch_bounds__(256, 2) void spinner_kernel(float* out, const float* src, int iters) {
  __shared__ mf4 lds[KIND == 14 ? 1024 : 1];
  const int t = blockIdx.x * 256 + threadIdx.x;
  f32x2 p, q; p[0] = 1.f + (float)(t & 7); p[1] = 2.f; q[0] = 1e-3f; q[1] = -1e-3f;
  mf4 acc = {0.f, 0.f, 0.f, 0.f};
  mf16 acc16;
  for (int e = 0; e < 16; ++e) acc16[e] = 0.f;
  mh8 ha, hb;
  for (int e = 0; e < 8; ++e) { ha[e] = (_Float16)(0.01f * (float)((t + e) & 15)); hb[e] = (_Float16)(0.02f * (float)((t * 3 + e) & 7)); }
  float f0 = (float)(t & 31), f1 = 0.5f;
  for (int it = 0; it < iters; ++it) {
    if constexpr (KIND == 10) {
      asm volatile("v_pk_add_f32 %0, %0, %1\n\tv_pk_add_f32 %0, %0, %1\n\tv_pk_add_f32 %0, %0, %1\n\tv_pk_add_f32 %0, %0, %1" : "+v"(p) : "v"(q));
    } else if constexpr (KIND == 11) {
      asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel:[1,0] op_sel_hi:[1,0]\n\ts_nop 1\n\tv_pk_mul_f32 %0, %0, %1 op_sel:[1,0] op_sel_hi:[1,0]" : "+v"(p) : "v"(q));
    } else if constexpr (KIND == 12) {
      acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, acc, 0, 0, 0);
    } else if constexpr (KIND == 16) {
      acc16 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, acc16, 0, 0, 0);
    } else if constexpr (KIND == 17) {
      acc = __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_shufflevector(ha, ha, 0, 1, 2, 3), __builtin_shufflevector(hb, hb, 0, 1, 2, 3), acc, 0, 0, 0);
    } else if constexpr (KIND == 18) {
      acc16 = __builtin_amdgcn_mfma_f32_32x32x2f32(f0, f1, acc16, 0, 0, 0);
    } else if constexpr (KIND == 19) {
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(f0, f1, acc, 0, 0, 0);
    } else if constexpr (KIND == 20) {
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(mb8, ha), __builtin_bit_cast(mb8, hb), acc, 0, 0, 0);
    } else if constexpr (KIND == 13) {
      unsigned r;
      asm volatile("v_cvt_pk_f16_f32 %0, %1, %2\n\tv_max3_f32 %1, %1, %2, %2" : "=&v"(r), "+v"(f0) : "v"(f1));
      acc[0] += (float)r;
    } else if constexpr (KIND == 14) {
      lds[threadIdx.x] = acc;
      __syncthreads();
      acc += lds[(threadIdx.x + 17) & 255];
      __syncthreads();
    } else {
      acc += *reinterpret_cast<const mf4*>(src + (((long)t * 4 + (long)it * 1024) & 0xffffc));
    }
  }
  out[t] = acc[0] + acc[1] + acc[2] + acc[3] + p[0] + p[1] + f0 + acc16[0] + acc16[7] + acc16[15];
}

static void launch_victim(int form, Report* rep, int blocks, int iters, hipStream_t s) {
  switch (form) {
    case 1: hipLaunchKernelGGL(victim_kernel<1>, dim3(blocks), dim3(256), 0, s, rep, iters); break;
    case 2: hipLaunchKernelGGL(victim_kernel<2>, dim3(blocks), dim3(256), 0, s, rep, iters); break;
    case 3: hipLaunchKernelGGL(victim_kernel<3>, dim3(blocks), dim3(256), 0, s, rep, iters); break;
    case 5: hipLaunchKernelGGL(victim_kernel<5>, dim3(blocks), dim3(256), 0, s, rep, iters); break;
    case 6: hipLaunchKernelGGL(victim_kernel<6>, dim3(blocks), dim3(256), 0, s, rep, iters); break;
    case 7: hipLaunchKernelGGL(victim_kernel<7>, dim3(blocks), dim3(256), 0, s, rep, iters); break;
    case 8: hipLaunchKernelGGL(victim_kernel<8>, dim3(blocks), dim3(256), 0, s, rep, iters); break;
    case 9: hipLaunchKernelGGL(victim_kernel<9>, dim3(blocks), dim3(256), 0, s, rep, iters); break;
    case 10: hipLaunchKernelGGL(victim_kernel<10>, dim3(blocks), dim3(256), 0, s, rep, iters); break;
    default: hipLaunchKernelGGL(victim_kernel<4>, dim3(blocks), dim3(256), 0, s, rep, iters); break;
  }
}

int main(int argc, char** argv) {
  const char* so = argc > 1 ? argv[1] : "glass-text-spotting_amd/libglass_hip.so";
  const int rounds = argc > 2 ? atoi(argv[2]) : 200;
  const int aggr = argc > 3 ? atoi(argv[3]) : 1;
  void* h = dlopen(so, RTLD_NOW);
  if (!h) { fprintf(stderr, "dlopen %s: %s\n", so, dlerror()); return 1; }
  auto pack = (decltype(&glass_conv_h16_pack_weights))dlsym(h, "glass_conv_h16_pack_weights");
  auto conv = (decltype(&glass_conv2d_nhwc_h16_packed))dlsym(h, "glass_conv2d_nhwc_h16_packed");
  auto err = (const char* (*)())dlsym(h, "glass_last_error");
  if (!pack || !conv) { fprintf(stderr, "symbols missing\n"); return 1; }

  const int N = 8, H = 64, W = 64, C = 256, KS = aggr == 3 ? 1 : 3;
  glass_conv_desc d;
  memset(&d, 0, sizeof d);
  d.N = N; d.H = H; d.W = W; d.Cin = C; d.Cout = C; d.KH = KS; d.KW = KS; d.stride_h = d.stride_w = 1; d.pad_h = d.pad_w = KS / 2;
  d.Ho = H; d.Wo = W; d.ldx = C; d.ldy = C; d.y_coff = 0; d.y_cstride = 1; d.relu = 1; d.res_mode = 0; d.ldr = 0;
  const size_t xn = (size_t)N * H * W * C, wn = (size_t)C * KS * KS * C;
  std::vector<_Float16> hx(xn);
  std::vector<float> hw(wn);
  for (size_t i = 0; i < xn; ++i) hx[i] = (_Float16)(0.01f * (float)((i * 7919u) % 200u) - 1.f);
  for (size_t i = 0; i < wn; ++i) hw[i] = 0.001f * (float)((i * 104729u) % 100u) - 0.05f;
  void *x, *u, *y; float* w;
  CK(hipMalloc(&x, xn * 2)); CK(hipMalloc(&y, xn * 2)); CK(hipMalloc(&u, wn * 2)); CK(hipMalloc(&w, wn * 4));
  CK(hipMemcpy(x, hx.data(), xn * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(w, hw.data(), wn * 4, hipMemcpyHostToDevice));
  hipStream_t sv, sa;
  CK(hipStreamCreateWithFlags(&sv, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
  if (pack(w, C, KS, KS, C, u, sa) != 0) { fprintf(stderr, "pack: %s\n", err()); return 1; }
  CK(hipStreamSynchronize(sa));
  Report* rep;
  CK(hipMalloc(&rep, sizeof(Report)));
  const char* fname[] = {"", "v_pk_fma_f32", "v_pk_mul_f32 op_sel bcast + v_pk_fma_f32", "v_pk_add_f32 neg + v_pk_fma_f32", "v_pk_fma_f32, 1/3 of lanes off", "v_pk_fma_f32 op_sel_hi:[0,1,1] (scalar bcast)", "v_pk_mul_f32 op_sel:[0,1] only", "v_pk_mul_f32 op_sel_hi:[0,1] only", "v_pk_mov_b32 op_sel:[1,0]", "v_cvt_f32_f16_sdwa src0_sel:WORD_1", "v_mov_b32_dpp quad_perm"};
  for (int form = 1; form <= 10; ++form) {
    CK(hipMemset(rep, 0, sizeof(Report)));
    for (int r = 0; r < rounds; ++r) {
      if ((aggr == 1 || aggr == 3) && conv(&d, x, u, nullptr, nullptr, y, 3, sa) != 0) { fprintf(stderr, "conv: %s\n", err()); return 1; }
      if (aggr == 5) hipLaunchKernelGGL(occupier_kernel<0>, dim3(512), dim3(256), 0, sa, (float*)y, 20000);
      if (aggr == 6) hipLaunchKernelGGL(occupier_kernel<1>, dim3(512), dim3(256), 0, sa, (float*)y, 20000);
      if (aggr == 7) hipLaunchKernelGGL(occupier_kernel<2>, dim3(512), dim3(256), 0, sa, (float*)y, 600);
      switch (aggr) {
        case 10: hipLaunchKernelGGL(spinner_kernel<10>, dim3(1024), dim3(256), 0, sa, (float*)y, (const float*)x, 20000); break;
        case 11: hipLaunchKernelGGL(spinner_kernel<11>, dim3(1024), dim3(256), 0, sa, (float*)y, (const float*)x, 20000); break;
        case 12: hipLaunchKernelGGL(spinner_kernel<12>, dim3(1024), dim3(256), 0, sa, (float*)y, (const float*)x, 20000); break;
        case 16: hipLaunchKernelGGL(spinner_kernel<16>, dim3(1024), dim3(256), 0, sa, (float*)y, (const float*)x, 20000); break;
        case 17: hipLaunchKernelGGL(spinner_kernel<17>, dim3(1024), dim3(256), 0, sa, (float*)y, (const float*)x, 20000); break;
        case 18: hipLaunchKernelGGL(spinner_kernel<18>, dim3(1024), dim3(256), 0, sa, (float*)y, (const float*)x, 10000); break;
        case 19: hipLaunchKernelGGL(spinner_kernel<19>, dim3(1024), dim3(256), 0, sa, (float*)y, (const float*)x, 20000); break;
        case 20: hipLaunchKernelGGL(spinner_kernel<20>, dim3(1024), dim3(256), 0, sa, (float*)y, (const float*)x, 20000); break;
        case 13: hipLaunchKernelGGL(spinner_kernel<13>, dim3(1024), dim3(256), 0, sa, (float*)y, (const float*)x, 20000); break;
        case 14: hipLaunchKernelGGL(spinner_kernel<14>, dim3(1024), dim3(256), 0, sa, (float*)y, (const float*)x, 4000); break;
        case 15: hipLaunchKernelGGL(spinner_kernel<15>, dim3(1024), dim3(256), 0, sa, (float*)y, (const float*)x, 4000); break;
        default: break;
      }
      launch_victim(form, rep, aggr == 4 ? 16384 : 1024, 3000, sv);
    }
    CK(hipDeviceSynchronize());
    Report hr;
    CK(hipMemcpy(&hr, rep, sizeof hr, hipMemcpyDeviceToHost));
    printf("aggressor %s | victim %-44s: %d launches: lo halves wrong %llu, hi halves wrong %llu, waves hit %llu; lanes 0-15/16-31/32-47/48-63: %llu %llu %llu %llu\n",
           aggr == 2 ? "none    " : aggr == 3 ? "conv 1x1" : aggr == 4 ? "none, victim grid x16" : aggr == 5 ? "VALU spinner" : aggr == 6 ? "VALU spinner + 64 KB LDS" : aggr == 7 ? "VALU spinner, ~240 VGPRs" : aggr == 10 ? "v_pk_add_f32 spinner" : aggr == 11 ? "v_pk_mul_f32 op_sel:[1,0] spinner" : aggr == 12 ? "v_mfma_f32_16x16x32_f16 spinner" : aggr == 16 ? "v_mfma_f32_32x32x16_f16 spinner" : aggr == 17 ? "v_mfma_f32_16x16x16_f16 spinner" : aggr == 18 ? "v_mfma_f32_32x32x2_f32 spinner" : aggr == 19 ? "v_mfma_f32_16x16x4_f32 spinner" : aggr == 20 ? "v_mfma_f32_16x16x32_bf16 spinner" : aggr == 13 ? "v_cvt_pk_f16_f32 + v_max3 spinner" : aggr == 14 ? "LDS + barrier spinner" : aggr == 15 ? "buffer/global load spinner" : "conv 3x3", fname[form], rounds, hr.bad_lo, hr.bad_hi, hr.waves, hr.lanes[0], hr.lanes[1], hr.lanes[2], hr.lanes[3]);
    if (hr.waves) {
      printf("   wrong / all wavefronts by VGPR base (x8 registers):");
      for (int i = 0; i < 64; ++i) if (hr.all_base[i]) printf(" %d:%llu/%llu", i * 8, hr.bad_base[i], hr.all_base[i]);
      printf("\n   by wave slot:");
      for (int i = 0; i < 16; ++i) if (hr.all_slot[i]) printf(" %d:%llu/%llu", i, hr.bad_slot[i], hr.all_slot[i]);
      printf("\n   by SIMD:");
      for (int i = 0; i < 4; ++i) printf(" %d:%llu/%llu", i, hr.bad_simd[i], hr.all_simd[i]);
      printf("\n");
    }
    fflush(stdout);
  }
  return 0;
}
