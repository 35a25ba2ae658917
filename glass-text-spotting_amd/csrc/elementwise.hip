// HBM-bound NHWC helpers: max pooling, image preprocess, u8->f32 resize, mean over H.
// All are one-pass streaming kernels with 16-byte (float4) channel-vector accesses where the
// channel count allows; grids are capped and grid-strided.
#include "common.h"

static inline int grid_for(long n, int block) {
  long g = (n + block - 1) / block;
  if (g > 256L * 16) g = 256L * 16;
  if (g < 1) g = 1;
  return (int)g;
}

// ---------------------------------------------------------------- max pool (C % 4 == 0)
// One thread per (output pixel, 4 channels); blockIdx.y = output row (n, ho) so the only integer division left
// is the 32-bit (wo, c4) split, and the window loops are compile-time for the shapes the path uses (3x3 / 2x2 /
// 2x1).  The first version (flat 64-bit index, three 64-bit divisions per output) was VALU-bound at 3x the
// kernel's HBM time.
typedef _Float16 mp_h4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 mp_widen(float4 v) { return v; }
__device__ __forceinline__ float4 mp_widen(mp_h4 v) { return make_float4((float)v.x, (float)v.y, (float)v.z, (float)v.w); }
__device__ __forceinline__ void mp_store(float4* p, float4 v) { *p = v; }
__device__ __forceinline__ void mp_store(mp_h4* p, float4 v) { *p = mp_h4{(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w}; }   // exact: max of fp16 values

// V4 = float4 (fp32 tensors) or mp_h4 (fp16 storage, glass_maxpool2d_nhwc_h16): 4 channels per thread either way
template <int KH, int KW, typename V4>
__global__ __launch_bounds__(256) void maxpool_nhwc_kernel(const V4* __restrict__ x, V4* __restrict__ y, int H, int W,
                                                           int C4, int kh_rt, int kw_rt, int sh, int sw, int ph, int pw,
                                                           int Ho, int Wo, int rows) {
  const int kh_n = KH > 0 ? KH : kh_rt, kw_n = KW > 0 ? KW : kw_rt;
  const int per_row = Wo * C4;
  for (int row = blockIdx.y; row < rows; row += gridDim.y) {      // row = n * Ho + ho
  const int n = row / Ho, ho = row - n * Ho;
  const int h0 = ho * sh - ph;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < per_row; e += gridDim.x * blockDim.x) {
    const int wo = e / C4, c4 = e - wo * C4;
    const int w0 = wo * sw - pw;
    float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
#pragma unroll
    for (int kh = 0; kh < kh_n; ++kh) {
      const int hi = h0 + kh;
      if ((unsigned)hi >= (unsigned)H) continue;
      const V4* xr = x + ((long)(n * H + hi) * W) * C4 + c4;
#pragma unroll
      for (int kw = 0; kw < kw_n; ++kw) {
        const int wi = w0 + kw;
        if ((unsigned)wi >= (unsigned)W) continue;
        const float4 v = mp_widen(xr[(long)wi * C4]);
        m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
      }
    }
    mp_store(&y[(long)row * per_row + e], m);
  }
  }
}

template <typename V4>
static int maxpool_launch(const void* x, void* y, int N, int H, int W, int C, int KH, int KW, int sh, int sw, int ph, int pw,
                          int Ho, int Wo, glass_stream_t stream, const char* what) {
  GLASS_CHECK_ARG(x && y, "%s: null pointer", what);
  GLASS_CHECK_ARG(C % 4 == 0 && C > 0, "%s: C=%d must be a multiple of 4", what, C);
  GLASS_CHECK_ARG(Ho == (H + 2 * ph - KH) / sh + 1 && Wo == (W + 2 * pw - KW) / sw + 1, "%s: bad Ho/Wo", what);
  if (N == 0) return GLASS_OK;
  const long rows = (long)N * Ho, per_row = (long)Wo * (C / 4);
  GLASS_CHECK_ARG(rows <= 0x7fffffffL && per_row <= 0x7fffffffL, "%s: tensor too large", what);
  const dim3 grid((unsigned)((per_row + 255) / 256 < 64 ? (per_row + 255) / 256 : 64), (unsigned)(rows < 65535 ? rows : 65535));
  const V4* x4 = reinterpret_cast<const V4*>(x);
  V4* y4 = reinterpret_cast<V4*>(y);
  hipStream_t s = (hipStream_t)stream;
  if (KH == 3 && KW == 3)
    hipLaunchKernelGGL((maxpool_nhwc_kernel<3, 3, V4>), grid, dim3(256), 0, s, x4, y4, H, W, C / 4, KH, KW, sh, sw, ph, pw, Ho, Wo, (int)rows);
  else if (KH == 2 && KW == 2)
    hipLaunchKernelGGL((maxpool_nhwc_kernel<2, 2, V4>), grid, dim3(256), 0, s, x4, y4, H, W, C / 4, KH, KW, sh, sw, ph, pw, Ho, Wo, (int)rows);
  else if (KH == 2 && KW == 1)
    hipLaunchKernelGGL((maxpool_nhwc_kernel<2, 1, V4>), grid, dim3(256), 0, s, x4, y4, H, W, C / 4, KH, KW, sh, sw, ph, pw, Ho, Wo, (int)rows);
  else
    hipLaunchKernelGGL((maxpool_nhwc_kernel<0, 0, V4>), grid, dim3(256), 0, s, x4, y4, H, W, C / 4, KH, KW, sh, sw, ph, pw, Ho, Wo, (int)rows);
  GLASS_CHECK_LAUNCH(what);
  return GLASS_OK;
}

extern "C" int glass_maxpool2d_nhwc(const float* x, float* y, int N, int H, int W, int C, int KH, int KW, int sh, int sw,
                                    int ph, int pw, int Ho, int Wo, glass_stream_t stream) {
  return maxpool_launch<float4>(x, y, N, H, W, C, KH, KW, sh, sw, ph, pw, Ho, Wo, stream, "glass_maxpool2d_nhwc");
}

extern "C" int glass_maxpool2d_nhwc_h16(const void* x, void* y, int N, int H, int W, int C, int KH, int KW, int sh, int sw,
                                        int ph, int pw, int Ho, int Wo, glass_stream_t stream) {
  return maxpool_launch<mp_h4>(x, y, N, H, W, C, KH, KW, sh, sw, ph, pw, Ho, Wo, stream, "glass_maxpool2d_nhwc_h16");
}

// ---------------------------------------------------------------- preprocess
__global__ void preprocess_kernel(const float* __restrict__ chw, int H, int W, float m0, float m1, float m2, float s0,
                                  float s1, float s2, float4* __restrict__ out, int Hp, int Wp) {
  const long total = (long)Hp * Wp;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int w = (int)(idx % Wp), h = (int)(idx / Wp);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (h < H && w < W) {
      const long o = (long)h * W + w, plane = (long)H * W;
      v.x = (chw[o] - m0) / s0;
      v.y = (chw[plane + o] - m1) / s1;
      v.z = (chw[2 * plane + o] - m2) / s2;
    }
    out[idx] = v;
  }
}

extern "C" int glass_preprocess_image(const float* chw, int H, int W, const float* mean3, const float* std3,
                                      float* batch_nhwc4, int n, int Hp, int Wp, glass_stream_t stream) {
  GLASS_CHECK_ARG(chw && mean3 && std3 && batch_nhwc4, "glass_preprocess_image: null pointer");
  GLASS_CHECK_ARG(H > 0 && W > 0 && H <= Hp && W <= Wp && n >= 0, "glass_preprocess_image: bad dims");
  float4* out = reinterpret_cast<float4*>(batch_nhwc4) + (long)n * Hp * Wp;
  hipLaunchKernelGGL(preprocess_kernel, dim3(grid_for((long)Hp * Wp, 256)), dim3(256), 0, (hipStream_t)stream, chw, H, W,
                     mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2], out, Hp, Wp);
  GLASS_CHECK_LAUNCH("glass_preprocess_image");
  return GLASS_OK;
}

// ---------------------------------------------------------------- u8 HWC -> f32 CHW (+ bilinear)
// F.interpolate(mode='bilinear', align_corners=False) source index rule:
// src = max(0, (dst + 0.5) * (in/out) - 0.5); i1 = min(i0 + 1, in - 1).
__global__ void u8_to_chw_resize_kernel(const uint8_t* __restrict__ hwc, int H, int W, float* __restrict__ chw, int Ho,
                                        int Wo, int flip) {
  const long total = (long)Ho * Wo;
  const float sy = (float)H / (float)Ho, sx = (float)W / (float)Wo;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int wo = (int)(idx % Wo), ho = (int)(idx / Wo);
    float v[3];
    if (Ho == H && Wo == W) {
      const uint8_t* p = hwc + ((long)ho * W + wo) * 3;
      v[0] = p[0]; v[1] = p[1]; v[2] = p[2];
    } else {
      float fy = ((float)ho + 0.5f) * sy - 0.5f;
      float fx = ((float)wo + 0.5f) * sx - 0.5f;
      fy = fy < 0.f ? 0.f : fy;
      fx = fx < 0.f ? 0.f : fx;
      const int y0 = (int)fy, x0 = (int)fx;
      const int y1 = y0 + (y0 < H - 1 ? 1 : 0), x1 = x0 + (x0 < W - 1 ? 1 : 0);
      const float ly = fy - (float)y0, lx = fx - (float)x0, hy = 1.f - ly, hx = 1.f - lx;
      const uint8_t* p00 = hwc + ((long)y0 * W + x0) * 3;
      const uint8_t* p01 = hwc + ((long)y0 * W + x1) * 3;
      const uint8_t* p10 = hwc + ((long)y1 * W + x0) * 3;
      const uint8_t* p11 = hwc + ((long)y1 * W + x1) * 3;
#pragma unroll
      for (int c = 0; c < 3; ++c)
        v[c] = hy * (hx * (float)p00[c] + lx * (float)p01[c]) + ly * (hx * (float)p10[c] + lx * (float)p11[c]);
    }
    const long plane = (long)Ho * Wo;
    chw[idx] = v[flip ? 2 : 0];
    chw[plane + idx] = v[1];
    chw[2 * plane + idx] = v[flip ? 0 : 2];
  }
}

extern "C" int glass_image_u8hwc_to_chw_resized(const uint8_t* hwc, int H, int W, float* chw, int Ho, int Wo,
                                                int flip_channels, glass_stream_t stream) {
  GLASS_CHECK_ARG(hwc && chw && H > 0 && W > 0 && Ho > 0 && Wo > 0, "glass_image_u8hwc_to_chw_resized: bad args");
  hipLaunchKernelGGL(u8_to_chw_resize_kernel, dim3(grid_for((long)Ho * Wo, 256)), dim3(256), 0, (hipStream_t)stream, hwc, H,
                     W, chw, Ho, Wo, flip_channels);
  GLASS_CHECK_LAUNCH("glass_image_u8hwc_to_chw_resized");
  return GLASS_OK;
}

// ---------------------------------------------------------------- mean over H
__global__ void mean_h_kernel(const float4* __restrict__ x, float4* __restrict__ y, int R, int H, int WC4) {
  const long total = (long)R * WC4;
  const float fh = (float)H;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int j = (int)(idx % WC4);
    const long r = idx / WC4;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int h = 0; h < H; ++h) {
      const float4 v = x[(r * H + h) * WC4 + j];
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    y[idx] = make_float4(s.x / fh, s.y / fh, s.z / fh, s.w / fh);
  }
}

extern "C" int glass_mean_over_h(const float* x, float* y, int R, int H, int W, int C, glass_stream_t stream) {
  GLASS_CHECK_ARG(x && y && C % 4 == 0 && H > 0, "glass_mean_over_h: bad args");
  if (R == 0) return GLASS_OK;
  const int WC4 = W * C / 4;
  hipLaunchKernelGGL(mean_h_kernel, dim3(grid_for((long)R * WC4, 256)), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const float4*>(x), reinterpret_cast<float4*>(y), R, H, WC4);
  GLASS_CHECK_LAUNCH("glass_mean_over_h");
  return GLASS_OK;
}

// ---------------------------------------------------------------- a *= b (SimpleAttention gate, fusion_modules.py:183)
__global__ void mul_inplace_kernel(float4* __restrict__ a, const float4* __restrict__ b, long n4) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    float4 x = a[i];
    const float4 y = b[i];
    x.x *= y.x; x.y *= y.y; x.z *= y.z; x.w *= y.w;
    a[i] = x;
  }
}

extern "C" int glass_mul_inplace(float* a, const float* b, int64_t n, glass_stream_t stream) {
  GLASS_CHECK_ARG((a && b) || n == 0, "glass_mul_inplace: null pointer");
  GLASS_CHECK_ARG(n >= 0 && n % 4 == 0 && (((uintptr_t)a | (uintptr_t)b) & 15) == 0, "glass_mul_inplace: n %% 4 == 0 and 16-byte alignment required");
  if (n == 0) return GLASS_OK;
  hipLaunchKernelGGL(mul_inplace_kernel, dim3(grid_for(n / 4, 256)), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<float4*>(a), reinterpret_cast<const float4*>(b), (long)(n / 4));
  GLASS_CHECK_LAUNCH("glass_mul_inplace");
  return GLASS_OK;
}

// ---------------------------------------------------------------- fp32 -> fp16 (round to nearest even): the operand rounding
// of the fp16 conv modes done once, so that an fp32 activation tensor can feed glass_conv2d_nhwc_h16_packed
typedef _Float16 cast_h4 __attribute__((ext_vector_type(4)));
__global__ void cast_f32_to_f16_kernel(const float4* __restrict__ x, cast_h4* __restrict__ y, long n4) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const float4 v = x[i];
    y[i] = cast_h4{(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w};
  }
}

extern "C" int glass_cast_f32_to_f16(const float* x, void* y, int64_t n, glass_stream_t stream) {
  GLASS_CHECK_ARG((x && y) || n == 0, "glass_cast_f32_to_f16: null pointer");
  GLASS_CHECK_ARG(n >= 0 && n % 4 == 0 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)y & 7) == 0,
                  "glass_cast_f32_to_f16: n %% 4 == 0, x 16-byte and y 8-byte aligned required");
  if (n == 0) return GLASS_OK;
  hipLaunchKernelGGL(cast_f32_to_f16_kernel, dim3(grid_for(n / 4, 256)), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const float4*>(x), static_cast<cast_h4*>(y), (long)(n / 4));
  GLASS_CHECK_LAUNCH("glass_cast_f32_to_f16");
  return GLASS_OK;
}
