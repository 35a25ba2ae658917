// Rotated mask branch at inference: the pieces around the conv stack that are not convolutions.
//   * 2x pixel shuffle that turns the 1x1-conv form of ConvTranspose2d(k=2, s=2) into the 28x28 map,
//   * sigmoid (d2 mask_rcnn_inference),
//   * rotated mask paste = the reference's own paste_masks_in_image / _do_paste_mask
//     (glass/postprocess/post_processor_academic.py:187-335): per RoI a sampling grid centred on the box,
//     rotated by its angle, normalised to the box, bilinear grid_sample (align_corners=False, zero padding)
//     of the MxM mask over the whole image, threshold.
#include "common.h"
#include <cstdint>

namespace {

static inline int grid_for_n(long n, int block) {
  long g = (n + block - 1) / block;
  if (g > 256L * 32) g = 256L * 32;
  if (g < 1) g = 1;
  return (int)g;
}

// x [N,H,W,4*C] with channel = (a*2+b)*C + c  ->  y [N,2H,2W,C],  y[n,2h+a,2w+b,c] = x[n,h,w,(a*2+b)*C+c]
__global__ void pixel_shuffle2x_kernel(const float4* __restrict__ x, float4* __restrict__ y, int N, int H, int W, int C4) {
  const long total = (long)N * H * W * 4 * C4;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int c4 = (int)(idx % C4);
    long t = idx / C4;
    const int ab = (int)(t & 3);
    t >>= 2;
    const int w = (int)(t % W);
    t /= W;
    const int h = (int)(t % H);
    const int n = (int)(t / H);
    const int a = ab >> 1, b = ab & 1;
    y[(((long)n * 2 * H + 2 * h + a) * 2 * W + 2 * w + b) * C4 + c4] = x[idx];
  }
}

__global__ void sigmoid_kernel(float* __restrict__ x, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    x[i] = 1.f / (1.f + expf(-x[i]));
}

struct PasteParams {
  const float* masks;   // [R,M,M]
  const float* boxes;   // [R,5]
  uint8_t* out;         // [R,H,W]
  int R, M, H, W;
  float threshold;
};

// one thread per 4 consecutive output pixels of one RoI row (32-bit packed store).
// BIT-EXACT against the reference's torch-CPU path (tests/golden/mask_paste.npz, produced by the reference's own
// paste_masks_in_image): every fp32 operation below is the one torch executes, in its order, with fusion exactly where
// torch's kernels fuse (established by emulation, oracle/README notes + tests/test_mask_branch.py):
//   * `stack([gx, gy]) @ rot` is an sgemm with K = 2: acc = gx * r0; acc = fma(gy, r1, acc);
//   * the following `+= c`, `- x0`, `/ (x1 - x0)`, `* 2`, `- 1` are separate elementwise kernels (each rounded);
//   * grid_sample (CPU, vectorised): unnormalise = fma(n + 1, M / 2, -0.5); weights e = 1 - w, s = 1 - n,
//     nw = e*s, ne = w*s, sw = e*n, se = w*n; out = nw_v*nw, then fma(ne_v, ne, out), fma(sw_v, sw, out), fma(se_v, se, out)
//     with out-of-range taps contributing a value of 0 (masked gather);
//   * cos / sin: correctly rounded fp32 (evaluated in fp64 and rounded), which is what torch's SLEEF kernels return
//     for all but rare inputs.
// Compiler contraction is off for this kernel; the fused operations are explicit __builtin_fmaf calls.
#pragma clang fp contract(off)
__global__ __launch_bounds__(256) void paste_rotated_masks_kernel(PasteParams p) {
  const int r = blockIdx.z;
  const int py = blockIdx.y;
  const float* b = p.boxes + (long)r * 5;
  const float cx = b[0], cy = b[1], bw = b[2], bh = b[3];
  // torch.deg2rad multiplies by pi/180 in fp32
  const float ang = b[4] * 0.017453292519943295f;
  const float cs = (float)cos((double)ang), sn = (float)sin((double)ang);
  const float nsn = -sn;
  // x0 = cx + (h * sin_t - w * cos_t) / 2 with sin_t = 0, cos_t = 1 (python ints in the reference)
  const float x0 = cx + (bh * 0.f - bw * 1.f) / 2.f, x1 = cx - (bh * 0.f - bw * 1.f) / 2.f;
  const float y0 = cy - (bh * 1.f + bw * 0.f) / 2.f, y1 = cy + (bh * 1.f + bw * 0.f) / 2.f;
  const float dx = x1 - x0, dy = y1 - y0;
  const float halfM = (float)p.M / 2.f;
  const float* mk = p.masks + (long)r * p.M * p.M;
  uint8_t* orow = p.out + ((long)r * p.H + py) * p.W;
  const float gy = ((float)py + 0.5f) - cy;
  for (int q = blockIdx.x * blockDim.x + threadIdx.x; q * 4 < p.W; q += gridDim.x * blockDim.x) {
    uint32_t packed = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int px = q * 4 + k;
      uint32_t bit = 0;
      if (px < p.W) {
        const float gx = ((float)px + 0.5f) - cx;
        // [gx, gy] @ [[cos, sin], [-sin, cos]] (K = 2 sgemm), recentred, normalised to the box
        float ix = __builtin_fmaf(gy, nsn, gx * cs);
        float iy = __builtin_fmaf(gy, cs, gx * sn);
        ix = ix + cx;
        iy = iy + cy;
        float nx = (ix - x0) / dx;
        nx = nx * 2.f;
        nx = nx - 1.f;
        float ny = (iy - y0) / dy;
        ny = ny * 2.f;
        ny = ny - 1.f;
        // grid_sample, align_corners=False, bilinear, zero padding
        const float fx = __builtin_fmaf(nx + 1.f, halfM, -0.5f);
        const float fy = __builtin_fmaf(ny + 1.f, halfM, -0.5f);
        const float flx = floorf(fx), fly = floorf(fy);
        float v = 0.f;
        if (flx >= -1.f && flx < (float)p.M && fly >= -1.f && fly < (float)p.M) {   // (also false for NaN: v stays 0)
          const int xw = (int)flx, yn = (int)fly;
          const float tw = fx - flx, te = 1.f - tw, tn = fy - fly, ts = 1.f - tn;
          const float wnw = te * ts, wne = tw * ts, wsw = te * tn, wse = tw * tn;
          const bool xin0 = xw >= 0, xin1 = xw + 1 < p.M;
          const bool yin0 = yn >= 0, yin1 = yn + 1 < p.M;
          const float vnw = (yin0 && xin0) ? mk[yn * p.M + xw] : 0.f;
          const float vne = (yin0 && xin1) ? mk[yn * p.M + xw + 1] : 0.f;
          const float vsw = (yin1 && xin0) ? mk[(yn + 1) * p.M + xw] : 0.f;
          const float vse = (yin1 && xin1) ? mk[(yn + 1) * p.M + xw + 1] : 0.f;
          v = vnw * wnw;
          v = __builtin_fmaf(vne, wne, v);
          v = __builtin_fmaf(vsw, wsw, v);
          v = __builtin_fmaf(vse, wse, v);
        }
        // threshold >= 0: boolean mask; threshold < 0: the reference's visualisation mode, (v * 255).to(uint8)
        const float v255 = v * 255.f;
        bit = p.threshold >= 0.f ? ((v >= p.threshold) ? 1u : 0u) : ((uint32_t)(int)v255 & 0xffu);
      }
      packed |= bit << (8 * k);
    }
    if (q * 4 + 3 < p.W && (((uintptr_t)(orow + q * 4)) & 3) == 0) {
      *reinterpret_cast<uint32_t*>(orow + q * 4) = packed;
    } else {
      for (int k = 0; k < 4; ++k)
        if (q * 4 + k < p.W) orow[q * 4 + k] = (uint8_t)((packed >> (8 * k)) & 0xff);
    }
  }
}

}  // namespace

extern "C" int glass_pixel_shuffle2x_nhwc(const float* x, float* y, int N, int H, int W, int C, glass_stream_t stream) {
  GLASS_CHECK_ARG(x && y, "glass_pixel_shuffle2x_nhwc: null pointer");
  GLASS_CHECK_ARG(N >= 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "glass_pixel_shuffle2x_nhwc: C=%d must be a positive multiple of 4", C);
  if (N == 0) return GLASS_OK;
  const long total = (long)N * H * W * C;      // float4 count of the input (= 4 * C/4 per pixel)
  hipLaunchKernelGGL(pixel_shuffle2x_kernel, dim3(grid_for_n(total, 256)), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const float4*>(x), reinterpret_cast<float4*>(y), N, H, W, C / 4);
  GLASS_CHECK_LAUNCH("glass_pixel_shuffle2x_nhwc");
  return GLASS_OK;
}

extern "C" int glass_sigmoid_inplace(float* x, int64_t n, glass_stream_t stream) {
  GLASS_CHECK_ARG(x != nullptr || n == 0, "glass_sigmoid_inplace: null pointer");
  if (n <= 0) return GLASS_OK;
  hipLaunchKernelGGL(sigmoid_kernel, dim3(grid_for_n(n, 256)), dim3(256), 0, (hipStream_t)stream, x, (long)n);
  GLASS_CHECK_LAUNCH("glass_sigmoid_inplace");
  return GLASS_OK;
}

extern "C" int glass_paste_rotated_masks(const float* masks, const float* boxes, int R, int M, int H, int W, float threshold,
                                         uint8_t* out, glass_stream_t stream) {
  GLASS_CHECK_ARG(R == 0 || (masks && boxes && out), "glass_paste_rotated_masks: null pointer");
  GLASS_CHECK_ARG(R >= 0 && M > 0 && H > 0 && W > 0 && H <= 65535 && R <= 65535, "glass_paste_rotated_masks: bad dims");
  if (R == 0) return GLASS_OK;
  PasteParams p{masks, boxes, out, R, M, H, W, threshold};
  const int quads = (W + 3) / 4;
  hipLaunchKernelGGL(paste_rotated_masks_kernel, dim3((unsigned)((quads + 255) / 256), (unsigned)H, (unsigned)R), dim3(256), 0,
                     (hipStream_t)stream, p);
  GLASS_CHECK_LAUNCH("glass_paste_rotated_masks");
  return GLASS_OK;
}
