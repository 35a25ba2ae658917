// Rotated RoIAlign over up to 5 NHWC pyramid levels (detectron2 ROIPooler + ROIAlignRotated
// semantics, see include/glass_hip.h).  Gather-bound: one thread per (roi, bin, 4-channel
// vector); the channel vector is the fastest index so the 4 bilinear taps of a sample are
// 16-byte loads that are contiguous across neighbouring lanes (coalesced 1 KiB per wave for
// C >= 256) and the output store is a coalesced float4 (or strided scalar stores when the
// output is channel-interleaved).
#include "common.h"

struct RoiParams {
  const float* feat[5];
  int H[5], W[5], ld[5];
  float scale[5];
  int num_levels, min_level;
  int C4, PH, PW, sampling_ratio;
  int ldy, ycoff, ycs;
  int R;
  const float* boxes;
  const int* batch_idx;
  float* out;
};

typedef _Float16 ra_h4 __attribute__((ext_vector_type(4)));
// FH: the pyramid levels are fp16 tensors (fp16 storage mode, glass_roi_align_rotated_h16); taps are widened to fp32,
// the interpolation and the output stay fp32
template <bool FH>
__device__ __forceinline__ float4 ra_tap(const float* base, long off) {
  if constexpr (FH) {
    const ra_h4 v = *reinterpret_cast<const ra_h4*>(reinterpret_cast<const _Float16*>(base) + off);
    return make_float4((float)v.x, (float)v.y, (float)v.z, (float)v.w);
  } else {
    return *reinterpret_cast<const float4*>(base + off);
  }
}

// UP2: every level tensor is HALF resolution ([N,H/2,W/2,*]) and is read through nearest-neighbour x2 upsampling - p.H / p.W are
// the UPSAMPLED dimensions, so the sampling grid, the clamps and the bilinear weights are those of the materialised upsampled
// map and only the tap address changes ((y, x) -> (y >> 1, x >> 1)): glass_roi_align_rotated_up2.
template <bool FH, bool UP2>
__global__ __launch_bounds__(256) void roi_align_rotated_kernel(RoiParams p) {
  const long total = (long)p.R * p.PH * p.PW * p.C4;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int c4 = (int)(idx % p.C4);
    long t = idx / p.C4;
    const int pw = (int)(t % p.PW);
    t /= p.PW;
    const int ph = (int)(t % p.PH);
    const int r = (int)(t / p.PH);

    const float* box = p.boxes + 5 * (long)r;
    const float bw = box[2], bh = box[3];
    int lvl = 0;
    if (p.num_levels > 1) {
      // d2 assign_boxes_to_levels: floor(4 + log2(sqrt(area)/224 + 1e-8)) clamped to the pooler's levels
      float l = floorf(4.f + log2f(sqrtf(bw * bh) / 224.f + 1e-8f));
      const float lo = (float)p.min_level, hi = (float)(p.min_level + p.num_levels - 1);
      l = fminf(fmaxf(l, lo), hi);
      lvl = (int)l - p.min_level;
    }
    // select per-level parameters without dynamic indexing into kernel-argument arrays
    const float* feat = p.feat[0];
    int H = p.H[0], W = p.W[0], ld = p.ld[0];
    float scale = p.scale[0];
#pragma unroll
    for (int k = 1; k < 5; ++k)
      if (lvl == k) { feat = p.feat[k]; H = p.H[k]; W = p.W[k]; ld = p.ld[k]; scale = p.scale[k]; }

    const int b = p.batch_idx[r];
    const float cw = box[0] * scale - 0.5f;
    const float ch = box[1] * scale - 0.5f;
    const float rw = bw * scale, rh = bh * scale;
    const float theta = (float)((double)box[4] * 3.14159265358979323846 / 180.0);
    float sin_t, cos_t;
    sincosf(theta, &sin_t, &cos_t);
    const float bin_h = rh / (float)p.PH, bin_w = rw / (float)p.PW;
    const int gh = p.sampling_ratio > 0 ? p.sampling_ratio : (int)ceilf(rh / (float)p.PH);
    const int gw = p.sampling_ratio > 0 ? p.sampling_ratio : (int)ceilf(rw / (float)p.PW);
    const float count = (float)(gh * gw > 1 ? gh * gw : 1);
    const float start_h = -rh / 2.0f, start_w = -rw / 2.0f;
    const int Ws = UP2 ? (W >> 1) : W;                        // stored row length
    const long boff = (UP2 ? (long)b * (H >> 1) * Ws : (long)b * H * W) * ld + c4 * 4;   // (image b, channel group c4) in the level

    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int iy = 0; iy < gh; ++iy) {
      const float yy = start_h + ph * bin_h + ((float)iy + .5f) * bin_h / (float)gh;
      for (int ix = 0; ix < gw; ++ix) {
        const float xx = start_w + pw * bin_w + ((float)ix + .5f) * bin_w / (float)gw;
        float y = yy * cos_t - xx * sin_t + ch;
        float x = yy * sin_t + xx * cos_t + cw;
        if (y < -1.0f || y > (float)H || x < -1.0f || x > (float)W) continue;
        if (y < 0.f) y = 0.f;
        if (x < 0.f) x = 0.f;
        int yl = (int)y, xl = (int)x, yh, xh;
        if (yl >= H - 1) { yh = yl = H - 1; y = (float)yl; } else yh = yl + 1;
        if (xl >= W - 1) { xh = xl = W - 1; x = (float)xl; } else xh = xl + 1;
        const float ly = y - (float)yl, lx = x - (float)xl, hy = 1.f - ly, hx = 1.f - lx;
        const float w1 = hy * hx, w2 = hy * lx, w3 = ly * hx, w4 = ly * lx;
        const int ya = UP2 ? (yl >> 1) : yl, yb = UP2 ? (yh >> 1) : yh, xa = UP2 ? (xl >> 1) : xl, xb = UP2 ? (xh >> 1) : xh;
        const float4 v1 = ra_tap<FH>(feat, boff + ((long)ya * Ws + xa) * ld);
        const float4 v2 = ra_tap<FH>(feat, boff + ((long)ya * Ws + xb) * ld);
        const float4 v3 = ra_tap<FH>(feat, boff + ((long)yb * Ws + xa) * ld);
        const float4 v4 = ra_tap<FH>(feat, boff + ((long)yb * Ws + xb) * ld);
        // same association order as the reference CPU op: w1*v1 + w2*v2 + w3*v3 + w4*v4
        acc.x += w1 * v1.x + w2 * v2.x + w3 * v3.x + w4 * v4.x;
        acc.y += w1 * v1.y + w2 * v2.y + w3 * v3.y + w4 * v4.y;
        acc.z += w1 * v1.z + w2 * v2.z + w3 * v3.z + w4 * v4.z;
        acc.w += w1 * v1.w + w2 * v2.w + w3 * v3.w + w4 * v4.w;
      }
    }
    acc.x /= count; acc.y /= count; acc.z /= count; acc.w /= count;
    float* o = p.out + ((long)(r * p.PH + ph) * p.PW + pw) * p.ldy + p.ycoff + (long)c4 * 4 * p.ycs;
    if (p.ycs == 1 && ((p.ldy | p.ycoff) & 3) == 0) {
      *reinterpret_cast<float4*>(o) = acc;
    } else {
      o[0] = acc.x; o[p.ycs] = acc.y; o[2 * p.ycs] = acc.z; o[3 * p.ycs] = acc.w;
    }
  }
}

static int roi_align_launch(const glass_roialign_desc* d, const float* boxes, const int* batch_idx, int R, float* out,
                            glass_stream_t stream, bool feat_half, bool up2 = false) {
  GLASS_CHECK_ARG(d && out, "glass_roi_align_rotated: null pointer");
  GLASS_CHECK_ARG(d->num_levels >= 1 && d->num_levels <= 5, "glass_roi_align_rotated: num_levels=%d", d->num_levels);
  GLASS_CHECK_ARG(d->C > 0 && d->C % 4 == 0, "glass_roi_align_rotated: C=%d must be a multiple of 4", d->C);
  GLASS_CHECK_ARG(d->PH > 0 && d->PW > 0 && d->sampling_ratio >= 0 && d->y_cstride >= 1, "glass_roi_align_rotated: bad output spec");
  if (R == 0) return GLASS_OK;
  GLASS_CHECK_ARG(boxes && batch_idx, "glass_roi_align_rotated: null boxes");
  RoiParams p;
  for (int i = 0; i < 5; ++i) {
    const int k = i < d->num_levels ? i : 0;
    GLASS_CHECK_ARG(d->feat[k] != nullptr && d->ld[k] % 4 == 0 && d->ld[k] >= d->C, "glass_roi_align_rotated: level %d", k);
    p.feat[i] = d->feat[k]; p.H[i] = d->H[k]; p.W[i] = d->W[k]; p.ld[i] = d->ld[k]; p.scale[i] = d->scale[k];
  }
  p.num_levels = d->num_levels; p.min_level = d->min_level;
  p.C4 = d->C / 4; p.PH = d->PH; p.PW = d->PW; p.sampling_ratio = d->sampling_ratio;
  p.ldy = d->ldy; p.ycoff = d->y_coff; p.ycs = d->y_cstride;
  p.R = R; p.boxes = boxes; p.batch_idx = batch_idx; p.out = out;
  const long total = (long)R * p.PH * p.PW * p.C4;
  long g = (total + 255) / 256;
  if (g > 256L * 32) g = 256L * 32;
  if (up2) {
    for (int i = 0; i < d->num_levels; ++i)
      GLASS_CHECK_ARG(d->H[i] % 2 == 0 && d->W[i] % 2 == 0, "glass_roi_align_rotated_up2: level %d: upsampled H, W must be even", i);
    GLASS_CHECK_ARG(!feat_half, "glass_roi_align_rotated_up2: fp32 levels only");
    hipLaunchKernelGGL((roi_align_rotated_kernel<false, true>), dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, p);
  } else if (feat_half)
    hipLaunchKernelGGL((roi_align_rotated_kernel<true, false>), dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, p);
  else
    hipLaunchKernelGGL((roi_align_rotated_kernel<false, false>), dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, p);
  GLASS_CHECK_LAUNCH("glass_roi_align_rotated");
  return GLASS_OK;
}

extern "C" int glass_roi_align_rotated(const glass_roialign_desc* d, const float* boxes, const int* batch_idx, int R,
                                       float* out, glass_stream_t stream) {
  return roi_align_launch(d, boxes, batch_idx, R, out, stream, false);
}

extern "C" int glass_roi_align_rotated_h16(const glass_roialign_desc* d, const float* boxes, const int* batch_idx, int R,
                                           float* out, glass_stream_t stream) {
  return roi_align_launch(d, boxes, batch_idx, R, out, stream, true);
}

// ROIAlignRotated of nearest_up2(level): d->feat[] are the HALF-resolution tensors, d->H / d->W / d->scale describe the upsampled
// map.  RoIAlign is linear in the feature map and P2P3Fusion (reference fusion_modules.py:281-286: conv1x1(p2) +
// up2(conv1x1(p3)), no bias, no norm) is linear too, so pool(conv1(p2) + up2(conv2(p3))) = W1 pool(p2) + W2 pool(up2(p3)): the
// recognizer pooler can run on p2 and on up2(p3) directly and the two 1x1 convolutions shrink from the whole map to the pooled
// bins (glass_amd/modeling/fusion/fusion_modules.py P2P3Fusion.pooled_nhwc).
extern "C" int glass_roi_align_rotated_up2(const glass_roialign_desc* d, const float* boxes, const int* batch_idx, int R,
                                           float* out, glass_stream_t stream) {
  return roi_align_launch(d, boxes, batch_idx, R, out, stream, false, true);
}
