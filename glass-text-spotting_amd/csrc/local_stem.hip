// Fused head of the local-crop feature extractor (reference glass/modeling/fusion/local_feature_extraction.py:103-112:
// conv0_1 3x3 3->16 + BN + ReLU, conv0_2 3x3 16->32 + BN + ReLU, maxpool1 2x2): [R,H,W,4] NHWC4 crops -> [R,H/2,W/2,32].
//
// Run as three launches these layers are HBM traffic and little else (per RoI 128 x 128: 256 KB in, 1 MB + 1 MB of
// 16-channel map written and re-read, 2 MB + 2 MB of 32-channel map, 0.5 MB out: 0.82 ms per step for 256 RoIs at
// 1.5 TB/s); fused, the two intermediate maps never leave the CU: 0.25 + 0.5 MB per RoI.
//
// One workgroup (256 threads, 4 wavefronts) per 32 x 32 tile of conv0_2 outputs (= 16 x 16 pooled pixels):
//   1. the 36 x 36 input patch (zero outside the image) -> LDS as float4 (3 channels + 0);
//   2. conv0_1 on the 34 x 34 pixels conv0_2 needs, on the vector ALU (576 fma per pixel; the 16 x 3 x 3 x 4 weights
//      are LDS broadcasts shared by the 5 pixels a thread computes together) -> LDS [34*34][16 + 4 pad]; pixels outside
//      the image are stored as 0 (conv0_2 pads its INPUT with zeros, it does not see conv0_1 evaluated out there);
//   3. conv0_2 as an implicit GEMM on v_mfma_f32_32x32x2_f32: M = 1024 pixels (wavefront w: image rows 8w .. 8w+7, one
//      32-pixel row per accumulator block), N = 32 channels, K = 9 taps x 16 channels; A fragments are ds_read_b128 of
//      the conv0_1 tile at the tap's offset (80-byte pixel rows: conflict-free), the 32 x 144 weights sit in 72
//      registers per lane for the whole block;
//   4. bias + ReLU + 2x2 max in registers (both pooling partners of a pixel live in the same lane of the C layout),
//      128-byte-per-pixel stores.
// H16 (glass_local_stem_fused_h16, conv precision "fp16s"): the arithmetic of the three fp16-storage launches it replaces -
// crops and weights rounded to fp16 (round to nearest even) as operands, fp32 accumulation, the conv0_1 map rounded to fp16
// where the unfused path stores it (it sits in LDS as fp16), the pooled output written as fp16.  conv0_2 then runs on
// v_mfma_f32_32x32x16_f16 - one MFMA per tap and 32-pixel row, K = the 16 channels - instead of 8 fp32 MFMAs.
#include "common.h"
#include <cstdint>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8s __attribute__((ext_vector_type(8)));
typedef _Float16 h4s __attribute__((ext_vector_type(4)));

namespace {

constexpr int TS = 32;                 // conv0_2 output tile (TS x TS pixels)
constexpr int C1W = TS + 2;            // conv0_1 tile side
constexpr int XW = TS + 4;             // input patch side
constexpr int C1LD = 20;               // floats per conv0_1 pixel in LDS (16 + 4 pad: 80-byte rows, conflict-free ds_read_b128)
constexpr int C1LDH = 24;              // H16: halves per conv0_1 pixel (16 + 8 pad: 48-byte rows, conflict-free ds_read_b128)
constexpr int STEM_LDS_BYTES = (XW * XW * 4 + C1W * C1W * C1LD + 16 * 9 * 4) * 4;

struct StemParams {
  const float* x;      // [R,H,W,4]
  const float* w1;     // [16][3][3][4]  (BN folded)
  const float* b1;     // [16]
  const float* w2;     // [32][3][3][16]
  const float* b2;     // [32]
  void* y;             // [R,H/2,W/2,32] fp32, or fp16 (H16)
  int R, H, W, tiles_h, tiles_w;
};

__device__ __forceinline__ float qh(float v) { return (float)(_Float16)v; }
__device__ __forceinline__ float4 qh4(float4 v) { return make_float4(qh(v.x), qh(v.y), qh(v.z), qh(v.w)); }

template <bool H16>
__global__ __launch_bounds__(256, 1) void local_stem_fused_kernel(StemParams p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float4* xin = reinterpret_cast<float4*>(smem);                     // [XW*XW]
  float* c1 = smem + XW * XW * 4;                                    // [C1W*C1W][C1LD]
  float4* w1s = reinterpret_cast<float4*>(c1 + C1W * C1W * C1LD);    // [16*9] float4 (cin4)

  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  int b = blockIdx.x;
  const int tw = b % p.tiles_w; b /= p.tiles_w;
  const int th = b % p.tiles_h;
  const int r = b / p.tiles_h;
  const int h0 = th * TS, w0 = tw * TS;                              // conv0_2 tile origin in the image

  // ---- 1. input patch (rows h0-2 .. h0+TS+1) and conv0_1 weights -> LDS
  const float4* xg = reinterpret_cast<const float4*>(p.x) + (long)r * p.H * p.W;
  for (int i = tid; i < XW * XW; i += 256) {
    const int py = i / XW, px = i - py * XW;
    const int hi = h0 - 2 + py, wi = w0 - 2 + px;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if ((unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W) v = xg[(long)hi * p.W + wi];
    xin[i] = H16 ? qh4(v) : v;
  }
  if (tid < 16 * 9) {
    const float4 v = reinterpret_cast<const float4*>(p.w1)[tid];
    w1s[tid] = H16 ? qh4(v) : v;
  }
  // conv0_2 weights: B fragment of the 32x32x2 MFMA = W2[n = lane&31][k], k-step s of group (tap, g) uses
  // ci = 8 g + 4 (lane>>5) + s: one float4 per (tap, g) and lane, 18 of them
  float4 w2r[H16 ? 1 : 18];
  h8s w2h[H16 ? 9 : 1];                // H16: B fragment of the 32x32x16 MFMA = W2[n = lane&31][tap][k = 8 (lane>>5) .. +7]
  if constexpr (H16) {
    const float4* w2g = reinterpret_cast<const float4*>(p.w2) + (long)(lane & 31) * 36 + 2 * (lane >> 5);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const float4 a = w2g[tap * 4], c = w2g[tap * 4 + 1];
      w2h[tap] = h8s{(_Float16)a.x, (_Float16)a.y, (_Float16)a.z, (_Float16)a.w, (_Float16)c.x, (_Float16)c.y, (_Float16)c.z, (_Float16)c.w};
    }
  } else {
    const float4* w2g = reinterpret_cast<const float4*>(p.w2) + (long)(lane & 31) * 36;       // 9 taps x 4 float4 per channel
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
      for (int g = 0; g < 2; ++g) w2r[tap * 2 + g] = w2g[tap * 4 + g * 2 + (lane >> 5)];
  }
  __syncthreads();

  // ---- 2. conv0_1 + ReLU on the 34 x 34 tile (vector ALU), 5 pixels per thread share each weight broadcast
  {
    constexpr int NPIX = C1W * C1W;                  // 1156
    constexpr int PPT = 5;                           // ceil(1156 / 256)
    int pix[PPT], base[PPT];
    bool inside[PPT];
#pragma unroll
    for (int q = 0; q < PPT; ++q) {
      const int i = tid + 256 * q;
      pix[q] = i < NPIX ? i : NPIX - 1;              // clamp: surplus slots recompute the last pixel (not stored)
      const int cy = pix[q] / C1W, cx = pix[q] - cy * C1W;
      base[q] = cy * XW + cx;                        // top-left tap of this conv0_1 pixel in the input patch
      const int hi = h0 - 1 + cy, wi = w0 - 1 + cx;
      inside[q] = (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
    }
#pragma unroll
    for (int cg = 0; cg < 4; ++cg) {                 // 4 output channels at a time: 20 accumulators
      float acc[PPT][4];
      const float4 bv = reinterpret_cast<const float4*>(p.b1)[cg];
#pragma unroll
      for (int q = 0; q < PPT; ++q) { acc[q][0] = bv.x; acc[q][1] = bv.y; acc[q][2] = bv.z; acc[q][3] = bv.w; }
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int toff = (tap / 3) * XW + (tap % 3);
        float4 wv4[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) wv4[c] = w1s[(cg * 4 + c) * 9 + tap];       // LDS broadcast
#pragma unroll
        for (int q = 0; q < PPT; ++q) {
          const float4 v = xin[base[q] + toff];
#pragma unroll
          for (int c = 0; c < 4; ++c)
            acc[q][c] = __builtin_fmaf(v.x, wv4[c].x, __builtin_fmaf(v.y, wv4[c].y, __builtin_fmaf(v.z, wv4[c].z, acc[q][c])));
        }
      }
#pragma unroll
      for (int q = 0; q < PPT; ++q) {
        if (tid + 256 * q < NPIX) {
          float4 o = make_float4(fmaxf(acc[q][0], 0.f), fmaxf(acc[q][1], 0.f), fmaxf(acc[q][2], 0.f), fmaxf(acc[q][3], 0.f));
          if (!inside[q]) o = make_float4(0.f, 0.f, 0.f, 0.f);
          if constexpr (H16)                              // where the unfused path stores this map as fp16
            *reinterpret_cast<h4s*>(reinterpret_cast<_Float16*>(c1) + pix[q] * C1LDH + cg * 4) =
                h4s{(_Float16)o.x, (_Float16)o.y, (_Float16)o.z, (_Float16)o.w};
          else
            *reinterpret_cast<float4*>(&c1[pix[q] * C1LD + cg * 4]) = o;
        }
      }
    }
  }
  __syncthreads();

  // ---- 3. conv0_2 on the matrix cores: wavefront wv owns tile rows 8 wv .. 8 wv + 7 (one 32-pixel row per block)
  f32x16 acc[8];
#pragma unroll
  for (int mb = 0; mb < 8; ++mb)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[mb][e] = 0.f;
  const int kg = lane >> 5, col = lane & 31;
  // A[i = lane&31 (pixel column)][k = lane>>5]: c1[(row + dy) * C1W + col + dx][8 g + 4 kg + s]
  if constexpr (H16) {
    // A[i = lane&31 (pixel column)][k = 8 kg .. 8 kg + 7]: 16 bytes of the fp16 conv0_1 pixel under the tap
    const _Float16* a_base = reinterpret_cast<const _Float16*>(c1) + ((8 * wv) * C1W + col) * C1LDH + 8 * kg;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int dy = tap / 3, dx = tap % 3;
#pragma unroll
      for (int mb = 0; mb < 8; ++mb) {
        const h8s af = *reinterpret_cast<const h8s*>(a_base + ((mb + dy) * C1W + dx) * C1LDH);
        acc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, w2h[tap], acc[mb], 0, 0, 0);
      }
    }
  } else {
    const float* a_base = c1 + ((8 * wv) * C1W + col) * C1LD + 4 * kg;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int dy = tap / 3, dx = tap % 3;
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        float4 af[8];
#pragma unroll
        for (int mb = 0; mb < 8; ++mb)
          af[mb] = *reinterpret_cast<const float4*>(a_base + ((mb + dy) * C1W + dx) * C1LD + 8 * g);
        const float4 bf = w2r[tap * 2 + g];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const float bs = s == 0 ? bf.x : s == 1 ? bf.y : s == 2 ? bf.z : bf.w;
#pragma unroll
          for (int mb = 0; mb < 8; ++mb) {
            const float as = s == 0 ? af[mb].x : s == 1 ? af[mb].y : s == 2 ? af[mb].z : af[mb].w;
            acc[mb] = __builtin_amdgcn_mfma_f32_32x32x2f32(as, bs, acc[mb], 0, 0, 0);
          }
        }
      }
    }
  }

  // ---- 4. bias + ReLU + 2x2 max + store.  C/D layout: column = lane&31 = channel, rows (= pixel columns of the tile row
  // mb) (e&3) + 8 (e>>2) + 4 kg: pixel columns 2j, 2j+1 are e, e+1 of one lane; image rows 2q, 2q+1 are blocks 2q, 2q+1.
  const float bias = p.b2[col];
  const int Hp = p.H >> 1, Wp = p.W >> 1;
  float* yg = static_cast<float*>(p.y) + (long)r * Hp * Wp * 32;
  _Float16* yh = static_cast<_Float16*>(p.y) + (long)r * Hp * Wp * 32;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int py = (h0 >> 1) + 4 * wv + q;
#pragma unroll
    for (int e = 0; e < 16; e += 2) {
      const float m = fmaxf(fmaxf(acc[2 * q][e], acc[2 * q][e + 1]), fmaxf(acc[2 * q + 1][e], acc[2 * q + 1][e + 1]));
      const int pxc = (e & 3) + 8 * (e >> 2) + 4 * kg;            // even pixel column of the pair
      const int px = (w0 >> 1) + (pxc >> 1);
      const float o = fmaxf(m + bias, 0.f);                       // relu(max(.) + b) = max(relu(. + b)); so is the fp16 rounding
      if constexpr (H16) yh[((long)py * Wp + px) * 32 + col] = (_Float16)o;
      else yg[((long)py * Wp + px) * 32 + col] = o;
    }
  }
}

}  // namespace

extern "C" int glass_local_stem_supported(int H, int W) { return H > 0 && W > 0 && H % TS == 0 && W % TS == 0; }

static int local_stem_launch(const float* x, const float* w1, const float* b1, const float* w2, const float* b2, void* y, int R,
                             int H, int W, glass_stream_t stream, bool h16) {
  GLASS_CHECK_ARG(x && w1 && b1 && w2 && b2 && y, "glass_local_stem_fused: null pointer");
  GLASS_CHECK_ARG(glass_local_stem_supported(H, W), "glass_local_stem_fused: H=%d, W=%d must be positive multiples of %d", H, W, TS);
  GLASS_CHECK_ARG((((uintptr_t)x | (uintptr_t)w1 | (uintptr_t)b1 | (uintptr_t)w2 | (uintptr_t)y) & 15) == 0,
                  "glass_local_stem_fused: pointers must be 16-byte aligned");
  if (R <= 0) return GLASS_OK;
  StemParams p;
  p.x = x; p.w1 = w1; p.b1 = b1; p.w2 = w2; p.b2 = b2; p.y = y;
  p.R = R; p.H = H; p.W = W; p.tiles_h = H / TS; p.tiles_w = W / TS;
  const long nblk = (long)R * p.tiles_h * p.tiles_w;
  GLASS_CHECK_ARG(nblk <= 0x7fffffffL, "glass_local_stem_fused: too many tiles");
  static int attr_rc = (int)hipFuncSetAttribute(reinterpret_cast<const void*>(local_stem_fused_kernel<false>),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, STEM_LDS_BYTES) |
                       (int)hipFuncSetAttribute(reinterpret_cast<const void*>(local_stem_fused_kernel<true>),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, STEM_LDS_BYTES);
  if (attr_rc != 0) {
    glass_set_error("glass_local_stem_fused: cannot reserve %d bytes of LDS (hip error %d)", STEM_LDS_BYTES, attr_rc);
    return GLASS_EHIP;
  }
  if (h16) hipLaunchKernelGGL(local_stem_fused_kernel<true>, dim3((unsigned)nblk), dim3(256), STEM_LDS_BYTES, (hipStream_t)stream, p);
  else hipLaunchKernelGGL(local_stem_fused_kernel<false>, dim3((unsigned)nblk), dim3(256), STEM_LDS_BYTES, (hipStream_t)stream, p);
  GLASS_CHECK_LAUNCH("glass_local_stem_fused");
  return GLASS_OK;
}

extern "C" int glass_local_stem_fused(const float* x, const float* w1, const float* b1, const float* w2, const float* b2,
                                      float* y, int R, int H, int W, glass_stream_t stream) {
  return local_stem_launch(x, w1, b1, w2, b2, y, R, H, W, stream, false);
}

extern "C" int glass_local_stem_fused_h16(const float* x, const float* w1, const float* b1, const float* w2, const float* b2,
                                          void* y, int R, int H, int W, glass_stream_t stream) {
  return local_stem_launch(x, w1, b1, w2, b2, y, R, H, W, stream, true);
}
