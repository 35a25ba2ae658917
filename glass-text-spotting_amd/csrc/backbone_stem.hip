// Fused ResNet stem (detectron2 BasicStem as restated in oracle/glass_cpu.py resnet50_fpn): conv 7x7 stride 2
// pad 3 (3 -> 64 channels, BatchNorm folded) + ReLU + max-pool 3x3 stride 2 pad 1 in ONE kernel:
//   x [N,H,W,4] NHWC4 (channel 3 = 0)  ->  y [N,H/4,W/4,64]        (H, W multiples of 4)
//
// As two launches this pair is the largest HBM round trip of the step: the implicit-GEMM kernel writes the [N,H/2,W/2,64]
// map (537 MB at 8 x 1024 x 1024) and the pool reads it back - 0.55 ms at 72 TFLOP/s (Cin = 4: K = 196 with a per-thread
// filter-tap decode) + 0.135 ms.  Fused, the convolution map never leaves the CU: 134 MB in, 134 MB out.
//
// One workgroup (256 threads = 4 wavefronts, one per SIMD, all 512 registers) owns 256 convolution columns x `tph` pooled
// rows of one image and walks DOWN the convolution rows 2 py0 - 1 ... 2 (py0 + tph) - 1:
//   * input rows stream through a 10-slot LDS ring, 3 floats per pixel (the zero 4th channel is dropped): convolution row oy
//     reads input rows 2 oy - 3 ... 2 oy + 3; the two rows the NEXT convolution row adds are fetched global -> registers at
//     the top of a row and written to the ring under its last MFMAs; one barrier per convolution row;
//   * the convolution is an implicit GEMM on v_mfma_f32_32x32x2_f32 with M = 32 consecutive convolution columns (wavefront w
//     owns columns 64 w ... 64 w + 63 = two M-blocks), N = 64 channels (two N-blocks), K = 7 filter rows x 24 (= 7 taps x 3
//     channels, padded from 21; the padding reads the next pixel against zero weights): 168 instead of the 196 of the
//     NHWC4 layout.  The B operand - ALL of the filter, 7 x 12 k-steps x 2 N-blocks = 168 registers per lane - is loaded
//     once per workgroup; the A operand is one ds_read_b64 per two k-steps (lane half kg supplies kk = 12 kg + s, so a lane
//     reads 12 consecutive floats of the packed row): 84 LDS reads against 336 MFMAs per M-block pair and row;
//   * pooling over y happens in registers (the same lane holds the same column and channel of consecutive rows:
//     m = max(row 2 py - 1, row 2 py, row 2 py + 1), the odd row is kept for the next pooled row), pooling over x through a
//     [257][64] LDS tile: column 0 is the LEFT HALO column 2 px0 - 1, which belongs to the neighbouring workgroup's range -
//     16 pooled columns need 33 convolution columns - and is computed here on the vector ALU (147 MACs x 64 channels per
//     row, split over the four wavefronts by filter row: < 2 % of the row's MFMA time);
//   * bias + ReLU after the max (relu(max(.) + b) = max(relu(. + b))), 256-byte-per-pixel stores.
// fp32 in, fp32 accumulate (exact fp32 fma chains on the matrix cores), fp32 out: the results of glass_conv2d_nhwc +
// glass_maxpool2d_nhwc up to fp32 summation order (tests/test_gpu_f_ops.py).
#include "common.h"
#include <cstdint>
#include <type_traits>
#include <utility>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

namespace {

constexpr int XCOLS = 256;                       // convolution columns per workgroup (8 M-blocks of 32, two per wavefront)
constexpr int RING = 10;                         // input rows resident in LDS: 7 in use + 2 arriving + 1 spare
constexpr int RPX = 2 * XCOLS + 8;               // input pixels per ring row: columns 2 cx0 - 5 ... 2 cx0 + 514 (520)
constexpr int RS = 3 * RPX;                      // floats per ring row (1560: even, so every ds_read_b64 below is aligned)
constexpr int SCOLS = XCOLS + 1;                 // pooling tile columns (0 = left halo)
constexpr int LDS_RING = RING * RS;              // floats
constexpr int LDS_S = SCOLS * 64;
constexpr int LDS_HP = 2 * 8 * 64;               // halo partial sums [wavefront][kg][channel], double-buffered by row parity
constexpr int STEM_LDS_BYTES = (LDS_RING + LDS_S + LDS_HP) * 4;
constexpr int LPT = (RPX + 255) / 256;           // float4 loads per thread and input row (3)

struct BStemParams {
  const float* x;      // [N,H,W,4]
  const float* w;      // [64][7][7][4]  (BN folded; w[..][3] ignored)
  const float* b;      // [64]
  float* y;            // [N,H/4,W/4,64]
  int N, H, W, Ho, Wo, Hp, Wp, tph, bands, xblocks;
};

__device__ __forceinline__ int ring_slot(int iy) { return (iy + 16 * RING) % RING; }      // iy >= -5

template <int B, int E, class F> __device__ __forceinline__ void static_for(F&& f) {
  if constexpr (B < E) {
    f(std::integral_constant<int, B>{});
    static_for<B + 1, E>(f);
  }
}

__global__ __launch_bounds__(256, 1) void backbone_stem_fused_kernel(BStemParams p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* ring = smem;                            // [RING][RS]
  float* S = smem + LDS_RING;                    // [SCOLS][64]
  float* hpart_all = S + LDS_S;                  // [2][4][2][64]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kg = lane >> 5, col = lane & 31;
  int bidx = blockIdx.x;
  const int xb = bidx % p.xblocks; bidx /= p.xblocks;
  const int band = bidx % p.bands;
  const int n = bidx / p.bands;
  const int py0 = band * p.tph;
  const int tph = min(p.tph, p.Hp - py0);
  const int cx0 = xb * XCOLS;                    // first convolution column of this workgroup
  const int ix0 = 2 * cx0 - 5;                   // input column of ring pixel 0
  const int oy_first = max(2 * py0 - 1, 0), oy_last = 2 * (py0 + tph) - 1;      // convolution rows computed here
  const bool has_halo = xb > 0;                  // column cx0 - 1 exists (else it is the pool's padding)

  // ---- the whole filter as B fragments: B[k = kg][j = col] of k-step (ky, s) and N-block nb = W[32 nb + col][ky][kk = 12 kg + s]
  float wr[7][12][2];
#pragma unroll
  for (int ky = 0; ky < 7; ++ky)
#pragma unroll
    for (int s = 0; s < 12; ++s) {
      const int kk = 12 * kg + s;                // 0 .. 23; kk = 3 kx + c, kx < 7
      const int kx = kk / 3, c = kk - 3 * kx;
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
        wr[ky][s][nb] = kk < 21 ? p.w[(((long)(32 * nb + col) * 7 + ky) * 7 + kx) * 4 + c] : 0.f;
    }

  const float4* xg = reinterpret_cast<const float4*>(p.x) + (long)n * p.H * p.W;
  // global -> registers: input row iy, this thread's pixels q = tid + 256 j (zero outside the image = the conv's padding)
  auto fetch_row = [&](int iy, float4 (&v)[LPT]) {
    const bool row_ok = (unsigned)iy < (unsigned)p.H;
#pragma unroll
    for (int j = 0; j < LPT; ++j) {
      const int q = tid + 256 * j, ix = ix0 + q;
      v[j] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (row_ok && q < RPX && (unsigned)ix < (unsigned)p.W) v[j] = xg[(long)iy * p.W + ix];
    }
  };
  auto stash_row = [&](int iy, const float4 (&v)[LPT]) {
    float* r = ring + ring_slot(iy) * RS;
#pragma unroll
    for (int j = 0; j < LPT; ++j) {
      const int q = tid + 256 * j;
      if (q < RPX) { r[3 * q] = v[j].x; r[3 * q + 1] = v[j].y; r[3 * q + 2] = v[j].z; }
    }
  };

  // ---- prologue: the 7 input rows of the first convolution row
  {
    float4 v[LPT];
    for (int iy = 2 * oy_first - 3; iy <= 2 * oy_first + 3; ++iy) {
      fetch_row(iy, v);
      stash_row(iy, v);
    }
  }
  __syncthreads();

  // A fragment address (floats, within a ring row): column x - cx0 = 32 (2 wv + mb) + col, tap kk = 12 kg + s -> 6 (x - cx0) + 6 + kk
  const int a_off0 = 6 * (32 * (2 * wv) + col) + 6 + 12 * kg;
  const int a_off1 = a_off0 + 6 * 32;
  const bool mb_ok0 = cx0 + 32 * (2 * wv) < p.Wo, mb_ok1 = cx0 + 32 * (2 * wv + 1) < p.Wo;       // wavefront-uniform

  // [M-block][N-block]: after an odd row 2 py - 1 it holds that row (the "row above" of pooled row py); the even row 2 py is
  // folded INTO it (max), and the odd row 2 py + 1 closes the pooled row: max(prev, row) goes out, the row itself stays
  f32x16 prev[2][2];
  float hprev = -INFINITY;                       // the same for the halo column (threads 0 .. 63: channel = tid)
#pragma unroll
  for (int mb = 0; mb < 2; ++mb)
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int e = 0; e < 16; ++e) prev[mb][nb][e] = -INFINITY;

  for (int oy = oy_first; oy <= oy_last; ++oy) {
    // (1) next row's two new input rows: global -> registers (consumed at (4))
    float4 nv0[LPT], nv1[LPT];
    const bool more = oy < oy_last;
    if (more) { fetch_row(2 * oy + 4, nv0); fetch_row(2 * oy + 5, nv1); }

    // (2) left halo column cx0 - 1 on the vector ALU, from the B fragments this lane already holds (no weight loads): lane
    // (kg, col) owns W[32 nb + col][ky][12 kg + s], i.e. half of the 24-float filter row of two channels; filter rows are split
    // over the wavefronts (2, 2, 2, 1), the 4 x 2 partial sums per channel are added in a fixed order at (5).
    // (double-buffered by row parity: wavefront 0 reads row oy's parts at (5) while the others may already write row oy + 1's)
    float* hpart = hpart_all + (oy & 1) * 512;
    if (has_halo) {
      float h0 = 0.f, h1 = 0.f;
      auto halo_ky = [&](auto ky_) {
        constexpr int ky = decltype(ky_)::value;
        const float* r = ring + ring_slot(2 * oy - 3 + ky) * RS + 12 * kg;      // halo column: floats 0 .. 23 of the row
        const float4 i0 = *reinterpret_cast<const float4*>(r), i1 = *reinterpret_cast<const float4*>(r + 4),
                     i2 = *reinterpret_cast<const float4*>(r + 8);
        const float in[12] = {i0.x, i0.y, i0.z, i0.w, i1.x, i1.y, i1.z, i1.w, i2.x, i2.y, i2.z, i2.w};
#pragma unroll
        for (int s = 0; s < 12; ++s) {
          h0 = __builtin_fmaf(in[s], wr[ky][s][0], h0);
          h1 = __builtin_fmaf(in[s], wr[ky][s][1], h1);
        }
      };
      if (wv == 0) { halo_ky(std::integral_constant<int, 0>{}); halo_ky(std::integral_constant<int, 1>{}); }
      else if (wv == 1) { halo_ky(std::integral_constant<int, 2>{}); halo_ky(std::integral_constant<int, 3>{}); }
      else if (wv == 2) { halo_ky(std::integral_constant<int, 4>{}); halo_ky(std::integral_constant<int, 5>{}); }
      else halo_ky(std::integral_constant<int, 6>{});
      hpart[(wv * 2 + kg) * 64 + col] = h0;                 // [wavefront][kg][channel]
      hpart[(wv * 2 + kg) * 64 + 32 + col] = h1;
    }

    // (3) the row's convolution: 2 M-blocks x 2 N-blocks x 84 k-steps
    f32x16 acc[2][2];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[mb][nb][e] = 0.f;
    if (mb_ok0) {
      // 21 groups of 4 k-steps (filter row ky = g / 3, floats 4 (g % 3) .. + 3 of the lane's 12): the NEXT group's A fragments
      // (two ds_read2_b64: this lane's 4 floats for each M-block) are requested before this group's 16 MFMAs
      // (an M-block past the image's right edge reads zero-filled ring pixels: finite values nobody stores)
      auto a_ptr = [&](int g) { return ring + ring_slot(2 * oy - 3 + g / 3) * RS + 4 * (g % 3); };
      // (ring offsets 6 col + 6 + 12 kg + 4 (g % 3) are 8-byte aligned, 16-byte aligned only for odd columns: the access
      //  type says so - `f4a8` - and the compiler emits ds_read2_b64; a float4 dereference promised an alignment the address
      //  does not have and a ds_read_b128 off its natural alignment is replayed by the LDS)
      struct __attribute__((aligned(8))) f4a8 { float x, y, z, w; };
      auto ld4 = [](const float* q) { const f4a8 v = *reinterpret_cast<const f4a8*>(q); return make_float4(v.x, v.y, v.z, v.w); };
      float4 c0 = ld4(a_ptr(0) + a_off0), c1 = ld4(a_ptr(0) + a_off1);
      static_for<0, 21>([&](auto g_) {
        constexpr int g = decltype(g_)::value;
        constexpr int ky = g / 3, s0 = 4 * (g % 3);
        float4 n0 = c0, n1 = c1;
        if constexpr (g + 1 < 21) {
          n0 = ld4(a_ptr(g + 1) + a_off0);
          n1 = ld4(a_ptr(g + 1) + a_off1);
        }
        __builtin_amdgcn_sched_barrier(0);
        const float a0[4] = {c0.x, c0.y, c0.z, c0.w}, a1[4] = {c1.x, c1.y, c1.z, c1.w};
#pragma unroll
        for (int h = 0; h < 4; ++h) {
          acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[h], wr[ky][s0 + h][0], acc[0][0], 0, 0, 0);
          acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[h], wr[ky][s0 + h][1], acc[0][1], 0, 0, 0);
          acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[h], wr[ky][s0 + h][0], acc[1][0], 0, 0, 0);
          acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[h], wr[ky][s0 + h][1], acc[1][1], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        c0 = n0; c1 = n1;
      });
    }

    // (3b) fold the row into the 3-row maxima.  Even row 2 py: prev = max(row 2 py - 1, this).  Odd row 2 py + 1: closes pooled
    // row py - max(prev, this) goes to the pooling tile - and stays as the "row above" of pooled row py + 1.  The band's first
    // row 2 py0 - 1 (odd) only seeds `prev`.
    const bool odd = oy & 1;
    const bool emit = odd && oy > 2 * py0 - 1;
    if (!odd) {
#pragma unroll
      for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
          for (int e = 0; e < 16; ++e) prev[mb][nb][e] = fmaxf(prev[mb][nb][e], acc[mb][nb][e]);
    } else {
      if (emit) {
        // y-pooled row -> S[1 + column][channel]; D layout: channel = 32 nb + col, column = 32 (2 wv + mb) + (e&3) + 8 (e>>2) + 4 kg
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
          if (mb == 0 ? mb_ok0 : mb_ok1) {
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
              for (int e = 0; e < 16; ++e) {
                const int c_ = 32 * (2 * wv + mb) + (e & 3) + 8 * (e >> 2) + 4 * kg;
                S[(1 + c_) * 64 + 32 * nb + col] = fmaxf(prev[mb][nb][e], acc[mb][nb][e]);
              }
          }
        }
      }
#pragma unroll
      for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) prev[mb][nb] = acc[mb][nb];
    }

    // (4) the prefetched input rows -> ring (slots of rows 2 oy - 6, 2 oy - 5: not read by anybody in this row)
    if (more) { stash_row(2 * oy + 4, nv0); stash_row(2 * oy + 5, nv1); }
    __syncthreads();                                                   // ring, hpart and S (rows that emit) are complete

    // (5) halo column: add the 4 x 2 partial sums (fixed order), fold like the main rows, publish as S column 0
    if (tid < 64) {
      float h = -INFINITY;
      if (has_halo) {
        h = hpart[tid];
#pragma unroll
        for (int q = 1; q < 8; ++q) h += hpart[q * 64 + tid];
      }
      if (!odd) hprev = fmaxf(hprev, h);
      else {
        if (emit) S[tid] = fmaxf(hprev, h);
        hprev = h;
      }
    }
    if (emit) {
      __syncthreads();
      // (6) pool over x, bias, ReLU, store: pooled column p (local) = max of S columns 2 p, 2 p + 1, 2 p + 2
      const int py = (oy - 1) >> 1;
      const int ch4 = tid & 15, pp = tid >> 4;
      const float4 bv = reinterpret_cast<const float4*>(p.b)[ch4];
      float4* yg = reinterpret_cast<float4*>(p.y) + (((long)n * p.Hp + py) * p.Wp + (cx0 >> 1)) * 16 + ch4;
      const float4* S4 = reinterpret_cast<const float4*>(S) + ch4;
#pragma unroll
      for (int it = 0; it < XCOLS / 2 / 16; ++it) {
        const int pl = pp + 16 * it;
        if ((cx0 >> 1) + pl < p.Wp) {
          const float4 a = S4[(2 * pl) * 16], b = S4[(2 * pl + 1) * 16], c = S4[(2 * pl + 2) * 16];
          float4 o;
          o.x = fmaxf(fmaxf(fmaxf(a.x, b.x), c.x) + bv.x, 0.f);
          o.y = fmaxf(fmaxf(fmaxf(a.y, b.y), c.y) + bv.y, 0.f);
          o.z = fmaxf(fmaxf(fmaxf(a.z, b.z), c.z) + bv.z, 0.f);
          o.w = fmaxf(fmaxf(fmaxf(a.w, b.w), c.w) + bv.w, 0.f);
          yg[(long)pl * 16] = o;
        }
      }
      // (the next write of S happens after the next emitting row's barrier at (4): two barriers away)
    }
  }
}

}  // namespace

extern "C" int glass_backbone_stem_supported(int H, int W) { return H > 0 && W > 0 && H % 4 == 0 && W % 4 == 0; }

extern "C" int glass_backbone_stem_fused(const float* x, const float* w, const float* bias, float* y, int N, int H, int W,
                                         glass_stream_t stream) {
  GLASS_CHECK_ARG(x && w && bias && y, "glass_backbone_stem_fused: null pointer");
  GLASS_CHECK_ARG(glass_backbone_stem_supported(H, W), "glass_backbone_stem_fused: H=%d, W=%d must be positive multiples of 4", H, W);
  GLASS_CHECK_ARG((((uintptr_t)x | (uintptr_t)w | (uintptr_t)bias | (uintptr_t)y) & 15) == 0,
                  "glass_backbone_stem_fused: pointers must be 16-byte aligned");
  GLASS_CHECK_ARG((long)N * H * W * 16 < 0x7fffffff00L, "glass_backbone_stem_fused: input too large");
  if (N <= 0) return GLASS_OK;
  BStemParams p;
  p.x = x; p.w = w; p.b = bias; p.y = y;
  p.N = N; p.H = H; p.W = W; p.Ho = H / 2; p.Wo = W / 2; p.Hp = H / 4; p.Wp = W / 4;
  p.xblocks = (p.Wo + XCOLS - 1) / XCOLS;
  // pooled rows per workgroup: every band recomputes ONE convolution row (2 py0 - 1), so tall bands waste less (16 rows: 3 %),
  // but the grid must still cover the chip: the tallest band that gives >= 256 workgroups, else 2 rows
  int tph = 16;
  while (tph > 2 && (long)N * ((p.Hp + tph - 1) / tph) * p.xblocks < 256) tph >>= 1;
  p.tph = tph;
  p.bands = (p.Hp + tph - 1) / tph;
  const long nblk = (long)N * p.bands * p.xblocks;
  GLASS_CHECK_ARG(nblk <= 0x7fffffffL, "glass_backbone_stem_fused: too many tiles");
  static int attr_rc = (int)hipFuncSetAttribute(reinterpret_cast<const void*>(backbone_stem_fused_kernel),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, STEM_LDS_BYTES);
  if (attr_rc != 0) {
    glass_set_error("glass_backbone_stem_fused: cannot reserve %d bytes of LDS (hip error %d)", STEM_LDS_BYTES, attr_rc);
    return GLASS_EHIP;
  }
  hipLaunchKernelGGL(backbone_stem_fused_kernel, dim3((unsigned)nblk), dim3(256), STEM_LDS_BYTES, (hipStream_t)stream, p);
  GLASS_CHECK_LAUNCH("glass_backbone_stem_fused");
  return GLASS_OK;
}
