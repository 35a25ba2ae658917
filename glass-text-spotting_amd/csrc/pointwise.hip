// 1x1 convolution (pointwise GEMM) on the fp32 matrix cores of gfx950, NHWC fp32:  Y[m][co] = sum_ci X[pix(m)][ci] W[co][ci].
//
// The implicit-GEMM kernel of conv.hip stages BOTH operands through LDS and transposes its accumulators through LDS
// again to store rows; on the 1x1 bottleneck convs (K = 64 .. 2048, k-loops of 2 .. 64 tiles) it sits at 0.62-0.70 of
// the MFMA peak whatever the tile shape - per-block prologue / epilogue of the order of the loop.  This kernel is
// the F(4x4) Winograd kernel's structure without the transform:
//   * weights are pre-packed in MFMA A-fragment order (glass_pointwise_pack_weights) and stream L2 -> registers, 1 KiB
//     coalesced per load, a k-tile ahead: no LDS traffic, no barrier for them;
//   * v_mfma_f32_16x16x4_f32 with A = weights (rows = 16 output channels), B = pixels: a lane ends up with 4 consecutive
//     channels of its pixel per channel block - 8 consecutive ones over its two blocks, the packed order interleaves
//     them - so the epilogue is bias / ReLU / residual on registers and 16-byte stores, 128 contiguous bytes per pixel and
//     4 lanes: no LDS transposition, no barrier;
//   * the input tile (16 PB pixels x 32 channels) goes global -> registers -> LDS (XOR-swizzled 16-byte slots,
//     conflict-free ds_read_b128 = 4 k-steps x 2 channel blocks), double buffered, ONE barrier per k-tile;
//   * block = 16 PB pixels x 128 channels, 4 wavefronts x 32 channels; 8 PB accumulator registers (128 for PB = 16), so
//     TWO workgroups share a CU and one's epilogue / prologue hides under the other's k-loop.
// Per k-tile and wavefront: 16 PB MFMAs against 2 PB LDS reads + 4 weight loads + PB/2 input loads and LDS writes.
// Handles stride (the 1x1 stride-2 shortcuts), fused bias / ReLU / residual incl. the x2-upsampled residual of the FPN
// laterals.  fp32 in / fp32 accumulate: an exact-fp32 fma chain like glass_conv2d_nhwc.
#include "wino_common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int PN = 128;                       // output channels per block
constexpr int PK = 32;                        // input channels per k-tile

struct PwParams {
  const float* x; const float* u; const float* bias; const float* res; float* y;
  int M, H, W, Ho, Wo, Cin, Cout, stride, nk;
  int ldx, ldy, ycoff, ldr, relu, res_mode;
  int tiles_m, tiles_n;
  unsigned x_bytes, u_bytes, y_bytes, r_bytes;
  unsigned magic_hw, magic_w;                 // floor(2^32 / (Ho*Wo)), floor(2^32 / Wo)
};

__device__ __forceinline__ float comp4p(const f32x4& v, int s) { return s == 0 ? v.x : s == 1 ? v.y : s == 2 ? v.z : v.w; }

template <int PB>
__global__ __launch_bounds__(256, 2) void conv1x1_pw_f32(PwParams p) {
  constexpr int PX = 16 * PB;                 // pixels per block
  constexpr int XL = PX / 32;                 // input float4 loads per thread and k-tile
  constexpr int XS = PX * PK;                 // floats per LDS stage
  __shared__ __attribute__((aligned(16))) float smem[2 * XS];

  const int nblk = gridDim.x, bid = blockIdx.x;
  const int xcd = bid & 7, q8 = nblk >> 3, r8 = nblk & 7;
  const int logical = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  const int tile_m = logical / p.tiles_n;
  const int tile_n = logical - tile_m * p.tiles_n;
  const int m0 = tile_m * PX, n0 = tile_n * PN;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int HoWo = p.Ho * p.Wo;

  __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, (int)p.x_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t ur = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.u), 0, (int)p.u_bytes, 0x00020000);

  // ---- input role: thread = (pixel tid>>3 + 32 i, 16-byte chunk tid&7 of the 32-channel k-tile) ----
  const int chunk = tid & 7, prow = tid >> 3;
  unsigned xoff[XL];
#pragma unroll
  for (int i = 0; i < XL; ++i) {
    const int m = m0 + prow + 32 * i;
    unsigned off = OOB;
    if (m < p.M) {
      int pix = m;
      if (p.stride != 1) {
        const int n = fast_div(m, HoWo, p.magic_hw);
        const int rem = m - n * HoWo;
        const int ho = fast_div(rem, p.Wo, p.magic_w);
        const int wo = rem - ho * p.Wo;
        pix = (n * p.H + ho * p.stride) * p.W + wo * p.stride;
      }
      off = (unsigned)(pix * p.ldx + chunk * 4) * 4u;
    }
    xoff[i] = off;
  }
  float4 xreg[XL];
  auto load_x1 = [&](int i, int kt) {
    xreg[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(xr, xoff[i], kt * (PK * 4), 0));
  };
  auto load_x = [&](int kt) {
#pragma unroll
    for (int i = 0; i < XL; ++i) load_x1(i, kt);
  };
  // X[stage][pixel][32]: 128-byte rows, consecutive pixels alternate bank halves; slot ^= (pixel/2)%8 (see winograd43.hip)
  auto store_x1 = [&](int i, int stage) {
    const int px = prow + 32 * i;
    *reinterpret_cast<float4*>(smem + stage * XS + px * PK + ((chunk ^ ((px >> 1) & 7)) * 4)) = xreg[i];
  };
  auto store_x = [&](int stage) {
#pragma unroll
    for (int i = 0; i < XL; ++i) store_x1(i, stage);
  };

  // ---- MFMA role: wave wv owns channels n0 + 32 wv + [0, 32) for all PX pixels ----
  f32x4 acc[PB][2];
#pragma unroll
  for (int pb = 0; pb < PB; ++pb)
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) acc[pb][cb] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int vj = lane & 15, kg = lane >> 4;
  const int vswz = (vj >> 1) & 7;
  const float* vb[2] = {smem + vj * PK + ((0 * 4 + kg) ^ vswz) * 4, smem + vj * PK + ((1 * 4 + kg) ^ vswz) * 4};
  const unsigned a_voff = (unsigned)lane * 16u;
  f32x4 aq[2][2][2];                          // [k-tile parity][half][cb]
  // packed U: [tile_n][kt][wave][half][cb] chunks of 1 KiB (64 lanes x float4)
  auto load_a1 = [&](int j, int kt, int par) {         // j = 2 half + cb
    const int base = (((tile_n * p.nk + kt) * 4 + wv) * 4 + j) * 1024;
    aq[par][j >> 1][j & 1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ur, a_voff, base, 0));
  };
  auto load_a = [&](int kt, int par) {
#pragma unroll
    for (int j = 0; j < 4; ++j) load_a1(j, kt, par);
  };

  load_x(0);
  load_a(0, 0);
  store_x(0);
  __syncthreads();

  // one k-tile; PAR = kt & 1 is a compile-time constant (the loop below is unrolled by two) so that the weight-fragment
  // ring and the LDS stage are statically indexed registers / addresses
  // The issue order is pinned (sched_barrier): left alone the compiler waits for each LDS read right after issuing it and
  // sinks the global loads to the end of the k-tile.  The B operand of group g + 1 is read while group g's 8 MFMAs run; the
  // 4 weight and XL input loads of the next k-tile go out with the first groups and are written to the other LDS stage
  // (last read in the previous k-tile, before its barrier) with the last groups: only the barrier separates two k-tiles.
  auto ktile = [&](int kt, auto par_) {
    constexpr int PAR = decltype(par_)::value;
    const int ktn = kt + 1 < p.nk ? kt + 1 : kt;     // clamped: a harmless re-read keeps the loop one block
    f32x4 vq[2];
    vq[0] = *reinterpret_cast<const f32x4*>(vb[0] + PAR * XS);
    static_for<2 * PB>([&](auto g_) {               // group g = (half, pixel block): one LDS read, 8 MFMAs
      constexpr int g = decltype(g_)::value;
      constexpr int h = g / PB, pb = g % PB;
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (g + 1 < 2 * PB) {
        constexpr int h1 = (g + 1) / PB, pb1 = (g + 1) % PB;
        vq[(g + 1) & 1] = *reinterpret_cast<const f32x4*>(vb[h1] + PAR * XS + pb1 * 16 * PK);
      }
      if constexpr (g < XL) load_x1(g, ktn);
      if constexpr (g < 4) load_a1(g, ktn, PAR ^ 1);
      if constexpr (g >= 2 * PB - XL) store_x1(g - (2 * PB - XL), PAR ^ 1);
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
          acc[pb][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(comp4p(aq[PAR][h][cb], s), comp4p(vq[g & 1], s), acc[pb][cb], 0, 0, 0);
    });
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
  };
  for (int kt = 0; kt < p.nk; kt += 2) {
    ktile(kt, ic<0>{});
    if (kt + 1 < p.nk) ktile(kt + 1, ic<1>{});
  }

  // ---- epilogue: lane = (pixel 16 pb + vj, channels n0 + 32 wv + 8 kg + 4 cb + e): the packed order interleaves the two
  // channel blocks, so a lane holds 8 consecutive channels and 4 lanes complete a 128-byte line of their pixel ----
  __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, (int)p.y_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.res_mode != 0 ? p.res : p.y), 0,
                                                                 (int)(p.res_mode != 0 ? p.r_bytes : 0u), 0x00020000);
  const int cbase = n0 + 32 * wv + 8 * kg;       // + 4 cb
  const unsigned ldy4 = (unsigned)p.ldy * 4u, ldr4 = (unsigned)p.ldr * 4u;
  const float lo2 = p.relu == 2 ? 0.f : __builtin_nanf(""), lo1 = p.relu == 1 ? 0.f : __builtin_nanf("");
  f32x4 bv[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
  if (p.bias != nullptr) {
    bv[0] = *reinterpret_cast<const f32x4*>(p.bias + cbase);
    bv[1] = *reinterpret_cast<const f32x4*>(p.bias + cbase + 4);
  }
  const int HoWo2 = (p.Ho >> 1) * (p.Wo >> 1);
  // compile-time residual variants, and the residual loads of a batch of pixel blocks all go out before its first store:
  // a load issued after a store waits for it (one in-order counter), which serialised the pixel blocks of a wavefront
  auto epilogue = [&](auto res_c) {
    constexpr bool RES = decltype(res_c)::value != 0;
    constexpr int PBB = PB < 8 ? PB : 8;
#pragma unroll
    for (int b0 = 0; b0 < PB; b0 += PBB) {
      unsigned yo[PBB];
      f32x4 rq[RES ? PBB : 1][2];
#pragma unroll
      for (int i = 0; i < PBB; ++i) {
        const int m = m0 + 16 * (b0 + i) + vj;
        const bool ok = m < p.M;
        yo[i] = ok ? (unsigned)m * ldy4 + (unsigned)(p.ycoff + cbase) * 4u : OOB;
        if constexpr (RES) {
          unsigned ro = OOB;
          if (ok) {
            int rp = m;
            if (p.res_mode == 2) {                      // x2 nearest-upsampled residual [N, Ho/2, Wo/2, ldr]
              const int n = fast_div(m, HoWo, p.magic_hw);
              const int rem = m - n * HoWo;
              const int ho = fast_div(rem, p.Wo, p.magic_w);
              const int wo = rem - ho * p.Wo;
              rp = n * HoWo2 + (ho >> 1) * (p.Wo >> 1) + (wo >> 1);
            }
            ro = (unsigned)rp * ldr4 + (unsigned)cbase * 4u;
          }
          rq[i][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rr, ro, 0, 0));
          rq[i][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rr, ro, 16, 0));
        }
      }
#pragma unroll
      for (int i = 0; i < PBB; ++i) {
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
          f32x4 v = acc[b0 + i][cb] + bv[cb];
          v.x = fmaxf(v.x, lo2); v.y = fmaxf(v.y, lo2); v.z = fmaxf(v.z, lo2); v.w = fmaxf(v.w, lo2);
          if constexpr (RES) v = v + rq[i][cb];
          v.x = fmaxf(v.x, lo1); v.y = fmaxf(v.y, lo1); v.z = fmaxf(v.z, lo1); v.w = fmaxf(v.w, lo1);
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), yr, yo[i], cb * 16, 0);
        }
      }
    }
  };
  if (p.res_mode != 0) epilogue(ic<1>{}); else epilogue(ic<0>{});
}

// W [Cout][1][1][Cin] -> [cout/128][cin/32][wave][half][cb][lane][s]  with
//   cout = 128 tn + 32 wave + 8 ((lane&15)>>2) + 4 cb + (lane&3),   cin = 32 kt + 16 half + 4 (lane>>4) + s
__global__ void pw_pack_weights_kernel(const float* __restrict__ w, float* __restrict__ u, int Cout, int Cin) {
  const long total = (long)Cout * Cin;
  const int nk = Cin / PK;
  for (long o = (long)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (long)gridDim.x * blockDim.x) {
    long r = o;
    const int s = (int)(r & 3); r >>= 2;
    const int lane = (int)(r & 63); r >>= 6;
    const int cb = (int)(r & 1); r >>= 1;
    const int half = (int)(r & 1); r >>= 1;
    const int wave = (int)(r & 3); r >>= 2;
    const int kt = (int)(r % nk);
    const int tn = (int)(r / nk);
    const int co = tn * PN + 32 * wave + 8 * ((lane & 15) >> 2) + 4 * cb + (lane & 3);
    const int ci = kt * PK + 16 * half + 4 * (lane >> 4) + s;
    u[o] = w[(long)co * Cin + ci];
  }
}

}  // namespace

extern "C" int glass_pointwise_supported(const glass_conv_desc* d) {
  if (!d) return 0;
  const long M = (long)d->N * d->Ho * d->Wo;
  const long xb = (long)d->N * d->H * d->W * d->ldx * 4, yb = M * d->ldy * 4;
  const long rb = d->res_mode == 1 ? M * d->ldr * 4 : d->res_mode == 2 ? (long)d->N * (d->Ho / 2) * (d->Wo / 2) * d->ldr * 4 : 0;
  return d->KH == 1 && d->KW == 1 && d->pad_h == 0 && d->pad_w == 0 && d->stride_h == d->stride_w && d->stride_h >= 1 &&
         d->Cin % PK == 0 && d->Cout % PN == 0 && d->ldx % 4 == 0 && d->ldx >= d->Cin && d->y_cstride == 1 && d->ldy % 4 == 0 &&
         d->y_coff % 4 == 0 && d->y_coff >= 0 && d->y_coff + d->Cout <= d->ldy &&
         (d->res_mode == 0 || (d->ldr % 4 == 0 && d->ldr >= d->Cout)) && (d->res_mode != 2 || (d->Ho % 2 == 0 && d->Wo % 2 == 0)) &&
         d->Ho == (d->H - 1) / d->stride_h + 1 && d->Wo == (d->W - 1) / d->stride_w + 1 &&
         M < 0x7fffffffL && xb < 0x7fffff00L && yb < 0x7fffff00L && rb < 0x7fffff00L && (long)d->Cout * d->Cin * 4 < 0x7fffff00L;
}

extern "C" size_t glass_pointwise_weight_floats(int Cout, int Cin) { return (size_t)Cout * (size_t)Cin; }

extern "C" int glass_pointwise_pack_weights(const float* w, int Cout, int Cin, float* u_packed, glass_stream_t stream) {
  GLASS_CHECK_ARG(w && u_packed, "glass_pointwise_pack_weights: null pointer");
  GLASS_CHECK_ARG(Cout > 0 && Cin > 0 && Cout % PN == 0 && Cin % PK == 0,
                  "glass_pointwise_pack_weights: Cout=%d must be a multiple of 128 and Cin=%d a multiple of 32", Cout, Cin);
  const long total = (long)Cout * Cin;
  const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(pw_pack_weights_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, u_packed, Cout, Cin);
  GLASS_CHECK_LAUNCH("glass_pointwise_pack_weights");
  return GLASS_OK;
}

extern "C" int glass_conv1x1_pointwise_nhwc(const glass_conv_desc* d, const float* x, const float* u_packed, const float* bias,
                                            const float* residual, float* y, glass_stream_t stream) {
  GLASS_CHECK_ARG(d && x && u_packed && y, "glass_conv1x1_pointwise_nhwc: null pointer");
  GLASS_CHECK_ARG(glass_pointwise_supported(d),
                  "glass_conv1x1_pointwise_nhwc: needs 1x1 / pad 0 / square stride, Cin%%32==0, Cout%%128==0, unit channel stride, "
                  "operands < 2 GiB (got Cin=%d Cout=%d k=%dx%d s=%d p=%d)", d->Cin, d->Cout, d->KH, d->KW, d->stride_h, d->pad_h);
  GLASS_CHECK_ARG(d->res_mode == 0 || residual != nullptr, "glass_conv1x1_pointwise_nhwc: res_mode set but residual is null");
  GLASS_CHECK_ARG(((uintptr_t)x & 15) == 0 && ((uintptr_t)u_packed & 15) == 0 && ((uintptr_t)y & 15) == 0 &&
                      (bias == nullptr || ((uintptr_t)bias & 15) == 0) && (residual == nullptr || ((uintptr_t)residual & 15) == 0),
                  "glass_conv1x1_pointwise_nhwc: pointers must be 16-byte aligned");
  if (d->N == 0) return GLASS_OK;
  PwParams p;
  p.x = x; p.u = u_packed; p.bias = bias; p.res = residual; p.y = y;
  p.M = d->N * d->Ho * d->Wo; p.H = d->H; p.W = d->W; p.Ho = d->Ho; p.Wo = d->Wo; p.Cin = d->Cin; p.Cout = d->Cout;
  p.stride = d->stride_h; p.nk = d->Cin / PK;
  p.ldx = d->ldx; p.ldy = d->ldy; p.ycoff = d->y_coff; p.ldr = d->ldr; p.relu = d->relu; p.res_mode = d->res_mode;
  p.x_bytes = (unsigned)((long)d->N * d->H * d->W * d->ldx * 4);
  p.u_bytes = (unsigned)((long)d->Cout * d->Cin * 4);
  p.y_bytes = (unsigned)((long)p.M * d->ldy * 4);
  p.r_bytes = d->res_mode == 1 ? (unsigned)((long)p.M * d->ldr * 4)
            : d->res_mode == 2 ? (unsigned)((long)d->N * (d->Ho / 2) * (d->Wo / 2) * d->ldr * 4) : 0u;
  p.magic_hw = (unsigned)(0x100000000ULL / (unsigned long long)(d->Ho * d->Wo));
  p.magic_w = (unsigned)(0x100000000ULL / (unsigned long long)d->Wo);
  p.tiles_n = d->Cout / PN;
  // 256-pixel blocks when they still give every CU ~1.5 workgroups, else 128-pixel blocks
  const bool big = (long)cdiv(p.M, 256) * p.tiles_n >= 384;
  p.tiles_m = cdiv(p.M, big ? 256 : 128);
  const long nblk = (long)p.tiles_m * p.tiles_n;
  GLASS_CHECK_ARG(nblk > 0 && nblk <= 0x7fffffffL, "glass_conv1x1_pointwise_nhwc: bad grid");
  if (big)
    hipLaunchKernelGGL(conv1x1_pw_f32<16>, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, p);
  else
    hipLaunchKernelGGL(conv1x1_pw_f32<8>, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, p);
  GLASS_CHECK_LAUNCH("glass_conv1x1_pointwise_nhwc");
  return GLASS_OK;
}
