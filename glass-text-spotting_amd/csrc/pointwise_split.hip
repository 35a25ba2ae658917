// 1x1 convolution (pointwise GEMM) with EXACT fp32 products on the bf16 matrix cores of gfx950, NHWC fp32 in / fp32 out.
//
// The fp32 MFMA (v_mfma_f32_16x16x4_f32: 2048 FLOP in 32 cycles) is the slowest matrix instruction of the chip, and the 1x1
// layers of the backbone sit on it at 0.77 of its peak (pointwise.hip) while they move a fraction of what HBM could.  An
// fp32 number is the EXACT sum of three bf16 numbers (8 + 8 + 8 significant bits: v = h + m + l, each piece a rounding to bf16,
// each remainder an exact subtraction), and the product of two bf16 numbers (16 significant bits) is exact in fp32.  So
//     x * w = sum over the nine pairs (xq, wr), q, r in {h, m, l}
// holds EXACTLY, and nine v_mfma_f32_16x16x32_bf16 (16384 FLOP in 16 cycles each) with one fp32 accumulator compute the
// same sum of products as eight v_mfma_f32_16x16x4_f32, with the same fp32 accumulation, in 9 x 16 = 144 instead of 8 x 32 =
// 256 matrix cycles - on a pipe that, unlike the fp32 MFMA, does not share its issue with the vector ALU.  Nothing is dropped
// (the six-product form that leaves out the three terms below 2^-23 is a compile-time variant kept for measurement only:
// NPROD), nothing is rounded before the accumulator: the result differs from the fp32-MFMA kernel's only by the order of the
// additions (measured against float64: scripts/exp_pw_split.py).  Not representable: inf / nan inputs (h = inf, v - h = nan) and
// |v| > 3.39e38 (bf16(v) = inf), which the fp32 kernel would carry through; activations and weights of a forward pass are finite.  A piece below 2^-126 (the
// low bits of an operand below ~2^-110) is a bf16 denormal and may be flushed by the matrix pipe: < 1e-37 |w| per product.
//
// Structure = pointwise.hip's (block = 16 PB pixels x 128 channels, 4 wavefronts x 32 channels, weights pre-split and
// pre-packed in MFMA A-fragment order streaming L2 -> registers a k-tile ahead, pixels global -> registers -> LDS, one barrier
// per k-tile, epilogue on registers with 16-byte stores), with the split of the pixels done once per block on the way into
// LDS (4 v_sub + 1.5 v_cvt_pk + 2 unpacks per element): three bf16 planes [pixel][32 k], 64-byte rows, 16-byte slots XOR-swizzled by (pixel / 4) % 4 so that the
// ds_read_b128 of a B fragment (lane = pixel lane & 15, k-octet lane >> 4) is conflict-free.
#include "wino_common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

namespace {

constexpr int SN = 128;                       // output channels per block
constexpr int SK = 32;                        // input channels per k-tile (one 16x16x32 MFMA deep)

struct PwsParams {
  const float* x; const void* u; const float* bias; const float* res; float* y;
  int M, H, W, Ho, Wo, Cin, Cout, stride, nk;
  int ldx, ldy, ycoff, ldr, relu, res_mode;
  int tiles_m, tiles_n;
  unsigned x_bytes, u_bytes, y_bytes, r_bytes;
  unsigned magic_hw, magic_w;                 // floor(2^32 / (Ho*Wo)), floor(2^32 / Wo)
  unsigned long long* dbg;                    // timing build (-DGLASS_PWS_STAMPS) only
  // DUAL form (a bottleneck block's shortcut folded into its conv3: Y = [X1(strided) | X2] [W1 | W2]^T, one accumulator): the
  // k-tiles [0, nk1) read x (Cin channels, `stride`), the k-tiles [nk1, nk) read x2 [M][ldx2] on the output grid
  const float* x2; int nk1, ldx2; unsigned x2_bytes;
};

#ifdef GLASS_PWS_STAMPS   // scripts/build_variant_lib.sh pwst -DGLASS_PWS_STAMPS: phase totals of wavefront 0 of a mid-grid workgroup
unsigned long long* g_pws_dbg = nullptr;
#define PWS_STAMP(k) { __builtin_amdgcn_sched_barrier(0); const unsigned long long tn = __builtin_amdgcn_s_memtime(); st[k] += tn - tlast; tlast = tn; __builtin_amdgcn_sched_barrier(0); }
#else
#define PWS_STAMP(k) {}
#endif

// v = h + m + l EXACTLY, each piece a bf16: h = bf16(v), m = bf16(v - h), l = v - h - m with round-to-nearest-even conversions
// (v_cvt_pk_bf16_f32, two elements per instruction).  v - h is exact (at most 16 significant bits: h and v share their leading
// bits), so is (v - h) - m, and what is left after two 8-bit roundings of a 24-bit number has at most 8 significant bits: the
// last conversion changes nothing.  Rounding instead of truncating halves every piece: |m| <= 2^-8 |v|, |l| <= 2^-16 |v| - it makes
// no difference to the nine-product sum (exact either way) and bounds what the six-product form leaves out (m l' + l m' + l l')
// by 2^-23 |v w|.  Returns the three pieces of (v0, v1) as packed bf16 pairs, element order v0, v1.
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split3_pair(float v0, float v1, unsigned& hp, unsigned& mp, unsigned& lp) {
  hp = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2{v0, v1}), bf16x2));
  const float r0 = v0 - __builtin_bit_cast(float, hp << 16), r1 = v1 - __builtin_bit_cast(float, hp & 0xffff0000u);
  mp = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2{r0, r1}), bf16x2));
  const float l0 = r0 - __builtin_bit_cast(float, mp << 16), l1 = r1 - __builtin_bit_cast(float, mp & 0xffff0000u);
  lp = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2{l0, l1}), bf16x2));
}

template <int PB, int NPROD, bool DUAL = false>
__global__ __launch_bounds__(256, 2) void conv1x1_pw_split(PwsParams p) {
  constexpr int PX = 16 * PB;                 // pixels per block
  constexpr int XL = PX / 32;                 // input float4 loads per thread and k-tile
  constexpr int NW = 4;                       // wavefronts, 32 output channels each
  constexpr int PLANE = PX * 64;              // bytes of one bf16 plane [PX][32]
  constexpr int STAGE = 3 * PLANE;
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * STAGE];

  const int nblk = gridDim.x, bid = blockIdx.x;
  const int xcd = bid & 7, q8 = nblk >> 3, r8 = nblk & 7;
  const int logical = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  const int tile_m = logical / p.tiles_n;
  const int tile_n = logical - tile_m * p.tiles_n;
  const int m0 = tile_m * PX, n0 = tile_n * SN;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int HoWo = p.Ho * p.Wo;

  __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, (int)p.x_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t ur = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.u), 0, (int)p.u_bytes, 0x00020000);

  // ---- input role: thread = (pixel tid>>3 + 32 i, 16-byte chunk tid&7 of the 32-channel k-tile) ----
  const int chunk = tid & 7, prow = tid >> 3;
  unsigned xoff[XL];
#pragma unroll
  for (int i = 0; i < XL; ++i) {
    const int m = m0 + prow + 8 * NW * i;
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
  unsigned xoff2[DUAL ? XL : 1];
  __amdgpu_buffer_rsrc_t xr2 = xr;
  if constexpr (DUAL) {
    xr2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x2), 0, (int)p.x2_bytes, 0x00020000);
#pragma unroll
    for (int i = 0; i < XL; ++i) {
      const int m = m0 + prow + 8 * NW * i;
      xoff2[i] = m < p.M ? (unsigned)(m * p.ldx2 + chunk * 4) * 4u : OOB;
    }
  }
  // (requesting the pixels two k-tiles ahead into a second register set, three workgroups per CU by launch bound, or eight
  //  wavefronts x 32 channels per block so that a pixel tile is split once for 256 channels: measured equal or slower - the
  //  kernel runs at the package power cap, profiles/r05_pw_split.txt)
  float4 xreg[XL];
  auto load_x = [&](int kt) {
    if constexpr (DUAL) {
      if (kt >= p.nk1) {                      // uniform: the second source's k-tiles
#pragma unroll
        for (int i = 0; i < XL; ++i)
          xreg[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(xr2, xoff2[i], (kt - p.nk1) * (SK * 4), 0));
        return;
      }
    }
#pragma unroll
    for (int i = 0; i < XL; ++i)
      xreg[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(xr, xoff[i], kt * (SK * 4), 0));
  };
  // plane q: [pixel][4 slots of 16 bytes = 8 k each], slot ^= (pixel / 4) % 4; this thread's 4 k are half (chunk & 1) of slot chunk >> 1
  auto store_x1 = [&](int i, int stage) {
    const int px = prow + 8 * NW * i;
    unsigned h[2], m[2], l[2];
    split3_pair(xreg[i].x, xreg[i].y, h[0], m[0], l[0]);
    split3_pair(xreg[i].z, xreg[i].w, h[1], m[1], l[1]);
    unsigned char* at = smem + stage * STAGE + px * 64 + (((chunk >> 1) ^ ((px >> 2) & 3)) * 16) + (chunk & 1) * 8;
    *reinterpret_cast<u32x2*>(at) = u32x2{h[0], h[1]};
    *reinterpret_cast<u32x2*>(at + PLANE) = u32x2{m[0], m[1]};
    *reinterpret_cast<u32x2*>(at + 2 * PLANE) = u32x2{l[0], l[1]};
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
  const unsigned char* vb = smem + vj * 64 + ((kg ^ ((vj >> 2) & 3)) * 16);
  const unsigned a_voff = (unsigned)lane * 16u;
  bf16x8 aq[2][2][3];                         // [k-tile parity][cb][piece]
  // packed U: [kt][32-channel group][cb][piece] chunks of 1 KiB (64 lanes x 8 bf16); this wavefront's group is tile_n NW + wv
  const int ngrp = p.Cout >> 5;
  auto load_a1 = [&](int j, int kt, int par) {         // j = 3 cb + piece
    const int base = ((kt * ngrp + tile_n * NW + wv) * 6 + j) * 1024;
    aq[par][j / 3][j % 3] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(ur, a_voff, base, 0));
  };
  auto load_a = [&](int kt, int par) {
#pragma unroll
    for (int j = 0; j < 6; ++j) load_a1(j, kt, par);
  };

#ifdef GLASS_PWS_STAMPS
  unsigned long long st[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = __builtin_amdgcn_s_memtime();
#endif
  load_x(0);
  load_a(0, 0);
  store_x(0);
  __syncthreads();
  PWS_STAMP(0)

  auto ktile = [&](int kt, auto par_) {
    constexpr int PAR = decltype(par_)::value;
    const int ktn = kt + 1 < p.nk ? kt + 1 : kt;     // clamped: a harmless re-read keeps the loop one block
    bf16x8 vq[2][3];
#pragma unroll
    for (int q = 0; q < 3; ++q) vq[0][q] = *reinterpret_cast<const bf16x8*>(vb + PAR * STAGE + q * PLANE);
    PWS_STAMP(1)
    static_for<PB>([&](auto g_) {                    // group = pixel block: 3 LDS reads, 2 NPROD MFMAs
      constexpr int pb = decltype(g_)::value;
      if constexpr (pb + 1 < PB) {
#pragma unroll
        for (int q = 0; q < 3; ++q)
          vq[(pb + 1) & 1][q] = *reinterpret_cast<const bf16x8*>(vb + PAR * STAGE + q * PLANE + (pb + 1) * 16 * 64);
      }
      // the memory counter retires in order: the weights of k-tile kt + 1 (needed at its first MFMA) go out BEFORE its pixels
      // (needed at the end of this k-tile), so that waiting for the former never waits for the latter
      if constexpr (pb == 0) {
        load_a(ktn, PAR ^ 1);
        load_x(ktn);
        __builtin_amdgcn_sched_barrier(0);
      }
      // smallest terms first; consecutive MFMAs alternate between the two channel blocks (independent accumulators)
      static_for<9>([&](auto t_) {
        constexpr int t = decltype(t_)::value;
        constexpr int qa = 2 - t / 3, qb = 2 - t % 3;       // (2,2) (2,1) (2,0) (1,2) (1,1) (1,0) (0,2) (0,1) (0,0)
        if constexpr (NPROD == 9 || qa + qb <= 2) {
#pragma unroll
          for (int cb = 0; cb < 2; ++cb)
            acc[pb][cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aq[PAR][cb][qa], vq[pb & 1][qb], acc[pb][cb], 0, 0, 0);
        }
      });
      if constexpr (pb == PB - XL - 1) PWS_STAMP(2)
      if constexpr (pb >= PB - XL) store_x1(pb - (PB - XL), PAR ^ 1);
    });
    PWS_STAMP(3)
    __syncthreads();
    PWS_STAMP(4)
  };
  for (int kt = 0; kt < p.nk; kt += 2) {
    ktile(kt, ic<0>{});
    if (kt + 1 < p.nk) ktile(kt + 1, ic<1>{});
  }

  // ---- epilogue (pointwise.hip's): lane = (pixel 16 pb + vj, channels n0 + 32 wv + 8 kg + 4 cb + e) ----
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
  auto epilogue = [&](auto res_c) {
    constexpr bool RES = decltype(res_c)::value != 0;
    unsigned yo[PB];
    f32x4 rq[RES ? PB : 1][2];
#pragma unroll
    for (int i = 0; i < PB; ++i) {
      const int m = m0 + 16 * i + vj;
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
    for (int i = 0; i < PB; ++i) {
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) {
        f32x4 v = acc[i][cb] + bv[cb];
        v.x = fmaxf(v.x, lo2); v.y = fmaxf(v.y, lo2); v.z = fmaxf(v.z, lo2); v.w = fmaxf(v.w, lo2);
        if constexpr (RES) v = v + rq[i][cb];
        v.x = fmaxf(v.x, lo1); v.y = fmaxf(v.y, lo1); v.z = fmaxf(v.z, lo1); v.w = fmaxf(v.w, lo1);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), yr, yo[i], cb * 16, 0);
      }
    }
  };
  if (p.res_mode != 0) epilogue(ic<1>{}); else epilogue(ic<0>{});
#ifdef GLASS_PWS_STAMPS
  __builtin_amdgcn_s_waitcnt(0);
  PWS_STAMP(5)
  if (p.dbg != nullptr && tid == 0 && (bid == nblk / 2 || bid == nblk / 2 + 8)) {
    unsigned long long* d = p.dbg + (bid == nblk / 2 ? 0 : 8);
    for (int k = 0; k < 8; ++k) d[k] = st[k];
  }
#endif
}

// W [Cout][1][1][Cin] fp32 -> bf16 [cin/32][cout/32][cb][piece][lane][8]  with
//   cout = 32 group + 8 ((lane&15)>>2) + 4 cb + (lane&3),   cin = 32 kt + 8 (lane>>4) + e
__global__ void pws_pack_weights_kernel(const float* __restrict__ w, unsigned short* __restrict__ u, int Cout, int Cin) {
  const long total = (long)Cout * Cin * 3;
  for (long o = (long)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (long)gridDim.x * blockDim.x) {
    long r = o;
    const int e = (int)(r & 7); r >>= 3;
    const int lane = (int)(r & 63); r >>= 6;
    const int q = (int)(r % 3); r /= 3;
    const int cb = (int)(r & 1); r >>= 1;
    const int ngrp = Cout >> 5;
    const int grp = (int)(r % ngrp);
    const int kt = (int)(r / ngrp);
    const int co = 32 * grp + 8 * ((lane & 15) >> 2) + 4 * cb + (lane & 3);
    const int ci = kt * SK + 8 * (lane >> 4) + e;
    unsigned h, m, l;
    split3_pair(w[(long)co * Cin + ci], 0.f, h, m, l);
    u[o] = (unsigned short)((q == 0 ? h : q == 1 ? m : l) & 0xffffu);
  }
}

}  // namespace

#ifdef GLASS_PWS_STAMPS
extern "C" void glass_pws_debug(void* p) { g_pws_dbg = (unsigned long long*)p; }
#endif

extern "C" int glass_pointwise_split_supported(const glass_conv_desc* d) {
  if (!d) return 0;
  const long M = (long)d->N * d->Ho * d->Wo;
  const long xb = (long)d->N * d->H * d->W * d->ldx * 4, yb = M * d->ldy * 4;
  const long rb = d->res_mode == 1 ? M * d->ldr * 4 : d->res_mode == 2 ? (long)d->N * (d->Ho / 2) * (d->Wo / 2) * d->ldr * 4 : 0;
  return d->KH == 1 && d->KW == 1 && d->pad_h == 0 && d->pad_w == 0 && d->stride_h == d->stride_w && d->stride_h >= 1 &&
         d->Cin % SK == 0 && d->Cout % SN == 0 && d->ldx % 4 == 0 && d->ldx >= d->Cin && d->y_cstride == 1 && d->ldy % 4 == 0 &&
         d->y_coff % 4 == 0 && d->y_coff >= 0 && d->y_coff + d->Cout <= d->ldy &&
         (d->res_mode == 0 || (d->ldr % 4 == 0 && d->ldr >= d->Cout)) && (d->res_mode != 2 || (d->Ho % 2 == 0 && d->Wo % 2 == 0)) &&
         d->Ho == (d->H - 1) / d->stride_h + 1 && d->Wo == (d->W - 1) / d->stride_w + 1 &&
         M < 0x7fffffffL && xb < 0x7fffff00L && yb < 0x7fffff00L && rb < 0x7fffff00L && (long)d->Cout * d->Cin * 6 < 0x7fffff00L;
}

extern "C" size_t glass_pointwise_split_weight_bytes(int Cout, int Cin) { return (size_t)Cout * (size_t)Cin * 6; }

extern "C" int glass_pointwise_split_pack_weights(const float* w, int Cout, int Cin, void* u_packed, glass_stream_t stream) {
  GLASS_CHECK_ARG(w && u_packed, "glass_pointwise_split_pack_weights: null pointer");
  GLASS_CHECK_ARG(Cout > 0 && Cin > 0 && Cout % SN == 0 && Cin % SK == 0,
                  "glass_pointwise_split_pack_weights: Cout=%d must be a multiple of 128 and Cin=%d a multiple of 32", Cout, Cin);
  const long total = (long)Cout * Cin * 3;
  const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(pws_pack_weights_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, (unsigned short*)u_packed, Cout, Cin);
  GLASS_CHECK_LAUNCH("glass_pointwise_split_pack_weights");
  return GLASS_OK;
}

namespace {
int pws_launch(const glass_conv_desc* d, const float* x, const float* x2, int Cin2, int ldx2, const void* u_packed, const float* bias,
               const float* residual, float* y, int products, glass_stream_t stream, const char* who);
}

// products: 9 = every pair of pieces (the exact product; what the model path uses); 6 = without the three terms < 2^-23 (measurement only)
extern "C" int glass_conv1x1_pointwise_split_nhwc(const glass_conv_desc* d, const float* x, const void* u_packed, const float* bias,
                                                  const float* residual, float* y, int products, glass_stream_t stream) {
  GLASS_CHECK_ARG(d && x && u_packed && y, "glass_conv1x1_pointwise_split_nhwc: null pointer");
  GLASS_CHECK_ARG(products == 9 || products == 6, "glass_conv1x1_pointwise_split_nhwc: products must be 9 or 6 (got %d)", products);
  GLASS_CHECK_ARG(glass_pointwise_split_supported(d),
                  "glass_conv1x1_pointwise_split_nhwc: needs 1x1 / pad 0 / square stride, Cin%%32==0, Cout%%128==0, unit channel stride, "
                  "operands < 2 GiB (got Cin=%d Cout=%d k=%dx%d s=%d p=%d)", d->Cin, d->Cout, d->KH, d->KW, d->stride_h, d->pad_h);
  GLASS_CHECK_ARG(d->res_mode == 0 || residual != nullptr, "glass_conv1x1_pointwise_split_nhwc: res_mode set but residual is null");
  GLASS_CHECK_ARG(((uintptr_t)x & 15) == 0 && ((uintptr_t)u_packed & 15) == 0 && ((uintptr_t)y & 15) == 0 &&
                      (bias == nullptr || ((uintptr_t)bias & 15) == 0) && (residual == nullptr || ((uintptr_t)residual & 15) == 0),
                  "glass_conv1x1_pointwise_split_nhwc: pointers must be 16-byte aligned");
  if (d->N == 0) return GLASS_OK;
  return pws_launch(d, x, nullptr, 0, 0, u_packed, bias, residual, y, products, stream, "glass_conv1x1_pointwise_split_nhwc");
}

namespace {
int pws_launch(const glass_conv_desc* d, const float* x, const float* x2, int Cin2, int ldx2, const void* u_packed, const float* bias,
               const float* residual, float* y, int products, glass_stream_t stream, const char* who) {
  PwsParams p;
  p.x = x; p.u = u_packed; p.bias = bias; p.res = residual; p.y = y;
  p.M = d->N * d->Ho * d->Wo; p.H = d->H; p.W = d->W; p.Ho = d->Ho; p.Wo = d->Wo; p.Cin = d->Cin; p.Cout = d->Cout;
  p.stride = d->stride_h; p.nk1 = d->Cin / SK; p.nk = (d->Cin + Cin2) / SK;
  p.x2 = x2; p.ldx2 = ldx2; p.x2_bytes = (unsigned)((long)p.M * ldx2 * 4);
  p.ldx = d->ldx; p.ldy = d->ldy; p.ycoff = d->y_coff; p.ldr = d->ldr; p.relu = d->relu; p.res_mode = d->res_mode;
  p.x_bytes = (unsigned)((long)d->N * d->H * d->W * d->ldx * 4);
  p.u_bytes = (unsigned)((long)d->Cout * (d->Cin + Cin2) * 6);
  p.y_bytes = (unsigned)((long)p.M * d->ldy * 4);
  p.r_bytes = d->res_mode == 1 ? (unsigned)((long)p.M * d->ldr * 4)
            : d->res_mode == 2 ? (unsigned)((long)d->N * (d->Ho / 2) * (d->Wo / 2) * d->ldr * 4) : 0u;
  p.magic_hw = (unsigned)(0x100000000ULL / (unsigned long long)(d->Ho * d->Wo));
  p.magic_w = (unsigned)(0x100000000ULL / (unsigned long long)d->Wo);
#ifdef GLASS_PWS_STAMPS
  p.dbg = g_pws_dbg;
#else
  p.dbg = nullptr;
#endif
  hipStream_t s = (hipStream_t)stream;
  p.tiles_n = d->Cout / SN;
  // 128-pixel blocks when they still give every CU ~2 workgroups, else 64-pixel blocks
  const bool big = (long)cdiv(p.M, 128) * p.tiles_n >= 512;
  p.tiles_m = cdiv(p.M, big ? 128 : 64);
  const long nblk = (long)p.tiles_m * p.tiles_n;
  GLASS_CHECK_ARG(nblk > 0 && nblk <= 0x7fffffffL, "%s: bad grid", who);
  const dim3 grid((unsigned)nblk), block(256);
  if (x2 != nullptr) {
    if (big) hipLaunchKernelGGL((conv1x1_pw_split<8, 9, true>), grid, block, 0, s, p);
    else hipLaunchKernelGGL((conv1x1_pw_split<4, 9, true>), grid, block, 0, s, p);
  } else if (products == 9) {
    if (big) hipLaunchKernelGGL((conv1x1_pw_split<8, 9>), grid, block, 0, s, p);
    else hipLaunchKernelGGL((conv1x1_pw_split<4, 9>), grid, block, 0, s, p);
  } else {
    if (big) hipLaunchKernelGGL((conv1x1_pw_split<8, 6>), grid, block, 0, s, p);
    else hipLaunchKernelGGL((conv1x1_pw_split<4, 6>), grid, block, 0, s, p);
  }
  GLASS_CHECK_LAUNCH(who);
  return GLASS_OK;
}
}  // namespace

// The shortcut of a bottleneck block folded into its conv3 (reference: detectron2 BottleneckBlock.forward behind
// glass/modeling/meta_arch/glass_rcnn.py:83, `out = conv3(out); out += shortcut(x); relu` [d2-recall]):
//   y = act([x1 strided | x2] . [W1 | W2]^T + bias)      x1 [N,H,W,ldx] with the desc's stride (the block input, Cin = d->Cin),
//   x2 [N,Ho,Wo,ldx2] (conv2's output, Cin2 channels), u_packed = glass_pointwise_split_pack_weights of the [Cout][Cin + Cin2]
//   concatenation, bias = the sum of the two folded biases.  One accumulator, nine exact bf16 piece products per element as in
//   glass_conv1x1_pointwise_split_nhwc: the [M][Cout] shortcut map is never written nor read back (res2: 2 x 537 MB at 8 images).
extern "C" int glass_pointwise_split_dual_supported(const glass_conv_desc* d, int Cin2, int ldx2) {
  if (!d || !glass_pointwise_split_supported(d)) return 0;
  const long M = (long)d->N * d->Ho * d->Wo;
  return d->res_mode == 0 && Cin2 > 0 && Cin2 % SK == 0 && ldx2 % 4 == 0 && ldx2 >= Cin2 && M * ldx2 * 4 < 0x7fffff00L &&
         (long)d->Cout * (d->Cin + Cin2) * 6 < 0x7fffff00L;
}

extern "C" int glass_conv1x1_pointwise_split_dual_nhwc(const glass_conv_desc* d, const float* x1, const float* x2, int Cin2, int ldx2,
                                                       const void* u_packed, const float* bias, float* y, glass_stream_t stream) {
  GLASS_CHECK_ARG(d && x1 && x2 && u_packed && y, "glass_conv1x1_pointwise_split_dual_nhwc: null pointer");
  GLASS_CHECK_ARG(glass_pointwise_split_dual_supported(d, Cin2, ldx2),
                  "glass_conv1x1_pointwise_split_dual_nhwc: needs what glass_conv1x1_pointwise_split_nhwc needs, res_mode 0, Cin2%%32==0, "
                  "ldx2%%4==0, ldx2>=Cin2 (got Cin=%d Cin2=%d ldx2=%d Cout=%d res_mode=%d)", d->Cin, Cin2, ldx2, d->Cout, d->res_mode);
  GLASS_CHECK_ARG(((uintptr_t)x1 & 15) == 0 && ((uintptr_t)x2 & 15) == 0 && ((uintptr_t)u_packed & 15) == 0 && ((uintptr_t)y & 15) == 0 &&
                      (bias == nullptr || ((uintptr_t)bias & 15) == 0),
                  "glass_conv1x1_pointwise_split_dual_nhwc: pointers must be 16-byte aligned");
  if (d->N == 0) return GLASS_OK;
  return pws_launch(d, x1, x2, Cin2, ldx2, u_packed, bias, nullptr, y, 9, stream, "glass_conv1x1_pointwise_split_dual_nhwc");
}
