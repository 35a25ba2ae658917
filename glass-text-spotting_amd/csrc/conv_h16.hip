// Convolution on the fp16 matrix cores of gfx950 for activations STORED as fp16 (BASELINE configs[4], conv precision
// "fp16s"): implicit GEMM  Y[m][co] = sum_{tap,ci} X[pix(m,tap)][ci] W[co][tap][ci]  with fp32 accumulation on
// v_mfma_f32_16x16x32_f16 (16x the fp32 matrix rate).
//
// glass_conv2d_nhwc_h16 (conv.hip) is the fp32 kernel template with fp16 operands: 32-channel k-tiles, both operands
// converted and staged through LDS, one barrier per 2 MFLOP - it sits at ~450 TF/s, 18 % of the fp16 peak.  This kernel is
// the structure of pointwise.hip (the fp32 weight-streaming 1x1 GEMM) widened to k x k taps and fp16:
//   * the weights are rounded to fp16 and laid out in MFMA A-fragment order ONCE (glass_conv_h16_pack_weights); each
//     wavefront streams its own fragments L2 -> registers with coalesced 1 KiB loads, a k-tile ahead.  No LDS, no
//     conversion, no barrier for them;
//   * a k-tile is 64 channels of ONE tap: 128 contiguous bytes per input pixel, fetched as bounds-checked 16-byte buffer
//     loads (zero padding = an out-of-range offset) and written to LDS unconverted (128-byte rows, XOR-swizzled 16-byte
//     slots -> conflict-free ds_read_b128 = one MFMA B operand of 32 k), double buffered, ONE barrier per 4-8 MFLOP;
//   * A = weights (16 output channels), B = pixels: a lane ends with 8 consecutive channels of its pixel (the packed
//     order interleaves the two channel blocks), so the epilogue is bias / ReLU / residual on registers and one 16-byte
//     store per pixel (fp16 out) - no LDS transposition;
//   * block = 256 pixels x 128 channels (4 wavefronts x 32 channels) or 256 x 64 (2 x 2 wavefronts of 128 pixels x 32
//     channels) or the 128- / 64-pixel fractions of those; <= 128 accumulator registers, TWO workgroups per CU.
// k-tile order is (channel block, tap) with the tap fastest, so the nine shifted reads of a channel block hit L2 / L1.
// Results: fp16?(act(sum over fp16(x) fp16(w) in fp32 + bias [+ residual])) - the same arithmetic as
// glass_conv2d_nhwc_h16 up to fp32 summation order (tests/test_gpu_f_ops.py compares the two and the fp64 reference).
#include "wino_common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

namespace {

constexpr int HK = 64;                        // input channels per k-tile (one tap)

struct H16Params {
  const void* x; const void* u; const float* bias; const void* res; void* y;
  int M, H, W, Ho, Wo, Cin, Cout, KH, KW, sh, sw, ph, pw, nk, ncc;
  int ldx, ldy, ycoff, ldr, relu, res_mode, yh, rh;
  int tiles_m, tiles_n;
  unsigned x_bytes, u_bytes, y_bytes, r_bytes;
  unsigned magic_hw, magic_w;                 // floor(2^32 / (Ho*Wo)), floor(2^32 / Wo)
  unsigned long long* dbg;                    // timing-instrumented builds only (GLASS_H16_ABL=6): per-workgroup cycle stamps
};

// PBW: 16-pixel blocks per wavefront; WM: wavefronts along the pixel dimension (4 / WM along the channels)
// ABL: timing ablations (GLASS_H16_ABL, wrong results; compiled only with -DGLASS_H16_ABLATIONS): 1 = no input loads after
// the first k-tile, 2 = neither input loads nor LDS stores, 3 = no weight loads, 4 = no LDS reads, 5 = no MFMAs,
// 6 = product code with s_memtime phase stamps
template <int PBW, int WM, int ABL = 0>
__global__ __launch_bounds__(256, 2) void conv_h16_kernel(H16Params p) {
  constexpr int NWN = 4 / WM;                 // wavefronts along the output channels
  constexpr int PN = 32 * NWN;                // output channels per block
  constexpr int PX = 16 * PBW * WM;           // pixels per block
  constexpr int XL = PX / 32;                 // 16-byte input loads per thread and k-tile
  constexpr int XS = PX * HK;                 // halves per LDS stage
  constexpr int VR = 2 * PBW < 6 ? 2 * PBW : 6; // B-operand ring (LDS reads in flight + 1)
  __shared__ __attribute__((aligned(16))) _Float16 smem[2 * XS];

  const int nblk = gridDim.x, bid = blockIdx.x;
  const int xcd = bid & 7, q8 = nblk >> 3, r8 = nblk & 7;
  const int logical = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  const int tile_m = logical / p.tiles_n;
  const int tile_n = logical - tile_m * p.tiles_n;
  const int m0 = tile_m * PX, n0 = tile_n * PN;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wv % NWN, wm = wv / NWN;
  const int HoWo = p.Ho * p.Wo;
  const int T = p.KH * p.KW;

  __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.x), 0, (int)p.x_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t ur = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.u), 0, (int)p.u_bytes, 0x00020000);

  // ---- input role: thread = (pixel tid>>3 + 32 i, 16-byte chunk tid&7 of the 64-channel k-tile) ----
  // xb: byte offset of the pixel under tap (0,0) (may lie before the tensor: padding); vm: bit t set = tap t is inside
  const int chunk = tid & 7, prow = tid >> 3;
  int xb[XL];
  unsigned vm[XL];
#pragma unroll
  for (int i = 0; i < XL; ++i) {
    const int m = m0 + prow + 32 * i;
    xb[i] = 0; vm[i] = 0u;
    if (m < p.M) {
      const int n = fast_div(m, HoWo, p.magic_hw);
      const int rem = m - n * HoWo;
      const int ho = fast_div(rem, p.Wo, p.magic_w);
      const int wo = rem - ho * p.Wo;
      const int hi0 = ho * p.sh - p.ph, wi0 = wo * p.sw - p.pw;
      xb[i] = (((n * p.H + hi0) * p.W + wi0) * p.ldx + chunk * 8) * 2;
      unsigned cm = 0u, msk = 0u;
#pragma clang loop vectorize(disable) unroll(disable)
      for (int kx = 0; kx < p.KW; ++kx) cm |= ((unsigned)(wi0 + kx) < (unsigned)p.W ? 1u : 0u) << kx;
#pragma clang loop vectorize(disable) unroll(disable)
      for (int ky = 0; ky < p.KH; ++ky) msk |= ((unsigned)(hi0 + ky) < (unsigned)p.H ? cm : 0u) << (ky * p.KW);
      vm[i] = msk;
    }
  }
  u32x4 xreg[XL];
  // k-tile kt = cc * T + tap: channels [64 cc, 64 cc + 64) of tap (ky, kx); tdelta = ((ky W + kx) ldx) * 2 bytes
  auto load_x1 = [&](int i, int cc, int tap, int tdelta) {
    const unsigned off = ((vm[i] >> tap) & 1u) ? (unsigned)(xb[i] + tdelta) : OOB;
    xreg[i] = __builtin_amdgcn_raw_buffer_load_b128(xr, off, cc * (HK * 2), 0);
  };
  auto load_x = [&](int cc, int tap, int tdelta) {
#pragma unroll
    for (int i = 0; i < XL; ++i) load_x1(i, cc, tap, tdelta);
  };
  // X[stage][pixel][64 halves]: 128-byte rows; slot ^= (pixel/2)%8 (see winograd43.hip)
  auto store_x1 = [&](int i, int stage) {
    const int px = prow + 32 * i;
    *reinterpret_cast<u32x4*>(smem + stage * XS + px * HK + ((chunk ^ ((px >> 1) & 7)) * 8)) = xreg[i];
  };
  auto store_x = [&](int stage) {
#pragma unroll
    for (int i = 0; i < XL; ++i) store_x1(i, stage);
  };

  // ---- MFMA role: wave (wm, wn) owns channels n0 + 32 wn + [0, 32) of pixels 16 PBW wm + [0, 16 PBW) ----
  f32x4 acc[PBW][2];
#pragma unroll
  for (int pb = 0; pb < PBW; ++pb)
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) acc[pb][cb] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int vj = lane & 15, kg = lane >> 4;
  const int vswz = (vj >> 1) & 7;
  const int vrow = (16 * PBW * wm + vj) * HK;
  const _Float16* vb[2] = {smem + vrow + ((0 * 4 + kg) ^ vswz) * 8, smem + vrow + ((1 * 4 + kg) ^ vswz) * 8};
  const unsigned a_voff = (unsigned)lane * 16u;
  h8 aq[2][2][2];                             // [k-tile parity][half][cb]
  // packed U: [tile_n][kt][wn][half][cb] chunks of 1 KiB (64 lanes x 8 halves)
  auto load_a1 = [&](int j, int kt, int par) {         // j = 2 half + cb
    const int base = (((tile_n * p.nk + kt) * NWN + wn) * 4 + j) * 1024;
    aq[par][j >> 1][j & 1] = __builtin_bit_cast(h8, __builtin_amdgcn_raw_buffer_load_b128(ur, a_voff, base, 0));
  };
  auto load_a = [&](int kt, int par) {
#pragma unroll
    for (int j = 0; j < 4; ++j) load_a1(j, kt, par);
  };

  // scalar state of the NEXT k-tile to fetch
  int n_kt = 0, n_cc = 0, n_tap = 0, n_kx = 0, n_td = 0;
  auto advance = [&]() {
    if (n_kt + 1 < p.nk) {                          // clamped at the last tile: a harmless re-read keeps the loop one block
      ++n_kt; ++n_tap; ++n_kx; n_td += p.ldx * 2;
      if (n_kx == p.KW) { n_kx = 0; n_td += (p.W - p.KW) * p.ldx * 2; }
      if (n_tap == T) { n_tap = 0; n_kx = 0; n_td = 0; ++n_cc; }
    }
  };

  unsigned long long st_begin = 0, st_loop = 0, st_end = 0, c_mfma = 0, c_store = 0, c_bar = 0;
  if constexpr (ABL == 6) st_begin = __builtin_amdgcn_s_memtime();
  load_x(0, 0, 0);
  load_a(0, 0);
  store_x(0);
  __syncthreads();
  if constexpr (ABL == 6) st_loop = __builtin_amdgcn_s_memtime();

  auto ktile = [&](auto par_) {
    constexpr int PAR = decltype(par_)::value;
    unsigned long long t0 = 0, t1 = 0, t2 = 0;
    if constexpr (ABL == 6) t0 = __builtin_amdgcn_s_memtime();
    advance();
    // The issue order is pinned (sched_barrier): left alone the compiler sinks the global loads to the end of the k-tile
    // and waits for every LDS read right after issuing it.  B operands: a ring of VR LDS reads - one ds_read_b128 feeds
    // only 2 MFMAs (34 cycles) against ~130 cycles of LDS latency, so the reads run VR - 1 groups ahead; the 4 weight and
    // XL input loads of the next k-tile go out with the first groups (a burst fills the CU's address queue and stalls the
    // MFMAs) and are written to LDS with the last groups, so that only the barrier is left between two k-tiles.
    h8 vq[VR];
    static_for<VR - 1>([&](auto g_) {
      constexpr int g = decltype(g_)::value;
      vq[g] = *reinterpret_cast<const h8*>(vb[g / PBW] + PAR * XS + (g % PBW) * 16 * HK);
    });
    static_for<2 * PBW>([&](auto g_) {              // group g = (half, pixel block): one LDS read, 2 MFMAs of 32 k
      constexpr int g = decltype(g_)::value;
      constexpr int h = g / PBW, pb = g % PBW;
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (g + VR - 1 < 2 * PBW && ABL != 4) {
        constexpr int h1 = (g + VR - 1) / PBW, pb1 = (g + VR - 1) % PBW;
        vq[(g + VR - 1) % VR] = *reinterpret_cast<const h8*>(vb[h1] + PAR * XS + pb1 * 16 * HK);
      }
      if constexpr (g < XL && ABL != 1 && ABL != 2) load_x1(g, n_cc, n_tap, n_td);
      if constexpr (g < 4 && ABL != 3) load_a1(g, n_kt, PAR ^ 1);
      // the tile lands in the other LDS stage (last read in the previous k-tile, before its barrier) under the last MFMAs
      if constexpr (g >= 2 * PBW - XL && ABL != 2) store_x1(g - (2 * PBW - XL), PAR ^ 1);
      if constexpr (ABL == 5) {
        acc[pb][0].x += (float)vq[g % VR][0] + (float)aq[PAR][h][0][0] + (float)aq[PAR][h][1][0];
      } else {
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
          acc[pb][cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(aq[PAR][h][cb], vq[g % VR], acc[pb][cb], 0, 0, 0);
      }
    });
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (ABL == 6) { t1 = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); }
    if constexpr (ABL == 6) { __builtin_amdgcn_sched_barrier(0); t2 = __builtin_amdgcn_s_memtime(); }
    __syncthreads();
    if constexpr (ABL == 6) {
      const unsigned long long t3 = __builtin_amdgcn_s_memtime();
      c_mfma += t1 - t0; c_store += t2 - t1; c_bar += t3 - t2;
    }
  };
  for (int kt = 0; kt < p.nk; kt += 2) {
    ktile(ic<0>{});
    if (kt + 1 < p.nk) ktile(ic<1>{});
  }

  if constexpr (ABL == 6) st_end = __builtin_amdgcn_s_memtime();
  // ---- epilogue: lane = (pixel 16 (PBW wm + pb) + vj, channels n0 + 32 wn + 8 kg + 4 cb + e) ----
  __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, (int)p.y_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.res_mode != 0 ? p.res : p.y), 0,
                                                                 (int)(p.res_mode != 0 ? p.r_bytes : 0u), 0x00020000);
  const int cbase = n0 + 32 * wn + 8 * kg;
  const unsigned yes = p.yh ? 2u : 4u, res_ = p.rh ? 2u : 4u;
  const unsigned ldyb = (unsigned)p.ldy * yes, ldrb = (unsigned)p.ldr * res_;
  const float lo2 = p.relu == 2 ? 0.f : __builtin_nanf(""), lo1 = p.relu == 1 ? 0.f : __builtin_nanf("");
  f32x4 bv[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
  if (p.bias != nullptr) {
    bv[0] = *reinterpret_cast<const f32x4*>(p.bias + cbase);
    bv[1] = *reinterpret_cast<const f32x4*>(p.bias + cbase + 4);
  }
  const int HoWo2 = (p.Ho >> 1) * (p.Wo >> 1);
  // RES: 0 none, 1 fp16 residual, 2 fp32 residual; YH: fp16 output.  Compile-time variants, and the residual loads of a
  // batch of pixel blocks all go out before its first store: a load issued after a store waits for it (one in-order
  // counter), which serialised the 16 blocks of a wavefront at ~2000 cycles each.
  auto epilogue = [&](auto res_c, auto yh_c) {
    constexpr int RES = decltype(res_c)::value;
    constexpr bool YH = decltype(yh_c)::value != 0;
    constexpr int PBB = PBW < 8 ? PBW : 8;          // pixel blocks per batch
#pragma unroll
    for (int b0 = 0; b0 < PBW; b0 += PBB) {
      unsigned yo[PBB];
      u32x4 rq[RES == 0 ? 1 : PBB][RES == 2 ? 2 : 1];
#pragma unroll
      for (int i = 0; i < PBB; ++i) {
        const int m = m0 + 16 * (PBW * wm + b0 + i) + vj;
        const bool ok = m < p.M;
        yo[i] = ok ? (unsigned)m * ldyb + (unsigned)(p.ycoff + cbase) * yes : OOB;
        if constexpr (RES != 0) {
          unsigned ro = OOB;
          if (ok) {
            int rp = m;
            if (p.res_mode == 2) {                  // x2 nearest-upsampled residual [N, Ho/2, Wo/2, ldr]
              const int n = fast_div(m, HoWo, p.magic_hw);
              const int rem = m - n * HoWo;
              const int ho = fast_div(rem, p.Wo, p.magic_w);
              const int wo = rem - ho * p.Wo;
              rp = n * HoWo2 + (ho >> 1) * (p.Wo >> 1) + (wo >> 1);
            }
            ro = (unsigned)rp * ldrb + (unsigned)cbase * res_;
          }
          rq[i][0] = __builtin_amdgcn_raw_buffer_load_b128(rr, ro, 0, 0);
          if constexpr (RES == 2) rq[i][1] = __builtin_amdgcn_raw_buffer_load_b128(rr, ro, 16, 0);
        }
      }
#pragma unroll
      for (int i = 0; i < PBB; ++i) {
        f32x4 r[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
        if constexpr (RES == 1) {
          const h8 rv = __builtin_bit_cast(h8, rq[i][0]);
          r[0] = f32x4{(float)rv[0], (float)rv[1], (float)rv[2], (float)rv[3]};
          r[1] = f32x4{(float)rv[4], (float)rv[5], (float)rv[6], (float)rv[7]};
        } else if constexpr (RES == 2) {
          r[0] = __builtin_bit_cast(f32x4, rq[i][0]);
          r[1] = __builtin_bit_cast(f32x4, rq[i][1]);
        }
        f32x4 v[2];
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
          f32x4 t = acc[b0 + i][cb] + bv[cb];
          t.x = fmaxf(t.x, lo2); t.y = fmaxf(t.y, lo2); t.z = fmaxf(t.z, lo2); t.w = fmaxf(t.w, lo2);
          if constexpr (RES != 0) t = t + r[cb];
          t.x = fmaxf(t.x, lo1); t.y = fmaxf(t.y, lo1); t.z = fmaxf(t.z, lo1); t.w = fmaxf(t.w, lo1);
          v[cb] = t;
        }
        if constexpr (YH) {
          const h8 o = h8{(_Float16)v[0].x, (_Float16)v[0].y, (_Float16)v[0].z, (_Float16)v[0].w,
                          (_Float16)v[1].x, (_Float16)v[1].y, (_Float16)v[1].z, (_Float16)v[1].w};
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), yr, yo[i], 0, 0);
        } else {
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v[0]), yr, yo[i], 0, 0);
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v[1]), yr, yo[i], 16, 0);
        }
      }
    }
  };
  if (p.res_mode == 0) {
    if (p.yh) epilogue(ic<0>{}, ic<1>{}); else epilogue(ic<0>{}, ic<0>{});
  } else if (p.rh) {
    if (p.yh) epilogue(ic<1>{}, ic<1>{}); else epilogue(ic<1>{}, ic<0>{});
  } else {
    if (p.yh) epilogue(ic<2>{}, ic<1>{}); else epilogue(ic<2>{}, ic<0>{});
  }
  if constexpr (ABL == 6) {
    if (p.dbg != nullptr && tid == 0) {
      unsigned long long* o = p.dbg + (long)blockIdx.x * 8;
      o[0] = st_begin; o[1] = st_loop; o[2] = st_end; o[3] = __builtin_amdgcn_s_memtime();
      o[4] = c_mfma; o[5] = c_store; o[6] = c_bar;
    }
  }
}

// W [Cout][KH][KW][Cin] fp32 -> fp16 (round to nearest even), [cout/PN][kt][wn][half][cb][lane][s] with
//   kt = (cin/64) * KH*KW + tap,   cin = 64 (kt / T) + 32 half + 8 (lane>>4) + s,
//   cout = PN tn + 32 wn + 8 ((lane&15)>>2) + 4 cb + (lane&3)      (A row lane&15 of channel block cb)
__global__ void h16_pack_weights_kernel(const float* __restrict__ w, _Float16* __restrict__ u, int Cout, int T, int Cin, int PN) {
  const long total = (long)Cout * T * Cin;
  const int nk = (Cin / HK) * T, nwn = PN / 32;
  for (long o = (long)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (long)gridDim.x * blockDim.x) {
    long r = o;
    const int s = (int)(r & 7); r >>= 3;
    const int lane = (int)(r & 63); r >>= 6;
    const int cb = (int)(r & 1); r >>= 1;
    const int half = (int)(r & 1); r >>= 1;
    const int wn = (int)(r % nwn); r /= nwn;
    const int kt = (int)(r % nk);
    const int tn = (int)(r / nk);
    const int cc = kt / T, tap = kt - cc * T;
    const int co = tn * PN + 32 * wn + 8 * ((lane & 15) >> 2) + 4 * cb + (lane & 3);
    const int ci = cc * HK + 32 * half + 8 * (lane >> 4) + s;
    u[o] = (_Float16)w[((long)co * T + tap) * Cin + ci];
  }
}

int h16_block_channels(int Cout) { return Cout % 128 == 0 ? 128 : 64; }

}  // namespace

extern "C" int glass_conv_h16_supported(const glass_conv_desc* d, int flags) {
  if (!d || (flags & ~7) != 0 || !(flags & 1)) return 0;          // the input must be an fp16 tensor
  const long yes = (flags & 2) ? 2 : 4, res_ = (flags & 4) ? 2 : 4;
  const long M = (long)d->N * d->Ho * d->Wo;
  const long xb = (long)d->N * d->H * d->W * d->ldx * 2, yb = M * d->ldy * yes;
  const long rb = d->res_mode == 1 ? M * d->ldr * res_ : d->res_mode == 2 ? (long)d->N * (d->Ho / 2) * (d->Wo / 2) * d->ldr * res_ : 0;
  const int ya = (flags & 2) ? 8 : 4, ra = (flags & 4) ? 8 : 4;   // 16-byte granules of the output / residual rows
  return d->KH >= 1 && d->KW >= 1 && d->KH * d->KW <= 32 && d->stride_h >= 1 && d->stride_w >= 1 && d->pad_h >= 0 && d->pad_w >= 0 &&
         d->Cin % HK == 0 && d->Cout % 64 == 0 && d->ldx % 8 == 0 && d->ldx >= d->Cin && d->y_cstride == 1 && d->ldy % ya == 0 &&
         d->y_coff % ya == 0 && d->y_coff >= 0 && d->y_coff + d->Cout <= d->ldy &&
         (d->res_mode == 0 || (d->ldr % ra == 0 && d->ldr >= d->Cout)) && (d->res_mode != 2 || (d->Ho % 2 == 0 && d->Wo % 2 == 0)) &&
         d->Ho == (d->H + 2 * d->pad_h - d->KH) / d->stride_h + 1 && d->Wo == (d->W + 2 * d->pad_w - d->KW) / d->stride_w + 1 &&
         d->Ho > 0 && d->Wo > 0 && M < 0x7fffffffL && xb < 0x7fffff00L && yb < 0x7fffff00L && rb < 0x7fffff00L &&
         (long)d->Cout * d->KH * d->KW * d->Cin * 2 < 0x7fffff00L;
}

extern "C" size_t glass_conv_h16_weight_halves(int Cout, int KH, int KW, int Cin) {
  return (size_t)Cout * (size_t)KH * (size_t)KW * (size_t)Cin;
}

extern "C" int glass_conv_h16_pack_weights(const float* w, int Cout, int KH, int KW, int Cin, void* u_packed, glass_stream_t stream) {
  GLASS_CHECK_ARG(w && u_packed, "glass_conv_h16_pack_weights: null pointer");
  GLASS_CHECK_ARG(Cout > 0 && Cin > 0 && KH > 0 && KW > 0 && Cout % 64 == 0 && Cin % HK == 0,
                  "glass_conv_h16_pack_weights: Cout=%d and Cin=%d must be multiples of 64", Cout, Cin);
  const long total = (long)Cout * KH * KW * Cin;
  const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(h16_pack_weights_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, static_cast<_Float16*>(u_packed),
                     Cout, KH * KW, Cin, h16_block_channels(Cout));
  GLASS_CHECK_LAUNCH("glass_conv_h16_pack_weights");
  return GLASS_OK;
}

extern "C" int glass_conv2d_nhwc_h16_packed(const glass_conv_desc* d, const void* x, const void* u_packed, const float* bias,
                                            const void* residual, void* y, int flags, glass_stream_t stream) {
  GLASS_CHECK_ARG(d && x && u_packed && y, "glass_conv2d_nhwc_h16_packed: null pointer");
  GLASS_CHECK_ARG(glass_conv_h16_supported(d, flags),
                  "glass_conv2d_nhwc_h16_packed: needs an fp16 input (flags bit 0), Cin%%64==0, Cout%%64==0, <= 32 taps, unit channel "
                  "stride, 16-byte aligned rows, operands < 2 GiB (got Cin=%d Cout=%d k=%dx%d flags=0x%x)",
                  d->Cin, d->Cout, d->KH, d->KW, flags);
  GLASS_CHECK_ARG(d->res_mode == 0 || residual != nullptr, "glass_conv2d_nhwc_h16_packed: res_mode set but residual is null");
  GLASS_CHECK_ARG(((uintptr_t)x & 15) == 0 && ((uintptr_t)u_packed & 15) == 0 && ((uintptr_t)y & 15) == 0 &&
                      (bias == nullptr || ((uintptr_t)bias & 15) == 0) && (residual == nullptr || ((uintptr_t)residual & 15) == 0),
                  "glass_conv2d_nhwc_h16_packed: pointers must be 16-byte aligned");
  if (d->N == 0) return GLASS_OK;
  H16Params p;
  p.x = x; p.u = u_packed; p.bias = bias; p.res = residual; p.y = y; p.dbg = nullptr;
  p.M = d->N * d->Ho * d->Wo; p.H = d->H; p.W = d->W; p.Ho = d->Ho; p.Wo = d->Wo; p.Cin = d->Cin; p.Cout = d->Cout;
  p.KH = d->KH; p.KW = d->KW; p.sh = d->stride_h; p.sw = d->stride_w; p.ph = d->pad_h; p.pw = d->pad_w;
  p.ncc = d->Cin / HK; p.nk = p.ncc * d->KH * d->KW;
  p.ldx = d->ldx; p.ldy = d->ldy; p.ycoff = d->y_coff; p.ldr = d->ldr; p.relu = d->relu; p.res_mode = d->res_mode;
  p.yh = (flags >> 1) & 1; p.rh = (flags >> 2) & 1;
  const long yes = p.yh ? 2 : 4, res_ = p.rh ? 2 : 4;
  p.x_bytes = (unsigned)((long)d->N * d->H * d->W * d->ldx * 2);
  p.u_bytes = (unsigned)((long)d->Cout * d->KH * d->KW * d->Cin * 2);
  p.y_bytes = (unsigned)((long)p.M * d->ldy * yes);
  p.r_bytes = d->res_mode == 1 ? (unsigned)((long)p.M * d->ldr * res_)
            : d->res_mode == 2 ? (unsigned)((long)d->N * (d->Ho / 2) * (d->Wo / 2) * d->ldr * res_) : 0u;
  p.magic_hw = (unsigned)(0x100000000ULL / (unsigned long long)(d->Ho * d->Wo));
  p.magic_w = (unsigned)(0x100000000ULL / (unsigned long long)d->Wo);
  const int PN = h16_block_channels(d->Cout);
  p.tiles_n = d->Cout / PN;
  // 256-pixel blocks when they still give every CU ~1.5 workgroups, else 128-pixel blocks, 64-pixel ones when even those
  // leave a quarter of the CUs idle (the linear layers: fc1 / fc2 at M = 800)
  const bool big = (long)cdiv(p.M, 256) * p.tiles_n >= 384;
  const bool tiny = !big && (long)cdiv(p.M, 128) * p.tiles_n < 192;
  p.tiles_m = cdiv(p.M, big ? 256 : tiny ? 64 : 128);
  const long nblk = (long)p.tiles_m * p.tiles_n;
  GLASS_CHECK_ARG(nblk > 0 && nblk <= 0x7fffffffL, "glass_conv2d_nhwc_h16_packed: bad grid");
  const dim3 grid((unsigned)nblk), block(256);
#ifdef GLASS_H16_ABLATIONS
  static const int abl = getenv("GLASS_H16_ABL") ? atoi(getenv("GLASS_H16_ABL")) : 0;      // timing ablations (wrong results)
  if (abl >= 1 && abl <= 6 && PN == 128 && big) {
    static unsigned long long* dbg_dev = nullptr;
    if (abl == 6) {
      if (!dbg_dev) (void)hipMalloc(&dbg_dev, 8L * 8 * 65536);
      p.dbg = nblk <= 65536 ? dbg_dev : nullptr;
    }
    switch (abl) {
      case 1: hipLaunchKernelGGL((conv_h16_kernel<16, 1, 1>), grid, block, 0, (hipStream_t)stream, p); break;
      case 2: hipLaunchKernelGGL((conv_h16_kernel<16, 1, 2>), grid, block, 0, (hipStream_t)stream, p); break;
      case 3: hipLaunchKernelGGL((conv_h16_kernel<16, 1, 3>), grid, block, 0, (hipStream_t)stream, p); break;
      case 4: hipLaunchKernelGGL((conv_h16_kernel<16, 1, 4>), grid, block, 0, (hipStream_t)stream, p); break;
      case 5: hipLaunchKernelGGL((conv_h16_kernel<16, 1, 5>), grid, block, 0, (hipStream_t)stream, p); break;
      default: hipLaunchKernelGGL((conv_h16_kernel<16, 1, 6>), grid, block, 0, (hipStream_t)stream, p); break;
    }
    GLASS_CHECK_LAUNCH("glass_conv2d_nhwc_h16_packed");
    if (abl == 6 && p.dbg) {      // instrumented build: phase times of the workgroups (drains the stream)
      static int printed = 0;
      if (printed++ < 3) {
        (void)hipStreamSynchronize((hipStream_t)stream);
        unsigned long long* h = (unsigned long long*)malloc(nblk * 64);
        (void)hipMemcpy(h, dbg_dev, nblk * 64, hipMemcpyDeviceToHost);
        double pro = 0, loop = 0, epi = 0, m = 0, st = 0, bar = 0;
        unsigned long long t_min = ~0ULL, t_max = 0;
        for (long b = 0; b < nblk; ++b) {
          pro += (double)(h[8 * b + 1] - h[8 * b]); loop += (double)(h[8 * b + 2] - h[8 * b + 1]); epi += (double)(h[8 * b + 3] - h[8 * b + 2]);
          m += (double)h[8 * b + 4]; st += (double)h[8 * b + 5]; bar += (double)h[8 * b + 6];
          if (h[8 * b] < t_min) t_min = h[8 * b];
          if (h[8 * b + 3] > t_max) t_max = h[8 * b + 3];
        }
        fprintf(stderr, "[h16 dbg] blocks %ld nk %d: mean ticks prologue %.0f  k-loop %.0f (%.0f / k-tile: mfma phase %.0f, store %.0f, barrier %.0f)  "
                "epilogue %.0f | kernel span %llu (s_memtime ticks)\n", nblk, p.nk, pro / nblk, loop / nblk, loop / nblk / p.nk,
                m / nblk / p.nk, st / nblk / p.nk, bar / nblk / p.nk, epi / nblk, t_max - t_min);
        free(h);
      }
    }
    return GLASS_OK;
  }
#endif
  if (PN == 128) {
    if (big) hipLaunchKernelGGL((conv_h16_kernel<16, 1>), grid, block, 0, (hipStream_t)stream, p);
    else if (!tiny) hipLaunchKernelGGL((conv_h16_kernel<8, 1>), grid, block, 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL((conv_h16_kernel<4, 1>), grid, block, 0, (hipStream_t)stream, p);
  } else {
    if (big) hipLaunchKernelGGL((conv_h16_kernel<8, 2>), grid, block, 0, (hipStream_t)stream, p);
    else if (!tiny) hipLaunchKernelGGL((conv_h16_kernel<4, 2>), grid, block, 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL((conv_h16_kernel<2, 2>), grid, block, 0, (hipStream_t)stream, p);
  }
  GLASS_CHECK_LAUNCH("glass_conv2d_nhwc_h16_packed");
  return GLASS_OK;
}
