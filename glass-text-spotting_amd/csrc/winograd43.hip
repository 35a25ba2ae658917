// Winograd F(4x4, 3x3) convolution on the fp32 matrix cores of gfx950 (3x3, stride 1, pad 1, NHWC fp32).
//
// F(2x2,3x3) (winograd.hip) issues 16 multiplies per 2x2 output tile = 4 per output pixel and channel pair; F(4x4,3x3)
// issues 36 per 4x4 tile = 2.25 (1.78x fewer again, 4x fewer than the direct convolution), reads 36 instead of 64
// input values per 16 outputs and streams 2.25x the weight bytes.  fp32 accuracy is the price of the larger
// transform: with the usual points (0, +-1, +-2) the error is 1.3e-5 of the output range on a 256-channel layer.
// Rounds 2-5 used the asymmetric points (0, 1, -1, 1/2, -2, inf): 4.8e-6, at the cost of a transform without the
// even / odd symmetry (16 operations per 6-point transform).  Round 6: the points (0, +-3/4, +-3/2, inf) - the best pair of a
// scan over +-a, +-b in eighths (fp32 emulation of this pipeline against fp64, 256 channels: 2.9e-6; +-1, +-2 is the same
// set scaled by 4/3 and 4.5x worse) - are symmetric, so a transform is 12 fused multiply-adds (even part + / - odd part) AND
// more accurate; every constant is dyadic, i.e. exact in fp32 (tests/test_gpu_f_ops.py holds the kernel to 2e-5 of the range).
//
//   Y(4x4) = At [ sum_c (G g G^t) . (Bt d B) ] A,   36 independent GEMMs  M_xi[cout][tile] = sum_c U_xi[cout][c] V_xi[c][tile]
//
// Work decomposition (256 threads = 4 waves, ONE workgroup per CU):
//   * block = 16 output tiles (4x4 pixels each = 256 output pixels) x 128 output channels x all 36 xi, k-tile = 32
//     input channels.  Wave w owns the 32 output channels 32w..32w+31 for ALL 36 xi: 36 x 2 MFMA blocks of
//     v_mfma_f32_16x16x4_f32 (rows = 16 channels, columns = the 16 tiles) = 288 accumulator registers.  Because a
//     lane then holds all 36 xi of its (tile, 4-channel) outputs, the output transform At . A runs entirely in
//     registers - no LDS exchange, no barrier in the epilogue - and ends in row stores (64 contiguous bytes per pixel and
//     store instruction).
//   * per k-tile: thread (tile, channel pair) fetches its raw 6x6 patch with 36 bounds-checked buffer_load_dwordx2
//     (offset = row part + column part, an invalid part is 2^30 so that the sum is out of range: padding and ragged
//     tiles are the hardware's zero fill; 12 offset registers instead of 36), applies Bt d B in place (12 six-point
//     transforms of 12 fma per channel) and writes the 36 transformed pairs to LDS V[xi][tile][32], XOR-swizzled
//     so that the MFMA-side ds_read_b128 (one per 8 MFMAs, feeding 4 k-steps x 2 channel blocks) is conflict free.
//   * weights never touch LDS: U = G g G^t is pre-packed (glass_winograd43_pack_weights) in MFMA A-fragment order,
//     each wave streams its own 1 KiB fragments L2 -> registers through a 6-slot ring, five groups ahead.
//   * 576 MFMAs (18.4 K cycles) per wave and k-tile against 144 weight loads + 72 LDS reads + 36 patch loads + 36 LDS
//     writes + ~300 VALU (round 6: 1367 instructions per k-tile, 1656 in round 5), all metered between the MFMAs (the source order IS the issue order, pinned with
//     sched_barrier like the F(2x2) kernels); V is double buffered, one barrier per k-tile.
#include "wino_common.h"
#include "splitk_common.h"
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

namespace {

// Two shapes of the same kernel (template parameters WT x WC = how the 4 waves split the block, KT = k-tile):
//   wide   WT = 1, WC = 4, KT = 32:  16 tiles x 128 channels  (Cout % 128 == 0, Cin % 32 == 0: the 128..512-channel layers)
//   narrow WT = 2, WC = 2, KT = 16:  32 tiles x  64 channels  (Cout %  64 == 0, Cin % 16 == 0: the 64-channel layers; two
//          waves share every weight fragment through the L1, V keeps its 72 KiB per stage with half the channels)
constexpr int WINO43_LDS_BYTES = 2 * 36 * 16 * 32 * 4;      // V: 2 stages x 36 xi x (16 WT tiles) x KT channels = 144 KiB for both shapes
#ifndef GLASS_W43_RING
#define GLASS_W43_RING 6       // 4 (three groups = ~770 cycles ahead) until round 6: under a full grid the L2 answers later (k-tile 22.1 K cycles at 32 workgroups, 23.1 K at 2048); 6 is +0.6 - 1.3 % end to end on two boxes, 9 spills and loses (profiles/r06_ab_w43_ring.txt)
#endif
constexpr int RING = GLASS_W43_RING;          // weight-fragment ring slots (groups in flight = RING - 1)
#ifndef GLASS_W43_VACC_XI
#define GLASS_W43_VACC_XI 32
#endif
#ifndef GLASS_W43_STORE_AUX
#define GLASS_W43_STORE_AUX 0
#endif
constexpr int W43_STORE_AUX = GLASS_W43_STORE_AUX;   // cache policy of the output stores (0 = default; 2 = nt: experiment, profiles/r06)
constexpr int VACC_XI = GLASS_W43_VACC_XI;    // xi >= VACC_XI accumulate in VGPRs (inline-asm MFMA); 36 = none (the round-5 code)
constexpr unsigned INV = 0x40000000u;         // "invalid" part of a split offset: any sum containing it is >= 1 GiB

__device__ __forceinline__ float comp4(const f32x4& v, int s) { return s == 0 ? v.x : s == 1 ? v.y : s == 2 ? v.z : v.w; }

// Cook-Toom F(4,3) on the points p = (0, a, -a, b, -b, inf), a = 3/4, b = 3/2.  M(x) = x (x^2 - a^2)(x^2 - b^2);
//   Bt row of p  = coefficients of M(x) / (x - p) (monic quartic; row inf = M itself),
//   G  row of p  = (1, p, p^2) / prod_{q != p} (p - q)   (row inf = (0, 0, 1)),
//   At[k][p]     = p^k, k = 0..3                          (column inf = (0, 0, 0, 1)).
constexpr float WA = 0.75f, WB = 1.5f;
constexpr float WA2 = WA * WA, WB2 = WB * WB, WS2 = WA2 + WB2, WP2 = WA2 * WB2, WA3 = WA2 * WA, WB3 = WB2 * WB;   // all exact

// six-point input transform, one scalar channel (12 fma: even part +- odd part for each symmetric pair):
//   o0 = a^2 b^2 d0 - (a^2 + b^2) d2 + d4                      o5 = a^2 b^2 d1 - (a^2 + b^2) d3 + d5
//   o1, o2 = (d4 - b^2 d2) +- a (d3 - b^2 d1)                  o3, o4 = (d4 - a^2 d2) +- b (d3 - a^2 d1)
struct Bt6 {
  float a, b, o0, o1, o2, o3, o4, o5;
  template <int S> __device__ __forceinline__ void step(float d0, float d1, float d2, float d3, float d4, float d5) {
    if constexpr (S == 0) { a = __builtin_fmaf(-WB2, d2, d4); b = __builtin_fmaf(-WB2, d1, d3); }
    if constexpr (S == 1) o0 = __builtin_fmaf(WP2, d0, __builtin_fmaf(-WS2, d2, d4));
    if constexpr (S == 2) { o1 = __builtin_fmaf(WA, b, a); o2 = __builtin_fmaf(-WA, b, a); }
    if constexpr (S == 3) { a = __builtin_fmaf(-WA2, d2, d4); b = __builtin_fmaf(-WA2, d1, d3); }
    if constexpr (S == 4) o5 = __builtin_fmaf(WP2, d1, __builtin_fmaf(-WS2, d3, d5));
    if constexpr (S == 5) { o3 = __builtin_fmaf(WB, b, a); o4 = __builtin_fmaf(-WB, b, a); }
  }
};

// the same on a channel PAIR (two scalar fma per operation: the library is built without packed-f32 instruction selection)
struct Bt6p {
  f32x2 a, b, o0, o1, o2, o3, o4, o5;
  static __device__ __forceinline__ f32x2 fma2(float c, f32x2 x, f32x2 y) {
    return __builtin_elementwise_fma(f32x2{c, c}, x, y);
  }
  template <int S> __device__ __forceinline__ void step(f32x2 d0, f32x2 d1, f32x2 d2, f32x2 d3, f32x2 d4, f32x2 d5) {
    if constexpr (S == 0) { a = fma2(-WB2, d2, d4); b = fma2(-WB2, d1, d3); }
    if constexpr (S == 1) o0 = fma2(WP2, d0, fma2(-WS2, d2, d4));
    if constexpr (S == 2) { o1 = fma2(WA, b, a); o2 = fma2(-WA, b, a); }
    if constexpr (S == 3) { a = fma2(-WA2, d2, d4); b = fma2(-WA2, d1, d3); }
    if constexpr (S == 4) o5 = fma2(WP2, d1, fma2(-WS2, d3, d5));
    if constexpr (S == 5) { o3 = fma2(WB, b, a); o4 = fma2(-WB, b, a); }
  }
};

// four-point output transform (rows of At): y0 = m0 + (m1 + m2) + (m3 + m4), y1 = a (m1 - m2) + b (m3 - m4),
// y2 = a^2 (m1 + m2) + b^2 (m3 + m4), y3 = a^3 (m1 - m2) + b^3 (m3 - m4) + m5
__device__ __forceinline__ void at4(float m0, float m1, float m2, float m3, float m4, float m5, float& y0, float& y1,
                                    float& y2, float& y3) {
  const float s1 = m1 + m2, d1 = m1 - m2, s2 = m3 + m4, d2 = m3 - m4;
  y0 = (m0 + s1) + s2;
  y1 = __builtin_fmaf(WB, d2, WA * d1);
  y2 = __builtin_fmaf(WB2, s2, WA2 * s1);
  y3 = __builtin_fmaf(WB3, d2, __builtin_fmaf(WA3, d1, m5));
}

template <int ABL, int WT, int WC, int KT>   // ABL: timing ablations (GLASS_W43_ABL): 0 = product, 1 = weights from one hot chunk, 2 = no transform VALU, 3 = no patch loads, 4 = phase stamps
__global__ __launch_bounds__(256, 1) void conv3x3_wino43_f32(WinoParams p) {
  static_assert(WT * WC == 4 && (KT == 32 || KT == 16) && 16 * WT * (KT / 2) == 256, "4 waves; one (tile, channel pair) per thread");
  constexpr int T4 = 16 * WT;                 // output tiles (4x4 pixels each) per block
  constexpr int N4 = 32 * WC;                 // output channels per block
  constexpr int K4 = KT;                      // input channels per k-tile
  constexpr int V4_FLOATS = 36 * T4 * K4;     // 72 KiB per stage
  constexpr int HALVES = KT / 16;             // 16-channel halves of a k-tile: one ds_read_b128 + 4 MFMA k-steps each
  constexpr int NG = 36 * HALVES;             // MFMA groups (xi, half) per k-tile, 8 MFMAs each
  constexpr int SPG = 144 / NG;               // transform / load micro-steps per group (108 steps per k-tile)
  constexpr int SLOTS = KT / 4;               // 16-byte slots per V row
  constexpr int RPW = 64 / KT;                // V rows per 256 bytes (one wrap of the 64 banks)
  static_assert(NG % RING == 0, "ring slot = group % RING must be consistent across k-tiles");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  // XCD-aware tile map (see conv.hip): cout-blocks innermost so the blocks that share an input patch sit on one L2
  const int nblk = gridDim.x, bid = blockIdx.x;
  const int xcd = bid & 7, q8 = nblk >> 3, r8 = nblk & 7;
  const int logical = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  const int tile_m = logical / p.tiles_n;
  const int tile_n = logical - tile_m * p.tiles_n;
  const int t0 = tile_m * T4, n0 = tile_n * N4;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tpi = p.TH * p.TW;
  // split-K (glass_conv3x3_winograd43_splitk_nhwc; gridDim.y slices of p.nk k-tiles each; one slice = the plain launch): this
  // workgroup's first k-tile, as the channel offset of its patch loads and the k-tile index of its weight fragments
  const int kt0 = blockIdx.y * p.nk;
  const int xk0 = kt0 * (KT * 4), uk0 = tile_n * p.nkt + kt0;
  unsigned long long stamp0 = 0, stamp1 = 0, stamp2 = 0, real0 = 0;
  if constexpr (ABL == 4) { stamp0 = __builtin_amdgcn_s_memtime(); real0 = __builtin_amdgcn_s_memrealtime(); }

  __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, (int)p.x_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t ur = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.u), 0, (int)p.u_bytes, 0x00020000);

  // ---- input role: thread = (tile tl, channel pair c2 of the k-tile) ----
  const int c2 = tid % (KT / 2), tl = tid / (KT / 2);
  const int wc = WC == 4 ? wv : (wv & (WC - 1)), wt = WC == 4 ? 0 : (wv / WC);
  unsigned rowoff[6], coloff[6];
  {
    const int t = t0 + tl;
    const bool tv = t < p.ntiles;
    const int n = fast_div(t, tpi, p.magic_tpi);
    const int rem = t - n * tpi;
    const int th = fast_div(rem, p.TW, p.magic_tw);
    const int tw = rem - th * p.TW;
    const int h0 = 4 * th - 1, w0 = 4 * tw - 1;
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      const int hi = h0 + r;
      rowoff[r] = (tv && (unsigned)hi < (unsigned)p.H) ? (unsigned)(((n * p.H + hi) * p.W) * p.ldx * 4) : INV;
    }
#pragma unroll
    for (int c = 0; c < 6; ++c) {
      const int wi = w0 + c;
      coloff[c] = ((unsigned)wi < (unsigned)p.W) ? (unsigned)((wi * p.ldx + 2 * c2) * 4) : INV;
    }
  }
  f32x2 dd[36];                               // the channel pair of the 6x6 patch, transformed in place
  auto load_patch1 = [&](int kt, auto i_) {
    constexpr int i = decltype(i_)::value;
    if constexpr (ABL == 3) { if (kt > 1) return; }
    // (bit_cast the whole vector: __builtin_bit_cast of a single vector ELEMENT reads element 0 with this compiler)
    dd[i] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(xr, rowoff[i / 6] + coloff[i % 6], xk0 + kt * (K4 * 4), 0));
  };
  // V[stage][xi][tile][KT]: a row is 128 (64) bytes = a half (quarter) of the 64 banks; the 16-byte slot is XOR-ed with
  // (tile / rows-per-256-bytes) % slots so that the 16 tiles a ds_read_b128 service group touches hit 16 different slots.
  float* vdst = smem + tl * K4 + (((c2 >> 1) ^ ((tl / RPW) % SLOTS)) * 4) + (c2 & 1) * 2;
  Bt6p bt;
  // transform micro-steps (compile-time index): 36 row steps (column c, step S), then per row i six column steps
  // (which write V[i][*] to LDS as they go)
  auto row_step = [&](auto c_, auto s_) {
    constexpr int c = decltype(c_)::value, S = decltype(s_)::value;
    if constexpr (ABL == 2) return;
    bt.template step<S>(dd[0 + c], dd[6 + c], dd[12 + c], dd[18 + c], dd[24 + c], dd[30 + c]);
    if constexpr (S == 5) {
      dd[0 + c] = bt.o0; dd[6 + c] = bt.o1; dd[12 + c] = bt.o2; dd[18 + c] = bt.o3; dd[24 + c] = bt.o4; dd[30 + c] = bt.o5;
    }
  };
  auto vstore = [&](int stage, int xi, f32x2 v) {
    *reinterpret_cast<f32x2*>(vdst + stage * V4_FLOATS + xi * (T4 * K4)) = v;
  };
  auto col_step = [&](int stage, auto i_, auto s_) {
    constexpr int i = decltype(i_)::value, S = decltype(s_)::value;
    if constexpr (ABL == 2) {
      if constexpr (S == 1) vstore(stage, i * 6 + 0, dd[i * 6 + 0]);
      if constexpr (S == 2) vstore(stage, i * 6 + 1, dd[i * 6 + 1]);
      if constexpr (S == 3) vstore(stage, i * 6 + 2, dd[i * 6 + 2]);
      if constexpr (S == 4) vstore(stage, i * 6 + 5, dd[i * 6 + 5]);
      if constexpr (S == 5) { vstore(stage, i * 6 + 3, dd[i * 6 + 3]); vstore(stage, i * 6 + 4, dd[i * 6 + 4]); }
      return;
    }
    bt.template step<S>(dd[i * 6 + 0], dd[i * 6 + 1], dd[i * 6 + 2], dd[i * 6 + 3], dd[i * 6 + 4], dd[i * 6 + 5]);
    if constexpr (S == 1) vstore(stage, i * 6 + 0, bt.o0);
    if constexpr (S == 2) vstore(stage, i * 6 + 1, bt.o1);
    if constexpr (S == 3) vstore(stage, i * 6 + 2, bt.o2);
    if constexpr (S == 4) vstore(stage, i * 6 + 5, bt.o5);
    if constexpr (S == 5) { vstore(stage, i * 6 + 3, bt.o3); vstore(stage, i * 6 + 4, bt.o4); }
  };

  // ---- MFMA role: wave (wt, wc) owns tiles 16 wt + [0, 16) x channels n0 + 32 wc + [0, 32) for all 36 xi ----
  // 16x16x4: A[i = lane&15][k = lane>>4] = weights (row i = 4 g' + e  <->  channel 32 wc + 16 cb + 4 g' + e),
  //          B[k = lane>>4][j = lane&15] = V (tile j);  C/D: lane holds rows 4 (lane>>4) + e, column lane&15.
  f32x4 acc[36][2];
#pragma unroll
  for (int i = 0; i < 36; ++i)
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) acc[i][cb] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int vj = lane & 15, kg = lane >> 4;
  const int vt = 16 * wt + vj;                // this lane's tile within the block
  const int vswz = (vt / RPW) % SLOTS;
  const float* vb[2] = {smem + vt * K4 + ((0 * 4 + kg) ^ vswz) * 4, smem + vt * K4 + (((HALVES - 1) * 4 + kg) ^ vswz) * 4};
  // packed U: [tile_n][kt][xi][wc][half][cb] chunks of 1 KiB (64 lanes x float4); group u = HALVES xi + half.
  // Address = wave part (lane, wc: one VGPR, set once) + (kt, xi) part (ONE scalar add per xi: the 2 HALVES chunks of an xi are
  // contiguous, <= 3 KiB apart) + (half, cb) part as the instruction's 12-bit immediate - round 6: the loop carried one
  // s_add_i32 per weight load (146 per k-tile, an issue slot each beside the MFMAs); now 36.
  const unsigned a_voff = (unsigned)lane * 16u + (unsigned)wc * (HALVES * 2 * 1024);
  f32x4 aq[RING][2];                          // weight fragments: [ring slot = group % RING][cb]
  f32x4 vq[2];                                // V fragments: [group & 1]
  auto load_a1 = [&](int kt, int u, int cb) {
    const int xi = u / HALVES, h = u % HALVES;
#ifdef GLASS_W43_R5_ADDR     // the round-5 form (A/B builds only): the whole chunk offset in the scalar operand
    const int sbase = ABL == 1 ? (h * 2 + cb) * 1024 : ((uk0 + kt) * 36 + xi) * (WC * HALVES * 2 * 1024) + (h * 2 + cb) * 1024;
    aq[u % RING][cb] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ur, a_voff, sbase, 0));
#else
    const int sbase = ABL == 1 ? 0 : ((uk0 + kt) * 36 + xi) * (WC * HALVES * 2 * 1024);
    aq[u % RING][cb] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ur, a_voff + (unsigned)((h * 2 + cb) * 1024), sbase, 0));
#endif
  };
  auto read_v = [&](int stage, int u) {
    vq[u & 1] = *reinterpret_cast<const f32x4*>(vb[u % HALVES] + stage * V4_FLOATS + (u / HALVES) * (T4 * K4));
  };

  // ---- prologue: patch 0 -> V[0], weight fragments of the first three groups, patch 1 in flight ----
  static_for<36>([&](auto i_) { load_patch1(0, i_); });
#pragma unroll
  for (int u = 0; u < RING - 1; ++u) { load_a1(0, u, 0); load_a1(0, u, 1); }
  static_for<6>([&](auto c_) { static_for<6>([&](auto s_) { row_step(c_, s_); }); });
  static_for<6>([&](auto i_) { static_for<6>([&](auto s_) { col_step(0, i_, s_); }); });
  {
    const int k1 = p.nk > 1 ? 1 : 0;
    static_for<36>([&](auto i_) { load_patch1(k1, i_); });
  }
  __syncthreads();
  if constexpr (ABL == 4) stamp1 = __builtin_amdgcn_s_memtime();

  for (int kt = 0; kt < p.nk; ++kt) {
    const int cur = kt & 1;
    const int ktn = kt + 1 < p.nk ? kt + 1 : kt;     // clamped at the end: harmless re-reads keep the loop one block
    const int ktnn = kt + 2 < p.nk ? kt + 2 : kt;
    read_v(cur, 0);
    static_for<NG>([&](auto u_) {
      constexpr int u = decltype(u_)::value;
      constexpr int xi = u / HALVES;
      static_for<8>([&](auto m_) {
        constexpr int m = decltype(m_)::value;
        constexpr int s = m >> 1, cb = m & 1;
        // side work issued BEFORE MFMA (u, m)
        if constexpr (m == 0 || m == 2) {                 // weight fragments of the group RING - 1 ahead
          constexpr int u3 = u + RING - 1;
          if constexpr (u3 < NG) load_a1(kt, u3, m >> 1); else load_a1(ktn, u3 - NG, m >> 1);
          __builtin_amdgcn_sched_barrier(0);
        } else if constexpr (m == 4) {
          if constexpr (u + 1 < NG) { read_v(cur, u + 1); __builtin_amdgcn_sched_barrier(0); }
        } else if constexpr (SPG == 2 ? (m == 1 || m == 5) : (m & 1) == 1) {
          // 144 slots for the 108 transform / load steps; the first 36 stay empty so that the patch loads issued at the
          // end of the previous k-tile (or by the prologue) have ~4 K cycles to land before the first row step waits on them
          constexpr int q = (SPG == 2 ? u * 2 + (m == 5 ? 1 : 0) : u * 4 + (m >> 1)) - 36;
          if constexpr (q < 0) {
          } else if constexpr (q < 36) {
            row_step(ic<q / 6>{}, ic<q % 6>{});           // Bt d of patch kt+1, column q/6
            __builtin_amdgcn_sched_barrier(0);
          } else if constexpr (q < 108) {
            constexpr int i = (q - 36) / 12, sub = (q - 36) % 12;
            if constexpr (sub < 6) col_step(cur ^ 1, ic<i>{}, ic<sub>{});       // (. B) of row i -> V[cur^1]
            else load_patch1(ktnn, ic<i * 6 + (sub - 6)>{});                     // row i is free: patch of k-tile kt+2
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        // 72 accumulators of 4 registers = 288 > the 256 AGPRs: hipcc selects the AGPR form for every builtin MFMA of a
        // 512-register kernel and shuttled the 8 accumulators that do not fit through scratch AGPRs around their MFMAs
        // (40 v_accvgpr_write + 40 v_accvgpr_read + 12 x `s_nop 8..9` for the results, per k-tile and wave).  The MFMAs of
        // the last four xi name their VGPR accumulator themselves (round 6): same instruction, C / D in VGPRs.
        if constexpr (xi >= VACC_XI) {
          f32x4 c = acc[xi][cb];                   // (asm operands cannot name a captured variable; the copies fold away)
          const float af = comp4(aq[u % RING][cb], s), bf = comp4(vq[u & 1], s);
          asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(c) : "v"(af), "v"(bf));
          acc[xi][cb] = c;
        } else {
          acc[xi][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(comp4(aq[u % RING][cb], s), comp4(vq[u & 1], s), acc[xi][cb], 0, 0, 0);
        }
      });
      __builtin_amdgcn_sched_barrier(0);
    });
    __syncthreads();     // V[cur] fully read, V[cur^1] fully written
  }

  // (the inline-asm MFMAs are invisible to the compiler's hazard recognizer: a VALU read of their VGPR results needs the matrix
  //  pipe drained - 8 passes + margin.  The loop ends with a barrier and ~40 address instructions precede the first read; these
  //  wait states make the distance explicit instead of incidental, and the empty statements tie every VGPR accumulator to them:
  //  volatile asm statements keep their order, so no read of acc[VACC_XI..35] can be scheduled above the wait states.)
  if constexpr (VACC_XI < 36) {
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
#pragma unroll
    for (int i = VACC_XI; i < 36; ++i) {
      asm volatile("" : "+v"(acc[i][0]));
      asm volatile("" : "+v"(acc[i][1]));
    }
  }
  if constexpr (ABL == 4) stamp2 = __builtin_amdgcn_s_memtime();
  // ---- epilogue: Y = At M A in registers; lane = (tile 16 wt + vj, channels n0 + 32 wc + 16 cb + 4 kg + e) ----
  // 32-bit buffer addressing with split offsets (row part + column part; an invalid part = 2^30 makes the sum out of
  // range): a pixel that does not exist (ragged last tile block, H or W not a multiple of 4) loads zeros / drops the store.
  __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(p.y + (long)blockIdx.y * p.y_slice, 0, (int)p.y_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.res_mode == 1 ? p.res : p.y), 0,
                                                                 (int)(p.res_mode == 1 ? p.r_bytes : 0u), 0x00020000);
  const int cbase = n0 + 32 * wc + 4 * kg;      // + 16 cb: one store instruction covers 64 contiguous bytes per pixel
  const unsigned ldy4 = (unsigned)p.ldy * 4u, ldr4 = (unsigned)p.ldr * 4u;
  unsigned yrow[4], ycol[4], rrow[4], rcol[4];
  {
    const int t = t0 + vt;
    const bool tv = t < p.ntiles;
    const int n = fast_div(t, tpi, p.magic_tpi);
    const int rem = t - n * tpi;
    const int th = fast_div(rem, p.TW, p.magic_tw);
    const int tw = rem - th * p.TW;
    const unsigned pix = (unsigned)((n * p.H + 4 * th) * p.W + 4 * tw);
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const bool ok = tv && 4 * th + a < p.H;
      yrow[a] = ok ? (pix + (unsigned)(a * p.W)) * ldy4 + (unsigned)(p.ycoff + cbase) * 4u : INV;
      rrow[a] = ok ? (pix + (unsigned)(a * p.W)) * ldr4 + (unsigned)cbase * 4u : INV;
    }
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const bool ok = 4 * tw + b < p.W;
      ycol[b] = ok ? (unsigned)b * ldy4 : INV;
      rcol[b] = ok ? (unsigned)b * ldr4 : INV;
    }
  }
  // ReLU before (2) / after (1) the residual add as an unconditional max: max(x, qNaN) = x keeps "no ReLU" exact
  const float lo2 = p.relu == 2 ? 0.f : __builtin_nanf(""), lo1 = p.relu == 1 ? 0.f : __builtin_nanf("");
  f32x4 rres[2][16];
  auto load_res = [&](auto cb_) {
    constexpr int cb = decltype(cb_)::value;
    if (p.res_mode == 1) {
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
          rres[cb][a * 4 + b] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rr, rrow[a] + rcol[b], cb * 64, 0));
    }
  };
#ifdef GLASS_W43_SPLIT_STORES   // the round-2..4 epilogue (A/B builds only: scripts/build_variant_lib.sh w43old -DGLASS_W43_SPLIT_STORES)
  load_res(ic<0>{});
  static_for<2>([&](auto cb_) {
    constexpr int cb = decltype(cb_)::value;
    f32x4 bv = f32x4{0.f, 0.f, 0.f, 0.f};
    if (p.bias != nullptr) bv = *reinterpret_cast<const f32x4*>(p.bias + cbase + 16 * cb);
    f32x4 out[16];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float z[6][4];
#pragma unroll
      for (int i = 0; i < 6; ++i)
        at4(acc[i * 6 + 0][cb][e], acc[i * 6 + 1][cb][e], acc[i * 6 + 2][cb][e], acc[i * 6 + 3][cb][e], acc[i * 6 + 4][cb][e],
            acc[i * 6 + 5][cb][e], z[i][0], z[i][1], z[i][2], z[i][3]);
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        float y0, y1, y2, y3;
        at4(z[0][b], z[1][b], z[2][b], z[3][b], z[4][b], z[5][b], y0, y1, y2, y3);
        out[0 * 4 + b][e] = y0; out[1 * 4 + b][e] = y1; out[2 * 4 + b][e] = y2; out[3 * 4 + b][e] = y3;
      }
    }
    // the other channel block's residual rows are requested as soon as this block's accumulators are dead, a whole
    // output transform ahead of their use
    if constexpr (cb == 0) { __builtin_amdgcn_sched_barrier(0); load_res(ic<1>{}); __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        f32x4 v = out[a * 4 + b] + bv;
        v.x = fmaxf(v.x, lo2); v.y = fmaxf(v.y, lo2); v.z = fmaxf(v.z, lo2); v.w = fmaxf(v.w, lo2);
        if (p.res_mode == 1) v = v + rres[cb][a * 4 + b];
        v.x = fmaxf(v.x, lo1); v.y = fmaxf(v.y, lo1); v.z = fmaxf(v.z, lo1); v.w = fmaxf(v.w, lo1);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), yr, yrow[a] + ycol[b], cb * 64, 0);
      }
  });
#else
  load_res(ic<0>{});
  // Both channel blocks' outputs are finished BEFORE the first store, and a pixel's two 64-byte halves (cb = 0 | 1: one 128-byte
  // line) leave in adjacent store instructions.  Storing block 0 right after its transform - a whole output transform (~1 us)
  // before block 1 - had the L2 write a line back half-filled and again when the other half arrived: PMC WRITE_SIZE was 1.36x
  // the output tensor on the FPN p2 layer (scripts/exp_write_size.py; a copy, the implicit-GEMM and the pointwise kernel
  // write 1.00x).  The accumulators are dead by then, so the 128 output registers cost nothing.
  f32x4 outs[2][16];
  static_for<2>([&](auto cb_) {
    constexpr int cb = decltype(cb_)::value;
    f32x4 bv = f32x4{0.f, 0.f, 0.f, 0.f};
    if (p.bias != nullptr) bv = *reinterpret_cast<const f32x4*>(p.bias + cbase + 16 * cb);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float z[6][4];
#pragma unroll
      for (int i = 0; i < 6; ++i)
        at4(acc[i * 6 + 0][cb][e], acc[i * 6 + 1][cb][e], acc[i * 6 + 2][cb][e], acc[i * 6 + 3][cb][e], acc[i * 6 + 4][cb][e],
            acc[i * 6 + 5][cb][e], z[i][0], z[i][1], z[i][2], z[i][3]);
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        float y0, y1, y2, y3;
        at4(z[0][b], z[1][b], z[2][b], z[3][b], z[4][b], z[5][b], y0, y1, y2, y3);
        outs[cb][0 * 4 + b][e] = y0; outs[cb][1 * 4 + b][e] = y1; outs[cb][2 * 4 + b][e] = y2; outs[cb][3 * 4 + b][e] = y3;
      }
    }
    // the other channel block's residual rows are requested as soon as this block's accumulators are dead, a whole
    // output transform ahead of their use
    if constexpr (cb == 0) { __builtin_amdgcn_sched_barrier(0); load_res(ic<1>{}); __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
    for (int ab = 0; ab < 16; ++ab) {
      f32x4 v = outs[cb][ab] + bv;
      v.x = fmaxf(v.x, lo2); v.y = fmaxf(v.y, lo2); v.z = fmaxf(v.z, lo2); v.w = fmaxf(v.w, lo2);
      if (p.res_mode == 1) v = v + rres[cb][ab];
      v.x = fmaxf(v.x, lo1); v.y = fmaxf(v.y, lo1); v.z = fmaxf(v.z, lo1); v.w = fmaxf(v.w, lo1);
      outs[cb][ab] = v;
    }
  });
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, outs[0][a * 4 + b]), yr, yrow[a] + ycol[b], 0, W43_STORE_AUX);
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, outs[1][a * 4 + b]), yr, yrow[a] + ycol[b], 64, W43_STORE_AUX);
    }
#endif
  if constexpr (ABL == 4) {
    __builtin_amdgcn_s_waitcnt(0);
    const unsigned long long stamp3 = __builtin_amdgcn_s_memtime(), real3 = __builtin_amdgcn_s_memrealtime();
    if (p.dbg != nullptr && tid == 0) {
      // + the block's life on the 100 MHz counter every CU shares, and WHERE it ran (HW_REG_HW_ID = 4: cu_id [11:8], sh_id
      // [12], se_id [15:13]; HW_REG_XCC_ID = 20): the gap accounting groups blocks by CU (scripts/exp_w43_gap.py)
      unsigned long long* o = p.dbg + (long)blockIdx.x * 8;
      o[0] = stamp0; o[1] = stamp1; o[2] = stamp2; o[3] = stamp3; o[4] = real0; o[5] = real3;
      o[6] = (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 4);
      o[7] = (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 20);
    }
  }
}

// U = G g G^t (6x6 per channel pair) in the fragment order the kernel streams:
// [cout/(32 WC)][cin/KT][xi][wc][half][cb][lane][s]   with
//   cout = 32 WC tn + 32 wc + 16 cb + 4 ((lane&15)>>2) + (lane&3),   cin = KT kt + 16 half + 4 (lane>>4) + s
__global__ void wino43_pack_weights_kernel(const float* __restrict__ w, float* __restrict__ u, int Cout, int Cin, int WC, int KT) {
  const long total = 36L * Cout * Cin;
  const int nk = Cin / KT, halves = KT / 16;
  // G rows of the points (0, a, -a, b, -b, inf): (1, p, p^2) / prod_{q != p} (p - q), in double
  const double P[5] = {0.0, (double)WA, -(double)WA, (double)WB, -(double)WB};
  double G[6][3];
  for (int i = 0; i < 5; ++i) {
    double f = 1.0;
    for (int j = 0; j < 5; ++j) if (j != i) f *= P[i] - P[j];
    G[i][0] = 1.0 / f; G[i][1] = P[i] / f; G[i][2] = P[i] * P[i] / f;
  }
  G[5][0] = 0.0; G[5][1] = 0.0; G[5][2] = 1.0;
  for (long o = (long)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (long)gridDim.x * blockDim.x) {
    long r = o;
    const int s = (int)(r & 3); r >>= 2;
    const int lane = (int)(r & 63); r >>= 6;
    const int cb = (int)(r & 1); r >>= 1;
    const int half = (int)(r % halves); r /= halves;
    const int wave = (int)(r % WC); r /= WC;
    const int xi = (int)(r % 36); r /= 36;
    const int kt = (int)(r % nk);
    const int tn = (int)(r / nk);
    const int co = tn * 32 * WC + 32 * wave + 16 * cb + 4 * ((lane & 15) >> 2) + (lane & 3);
    const int ci = kt * KT + 16 * half + 4 * (lane >> 4) + s;
    const int i = xi / 6, j = xi % 6;
    double acc = 0.0;
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int b = 0; b < 3; ++b) acc += G[i][a] * G[j][b] * (double)w[(((long)co * 3 + a) * 3 + b) * Cin + ci];
    u[o] = (float)acc;
  }
}

}  // namespace

// wide shape when the channel counts allow it, else the narrow one (both need Cout % 64 == 0 and Cin % 16 == 0)
static bool wino43_wide(int Cout, int Cin) { return Cout % 128 == 0 && Cin % 32 == 0; }

extern "C" int glass_winograd43_supported(const glass_conv_desc* d) {
  if (!d) return 0;
  const long xb = (long)d->N * d->H * d->W * d->ldx * 4;
  const long yb = (long)d->N * d->H * d->W * d->ldy * 4, rb = d->res_mode == 1 ? (long)d->N * d->H * d->W * d->ldr * 4 : 0;
  const long lim = 0x40000000L;               // split offsets: every operand below 1 GiB
  return d->KH == 3 && d->KW == 3 && d->stride_h == 1 && d->stride_w == 1 && d->pad_h == 1 && d->pad_w == 1 &&
         d->Cin % 16 == 0 && d->Cout % 64 == 0 && d->ldx % 2 == 0 && d->ldx >= d->Cin && d->y_cstride == 1 && d->ldy % 4 == 0 &&
         d->y_coff % 4 == 0 && d->y_coff >= 0 && d->y_coff + d->Cout <= d->ldy &&
         (d->res_mode == 0 || (d->res_mode == 1 && d->ldr % 4 == 0 && d->ldr >= d->Cout)) && xb < lim && yb < lim && rb < lim &&
         36L * d->Cout * d->Cin * 4 < 0x7fffff00L && d->Ho == d->H && d->Wo == d->W;
}

extern "C" size_t glass_winograd43_weight_floats(int Cout, int Cin) { return (size_t)36 * (size_t)Cout * (size_t)Cin; }

extern "C" int glass_winograd43_pack_weights(const float* w, int Cout, int Cin, float* u_packed, glass_stream_t stream) {
  GLASS_CHECK_ARG(w && u_packed, "glass_winograd43_pack_weights: null pointer");
  GLASS_CHECK_ARG(Cout > 0 && Cin > 0 && Cout % 64 == 0 && Cin % 16 == 0,
                  "glass_winograd43_pack_weights: Cout=%d must be a multiple of 64 and Cin=%d a multiple of 16", Cout, Cin);
  const long total = 36L * Cout * Cin;
  const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
  const bool wide = wino43_wide(Cout, Cin);
  hipLaunchKernelGGL(wino43_pack_weights_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, u_packed, Cout, Cin,
                     wide ? 4 : 2, wide ? 32 : 16);
  GLASS_CHECK_LAUNCH("glass_winograd43_pack_weights");
  return GLASS_OK;
}

static int wino43_launch(const glass_conv_desc* d, const float* x, const float* u_packed, const float* bias, const float* residual,
                         float* y, glass_stream_t stream, bool body_only, int splits = 1, void* workspace = nullptr);

extern "C" int glass_conv3x3_winograd43_nhwc(const glass_conv_desc* d, const float* x, const float* u_packed,
                                             const float* bias, const float* residual, float* y, glass_stream_t stream) {
  return wino43_launch(d, x, u_packed, bias, residual, y, stream, false);
}

// Only the FULL 4-column tile columns: output columns [0, 4 * (W / 4)) of every row.  A map whose width is 4 k + 1 (the local
// extractor's 16 x 33 maps, reference local_feature_extraction.py:123 `MaxPool2d(2, (2, 1), (0, 1))`) otherwise pays a whole
// tile column - 36 multiplies per tile, exactly the direct convolution of its 16 outputs - for ONE pixel column, and its
// tile count (36 per map instead of 32) turns 4 rounds of workgroups into 4.5; the caller computes the last column with
// glass_conv2d_nhwc on the 2-column strip (KH = 3, KW = 1 over channels = (kw, cin); ops/native.py conv2d_nhwc).
extern "C" int glass_conv3x3_winograd43_body_nhwc(const glass_conv_desc* d, const float* x, const float* u_packed,
                                                  const float* bias, const float* residual, float* y, glass_stream_t stream) {
  GLASS_CHECK_ARG(d && d->W >= 4, "glass_conv3x3_winograd43_body_nhwc: needs W >= 4");
  return wino43_launch(d, x, u_packed, bias, residual, y, stream, true);
}

// Split-K (round 6): the layers whose F(4x4) grid leaves most of the chip idle when ONE image is in flight (reference predictor:
// glass/inference/glass_runner.py:93-96) - res4 / res5 3x3 on 64 x 64 / 32 x 32 maps (32 / 16 workgroups of 8 / 16 k-tiles), the
// fusion conv, FPN / RPN on the small levels.  `splits` k-slices of the layer run as gridDim.y (slice s = k-tiles [s nk / splits,
// (s+1) nk / splits)); a slice's output transform is linear, so it writes its raw partial Y to workspace[s][N H W][Cout] and the
// ordered reduction of splitk_common.h adds the slices, then bias / ReLU / residual, and writes y with its strides.  With
// `body_only` the last pixel column of y is written by the reduction from unwritten workspace and must be overwritten by the
// caller's strip convolution, as after glass_conv3x3_winograd43_body_nhwc.  Wide shape only (Cout % 128 == 0, Cin % 32 == 0).
extern "C" int glass_winograd43_splitk_supported(const glass_conv_desc* d, int splits) {
  return glass_winograd43_supported(d) && wino43_wide(d->Cout, d->Cin) && splits >= 2 && splits <= 32 && (d->Cin / 32) % splits == 0 &&
         (long)d->N * d->H * d->W * d->Cout * 4 < 0x40000000L;
}

extern "C" int64_t glass_winograd43_splitk_workspace_bytes(const glass_conv_desc* d, int splits) {
  if (!d) return 0;
  return (int64_t)splits * d->N * d->H * d->W * d->Cout * (int64_t)sizeof(float);
}

extern "C" int glass_conv3x3_winograd43_splitk_nhwc(const glass_conv_desc* d, const float* x, const float* u_packed, const float* bias,
                                                    const float* residual, float* y, int splits, int body_only, void* workspace,
                                                    int64_t workspace_bytes, glass_stream_t stream) {
  GLASS_CHECK_ARG(d && workspace, "glass_conv3x3_winograd43_splitk_nhwc: null pointer");
  GLASS_CHECK_ARG(glass_winograd43_splitk_supported(d, splits),
                  "glass_conv3x3_winograd43_splitk_nhwc: needs the wide F(4x4) shape (Cout %% 128 == 0, Cin %% 32 == 0), 2 <= splits <= 32 "
                  "dividing Cin / 32 (got Cin=%d Cout=%d splits=%d)", d->Cin, d->Cout, splits);
  GLASS_CHECK_ARG(workspace_bytes >= glass_winograd43_splitk_workspace_bytes(d, splits) && ((uintptr_t)workspace & 15) == 0,
                  "glass_conv3x3_winograd43_splitk_nhwc: workspace too small or not 16-byte aligned");
  GLASS_CHECK_ARG(!body_only || d->W >= 4, "glass_conv3x3_winograd43_splitk_nhwc: body_only needs W >= 4");
  return wino43_launch(d, x, u_packed, bias, residual, y, stream, body_only != 0, splits, workspace);
}

static int wino43_launch(const glass_conv_desc* d, const float* x, const float* u_packed, const float* bias, const float* residual,
                         float* y, glass_stream_t stream, bool body_only, int splits, void* workspace) {
  GLASS_CHECK_ARG(d && x && u_packed && y, "glass_conv3x3_winograd43_nhwc: null pointer");
  GLASS_CHECK_ARG(glass_winograd43_supported(d),
                  "glass_conv3x3_winograd43_nhwc: needs 3x3/stride 1/pad 1, Cin%%16==0, Cout%%64==0, unit channel stride, "
                  "res_mode 0/1 and operands < 1 GiB (got Cin=%d Cout=%d k=%dx%d s=%d p=%d)", d->Cin, d->Cout, d->KH, d->KW,
                  d->stride_h, d->pad_h);
  GLASS_CHECK_ARG(d->res_mode == 0 || residual != nullptr, "glass_conv3x3_winograd43_nhwc: res_mode set but residual is null");
  GLASS_CHECK_ARG(((uintptr_t)x & 15) == 0 && ((uintptr_t)u_packed & 15) == 0 && ((uintptr_t)y & 15) == 0 &&
                      (bias == nullptr || ((uintptr_t)bias & 15) == 0) && (residual == nullptr || ((uintptr_t)residual & 15) == 0),
                  "glass_conv3x3_winograd43_nhwc: pointers must be 16-byte aligned");
  if (d->N == 0) return GLASS_OK;
  WinoParams p;
  p.x = x; p.u = u_packed; p.bias = bias; p.res = residual; p.y = y; p.dbg = nullptr;
  p.N = d->N; p.H = d->H; p.W = d->W; p.Cin = d->Cin; p.Cout = d->Cout;
  p.TH = (d->H + 3) / 4; p.TW = body_only ? d->W / 4 : (d->W + 3) / 4;      // (input / output bounds still use the true W)
  const long nt = (long)d->N * p.TH * p.TW;
  GLASS_CHECK_ARG(nt < 0x7fffffffL, "glass_conv3x3_winograd43_nhwc: too many tiles");
  p.ntiles = (int)nt;
  const bool wide = wino43_wide(d->Cout, d->Cin);
  const int T4 = wide ? 16 : 32, N4 = wide ? 128 : 64, K4 = wide ? 32 : 16;
  p.nkt = d->Cin / K4;
  p.nk = p.nkt / splits;
  p.y_slice = 0;
  p.ldx = d->ldx; p.ldy = d->ldy; p.ycoff = d->y_coff; p.ldr = d->ldr; p.relu = d->relu; p.res_mode = d->res_mode;
  p.tiles_m = cdiv(p.ntiles, T4);
  p.tiles_n = d->Cout / N4;
  p.x_bytes = (unsigned)((long)d->N * d->H * d->W * d->ldx * 4);
  p.magic_tpi = (unsigned)(0x100000000ULL / (unsigned long long)(p.TH * p.TW));
  p.magic_tw = (unsigned)(0x100000000ULL / (unsigned long long)p.TW);
  p.u_bytes = (unsigned)(36L * d->Cout * d->Cin * 4);
  p.y_bytes = (unsigned)((long)d->N * d->H * d->W * d->ldy * 4);
  p.r_bytes = d->res_mode == 1 ? (unsigned)((long)d->N * d->H * d->W * d->ldr * 4) : 0u;
  const long Mpix = (long)d->N * d->H * d->W;
  if (splits > 1) {       // the slices write raw partial sums: dense [M][Cout] rows, no bias / ReLU / residual
    p.y = static_cast<float*>(workspace); p.bias = nullptr; p.res = nullptr;
    p.ldy = d->Cout; p.ycoff = 0; p.relu = 0; p.res_mode = 0; p.ldr = 0; p.r_bytes = 0;
    p.y_slice = Mpix * d->Cout;
    p.y_bytes = (unsigned)(Mpix * d->Cout * 4);
  }
  const long nblk = (long)p.tiles_m * p.tiles_n;
  GLASS_CHECK_ARG(nblk > 0 && nblk <= 0x7fffffffL, "glass_conv3x3_winograd43_nhwc: bad grid");
  static const int abl = getenv("GLASS_W43_ABL") ? atoi(getenv("GLASS_W43_ABL")) : 0;      // timing ablations (wrong results)
  // (the ablation instantiations 1..3 are compiled only with -DGLASS_W43_ABLATIONS: they triple the build time of this file)
  auto kern = !wide ? (abl == 4 ? conv3x3_wino43_f32<4, 2, 2, 16> : conv3x3_wino43_f32<0, 2, 2, 16>)
#ifdef GLASS_W43_ABLATIONS
            : abl == 1 ? conv3x3_wino43_f32<1, 1, 4, 32> : abl == 2 ? conv3x3_wino43_f32<2, 1, 4, 32> : abl == 3 ? conv3x3_wino43_f32<3, 1, 4, 32>
#endif
            : abl == 4 ? conv3x3_wino43_f32<4, 1, 4, 32> : conv3x3_wino43_f32<0, 1, 4, 32>;
  static unsigned long long* dbg_dev = nullptr;
  if (abl == 4) {
    if (!dbg_dev) (void)hipMalloc(&dbg_dev, 8L * 8 * 65536);
    p.dbg = nblk <= 65536 ? dbg_dev : nullptr;
  }
  static int attr_rc_w = -1, attr_rc_n = -1;
  int& attr_rc = wide ? attr_rc_w : attr_rc_n;
  if (attr_rc == -1)
    attr_rc = (int)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, WINO43_LDS_BYTES);
  if (attr_rc != 0) {
    glass_set_error("glass_conv3x3_winograd43_nhwc: cannot reserve %d bytes of LDS (hip error %d)", WINO43_LDS_BYTES, attr_rc);
    return GLASS_EHIP;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)nblk, (unsigned)splits), dim3(256), WINO43_LDS_BYTES, (hipStream_t)stream, p);
  GLASS_CHECK_LAUNCH("glass_conv3x3_winograd43_nhwc");
  if (splits > 1) {
    launch_splitk_reduce(static_cast<const float*>(workspace), bias, residual, y, Mpix, d->Cout, splits, d->ldy, d->y_coff, 1, d->ldr, d->relu,
                         d->res_mode, (hipStream_t)stream);
    GLASS_CHECK_LAUNCH("glass_conv3x3_winograd43_splitk_nhwc(reduce)");
  }
  if (abl == 4 && p.dbg) {      // instrumented build: print the phase times of a few workgroups (drains the stream)
    static int printed = 0;
    if (printed++ < (getenv("GLASS_W43_DBG_DUMP") ? 256 : 4)) {
      (void)hipStreamSynchronize((hipStream_t)stream);
      std::vector<unsigned long long> h(8 * nblk);
      (void)hipMemcpy(h.data(), dbg_dev, h.size() * 8, hipMemcpyDeviceToHost);
      unsigned long long t_min = ~0ull, t_max = 0, r_min = ~0ull, r_max = 0;
      double pro = 0, loop = 0, epi = 0, life = 0;
      for (long b = 0; b < nblk; ++b) {
        t_min = h[8 * b] < t_min ? h[8 * b] : t_min; t_max = h[8 * b + 3] > t_max ? h[8 * b + 3] : t_max;
        r_min = h[8 * b + 4] < r_min ? h[8 * b + 4] : r_min; r_max = h[8 * b + 5] > r_max ? h[8 * b + 5] : r_max;
        pro += (double)(h[8 * b + 1] - h[8 * b]); loop += (double)(h[8 * b + 2] - h[8 * b + 1]); epi += (double)(h[8 * b + 3] - h[8 * b + 2]);
        life += (double)(h[8 * b + 5] - h[8 * b + 4]);
      }
      fprintf(stderr, "[w43 dbg] blocks %ld nk %d: mean s_memtime ticks prologue %.0f  k-loop %.0f (%.0f / k-tile)  epilogue %.0f | block life %.2f us, kernel span %.2f us "
              "(s_memrealtime) = %llu ticks -> s_memtime at %.1f MHz\n",
              nblk, p.nk, pro / nblk, loop / nblk, loop / nblk / p.nk, epi / nblk, life / nblk / 100.0, (double)(r_max - r_min) / 100.0, t_max - t_min,
              (double)(t_max - t_min) / ((double)(r_max - r_min) / 100.0));
      if (const char* path = getenv("GLASS_W43_DBG_DUMP")) {      // raw per-block records for scripts/exp_w43_gap.py
        if (FILE* f = fopen(path, "ab")) {
          const long hdr[4] = {nblk, p.nk, p.tiles_n, wide ? 1 : 0};
          fwrite(hdr, sizeof(long), 4, f); fwrite(h.data(), 8, h.size(), f); fclose(f);
        }
      }
    }
  }
  return GLASS_OK;
}
