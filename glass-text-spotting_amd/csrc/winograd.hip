// Winograd F(2x2, 3x3) convolution on the fp32 matrix cores of gfx950 (3x3, stride 1, pad 1, NHWC fp32).
//
// fp32 has no reduced-precision MFMA shortcut on CDNA4 (no TF32), so past ~85% of the 157 TFLOP/s matrix
// peak the only way to make the 3x3 layers (78% of the path's conv time) faster is to do fewer multiplies:
// F(2x2,3x3) needs 16 instead of 36 MACs per 2x2 output tile and channel pair (2.25x).
//
//   Y(2x2) = At [ sum_c  (G g G^t)  .  (Bt d B) ] A         per output tile, over input channels c
//
// which is 16 independent GEMMs  M_xi[tile][cout] = sum_c V_xi[tile][c] * U_xi[c][cout],  xi = 0..15.
//
// Work decomposition (one workgroup = 256 threads = 4 waves, ONE workgroup per CU):
//   * block tile = 64 output tiles (linear over n, tile-row, tile-col) x 64 output channels x all 16 xi;
//     wave w owns the four xi = 4w..4w+3 (row w of the 4x4 transform domain): 4 xi x (2x2 MFMA blocks of
//     32x32) = 256 accumulator registers per lane (the whole AGPR half of the unified 512-entry file).
//   * per k-tile of 16 input channels: thread (tile, 4-channel chunk) fetches the raw 4x4 input patch with
//     16 bounds-checked buffer_load_dwordx4 (the image border / padding is the hardware's out-of-range
//     zero fill; the 16 offsets are loop invariant, only the scalar offset advances), applies Bt d B in
//     registers (128 VALU ops) and writes the 16 transformed float4 to LDS  V[xi][tile][16+4].
//   * the weight operand never touches LDS: U is pre-packed (glass_winograd_pack_weights) in exact MFMA
//     B-fragment order, so each wave streams its own xi's fragments L2 -> registers with fully coalesced
//     1 KiB loads, double buffered one 8-channel group ahead.  No two waves of a block read the same bytes.
//   * 128 MFMAs (8192 cycles) per wave per k-tile against ~150 VALU + 16 LDS writes + 16 LDS reads + 32
//     global loads; the next k-tile's patch loads are in flight during the MFMAs.
//   * epilogue: each wave folds its row of xi with the column half of At . A in registers, the four rows
//     meet in LDS, and threads (tile, 4-channel chunk) finish At . , add bias / residual, ReLU and store
//     float4 rows (same epilogue semantics as glass_conv2d_nhwc).
#include "common.h"
#include <cstdint>
#include <cstdlib>
#include <type_traits>
#include <utility>

#include "wino_common.h"

namespace {

constexpr int WT = 64;                        // output tiles (2x2 pixels each) per block
constexpr int WN = 64;                        // output channels per block
constexpr int WK = 16;                        // input channels per k-tile
constexpr int VROW = WK;                      // unpadded V row (64 B); 16-byte slots are XOR-swizzled instead
constexpr int ZLD = WN + 4;
constexpr int V_FLOATS = 16 * WT * VROW;      // 64 KiB per stage, two stages
constexpr int Z_FLOATS = 4 * 2 * WT * ZLD;    // 136 KiB
constexpr int WINO_LDS_BYTES = (2 * V_FLOATS > Z_FLOATS ? 2 * V_FLOATS : Z_FLOATS) * 4;

__global__ __launch_bounds__(256, 1) void conv3x3_wino_f32(WinoParams p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];

  // XCD-aware tile map (see conv.hip): contiguous run of logical tiles per XCD, cout-blocks innermost so the
  // blocks that share an input patch run side by side on one L2
  const int nblk = gridDim.x, bid = blockIdx.x;
  const int xcd = bid & 7, q = nblk >> 3, r8 = nblk & 7;
  const int logical = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + (bid >> 3);
  const int tile_m = logical / p.tiles_n;
  const int tile_n = logical - tile_m * p.tiles_n;
  const int t0 = tile_m * WT, n0 = tile_n * WN;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tpi = p.TH * p.TW;

  __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, (int)p.x_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t ur = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.u), 0, (int)p.u_bytes, 0x00020000);

  // ---- input role: thread = (tile tl, 4-channel chunk) ----
  const int chunk = tid & 3, tl = tid >> 2;
  unsigned voff[16];
  {
    const int t = t0 + tl;
    const bool tv = t < p.ntiles;
    const int n = fast_div(t, tpi, p.magic_tpi);
    const int rem = t - n * tpi;
    const int th = fast_div(rem, p.TW, p.magic_tw);
    const int tw = rem - th * p.TW;
    const int h0 = 2 * th - 1, w0 = 2 * tw - 1;
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int hi = h0 + r, wi = w0 + c;
        const bool ok = tv && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
        voff[r * 4 + c] = ok ? (unsigned)((((n * p.H + hi) * p.W + wi) * p.ldx + chunk * 4) * 4) : OOB;
      }
  }
  float4 d[16], t[16];
  auto load_patch1 = [&](int kt, int i, __amdgpu_buffer_rsrc_t rs) {
    d[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff[i], kt * (WK * 4), 0));
  };
  auto load_patch = [&](int kt) {
#pragma unroll
    for (int i = 0; i < 16; ++i) load_patch1(kt, i, xr);
  };
  // V[stage][xi][tile][16]: a row is 64 bytes, so four consecutive rows span the 64 banks once; XOR-ing the
  // 16-byte slot with (tile/4)%4 spreads the 16 rows a ds_read_b128 service group touches over all 16 slots.
  float* vdst = smem + tl * VROW + ((chunk ^ ((tl >> 2) & 3)) * 4);
  // Bt d B in 20 pieces so that the main loop can meter it out between MFMAs:
  auto row_piece = [&](int c) {              // t[.][c] = Bt . d[.][c]
    t[0 + c] = sub4(d[0 + c], d[8 + c]);
    t[4 + c] = add4(d[4 + c], d[8 + c]);
    t[8 + c] = sub4(d[8 + c], d[4 + c]);
    t[12 + c] = sub4(d[4 + c], d[12 + c]);
  };
  auto col_piece = [&](int stage, int i, int j) {   // V[i][j] = (t . B)[i][j]  -> LDS
    const float4 v = j == 0 ? sub4(t[i * 4 + 0], t[i * 4 + 2]) : j == 1 ? add4(t[i * 4 + 1], t[i * 4 + 2])
                   : j == 2 ? sub4(t[i * 4 + 2], t[i * 4 + 1]) : sub4(t[i * 4 + 1], t[i * 4 + 3]);
    *reinterpret_cast<float4*>(vdst + stage * V_FLOATS + (i * 4 + j) * (WT * VROW)) = v;
  };
  auto transform_store = [&](int stage) {
#pragma unroll
    for (int c = 0; c < 4; ++c) row_piece(c);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) col_piece(stage, i, j);
  };

  // ---- MFMA role: wave wv owns xi = 4 wv + j ----
  f32x16 acc[4][2][2];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][mb][nb][e] = 0.f;

  const int a_sw = ((lane & 31) >> 2) & 3;
  const float* a_frag_g0 = smem + (4 * wv * WT + (lane & 31)) * VROW + (((lane >> 5)) ^ a_sw) * 4;
  const float* a_frag_g1 = smem + (4 * wv * WT + (lane & 31)) * VROW + ((2 + (lane >> 5)) ^ a_sw) * 4;
  const unsigned b_voff = (unsigned)lane * 16u;
  float4 bq[2][4][2];
  // packed U: [tile_n][kt][xi][nb][g] chunks of 1 KiB (64 lanes x float4)
  auto load_b1r = [&](float4 (&dst)[4][2], int kt, int g, int idx, __amdgpu_buffer_rsrc_t rs) {
    const int base = ((((tile_n * p.nk + kt) * 16 + 4 * wv) * 2) * 2 + g) * 1024;
    const int j = idx >> 1, nb = idx & 1;
    dst[j][nb] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, b_voff, base + (j * 4 + nb * 2) * 1024, 0));
  };
  auto load_b1 = [&](float4 (&dst)[4][2], int kt, int g, int idx) { load_b1r(dst, kt, g, idx, ur); };
  auto load_b = [&](float4 (&dst)[4][2], int kt, int g) {
#pragma unroll
    for (int idx = 0; idx < 8; ++idx) load_b1(dst, kt, g, idx);
  };

  // Software pipeline (one barrier per k-tile, V double buffered):
  //   g = 0 of k-tile kt:  MFMAs on V[kt&1]  ||  weight fragments for g = 1  ||  Bt d B of patch kt+1 -> V[(kt+1)&1]
  //                                            ||  then the patch loads of k-tile kt+2
  //   g = 1 of k-tile kt:  MFMAs on V[kt&1]  ||  weight fragments for kt+1
  // A burst of vector-memory instructions fills the CU's address queue and the wave then sits on its next
  // load instead of issuing the next MFMA (all four waves burst together after the barrier), so the loads,
  // the transform's VALU work and its LDS writes are metered out between the MFMAs: the source below IS
  // the issue order, pinned chunk by chunk with sched_barrier(0).
  load_patch(0);
  load_b(bq[0], 0, 0);
  transform_store(0);
  load_patch(p.nk > 1 ? 1 : 0);
  __syncthreads();
  for (int kt = 0; kt < p.nk; ++kt) {
    const int cur = kt & 1;
    const int ktn = kt + 1 < p.nk ? kt + 1 : kt;     // clamped at the end: harmless re-reads keep the loop one block
    const int ktnn = kt + 2 < p.nk ? kt + 2 : kt;
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      const float* af = (g == 0 ? a_frag_g0 : a_frag_g1) + cur * V_FLOATS;
      float4 a[4][2];
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
          a[j][mb] = *reinterpret_cast<const float4*>(af + (j * WT + mb * 32) * VROW);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int m = 0; m < 64; ++m) {
        // side work issued BEFORE MFMA m
        if (m < 16) {
          if ((m & 1) == 0) {                                   // 8 weight-fragment loads, one per 2 MFMAs
            if (g == 0) load_b1(bq[1], kt, 1, m >> 1); else load_b1(bq[0], ktn, 0, m >> 1);
            __builtin_amdgcn_sched_barrier(0);
          }
        } else if (g == 0) {
          // transform of patch kt+1: 4 row pieces (every 3rd MFMA from 16), then 16 column pieces (every 2nd from 28)
          if (m < 28) {
            if ((m - 16) % 3 == 0) { row_piece((m - 16) / 3); __builtin_amdgcn_sched_barrier(0); }
          } else if (m < 60) {
            const int q = (m - 28) >> 1;
            if (((m - 28) & 1) == 0) col_piece(cur ^ 1, q >> 2, q & 3);
            else load_patch1(ktnn, q, xr);        // d is free again: patch of k-tile kt+2, a whole g ahead of its use
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        const int s_ = m >> 4, j = (m >> 2) & 3, mb = (m >> 1) & 1, nb = m & 1;
        acc[j][mb][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(comp(a[j][mb], s_), comp(bq[g][j][nb], s_), acc[j][mb][nb], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();     // V[cur] fully read, V[cur^1] fully written
  }

  // ---- epilogue ----
  // Y = At M A with At = [[1,1,1,0],[0,1,-1,-1]].  This wave holds row i = wv of M (its four xi are the
  // columns j): fold the columns first, Z[i][b] = sum_j M[i][j] At[b][j], then meet the other rows in LDS.
  // C/D layout of the 32x32 MFMA: col = lane&31, row = (e&3) + 8*(e>>2) + 4*(lane>>5).
  // Output role: thread = (4-channel chunk c4, tile ts + 16*pass).  The residual float4s of all 16 output pixels
  // of this thread are requested here, before the LDS exchange, address-selected so that the loads are
  // unconditional and all in flight together (one HBM round trip per block instead of four in series).
  // 32-bit buffer addressing: a pixel that does not exist (ragged last tile block, odd H / W) gets the out-of-range
  // offset - its residual load returns zeros and its store is dropped by the hardware; no branches, no 64-bit math.
  const int c4 = tid & 15, ts = tid >> 4;
  const int co = n0 + c4 * 4;
  float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);    // requested now, used after the exchange
  if (p.bias != nullptr) bv = *reinterpret_cast<const float4*>(p.bias + co);
  __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, (int)p.y_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.res_mode == 1 ? p.res : p.y), 0,
                                                                 (int)(p.res_mode == 1 ? p.r_bytes : 0u), 0x00020000);
  const unsigned ldy4 = (unsigned)p.ldy * 4u, ldr4 = (unsigned)p.ldr * 4u;
  const unsigned rowy = (unsigned)p.W * ldy4, rowr = (unsigned)p.W * ldr4;
  unsigned yoff[16];                              // [pass][b][a]: byte offset of the output pixel, OOB if it does not exist
  float4 rres[16];
#pragma unroll
  for (int pass = 0; pass < 4; ++pass) {
    const int t = t0 + pass * 16 + ts;
    const bool tv = t < p.ntiles;
    const int n = fast_div(t, tpi, p.magic_tpi);
    const int rem = t - n * tpi;
    const int th = fast_div(rem, p.TW, p.magic_tw);
    const int tw = rem - th * p.TW;
    const unsigned pix = (unsigned)((n * p.H + 2 * th) * p.W + 2 * tw);
    const unsigned yb = pix * ldy4 + (unsigned)(p.ycoff + co) * 4u;
    const unsigned rb = pix * ldr4 + (unsigned)co * 4u;
    const bool h1 = 2 * th + 1 < p.H, w1 = 2 * tw + 1 < p.W;
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        const bool ok = tv && (a == 0 || h1) && (b == 0 || w1);
        yoff[pass * 4 + b * 2 + a] = ok ? yb + (a ? rowy : 0u) + (b ? ldy4 : 0u) : OOB;
        if (p.res_mode == 1)
          rres[pass * 4 + b * 2 + a] = __builtin_bit_cast(
              float4, __builtin_amdgcn_raw_buffer_load_b128(rr, ok ? rb + (a ? rowr : 0u) + (b ? ldr4 : 0u) : OOB, 0, 0));
      }
  }
  float* zs = smem;
#pragma unroll
  for (int mb = 0; mb < 2; ++mb)
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = mb * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
        const int col = nb * 32 + (lane & 31);
        const float m0 = acc[0][mb][nb][e], m1 = acc[1][mb][nb][e], m2 = acc[2][mb][nb][e], m3 = acc[3][mb][nb][e];
        zs[((wv * 2 + 0) * WT + row) * ZLD + col] = m0 + m1 + m2;
        zs[((wv * 2 + 1) * WT + row) * ZLD + col] = m1 - m2 - m3;
      }
  __syncthreads();
  // ReLU before (2) / after (1) the residual add as an unconditional max: max(x, qNaN) = x keeps "no ReLU" exact
  const float lo2 = p.relu == 2 ? 0.f : __builtin_nanf(""), lo1 = p.relu == 1 ? 0.f : __builtin_nanf("");
#pragma unroll
  for (int pass = 0; pass < 4; ++pass) {
    const int tile = pass * 16 + ts;
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const float4 z0 = *reinterpret_cast<const float4*>(&zs[((0 * 2 + b) * WT + tile) * ZLD + c4 * 4]);
      const float4 z1 = *reinterpret_cast<const float4*>(&zs[((1 * 2 + b) * WT + tile) * ZLD + c4 * 4]);
      const float4 z2 = *reinterpret_cast<const float4*>(&zs[((2 * 2 + b) * WT + tile) * ZLD + c4 * 4]);
      const float4 z3 = *reinterpret_cast<const float4*>(&zs[((3 * 2 + b) * WT + tile) * ZLD + c4 * 4]);
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        float4 v = a == 0 ? add4(add4(z0, z1), z2) : sub4(sub4(z1, z2), z3);
        v = add4(v, bv);
        v.x = fmaxf(v.x, lo2); v.y = fmaxf(v.y, lo2); v.z = fmaxf(v.z, lo2); v.w = fmaxf(v.w, lo2);
        if (p.res_mode == 1) v = add4(v, rres[pass * 4 + b * 2 + a]);
        v.x = fmaxf(v.x, lo1); v.y = fmaxf(v.y, lo1); v.z = fmaxf(v.z, lo1); v.w = fmaxf(v.w, lo1);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), yr, yoff[pass * 4 + b * 2 + a], 0, 0);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Wide variant for Cout % 128 == 0, Cin % 32 == 0 (the 256-channel layers = most of the path's 3x3 time):
// block tile = 32 output tiles x 128 output channels x 16 xi, k-tile = 32 input channels.  Same 256 accumulators
// (wave w: xi = 4w..4w+3, each 1 x 4 MFMA blocks of 32x32), same 128 KiB of V, but every transformed input value
// now feeds 128 instead of 64 output channels: per MFMA the transform's VALU work, the patch loads and the LDS
// writes / reads are halved; the weight-fragment loads double (they were the cheap part: fully coalesced 1 KiB
// L2 reads).  Weight fragments stream j-major through a 4-slot ring (slot = j; the fragments of the group three
// ahead are requested while a group's 16 MFMAs issue), which keeps them at 64 registers.
constexpr int T2 = 32;                        // output tiles per block
constexpr int N2 = 128;                       // output channels per block
constexpr int K2 = 32;                        // input channels per k-tile
constexpr int V2_FLOATS = 16 * T2 * K2;       // 64 KiB per stage
constexpr int ZLD2 = N2 + 4;
constexpr int Z2_FLOATS = 4 * 2 * T2 * ZLD2;  // 132 KiB
constexpr int WINO128_LDS_BYTES = (2 * V2_FLOATS > Z2_FLOATS ? 2 * V2_FLOATS : Z2_FLOATS) * 4;

__global__ __launch_bounds__(256, 1) void conv3x3_wino128_f32(WinoParams p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int nblk = gridDim.x, bid = blockIdx.x;
  const int xcd = bid & 7, q8 = nblk >> 3, r8 = nblk & 7;
  const int logical = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  const int tile_m = logical / p.tiles_n;
  const int tile_n = logical - tile_m * p.tiles_n;
  const int t0 = tile_m * T2, n0 = tile_n * N2;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tpi = p.TH * p.TW;

  __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, (int)p.x_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t ur = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.u), 0, (int)p.u_bytes, 0x00020000);

  // ---- input role: thread = (tile tl, 4-channel chunk of the 32) ----
  const int chunk = tid & 7, tl = tid >> 3;
  unsigned voff[16];
  {
    const int t = t0 + tl;
    const bool tv = t < p.ntiles;
    const int n = fast_div(t, tpi, p.magic_tpi);
    const int rem = t - n * tpi;
    const int th = fast_div(rem, p.TW, p.magic_tw);
    const int tw = rem - th * p.TW;
    const int h0 = 2 * th - 1, w0 = 2 * tw - 1;
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int hi = h0 + r, wi = w0 + c;
        const bool ok = tv && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
        voff[r * 4 + c] = ok ? (unsigned)((((n * p.H + hi) * p.W + wi) * p.ldx + chunk * 4) * 4) : OOB;
      }
  }
  float4 d[16], t[16];
  auto load_patch1 = [&](int kt, int i) {
    d[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(xr, voff[i], kt * (K2 * 4), 0));
  };
  // V[stage][xi][tile][32]: a row is 128 bytes = half of the 64 banks; the 16-byte slot is XOR-ed with (tile/2)%8,
  // so the 16 rows a ds_read_b128 / ds_write_b128 service group touches land on 16 different slots of the 256 bytes.
  float* vdst = smem + tl * K2 + ((chunk ^ ((tl >> 1) & 7)) * 4);
  auto row_piece = [&](int c) {
    t[0 + c] = sub4(d[0 + c], d[8 + c]);
    t[4 + c] = add4(d[4 + c], d[8 + c]);
    t[8 + c] = sub4(d[8 + c], d[4 + c]);
    t[12 + c] = sub4(d[4 + c], d[12 + c]);
  };
  auto col_piece = [&](int stage, int i, int j) {
    const float4 v = j == 0 ? sub4(t[i * 4 + 0], t[i * 4 + 2]) : j == 1 ? add4(t[i * 4 + 1], t[i * 4 + 2])
                   : j == 2 ? sub4(t[i * 4 + 2], t[i * 4 + 1]) : sub4(t[i * 4 + 1], t[i * 4 + 3]);
    *reinterpret_cast<float4*>(vdst + stage * V2_FLOATS + (i * 4 + j) * (T2 * K2)) = v;
  };

  // ---- MFMA role: wave wv owns xi = 4 wv + j, tiles 0..31 x channels nb*32.. ----
  f32x16 acc[4][4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int nb = 0; nb < 4; ++nb)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[j][nb][e] = 0.f;

  const int a_sw = ((lane & 31) >> 1) & 7;
  const float* a_base = smem + (4 * wv * T2 + (lane & 31)) * K2;
  const int a_kh = lane >> 5;
  const unsigned b_voff = (unsigned)lane * 16u;
  float4 bq[4][4];                           // [slot = j][nb]
  // packed U: [tile_n][kt][xi][g][nb] chunks of 1 KiB (64 lanes x float4)
  auto load_b1 = [&](int kt, int g, int j, int nb) {
    const int base = (((tile_n * p.nk + kt) * 16 + 4 * wv + j) * 16 + g * 4 + nb) * 1024;
    bq[j][nb] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(ur, b_voff, base, 0));
  };

  // prologue: first patch -> V[0], fragments of the first three groups, second patch in flight
#pragma unroll
  for (int i = 0; i < 16; ++i) load_patch1(0, i);
#pragma unroll
  for (int j = 0; j < 3; ++j)
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) load_b1(0, 0, j, nb);
#pragma unroll
  for (int c = 0; c < 4; ++c) row_piece(c);
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) col_piece(0, i, j);
  {
    const int k1 = p.nk > 1 ? 1 : 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) load_patch1(k1, i);
  }
  __syncthreads();

  for (int kt = 0; kt < p.nk; ++kt) {
    const int cur = kt & 1;
    const int ktn = kt + 1 < p.nk ? kt + 1 : kt;
    const int ktnn = kt + 2 < p.nk ? kt + 2 : kt;
    float4 a[2][4];
    auto read_a = [&](auto g_) {
      constexpr int g = decltype(g_)::value;
      const float* af = a_base + cur * V2_FLOATS + (((2 * g + a_kh) ^ a_sw) * 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) a[g & 1][j] = *reinterpret_cast<const float4*>(af + j * (T2 * K2));
    };
    read_a(ic<0>{});
    static_for<4>([&](auto g_) {
      constexpr int g = decltype(g_)::value;
      __builtin_amdgcn_sched_barrier(0);
      static_for<64>([&](auto mm_) {
        constexpr int mm = decltype(mm_)::value;
        constexpr int M = g * 64 + mm;             // position in the k-tile's 256 MFMAs
        constexpr int j = mm >> 4, s_ = (mm >> 2) & 3, nb = mm & 3;
        // side work issued BEFORE MFMA M
        if constexpr ((mm & 3) == 0) {             // weight fragments of the group three ahead, one load per 4 MFMAs
          constexpr int u3 = g * 4 + j + 3;        // group index within (this, next) k-tile
          constexpr int g3 = (u3 >> 2) & 3, j3 = u3 & 3;
          load_b1(u3 >= 16 ? ktn : kt, g3, j3, s_);
          __builtin_amdgcn_sched_barrier(0);
        } else if constexpr ((M & 7) == 2 && M >= 10 && M < 40) {
          row_piece((M - 10) >> 3);                // Bt d B of patch kt+1 -> V[cur^1]: M = 10, 18, 26, 34
          __builtin_amdgcn_sched_barrier(0);
        } else if constexpr ((M & 7) == 2 && M >= 42 && M < 42 + 128) {
          col_piece(cur ^ 1, ((M - 42) >> 3) >> 2, ((M - 42) >> 3) & 3);       // M = 42 .. 162
          __builtin_amdgcn_sched_barrier(0);
        } else if constexpr ((M & 7) == 6 && M >= 46 && M < 46 + 128) {
          load_patch1(ktnn, (M - 46) >> 3);        // d is free again: patch of k-tile kt+2
          __builtin_amdgcn_sched_barrier(0);
        } else if constexpr (g < 3 && mm == 49) {
          read_a(ic<g + 1>{});                     // next group's A fragments, a quarter group ahead
          __builtin_amdgcn_sched_barrier(0);
        }
        acc[j][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(comp(a[g & 1][j], s_), comp(bq[j][nb], s_), acc[j][nb], 0, 0, 0);
      });
      __builtin_amdgcn_sched_barrier(0);
    });
    __syncthreads();
  }

  // ---- epilogue (same algebra as conv3x3_wino_f32; 32 tiles x 128 channels) ----
  // 32-bit buffer addressing throughout: a pixel that does not exist (ragged last tile block, odd H / W) gets the
  // out-of-range offset, so its residual load returns zeros and its store is dropped by the hardware - no branches,
  // no 64-bit address arithmetic, one multiply per pass (the epilogue is ~2600 instructions of pure overhead).
  const int c4 = tid & 31, ts = tid >> 5;
  const int co = n0 + c4 * 4;
  float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
  if (p.bias != nullptr) bv = *reinterpret_cast<const float4*>(p.bias + co);
  __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, (int)p.y_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.res_mode == 1 ? p.res : p.y), 0,
                                                                 (int)(p.res_mode == 1 ? p.r_bytes : 0u), 0x00020000);
  const unsigned ldy4 = (unsigned)p.ldy * 4u, ldr4 = (unsigned)p.ldr * 4u;
  const unsigned rowy = (unsigned)p.W * ldy4, rowr = (unsigned)p.W * ldr4;
  unsigned yoff[16];                              // [pass][b][a]: byte offset of the output pixel, OOB if it does not exist
  float4 rres[16];
  // tile -> pixel offsets of one pass: computed TWICE (mul-high divisions, ~40 instructions) - for the residual loads before
  // the LDS transposition and for the output offsets after it.  Kept across the transposition, the sixteen output offsets
  // pushed the register allocation over: 8 registers went to scratch and the scratch stores showed up as 12.5 % more HBM
  // writes than the output tensor (PMC WRITE_SIZE, scripts/exp_write_size.py; round 5).
  auto pass_pixels = [&](int pass, unsigned ld4, unsigned row4, unsigned coff, unsigned (&off)[4]) {
    const int tt = t0 + pass * 8 + ts;
    const bool tv = tt < p.ntiles;
    const int n = fast_div(tt, tpi, p.magic_tpi);
    const int rem = tt - n * tpi;
    const int th = fast_div(rem, p.TW, p.magic_tw);
    const int tw = rem - th * p.TW;
    const unsigned pix = (unsigned)((n * p.H + 2 * th) * p.W + 2 * tw);
    const unsigned base = pix * ld4 + coff;
    const bool h1 = 2 * th + 1 < p.H, w1 = 2 * tw + 1 < p.W;
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        const bool ok = tv && (a == 0 || h1) && (b == 0 || w1);
        off[b * 2 + a] = ok ? base + (a ? row4 : 0u) + (b ? ld4 : 0u) : OOB;
      }
  };
  if (p.res_mode == 1) {
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
      unsigned ro[4];
      pass_pixels(pass, ldr4, rowr, (unsigned)co * 4u, ro);
#pragma unroll
      for (int i = 0; i < 4; ++i) rres[pass * 4 + i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rr, ro[i], 0, 0));
    }
  }
  float* zs = smem;
#pragma unroll
  for (int nb = 0; nb < 4; ++nb)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int row = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
      const int col = nb * 32 + (lane & 31);
      const float m0 = acc[0][nb][e], m1 = acc[1][nb][e], m2 = acc[2][nb][e], m3 = acc[3][nb][e];
      zs[((wv * 2 + 0) * T2 + row) * ZLD2 + col] = m0 + m1 + m2;
      zs[((wv * 2 + 1) * T2 + row) * ZLD2 + col] = m1 - m2 - m3;
    }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int pass = 0; pass < 4; ++pass) {
    unsigned yo[4];
    pass_pixels(pass, ldy4, rowy, (unsigned)(p.ycoff + co) * 4u, yo);
#pragma unroll
    for (int i = 0; i < 4; ++i) yoff[pass * 4 + i] = yo[i];
  }
  __syncthreads();
  // ReLU before (2) / after (1) the residual add as an unconditional max: max(x, qNaN) = x keeps "no ReLU" exact,
  // NaN inputs included (max(x, -inf) would turn a NaN into -inf)
  const float lo2 = p.relu == 2 ? 0.f : __builtin_nanf(""), lo1 = p.relu == 1 ? 0.f : __builtin_nanf("");
#pragma unroll
  for (int pass = 0; pass < 4; ++pass) {
    const int tile = pass * 8 + ts;
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const float4 z0 = *reinterpret_cast<const float4*>(&zs[((0 * 2 + b) * T2 + tile) * ZLD2 + c4 * 4]);
      const float4 z1 = *reinterpret_cast<const float4*>(&zs[((1 * 2 + b) * T2 + tile) * ZLD2 + c4 * 4]);
      const float4 z2 = *reinterpret_cast<const float4*>(&zs[((2 * 2 + b) * T2 + tile) * ZLD2 + c4 * 4]);
      const float4 z3 = *reinterpret_cast<const float4*>(&zs[((3 * 2 + b) * T2 + tile) * ZLD2 + c4 * 4]);
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        float4 v = a == 0 ? add4(add4(z0, z1), z2) : sub4(sub4(z1, z2), z3);
        v = add4(v, bv);
        v.x = fmaxf(v.x, lo2); v.y = fmaxf(v.y, lo2); v.z = fmaxf(v.z, lo2); v.w = fmaxf(v.w, lo2);
        if (p.res_mode == 1) v = add4(v, rres[pass * 4 + b * 2 + a]);
        v.x = fmaxf(v.x, lo1); v.y = fmaxf(v.y, lo1); v.z = fmaxf(v.z, lo1); v.w = fmaxf(v.w, lo1);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), yr, yoff[pass * 4 + b * 2 + a], 0, 0);
      }
    }
  }
}

// wide layout: [cout/128][cin/32][xi][g][nb][lane][s]  with  cout = 128 tn + 32 nb + (lane&31),  cin = 32 kt + 8 g + 4 (lane>>5) + s
__global__ void wino128_pack_weights_kernel(const float* __restrict__ w, float* __restrict__ u, int Cout, int Cin) {
  const long total = 16L * Cout * Cin;
  const int nk = Cin / K2;
  for (long o = (long)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (long)gridDim.x * blockDim.x) {
    long r = o;
    const int s = (int)(r & 3); r >>= 2;
    const int lane = (int)(r & 63); r >>= 6;
    const int nb = (int)(r & 3); r >>= 2;
    const int g = (int)(r & 3); r >>= 2;
    const int xi = (int)(r & 15); r >>= 4;
    const int kt = (int)(r % nk);
    const int tn = (int)(r / nk);
    const int co = tn * N2 + nb * 32 + (lane & 31);
    const int ci = kt * K2 + g * 8 + (lane >> 5) * 4 + s;
    const int i = xi >> 2, j = xi & 3;
    const double G[4][3] = {{1.0, 0.0, 0.0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0.0, 0.0, 1.0}};
    double acc = 0.0;
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int b = 0; b < 3; ++b) acc += G[i][a] * G[j][b] * (double)w[(((long)co * 3 + a) * 3 + b) * Cin + ci];
    u[o] = (float)acc;
  }
}

// U = G g G^t per (cout, cin), written in the fragment order the kernel streams:
// [cout/64][cin/16][xi][nb][g][lane][s]  with  cout = 64 tn + 32 nb + (lane&31),  cin = 16 kt + 8 g + 4 (lane>>5) + s
__global__ void wino_pack_weights_kernel(const float* __restrict__ w, float* __restrict__ u, int Cout, int Cin) {
  const long total = 16L * Cout * Cin;
  const int nk = Cin / WK;
  for (long o = (long)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (long)gridDim.x * blockDim.x) {
    long r = o;
    const int s = (int)(r & 3); r >>= 2;
    const int lane = (int)(r & 63); r >>= 6;
    const int g = (int)(r & 1); r >>= 1;
    const int nb = (int)(r & 1); r >>= 1;
    const int xi = (int)(r & 15); r >>= 4;
    const int kt = (int)(r % nk);
    const int tn = (int)(r / nk);
    const int co = tn * WN + nb * 32 + (lane & 31);
    const int ci = kt * WK + g * 8 + (lane >> 5) * 4 + s;
    const int i = xi >> 2, j = xi & 3;
    const double G[4][3] = {{1.0, 0.0, 0.0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0.0, 0.0, 1.0}};
    double acc = 0.0;
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int b = 0; b < 3; ++b) acc += G[i][a] * G[j][b] * (double)w[(((long)co * 3 + a) * 3 + b) * Cin + ci];
    u[o] = (float)acc;
  }
}

// the wide kernel takes every layer whose channel counts allow it (GLASS_WINO128=0: 64x64 blocks everywhere)
bool wino_wide(int Cout, int Cin) {
  static const bool enabled = [] { const char* e = getenv("GLASS_WINO128"); return !(e && e[0] == '0'); }();
  return enabled && Cout % N2 == 0 && Cin % K2 == 0;
}

}  // namespace

extern "C" int glass_winograd_supported(const glass_conv_desc* d) {
  if (!d) return 0;
  const long xb = (long)d->N * d->H * d->W * d->ldx * 4;
  const long yb = (long)d->N * d->H * d->W * d->ldy * 4, rb = d->res_mode == 1 ? (long)d->N * d->H * d->W * d->ldr * 4 : 0;
  return d->KH == 3 && d->KW == 3 && d->stride_h == 1 && d->stride_w == 1 && d->pad_h == 1 && d->pad_w == 1 &&
         d->Cin % WK == 0 && d->Cout % WN == 0 && d->ldx % 4 == 0 && d->ldx >= d->Cin && d->y_cstride == 1 && d->ldy % 4 == 0 &&
         d->y_coff % 4 == 0 && d->y_coff >= 0 && d->y_coff + d->Cout <= d->ldy &&
         (d->res_mode == 0 || (d->res_mode == 1 && d->ldr % 4 == 0 && d->ldr >= d->Cout)) && xb < 0x7fffff00L && yb < 0x7fffff00L && rb < 0x7fffff00L &&
         d->Ho == d->H && d->Wo == d->W;
}

extern "C" int glass_winograd_block_channels(int Cout, int Cin) { return wino_wide(Cout, Cin) ? N2 : WN; }

extern "C" size_t glass_winograd_weight_floats(int Cout, int Cin) { return (size_t)16 * (size_t)Cout * (size_t)Cin; }

extern "C" int glass_winograd_pack_weights(const float* w, int Cout, int Cin, float* u_packed, glass_stream_t stream) {
  GLASS_CHECK_ARG(w && u_packed, "glass_winograd_pack_weights: null pointer");
  GLASS_CHECK_ARG(Cout > 0 && Cin > 0 && Cout % WN == 0 && Cin % WK == 0,
                  "glass_winograd_pack_weights: Cout=%d must be a multiple of 64 and Cin=%d a multiple of 16", Cout, Cin);
  const long total = 16L * Cout * Cin;
  const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  if (wino_wide(Cout, Cin))
    hipLaunchKernelGGL(wino128_pack_weights_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, u_packed, Cout, Cin);
  else
    hipLaunchKernelGGL(wino_pack_weights_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, u_packed, Cout, Cin);
  GLASS_CHECK_LAUNCH("glass_winograd_pack_weights");
  return GLASS_OK;
}

static int wino_launch(const glass_conv_desc* d, const float* x, const float* u_packed, const float* bias, const float* residual,
                       float* y, glass_stream_t stream, bool body_only);

extern "C" int glass_conv3x3_winograd_nhwc(const glass_conv_desc* d, const float* x, const float* u_packed,
                                           const float* bias, const float* residual, float* y, glass_stream_t stream) {
  return wino_launch(d, x, u_packed, bias, residual, y, stream, false);
}

// Only the FULL 2-column tile columns: output columns [0, 2 * (W / 2)) of every row (the F(2x2) sibling of
// glass_conv3x3_winograd43_body_nhwc).  A map of odd width - the local extractor's 16 x 33 maps, reference
// local_feature_extraction.py:123 - otherwise pays a 17th tile column for one pixel column, and with one image in flight
// (32 RoIs: 17 x 8 x 32 = 4352 tiles = 272 workgroups of 32 tiles x 128 channels) that extra column is what turns ONE round
// on the 256 CUs into two: 143 -> 75 us per layer + the strip.  The caller computes the last column with
// glass_conv2d_nhwc on the 2-column strip (ops/native.py _last_column_strip).
extern "C" int glass_conv3x3_winograd_body_nhwc(const glass_conv_desc* d, const float* x, const float* u_packed,
                                                const float* bias, const float* residual, float* y, glass_stream_t stream) {
  GLASS_CHECK_ARG(d && d->W >= 2, "glass_conv3x3_winograd_body_nhwc: needs W >= 2");
  return wino_launch(d, x, u_packed, bias, residual, y, stream, true);
}

static int wino_launch(const glass_conv_desc* d, const float* x, const float* u_packed, const float* bias, const float* residual,
                       float* y, glass_stream_t stream, bool body_only) {
  GLASS_CHECK_ARG(d && x && u_packed && y, "glass_conv3x3_winograd_nhwc: null pointer");
  GLASS_CHECK_ARG(glass_winograd_supported(d),
                  "glass_conv3x3_winograd_nhwc: needs 3x3/stride 1/pad 1, Cin%%16==0, Cout%%64==0, unit channel stride, "
                  "res_mode 0/1 and x < 2 GiB (got Cin=%d Cout=%d k=%dx%d s=%d p=%d)", d->Cin, d->Cout, d->KH, d->KW,
                  d->stride_h, d->pad_h);
  GLASS_CHECK_ARG(d->res_mode == 0 || residual != nullptr, "glass_conv3x3_winograd_nhwc: res_mode set but residual is null");
  GLASS_CHECK_ARG(((uintptr_t)x & 15) == 0 && ((uintptr_t)u_packed & 15) == 0 && ((uintptr_t)y & 15) == 0 &&
                      (bias == nullptr || ((uintptr_t)bias & 15) == 0) && (residual == nullptr || ((uintptr_t)residual & 15) == 0),
                  "glass_conv3x3_winograd_nhwc: pointers must be 16-byte aligned");
  if (d->N == 0) return GLASS_OK;
  WinoParams p;
  p.x = x; p.u = u_packed; p.bias = bias; p.res = residual; p.y = y; p.dbg = nullptr;
  p.N = d->N; p.H = d->H; p.W = d->W; p.Cin = d->Cin; p.Cout = d->Cout;
  p.TH = (d->H + 1) / 2; p.TW = body_only ? d->W / 2 : (d->W + 1) / 2;      // (input / output bounds still use the true W)
  const long nt = (long)d->N * p.TH * p.TW;
  GLASS_CHECK_ARG(nt < 0x7fffffffL, "glass_conv3x3_winograd_nhwc: too many tiles");
  p.ntiles = (int)nt;
  const bool wide = wino_wide(d->Cout, d->Cin);
  p.nk = d->Cin / (wide ? K2 : WK);
  p.ldx = d->ldx; p.ldy = d->ldy; p.ycoff = d->y_coff; p.ldr = d->ldr; p.relu = d->relu; p.res_mode = d->res_mode;
  p.tiles_m = cdiv(p.ntiles, wide ? T2 : WT);
  p.tiles_n = d->Cout / (wide ? N2 : WN);
  p.x_bytes = (unsigned)((long)d->N * d->H * d->W * d->ldx * 4);
  p.magic_tpi = (unsigned)(0x100000000ULL / (unsigned long long)(p.TH * p.TW));
  p.magic_tw = (unsigned)(0x100000000ULL / (unsigned long long)p.TW);
  p.u_bytes = (unsigned)(16L * d->Cout * d->Cin * 4);
  p.y_bytes = (unsigned)((long)d->N * d->H * d->W * d->ldy * 4);
  p.r_bytes = d->res_mode == 1 ? (unsigned)((long)d->N * d->H * d->W * d->ldr * 4) : 0u;
  const long nblk = (long)p.tiles_m * p.tiles_n;
  GLASS_CHECK_ARG(nblk > 0 && nblk <= 0x7fffffffL, "glass_conv3x3_winograd_nhwc: bad grid");
  static int attr_rc = (int)hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_wino_f32),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, WINO_LDS_BYTES);
  static int attr_rc2 = (int)hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_wino128_f32),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, WINO128_LDS_BYTES);
  if (attr_rc != 0 || attr_rc2 != 0) {
    glass_set_error("glass_conv3x3_winograd_nhwc: cannot reserve %d bytes of LDS (hip error %d / %d)", WINO_LDS_BYTES, attr_rc, attr_rc2);
    return GLASS_EHIP;
  }
  if (wide)
    hipLaunchKernelGGL(conv3x3_wino128_f32, dim3((unsigned)nblk), dim3(256), WINO128_LDS_BYTES, (hipStream_t)stream, p);
  else
    hipLaunchKernelGGL(conv3x3_wino_f32, dim3((unsigned)nblk), dim3(256), WINO_LDS_BYTES, (hipStream_t)stream, p);
  GLASS_CHECK_LAUNCH("glass_conv3x3_winograd_nhwc");
  return GLASS_OK;
}
