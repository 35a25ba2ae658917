// Implicit-GEMM convolution on the fp32 matrix cores of gfx950 (v_mfma_f32_32x32x2_f32).
//
// GEMM view:  Y[m][co] = sum_k A[m][k] * Wt[co][k],  m = (n,ho,wo) output pixel,
// k = (kh,kw,ci) with ci fastest, so a 32-wide k-tile of A is (for Cin >= 32) 128 contiguous
// bytes of ONE input pixel -> 16-byte coalesced NHWC loads, and Wt rows are K-contiguous.
//
// Block = 256 threads = 4 waves (64 lanes each).  Each wave owns TM x TN tiles of 32x32
// accumulators (f32x16 each, in the unified VGPR/AGPR file).  Both operand tiles are staged
// through LDS as [row][BK+4] floats: the +4 pad makes the 16-byte fragment reads
// (ds_read_b128: lane l reads row l&31, k = 4*(l>>5)..+3) conflict-free (row stride 144 B ->
// 16-B slot index 9*row mod 16 is a bijection over each 16-lane service group).
// One ds_read_b128 per operand feeds 4 MFMA k-steps: step s uses k = 8g+s on lanes 0-31 and
// k = 8g+4+s on lanes 32-63 for A and B alike, so the k-permutation cancels in the sum.
//
// Global->LDS staging goes through registers: the loads of k-tile t+1 are issued before the
// MFMAs of tile t and written to LDS after them, so HBM/L2 latency hides under 64 MFMAs
// (4096 cycles) per wave per tile.
//
// blockIdx -> tile map is XCD-aware: hardware places block b on XCD b%8 (each XCD has its
// own 4 MiB L2), so each XCD gets a contiguous run of tiles whose neighbours share A rows.
#include "splitk_common.h"
#include "common.h"
#include <cstdlib>

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct ConvParams {
  const float* x;
  const float* w;
  const float* bias;
  const float* res;
  float* y;
  int N, H, W, Cin, Cout, KH, KW, sh, sw, ph, pw, Ho, Wo;
  int ldx, ldy, ycoff, ycs, relu, res_mode, ldr;
  int M, Ktot, nk, tiles_m, tiles_n, vec_epi;
  unsigned x_bytes, w_bytes;      // buffer sizes for the bounds-checked load paths (0: tensors too large)
  unsigned y_bytes, r_bytes;      // output / residual spans of the vector epilogue (it is enabled only if they fit 31 bits)
  unsigned magic_cin, magic_kw;   // floor(2^32 / Cin), floor(2^32 / KW) for the per-thread tap decode (MODE 2)
  int ksplit;                     // > 1: split-K launch (glass_conv2d_nhwc_splitk, MODE 1 only): blockIdx.y = k-slice s takes the
  long split_y;                   // k-tiles [s * nk / ksplit, (s + 1) * nk / ksplit) and writes its partial sums to y + s * split_y
  int half_mode;                  // 1: fp16 operands on v_mfma_f32_32x32x16_f16 (glass_conv2d_nhwc_f16)
  int xh, yh, rh;                 // half_mode only (glass_conv2d_nhwc_h16): x / y / residual are fp16 tensors in HBM
};

__device__ __forceinline__ float4 sel4(bool ok, float4 v) {
  // component-wise: a whole-float4 ?: is lowered through scratch memory by hipcc
  return make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f);
}


typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

// MODE 0: plain loads, 64-bit addresses (tensors >= 2 GiB).  MODE 1 ("FAST"): Cin % 32 == 0, buffer loads, uniform
// scalar tap tracking.  MODE 2: any Cin % 4 == 0 (stem, Cin = 4 / 16 first layers), buffer loads, per-thread tap decode
// by mul-high.
// HALF: operands are rounded to fp16 (round to nearest even) when they are staged into LDS and multiplied on
// v_mfma_f32_32x32x16_f16 (fp32 accumulate, 16x the fp32 matrix rate); storage stays fp32.  Opt-in precision mode
// (glass_conv2d_nhwc_f16, BASELINE configs[4]); never used by the fp32 path the headline metric is measured on.
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));

template <int WAVES_M, int WAVES_N, int TM, int TN, int NSTAGE, int MINW, int BK, int MODE, bool HALF>
__global__ __launch_bounds__(256, MINW) void conv_igemm_f32(ConvParams pin) {
  ConvParams p = pin;
  int kt0 = 0, kt1 = pin.nk;       // this workgroup's k-tiles
  if (pin.ksplit > 1) {           // split-K: blockIdx.y's share of the k-tiles; partial sums go to slice blockIdx.y of the workspace
    const int per = pin.nk / pin.ksplit;
    kt0 = (int)blockIdx.y * per;
    kt1 = kt0 + per;
    p.y += (long)blockIdx.y * pin.split_y;
  }
  constexpr bool FAST = MODE == 1, BUF = MODE != 0;
  constexpr int LDS_LD = HALF ? BK + 8 : BK + 4;      // elements (halfs / floats) per staged row incl. the conflict pad
  constexpr int KCH = BK / 4;            // 16-byte chunks per staged row
  constexpr int RPP = 256 / KCH;         // rows staged per pass of the 256 threads
  constexpr int BM = WAVES_M * TM * 32;
  constexpr int BN = WAVES_N * TN * 32;
  constexpr int A_LOADS = BM / RPP;
  constexpr int B_LOADS = BN / RPP;
  // NSTAGE LDS stages: with 2, tile t is read by the MFMAs while tile t+1 is written (one barrier per
  // k-tile); measured neutral vs 1 stage on MI355X (the 2 co-resident blocks already overlap), and one
  // stage (36 KiB) keeps more blocks resident, which shortens the last partial wave of tiles.
  constexpr int STAGE = (BM + BN) * LDS_LD;
  constexpr int ESZ = HALF ? 2 : 4;
  // epilogue staging (fp32, LDS-transposed accumulators) shares the buffer: 64 columns per chunk when they fit
  constexpr int CW = (NSTAGE * STAGE * ESZ >= BM * 68 * 4 && BN >= 64) ? 64 : 32;
  constexpr int SMEM_BYTES = NSTAGE * STAGE * ESZ > BM * (CW + 4) * 4 ? NSTAGE * STAGE * ESZ : BM * (CW + 4) * 4;
  __shared__ __attribute__((aligned(16))) unsigned char smem_raw[SMEM_BYTES];
  float* smem = reinterpret_cast<float*>(smem_raw);
  _Float16* hmem = reinterpret_cast<_Float16*>(smem_raw);

  // bijective XCD swizzle: XCD (bid % 8) owns a contiguous chunk of logical tile ids
  const int nblk = gridDim.x, bid = blockIdx.x;
  const int xcd = bid & 7, q = nblk >> 3, r = nblk & 7;
  const int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  const int tile_m = logical / p.tiles_n;
  const int tile_n = logical - tile_m * p.tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int cc = tid % KCH;  // 16-byte chunk column inside the k-tile
  const int r0 = tid / KCH;  // first staged row of this thread

  long a_base[A_LOADS];
  int a_hi0[A_LOADS], a_wi0[A_LOADS];
  const int HoWo = p.Ho * p.Wo;
#pragma unroll
  for (int i = 0; i < A_LOADS; ++i) {
    const int m = m0 + r0 + RPP * i;
    if (m < p.M) {
      const int n = m / HoWo;
      const int rem = m - n * HoWo;
      const int ho = rem / p.Wo;
      const int wo = rem - ho * p.Wo;
      a_hi0[i] = ho * p.sh - p.ph;
      a_wi0[i] = wo * p.sw - p.pw;
      a_base[i] = ((long)n * p.H * p.W + (long)a_hi0[i] * p.W + a_wi0[i]) * p.ldx;
    } else {
      a_hi0[i] = -(1 << 24);
      a_wi0[i] = 0;
      a_base[i] = 0;
    }
  }
  long b_off[B_LOADS];
  bool b_ok[B_LOADS];
#pragma unroll
  for (int i = 0; i < B_LOADS; ++i) {
    const int co = n0 + r0 + RPP * i;
    b_ok[i] = co < p.Cout;
    b_off[i] = (long)(b_ok[i] ? co : 0) * p.Ktot + cc * 4;
  }

  float4 areg[A_LOADS], breg[B_LOADS];

  // FAST path (Cin % BK == 0, tensors < 2 GiB): a k-tile is ONE filter tap and BK consecutive channels, so
  // (dh, dw, c0) are workgroup-uniform and advance with scalar adds (no integer division in the loop), and
  // the loads are bounds-checked buffer loads: an out-of-image / out-of-range lane just gets a huge offset
  // and the hardware returns zeros (no per-element select).  The VALU work per k-tile drops from ~150 to
  // ~35 instructions per wave, which matters because VALU issue competes with MFMA issue on the SIMD
  // (measured: the same loop without its global loads runs 145 instead of 122 TFLOP/s).
  constexpr unsigned OOB = 0x7fffffffu;
  int a_off[A_LOADS];
  unsigned b_voff[B_LOADS];
  __amdgpu_buffer_rsrc_t xr, wr;
  int f_dh = 0, f_dw = 0, f_c0 = 0;          // uniform: tap row / column, first channel of the current k-tile
  if (FAST && kt0 != 0) {                    // a split-K slice starts in the middle of the (tap, channel) walk
    const int tap = (kt0 * BK) / p.Cin;
    f_c0 = kt0 * BK - tap * p.Cin;
    f_dh = tap / p.KW;
    f_dw = tap - f_dh * p.KW;
  }
  if constexpr (BUF) {
    xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, (int)p.x_bytes, 0x00020000);
    wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.w), 0, (int)p.w_bytes, 0x00020000);
#pragma unroll
    for (int i = 0; i < A_LOADS; ++i) a_off[i] = (int)(a_base[i] * 4) + cc * 16;       // bytes, may be negative
#pragma unroll
    for (int i = 0; i < B_LOADS; ++i) b_voff[i] = b_ok[i] ? (unsigned)(b_off[i] * 4) : OOB;
  }

  // 4 consecutive input channels at fp32 byte offset `off` (OOB -> zeros).  fp16 storage (half_mode, p.xh): the same 4
  // channels are 8 bytes at half the offset; they are widened here and rounded back (exactly) when the tile is staged.
  auto load_x4 = [&](unsigned off) -> float4 {
    if constexpr (HALF) {
      if (p.xh) {
        const h4 v = __builtin_bit_cast(h4, __builtin_amdgcn_raw_buffer_load_b64(xr, off == OOB ? OOB : off >> 1, 0, 0));
        return make_float4((float)v.x, (float)v.y, (float)v.z, (float)v.w);
      }
    }
    return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(xr, off, 0, 0));
  };
  auto load_tile_fast = [&](int kt) {
    const int koff = ((f_dh * p.W + f_dw) * p.ldx + f_c0) * 4;                         // uniform, bytes
#pragma unroll
    for (int i = 0; i < A_LOADS; ++i) {
      const int hi = a_hi0[i] + f_dh, wi = a_wi0[i] + f_dw;
      const bool ok = (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
      const unsigned off = ok ? (unsigned)(a_off[i] + koff) : OOB;
      areg[i] = load_x4(off);
    }
#pragma unroll
    for (int i = 0; i < B_LOADS; ++i)
      breg[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(wr, b_voff[i], kt * (BK * 4), 0));
    // advance the uniform tap/channel position to the next k-tile
    // (branch-free, so the k-loop body stays one scheduling region)
    f_c0 += BK;
    const int wrap_c = f_c0 >= p.Cin ? 1 : 0;
    f_c0 = wrap_c ? 0 : f_c0;
    f_dw += wrap_c;
    const int wrap_w = f_dw == p.KW ? 1 : 0;
    f_dw = wrap_w ? 0 : f_dw;
    f_dh += wrap_w;
  };

  // MODE 2: the k-tile spans several taps (Cin < 32) or straddles them; every thread decodes its own 16-byte chunk
  // (tap, channel) with two mul-high divisions and goes through the same bounds-checked loads (~50 VALU per k-tile
  // instead of ~150 for the plain path: two real integer divisions, 64-bit addresses and a 4-component select per load)
  auto udiv = [](int t, int d, unsigned magic) {
    int q = d == 1 ? t : (int)__umulhi((unsigned)t, magic);
    if (t - q * d >= d) ++q;
    return q;
  };
  auto load_tile_buf = [&](int kt) {
    const int kpos = kt * BK + cc * 4;
    const bool kvalid = kpos < p.Ktot;
    const int tap = udiv(kpos, p.Cin, p.magic_cin);
    const int c = kpos - tap * p.Cin;
    const int dh = udiv(tap, p.KW, p.magic_kw);
    const int dw = tap - dh * p.KW;
    const int koff = ((dh * p.W + dw) * p.ldx + c) * 4 - cc * 16;       // a_off already carries the chunk column
#pragma unroll
    for (int i = 0; i < A_LOADS; ++i) {
      const int hi = a_hi0[i] + dh, wi = a_wi0[i] + dw;
      const bool ok = kvalid && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
      const unsigned off = ok ? (unsigned)(a_off[i] + koff) : OOB;
      areg[i] = load_x4(off);
    }
#pragma unroll
    for (int i = 0; i < B_LOADS; ++i)
      breg[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(wr, kvalid ? b_voff[i] : OOB, kt * (BK * 4), 0));
  };

  auto load_tile_generic = [&](int kt) {
    const int kpos = kt * BK + cc * 4;
    const bool kvalid = kpos < p.Ktot;
    const int tap = kpos / p.Cin;
    const int c = kpos - tap * p.Cin;
    const int dh = tap / p.KW;
    const int dw = tap - dh * p.KW;
    const long koff = ((long)dh * p.W + dw) * p.ldx + c;
#pragma unroll
    for (int i = 0; i < A_LOADS; ++i) {
      const int hi = a_hi0[i] + dh, wi = a_wi0[i] + dw;
      const bool ok = kvalid && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
      // select the ADDRESS (not the load) so the load stays unconditional and pipelined
      const float* src = ok ? (p.x + a_base[i] + koff) : p.x;
      float4 v = *reinterpret_cast<const float4*>(src);
      areg[i] = sel4(ok, v);
    }
#pragma unroll
    for (int i = 0; i < B_LOADS; ++i) {
      const bool ok = kvalid && b_ok[i];
      const float* src = ok ? (p.w + b_off[i] + (long)kt * BK) : p.w;
      float4 v = *reinterpret_cast<const float4*>(src);
      breg[i] = sel4(ok, v);
    }
  };
  auto load_tile = [&](int kt) {
    if constexpr (MODE == 1) load_tile_fast(kt); else if constexpr (MODE == 2) load_tile_buf(kt); else load_tile_generic(kt);
  };
  auto store_tile = [&](int stage) {
    if constexpr (HALF) {
      _Float16* As = hmem + stage * STAGE;
      _Float16* Bs = As + BM * LDS_LD;
      auto to_h4 = [](float4 v) { return h4{(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w}; };   // v_cvt_f16_f32: RNE
#pragma unroll
      for (int i = 0; i < A_LOADS; ++i)
        *reinterpret_cast<h4*>(&As[(r0 + RPP * i) * LDS_LD + cc * 4]) = to_h4(areg[i]);
#pragma unroll
      for (int i = 0; i < B_LOADS; ++i)
        *reinterpret_cast<h4*>(&Bs[(r0 + RPP * i) * LDS_LD + cc * 4]) = to_h4(breg[i]);
    } else {
      float* As = smem + stage * STAGE;
      float* Bs = As + BM * LDS_LD;
#pragma unroll
      for (int i = 0; i < A_LOADS; ++i)
        *reinterpret_cast<float4*>(&As[(r0 + RPP * i) * LDS_LD + cc * 4]) = areg[i];
#pragma unroll
      for (int i = 0; i < B_LOADS; ++i)
        *reinterpret_cast<float4*>(&Bs[(r0 + RPP * i) * LDS_LD + cc * 4]) = breg[i];
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int frag_row = lane & 31;
  const int frag_k = (lane >> 5) * 4;
  const float* a_frag0 = smem + (wm * TM * 32 + frag_row) * LDS_LD + frag_k;
  const float* b_frag0 = smem + BM * LDS_LD + (wn * TN * 32 + frag_row) * LDS_LD + frag_k;
  // fp16: lane l supplies 8 consecutive k of row l&31 starting at 8*(l>>5) (one ds_read_b128 = one MFMA operand)
  const _Float16* a_frag0h = hmem + (wm * TM * 32 + frag_row) * LDS_LD + (lane >> 5) * 8;
  const _Float16* b_frag0h = hmem + BM * LDS_LD + (wn * TN * 32 + frag_row) * LDS_LD + (lane >> 5) * 8;

  load_tile(kt0);
  store_tile(0);
  __syncthreads();
  for (int kt = kt0; kt < kt1; ++kt) {
    const bool more = kt + 1 < kt1;
    // buffer-load modes: unconditional (after the last k-tile the bounds-checked loads just return zeros / unused data),
    // so the whole k-tile is one scheduling region and the loads can be metered out between the MFMAs below
    if (BUF || more) load_tile(kt + 1);
    const int cur = NSTAGE == 2 ? ((kt - kt0) & 1) : 0;
    if constexpr (HALF) {
      const _Float16* a_frag = a_frag0h + cur * STAGE;
      const _Float16* b_frag = b_frag0h + cur * STAGE;
#pragma unroll
      for (int g = 0; g < BK / 16; ++g) {
        h8 af[TM], bf[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const h8*>(a_frag + i * 32 * LDS_LD + g * 16);
#pragma unroll
        for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const h8*>(b_frag + j * 32 * LDS_LD + g * 16);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i], bf[j], acc[i][j], 0, 0, 0);
      }
    } else {
    const float* a_frag = a_frag0 + cur * STAGE;
    const float* b_frag = b_frag0 + cur * STAGE;
#pragma unroll
    for (int g = 0; g < BK / 8; ++g) {
      float4 af[TM], bf[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const float4*>(a_frag + i * 32 * LDS_LD + g * 8);
#pragma unroll
      for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const float4*>(b_frag + j * 32 * LDS_LD + g * 8);
#pragma unroll
      for (int s = 0; s < 4; ++s) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const float a = s == 0 ? af[i].x : s == 1 ? af[i].y : s == 2 ? af[i].z : af[i].w;
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            const float b = s == 0 ? bf[j].x : s == 1 ? bf[j].y : s == 2 ? bf[j].z : bf[j].w;
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i][j], 0, 0, 0);
          }
        }
      }
    }
    }
    if constexpr (BUF) {
      // A burst of vector-memory instructions fills the CU's address queue and the wave then sits on its next
      // load instead of issuing an MFMA; one load per PER MFMAs keeps the queue shallow (measured on the
      // Winograd kernel: +10%).
      constexpr int NL = A_LOADS + B_LOADS, NM = HALF ? (BK / 16) * TM * TN : (BK / 8) * 4 * TM * TN, PER = NM / NL > 0 ? NM / NL : 1;
#pragma unroll
      for (int i = 0; i < NL; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, PER, 0);
      }
    }
    if (NSTAGE == 2) {
      // stage (kt+1)&1 was last read in iteration kt-1, which every wave left through the barrier below
      if (more) store_tile((kt + 1 - kt0) & 1);
      __syncthreads();
    } else {
      __syncthreads();
      if (more) {
        store_tile(0);
        __syncthreads();
      }
    }
  }

  // epilogue.  C/D layout of the 32x32 MFMA: col = lane&31, row = (e&3) + 8*(e>>2) + 4*(lane>>5).
  const int HoWo2 = (p.Ho >> 1) * (p.Wo >> 1);
  if (p.vec_epi) {
    // Vector path (unit channel stride, 16-byte aligned rows): the accumulators are transposed through
    // the (now idle) operand LDS so that every lane handles 4 consecutive channels of one pixel:
    // 16-byte residual loads / output stores, 16 lanes covering 256 contiguous bytes of a row.
    constexpr int SLD = CW + 4;
    constexpr int NCHUNK = BN / CW;
    constexpr int VPR = CW / 4;                    // float4 per staged row
    constexpr int ITERS = BM * VPR / 256;
    float* stage = smem;
    // Residual prefetch: a chunk's residual float4s are requested together before its LDS transposition
    // (address-selected, so the loads are unconditional and all in flight) instead of one dependent HBM round
    // trip per output row inside the store loop.
    // 32-bit buffer addressing (the host enables this path only when the output / residual spans fit 31 bits): a
    // row or channel group past the edge gets the out-of-range offset - its residual load returns zeros, its store is
    // dropped by the hardware - so the loops below carry no branches and no 64-bit address arithmetic, and the row
    // offsets advance by a loop-invariant stride.
    static_assert(256 % VPR == 0, "a thread keeps its channel group across iterations");
    constexpr int RPI = 256 / VPR;                 // rows per iteration
    constexpr unsigned EOOB = 0x7fffffffu;
    __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, (int)p.y_bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.res_mode != 0 ? p.res : p.y), 0,
                                                                   (int)(p.res_mode != 0 ? p.r_bytes : 0u), 0x00020000);
    const int row_t = tid / VPR, c4 = tid - row_t * VPR;
    const unsigned ldy4 = (unsigned)p.ldy * 4u, ldr4 = (unsigned)p.ldr * 4u;
    const float lo2 = p.relu == 2 ? 0.f : __builtin_nanf(""), lo1 = p.relu == 1 ? 0.f : __builtin_nanf("");   // max(x, qNaN) = x
    float4 rres[ITERS];
    auto prefetch_res = [&](int c) {
      const int co = n0 + c * CW + c4 * 4;
      const bool cok = co < p.Cout;
#pragma unroll
      for (int it = 0; it < ITERS; ++it) {
        const int m = m0 + row_t + it * RPI;
        const bool ok = m < p.M && cok;
        unsigned roff;
        if (p.res_mode == 1) {
          roff = (unsigned)m * ldr4 + (unsigned)co * 4u;
        } else {
          const int n = m / HoWo;
          const int rem = m - n * HoWo;
          const int ho = rem / p.Wo;
          const int wo = rem - ho * p.Wo;
          roff = (unsigned)(n * HoWo2 + (ho >> 1) * (p.Wo >> 1) + (wo >> 1)) * ldr4 + (unsigned)co * 4u;
        }
        if (HALF && p.rh) {
          const h4 hv = __builtin_bit_cast(h4, __builtin_amdgcn_raw_buffer_load_b64(rr, ok ? roff >> 1 : EOOB, 0, 0));
          rres[it] = make_float4((float)hv.x, (float)hv.y, (float)hv.z, (float)hv.w);
        } else {
          rres[it] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rr, ok ? roff : EOOB, 0, 0));
        }
      }
    };
#pragma unroll
    for (int c = 0; c < NCHUNK; ++c) {
      if (p.res_mode != 0) prefetch_res(c);        // in flight across the LDS transposition below
      const int co = n0 + c * CW + c4 * 4;
      const bool cok = co < p.Cout;
      float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
      if (p.bias != nullptr && cok) bv = *reinterpret_cast<const float4*>(p.bias + co);
      __syncthreads();                             // operand reads / previous chunk's read-back done
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int col0 = (wn * TN + j) * 32;
        if (col0 / CW == c) {
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
              const int row = (wm * TM + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
              stage[row * SLD + (col0 % CW) + (lane & 31)] = acc[i][j][e];
            }
        }
      }
      __syncthreads();
      const unsigned ybase = (unsigned)(m0 + row_t) * ldy4 + (unsigned)(p.ycoff + co) * 4u;
#pragma unroll
      for (int it = 0; it < ITERS; ++it) {
        const int row = row_t + it * RPI;
        const bool ok = m0 + row < p.M && cok;
        float4 v = *reinterpret_cast<const float4*>(&stage[row * SLD + c4 * 4]);
        v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
        v.x = fmaxf(v.x, lo2); v.y = fmaxf(v.y, lo2); v.z = fmaxf(v.z, lo2); v.w = fmaxf(v.w, lo2);
        if (p.res_mode != 0) {
          const float4 r = rres[it];
          v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
        }
        v.x = fmaxf(v.x, lo1); v.y = fmaxf(v.y, lo1); v.z = fmaxf(v.z, lo1); v.w = fmaxf(v.w, lo1);
        if (HALF && p.yh) {
          const h4 hv = h4{(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w};          // v_cvt_f16_f32: RNE
          __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, hv), yr,
                                                ok ? (ybase + (unsigned)(it * RPI) * ldy4) >> 1 : EOOB, 0, 0);
        } else {
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), yr,
                                                 ok ? ybase + (unsigned)(it * RPI) * ldy4 : EOOB, 0, 0);
        }
      }
    }
    return;
  }
  // Scalar path (channel-strided / unaligned outputs)
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int co = n0 + (wn * TN + j) * 32 + (lane & 31);
    const bool cok = co < p.Cout;
    const float bv = (p.bias != nullptr && cok) ? p.bias[co] : 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
        const int m = m0 + (wm * TM + i) * 32 + row;
        if (cok && m < p.M) {
          float v = acc[i][j][e] + bv;
          if (p.relu == 2) v = fmaxf(v, 0.f);
          const _Float16* resh = reinterpret_cast<const _Float16*>(p.res);
          const bool rh = HALF && p.rh;
          if (p.res_mode == 1) {
            v += rh ? (float)resh[(long)m * p.ldr + co] : p.res[(long)m * p.ldr + co];
          } else if (p.res_mode == 2) {
            const int n = m / HoWo;
            const int rem = m - n * HoWo;
            const int ho = rem / p.Wo;
            const int wo = rem - ho * p.Wo;
            const long ri = ((long)n * HoWo2 + (long)(ho >> 1) * (p.Wo >> 1) + (wo >> 1)) * p.ldr + co;
            v += rh ? (float)resh[ri] : p.res[ri];
          }
          if (p.relu == 1) v = fmaxf(v, 0.f);
          if (HALF && p.yh) reinterpret_cast<_Float16*>(p.y)[(long)m * p.ldy + p.ycoff + (long)co * p.ycs] = (_Float16)v;
          else p.y[(long)m * p.ldy + p.ycoff + (long)co * p.ycs] = v;
        }
      }
    }
  }
}

template <int WAVES_M, int WAVES_N, int TM, int TN, int NSTAGE, int MINW, int BK, bool HALF>
static int launch_conv_cfg(ConvParams& p, hipStream_t stream) {
  constexpr int BM = WAVES_M * TM * 32, BN = WAVES_N * TN * 32;
  p.tiles_m = cdiv(p.M, BM);
  p.tiles_n = cdiv(p.Cout, BN);
  const long nblk = (long)p.tiles_m * p.tiles_n;
  if (nblk <= 0 || nblk > 0x7fffffffL) {
    glass_set_error("glass_conv2d_nhwc: bad grid %ld", nblk);
    return GLASS_EINVAL;
  }
  p.nk = cdiv(p.Ktot, BK);
  const dim3 grid((unsigned)nblk, (unsigned)(p.ksplit > 1 ? p.ksplit : 1));
  if (p.x_bytes != 0 && p.Cin % BK == 0)
    hipLaunchKernelGGL((conv_igemm_f32<WAVES_M, WAVES_N, TM, TN, NSTAGE, MINW, BK, 1, HALF>), grid, dim3(256), 0, stream, p);
  else if (p.x_bytes != 0)
    hipLaunchKernelGGL((conv_igemm_f32<WAVES_M, WAVES_N, TM, TN, NSTAGE, MINW, BK, 2, HALF>), grid, dim3(256), 0, stream, p);
  else if constexpr (!HALF)
    hipLaunchKernelGGL((conv_igemm_f32<WAVES_M, WAVES_N, TM, TN, NSTAGE, MINW, BK, 0, false>), grid, dim3(256), 0, stream, p);
  else {
    glass_set_error("glass_conv2d_nhwc_f16: tensors of 2 GiB and more are not supported in the fp16 mode");
    return GLASS_EINVAL;
  }
  GLASS_CHECK_LAUNCH("glass_conv2d_nhwc");
  return GLASS_OK;
}

template <int WAVES_M, int WAVES_N, int TM, int TN, int NSTAGE, int MINW, int BK>
static int launch_conv_impl(ConvParams& p, hipStream_t stream) {
  // (64-wide k-tiles for the fp16 kernels - 4 instead of 2 MFMAs per accumulator block and barrier pair - were measured
  //  2.6x SLOWER: 148 vs 388 images/s in fp16 mode; the 144-byte LDS rows they need are not conflict-free)
  return p.half_mode ? launch_conv_cfg<WAVES_M, WAVES_N, TM, TN, NSTAGE, MINW, BK, true>(p, stream)
                     : launch_conv_cfg<WAVES_M, WAVES_N, TM, TN, NSTAGE, MINW, BK, false>(p, stream);
}

static int conv_dispatch(const glass_conv_desc* d, const float* x, const float* w, const float* bias, const float* residual,
                         float* y, glass_stream_t stream, int half_mode, int h16_flags = 0);

extern "C" int glass_conv2d_nhwc(const glass_conv_desc* d, const float* x, const float* w, const float* bias,
                                 const float* residual, float* y, glass_stream_t stream) {
  return conv_dispatch(d, x, w, bias, residual, y, stream, 0);
}

extern "C" int glass_conv2d_nhwc_f16(const glass_conv_desc* d, const float* x, const float* w, const float* bias,
                                     const float* residual, float* y, glass_stream_t stream) {
  return conv_dispatch(d, x, w, bias, residual, y, stream, 1);
}

extern "C" int glass_conv2d_nhwc_h16(const glass_conv_desc* d, const void* x, const float* w, const float* bias,
                                     const void* residual, void* y, int flags, glass_stream_t stream) {
  GLASS_CHECK_ARG((flags & ~7) == 0, "glass_conv2d_nhwc_h16: unknown flags 0x%x", flags);
  return conv_dispatch(d, static_cast<const float*>(x), w, bias, static_cast<const float*>(residual), static_cast<float*>(y),
                       stream, 1, flags);
}

static int conv_dispatch(const glass_conv_desc* d, const float* x, const float* w, const float* bias, const float* residual,
                         float* y, glass_stream_t stream, int half_mode, int h16_flags) {
  GLASS_CHECK_ARG(d && x && w && y, "glass_conv2d_nhwc: null pointer");
  GLASS_CHECK_ARG(d->Cin > 0 && d->Cin % 4 == 0, "glass_conv2d_nhwc: Cin=%d must be a positive multiple of 4", d->Cin);
  GLASS_CHECK_ARG(d->ldx % 4 == 0 && d->ldx >= d->Cin, "glass_conv2d_nhwc: ldx=%d", d->ldx);
  GLASS_CHECK_ARG(d->N >= 0 && d->H > 0 && d->W > 0 && d->Cout > 0 && d->KH > 0 && d->KW > 0, "glass_conv2d_nhwc: bad dims");
  GLASS_CHECK_ARG(d->stride_h > 0 && d->stride_w > 0, "glass_conv2d_nhwc: bad stride");
  GLASS_CHECK_ARG(d->Ho == (d->H + 2 * d->pad_h - d->KH) / d->stride_h + 1 &&
                      d->Wo == (d->W + 2 * d->pad_w - d->KW) / d->stride_w + 1,
                  "glass_conv2d_nhwc: Ho/Wo (%d,%d) inconsistent with input/kernel/stride/pad", d->Ho, d->Wo);
  GLASS_CHECK_ARG(d->y_cstride >= 1 && d->ldy >= 1 && d->y_coff >= 0, "glass_conv2d_nhwc: bad output strides");
  // the channel window [y_coff, y_coff + (Cout-1) * y_cstride] must stay inside one pixel's ldy channels: an over-wide
  // window would silently spill into the next pixel (and be dropped by the bounds check only on the last row)
  GLASS_CHECK_ARG((long)d->y_coff + (long)(d->Cout - 1) * d->y_cstride < (long)d->ldy,
                  "glass_conv2d_nhwc: output channel window (y_coff=%d, Cout=%d, y_cstride=%d) exceeds ldy=%d", d->y_coff,
                  d->Cout, d->y_cstride, d->ldy);
  GLASS_CHECK_ARG(d->res_mode == 0 || d->ldr >= d->Cout, "glass_conv2d_nhwc: residual ldr=%d < Cout=%d", d->ldr, d->Cout);
  GLASS_CHECK_ARG(d->res_mode == 0 || residual != nullptr, "glass_conv2d_nhwc: res_mode set but residual is null");
  GLASS_CHECK_ARG(d->res_mode != 2 || (d->Ho % 2 == 0 && d->Wo % 2 == 0), "glass_conv2d_nhwc: upsampled residual needs even Ho/Wo");
  GLASS_CHECK_ARG(((uintptr_t)x & 15) == 0 && ((uintptr_t)w & 15) == 0, "glass_conv2d_nhwc: x/w must be 16-byte aligned");
  if (d->N == 0) return GLASS_OK;
  ConvParams p;
  p.half_mode = half_mode;
  p.xh = h16_flags & 1; p.yh = (h16_flags >> 1) & 1; p.rh = (h16_flags >> 2) & 1;
  const long xes = p.xh ? 2 : 4, yes = p.yh ? 2 : 4, res_ = p.rh ? 2 : 4;      // bytes per stored element
  p.x = x; p.w = w; p.bias = bias; p.res = residual; p.y = y;
  p.N = d->N; p.H = d->H; p.W = d->W; p.Cin = d->Cin; p.Cout = d->Cout; p.KH = d->KH; p.KW = d->KW;
  p.sh = d->stride_h; p.sw = d->stride_w; p.ph = d->pad_h; p.pw = d->pad_w; p.Ho = d->Ho; p.Wo = d->Wo;
  p.ldx = d->ldx; p.ldy = d->ldy; p.ycoff = d->y_coff; p.ycs = d->y_cstride; p.relu = d->relu;
  p.res_mode = d->res_mode; p.ldr = d->ldr;
  const long M = (long)d->N * d->Ho * d->Wo;
  GLASS_CHECK_ARG(M < 0x7fffffffL, "glass_conv2d_nhwc: too many output pixels");
  p.M = (int)M;
  p.Ktot = d->KH * d->KW * d->Cin;
  p.ksplit = 1; p.split_y = 0;
  {
    // the buffer-load paths need 31-bit byte offsets (MODE 1 additionally Cin % 32 == 0, checked at launch)
    // (offsets are computed as fp32 byte offsets and halved for fp16 tensors: the fp32-sized span must fit 31 bits)
    const long xb = (long)d->N * d->H * d->W * d->ldx * 4, wb = (long)d->Cout * p.Ktot * 4;
    const bool small = xb < 0x7fffff00L && wb < 0x7fffff00L;
    p.x_bytes = small ? (unsigned)(xb / 4 * xes) : 0u;
    p.w_bytes = small ? (unsigned)wb : 0u;
    p.magic_cin = (unsigned)(0x100000000ULL / (unsigned long long)d->Cin);
    p.magic_kw = (unsigned)(0x100000000ULL / (unsigned long long)d->KW);
  }
  p.vec_epi = (d->y_cstride == 1 && d->ldy % 4 == 0 && d->y_coff % 4 == 0 && d->Cout % 4 == 0 &&
               ((uintptr_t)y & 15) == 0 && (bias == nullptr || ((uintptr_t)bias & 15) == 0) &&
               (d->res_mode == 0 || (d->ldr % 4 == 0 && ((uintptr_t)residual & 15) == 0))) ? 1 : 0;
  {
    const long yb = M * d->ldy * 4;
    const long rb = d->res_mode == 1 ? M * d->ldr * 4
                  : d->res_mode == 2 ? (long)d->N * (d->Ho / 2) * (d->Wo / 2) * d->ldr * 4 : 0;
    if (yb >= 0x7fffff00L || rb >= 0x7fffff00L) p.vec_epi = 0;      // > 2 GiB output: scalar epilogue (64-bit addresses)
    p.y_bytes = (unsigned)(p.vec_epi ? yb / 4 * yes : 0);
    p.r_bytes = (unsigned)(p.vec_epi ? rb / 4 * res_ : 0);
  }
  hipStream_t s = (hipStream_t)stream;
  static const int force_cfg = getenv("GLASS_CONV_CFG") ? atoi(getenv("GLASS_CONV_CFG")) : 0;   // tuning aid
  if (force_cfg == 1) return launch_conv_impl<2, 2, 2, 2, 1, 3, 32>(p, s);
  if (force_cfg == 2) return launch_conv_impl<1, 4, 2, 1, 1, 4, 32>(p, s);
  if (force_cfg == 3) return launch_conv_impl<2, 2, 2, 1, 1, 4, 32>(p, s);
  if (force_cfg == 4) return launch_conv_impl<2, 2, 1, 1, 1, 8, 32>(p, s);   // 64 x 64, 8 blocks/CU
  if (force_cfg == 5) return launch_conv_impl<1, 4, 1, 1, 1, 6, 32>(p, s);   // 32 x 128
  if (force_cfg == 6) return launch_conv_impl<4, 1, 1, 3, 1, 4, 32>(p, s);   // 128 x 96
  if (force_cfg == 7) return launch_conv_impl<2, 2, 4, 2, 1, 2, 32>(p, s);   // 256 x 128, 2 blocks/CU
  if (force_cfg == 8) return launch_conv_impl<2, 2, 2, 4, 1, 2, 32>(p, s);   // 128 x 256, 2 blocks/CU
  if (d->Cout <= 32) return launch_conv_impl<4, 1, 1, 1, 1, 4, 32>(p, s);   // 128 x 32
  if (d->Cout <= 64) return launch_conv_impl<2, 2, 2, 1, 1, 4, 32>(p, s);   // 128 x 64
  if (d->Cout <= 96) return launch_conv_impl<4, 1, 1, 3, 1, 4, 32>(p, s);   // 128 x 96 (the merged 72-channel RPN heads: 0.36 -> 0.26 ms)
  // short K (1x1 convs with Cin <= 256; most carry a fused residual): few k-tiles per block, so the block is mostly
  // prologue + epilogue and the layer is HBM-bound - 64x64 tiles at 8 blocks/CU hide those latencies far better
  // than 3 big blocks (64->256 +residual: 3.1 -> 4.7 TB/s; 128->512: 2.3 -> 3.1; 256->1024: 88 -> 108 TFLOP/s)
  static const int k64 = getenv("GLASS_CONV_K64") ? atoi(getenv("GLASS_CONV_K64")) : 256;       // tuning aid
  if (p.Ktot <= k64) return launch_conv_impl<2, 2, 1, 1, 1, 8, 32>(p, s);   // 64 x 64
  // few 128x128 tiles (deep small maps, linear layers on <=800 rows): halve the tile height so the
  // grid covers the 256 CUs at least ~2x
  const long tiles128 = (long)cdiv(p.M, 128) * cdiv(d->Cout, 128);
  if (p.M <= 2048 && tiles128 < 640 && d->Cout >= 256) return launch_conv_impl<2, 2, 1, 1, 1, 8, 32>(p, s);   // 64 x 64: few-row GEMMs (box head fc on 800 rows)
  // the last-column strip of a 16 x 33 map (M = 4096 rows, K = 1536, Cout = 256; ops/native.py _last_column_strip): 64 x 128
  // tiles would be 128 workgroups = half the chip at 48 k-tiles each
  if (p.M <= 8192 && tiles128 < 128 && d->Cout >= 256) return launch_conv_impl<2, 2, 1, 1, 1, 8, 32>(p, s);
  if (p.M <= 64 || tiles128 < 640) return launch_conv_impl<1, 4, 2, 1, 1, 4, 32>(p, s);   // 64 x 128
  return launch_conv_impl<2, 2, 2, 2, 1, 3, 32>(p, s);                      // 128 x 128, 3 blocks/CU
  // (measured on MI355X: BK=64 with 2 blocks/CU and a 2-stage LDS pipeline are both within 2% of this)
}


// ---------------------------------------------------------------------------------------------------------------- split-K
// y = act(conv(x, w) + bias [+ residual]) for FEW output pixels and a LONG K - the layers that leave most of the chip idle
// when one image is in flight (reference predictor: glass/inference/glass_runner.py:93-96 is one image per call): the box
// head's fc1 on 100 proposals (M = 100, K = 12544: 64 workgroups x 392 sequential k-tiles, 0.38 ms at 13 TFLOP/s), res4 /
// res5 3x3 layers on 64 x 64 / 32 x 32 maps (K = 2304 / 4608), the 11-row box predictors (ONE workgroup).  The same
// implicit-GEMM kernel runs `splits` k-slices as grid.y (slice s = k-tiles [s nk / splits, (s+1) nk / splits) of the
// (tap, channel) walk), slice s writes raw partial sums to workspace[s][M][Cout], and a second kernel adds them IN SLICE
// ORDER (deterministic), then bias, ReLU, residual as the single-slice epilogue does, and writes y with its strides.

extern "C" int64_t glass_conv2d_splitk_workspace_bytes(const glass_conv_desc* d, int splits) {
  if (!d) return 0;
  return (int64_t)splits * d->N * d->Ho * d->Wo * d->Cout * (int64_t)sizeof(float);
}

// every k-slice a whole number of 32-channel k-tiles (the uniform-tap load path), residual not upsampled, operands < 2 GiB
extern "C" int glass_conv2d_splitk_supported(const glass_conv_desc* d, int splits) {
  if (!d) return 0;
  const long M = (long)d->N * d->Ho * d->Wo, K = (long)d->KH * d->KW * d->Cin;
  return M > 0 && d->Cout > 0 && d->Cin % 32 == 0 && splits >= 2 && splits <= 32 && (K / 32) % splits == 0 && d->ldx % 4 == 0 &&
         d->ldx >= d->Cin && (d->res_mode == 0 || d->res_mode == 1) &&
         (long)d->N * d->H * d->W * d->ldx * 4 < 0x7fffff00L && (long)d->Cout * K * 4 < 0x7fffff00L && M * d->Cout * 4 < 0x7fffff00L;
}

extern "C" int glass_conv2d_nhwc_splitk(const glass_conv_desc* d, const float* x, const float* w, const float* bias,
                                        const float* residual, float* y, int splits, void* workspace, int64_t workspace_bytes,
                                        glass_stream_t stream) {
  GLASS_CHECK_ARG(d && x && w && y && workspace, "glass_conv2d_nhwc_splitk: null pointer");
  GLASS_CHECK_ARG(glass_conv2d_splitk_supported(d, splits),
                  "glass_conv2d_nhwc_splitk: needs Cin %% 32 == 0, (KH KW Cin / 32) %% splits == 0, 2 <= splits <= 32, res_mode 0/1 and "
                  "operands < 2 GiB (got Cin=%d k=%dx%d splits=%d res_mode=%d)", d->Cin, d->KH, d->KW, splits, d->res_mode);
  GLASS_CHECK_ARG(d->Ho == (d->H + 2 * d->pad_h - d->KH) / d->stride_h + 1 && d->Wo == (d->W + 2 * d->pad_w - d->KW) / d->stride_w + 1,
                  "glass_conv2d_nhwc_splitk: Ho/Wo (%d,%d) inconsistent with input/kernel/stride/pad", d->Ho, d->Wo);
  GLASS_CHECK_ARG(d->y_cstride >= 1 && d->y_coff >= 0 && (long)d->y_coff + (long)(d->Cout - 1) * d->y_cstride < (long)d->ldy,
                  "glass_conv2d_nhwc_splitk: output channel window exceeds ldy=%d", d->ldy);
  GLASS_CHECK_ARG(d->res_mode == 0 || (residual != nullptr && d->ldr >= d->Cout), "glass_conv2d_nhwc_splitk: bad residual");
  GLASS_CHECK_ARG(workspace_bytes >= glass_conv2d_splitk_workspace_bytes(d, splits), "glass_conv2d_nhwc_splitk: workspace too small");
  GLASS_CHECK_ARG((((uintptr_t)x | (uintptr_t)w | (uintptr_t)workspace) & 15) == 0, "glass_conv2d_nhwc_splitk: x / w / workspace must be 16-byte aligned");
  GLASS_CHECK_ARG(d->relu >= 0 && d->relu <= 2, "glass_conv2d_nhwc_splitk: relu must be 0, 1 or 2");
  const long M = (long)d->N * d->Ho * d->Wo;
  ConvParams p;
  p.half_mode = 0; p.xh = p.yh = p.rh = 0;
  p.x = x; p.w = w; p.bias = nullptr; p.res = nullptr; p.y = static_cast<float*>(workspace);
  p.N = d->N; p.H = d->H; p.W = d->W; p.Cin = d->Cin; p.Cout = d->Cout; p.KH = d->KH; p.KW = d->KW;
  p.sh = d->stride_h; p.sw = d->stride_w; p.ph = d->pad_h; p.pw = d->pad_w; p.Ho = d->Ho; p.Wo = d->Wo;
  p.ldx = d->ldx; p.ldy = d->Cout; p.ycoff = 0; p.ycs = 1; p.relu = 0; p.res_mode = 0; p.ldr = 0;
  p.M = (int)M; p.Ktot = d->KH * d->KW * d->Cin; p.ksplit = splits; p.split_y = M * d->Cout;
  p.x_bytes = (unsigned)((long)d->N * d->H * d->W * d->ldx * 4);
  p.w_bytes = (unsigned)((long)d->Cout * p.Ktot * 4);
  p.magic_cin = (unsigned)(0x100000000ULL / (unsigned long long)d->Cin);
  p.magic_kw = (unsigned)(0x100000000ULL / (unsigned long long)d->KW);
  p.vec_epi = d->Cout % 4 == 0 ? 1 : 0;
  p.y_bytes = (unsigned)(M * d->Cout * 4);     // one slice: the kernel's y is already the slice's base
  p.r_bytes = 0;
  hipStream_t s = (hipStream_t)stream;
  const int rc = d->Cout <= 32 ? launch_conv_impl<4, 1, 1, 1, 1, 4, 32>(p, s)          // 128 x 32 tiles (box predictors)
                               : launch_conv_impl<2, 2, 1, 1, 1, 8, 32>(p, s);         // 64 x 64 tiles, 8 workgroups per CU
  if (rc != GLASS_OK) return rc;
  launch_splitk_reduce(static_cast<const float*>(workspace), bias, residual, y, M, d->Cout, splits, d->ldy, d->y_coff, d->y_cstride, d->ldr,
                       d->relu, d->res_mode, s);
  GLASS_CHECK_LAUNCH("glass_conv2d_nhwc_splitk");
  return GLASS_OK;
}
