// Small device helpers shared by the per-RoI recognition kernels (recognition.hip: one launch per recurrent step;
// recurrent_persistent.hip: one launch per recurrent layer / per decode).
#pragma once
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

// Wavefront reductions on DPP (data-parallel primitives: the cross-lane operand path of the vector ALU, ~8 cycles per step)
// instead of __shfl_xor (ds_bpermute_b32: an LDS-crossbar round trip, ~100 cycles per dependent step - the 24 dependent
// shuffles of a soft-max + arg-max were 2.4 K cycles of the persistent decoder's step).  gfx9 DPP controls: quad_perm,
// row_half_mirror (0x141), row_mirror (0x140) leave every lane of a 16-lane row with the row's result; row_bcast15 (0x142,
// rows 1 and 3) and row_bcast31 (0x143, rows 2 and 3) carry it on so that lane 63 holds the wavefront's, read with readlane.
// `old` is the operation's identity: lanes a row_mask disables get it.
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ float dpp_move(float v, float identity) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(identity), __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
}
// sum / max over each aligned group of 16 lanes, result in every lane of the group
__device__ __forceinline__ float row16_sum(float v) {
  v += dpp_move<0xB1>(v, 0.f);      // quad_perm [1,0,3,2]
  v += dpp_move<0x4E>(v, 0.f);      // quad_perm [2,3,0,1]
  v += dpp_move<0x141>(v, 0.f);     // row_half_mirror
  v += dpp_move<0x140>(v, 0.f);     // row_mirror
  return v;
}
__device__ __forceinline__ float row16_max(float v) {
  v = fmaxf(v, dpp_move<0xB1>(v, -INFINITY));
  v = fmaxf(v, dpp_move<0x4E>(v, -INFINITY));
  v = fmaxf(v, dpp_move<0x141>(v, -INFINITY));
  v = fmaxf(v, dpp_move<0x140>(v, -INFINITY));
  return v;
}
// over the whole wavefront, result wavefront-uniform
__device__ __forceinline__ float wave_sum(float v) {
  v = row16_sum(v);
  v += dpp_move<0x142, 0xa>(v, 0.f);
  v += dpp_move<0x143, 0xc>(v, 0.f);
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ float wave_max(float v) {
  v = row16_max(v);
  v = fmaxf(v, dpp_move<0x142, 0xa>(v, -INFINITY));
  v = fmaxf(v, dpp_move<0x143, 0xc>(v, -INFINITY));
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
// arg-max with the FIRST maximum winning (torch.max semantics): (value, index) per lane -> the wavefront's, uniform
__device__ __forceinline__ void wave_argmax_first(float& best, int& besti) {
  const float m = wave_max(best);
  // lanes that hold the maximum offer their index, the others INT_MAX; the minimum index wins
  int cand = best == m ? besti : 0x7fffffff;
  // min over the wavefront as -max(-x) on floats would lose bits: 32-bit integer min on DPP
  auto imin = [](int a, int b) { return a < b ? a : b; };
  cand = imin(cand, __builtin_amdgcn_update_dpp(0x7fffffff, cand, 0xB1, 0xf, 0xf, false));
  cand = imin(cand, __builtin_amdgcn_update_dpp(0x7fffffff, cand, 0x4E, 0xf, 0xf, false));
  cand = imin(cand, __builtin_amdgcn_update_dpp(0x7fffffff, cand, 0x141, 0xf, 0xf, false));
  cand = imin(cand, __builtin_amdgcn_update_dpp(0x7fffffff, cand, 0x140, 0xf, 0xf, false));
  cand = imin(cand, __builtin_amdgcn_update_dpp(0x7fffffff, cand, 0x142, 0xa, 0xf, false));
  cand = imin(cand, __builtin_amdgcn_update_dpp(0x7fffffff, cand, 0x143, 0xc, 0xf, false));
  besti = __builtin_amdgcn_readlane(cand, 63);
  best = m;
}
__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }
// tanh(x) = 1 - 2 / (exp(2x) + 1) on the hardware exp/rcp (|err| < 2e-7 absolute; saturates cleanly)
__device__ __forceinline__ float tanh_fast(float x) { return 1.f - 2.f * __frcp_rn(__expf(2.f * x) + 1.f); }


// logistic function on the hardware exp / rcp (v_exp_f32, v_rcp_f32; |err| < 2e-7 absolute, saturates cleanly: exp -> inf -> 0)
__device__ __forceinline__ float sigmoid_fast(float x) { return __frcp_rn(1.f + __expf(-x)); }

// nn.LSTM cell update from the four gate pre-activations (order i, f, g, o) - ONE definition for the step kernel and the
// persistent kernel, with the multiply-add of the cell state spelled out so that both compile to the same rounding.
// The five transcendentals run on the hardware exp / rcp: libm's expf / tanhf are ~80 instructions each, and with one
// (RoI, unit) element per thread the gate functions were 1.3 of the 3.8 us a chain-step took (the matrix-core phase is 1.7).
struct LstmCell { float c, h; };
__device__ __forceinline__ LstmCell lstm_cell(float pi, float pf, float pg, float po, float c_prev) {
  const float ig = sigmoid_fast(pi), fg = sigmoid_fast(pf), gg = tanh_fast(pg), og = sigmoid_fast(po);
  LstmCell r;
  r.c = __fmaf_rn(fg, c_prev, ig * gg);
  r.h = og * tanh_fast(r.c);
  return r;
}

// reference early break (prediction_aster.py:91-93): after step i, if every RoI of the call has
// emitted `eos` at least once the loop stops and later rows stay zero.  One workgroup per image.
static __global__ void decode_break_mask_kernel(const int* __restrict__ pred, const int* __restrict__ roi_image, int R, int max_len,
                                         int C, int eos, float* __restrict__ out) {
  __shared__ int s_lo, s_hi, s_break;
  const int img = blockIdx.x;
  if (threadIdx.x == 0) { s_lo = R; s_hi = -1; s_break = -1; }
  __syncthreads();
  for (int r = threadIdx.x; r < R; r += blockDim.x)
    if (roi_image[r] == img) { atomicMin(&s_lo, r); atomicMax(&s_hi, r); }
  __syncthreads();
  if (s_hi < 0) return;
  const int lo = s_lo, hi = s_hi;
  for (int r = lo + threadIdx.x; r <= hi; r += blockDim.x) {
    int first = max_len;   // first step with pred == eos (max_len: never)
    for (int t = 0; t < max_len; ++t)
      if (pred[(long)r * max_len + t] == eos) { first = t; break; }
    atomicMax(&s_break, first);
  }
  __syncthreads();
  const int brk = s_break;   // loop ran steps 0..brk (inclusive) if brk < max_len
  if (brk >= max_len - 1) return;
  const long per_row = (long)(max_len - 1 - brk) * C;
  for (int r = lo; r <= hi; ++r) {
    float* o = out + ((long)r * max_len + brk + 1) * C;
    for (long i = threadIdx.x; i < per_row; i += blockDim.x) o[i] = 0.f;
  }
}

