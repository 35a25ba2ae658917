// Small device helpers shared by the per-RoI recognition kernels (recognition.hip: one launch per recurrent step;
// recurrent_persistent.hip: one launch per recurrent layer / per decode).
#pragma once
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off));
  return v;
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

