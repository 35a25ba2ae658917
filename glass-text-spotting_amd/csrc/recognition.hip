// Per-RoI recognition kernels: global-to-local fusion attention (softmax pooling + channel MLP),
// the BiLSTM recurrence and the greedy attention-GRU decoder.
//
// All three are independent across RoIs.  The fusion attention is one workgroup per RoI.  The two recurrences are
// ONE LAUNCH PER DEPENDENT STEP (T launches of lstm_step_kernel per layer; dec_fc_att_kernel + dec_gru_kernel per
// decoding step), all issued by a single ABI call with no host synchronisation - the arg-max feedback and the
// recurrent state stay on the device (state double-buffered in a global workspace, previous rows staged in LDS).
// A persistent workgroup per RoI group (no launch per step) was the first design and lost: it has to pull the whole
// 1 MiB recurrent matrix through ONE CU per step (27 us per step measured vs 8 us with each step spread over the chip
// on 16x16x4 fp32 MFMA tiles).  Weights stream from L2 in a k-blocked layout
// (packed[k/4][row][4]) so that the 64 lanes of a wavefront read 1 KiB contiguous per load; reductions (softmax,
// LayerNorm, attention energies) are wavefront shuffle reductions (64 lanes) + one LDS hop across the 4 wavefronts.
#include "recognition_common.h"

// acc[g][r] += sum_k W[g*gstride + u][k] * vec[r][k] for one output unit `u` per thread, weights in the
// k-blocked layout (w4[k4 * rows_total + row] = 4 consecutive k of `row`).  L2 latency (~300 ns) is far
// longer than one k-step of FMAs, so the weight stream runs PF k-steps ahead through a register ring
// (statically indexed: fully unrolled), with the reload unconditional in the main loop and absent in
// the tail (a predicated reload would make hipcc wait vmcnt(0) per element).
template <int G, int RB, int PF>
__device__ __forceinline__ void stream_matvec(const float4* __restrict__ w4, int rows_total, int gstride, int u, int K4,
                                              const float* vec, int ldv, float (&acc)[G][RB]) {
  float4 ring[PF][G];
#pragma unroll
  for (int j = 0; j < PF; ++j)
#pragma unroll
    for (int g = 0; g < G; ++g) ring[j][g] = w4[(long)j * rows_total + g * gstride + u];
  int k0 = 0;
#pragma unroll 1
  for (; k0 + PF < K4; k0 += PF) {
#pragma unroll
    for (int j = 0; j < PF; ++j) {
      float4 wv[G];
#pragma unroll
      for (int g = 0; g < G; ++g) {
        wv[g] = ring[j][g];
        ring[j][g] = w4[(long)(k0 + PF + j) * rows_total + g * gstride + u];
      }
#pragma unroll
      for (int r = 0; r < RB; ++r) {
        const float4 hv = *reinterpret_cast<const float4*>(vec + r * ldv + (k0 + j) * 4);
#pragma unroll
        for (int g = 0; g < G; ++g) acc[g][r] += wv[g].x * hv.x + wv[g].y * hv.y + wv[g].z * hv.z + wv[g].w * hv.w;
      }
    }
  }
#pragma unroll
  for (int j = 0; j < PF; ++j) {
#pragma unroll
    for (int r = 0; r < RB; ++r) {
      const float4 hv = *reinterpret_cast<const float4*>(vec + r * ldv + (k0 + j) * 4);
#pragma unroll
      for (int g = 0; g < G; ++g)
        acc[g][r] += ring[j][g].x * hv.x + ring[j][g].y * hv.y + ring[j][g].z * hv.z + ring[j][g].w * hv.w;
    }
  }
}

// ================================================================== fusion attention
// one workgroup (256 threads) per RoI; x slab [HW=256][C=512]
constexpr int GC_C = 512, GC_HW = 256, GC_HEADS = 8, GC_P = 256;

__global__ __launch_bounds__(256) void gc_attention_kernel(float* __restrict__ xall, const float* __restrict__ w_mask,
                                                           const float* __restrict__ b_mask, const float* __restrict__ w1,
                                                           const float* __restrict__ b1, const float* __restrict__ ln_g,
                                                           const float* __restrict__ ln_b, const float* __restrict__ w2,
                                                           const float* __restrict__ b2) {
  __shared__ float prob[GC_HEADS][GC_HW];   // mask logits, then softmax probabilities
  __shared__ float ctx[GC_C];
  __shared__ float hid[GC_P];
  __shared__ float red[8];
  float* x = xall + (long)blockIdx.x * GC_HW * GC_C;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

  // phase 1: mask logits. A wavefront covers one position per iteration: lane l holds channels
  // 4l..4l+3 and 256+4l..; a head is 64 channels = 16 lanes -> 4-step shuffle reduction.
  // (8 positions = 16 independent 16-byte loads per lane in flight per trip: the slab streams from HBM, and one position per
  //  trip - the shuffles and the LDS store keep the compiler from overlapping trips - made this phase 64 serial round trips)
  const float4 wm = *reinterpret_cast<const float4*>(w_mask + ((lane * 4) & 63));
  const float bm = b_mask[0];
  constexpr int GC_U1 = 8;
  static_assert(GC_HW % (4 * GC_U1) == 0, "phase 1 unroll");
  for (int pos0 = wave; pos0 < GC_HW; pos0 += 4 * GC_U1) {
    float4 a[GC_U1], b[GC_U1];
#pragma unroll
    for (int i = 0; i < GC_U1; ++i) {
      a[i] = *reinterpret_cast<const float4*>(x + (long)(pos0 + 4 * i) * GC_C + lane * 4);
      b[i] = *reinterpret_cast<const float4*>(x + (long)(pos0 + 4 * i) * GC_C + 256 + lane * 4);
    }
#pragma unroll
    for (int i = 0; i < GC_U1; ++i) {
      float sa = a[i].x * wm.x + a[i].y * wm.y + a[i].z * wm.z + a[i].w * wm.w;
      float sb = b[i].x * wm.x + b[i].y * wm.y + b[i].z * wm.z + b[i].w * wm.w;
      sa = row16_sum(sa);
      sb = row16_sum(sb);
      if ((lane & 15) == 0) {
        prob[lane >> 4][pos0 + 4 * i] = sa + bm;
        prob[4 + (lane >> 4)][pos0 + 4 * i] = sb + bm;
      }
    }
  }
  __syncthreads();
  // phase 2: softmax over the 256 positions of each head (2 heads per wavefront)
  for (int h = wave * 2; h < wave * 2 + 2; ++h) {
    float v[4], m = -INFINITY;
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[i] = prob[h][lane + 64 * i]; m = fmaxf(m, v[i]); }
    m = wave_max(m);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[i] = expf(v[i] - m); s += v[i]; }
    s = wave_sum(s);
#pragma unroll
    for (int i = 0; i < 4; ++i) prob[h][lane + 64 * i] = v[i] / s;
  }
  __syncthreads();
  // phase 3: context[c] = sum_pos x[pos][c] * prob[head(c)][pos]; thread owns channels tid, tid+256
  {
    float c0 = 0.f, c1 = 0.f;
    const int h0 = tid >> 6, h1 = 4 + (tid >> 6);
    constexpr int GC_U3 = 16;                                   // 32 independent loads per thread in flight; same summation order
    for (int pos0 = 0; pos0 < GC_HW; pos0 += GC_U3) {
      float v0[GC_U3], v1[GC_U3];
#pragma unroll
      for (int i = 0; i < GC_U3; ++i) {
        v0[i] = x[(long)(pos0 + i) * GC_C + tid];
        v1[i] = x[(long)(pos0 + i) * GC_C + 256 + tid];
      }
#pragma unroll
      for (int i = 0; i < GC_U3; ++i) {
        c0 += v0[i] * prob[h0][pos0 + i];
        c1 += v1[i] * prob[h1][pos0 + i];
      }
    }
    ctx[tid] = c0;
    ctx[256 + tid] = c1;
  }
  __syncthreads();
  // phase 4a: hid = W1 ctx + b1 (256 x 512): each wavefront takes rows wave, wave+4, ...
  // (8 rows per trip: the weight rows come from L2 and the reduction after each row kept one row in flight - 64 serial L2
  //  round trips per wavefront here, 128 in phase 4b)
  {
    constexpr int GC_U4 = 8;
    static_assert(GC_P % (4 * GC_U4) == 0 && GC_C % (4 * GC_U4) == 0, "phase 4 unroll");
    const float4 ca = *reinterpret_cast<const float4*>(ctx + lane * 4);
    const float4 cb = *reinterpret_cast<const float4*>(ctx + 256 + lane * 4);
    for (int j0 = wave; j0 < GC_P; j0 += 4 * GC_U4) {
      float4 wa[GC_U4], wb[GC_U4];
#pragma unroll
      for (int i = 0; i < GC_U4; ++i) {
        wa[i] = *reinterpret_cast<const float4*>(w1 + (long)(j0 + 4 * i) * GC_C + lane * 4);
        wb[i] = *reinterpret_cast<const float4*>(w1 + (long)(j0 + 4 * i) * GC_C + 256 + lane * 4);
      }
#pragma unroll
      for (int i = 0; i < GC_U4; ++i) {
        float s = wa[i].x * ca.x + wa[i].y * ca.y + wa[i].z * ca.z + wa[i].w * ca.w + wb[i].x * cb.x + wb[i].y * cb.y + wb[i].z * cb.z +
                  wb[i].w * cb.w;
        s = wave_sum(s);
        if (lane == 0) hid[j0 + 4 * i] = s + b1[j0 + 4 * i];
      }
    }
  }
  __syncthreads();
  // LayerNorm over the 256 hidden values (biased variance, eps 1e-5) + ReLU
  {
    const float v = hid[tid];
    float s = wave_sum(v);
    if (lane == 0) red[wave] = s;
    __syncthreads();
    const float mean = (red[0] + red[1] + red[2] + red[3]) / (float)GC_P;
    const float d = v - mean;
    float q = wave_sum(d * d);
    if (lane == 0) red[4 + wave] = q;
    __syncthreads();
    const float var = (red[4] + red[5] + red[6] + red[7]) / (float)GC_P;
    const float y = d / sqrtf(var + 1e-5f) * ln_g[tid] + ln_b[tid];
    __syncthreads();
    hid[tid] = fmaxf(y, 0.f);
  }
  __syncthreads();
  // phase 4b: t = W2 hid + b2 (512 x 256) -> reuse ctx[] for t
  {
    constexpr int GC_U4 = 8;
    const float4 hv = *reinterpret_cast<const float4*>(hid + lane * 4);
    for (int c0 = wave; c0 < GC_C; c0 += 4 * GC_U4) {
      float4 wv[GC_U4];
#pragma unroll
      for (int i = 0; i < GC_U4; ++i) wv[i] = *reinterpret_cast<const float4*>(w2 + (long)(c0 + 4 * i) * GC_P + lane * 4);
#pragma unroll
      for (int i = 0; i < GC_U4; ++i) {
        float s = wv[i].x * hv.x + wv[i].y * hv.y + wv[i].z * hv.z + wv[i].w * hv.w;
        s = wave_sum(s);
        if (lane == 0) ctx[c0 + 4 * i] = s + b2[c0 + 4 * i];
      }
    }
  }
  __syncthreads();
  // phase 5: x += t (broadcast over positions), float4 streaming
  {
    const int c4 = tid & 127;           // 128 float4 per position
    const float4 t = *reinterpret_cast<const float4*>(ctx + c4 * 4);
    constexpr int GC_U5 = 16;
    for (int pos0 = tid >> 7; pos0 < GC_HW; pos0 += 2 * GC_U5) {
      float4 v[GC_U5];
#pragma unroll
      for (int i = 0; i < GC_U5; ++i) v[i] = *(reinterpret_cast<const float4*>(x + (long)(pos0 + 2 * i) * GC_C) + c4);
#pragma unroll
      for (int i = 0; i < GC_U5; ++i) {
        v[i].x += t.x; v[i].y += t.y; v[i].z += t.z; v[i].w += t.w;
        *(reinterpret_cast<float4*>(x + (long)(pos0 + 2 * i) * GC_C) + c4) = v[i];
      }
    }
  }
}

extern "C" int glass_gc_attention_inplace(float* x, int R, int HW, int C, int heads, int P, const float* w_mask,
                                          const float* b_mask, const float* w1, const float* b1, const float* ln_g,
                                          const float* ln_b, const float* w2, const float* b2, glass_stream_t stream) {
  GLASS_CHECK_ARG(C == GC_C && HW == GC_HW && heads == GC_HEADS && P == GC_P,
                  "glass_gc_attention_inplace: only C=512, HW=256, heads=8, P=256 is built (got %d,%d,%d,%d)", C, HW, heads, P);
  if (R == 0) return GLASS_OK;
  GLASS_CHECK_ARG(x && w_mask && b_mask && w1 && b1 && ln_g && ln_b && w2 && b2, "glass_gc_attention_inplace: null pointer");
  hipLaunchKernelGGL(gc_attention_kernel, dim3(R), dim3(256), 0, (hipStream_t)stream, x, w_mask, b_mask, w1, b1, ln_g, ln_b,
                     w2, b2);
  GLASS_CHECK_LAUNCH("glass_gc_attention_inplace");
  return GLASS_OK;
}

// ================================================================== BiLSTM recurrence
// One launch per time step (T launches per layer, issued back to back by the host function).  A persistent
// workgroup per RoI group would have to stream the whole 1 MiB recurrent matrix through ONE CU every step
// (~40 GB/s per CU -> 27 us/step); instead each step is spread over the chip: grid = (RoI groups of 16,
// 8 blocks of 32 hidden units, 2 directions), every workgroup streams only its 128 KiB weight slice
// (4 gates x 32 units x 256) from L2 and keeps the 16 previous hidden rows in LDS.  The hidden/cell state
// lives in a small global workspace, double-buffered across launches (stream order is the barrier).
constexpr int LSTM_HD = 256;
constexpr int LS_RB = 16;   // RoIs per workgroup (= the N of a 16x16x4 MFMA)
constexpr int LS_UB = 32;   // hidden units per workgroup

// C[16 rows][16 RoIs] += W[row0 .. row0+16)[0..K) . hs[roi][0..K)^T on v_mfma_f32_16x16x4_f32.
// W is the plain row-major [rows][K] matrix (nn.LSTM / nn.GRU layout).  Lane l = (row r = l&15,
// k-quarter q = l>>4) loads 16 bytes = W[row0+r][16S + 4q .. +3] and hs[l&15][16S + 4q .. +3]; MFMA c of
// super-step S then contracts k = 16S + 4q + c on both operands, so the 16 k of a super-step are all
// covered once (the order inside the fp32 sum differs from k-ascending, nothing else).
template <int K>
__device__ __forceinline__ void mfma_rows16(const float* __restrict__ W, int ldw, int row0, const float* hs, int ldh,
                                            int lane, f32x4& acc) {
  const float* wp = W + (long)(row0 + (lane & 15)) * ldw + (lane >> 4) * 4;
  const float* hp = hs + (lane & 15) * ldh + (lane >> 4) * 4;
  // all K/16 weight loads of the job are issued before the first MFMA (one L2 round trip per job, not per
  // super-step); two accumulators halve the dependent-MFMA chain
  float4 a[K / 16];
#pragma unroll
  for (int S = 0; S < K / 16; ++S) a[S] = *reinterpret_cast<const float4*>(wp + S * 16);
  f32x4 acc2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int S = 0; S < K / 16; ++S) {
    const float4 b = *reinterpret_cast<const float4*>(hp + S * 16);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[S].x, b.x, acc, 0, 0, 0);
    acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[S].y, b.y, acc2, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[S].z, b.z, acc, 0, 0, 0);
    acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[S].w, b.w, acc2, 0, 0, 0);
  }
  acc += acc2;
}

__global__ __launch_bounds__(256) void lstm_step_kernel(const float* __restrict__ xg, const float* __restrict__ whh,
                                                        const float* __restrict__ h_prev, float* __restrict__ h_next,
                                                        float* __restrict__ c_state, float* __restrict__ out, int R, int T,
                                                        int step) {
  __shared__ __attribute__((aligned(16))) float hs[LS_RB][LSTM_HD + 4];   // +4: 16-byte row skew against bank conflicts
  __shared__ float gates[LS_RB][4 * LS_UB + 4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int dir = blockIdx.z, ub = blockIdx.y, r0 = blockIdx.x * LS_RB;
  const int t = dir == 0 ? step : T - 1 - step;
  const int nr = min(LS_RB, R - r0);
  for (int i = tid; i < LS_RB * (LSTM_HD / 4); i += 256) {
    const int r = i / (LSTM_HD / 4), k4 = i % (LSTM_HD / 4);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r < nr) v = reinterpret_cast<const float4*>(h_prev + ((long)(r0 + r) * 2 + dir) * LSTM_HD)[k4];
    *reinterpret_cast<float4*>(&hs[r][k4 * 4]) = v;
  }
  __syncthreads();
  // the workgroup's 128 gate rows = 8 row tiles of 16: tile = gate * 2 + half-of-32-units; 2 tiles per wavefront
  const float* W = whh + (long)dir * (4 * LSTM_HD) * LSTM_HD;
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int tile = wave * 2 + q;
    const int g = tile >> 1, uh = tile & 1;
    const int row0 = g * LSTM_HD + ub * LS_UB + uh * 16;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    mfma_rows16<LSTM_HD>(W, LSTM_HD, row0, &hs[0][0], LSTM_HD + 4, lane, acc);
    // C layout: col = lane & 15 (RoI), row = (lane >> 4) * 4 + e
#pragma unroll
    for (int e = 0; e < 4; ++e) gates[lane & 15][g * LS_UB + uh * 16 + (lane >> 4) * 4 + e] = acc[e];
  }
  __syncthreads();
  for (int i = tid; i < LS_RB * LS_UB; i += 256) {
    const int r = i / LS_UB, ul = i % LS_UB;
    if (r < nr) {
      const int u = ub * LS_UB + ul;
      const long idx = ((long)(r0 + r) * 2 + dir) * LSTM_HD + u;
      const float* xr = xg + (((long)(r0 + r) * T + t) * 2 + dir) * (4 * LSTM_HD) + u;
      const LstmCell cell = lstm_cell(xr[0] + gates[r][ul], xr[LSTM_HD] + gates[r][LS_UB + ul], xr[2 * LSTM_HD] + gates[r][2 * LS_UB + ul],
                                      xr[3 * LSTM_HD] + gates[r][3 * LS_UB + ul], c_state[idx]);
      c_state[idx] = cell.c;
      const float hn = cell.h;
      h_next[idx] = hn;
      out[((long)(r0 + r) * T + t) * (2 * LSTM_HD) + dir * LSTM_HD + u] = hn;
    }
  }
}

extern "C" int64_t glass_bilstm_workspace_bytes(int R, int Hd) { return (int64_t)3 * R * 2 * Hd * sizeof(float); }

extern "C" int glass_bilstm_recurrence(const float* xg, const float* w_hh_packed, float* out, int R, int T, int Hd,
                                       void* workspace, int64_t workspace_bytes, glass_stream_t stream) {
  GLASS_CHECK_ARG(Hd == LSTM_HD, "glass_bilstm_recurrence: only Hd=256 is built (got %d)", Hd);
  if (R == 0) return GLASS_OK;
  GLASS_CHECK_ARG(xg && w_hh_packed && out && T > 0 && workspace, "glass_bilstm_recurrence: bad args");
  GLASS_CHECK_ARG(workspace_bytes >= glass_bilstm_workspace_bytes(R, Hd), "glass_bilstm_recurrence: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  const size_t n = (size_t)R * 2 * Hd;
  float* hbuf[2] = {static_cast<float*>(workspace), static_cast<float*>(workspace) + n};
  float* c = static_cast<float*>(workspace) + 2 * n;
  hipError_t e = hipMemsetAsync(workspace, 0, 3 * n * sizeof(float), s);     // h0 = c0 = 0
  if (e != hipSuccess) { glass_set_error("glass_bilstm_recurrence: memset: %s", hipGetErrorString(e)); return GLASS_EHIP; }
  const dim3 grid(cdiv(R, LS_RB), LSTM_HD / LS_UB, 2);
  for (int step = 0; step < T; ++step)
    hipLaunchKernelGGL(lstm_step_kernel, grid, dim3(256), 0, s, xg, w_hh_packed, hbuf[step & 1], hbuf[(step + 1) & 1], c, out,
                       R, T, step);
  GLASS_CHECK_LAUNCH("glass_bilstm_recurrence");
  return GLASS_OK;
}

// ================================================================== attention GRU decoder
// Two launches per decoding step, issued back to back by the host function (no host sync, argmax feedback
// stays on device):
//   dec_fc_att_kernel  (4 RoIs per workgroup): [fc + softmax + argmax of the state h_i -> out[:, i], y_i]
//                       then [additive attention with h_i -> context, embedding(y_i)] -> inp_{i+1}
//   dec_gru_kernel     (16 RoIs x GRU_UB hidden units per workgroup): GRU cell on v_mfma_f32_16x16x4_f32,
//                       every workgroup streams only its 3*GRU_UB-row slice of W_ih / W_hh from L2.
// A persistent workgroup per RoI group has to pull all 2.6 MB of decoder weights through ONE CU per step
// (80 us/step measured); spreading each step over the chip costs two launch boundaries (~2 us each).
constexpr int DEC_D = 256;
constexpr int DEC_TMAX = 64;
constexpr int DEC_CMAX = 256;
constexpr int GRU_RB = 16;
constexpr int GRU_UB = 16;      // hidden units per workgroup: 16 -> 16 x 16 = 256 workgroups at R = 256 (one per CU; 32 left half the chip idle)

struct DecParams {
  const float* x; const float* xproj;
  const float *sW, *sB, *wW, *wB, *emb, *w_ih, *w_hh, *b_ih, *b_hh, *fcW, *fcB;
  float temperature;
  int R, T, C, max_len;
  float* out;
  int* pred;
  float* h0; float* h1; float* inp; int* yprev;     // workspace: state (double-buffered), [emb|ctx], last argmax
  float* logits;                                    // glass_attention_decode_step only: raw fc outputs [R, C] (else null)
};

// step == -1: only the attention part with the initial input (y = 0); step == -2: only the attention part with y read from
// p.yprev (one decoder step of a caller-driven search, glass_attention_decode_step); do_att == 0: only the fc part (last step)
// DEC_RB = RoIs per workgroup: 4 when there are enough RoIs to fill the chip with 4-RoI workgroups, else 2 or 1
// (R = 256: 64 workgroups of 4 leave 3/4 of the CUs idle for 38 us per step; 256 workgroups of 1 take 12 us).
template <int DEC_RB>
__global__ __launch_bounds__(256) void dec_fc_att_kernel(DecParams p, const float* __restrict__ hcur, int step, int do_att) {
  __shared__ __attribute__((aligned(16))) float h[DEC_RB][DEC_D];
  __shared__ float sproj[DEC_RB][DEC_D];
  __shared__ float energy[DEC_RB][DEC_TMAX];
  __shared__ float logit[DEC_RB][DEC_CMAX];
  __shared__ int ycur[DEC_RB];
  const int u = threadIdx.x, lane = u & 63, wave = u >> 6;
  const int r0 = blockIdx.x * DEC_RB;
  const int nr = min(DEC_RB, p.R - r0);
  const int T = p.T, C = p.C;
#pragma unroll
  for (int r = 0; r < DEC_RB; ++r) h[r][u] = (r < nr) ? hcur[(long)(r0 + r) * DEC_D + u] : 0.f;
  // previous symbol: 0 ([GO]) before the first step; not read at all by an fc-only launch (do_att == 0: the beam-search entry
  // points p.yprev at scratch it has not written yet); a caller-supplied symbol (glass_attention_decode_step) is clamped to
  // the embedding table - an index outside [0, C) must not become an out-of-bounds read behind a C ABI
  if (u < DEC_RB) ycur[u] = (step == -1 || u >= nr || !do_att) ? 0 : min(max(p.yprev[r0 + u], 0), C - 1);
  __syncthreads();
  if (step >= 0) {
    // ---- logits = fc(h) * temperature; softmax over C; argmax (first maximum)
    if (u < C) {
      float acc[1][DEC_RB];
#pragma unroll
      for (int r = 0; r < DEC_RB; ++r) acc[0][r] = p.fcB[u];
      stream_matvec<1, DEC_RB, 16>(reinterpret_cast<const float4*>(p.fcW), C, 0, u, DEC_D / 4, &h[0][0], DEC_D, acc);
#pragma unroll
      for (int r = 0; r < DEC_RB; ++r) {
        logit[r][u] = acc[0][r] * p.temperature;
        if (p.logits != nullptr && r < nr) p.logits[(long)(r0 + r) * C + u] = logit[r][u];
      }
    }
    __syncthreads();
    if (wave < nr) {
      const int r = wave;
      float v[DEC_CMAX / 64];
      float m = -INFINITY;
#pragma unroll
      for (int i = 0; i < DEC_CMAX / 64; ++i) {
        const int cidx = lane + 64 * i;
        v[i] = cidx < C ? logit[r][cidx] : -INFINITY;
        m = fmaxf(m, v[i]);
      }
      m = wave_max(m);
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < DEC_CMAX / 64; ++i) { v[i] = (lane + 64 * i) < C ? expf(v[i] - m) : 0.f; s += v[i]; }
      s = wave_sum(s);
      float best = -1.f;
      int besti = 0x7fffffff;
      float* o = p.out + ((long)(r0 + r) * p.max_len + step) * C;
#pragma unroll
      for (int i = 0; i < DEC_CMAX / 64; ++i) {
        const int cidx = lane + 64 * i;
        if (cidx < C) {
          const float pr = v[i] / s;
          o[cidx] = pr;
          if (pr > best) { best = pr; besti = cidx; }
        }
      }
      wave_argmax_first(best, besti);
      if (lane == 0) {
        ycur[r] = besti;
        p.yprev[r0 + r] = besti;
        p.pred[(long)(r0 + r) * p.max_len + step] = besti;
      }
    }
    __syncthreads();
  }
  if (!do_att) return;
  // ---- sProj = sEmbed(h)
  {
    float acc[1][DEC_RB];
#pragma unroll
    for (int r = 0; r < DEC_RB; ++r) acc[0][r] = p.sB[u];
    stream_matvec<1, DEC_RB, 16>(reinterpret_cast<const float4*>(p.sW), DEC_D, 0, u, DEC_D / 4, &h[0][0], DEC_D, acc);
#pragma unroll
    for (int r = 0; r < DEC_RB; ++r) sproj[r][u] = acc[0][r];
  }
  __syncthreads();
  // ---- energies e[r][t] = wEmbed(tanh(sProj + xProj[t])): wavefront per (r,t), lanes over D
  {
    const float4 ww = *reinterpret_cast<const float4*>(p.wW + lane * 4);
    const float wb = p.wB[0];
    const int npairs = nr * T;
    for (int base = wave; base < npairs; base += 4 * 8) {
      float4 xp[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int pr = min(base + 4 * i, npairs - 1);
        const int r = pr / T, t = pr - r * T;
        xp[i] = *reinterpret_cast<const float4*>(p.xproj + ((long)(r0 + r) * T + t) * DEC_D + lane * 4);
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int pr = base + 4 * i;
        if (pr < npairs) {
          const int r = pr / T, t = pr - r * T;
          const float4 sp = *reinterpret_cast<const float4*>(&sproj[r][lane * 4]);
          float s = ww.x * tanh_fast(sp.x + xp[i].x) + ww.y * tanh_fast(sp.y + xp[i].y) + ww.z * tanh_fast(sp.z + xp[i].z) +
                    ww.w * tanh_fast(sp.w + xp[i].w);
          s = wave_sum(s);
          if (lane == 0) energy[r][t] = s + wb;
        }
      }
    }
  }
  __syncthreads();
  if (wave < nr) {
    const int r = wave;
    const float v = lane < T ? energy[r][lane] : -INFINITY;
    const float m = wave_max(v);
    const float e = lane < T ? expf(v - m) : 0.f;
    const float s = wave_sum(e);
    if (lane < T) energy[r][lane] = e / s;
  }
  __syncthreads();
  // ---- inp = [embedding(y) | context = alpha . x]
  for (int r = 0; r < nr; ++r) {
    float s = 0.f;
    const float* xr = p.x + (long)(r0 + r) * T * DEC_D + u;
    int t = 0;
    for (; t + 16 <= T; t += 16) {
      float v[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) v[i] = xr[(long)(t + i) * DEC_D];
#pragma unroll
      for (int i = 0; i < 16; ++i) s += energy[r][t + i] * v[i];
    }
    for (; t < T; ++t) s += energy[r][t] * xr[(long)t * DEC_D];
    float* ip = p.inp + (long)(r0 + r) * (2 * DEC_D);
    ip[u] = p.emb[(long)ycur[r] * DEC_D + u];
    ip[DEC_D + u] = s;
  }
}

// h_next = GRUCell(inp, h_prev): grid (ceil(R/16), D/GRU_UB).  The workgroup's 3*GRU_UB gate rows (3 gates x GRU_UB
// units) are 3*GRU_UB/16 row tiles of 16; each tile has three K-slices of 256 (W_ih[:, :256], W_ih[:, 256:], W_hh):
// 9*GRU_UB/16 MFMA jobs of 64 MFMAs, dealt round-robin to the 4 wavefronts; slices are summed in the pointwise phase.
__global__ __launch_bounds__(256) void dec_gru_kernel(DecParams p, const float* __restrict__ hprev, float* __restrict__ hnext) {
  __shared__ __attribute__((aligned(16))) float xin[GRU_RB][2 * DEC_D + 4];
  __shared__ __attribute__((aligned(16))) float hs[GRU_RB][DEC_D + 4];
  __shared__ float gbuf[3][GRU_RB][3 * GRU_UB + 4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r0 = blockIdx.x * GRU_RB, ub = blockIdx.y;
  const int nr = min(GRU_RB, p.R - r0);
  for (int i = tid; i < GRU_RB * (2 * DEC_D / 4); i += 256) {
    const int r = i / (2 * DEC_D / 4), k4 = i % (2 * DEC_D / 4);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r < nr) v = reinterpret_cast<const float4*>(p.inp + (long)(r0 + r) * (2 * DEC_D))[k4];
    *reinterpret_cast<float4*>(&xin[r][k4 * 4]) = v;
  }
  for (int i = tid; i < GRU_RB * (DEC_D / 4); i += 256) {
    const int r = i / (DEC_D / 4), k4 = i % (DEC_D / 4);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r < nr) v = reinterpret_cast<const float4*>(hprev + (long)(r0 + r) * DEC_D)[k4];
    *reinterpret_cast<float4*>(&hs[r][k4 * 4]) = v;
  }
  __syncthreads();
  constexpr int NT = GRU_UB / 16;                           // 16-row MFMA tiles per gate
  for (int job = wave; job < 9 * NT; job += 4) {
    const int tile = job / 3, part = job - tile * 3;        // part 0/1: W_ih column halves, 2: W_hh
    const int g = tile / NT, uh = tile - g * NT;
    const int row0 = g * DEC_D + ub * GRU_UB + uh * 16;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (part < 2) mfma_rows16<DEC_D>(p.w_ih + part * DEC_D, 2 * DEC_D, row0, &xin[0][part * DEC_D], 2 * DEC_D + 4, lane, acc);
    else mfma_rows16<DEC_D>(p.w_hh, DEC_D, row0, &hs[0][0], DEC_D + 4, lane, acc);
#pragma unroll
    for (int e = 0; e < 4; ++e) gbuf[part][lane & 15][g * GRU_UB + uh * 16 + (lane >> 4) * 4 + e] = acc[e];
  }
  __syncthreads();
  for (int i = tid; i < GRU_RB * GRU_UB; i += 256) {
    const int r = i / GRU_UB, ul = i % GRU_UB;
    if (r < nr) {
      const int u = ub * GRU_UB + ul;
      const float ir = gbuf[0][r][ul] + gbuf[1][r][ul] + p.b_ih[u];
      const float iz = gbuf[0][r][GRU_UB + ul] + gbuf[1][r][GRU_UB + ul] + p.b_ih[DEC_D + u];
      const float in_ = gbuf[0][r][2 * GRU_UB + ul] + gbuf[1][r][2 * GRU_UB + ul] + p.b_ih[2 * DEC_D + u];
      const float hr = gbuf[2][r][ul] + p.b_hh[u];
      const float hz = gbuf[2][r][GRU_UB + ul] + p.b_hh[DEC_D + u];
      const float hn = gbuf[2][r][2 * GRU_UB + ul] + p.b_hh[2 * DEC_D + u];
      const float rg = sigmoidf_(ir + hr), zg = sigmoidf_(iz + hz);
      const float ng = tanhf(in_ + rg * hn);
      hnext[(long)(r0 + r) * DEC_D + u] = (1.f - zg) * ng + zg * hs[r][u];
    }
  }
}

extern "C" int64_t glass_decode_workspace_bytes(int R, int D) {
  return (int64_t)((size_t)R * D * 2 + (size_t)R * 2 * D) * sizeof(float) + (int64_t)R * sizeof(int);
}

extern "C" int glass_attention_decode(const float* x, const float* xproj, const glass_decoder_weights* w, const int* roi_image,
                                      int R, int num_images, int T, int D, int C, int max_len, int eos, float* out,
                                      int* pred_scratch, void* workspace, int64_t workspace_bytes, glass_stream_t stream) {
  GLASS_CHECK_ARG(D == DEC_D && T > 0 && T <= DEC_TMAX && C > 0 && C <= DEC_CMAX && max_len > 0,
                  "glass_attention_decode: needs D=256, T<=64, C<=256 (got D=%d T=%d C=%d)", D, T, C);
  if (R == 0) return GLASS_OK;
  GLASS_CHECK_ARG(x && xproj && w && roi_image && out && pred_scratch && num_images > 0 && workspace,
                  "glass_attention_decode: null pointer");
  GLASS_CHECK_ARG(workspace_bytes >= glass_decode_workspace_bytes(R, D), "glass_attention_decode: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  DecParams p;
  p.x = x; p.xproj = xproj; p.sW = w->sW; p.sB = w->sB; p.wW = w->wW; p.wB = w->wB; p.emb = w->emb; p.w_ih = w->w_ih;
  p.w_hh = w->w_hh; p.b_ih = w->b_ih; p.b_hh = w->b_hh; p.fcW = w->fcW; p.fcB = w->fcB; p.temperature = w->temperature;
  p.R = R; p.T = T; p.C = C; p.max_len = max_len; p.out = out; p.pred = pred_scratch; p.logits = nullptr;
  float* ws = static_cast<float*>(workspace);
  p.h0 = ws; p.h1 = ws + (size_t)R * D; p.inp = ws + (size_t)R * D * 2;
  p.yprev = reinterpret_cast<int*>(ws + (size_t)R * D * 4);
  hipError_t e = hipMemsetAsync(p.h0, 0, (size_t)R * D * sizeof(float), s);      // initial state h = 0
  if (e != hipSuccess) { glass_set_error("glass_attention_decode: memset: %s", hipGetErrorString(e)); return GLASS_EHIP; }
  const int rb = R >= 1024 ? 4 : R >= 512 ? 2 : 1;
  const dim3 ga(cdiv(R, rb)), gg(cdiv(R, GRU_RB), DEC_D / GRU_UB);
  float* hb[2] = {p.h0, p.h1};
  auto fc_att = [&](const float* hcur, int step, int do_att) {
    if (rb == 4) hipLaunchKernelGGL(dec_fc_att_kernel<4>, ga, dim3(256), 0, s, p, hcur, step, do_att);
    else if (rb == 2) hipLaunchKernelGGL(dec_fc_att_kernel<2>, ga, dim3(256), 0, s, p, hcur, step, do_att);
    else hipLaunchKernelGGL(dec_fc_att_kernel<1>, ga, dim3(256), 0, s, p, hcur, step, do_att);
  };
  fc_att(hb[0], -1, 1);                                                                       // attention for step 0
  for (int step = 0; step < max_len; ++step) {
    hipLaunchKernelGGL(dec_gru_kernel, gg, dim3(256), 0, s, p, hb[step & 1], hb[(step + 1) & 1]);
    fc_att(hb[(step + 1) & 1], step, step + 1 < max_len ? 1 : 0);
  }
  GLASS_CHECK_LAUNCH("glass_attention_decode");
  hipLaunchKernelGGL(decode_break_mask_kernel, dim3(num_images), dim3(256), 0, s, pred_scratch, roi_image, R, max_len, C, eos,
                     out);
  GLASS_CHECK_LAUNCH("glass_attention_decode(mask)");
  return GLASS_OK;
}

// One decoder step for a caller-driven search (AttentionRecognitionHead.beam_search, reference prediction_aster.py:133-134:
// `output, state, alpha = self.decoder(x, state, y_prev)`): attention with h_in and the embedding of y_prev, GRU cell, fc.
extern "C" int64_t glass_decode_step_workspace_bytes(int R, int D) {
  return (int64_t)((size_t)R * 2 * D) * sizeof(float) + (int64_t)R * 2 * sizeof(int);
}

extern "C" int glass_attention_decode_step(const float* x, const float* xproj, const glass_decoder_weights* w, int R, int T, int D,
                                           int C, const float* h_in, const int* y_prev, float* h_out, float* logits_out,
                                           float* probs_out, void* workspace, int64_t workspace_bytes, glass_stream_t stream) {
  GLASS_CHECK_ARG(D == DEC_D && T > 0 && T <= DEC_TMAX && C > 0 && C <= DEC_CMAX,
                  "glass_attention_decode_step: needs D=256, T<=64, C<=256 (got D=%d T=%d C=%d)", D, T, C);
  if (R == 0) return GLASS_OK;
  GLASS_CHECK_ARG(x && xproj && w && h_in && y_prev && h_out && logits_out && probs_out && workspace && h_in != h_out,
                  "glass_attention_decode_step: null pointer (or h_out aliases h_in)");
  GLASS_CHECK_ARG(workspace_bytes >= glass_decode_step_workspace_bytes(R, D), "glass_attention_decode_step: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  DecParams p;
  p.x = x; p.xproj = xproj; p.sW = w->sW; p.sB = w->sB; p.wW = w->wW; p.wB = w->wB; p.emb = w->emb; p.w_ih = w->w_ih;
  p.w_hh = w->w_hh; p.b_ih = w->b_ih; p.b_hh = w->b_hh; p.fcW = w->fcW; p.fcB = w->fcB; p.temperature = w->temperature;
  p.R = R; p.T = T; p.C = C; p.max_len = 1; p.out = probs_out; p.logits = nullptr;
  float* ws = static_cast<float*>(workspace);
  p.h0 = p.h1 = nullptr; p.inp = ws;
  int* iscratch = reinterpret_cast<int*>(ws + (size_t)R * 2 * D);
  p.pred = iscratch;
  const int rb = R >= 1024 ? 4 : R >= 512 ? 2 : 1;
  const dim3 ga(cdiv(R, rb)), gg(cdiv(R, GRU_RB), DEC_D / GRU_UB);
  auto fc_att = [&](const float* hcur, int step, int do_att) {
    if (rb == 4) hipLaunchKernelGGL(dec_fc_att_kernel<4>, ga, dim3(256), 0, s, p, hcur, step, do_att);
    else if (rb == 2) hipLaunchKernelGGL(dec_fc_att_kernel<2>, ga, dim3(256), 0, s, p, hcur, step, do_att);
    else hipLaunchKernelGGL(dec_fc_att_kernel<1>, ga, dim3(256), 0, s, p, hcur, step, do_att);
  };
  p.yprev = const_cast<int*>(y_prev);            // read only in the attention-only launch
  fc_att(h_in, -2, 1);                           // inp = [embedding(y_prev) | attention context of h_in]
  hipLaunchKernelGGL(dec_gru_kernel, gg, dim3(256), 0, s, p, h_in, h_out);
  p.yprev = iscratch + R;                        // the fc launch records its arg-max: into the scratch, not into the caller's y
  p.logits = logits_out;
  fc_att(h_out, 0, 0);                           // logits = fc(h_out) * temperature, probs = softmax(logits)
  GLASS_CHECK_LAUNCH("glass_attention_decode_step");
  return GLASS_OK;
}
