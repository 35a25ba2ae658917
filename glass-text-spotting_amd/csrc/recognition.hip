// Per-RoI recognition kernels: global-to-local fusion attention (softmax pooling + channel MLP),
// the BiLSTM recurrence and the greedy attention-GRU decoder.
//
// All three are independent across RoIs, so a workgroup owns a small group of RoIs for the
// whole sequence (persistent over the T / max_len dependent steps: no launch per step and no
// host sync), keeps the recurrent state in LDS, and streams the recurrent weights from L2 in
// a k-blocked layout (packed[k/4][row][4]) so that the 64 lanes of a wavefront read 1 KiB
// contiguous per load.  Reductions (softmax, LayerNorm, attention energies) are wavefront
// shuffle reductions (64 lanes) + one LDS hop across the 4 wavefronts.
#include "common.h"

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
  const float4 wm = *reinterpret_cast<const float4*>(w_mask + ((lane * 4) & 63));
  const float bm = b_mask[0];
  for (int pos = wave; pos < GC_HW; pos += 4) {
    const float4 a = *reinterpret_cast<const float4*>(x + (long)pos * GC_C + lane * 4);
    const float4 b = *reinterpret_cast<const float4*>(x + (long)pos * GC_C + 256 + lane * 4);
    float sa = a.x * wm.x + a.y * wm.y + a.z * wm.z + a.w * wm.w;
    float sb = b.x * wm.x + b.y * wm.y + b.z * wm.z + b.w * wm.w;
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) {
      sa += __shfl_xor(sa, off);
      sb += __shfl_xor(sb, off);
    }
    if ((lane & 15) == 0) {
      prob[lane >> 4][pos] = sa + bm;
      prob[4 + (lane >> 4)][pos] = sb + bm;
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
    for (int pos = 0; pos < GC_HW; ++pos) {
      c0 += x[(long)pos * GC_C + tid] * prob[h0][pos];
      c1 += x[(long)pos * GC_C + 256 + tid] * prob[h1][pos];
    }
    ctx[tid] = c0;
    ctx[256 + tid] = c1;
  }
  __syncthreads();
  // phase 4a: hid = W1 ctx + b1 (256 x 512): each wavefront takes rows wave, wave+4, ...
  for (int j = wave; j < GC_P; j += 4) {
    const float4 wa = *reinterpret_cast<const float4*>(w1 + (long)j * GC_C + lane * 4);
    const float4 wb = *reinterpret_cast<const float4*>(w1 + (long)j * GC_C + 256 + lane * 4);
    const float4 ca = *reinterpret_cast<const float4*>(ctx + lane * 4);
    const float4 cb = *reinterpret_cast<const float4*>(ctx + 256 + lane * 4);
    float s = wa.x * ca.x + wa.y * ca.y + wa.z * ca.z + wa.w * ca.w + wb.x * cb.x + wb.y * cb.y + wb.z * cb.z + wb.w * cb.w;
    s = wave_sum(s);
    if (lane == 0) hid[j] = s + b1[j];
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
  for (int c = wave; c < GC_C; c += 4) {
    const float4 wv = *reinterpret_cast<const float4*>(w2 + (long)c * GC_P + lane * 4);
    const float4 hv = *reinterpret_cast<const float4*>(hid + lane * 4);
    float s = wv.x * hv.x + wv.y * hv.y + wv.z * hv.z + wv.w * hv.w;
    s = wave_sum(s);
    if (lane == 0) ctx[c] = s + b2[c];
  }
  __syncthreads();
  // phase 5: x += t (broadcast over positions), float4 streaming
  {
    const int c4 = tid & 127;           // 128 float4 per position
    const float4 t = *reinterpret_cast<const float4*>(ctx + c4 * 4);
    for (int pos = tid >> 7; pos < GC_HW; pos += 2) {
      float4* px = reinterpret_cast<float4*>(x + (long)pos * GC_C) + c4;
      float4 v = *px;
      v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
      *px = v;
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
// grid = (ceil(R/RB), 2 directions); 256 threads, thread u owns hidden unit u: its i,f,g,o rows
// (u, 256+u, 512+u, 768+u of W_hh) and the cell state of every RoI of the group in registers.
constexpr int LSTM_HD = 256;
constexpr int LSTM_RB = 8;

__global__ __launch_bounds__(256) void bilstm_kernel(const float* __restrict__ xg, const float* __restrict__ whh,
                                                     float* __restrict__ out, int R, int T) {
  __shared__ __attribute__((aligned(16))) float h[LSTM_RB][LSTM_HD];
  const int u = threadIdx.x;
  const int dir = blockIdx.y;
  const int r0 = blockIdx.x * LSTM_RB;
  const int nr = min(LSTM_RB, R - r0);
  const float4* w = reinterpret_cast<const float4*>(whh) + (long)dir * (LSTM_HD / 4) * (4 * LSTM_HD);
  float c[LSTM_RB];
#pragma unroll
  for (int r = 0; r < LSTM_RB; ++r) { c[r] = 0.f; h[r][u] = 0.f; }
  __syncthreads();
  for (int step = 0; step < T; ++step) {
    const int t = dir == 0 ? step : T - 1 - step;
    float acc[4][LSTM_RB];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int r = 0; r < LSTM_RB; ++r)
        acc[g][r] = (r < nr) ? xg[(((long)(r0 + r) * T + t) * 2 + dir) * (4 * LSTM_HD) + g * LSTM_HD + u] : 0.f;
    stream_matvec<4, LSTM_RB, 4>(w, 4 * LSTM_HD, LSTM_HD, u, LSTM_HD / 4, &h[0][0], LSTM_HD, acc);
    __syncthreads();   // everyone has finished reading h of the previous step
#pragma unroll
    for (int r = 0; r < LSTM_RB; ++r) {
      const float ig = sigmoidf_(acc[0][r]), fg = sigmoidf_(acc[1][r]), gg = tanhf(acc[2][r]), og = sigmoidf_(acc[3][r]);
      c[r] = fg * c[r] + ig * gg;
      const float hn = og * tanhf(c[r]);
      h[r][u] = hn;
      if (r < nr) out[((long)(r0 + r) * T + t) * (2 * LSTM_HD) + dir * LSTM_HD + u] = hn;
    }
    __syncthreads();
  }
}

extern "C" int glass_bilstm_recurrence(const float* xg, const float* w_hh_packed, float* out, int R, int T, int Hd,
                                       glass_stream_t stream) {
  GLASS_CHECK_ARG(Hd == LSTM_HD, "glass_bilstm_recurrence: only Hd=256 is built (got %d)", Hd);
  if (R == 0) return GLASS_OK;
  GLASS_CHECK_ARG(xg && w_hh_packed && out && T > 0, "glass_bilstm_recurrence: bad args");
  hipLaunchKernelGGL(bilstm_kernel, dim3(cdiv(R, LSTM_RB), 2), dim3(256), 0, (hipStream_t)stream, xg, w_hh_packed, out, R, T);
  GLASS_CHECK_LAUNCH("glass_bilstm_recurrence");
  return GLASS_OK;
}

// ================================================================== attention GRU decoder
constexpr int DEC_D = 256;
constexpr int DEC_RB = 4;
constexpr int DEC_TMAX = 64;
constexpr int DEC_CMAX = 256;

struct DecParams {
  const float* x; const float* xproj;
  const float *sW, *sB, *wW, *wB, *emb, *w_ih, *w_hh, *b_ih, *b_hh, *fcW, *fcB;
  float temperature;
  int R, T, C, max_len;
  float* out;
  int* pred;
};

__global__ __launch_bounds__(256) void attn_decode_kernel(DecParams p) {
  __shared__ __attribute__((aligned(16))) float h[DEC_RB][DEC_D];        // GRU state
  __shared__ __attribute__((aligned(16))) float inp[DEC_RB][2 * DEC_D];  // [embedding | context]
  __shared__ float sproj[DEC_RB][DEC_D];
  __shared__ float energy[DEC_RB][DEC_TMAX];
  __shared__ float logit[DEC_RB][DEC_CMAX];
  __shared__ int yprev[DEC_RB];
  const int u = threadIdx.x, lane = u & 63, wave = u >> 6;
  const int r0 = blockIdx.x * DEC_RB;
  const int nr = min(DEC_RB, p.R - r0);
  const int T = p.T, C = p.C;
  const float4* sW4 = reinterpret_cast<const float4*>(p.sW);
  const float4* wih4 = reinterpret_cast<const float4*>(p.w_ih);
  const float4* whh4 = reinterpret_cast<const float4*>(p.w_hh);
  const float4* fc4 = reinterpret_cast<const float4*>(p.fcW);
#pragma unroll
  for (int r = 0; r < DEC_RB; ++r) h[r][u] = 0.f;
  if (u < DEC_RB) yprev[u] = 0;
  __syncthreads();

  for (int step = 0; step < p.max_len; ++step) {
    // ---- 1. sProj = sEmbed(h)
    {
      float acc[1][DEC_RB];
#pragma unroll
      for (int r = 0; r < DEC_RB; ++r) acc[0][r] = p.sB[u];
      stream_matvec<1, DEC_RB, 4>(sW4, DEC_D, 0, u, DEC_D / 4, &h[0][0], DEC_D, acc);
#pragma unroll
      for (int r = 0; r < DEC_RB; ++r) sproj[r][u] = acc[0][r];
    }
    __syncthreads();
    // ---- 2. energies e[r][t] = wEmbed(tanh(sProj + xProj[t])): wavefront per (r,t), lanes over D
    {
      const float4 ww = *reinterpret_cast<const float4*>(p.wW + lane * 4);
      const float wb = p.wB[0];
      for (int pr = wave; pr < nr * T; pr += 4) {
        const int r = pr / T, t = pr - r * T;
        const float4 xp = *reinterpret_cast<const float4*>(p.xproj + ((long)(r0 + r) * T + t) * DEC_D + lane * 4);
        const float4 sp = *reinterpret_cast<const float4*>(&sproj[r][lane * 4]);
        float s = ww.x * tanhf(sp.x + xp.x) + ww.y * tanhf(sp.y + xp.y) + ww.z * tanhf(sp.z + xp.z) + ww.w * tanhf(sp.w + xp.w);
        s = wave_sum(s);
        if (lane == 0) energy[r][t] = s + wb;
      }
    }
    __syncthreads();
    // ---- 3. softmax over t (one wavefront per RoI), 4. embedding lookup
    if (wave < nr) {
      const int r = wave;
      const float v = lane < T ? energy[r][lane] : -INFINITY;
      const float m = wave_max(v);
      const float e = lane < T ? expf(v - m) : 0.f;
      const float s = wave_sum(e);
      if (lane < T) energy[r][lane] = e / s;
    }
#pragma unroll
    for (int r = 0; r < DEC_RB; ++r) inp[r][u] = (r < nr) ? p.emb[(long)yprev[r] * DEC_D + u] : 0.f;
    __syncthreads();
    // ---- 4. context = alpha . x  (thread per feature)
#pragma unroll
    for (int r = 0; r < DEC_RB; ++r) {
      float s = 0.f;
      if (r < nr)
        for (int t = 0; t < T; ++t) s += energy[r][t] * p.x[((long)(r0 + r) * T + t) * DEC_D + u];
      inp[r][DEC_D + u] = s;
    }
    __syncthreads();
    // ---- 5. GRU cell: thread u owns hidden unit u (rows u, D+u, 2D+u = r,z,n gates)
    {
      float gi[3][DEC_RB], gh[3][DEC_RB];
#pragma unroll
      for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int r = 0; r < DEC_RB; ++r) { gi[g][r] = p.b_ih[g * DEC_D + u]; gh[g][r] = p.b_hh[g * DEC_D + u]; }
      stream_matvec<3, DEC_RB, 4>(wih4, 3 * DEC_D, DEC_D, u, 2 * DEC_D / 4, &inp[0][0], 2 * DEC_D, gi);
      stream_matvec<3, DEC_RB, 4>(whh4, 3 * DEC_D, DEC_D, u, DEC_D / 4, &h[0][0], DEC_D, gh);
      __syncthreads();   // all reads of the old state are done
#pragma unroll
      for (int r = 0; r < DEC_RB; ++r) {
        const float rg = sigmoidf_(gi[0][r] + gh[0][r]);
        const float zg = sigmoidf_(gi[1][r] + gh[1][r]);
        const float ng = tanhf(gi[2][r] + rg * gh[2][r]);
        h[r][u] = (1.f - zg) * ng + zg * h[r][u];
      }
    }
    __syncthreads();
    // ---- 6. logits = fc(h) * temperature
    if (u < C) {
      float acc[1][DEC_RB];
#pragma unroll
      for (int r = 0; r < DEC_RB; ++r) acc[0][r] = p.fcB[u];
      stream_matvec<1, DEC_RB, 4>(fc4, C, 0, u, DEC_D / 4, &h[0][0], DEC_D, acc);
#pragma unroll
      for (int r = 0; r < DEC_RB; ++r) logit[r][u] = acc[0][r] * p.temperature;
    }
    __syncthreads();
    // ---- 7. softmax over C classes + argmax (first maximum), one wavefront per RoI
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
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) {
        const float ob = __shfl_xor(best, off);
        const int oi = __shfl_xor(besti, off);
        if (ob > best || (ob == best && oi < besti)) { best = ob; besti = oi; }
      }
      if (lane == 0) {
        yprev[r] = besti;
        p.pred[(long)(r0 + r) * p.max_len + step] = besti;
      }
    }
    __syncthreads();
  }
}

// reference early break (prediction_aster.py:91-93): after step i, if every RoI of the call has
// emitted `eos` at least once the loop stops and later rows stay zero.  One workgroup per image.
__global__ void decode_break_mask_kernel(const int* __restrict__ pred, const int* __restrict__ roi_image, int R, int max_len,
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

extern "C" int glass_attention_decode(const float* x, const float* xproj, const glass_decoder_weights* w, const int* roi_image,
                                      int R, int num_images, int T, int D, int C, int max_len, int eos, float* out,
                                      int* pred_scratch, glass_stream_t stream) {
  GLASS_CHECK_ARG(D == DEC_D && T > 0 && T <= DEC_TMAX && C > 0 && C <= DEC_CMAX && max_len > 0,
                  "glass_attention_decode: needs D=256, T<=64, C<=256 (got D=%d T=%d C=%d)", D, T, C);
  if (R == 0) return GLASS_OK;
  GLASS_CHECK_ARG(x && xproj && w && roi_image && out && pred_scratch && num_images > 0, "glass_attention_decode: null pointer");
  DecParams p;
  p.x = x; p.xproj = xproj; p.sW = w->sW; p.sB = w->sB; p.wW = w->wW; p.wB = w->wB; p.emb = w->emb; p.w_ih = w->w_ih;
  p.w_hh = w->w_hh; p.b_ih = w->b_ih; p.b_hh = w->b_hh; p.fcW = w->fcW; p.fcB = w->fcB; p.temperature = w->temperature;
  p.R = R; p.T = T; p.C = C; p.max_len = max_len; p.out = out; p.pred = pred_scratch;
  hipLaunchKernelGGL(attn_decode_kernel, dim3(cdiv(R, DEC_RB)), dim3(256), 0, (hipStream_t)stream, p);
  GLASS_CHECK_LAUNCH("glass_attention_decode");
  hipLaunchKernelGGL(decode_break_mask_kernel, dim3(num_images), dim3(256), 0, (hipStream_t)stream, pred_scratch, roi_image, R,
                     max_len, C, eos, out);
  GLASS_CHECK_LAUNCH("glass_attention_decode(mask)");
  return GLASS_OK;
}
